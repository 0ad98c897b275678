"""splintr_amd -- MI355X-native batch BPE encoder behind the `splintr.Tokenizer` surface.

`Tokenizer.encode_batch()` dispatches through a C ABI (include/splintr_hip.h) into hand-written
gfx950 kernels.  See DESIGN.md for the path and INTEGRATION.md for the drop-in story.
The module exports what `splintr/__init__.py` of the reference exports (python/splintr/__init__.py:110-141).
"""
from .agent_tokens import *  # noqa: F401,F403
from .agent_tokens import __all__ as _agent_all
from .streaming import ByteLevelStreamingDecoder, StreamingDecoder
from .tokenizer import CL100K_BASE_PATTERN, LLAMA3_PATTERN, MISTRAL_V3_PATTERN, O200K_BASE_PATTERN, Tokenizer

__all__ = ["Tokenizer", "StreamingDecoder", "ByteLevelStreamingDecoder", "CL100K_BASE_PATTERN", "O200K_BASE_PATTERN",
           "LLAMA3_PATTERN", "MISTRAL_V3_PATTERN"] + list(_agent_all)
__version__ = "0.2.0"
