"""splintr_amd -- MI355X-native batch BPE encoder behind the `splintr.Tokenizer` surface.

`Tokenizer.encode_batch()` dispatches through a C ABI (include/splintr_hip.h) into hand-written
gfx950 kernels.  See DESIGN.md for the path and INTEGRATION.md for the drop-in story.
"""
from .tokenizer import CL100K_BASE_PATTERN, LLAMA3_PATTERN, O200K_BASE_PATTERN, Tokenizer

__all__ = ["Tokenizer", "CL100K_BASE_PATTERN", "O200K_BASE_PATTERN", "LLAMA3_PATTERN"]
__version__ = "0.1.0"
