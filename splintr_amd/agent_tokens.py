"""Agent-token constant classes: `CL100K_AGENT_TOKENS.SYSTEM` etc.

The reference generates one frozen class of integer class attributes per vocabulary
(src/python/agent_tokens_generated.rs, registered in src/lib.rs); here the same NAME -> id tables
are data (splintr_amd/data/agent_tokens.json, tools/pack_agent_tokens.py) and the classes are built
from it.  Instances cannot be created and attributes cannot be rebound, as with `frozen` pyclasses."""
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "agent_tokens.json")


class _Frozen(type):
    def __setattr__(cls, key, value):
        raise AttributeError(f"{cls.__name__} is frozen")

    def __delattr__(cls, key):
        raise AttributeError(f"{cls.__name__} is frozen")

    def __call__(cls, *a, **k):
        raise TypeError(f"cannot create '{cls.__name__}' instances")


def _build():
    with open(_PATH) as f:
        tables = json.load(f)
    return {name: _Frozen(name, (), dict(consts, __doc__=f"Agent token ids ({min(consts.values())}-{max(consts.values())})",
                                         __slots__=()))
            for name, consts in tables.items()}


globals().update(_build())
__all__ = [k for k in list(globals()) if k.endswith("_AGENT_TOKENS")]
