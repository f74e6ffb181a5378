"""`benchmark.evaluator` -> tools/bbc_scoring.py (see ../README.md)."""
import os
import sys

_TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "tools")
if _TOOLS not in sys.path:
    sys.path.insert(0, _TOOLS)
import bbc_scoring as _m  # noqa: E402

globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
