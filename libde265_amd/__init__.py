"""libde265_amd -- MI355X (gfx950) HEVC pixel-reconstruction back end.

Holds only what the hot path needs: csrc/ (hand-written HIP kernels + the C ABI
of include/de265_hip.h) and a thin ctypes mirror of that ABI (backend.py).
"""
from . import _abi  # noqa: F401

__all__ = ["_abi", "backend", "build"]
