"""xbitops_amd — MI355X-native (gfx950) low-bit dequant + A16Wx GEMV behind the XbitOps operator surface.

``import XbitOps`` (repo root) gives the reference's module name; this package holds the
implementation: hand-written HIP kernels + a C-ABI shared library (``csrc/``, ``lib/``), the
ctypes binding (``_lib``), the two operators (``ops``), the host-side packer (``packing``), the
column-sharded multi-GPU form (``sharded``) and seeded synthetic inputs (``synth``).
"""
from .ops import dequant, gemv  # noqa: F401

__all__ = ["dequant", "gemv"]
