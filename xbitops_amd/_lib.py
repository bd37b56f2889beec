"""ctypes binding of libxbitops_hip.so — the C ABI declared in include/xbitops_hip.h.

This is the only place the native library is loaded.  There is no fallback of any kind: if the
library is missing or does not export the ABI, importing the ops raises.  (The reference binds
its launchers with pybind11 inside a torch C++ extension, /root/reference/src/dq_torch_ops.cc:
80-85; here the boundary is a plain C ABI so that any host language can bind it — see
INTEGRATION.md.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libxbitops_hip.so")
ABI_VERSION = 1

# every symbol include/xbitops_hip.h declares
EXPORTS = (
    "xbit_abi_version",
    "xbit_status_string",
    "xbit_dequant_f16",
    "xbit_gemv_f16",
    "xbit_gemv_multi_f16",
    "xbit_profile_enable",
    "xbit_profile_last_ms",
)

_lib = None


class XbitLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the native library.  Raises XbitLibraryError if it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XbitLibraryError(
            "xbitops: native library %s not found. Build it with `python -m xbitops_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise XbitLibraryError("xbitops: cannot load %s: %s" % (LIB_PATH, e)) from e
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise XbitLibraryError("xbitops: %s does not export %s" % (LIB_PATH, ", ".join(missing)))

    lib.xbit_abi_version.restype = ctypes.c_int
    lib.xbit_abi_version.argtypes = []
    lib.xbit_status_string.restype = ctypes.c_char_p
    lib.xbit_status_string.argtypes = [ctypes.c_int]
    vp, u32, u8, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint8, ctypes.c_int
    lib.xbit_dequant_f16.restype = ci
    lib.xbit_dequant_f16.argtypes = [vp, vp, vp, vp, ci, ci, u32, u32, u8, vp]
    lib.xbit_gemv_f16.restype = ci
    lib.xbit_gemv_f16.argtypes = [vp, vp, vp, vp, vp, ci, ci, u32, u32, u32, u8, vp]
    lib.xbit_gemv_multi_f16.restype = ci
    lib.xbit_gemv_multi_f16.argtypes = [ci, ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                        ctypes.POINTER(u32), ci, ci, u32, u32, u8, vp]
    lib.xbit_profile_enable.restype = ci
    lib.xbit_profile_enable.argtypes = [ci]
    lib.xbit_profile_last_ms.restype = ci
    lib.xbit_profile_last_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
    if lib.xbit_abi_version() != ABI_VERSION:
        raise XbitLibraryError("xbitops: ABI version mismatch (library %d, binding %d); rebuild" % (lib.xbit_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def status_string(status: int) -> str:
    return load().xbit_status_string(int(status)).decode("utf-8", "replace")


def check(status: int, what: str) -> None:
    """Turn a non-zero status into RuntimeError (the reference aborts the process instead)."""
    if status != 0:
        raise RuntimeError("%s failed: %s (status %d)" % (what, status_string(status), status))
