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
TUNING_LIB_PATH = os.path.join(_HERE, "lib", "libxbitops_hip_tuning.so")  # measurement build: tools/ and two A/B tests only
ABI_VERSION = 2

# every symbol include/xbitops_hip.h declares
EXPORTS = (
    "xbit_abi_version",
    "xbit_status_string",
    "xbit_dequant_f16",
    "xbit_gemv_f16",
    "xbit_gemv_f16_ex",
    "xbit_gemv_workspace_bytes",
    "xbit_workspace_reset",
    "xbit_workspace_release",
    "xbit_workspace_capture_bytes_free",
    "xbit_gemv_gidx_f16",
    "xbit_dequant_gidx_f16",
    "xbit_gemv_multi_f16",
    "xbit_profile_enable",
    "xbit_profile_last_ms",
    "xbit_profile_series_begin",
    "xbit_profile_series_end",
    "xbit_dry_run_begin",
    "xbit_dry_run_end",
)

_lib = None
_tuning = None


class XbitLibraryError(RuntimeError):
    pass


class LaunchRecord(ctypes.Structure):
    """``xbit_launch_record`` of include/xbitops_hip.h."""

    _fields_ = [("kernel_offset", ctypes.c_uint64), ("grid", ctypes.c_uint32 * 3), ("block", ctypes.c_uint32 * 3),
                ("dynamic_lds_bytes", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class GemvOptions(ctypes.Structure):
    """``xbit_gemv_options`` of include/xbitops_hip.h."""

    _fields_ = [("struct_bytes", ctypes.c_uint32), ("split_k", ctypes.c_int32), ("workspace", ctypes.c_void_p),
                ("workspace_bytes", ctypes.c_uint64), ("variant", ctypes.c_int32), ("reserved", ctypes.c_int32), ("trace", ctypes.c_void_p)]

    def __init__(self, split_k: int = 0, workspace: int = 0, workspace_bytes: int = 0, variant: int = -1, trace: int = 0):
        super().__init__(ctypes.sizeof(GemvOptions), split_k, workspace or None, workspace_bytes, variant, 0, trace or None)


def _typed(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise XbitLibraryError(
            "xbitops: native library %s not found. Build it with `python -m xbitops_amd.build%s` "
            "(hipcc, gfx950). There is no CPU fallback." % (path, " --tuning" if path == TUNING_LIB_PATH else "")
        )
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # missing libamdhip64 etc.
        raise XbitLibraryError("xbitops: cannot load %s: %s" % (path, e)) from e
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise XbitLibraryError("xbitops: %s does not export %s" % (path, ", ".join(missing)))

    lib.xbit_abi_version.restype = ctypes.c_int
    lib.xbit_abi_version.argtypes = []
    lib.xbit_status_string.restype = ctypes.c_char_p
    lib.xbit_status_string.argtypes = [ctypes.c_int]
    vp, u32, u8, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint8, ctypes.c_int
    lib.xbit_dequant_f16.restype = ci
    lib.xbit_dequant_f16.argtypes = [vp, vp, vp, vp, ci, ci, u32, u32, u8, vp]
    lib.xbit_gemv_f16.restype = ci
    lib.xbit_gemv_f16.argtypes = [vp, vp, vp, vp, vp, ci, ci, u32, u32, u32, u8, vp]
    lib.xbit_gemv_f16_ex.restype = ci
    lib.xbit_gemv_f16_ex.argtypes = [vp, vp, vp, vp, vp, ci, ci, u32, u32, u32, u8, vp, ctypes.POINTER(GemvOptions)]
    lib.xbit_gemv_workspace_bytes.restype = ctypes.c_uint64
    lib.xbit_gemv_workspace_bytes.argtypes = [u32, u32]
    lib.xbit_workspace_reset.restype = ci
    lib.xbit_workspace_reset.argtypes = []
    lib.xbit_workspace_release.restype = ci
    lib.xbit_workspace_release.argtypes = []
    lib.xbit_workspace_capture_bytes_free.restype = ctypes.c_uint64
    lib.xbit_workspace_capture_bytes_free.argtypes = []
    lib.xbit_gemv_gidx_f16.restype = ci
    lib.xbit_gemv_gidx_f16.argtypes = [vp, vp, vp, vp, vp, vp, ci, u32, u32, u32, u32, u8, vp]
    lib.xbit_dequant_gidx_f16.restype = ci
    lib.xbit_dequant_gidx_f16.argtypes = [vp, vp, vp, vp, vp, ci, u32, u32, u32, u8, vp]
    lib.xbit_gemv_multi_f16.restype = ci
    lib.xbit_gemv_multi_f16.argtypes = [ci, ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                        ctypes.POINTER(u32), ci, ci, u32, u32, u8, vp]
    lib.xbit_profile_enable.restype = ci
    lib.xbit_profile_enable.argtypes = [ci]
    lib.xbit_profile_last_ms.restype = ci
    lib.xbit_profile_last_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
    lib.xbit_profile_series_begin.restype = ci
    lib.xbit_profile_series_begin.argtypes = [ci]
    lib.xbit_profile_series_end.restype = ci
    lib.xbit_profile_series_end.argtypes = [ctypes.POINTER(ctypes.c_float), ci, ctypes.POINTER(ci)]
    lib.xbit_dry_run_begin.restype = ci
    lib.xbit_dry_run_begin.argtypes = []
    lib.xbit_dry_run_end.restype = ci
    lib.xbit_dry_run_end.argtypes = [ctypes.POINTER(LaunchRecord), ci, ctypes.POINTER(ci)]
    if lib.xbit_abi_version() != ABI_VERSION:
        raise XbitLibraryError("xbitops: ABI version mismatch (library %d, binding %d); rebuild" % (lib.xbit_abi_version(), ABI_VERSION))
    return lib


def load() -> ctypes.CDLL:
    """Load (once) and type the native library.  Raises XbitLibraryError if it is unusable."""
    global _lib
    if _lib is None:
        _lib = _typed(LIB_PATH)
    return _lib


def load_tuning() -> ctypes.CDLL:
    """The measurement build (same ABI plus ``options.variant`` / ``options.trace``).  Never used by the
    operators; ``tools/`` and the A/B tests call its C ABI directly."""
    global _tuning
    if _tuning is None:
        _tuning = _typed(TUNING_LIB_PATH)
    return _tuning


def status_string(status: int) -> str:
    return load().xbit_status_string(int(status)).decode("utf-8", "replace")


def check(status: int, what: str) -> None:
    """Turn a non-zero status into RuntimeError (the reference aborts the process instead)."""
    if status != 0:
        raise RuntimeError("%s failed: %s (status %d)" % (what, status_string(status), status))


_symbols = {}


def kernel_names(path: str = LIB_PATH):
    """{offset of a kernel's handle inside the library: demangled name} (``nm -C``; cached per library file)."""
    import subprocess

    if path not in _symbols:
        out = subprocess.run(["nm", "-C", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        table = {}
        for line in out.splitlines():
            parts = line.split(" ", 2)
            # the address a launch passes is the kernel's HANDLE: an object symbol that carries the kernel's own name
            # (-fhip-new-launch-api); the host stub is a function symbol next to it
            if len(parts) == 3 and parts[1] in "vVdDbB" and "_kernel" in parts[2] and "__device_stub__" not in parts[2]:
                table[int(parts[0], 16)] = parts[2]
        _symbols[path] = table
    return _symbols[path]


def dry_run(lib: ctypes.CDLL, call, path: str = None):
    """Run ``call()`` (anything that calls into ``lib`` on this thread) with launches recorded instead of issued; returns
    a list of dicts {kernel, grid, block, lds}.  No GPU needed: device pointers are never followed."""
    if path is None:
        path = TUNING_LIB_PATH if lib is _tuning else LIB_PATH
    check(lib.xbit_dry_run_begin(), "xbit_dry_run_begin")
    try:
        call()
    finally:
        buf, cnt = (LaunchRecord * 64)(), ctypes.c_int()
        st = lib.xbit_dry_run_end(buf, 64, ctypes.byref(cnt))
    check(st, "xbit_dry_run_end")
    names = kernel_names(path)
    return [{"kernel": names.get(buf[i].kernel_offset, "?@%x" % buf[i].kernel_offset), "grid": tuple(buf[i].grid), "block": tuple(buf[i].block),
             "lds": int(buf[i].dynamic_lds_bytes)} for i in range(cnt.value)]
