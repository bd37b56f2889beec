"""oracle — CPU checkers for the XbitOps dequant / A16Wx GEMV path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker / the timed CPU baseline.  The product
(``xbitops_amd``, ``XbitOps``) never imports it and has no CPU fallback.

Three independent statements of the same algorithm live here:

* ``xbit_oracle.c``  -> ``_build/libxbit_oracle.so`` — this repo's C restatement (two rounding
  modes: ``RNE`` = what the reference's CUDA kernels compute on hardware, ``SIM`` = the
  reference simulator's round-half-up soft-float);
* ``np_oracle.py``   — a numpy bit-stream restatement (independent unpack, RNE only);
* ``ref_driver.cc``  -> ``_ref/libxbit_ref.so`` — the REFERENCE's own simulator
  (/root/reference/src/cpp_simulate.cc compiled in place).  Present only where it was built
  (this container) or shipped prebuilt (the GPU box); ``ref_available()`` says which.

Parity status: pinned — ``tests/test_oracle_pinning.py`` checks C(SIM) == reference simulator
bit-for-bit (live when ``_ref`` is present, and always against ``tests/golden/*.npz``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

RNE = 0
SIM = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libxbit_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libxbit_ref.so")

_lib = None
_ref = None


def build(quiet: bool = True) -> None:
    """(Re)build the checkers with oracle/Makefile (gcc/g++ only; no GPU toolchain)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL if quiet else None)


def _u16p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))


def _u32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "xbit_oracle.c")
        path = os.environ.get("XBIT_ORACLE_LIB")  # another build of the same source (tests/test_sanitizers.py: the ASan + UBSan one)
        if not path:
            path = _LIB_PATH
            if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
                build()
        _lib = ctypes.CDLL(path)
        _lib.xo_h2f.restype = ctypes.c_float
        _lib.xo_h2f.argtypes = [ctypes.c_uint16]
        _lib.xo_f2h_sim.restype = ctypes.c_uint16
        _lib.xo_f2h_sim.argtypes = [ctypes.c_float]
        _lib.xo_d2h_rne.restype = ctypes.c_uint16
        _lib.xo_d2h_rne.argtypes = [ctypes.c_double]
        _lib.xo_hmul.restype = ctypes.c_uint16
        _lib.xo_hmul.argtypes = [ctypes.c_uint16, ctypes.c_uint16, ctypes.c_int]
        _lib.xo_hfma_negc.restype = ctypes.c_uint16
        _lib.xo_hfma_negc.argtypes = [ctypes.c_uint16, ctypes.c_uint16, ctypes.c_uint16, ctypes.c_int]
        _lib.xo_num_threads.restype = ctypes.c_int
    return _lib


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref() -> ctypes.CDLL:
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(_REF_PATH)
        _ref.ref_float_to_half.restype = ctypes.c_uint16
        _ref.ref_float_to_half.argtypes = [ctypes.c_float]
        _ref.ref_half_to_float.restype = ctypes.c_float
        _ref.ref_half_to_float.argtypes = [ctypes.c_uint16]
    return _ref


def _prep(qweight, scales, qzeros):
    qw = np.ascontiguousarray(qweight).view(np.uint32)
    sc = np.ascontiguousarray(scales, dtype=np.float16).view(np.uint16)
    qz = np.ascontiguousarray(qzeros).view(np.uint32)
    return qw, sc, qz


def unpack_q(qweight: np.ndarray, bits: int, k: int) -> np.ndarray:
    """Integer weights q[K, N] (uint8)."""
    qw = np.ascontiguousarray(qweight).view(np.uint32)
    n = qw.shape[1]
    out = np.empty((k, n), np.uint8)
    lib().xo_unpack_q(_u8p(out), _u32p(qw), ctypes.c_int(bits), ctypes.c_uint32(k), ctypes.c_uint32(n))
    return out


def unpack_z(qzeros: np.ndarray, bits: int, n: int) -> np.ndarray:
    """Integer zero-points z[G, N] (uint8), before add_zero_bias."""
    qz = np.ascontiguousarray(qzeros).view(np.uint32)
    groups = qz.shape[0]
    out = np.empty((groups, n), np.uint8)
    lib().xo_unpack_z(_u8p(out), _u32p(qz), ctypes.c_int(bits), ctypes.c_uint32(groups), ctypes.c_uint32(n))
    return out


def dequant(qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, add_zero_bias: int = 0, mode: int = RNE) -> np.ndarray:
    """W[K, N] float16 — same argument order as the reference op (dq_torch_ops.cc:23-24)."""
    qw, sc, qz = _prep(qweight, scales, qzeros)
    n = qw.shape[1]
    out = np.empty((in_features, n), np.uint16)
    lib().xo_dequant(_u16p(out), _u32p(qw), _u16p(sc), _u32p(qz), ctypes.c_int(bits), ctypes.c_int(groupsize),
                     ctypes.c_uint32(in_features), ctypes.c_uint32(n), ctypes.c_int(add_zero_bias), ctypes.c_int(mode))
    return out.view(np.float16)


def gemv(act, qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, add_zero_bias: int = 0, mode: int = RNE) -> np.ndarray:
    """y[M, N] float16 = fp16(sum_k a[m,k] * W[k,n]) with W = dequant(mode), double accumulation."""
    qw, sc, qz = _prep(qweight, scales, qzeros)
    a = np.ascontiguousarray(act, dtype=np.float16).reshape(-1, in_features).view(np.uint16)
    n = qw.shape[1]
    out = np.empty((a.shape[0], n), np.uint16)
    lib().xo_gemv(_u16p(out), _u16p(a), _u32p(qw), _u16p(sc), _u32p(qz), ctypes.c_int(bits), ctypes.c_int(groupsize),
                  ctypes.c_uint32(a.shape[0]), ctypes.c_uint32(in_features), ctypes.c_uint32(n), ctypes.c_int(add_zero_bias),
                  ctypes.c_int(mode))
    return out.view(np.float16)


def _i32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def dequant_gidx(qweight, scales, qzeros, g_idx, bits: int, in_features: int, add_zero_bias: int = 0, mode: int = RNE) -> np.ndarray:
    """W[K, N] float16 with the group of row k given by ``g_idx[k]`` (act-order; quant_cuda_kernel.cu:257-336)."""
    qw, sc, qz = _prep(qweight, scales, qzeros)
    gi = np.ascontiguousarray(g_idx, dtype=np.int32)
    assert gi.shape == (in_features,) and gi.min() >= 0 and gi.max() < sc.shape[0]
    n = qw.shape[1]
    out = np.empty((in_features, n), np.uint16)
    lib().xo_dequant_gidx(_u16p(out), _u32p(qw), _u16p(sc), _u32p(qz), _i32p(gi), ctypes.c_int(bits), ctypes.c_uint32(in_features),
                          ctypes.c_uint32(n), ctypes.c_int(add_zero_bias), ctypes.c_int(mode))
    return out.view(np.float16)


def gemv_gidx(act, qweight, scales, qzeros, g_idx, bits: int, in_features: int, add_zero_bias: int = 0, mode: int = RNE) -> np.ndarray:
    """y[M, N] float16 = fp16(sum_k a[m,k] * W[k,n]) with W = dequant_gidx(mode), double accumulation."""
    qw, sc, qz = _prep(qweight, scales, qzeros)
    gi = np.ascontiguousarray(g_idx, dtype=np.int32)
    assert gi.shape == (in_features,) and gi.min() >= 0 and gi.max() < sc.shape[0]
    a = np.ascontiguousarray(act, dtype=np.float16).reshape(-1, in_features).view(np.uint16)
    n = qw.shape[1]
    out = np.empty((a.shape[0], n), np.uint16)
    lib().xo_gemv_gidx(_u16p(out), _u16p(a), _u32p(qw), _u16p(sc), _u32p(qz), _i32p(gi), ctypes.c_int(bits), ctypes.c_uint32(a.shape[0]),
                       ctypes.c_uint32(in_features), ctypes.c_uint32(n), ctypes.c_int(add_zero_bias), ctypes.c_int(mode))
    return out.view(np.float16)


def gemv_w4_transposed(act, qweight_t, scales_t, qzeros_t, groupsize: int, in_features: int, add_zero_bias: int = 0, mode: int = RNE) -> np.ndarray:
    """y[M, N] for the transposed 4-bit layout of the reference's gemv_NT (gemv_w4a16_nt.cu:64-159):
    qweight_t [N, ceil(K/8)], scales_t [N, G], qzeros_t [N/8, G]."""
    qw = np.ascontiguousarray(qweight_t).view(np.uint32)
    sc = np.ascontiguousarray(scales_t, dtype=np.float16).view(np.uint16)
    qz = np.ascontiguousarray(qzeros_t).view(np.uint32)
    n = qw.shape[0]
    a = np.ascontiguousarray(act, dtype=np.float16).reshape(-1, in_features).view(np.uint16)
    out = np.empty((a.shape[0], n), np.uint16)
    lib().xo_gemv_w4_transposed(_u16p(out), _u16p(a), _u32p(qw), _u16p(sc), _u32p(qz), ctypes.c_int(groupsize), ctypes.c_uint32(a.shape[0]),
                                ctypes.c_uint32(in_features), ctypes.c_uint32(n), ctypes.c_int(add_zero_bias), ctypes.c_int(mode))
    return out.view(np.float16)


def matvec_f16(act, w) -> np.ndarray:
    """y[M, N] float16 = fp16(a @ W) for a dequantised fp16 W[K, N] (double accumulation)."""
    wv = np.ascontiguousarray(w, dtype=np.float16).view(np.uint16)
    k, n = wv.shape
    a = np.ascontiguousarray(act, dtype=np.float16).reshape(-1, k).view(np.uint16)
    out = np.empty((a.shape[0], n), np.uint16)
    lib().xo_matvec_f16(_u16p(out), _u16p(a), _u16p(wv), ctypes.c_uint32(a.shape[0]), ctypes.c_uint32(k), ctypes.c_uint32(n))
    return out.view(np.float16)


def gemv_w4_cuda_order(act, qweight, scales, qzeros, groupsize: int, in_features: int, add_zero_bias: int = 0) -> np.ndarray:
    """Model of the rounding order of the reference CUDA kernel (gemv_w4a16_pt.cu:67-143), 4-bit."""
    qw, sc, qz = _prep(qweight, scales, qzeros)
    a = np.ascontiguousarray(act, dtype=np.float16).reshape(-1, in_features).view(np.uint16)
    n = qw.shape[1]
    out = np.empty((a.shape[0], n), np.uint16)
    lib().xo_gemv_w4_cuda_order(_u16p(out), _u16p(a), _u32p(qw), _u16p(sc), _u32p(qz), ctypes.c_int(groupsize),
                                ctypes.c_uint32(a.shape[0]), ctypes.c_uint32(in_features), ctypes.c_uint32(n),
                                ctypes.c_int(add_zero_bias))
    return out.view(np.float16)


def ref_dequant(qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, variant: int = 0) -> np.ndarray:
    """W[K, N] float16 from the REFERENCE simulator (no add_zero_bias there). Needs ``_ref``."""
    qw, sc, qz = _prep(qweight, scales, qzeros)
    n = qw.shape[1]
    out = np.zeros((in_features, n), np.uint16)
    rc = ref().ref_dequant(_u16p(out), _u32p(qw), _u16p(sc), _u32p(qz), ctypes.c_int(bits), ctypes.c_int(groupsize),
                           ctypes.c_int(in_features), ctypes.c_int(n), ctypes.c_int(variant))
    if rc != 0:
        raise ValueError("reference simulator does not support bits=%d variant=%d n=%d" % (bits, variant, n))
    return out.view(np.float16)


def num_threads() -> int:
    return int(lib().xo_num_threads())


def set_num_threads(t: int) -> None:
    lib().xo_set_num_threads(ctypes.c_int(t))
