"""numpy restatement of the packed layout and of the dequant arithmetic.  TEST INFRASTRUCTURE ONLY.

Independent of ``xbit_oracle.c``: the unpack treats each column (or zero-point row) as one
LSB-first bit-stream and slices it with 64-bit windows, instead of the per-field word/shift
arithmetic of the reference kernels (/root/reference/src/cu/unpack_weight_2_to_7.cu:56-59,
196-217, 258-281).  Two statements agreeing with each other and with the reference simulator
is the pin (tests/test_oracle_pinning.py).
"""
from __future__ import annotations

import numpy as np


def unpack_stream(words: np.ndarray, bits: int, count: int) -> np.ndarray:
    """``words`` uint32 ``[W, C]`` -> fields ``[count, C]`` (uint8); stream runs down axis 0."""
    w = np.ascontiguousarray(words).view(np.uint32).astype(np.uint64)
    nwords, cols = w.shape
    ext = np.concatenate([w, np.zeros((1, cols), np.uint64)], axis=0)
    win = ext[:-1] | (ext[1:] << np.uint64(32))  # 64-bit window starting at every word
    bitpos = np.arange(count, dtype=np.int64) * bits
    wi = bitpos >> 5
    sh = (bitpos & 31).astype(np.uint64)
    mask = np.uint64((1 << bits) - 1)
    return ((win[wi] >> sh[:, None]) & mask).astype(np.uint8)


def unpack_q(qweight: np.ndarray, bits: int, k: int) -> np.ndarray:
    return unpack_stream(qweight, bits, k)


def unpack_z(qzeros: np.ndarray, bits: int, n: int) -> np.ndarray:
    return np.ascontiguousarray(unpack_stream(np.ascontiguousarray(np.ascontiguousarray(qzeros).view(np.uint32).T), bits, n).T)


def dequant_rne(qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, add_zero_bias: int = 0) -> np.ndarray:
    """W[K,N] float16 with IEEE fp16 arithmetic: zs = rne16((z+bias)*s); W = rne16(q*s - zs).

    Products and the sum are exact in float64 (<= 8+11 and 11 significant bits), and numpy's
    float64->float16 cast rounds to nearest-even once, so this equals a hardware hmul + hfma
    (unpack_weight_2_to_7.cu:60,75 / :285,305).
    """
    q = unpack_q(qweight, bits, in_features).astype(np.float64)
    n = q.shape[1]
    z = unpack_z(qzeros, bits, n).astype(np.float64) + float(add_zero_bias)
    s = np.asarray(scales, np.float16).astype(np.float64)
    zs = (z * s).astype(np.float16).astype(np.float64)
    gi = np.arange(in_features) // groupsize
    with np.errstate(over="ignore", invalid="ignore"):
        return (q * s[gi] - zs[gi]).astype(np.float16)


def gemv_rne(act, qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, add_zero_bias: int = 0) -> np.ndarray:
    """y = fp16( a @ W ) with float64 accumulation of the fp16 W above."""
    w = dequant_rne(qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias).astype(np.float64)
    a = np.asarray(act, np.float16).reshape(-1, in_features).astype(np.float64)
    return (a @ w).astype(np.float16)
