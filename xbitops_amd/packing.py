"""Host-side packer for the XbitOps wire format (numpy, no device code).

The reference repository ships only the *consumers* of the packed tensors; the producer lives in
an external quantisation pipeline.  This module is the inverse of the
layouts the reference kernels index (all citations under /root/reference):

* ``qweight``  int32 ``[ceil(K*b/32), N]`` — per column, the K values form one LSB-first
  contiguous bit-stream running down the rows (``src/cu/unpack_weight_2_to_7.cu:64,72``
  for 2/4/8 bit, ``:196-217,228-239`` for 3/5/6/7 bit, ``src/cu/gemv_w4a16_pt.cu:73,90-94``).
* ``qzeros``   int32 ``[G, ceil(N*b/32)]`` — per group, the N zero-points form one LSB-first
  bit-stream along the row (``unpack_weight_2_to_7.cu:56-59,258-281``).
* ``scales``   fp16 ``[G, N]``.

``add_zero_bias`` is the GPTQ "stored zero-point minus one" convention
(``unpack_weight_2_to_7.cu:60,285``): pass ``zeros - 1`` here and ``add_zero_bias=1`` to the ops.
"""
from __future__ import annotations

import numpy as np

__all__ = ["packed_rows", "pack_stream", "unpack_stream", "pack_qweight", "pack_qzeros", "quantize_rtn", "sort_rows_by_group",
           "from_transposed_w4"]


def packed_rows(count: int, bits: int) -> int:
    """Number of int32 words that hold ``count`` fields of ``bits`` bits (dq_torch_ops.cc:31)."""
    return (count * bits + 31) // 32


def pack_stream(values: np.ndarray, bits: int) -> np.ndarray:
    """Pack ``values[L, C]`` (uint, < 2**bits) into ``uint32[ceil(L*b/32), C]``.

    Field ``i`` of column ``c`` occupies stream bits ``[i*b, (i+1)*b)``; stream bit ``t`` is bit
    ``t % 32`` of word ``t // 32``.
    """
    if not 1 <= bits <= 8:
        raise ValueError("bits must be in 1..8")
    v = np.ascontiguousarray(values).astype(np.uint64)
    if v.ndim != 2:
        raise ValueError("values must be 2-D [length, columns]")
    if v.size and int(v.max()) >> bits:
        raise ValueError("value does not fit in %d bits" % bits)
    length, cols = v.shape
    nblk = (length + 31) // 32  # 32 fields == exactly `bits` words
    pad = nblk * 32 - length
    if pad:
        v = np.concatenate([v, np.zeros((pad, cols), np.uint64)], axis=0)
    v = v.reshape(nblk, 32, cols)
    # one spare word per block absorbs the (always zero) spill of the last field
    words = np.zeros((nblk, bits + 1, cols), np.uint64)
    for i in range(32):
        w, sh = divmod(i * bits, 32)
        f = v[:, i, :] << np.uint64(sh)
        words[:, w, :] |= f & np.uint64(0xFFFFFFFF)
        words[:, w + 1, :] |= f >> np.uint64(32)
    out = words[:, :bits, :].reshape(nblk * bits, cols)[: packed_rows(length, bits)]
    return np.ascontiguousarray(out.astype(np.uint32))


def pack_qweight(q: np.ndarray, bits: int) -> np.ndarray:
    """``q[K, N]`` integer weights -> ``qweight`` int32 ``[ceil(K*b/32), N]``."""
    return pack_stream(q, bits).view(np.int32)


def pack_qzeros(z: np.ndarray, bits: int) -> np.ndarray:
    """``z[G, N]`` integer zero-points -> ``qzeros`` int32 ``[G, ceil(N*b/32)]``."""
    return np.ascontiguousarray(pack_stream(np.ascontiguousarray(z.T), bits).T).view(np.int32)


def quantize_rtn(w: np.ndarray, bits: int, groupsize: int, add_zero_bias: int = 0):
    """Asymmetric round-to-nearest group quantisation of ``w[K, N]`` (float) -> packed tensors.

    Returns ``(qweight, scales, qzeros)`` such that the ops reconstruct
    ``(q - z) * s`` with ``z = stored + add_zero_bias``.
    """
    w = np.asarray(w, np.float32)
    k, n = w.shape
    groups = (k + groupsize - 1) // groupsize
    maxq = (1 << bits) - 1
    q = np.zeros((k, n), np.uint8)
    scales = np.zeros((groups, n), np.float16)
    zeros = np.zeros((groups, n), np.int64)
    for g in range(groups):
        blk = w[g * groupsize : (g + 1) * groupsize]
        lo = np.minimum(blk.min(axis=0), 0.0)
        hi = np.maximum(blk.max(axis=0), 0.0)
        s = ((hi - lo) / maxq).astype(np.float16)
        s = np.where(s == 0, np.float16(1.0), s)
        sf = s.astype(np.float32)
        z = np.clip(np.rint(-lo / sf), add_zero_bias, maxq).astype(np.int64)
        q[g * groupsize : (g + 1) * groupsize] = np.clip(np.rint(blk / sf) + z, 0, maxq).astype(np.uint8)
        scales[g] = s
        zeros[g] = z - add_zero_bias
    return pack_qweight(q, bits), scales, pack_qzeros(zeros.astype(np.uint8), bits)


def unpack_stream(words: np.ndarray, bits: int, length: int) -> np.ndarray:
    """Inverse of :func:`pack_stream`: ``uint32[ceil(L*b/32), C]`` -> ``uint8[L, C]``."""
    w = np.ascontiguousarray(words).view(np.uint32).astype(np.uint64)
    rows, cols = w.shape
    w = np.concatenate([w, np.zeros((1, cols), np.uint64)], axis=0)  # spare word for the last field's (empty) spill
    i = np.arange(length, dtype=np.int64) * bits
    wi, sh = i >> 5, (i & 31).astype(np.uint64)
    pair = w[wi] | (w[wi + 1] << np.uint64(32))
    return ((pair >> sh[:, None]) & np.uint64((1 << bits) - 1)).astype(np.uint8)


def sort_rows_by_group(qweight: np.ndarray, g_idx: np.ndarray, bits: int, groupsize: int):
    """Act-order checkpoint -> the contiguous-group layout the ordinary kernels take.

    ``g_idx[k]`` is the group of row ``k`` (GPTQ ``desc_act``).  When every group holds exactly
    ``groupsize`` rows (the last one may hold fewer), sorting the rows by group makes
    ``g = k' // groupsize`` true again: returns ``(qweight_sorted, perm)`` with
    ``W_sorted[k'] = W[perm[k']]``, so ``x @ W == x[..., perm] @ W_sorted`` (a sum over k in another
    order).  Raises ``ValueError`` for any other ``g_idx`` — use the general ``*_gidx`` entry points then.
    """
    gi = np.ascontiguousarray(g_idx).astype(np.int64)
    k = gi.shape[0]
    groups = (k + groupsize - 1) // groupsize
    if gi.min() < 0 or gi.max() >= groups:
        raise ValueError("g_idx values must lie in [0, %d)" % groups)
    counts = np.bincount(gi, minlength=groups)
    want = np.full(groups, groupsize, np.int64)
    want[-1] = k - (groups - 1) * groupsize
    if not np.array_equal(counts, want):
        raise ValueError("groups are not all of size %d: the rows cannot be sorted into the contiguous-group layout" % groupsize)
    perm = np.argsort(gi, kind="stable")
    q = unpack_stream(np.ascontiguousarray(qweight).view(np.uint32), bits, k)
    return pack_qweight(q[perm], bits), perm.astype(np.int64)


def from_transposed_w4(qweight_t: np.ndarray, scales_t: np.ndarray, qzeros_t: np.ndarray):
    """The transposed 4-bit layout of the reference's experimental ``gemv_NT``
    (/root/reference/src/cu/gemv_w4a16_nt.cu:64-159: ``qweight_t [N, K/8]``, ``scales_t [N, G]``,
    ``qzeros_t [N/8, G]``) -> the registered ops' layout.  With 8 nibbles per word on both axes the
    conversion is three plain transposes: word ``(n, k/8)`` holds k..k+7 of column n exactly as word
    ``(k/8, n)`` of ``qweight`` does (``:104`` vs gemv_w4a16_pt.cu:73), and word ``(n/8, g)`` holds the
    zero-points of columns n..n+7 exactly as word ``(g, n/8)`` of ``qzeros`` (``:87-89`` vs
    gemv_w4a16_pt.cu:53-58)."""
    return (np.ascontiguousarray(np.asarray(qweight_t).T), np.ascontiguousarray(np.asarray(scales_t).T),
            np.ascontiguousarray(np.asarray(qzeros_t).T))
