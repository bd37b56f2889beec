"""Deterministic synthetic inputs for the dequant / GEMV path (SURVEY.md §8d "Synthetic inputs").

A counter-based splitmix64 stream makes every tensor a pure function of ``(seed, tag, index)``,
so the same problem can be regenerated on the GPU box, in tests and in ``oracle/gen_golden.py``
without shipping data.  The reference's own golden ``.bin`` files are author-local
(/root/reference/src/main.cc:14) and absent from its tree.

* ``qweight`` / ``qzeros``: uniform random 32-bit words — every bit pattern is a legal packing,
  so q and z are uniform on ``[0, 2**bits)``;
* ``scales``: fp16 of ``U(0.002, 0.01) * 16 / 2**bits`` (keeps \\|W\\| <~ 0.15);
* activation: fp16 of ``U(-1, 1)``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .packing import packed_rows

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(index: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 output for counter values ``index`` (uint64 array) under ``seed``."""
    with np.errstate(over="ignore"):
        z = (index.astype(np.uint64) + np.uint64(1)) * _GOLDEN + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(count: int, seed: int, tag: int) -> np.ndarray:
    return splitmix64(np.arange(count, dtype=np.uint64), (seed * 0x100000001B3 + tag * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)


def random_words(count: int, seed: int, tag: int) -> np.ndarray:
    """``count`` uniform uint32 words."""
    return (_stream(count, seed, tag) >> np.uint64(32)).astype(np.uint32)


def random_unit(count: int, seed: int, tag: int) -> np.ndarray:
    """``count`` float64 in [0, 1) with 53 random bits."""
    return (_stream(count, seed, tag) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


@dataclass
class Problem:
    """One synthetic call of the ops (numpy, host memory)."""

    k: int
    n: int
    bits: int
    groupsize: int
    m: int
    qweight: np.ndarray  # int32 [ceil(K*b/32), N]
    scales: np.ndarray  # float16 [G, N]
    qzeros: np.ndarray  # int32 [G, ceil(N*b/32)]
    act: np.ndarray  # float16 [M, K]

    @property
    def groups(self) -> int:
        return (self.k + self.groupsize - 1) // self.groupsize


def make_problem(k: int, n: int, bits: int, groupsize: int = 128, m: int = 1, seed: int = 0) -> Problem:
    kw = packed_rows(k, bits)
    groups = (k + groupsize - 1) // groupsize
    zw = packed_rows(n, bits)
    qweight = random_words(kw * n, seed, 1).reshape(kw, n).view(np.int32)
    qzeros = random_words(groups * zw, seed, 2).reshape(groups, zw).view(np.int32)
    scales = ((0.002 + 0.008 * random_unit(groups * n, seed, 3)) * (16.0 / (1 << bits))).astype(np.float16).reshape(groups, n)
    act = (2.0 * random_unit(m * k, seed, 4) - 1.0).astype(np.float16).reshape(m, k)
    return Problem(k, n, bits, groupsize, m, qweight, scales, qzeros, act)


def algorithmic_bytes(k: int, n: int, bits: int, groupsize: int, m: int = 1) -> int:
    """Bytes one GEMV call must move."""
    groups = (k + groupsize - 1) // groupsize
    return 4 * packed_rows(k, bits) * n + 2 * groups * n + 4 * groups * packed_rows(n, bits) + 2 * m * k + 2 * m * n


def dequant_bytes(k: int, n: int, bits: int, groupsize: int) -> int:
    """Bytes one dequant call must move: packed inputs read + fp16 W[K,N] written."""
    groups = (k + groupsize - 1) // groupsize
    return 4 * packed_rows(k, bits) * n + 2 * groups * n + 4 * groups * packed_rows(n, bits) + 2 * k * n
