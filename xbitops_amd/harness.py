"""Reader / writer / replay for the reference's stand-alone harness data set.

The reference benchmark executable loads seven files from a directory
(/root/reference/src/main.cc:14-80; the author's data is not in its repository):

    shape.bin    text  "K,N,bits,group"        (main.cc:18-19; any single-character separators)
    input.bin    fp16  [1, K]                  (main.cc:47-50)
    qweight.bin  int32 [ceil(K*bits/32), N]    (:52-56)
    scales.bin   fp16  [ceil(K/group), N]      (:58-62)
    qzeros.bin   int32 [ceil(K/group), ceil(N*bits/32)]   (:64-68)
    weight.bin   fp16  [K, N]   golden dequantised weight  (:70-74)
    out.bin      fp16  [1, N]   golden GEMV output         (:76-80)

``replay`` runs XbitOps.dequant / XbitOps.gemv on such a set and applies the reference's own
acceptance tests: dequant |diff| <= 1e-3 (unpack_weight_2_to_7.cu:570); GEMV error iff relative
> 1e-3 AND absolute > 4e-4 (gemv_w4a16.cu:406-407).  So a layer dumped by the quantisation
pipeline the reference was developed against can be replayed against this library unchanged.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass

import numpy as np

from .packing import packed_rows

FILES = ("shape.bin", "input.bin", "qweight.bin", "scales.bin", "qzeros.bin", "weight.bin", "out.bin")


@dataclass
class Sample:
    k: int
    n: int
    bits: int
    groupsize: int
    input: np.ndarray  # fp16 [1, K]
    qweight: np.ndarray  # int32 [kw, N]
    scales: np.ndarray  # fp16 [G, N]
    qzeros: np.ndarray  # int32 [G, zw]
    weight: np.ndarray  # fp16 [K, N] golden
    out: np.ndarray  # fp16 [1, N] golden


def save_sample(directory: str, s: Sample) -> None:
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "shape.bin"), "w") as f:
        f.write("%d,%d,%d,%d" % (s.k, s.n, s.bits, s.groupsize))
    for name, arr, dt in (("input", s.input, np.float16), ("qweight", s.qweight, np.int32), ("scales", s.scales, np.float16),
                          ("qzeros", s.qzeros, np.int32), ("weight", s.weight, np.float16), ("out", s.out, np.float16)):
        np.ascontiguousarray(arr, dtype=dt).tofile(os.path.join(directory, name + ".bin"))


def load_sample(directory: str) -> Sample:
    with open(os.path.join(directory, "shape.bin")) as f:
        nums = [int(x) for x in re.findall(r"-?\d+", f.read())]
    if len(nums) < 4:
        raise ValueError("shape.bin must hold K,N,bits,group")
    k, n, bits, g = nums[:4]
    kw, groups, zw = packed_rows(k, bits), (k + g - 1) // g, packed_rows(n, bits)

    def rd(name, dt, shape):
        a = np.fromfile(os.path.join(directory, name + ".bin"), dtype=dt)
        need = int(np.prod(shape))
        if a.size < need:
            raise ValueError("%s.bin holds %d elements, %s needs %d" % (name, a.size, shape, need))
        return a[:need].reshape(shape)

    return Sample(k, n, bits, g, rd("input", np.float16, (1, k)), rd("qweight", np.int32, (kw, n)), rd("scales", np.float16, (groups, n)),
                  rd("qzeros", np.int32, (groups, zw)), rd("weight", np.float16, (k, n)), rd("out", np.float16, (1, n)))


def dequant_errors(w: np.ndarray, golden: np.ndarray) -> int:
    """Number of elements the reference harness would flag (unpack_weight_2_to_7.cu:562-575)."""
    return int((np.abs(w.astype(np.float32) - golden.astype(np.float32)) > 1e-3).sum())


def gemv_errors(y: np.ndarray, golden: np.ndarray) -> int:
    """Number of elements the reference harness would flag (gemv_w4a16.cu:396-412)."""
    v1, ref = y.astype(np.float32).ravel(), golden.astype(np.float32).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = (np.abs(ref / v1) > 1.001) | (np.abs(v1 / ref) > 1.001)  # abs() as the reference: a sign flip is an error
    return int((rel & (np.abs(ref - v1) > 4e-4)).sum())


def replay(directory: str, add_zero_bias: int = 0, device: str = "cuda:0", iters: int = 10):
    """Run both ops on a harness data set; returns a dict with error counts and kernel milliseconds."""
    import torch

    from . import ops

    s = load_sample(directory)
    dev = torch.device(device)
    x, qw, sc, qz = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (s.input, s.qweight, s.scales, s.qzeros))
    w = ops.dequant(qw, sc, qz, s.groupsize, s.bits, s.k, add_zero_bias)
    y = ops.gemv(x, qw, sc, qz, s.groupsize, s.bits, s.k, add_zero_bias)
    torch.cuda.synchronize(dev)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):  # the reference times 10 launches between two events (unpack_weight_2_to_7.cu:538-549)
        ops.gemv(x, qw, sc, qz, s.groupsize, s.bits, s.k, add_zero_bias)
    t1.record()
    torch.cuda.synchronize(dev)
    return {
        "k": s.k, "n": s.n, "bits": s.bits, "groupsize": s.groupsize,
        "dequant_errors": dequant_errors(w.cpu().numpy(), s.weight), "gemv_errors": gemv_errors(y.cpu().numpy(), s.out),
        "gemv_ms": t0.elapsed_time(t1) / iters,
    }
