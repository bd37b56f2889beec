#!/usr/bin/env python3
"""Host-side cost of one XbitOps.gemv / XbitOps.dequant call (Python checks + ctypes + launch)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import XbitOps  # noqa: E402
from xbitops_amd import _lib, ops, synth  # noqa: E402


def timeit(f, n=3000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


def main():
    dev = torch.device("cuda", 0)
    p = synth.make_problem(4096, 11008, 4, 128, m=1, seed=3)
    a, qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (p.act, p.qweight, p.scales, p.qzeros))
    lib = _lib.load()
    out = torch.empty((1, 11008), dtype=torch.float16, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    ptrs = (out.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr())
    print("XbitOps.gemv (full operator)            %6.2f us" % timeit(lambda: XbitOps.gemv(a, qw, sc, qz, 128, 4, 4096, 1)))
    plan = ops.GemvPlan(qw, sc, qz, 128, 4, 4096, 1)
    print("ops.GemvPlan call (weight checks hoisted)  %6.2f us" % timeit(lambda: plan(a)))
    print("  raw ctypes call, cached pointers      %6.2f us" % timeit(lambda: lib.xbit_gemv_f16(*ptrs, 4, 128, 1, 4096, 11008, 1, sp)))
    print("  argument checks only                  %6.2f us" % timeit(lambda: (ops._check_input(a, "input_a"), ops._check_common(qw, sc, qz, 128, 4, 4096))))
    print("  torch.empty output                    %6.2f us" % timeit(lambda: torch.empty((1, 11008), dtype=torch.float16, device=dev)))
    print("  5 x data_ptr()                        %6.2f us" % timeit(lambda: (out.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr())))
    print("  current stream handle                 %6.2f us" % timeit(lambda: ops._stream_ptr(0)))
    print("XbitOps.dequant (full operator)         %6.2f us" % timeit(lambda: XbitOps.dequant(qw, sc, qz, 128, 4, 4096, 1), 500))


if __name__ == "__main__":
    main()
