#!/usr/bin/env python3
"""Where should QuantLinear stop calling the fused kernel and dequantise + call the dense GEMM instead?  Per row count: the
fused path (xbit_gemv_f16 with M rows) against xbit_dequant_f16 + torch's fp16 matmul, both over rotating weight copies
(cold HBM), time per call from stream events around a queued run.
    python tools/linear_crossover.py [--shape K,N] [--bits 4] [--rows 8,16,32,48,64,96,128]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import XbitOps  # noqa: E402
from xbitops_amd import synth  # noqa: E402


def timed(fn, iters):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)  # ~10 ms: the host queues everything behind it
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096,11008")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--rows", default="8,16,32,48,64,96,128")
    args = ap.parse_args()
    k, n = (int(x) for x in args.shape.split(","))
    dev = torch.device("cuda", 0)
    base = synth.make_problem(k, n, args.bits, args.group, m=1, seed=1)
    copies = (512 << 20) // (base.qweight.nbytes) + 2
    qw = [(torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
    sc = [torch.from_numpy(base.scales).to(dev).clone() for _ in range(copies)]
    qz = [torch.from_numpy(base.qzeros).to(dev).clone() for _ in range(copies)]
    print("# %dx%d w%d g%d: us per call, cold weights (%d copies)" % (k, n, args.bits, args.group, copies))
    for m in (int(r) for r in args.rows.split(",")):
        x = (torch.randn((m, k), device=dev) * 0.1).to(torch.float16)

        def fused(i):
            c = i % copies
            XbitOps.gemv(x, qw[c], sc[c], qz[c], args.group, args.bits, k, 1)

        def dense(i):
            c = i % copies
            w = XbitOps.dequant(qw[c], sc[c], qz[c], args.group, args.bits, k, 1)
            torch.matmul(x, w)

        tf, td = timed(fused, 60), timed(dense, 60)
        print("rows %4d   fused %7.2f us   dequant + GEMM %7.2f us   %s" % (m, tf, td, "fused" if tf < td else "dense"), flush=True)


if __name__ == "__main__":
    main()
