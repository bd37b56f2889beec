#!/usr/bin/env python3
"""Context for the roofline fraction: what the vendor library's own fp16 GEMV (torch -> hipBLASLt /
rocBLAS) achieves when it streams the SAME number of bytes cold from HBM, timed per launch with
events around each call.  fp16 weights, batch 1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(k, n, label):
    dev = torch.device("cuda", 0)
    nbytes = k * n * 2
    copies = max(2, (600 << 20) // nbytes + 1)
    ws = [torch.randn(n, k, device=dev, dtype=torch.float16) * 0.02 for _ in range(copies)]  # [out, in] as nn.Linear stores it
    x = torch.randn(1, k, device=dev, dtype=torch.float16)
    for i in range(20):
        torch.nn.functional.linear(x, ws[i % copies])
    torch.cuda.synchronize()
    ts = []
    for i in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.nn.functional.linear(x, ws[i % copies])
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    avg = sum(ts) / len(ts)
    print("%-44s %7.1f MB  avg %6.2f us  med %6.2f  min %6.2f   %6.0f GB/s = %4.1f %% of 8 TB/s" % (
        label, nbytes / 1e6, avg, ts[len(ts) // 2], ts[0], nbytes / avg * 1e-3, nbytes / avg * 1e-3 / 80))


def run_xbit(k, n, label):
    import XbitOps
    from xbitops_amd import synth

    dev = torch.device("cuda", 0)
    nbytes = synth.algorithmic_bytes(k, n, 4, 128, 1)
    copies = max(2, (600 << 20) // nbytes + 1)
    p = synth.make_problem(k, n, 4, 128, m=1, seed=1)
    a, sc, qz = (torch.from_numpy(t).to(dev) for t in (p.act, p.scales, p.qzeros))
    qw0 = torch.from_numpy(p.qweight).to(dev)
    qws = [(qw0 ^ (0x01010101 * (i % 127))).contiguous() for i in range(copies)]
    for i in range(20):
        XbitOps.gemv(a, qws[i % copies], sc, qz, 128, 4, k, 1)
    torch.cuda.synchronize()
    ts = []
    for i in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        XbitOps.gemv(a, qws[i % copies], sc, qz, 128, 4, k, 1)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    avg = sum(ts) / len(ts)
    print("%-44s %7.1f MB  avg %6.2f us  med %6.2f  min %6.2f   %6.0f GB/s = %4.1f %% of 8 TB/s" % (
        label, nbytes / 1e6, avg, ts[len(ts) // 2], ts[0], nbytes / avg * 1e-3, nbytes / avg * 1e-3 / 80))


def main():
    print("same timer (torch events recorded around each call), cold weights, batch 1")
    run_xbit(4096, 11008, "XbitOps.gemv w4 4096x11008")
    run_xbit(8192, 28672, "XbitOps.gemv w4 8192x28672")
    print("vendor fp16 GEMV (torch.nn.functional.linear -> hipBLASLt / rocBLAS):")
    run(4096, 2752, "fp16 4096x2752  (= bytes of w4 4096x11008)")
    run(4096, 11008, "fp16 4096x11008")
    run(8192, 7168, "fp16 8192x7168  (= bytes of w4 8192x28672)")
    run(8192, 28672, "fp16 8192x28672")


if __name__ == "__main__":
    main()
