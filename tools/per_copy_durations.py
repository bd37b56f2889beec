#!/usr/bin/env python3
"""Does the kernel's duration depend on WHERE the weights sit?  Rotates over distinct weight copies (cold HBM) and
reports the mean / spread of the kernel-only duration per copy next to the copy's address.
    python tools/per_copy_durations.py [--shape K,N,bits,g,M] [--rounds 30]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096,11008,4,128,1")
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--align", type=int, default=0, help="allocate each copy at this byte alignment (0 = torch's allocator)")
    args = ap.parse_args()
    k, n, bits, g, m = (int(x) for x in args.shape.split(","))
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    nbytes = synth.algorithmic_bytes(k, n, bits, g, m)
    copies = (512 << 20) // nbytes + 2
    base = synth.make_problem(k, n, bits, g, m=m, seed=1)
    act = torch.from_numpy(base.act).to(dev)
    host = torch.from_numpy(base.qweight)
    qw = []
    keep = []
    for c in range(copies):
        if args.align:
            raw = torch.empty(host.numel() * 4 + args.align, dtype=torch.uint8, device=dev)
            off = (-raw.data_ptr()) % args.align
            t = raw[off:off + host.numel() * 4].view(torch.int32).view(host.shape)
            keep.append(raw)
        else:
            t = torch.empty_like(host, device=dev)
        t.copy_(host)
        qw.append(t)
    sc = torch.from_numpy(base.scales).to(dev)
    qz = torch.from_numpy(base.qzeros).to(dev)
    out = torch.empty((m, n), dtype=torch.float16, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    ms = ctypes.c_float()
    per = [[] for _ in range(copies)]
    for r in range(args.rounds + 2):
        if r == 2:
            lib.xbit_profile_enable(1)
        for c in range(copies):
            st = lib.xbit_gemv_f16(out.data_ptr(), act.data_ptr(), qw[c].data_ptr(), sc.data_ptr(), qz.data_ptr(), bits, g, m, k, n, 1, sp)
            assert st == 0
            if r >= 2:
                _lib.check(lib.xbit_profile_last_ms(ctypes.byref(ms)), "profile")
                per[c].append(ms.value * 1e3)
    lib.xbit_profile_enable(0)
    allv = np.array(per)
    print("shape %s, %d copies x %d rounds, align %d: all launches mean %.2f median %.2f us" % (args.shape, copies, args.rounds, args.align, allv.mean(), np.median(allv)))
    print("per-copy medians: min %.2f max %.2f; std of per-copy medians %.3f us, mean within-copy std %.3f us" % (
        np.median(allv, axis=1).min(), np.median(allv, axis=1).max(), np.median(allv, axis=1).std(), allv.std(axis=1).mean()))
    for c in range(copies):
        a = qw[c].data_ptr()
        print("  copy %2d  addr 0x%012x  (mod 2 MiB 0x%06x, mod 4 KiB 0x%03x)  median %.2f  mean %.2f  min %.2f" % (
            c, a, a % (2 << 20), a % 4096, np.median(allv[c]), allv[c].mean(), allv[c].min()))


if __name__ == "__main__":
    main()
