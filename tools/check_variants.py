#!/usr/bin/env python3
"""Correctness of tuning variants against the default GEMV path (GPU): python tools/check_variants.py 12,13,21"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth  # noqa: E402


def main():
    variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,13,15,21,22,28,39,105,51,53,55,56,64").split(",")]
    dev = torch.device("cuda", 0)
    lib = _lib.load_tuning()  # measurement build: python -m xbitops_amd.build --tuning
    sp = torch.cuda.current_stream(dev).cuda_stream
    for (k, n, bits, g, m) in [(4096, 11008, 4, 128, 1), (4096, 4096, 4, 128, 1), (4096, 1000, 4, 128, 1), (2048, 512, 4, 128, 1)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=5)
        a, qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (p.act, p.qweight, p.scales, p.qzeros))

        def run(v):
            out = torch.zeros((m, n), dtype=torch.float16, device=dev)
            st = lib.xbit_gemv_f16_ex(out.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), bits, g, m, k, n, 1, sp,
                                      _lib.GemvOptions(variant=v))
            torch.cuda.synchronize()
            return st, out

        _, ref = run(-1)
        for v in variants:
            st, out = run(v)
            diff = (out.float() - ref.float()).abs().max().item() if st == 0 else float("nan")
            ulp1 = ((out.view(torch.int16).int() - ref.view(torch.int16).int()).abs() <= 1).float().mean().item() if st == 0 else 0
            print("k=%d n=%d variant %d: status %d  max|diff| %.3e  within-1-ulp %.5f" % (k, n, v, st, diff, ulp1), flush=True)


if __name__ == "__main__":
    main()
