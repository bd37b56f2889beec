#!/usr/bin/env python3
"""How far is the automatic split-K plan from the best of an exhaustive sweep on shapes NO table entry and no earlier sweep
covers?  Few-column, long-K layers; per shape: the automatic launch, the unsplit launch, and split factors 2..12 on 32- and
64-column tiles of both kernels (202 / 205: xbit_gemv_mfma.h, 332 / 334: xbit_gemv_dyn.h).  Chain period, cold weights.
    python tools/sweep_split_unseen.py [--bits 4]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["XBIT_SWEEP_PERIOD"] = "1"
from xbitops_amd import _lib  # noqa: E402
import gemv_sweep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", default="4")
    ap.add_argument("--ks", default="10240,12288,16384,20480")
    ap.add_argument("--ns", default="2048,3072,4096,5120")
    ap.add_argument("--verbose", action="store_true", help="also print every (variant, split factor) time")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib, tun = _lib.load(), _lib.load_tuning()
    print("# us per launch (chain period, cold): automatic plan, unsplit, best of the sweep (variant x split), automatic / best")
    worst = 0.0
    for bits in (int(b) for b in args.bits.split(",")):
        for k in (int(x) for x in args.ks.split(",")):
            for n in (int(x) for x in args.ns.split(",")):
                auto = gemv_sweep.gemv_case(lib, dev, k, n, bits, 128, 1, True, 60, -1, 0)[2]
                unsplit = gemv_sweep.gemv_case(lib, dev, k, n, bits, 128, 1, True, 60, -1, 1)[2]
                best, how = unsplit, "unsplit"
                fill = max(2, 256 // ((n + 63) // 64))  # 64-column tiles x as many K ranges as fill the chip once
                t_fill = None
                table = {}
                for v in (202, 205, 332, 334):
                    for sp in sorted(set((2, 3, 4, 5, 6, 7, 8, 10, 12, fill))):
                        if sp > 15:
                            continue
                        try:
                            t = gemv_sweep.gemv_case(tun, dev, k, n, bits, 128, 1, True, 60, v, sp)[2]
                        except AssertionError:
                            continue
                        if v == 334 and sp == fill:
                            t_fill = t
                        table.setdefault(v, []).append("%d:%.2f" % (sp, t))
                        if t < best:
                            best, how = t, "variant %d split %d" % (v, sp)
                auto = min(auto, gemv_sweep.gemv_case(lib, dev, k, n, bits, 128, 1, True, 60, -1, 0)[2])  # once more at the end: the first case of a shape reads a few per cent high
                worst = max(worst, auto / best)
                print("w%d %6dx%-5d  auto %6.2f  unsplit %6.2f  best %6.2f (%s)  auto/best %.3f   fill-once (334 x %d) %s" % (
                    bits, k, n, auto, unsplit, best, how, auto / best, fill, "%6.2f = %.3f of best" % (t_fill, t_fill / best) if t_fill else "n/a"), flush=True)
                if args.verbose:
                    for v in sorted(table):
                        print("      variant %d  %s" % (v, "  ".join(table[v])), flush=True)
    print("# worst automatic / best: %.3f" % worst)


if __name__ == "__main__":
    main()
