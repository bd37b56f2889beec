#!/usr/bin/env python3
"""One activation row: the staged-operand kernel (xbit_gemv_dyn.h) against the kernels of xbit_gemv_mfma.h on decoder-layer
shapes, every bit-width.  Tuning library; per-launch kernel durations (hipExtLaunchKernel events, cold HBM), median us.
Variants: 399 = never the staged-operand kernel (round-2 launch rules), 300 = forced with the automatic tile, 301 = 16-column
tiles (8 lanes x 8 B), 302 / 303 / 304 = tiles of 32 / 48 / 64 columns, -1 = what the production rule picks.
    python tools/sweep_dyn_vs_mfma.py [--bits 2,3,4,5,6,7,8] [--shapes 4096x11008,...] [--variants 399,300,...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from xbitops_amd import _lib, synth  # noqa: E402
import gemv_sweep  # noqa: E402

SHAPES = ["4096x4096", "4096x11008", "11008x4096", "4096x14336", "14336x4096", "5120x5120", "5120x13824", "13824x5120", "8192x8192",
          "8192x3584", "8192x1024", "8192x28672", "4096x32000"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", default="2,3,4,5,6,7,8")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--variants", default="399,300,301,302,303,304,-1")
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--rows", default="1", help="activation rows per launch (the staged-operand kernel takes 1..4)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load_tuning()
    variants = [int(v) for v in args.variants.split(",")]
    print("# us per launch, cold, group %d: %s; columns: variant %s" % (
        args.group, "period of a dependent chain of launches" if os.environ.get("XBIT_SWEEP_PERIOD") else "median kernel duration inside a chain (per-launch events)",
        " ".join(str(v) for v in variants)), flush=True)
    for shape in args.shapes.split(","):
        k, n = (int(x) for x in shape.split("x"))
        for bits, m in ((int(b), int(r)) for b in args.bits.split(",") for r in args.rows.split(",")):
            cells = []
            for v in variants:
                try:
                    nbytes, _avg, med, _mn = gemv_sweep.gemv_case(lib, dev, k, n, bits, args.group, m, True, 80 if k * n > 5e7 else 120, v, 0)
                    cells.append("%7.2f" % med)
                except AssertionError:
                    cells.append("    n/a")
            print("w%d %-12s m%d %6.1f MB  %s" % (bits, shape, m, synth.algorithmic_bytes(k, n, bits, args.group, m) / 1e6, " ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
