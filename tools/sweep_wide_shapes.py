#!/usr/bin/env python3
"""Round 4: wide layers (the 70B FFN up / gate projection and its relatives) - which launch uses the whole chip?
Per (shape, launch): median and min kernel duration inside a chain (per-launch events) and the chain period, cold HBM.
Launches: -1 production rule; 302 / 303 / 304 staged-operand kernel, whole K, tiles of 32 / 48 / 64 columns;
334:s / 332:s staged-operand kernel, 64- / 32-column tiles x s K ranges (split-K); 107 128-column tiles (round-2 kernel).
    python tools/sweep_wide_shapes.py [--shapes 8192x28672,...] [--launches -1,304,334:4,...] [--bits 4] [--rows 1]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="8192x28672,4096x28672,4096x32000,8192x22016")
    ap.add_argument("--launches", default="-1,302,303,304,334:2,334:3,334:4,334:6,334:8,332:2,332:4")
    ap.add_argument("--bits", default="4")
    ap.add_argument("--rows", default="1")
    ap.add_argument("--group", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    prod, tun = _lib.load(), _lib.load_tuning()
    print("# device: %s; us per launch, cold: events median / events min / chain period" % torch.cuda.get_device_name(0), flush=True)
    for shape in args.shapes.split(","):
        k, n = (int(x) for x in shape.split("x"))
        for bits in (int(b) for b in args.bits.split(",")):
            for m in (int(r) for r in args.rows.split(",")):
                nbytes = synth.algorithmic_bytes(k, n, bits, args.group, m)
                for launch in args.launches.split(","):
                    v, _, sp = launch.partition(":")
                    v, sp = int(v), int(sp or 0)
                    try:
                        _nb, _avg, med, mn, per = gemv_sweep.gemv_case(tun if v > 0 else prod, dev, k, n, bits, args.group, m, True,
                                                                       80 if k * n > 5e7 else 150, v, sp, both=True)
                        print("w%d %-12s m%-2d %6.1f MB  launch %-7s events med %7.2f min %7.2f  period %7.2f us   %5.3f of 8 TB/s (med)" % (
                            bits, shape, m, nbytes / 1e6, launch, med, mn, per, nbytes / (med * 1e-6) / 8e12), flush=True)
                    except AssertionError as e:
                        print("w%d %-12s m%-2d launch %-7s n/a (%s)" % (bits, shape, m, launch, e), flush=True)


if __name__ == "__main__":
    main()
