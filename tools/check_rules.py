#!/usr/bin/env python3
"""Do the launch rules (xbitops_amd/csrc/xbit_gemv_rules.h) still pick the fastest launch?  GPU.

For every point of a held-out list of (bits, rows, K, N, group) - shapes no sweep behind a rule was run on - the automatic launch of the
PRODUCTION library is timed next to every forced launch the measurement library offers for that point (the staged-operand kernel on each
tile width, the round-2 kernels with and without their split plans, 128-column tiles, forced split counts).  Clock: the period of a
dependent chain of launches, cold HBM - what a decoder pays per layer.  The check FAILS (exit 1) when the automatic launch takes more than
`--slack` (default 1.08) times the best forced one on any point; a kernel or rule edit that silently invalidates a threshold shows up here.

    python tools/check_rules.py [--slack 1.08] [--points all|quick] > profiles/rNN/check_rules.txt
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

# (bits, rows, K, N, group): none of these shapes appears in a sweep under profiles/r01..r03 that a rule cites
POINTS = [
    # one row, many tiles / few tiles / wide
    (4, 1, 3072, 9216, 128), (4, 1, 6656, 6656, 128), (4, 1, 8192, 16384, 128), (4, 1, 4096, 22016, 128), (4, 1, 6144, 16384, 128), (4, 1, 5120, 27648, 128),
    (4, 1, 4096, 1536, 128), (4, 1, 6144, 2048, 128), (4, 1, 2048, 8192, 128), (3, 1, 6656, 6656, 128), (3, 1, 3072, 9216, 128), (3, 1, 8192, 2560, 128),
    (2, 1, 6144, 16384, 128), (2, 1, 4096, 6144, 128), (5, 1, 3072, 9216, 128), (5, 1, 6144, 4096, 64), (6, 1, 6656, 6656, 128), (7, 1, 4096, 6144, 64),
    (8, 1, 6144, 16384, 128), (8, 1, 4096, 2048, 128), (8, 1, 4096, 22016, 128),
    # long K, few columns: split plans
    (4, 1, 10240, 2560, 128), (4, 1, 22016, 4096, 128), (4, 1, 16384, 3072, 128), (4, 1, 24576, 1536, 128), (3, 1, 10240, 5120, 128), (5, 1, 16384, 3072, 128),
    (8, 1, 10240, 2560, 128), (2, 1, 22016, 4096, 128), (4, 1, 16384, 7168, 128),
    # group sizes 64 / 32
    (4, 1, 6656, 6656, 64), (3, 1, 6144, 16384, 64), (4, 1, 3072, 9216, 32), (8, 1, 6144, 4096, 32),
    # several rows
    (4, 2, 3072, 9216, 128), (4, 4, 6144, 16384, 128), (8, 4, 6656, 6656, 128), (3, 4, 6144, 16384, 128), (4, 8, 3072, 9216, 128), (4, 8, 6144, 2048, 128),
    (5, 8, 6656, 6656, 128), (4, 16, 6144, 16384, 128), (3, 16, 3072, 9216, 128), (4, 16, 22016, 4096, 128), (4, 16, 4096, 1536, 128), (8, 12, 6656, 6656, 128),
    (4, 32, 3072, 9216, 128), (4, 24, 6144, 16384, 128), (2, 32, 6656, 6656, 128), (4, 32, 6144, 2048, 128), (4, 64, 3072, 9216, 128), (4, 48, 6656, 6656, 128),
    # round 4's rule edits: 128-column tiles from 21504 columns (also at several rows), 17..32 rows of a down-projection on the split build
    (4, 4, 8192, 22016, 128), (8, 2, 4096, 22016, 128), (4, 8, 4096, 28672, 128), (4, 1, 8192, 24576, 128), (4, 24, 22016, 4096, 128), (4, 32, 10240, 2560, 128),
    (5, 20, 16384, 3072, 128), (3, 17, 10240, 5120, 128),
    # 33..128 rows: the shared-activation-tile kernel's shape choice, and the rule that sends a layer to it
    (4, 128, 3072, 9216, 128), (4, 96, 6144, 16384, 128), (4, 40, 6656, 6656, 128), (8, 64, 6656, 6656, 128), (3, 128, 6144, 4096, 128), (5, 80, 3072, 9216, 64),
    (4, 128, 4096, 22016, 128), (4, 64, 6144, 2048, 128),
    # 129..256 rows
    (4, 256, 6144, 2048, 128), (4, 192, 3072, 9216, 128), (8, 256, 6656, 6656, 128), (3, 160, 4096, 11008, 64), (4, 256, 6144, 16384, 128), (5, 192, 6144, 4096, 128),
    # 2-bit beyond 32 rows
    (2, 64, 4096, 11008, 128), (2, 128, 6656, 6656, 128), (2, 256, 3072, 9216, 64), (2, 96, 6144, 4096, 128),
    # 257..512 rows
    (4, 512, 3072, 9216, 128), (4, 384, 6656, 6656, 128), (8, 512, 4096, 11008, 128), (4, 320, 6144, 2048, 128),
]
# beyond 32 rows on layers whose tables of all groups do not fit in LDS beside the activation images: the windowed-table builds of the shared-activation-tile kernel
LONG_K = [(4, 96, 24576, 5120, 128), (4, 256, 28672, 6144, 128), (8, 64, 16384, 7168, 64), (3, 160, 20480, 5120, 128), (2, 128, 6656, 6656, 64), (4, 384, 24576, 1536, 128),
          (5, 128, 16384, 3072, 32)]
POINTS += LONG_K
QUICK = POINTS[::4]


def candidates(bits, m, k, n, g):
    """(label, variant, split) of every forced launch that applies to the point; the measurement library falls back to the round-2
    kernels where a forced launch does not take the shape, so a candidate can only be as slow as those."""
    kb, t32 = (k + 31) // 32, (n + 31) // 32
    c = [("round-2 kernels, unsplit", 399, 0), ("round-2 rules (their split plans)", 398, 0), ("staged-operand, automatic tile", 300, 0),
         ("staged-operand, 32 columns", 302, 0), ("staged-operand, 48 columns", 303, 0), ("staged-operand, 64 columns", 304, 0)]
    if m == 1:
        c.append(("staged-operand, 16 columns", 301, 0))
    if bits in (4, 8) and m <= 4 and n >= 16384:
        c.append(("staged-operand, 128 columns", 340, 0))
        if m == 1:
            c.append(("round-2 kernel, 128 columns", 107, 0))
    if m > 32:
        c += [("32-row build, blocks of rows in grid.y (up to 256 rows)", 348, 0), ("shared activation tile, 2 x 48 columns", 351, 0), ("shared activation tile, 4 x 48 columns", 352, 0),
              ("shared activation tile, 2 x 64 columns", 353, 0), ("shared activation tile, 4 x 64 columns", 354, 0)]
        if m > 64:
            c += [("shared activation tile, 2 x 48 columns x 64 rows", 355, 0), ("shared activation tile, 4 x 48 columns x 64 rows", 356, 0)]
    if m == 1 and kb >= 256 and t32 <= 256 and g % 128 == 0:
        for s in (2, 3, 4, 6, 8, 12):
            c.append(("staged-operand split x%d, 64 columns" % s, 334, s))
        for s in (4, 8):
            c.append(("staged-operand split x%d, 32 columns" % s, 332, s))
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slack", type=float, default=1.08)
    ap.add_argument("--points", default="all", choices=("all", "quick", "long_k"))
    ap.add_argument("--only", default="", help="bits,rows,K,N,group[;...]: these points instead of a list")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    prod, tun = _lib.load(), _lib.load_tuning()
    points = POINTS if args.points == "all" else LONG_K if args.points == "long_k" else QUICK
    if args.only:
        points = [tuple(int(x) for x in pt.split(",")) for pt in args.only.split(";")]
    print("# device %s; period of a dependent chain of launches, cold, us; the automatic launch (production library) against every forced launch; slack %.2f" % (
        torch.cuda.get_device_name(0), args.slack), flush=True)
    failed = 0
    for bits, m, k, n, g in points:
        iters = 60 if k * n > 5e7 else 120

        def period(lib, variant, split):
            return gemv_sweep.gemv_case(lib, dev, k, n, bits, g, m, True, iters, variant, split, both=True)[4]

        def measure():
            auto = period(prod, -1, 0)
            best, best_label, cells = auto, "automatic", []
            for label, v, s in candidates(bits, m, k, n, g):
                try:
                    t = period(tun, v, s)
                except AssertionError:
                    continue
                cells.append("%s %.2f" % (label, t))
                if t < best:
                    best, best_label = t, label
            auto2 = period(prod, -1, 0)  # the automatic launch once more, after the others: the box drifts by a per cent or two over a point
            auto = min(auto, auto2)
            return auto, best, best_label, cells

        # "best" is the minimum of a dozen single measurements, each with a spread of a few per cent (the same launch under two labels has read 20.3 and 22.2 us
        # in one run: profiles/r04/check_rules_10.txt), so a ratio just above the slack may be that spread: a point fails only if a second pass over it fails too
        for attempt in (1, 2):
            auto, best, best_label, cells = measure()
            ok = auto <= args.slack * best
            tag = "ok  " if ok else ("FAIL" if attempt == 2 else "again")
            print("%s w%d m%-3d %5dx%-5d g%-3d  automatic %7.2f  best %7.2f (%s)  ratio %.3f" % (tag, bits, m, k, n, g, auto, best, best_label, auto / best), flush=True)
            print("       " + "; ".join(cells), flush=True)
            if ok:
                break
        failed += 0 if ok else 1
    print("# %d of %d points within %.2f x the best forced launch" % (len(points) - failed, len(points), args.slack), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
