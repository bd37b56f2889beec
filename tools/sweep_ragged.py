#!/usr/bin/env python3
"""Shapes outside the fast path's alignment rules, beside their aligned neighbours: ragged K (K % 8 != 0, one row: the
staged-operand kernel with a masked tail; several rows: the coalesced column kernel) and group sizes that are not a
multiple of 32 (the staged-operand kernel's by-id builds; the column kernel beyond sixteen rows).  Period of a dependent chain of launches, cold HBM, production library.
    python tools/sweep_ragged.py [--bits 4,3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["XBIT_SWEEP_PERIOD"] = "1"
from xbitops_amd import _lib, synth  # noqa: E402
import gemv_sweep  # noqa: E402

CASES = [  # (K, N, group, rows, note)
    (7168, 4096, 128, 1, "aligned neighbour"),
    (7167, 4096, 128, 1, "K % 8 != 0: staged-operand kernel, masked tail"),
    (4096, 11008, 128, 1, "aligned neighbour (the metric shape)"),
    (4100, 11008, 128, 1, "K % 32 != 0"),
    (7168, 4096, 128, 4, "aligned neighbour, 4 rows"),
    (7167, 4096, 128, 4, "ragged ODD K with 4 rows (rows begin on odd halfwords)"),
    (7166, 4096, 128, 4, "ragged even K with 4 rows: 4-row build, masked tail (round 3: column kernel, 85 us)"),
    (7166, 4096, 128, 16, "ragged even K with 16 rows: 16-row build, masked tail"),
    (7166, 4096, 128, 32, "ragged even K with 32 rows: 32-row build, masked tail"),
    (4096, 4096, 64, 1, "aligned neighbour, group 64"),
    (4096, 4096, 48, 1, "group 48: by-id build, ids computed"),
    (4096, 4096, 16, 1, "group 16: by-id build (256 groups: 128 KB of table)"),
    (4096, 11008, 48, 1, "group 48: by-id build, ids computed"),
    (4096, 4096, 48, 4, "group 48, 4 rows: by-id build"),
    (4096, 4096, 48, 8, "group 48, 8 rows: by-id build (8-row)"),
    (4096, 4096, 48, 16, "group 48, 16 rows: by-id build (16-row)"),
    (4096, 4096, 48, 24, "group 48, 24 rows: 32-row by-id build (round 3: column kernel, 93 us)"),
    (4096, 4096, 48, 40, "group 48, 40 rows: launches of 32 + 8"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", default="4,3")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    print("# us per launch: period of a dependent chain of launches, cold; GB/s over the algorithmic bytes", flush=True)
    for bits in (int(b) for b in args.bits.split(",")):
        for k, n, g, m, note in CASES:
            nbytes, _avg, med, _mn = gemv_sweep.gemv_case(lib, dev, k, n, bits, g, m, True, 80, -1, 0)
            print("w%d %5dx%-5d g%-3d m%d  %7.2f us  %6.0f GB/s   %s" % (bits, k, n, g, m, med, synth.algorithmic_bytes(k, n, bits, g, m) / med * 1e-3, note), flush=True)


if __name__ == "__main__":
    main()
