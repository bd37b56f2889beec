#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: where does (step time - kernel time) go?
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 bench.py --no-cpu --config cfg4
    python tools/kernel_gaps.py OUT [name-substring]
Prints, for the longest run of back-to-back launches of the dominant kernel, the distribution of start[i+1] - end[i]."""
import csv
import glob
import sys

import numpy as np


def main():
    out = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "gemv"
    files = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sel = [(s, e) for s, e, n in rows if want in n]
    print("# %d kernels in the trace, %d match '%s'" % (len(rows), len(sel), want))
    s = np.array([x[0] for x in sel], dtype=np.int64)
    e = np.array([x[1] for x in sel], dtype=np.int64)
    gap = (s[1:] - e[:-1]) * 1e-3
    dur = (e - s) * 1e-3
    # runs of launches with gaps under 50 us (one timed region, one sample ...)
    cut = np.where(gap > 50.0)[0]
    bounds = [0] + [int(c) + 1 for c in cut] + [len(sel)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b - a < 20:
            continue
        g, d = gap[a:b - 1], dur[a:b]
        print("run of %4d launches: kernel avg %.2f us | gap avg %.2f median %.2f p90 %.2f max %.2f us | gaps > 3 us: %d (sum %.0f us) | period %.2f us" % (
            b - a, d.mean(), g.mean(), np.median(g), np.percentile(g, 90), g.max(), int((g > 3).sum()), g[g > 3].sum(), (e[b - 1] - s[a]) * 1e-3 / (b - a)))
        big = np.where(g > 3)[0]
        if len(big):
            print("      positions of the gaps > 3 us in the run:", " ".join(str(int(i)) for i in big[:40]))


if __name__ == "__main__":
    main()
