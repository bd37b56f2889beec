#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the small files kept under profiles/:
bench_kernel_stats.csv (the --stats table), bench_kernel_trace_phases.txt (GEMV launch durations by
bench phase) and pmc_traffic.json (HBM bytes per launch, FETCH_SIZE x2 on gfx950 + WRITE_SIZE)."""
import csv
import glob
import json
import os
import statistics
import sys

KERNEL = "gemv_mfma_kernel"
ALGO_BYTES = 23455232  # 4096x11008 w4 g128 batch 1 (bench.py)
WORKLOAD = "A16W4 GEMV in=4096 out=11008 group=128 batch=1"


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def counter(root, name):
    path = find(root, "counter_collection.csv")
    vals, kname = [], None
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
                kname = row["Kernel_Name"]
    return vals, kname


def main(out):
    stats = find(os.path.join(out, "trace"), "kernel_stats.csv")
    if stats:
        with open(stats) as f, open(os.path.join(out, "bench_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    trace = find(os.path.join(out, "trace"), "kernel_trace.csv")
    if trace:
        rows = []
        with open(trace) as f:
            for row in csv.DictReader(f):
                if KERNEL in row["Kernel_Name"]:
                    rows.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
        rows.sort()
        d = [x for _, x in rows]
        with open(os.path.join(out, "bench_kernel_trace_phases.txt"), "w") as g:
            g.write("# rocprofv3 --kernel-trace of `python3 bench.py --no-cpu`: %s launches in start order\n" % KERNEL)
            g.write("all launches: n=%d avg %.2f us  median %.2f  min %.2f\n" % (len(d), statistics.mean(d), statistics.median(d), min(d)))
            # bench.py order: warm-up + graph replay (cold), event-bracketed cold launches, event-bracketed warm launches
            n = len(d)
            for name, part in (("first third (warm-up + hipGraph replay, cold)", d[: n // 3]), ("middle third (event-bracketed, cold)", d[n // 3: 2 * n // 3]),
                               ("last third (event-bracketed, one weight copy = cache-warm)", d[2 * n // 3:])):
                if part:
                    g.write("%-62s n=%d avg %.2f us  median %.2f  min %.2f\n" % (name, len(part), statistics.mean(part), statistics.median(part), min(part)))
    fetch, kname = counter(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
    write, _ = counter(os.path.join(out, "pmc_write"), "WRITE_SIZE")
    if fetch and write:
        fb, wb = statistics.mean(fetch) * 1024.0 * 2.0, statistics.mean(write) * 1024.0
        rec = {
            "workload": WORKLOAD,
            "kernel": kname,
            "command": "tools/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu --no-graph --steps 60 --warmup 10",
            "raw": {"FETCH_SIZE": {"launches": len(fetch), "mean_KB": statistics.mean(fetch), "min_KB": min(fetch), "max_KB": max(fetch)},
                    "WRITE_SIZE": {"launches": len(write), "mean_KB": statistics.mean(write), "min_KB": min(write), "max_KB": max(write)}},
            "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced streaming reads); WRITE_SIZE as is",
            "fetch_bytes_per_launch": fb,
            "write_bytes_per_launch": wb,
            "traffic_bytes_per_launch": fb + wb,
            "algorithmic_bytes_per_launch": ALGO_BYTES,
            "traffic_over_algorithmic": round((fb + wb) / ALGO_BYTES, 4),
        }
        with open(os.path.join(out, "pmc_traffic.json"), "w") as g:
            json.dump(rec, g, indent=1)
    for name in ("bench_kernel_stats.csv", "bench_kernel_trace_phases.txt", "pmc_traffic.json"):
        path = os.path.join(out, name)
        print(name, "ok" if os.path.exists(path) else "MISSING")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof")
