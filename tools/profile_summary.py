#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the small files kept under profiles/:
bench_kernel_stats.csv (the --stats table), bench_kernel_trace.txt (durations of the dominant kernel,
all launches of the run in start order) and pmc_traffic.json (HBM bytes per launch of that kernel:
FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes).  The workload, its algorithmic bytes and the
library's sha256 come from the bench line of the same run (bench.json.log)."""
import csv
import glob
import json
import os
import statistics
import sys

SKIP = ("__amd_rocclr", "at::native", "elementwise", "Cijk", "sleep", "fill", "copy", "floor_")  # torch / runtime helpers, never the measured kernel


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def dominant_kernel(stats_path):
    best = None
    with open(stats_path) as f:
        for row in csv.DictReader(f):
            name = row["Name"]
            if any(s in name for s in SKIP) or "xbit" not in name:
                continue
            tot = float(row["TotalDurationNs"])
            if best is None or tot > best[1]:
                best = (name, tot, float(row["AverageNs"]), int(row["Calls"]))
    return best


def counter(root, name, kernel):
    path = find(root, "counter_collection.csv")
    vals = []
    if not path:
        return vals
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Kernel_Name"] == kernel and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    return vals


def main(out):
    line = [ln for ln in open(os.path.join(out, "bench.json.log")).read().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(line)
    stats = find(os.path.join(out, "trace"), "kernel_stats.csv")
    kernel = None
    if stats:
        with open(stats) as f, open(os.path.join(out, "bench_kernel_stats.csv"), "w") as g:
            g.write(f.read())
        dom = dominant_kernel(stats)
        if dom:
            kernel = dom[0]
    trace = find(os.path.join(out, "trace"), "kernel_trace.csv")
    if trace and kernel:
        rows = []
        with open(trace) as f:
            for row in csv.DictReader(f):
                if row["Kernel_Name"] == kernel:
                    rows.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3,
                                 row.get("VGPR_Count", "?"), row.get("Accum_VGPR_Count", "?"), row.get("SGPR_Count", "?"), row.get("LDS_Block_Size", "?"),
                                 row.get("Workgroup_Size_X", "?"), row.get("Grid_Size_X", "?")))
        rows.sort()
        d = [r[1] for r in rows]
        with open(os.path.join(out, "bench_kernel_trace.txt"), "w") as g:
            g.write("# rocprofv3 --kernel-trace of `python3 bench.py --no-cpu %s`\n" % " ".join(sys.argv[2:]))
            g.write("# workload: %s\n# kernel: %s\n" % (bench["config"]["workload"], kernel))
            g.write("# VGPR %s  AGPR %s  SGPR %s  LDS %s B  workgroup %s threads  grid %s threads\n" % rows[0][2:])
            g.write("all launches (every one streams cold weights: touch pass, warm-up, timed steps, roofline sample): "
                    "n=%d avg %.3f us  median %.3f  min %.3f  max %.3f\n" % (len(d), statistics.mean(d), statistics.median(d), min(d), max(d)))
            g.write("bench.py's own figures of the same kind of run (un-profiled): avg kernel %.3f us by hipExtLaunchKernel events, %.3f us per step end to end\n"
                    % (bench["roofline"]["avg_kernel_us"], bench["end_to_end_us_per_step"]))
    fetch = counter(os.path.join(out, "pmc_fetch"), "FETCH_SIZE", kernel) if kernel else []
    write = counter(os.path.join(out, "pmc_write"), "WRITE_SIZE", kernel) if kernel else []
    if fetch and write:
        fb, wb = statistics.mean(fetch) * 1024.0 * 2.0, statistics.mean(write) * 1024.0
        algo = bench["config"]["algorithmic_bytes_per_step"]
        rec = {
            "workload": bench["config"]["workload"],
            "kernel": kernel,
            "lib_sha256": bench.get("lib_sha256"),
            "src_sha256": bench.get("src_sha256"),
            "command": "tools/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu --steps 60 --warmup 10 ...",
            "raw": {"FETCH_SIZE": {"launches": len(fetch), "mean_KB": statistics.mean(fetch), "min_KB": min(fetch), "max_KB": max(fetch)},
                    "WRITE_SIZE": {"launches": len(write), "mean_KB": statistics.mean(write), "min_KB": min(write), "max_KB": max(write)}},
            "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced streaming reads, MI355X_MICROARCH.md); WRITE_SIZE as is",
            "fetch_bytes_per_launch": fb,
            "write_bytes_per_launch": wb,
            "traffic_bytes_per_launch": fb + wb,
            "algorithmic_bytes_per_launch": algo,
            "traffic_over_algorithmic": round((fb + wb) / algo, 4),
        }
        with open(os.path.join(out, "pmc_traffic.json"), "w") as g:
            json.dump(rec, g, indent=1)
    for name in ("bench.json.log", "bench_kernel_stats.csv", "bench_kernel_trace.txt", "pmc_traffic.json"):
        path = os.path.join(out, name)
        print(name, "ok" if os.path.exists(path) else "MISSING")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof")
