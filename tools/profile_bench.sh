#!/bin/bash
# Collect, on the GPU box, the profiles DESIGN.md / bench.py cite for ONE bench.py workload:
#   1. bench.py's own JSON line                                   -> $OUT/bench.json.log
#   2. rocprofv3 --kernel-trace --stats of the same command       -> $OUT/bench_kernel_stats.csv, bench_kernel_trace.txt
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (no trace domains besides
#      --kernel-trace, as the pool requires)                       -> $OUT/pmc_traffic.json
# Usage (from the repo root on the box):  bash tools/profile_bench.sh gpurun_out/prof_cfg2 [bench.py arguments ...]
#   e.g.  bash tools/profile_bench.sh gpurun_out/prof_cfg4 --config cfg4
#         bash tools/profile_bench.sh gpurun_out/prof_cfg5_m16 --shape 4096,11008,3,128,16
set -eo pipefail
OUT=${1:-gpurun_out/prof}
shift || true
ROOT=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
python3 bench.py --no-cpu "$@" > "$OUT/bench.json.log" 2> "$OUT/bench.err.log"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$ROOT/bench.py" --no-cpu "$@" > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$ROOT/bench.py" --no-cpu --steps 60 --warmup 10 --roofline-launches 200 "$@" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$ROOT/bench.py" --no-cpu --steps 60 --warmup 10 --roofline-launches 200 "$@" > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
python3 tools/profile_summary.py "$OUT"
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"   # the condensed files are what is kept
