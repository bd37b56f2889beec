#!/bin/bash
# Every BASELINE.json config (and the dequant op) through tools/profile_bench.sh, on the GPU box:
#   bash tools/profile_all.sh gpurun_out/prof_r03 [a|b|c]    (a / b: half of the list each, to stay inside one gpurun call; c: the metric config, cfg4 and the
#                                                             launches of the shared-activation-tile kernel only - what a change to xbit_gemv_rows.h touches)
# Leaves one directory per workload with bench.json.log, bench_kernel_stats.csv, bench_kernel_trace.txt, pmc_traffic.json.
set -o pipefail
OUT=${1:-gpurun_out/prof_all}
PART=${2:-ab}
mkdir -p "$OUT"
run() {  # name, bench.py arguments...
  local name=$1; shift
  echo "== $name: bench.py $*"
  bash tools/profile_bench.sh "$OUT/$name" "$@" > "$OUT/$name.log" 2>&1 || echo "   FAILED (see $OUT/$name.log)"
  grep -h "all launches" "$OUT/$name/bench_kernel_trace.txt" 2>/dev/null
}
if [[ $PART == *a* ]]; then
  run cfg2_w4_4096x11008_m1
  run cfg4_w4_8192x28672_m1_one_gpu --config cfg4
  run cfg4_shard8_w4_8192x3584_m1 --shape 8192,3584,4,128,1
  for b in 2 3 4 5 6 7 8; do run cfg3_w${b}_4096x4096_m1 --shape 4096,4096,$b,128,1; done
fi
if [[ $PART == *b* ]]; then
  for m in 1 4 16; do run cfg5_w3_4096x11008_m$m --shape 4096,11008,3,128,$m; done
  run w4_4096x11008_m4 --shape 4096,11008,4,128,4
  run w4_4096x11008_m8 --shape 4096,11008,4,128,8
  run w4_4096x11008_m16 --shape 4096,11008,4,128,16
  run w4_4096x11008_m32 --shape 4096,11008,4,128,32
  run w4_4096x11008_m64 --shape 4096,11008,4,128,64
  run w4_4096x11008_m128 --shape 4096,11008,4,128,128
  run w4_11008x4096_m1_down_proj --shape 11008,4096,4,128,1
  run w5_4096x14336_m1 --shape 4096,14336,5,128,1
  run dequant_w4_4096x11008 --op dequant --shape 4096,11008,4,128,1
  run dequant_w4_4096x4096 --op dequant --shape 4096,4096,4,128,1
  run dequant_w3_4096x4096 --op dequant --shape 4096,4096,3,128,1
fi
if [[ $PART == *c* ]]; then
  run cfg2_w4_4096x11008_m1
  run cfg4_w4_8192x28672_m1_one_gpu --config cfg4
  run w4_4096x11008_m64 --shape 4096,11008,4,128,64
  run w4_4096x11008_m128 --shape 4096,11008,4,128,128
  run w4_28672x8192_m128_down_proj_70b --shape 28672,8192,4,128,128
fi
