#!/bin/bash
# round 4, GPU call 19: shared-activation-tile kernel with 16 waves per workgroup (4 per SIMD: vector instructions issue 1.45-2x faster than at 2 per SIMD, profiles/r01/valu_issue_rates.txt)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rows_lab_16_waves.txt; : > $out
for shape in "4096 4096 64" "4096 11008 64" "4096 11008 128" "8192 8192 128"; do
  timeout -k 10 120 tools/microbench/rows_lab $shape >> $out 2>&1 || exit 1
done
sed -E 's/ worst [^|]*\|//; s/ok  bad 0//' $out | cut -c1-170
