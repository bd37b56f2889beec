#!/bin/bash
# round 4, GPU call 18: shared-activation-tile kernel with the unit's A fragments and table entry read ahead of the first unpack (lab harness on the library's header)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rows_lab_prefetch.txt; : > $out
for shape in "4096 4096 64" "4096 11008 64" "4096 11008 128" "8192 8192 128"; do
  timeout -k 10 120 tools/microbench/rows_lab $shape >> $out 2>&1 || exit 1
done
cut -c1-75,100-125,150-230 $out
