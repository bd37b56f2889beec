#!/bin/bash
# round 4, GPU call 27: 64 rows per workgroup (four A operands per unpacked B fragment) against 32, shared-activation-tile kernel (lab harness)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rows_lab_64_rows.txt; : > $out
for shape in "4096 11008 64" "4096 11008 128" "4096 11008 256" "8192 8192 128" "8192 8192 256" "4096 4096 256"; do
  timeout -k 10 200 tools/microbench/rows_lab $shape 0xff >> $out 2>&1 || exit 1
done
sed -E 's/ worst [^|]*\|//; s/ok  bad 0//' $out | cut -c1-170
