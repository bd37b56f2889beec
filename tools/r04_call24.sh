#!/bin/bash
# round 4, GPU call 24: row blocks of a column tile on ONE XCD (1-D grid, ids mapped in the kernel) against the shipped library's blocks in grid.y (lab harness)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rows_lab_xcd_rows.txt; : > $out
for shape in "4096 11008 64" "4096 11008 128" "8192 8192 128" "4096 22016 128"; do
  timeout -k 10 150 tools/microbench/rows_lab $shape 0x0f >> $out 2>&1 || exit 1
done
sed -E 's/ worst [^|]*\|//; s/ok  bad 0//' $out | cut -c1-170
