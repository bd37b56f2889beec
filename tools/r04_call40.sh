#!/bin/bash
# round 4, GPU call 40: four waves per workgroup (half the activation images in LDS: room for the tables of a long K) against eight (lab harness)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rows_lab_4_waves.txt; : > $out
for shape in "4096 11008 64" "4096 11008 128" "8192 8192 128" "8192 8192 256"; do
  timeout -k 10 200 tools/microbench/rows_lab $shape 0x715 >> $out 2>&1 || exit 1
done
sed -E 's/ worst [^|]*\|//; s/ok  bad 0//' $out | cut -c1-170
