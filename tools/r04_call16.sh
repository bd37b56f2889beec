#!/bin/bash
# round 4, GPU call 16: shared-activation-tile kernel with two whole units of weight rows in flight (lab harness on the library's header)
mkdir -p gpurun_out/r04
timeout -k 10 120 tools/microbench/rows_lab 4096 11008 128 > gpurun_out/r04/rows_lab2_m128.txt 2>&1
echo "m128 rc $?"; cat gpurun_out/r04/rows_lab2_m128.txt
timeout -k 10 120 tools/microbench/rows_lab 4096 11008 64 > gpurun_out/r04/rows_lab2_m64.txt 2>&1 || exit 1
cat gpurun_out/r04/rows_lab2_m64.txt
timeout -k 10 120 tools/microbench/rows_lab 4096 4096 64 > gpurun_out/r04/rows_lab2_4096_m64.txt 2>&1 || exit 1
cat gpurun_out/r04/rows_lab2_4096_m64.txt
timeout -k 10 120 tools/microbench/rows_lab 8192 8192 128 > gpurun_out/r04/rows_lab2_8192_m128.txt 2>&1 || exit 1
cat gpurun_out/r04/rows_lab2_8192_m128.txt
