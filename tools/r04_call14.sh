#!/bin/bash
# round 4, GPU call 14: lab prototype - 33..128 rows with the activation tile shared by 2 / 4 column groups of a workgroup (tools/microbench/rows_lab.hip)
mkdir -p gpurun_out/r04
timeout -k 10 120 tools/microbench/rows_lab 4096 11008 128 > gpurun_out/r04/rows_lab_m128.txt 2>&1
echo "m128 rc $?"; cat gpurun_out/r04/rows_lab_m128.txt
timeout -k 10 120 tools/microbench/rows_lab 4096 11008 64 > gpurun_out/r04/rows_lab_m64.txt 2>&1 || exit 1
echo "m64 rc $?"; cat gpurun_out/r04/rows_lab_m64.txt
timeout -k 10 120 tools/microbench/rows_lab 8192 8192 128 > gpurun_out/r04/rows_lab_8192_m128.txt 2>&1 || exit 1
cat gpurun_out/r04/rows_lab_8192_m128.txt
