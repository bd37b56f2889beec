#!/bin/bash
# round 4, GPU call 1: wide-layer launch sweep with the existing kernels + ablations / 16-wave builds of the metric kernel in the lab
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 700 python tools/sweep_wide_shapes.py > gpurun_out/r04/sweep_wide_1.txt 2>&1
echo "sweep rc $?"
timeout -k 10 200 tools/microbench/gemv_lab 4096 11008 > gpurun_out/r04/lab_1.txt 2>&1
echo "lab rc $?"
timeout -k 10 200 tools/microbench/gemv_lab 4096 11008 > gpurun_out/r04/lab_1_repeat.txt 2>&1
echo "lab rc $?"
tail -5 gpurun_out/r04/sweep_wide_1.txt
