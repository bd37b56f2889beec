#!/bin/bash
# round 4, GPU call 43: the whole rule check (89 held-out points) on the final library; QuantLinear's fused path against dequantise + dense GEMM on the long-K layers the
# windowed-table builds now take
mkdir -p gpurun_out/r04
timeout -k 10 120 python tools/linear_crossover.py --shape 28672,8192 --rows 64,128,256,512 > gpurun_out/r04/linear_crossover_long_k.txt 2>&1
echo "crossover rc $?"; grep -v "amdgpu.ids" gpurun_out/r04/linear_crossover_long_k.txt | cut -c1-200 | tail -8
timeout -k 10 100 python tools/linear_crossover.py --shape 14336,4096 --rows 128,256,512 >> gpurun_out/r04/linear_crossover_long_k.txt 2>&1
echo "crossover 2 rc $?"; grep -v "amdgpu.ids" gpurun_out/r04/linear_crossover_long_k.txt | cut -c1-200 | tail -5
timeout -k 10 900 python tools/check_rules.py > gpurun_out/r04/check_rules_10.txt 2>&1
echo "check_rules rc $?"; grep "^FAIL" gpurun_out/r04/check_rules_10.txt | cut -c1-220; tail -1 gpurun_out/r04/check_rules_10.txt
