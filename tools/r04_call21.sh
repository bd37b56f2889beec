#!/bin/bash
# round 4, GPU call 21: rebuilt library (shared-activation-tile kernel, shape by cost model, few-tile layers on the 32-row build): GPU suite, rule check, group sizes 64 / 32, crossover
mkdir -p gpurun_out/r04
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu_call21.txt 2>&1
rc=$?; tail -4 gpurun_out/r04/pytest_gpu_call21.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/check_rules.py > gpurun_out/r04/check_rules_3.txt 2>&1
rc=$?; grep -c "^ok" gpurun_out/r04/check_rules_3.txt; grep "^FAIL\|^#" gpurun_out/r04/check_rules_3.txt; [ $rc -le 1 ] || exit 1
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 4096x11008,4096x4096,8192x8192 --bits 4,3 --group 64 --rows 64,128 --launches=-1,348,350,399 > gpurun_out/r04/sweep_rows_group64.txt 2>&1 || exit 1
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 4096x11008,4096x4096 --bits 4,8 --group 32 --rows 64,128 --launches=-1,348,350,399 > gpurun_out/r04/sweep_rows_group32.txt 2>&1 || exit 1
cat gpurun_out/r04/sweep_rows_group64.txt gpurun_out/r04/sweep_rows_group32.txt | cut -c1-130
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192,4096x4096 --rows 192,256 --launches=-1,348,350 > gpurun_out/r04/sweep_rows_192_256.txt 2>&1 || exit 1
cut -c1-130 gpurun_out/r04/sweep_rows_192_256.txt
timeout -k 10 300 python tools/linear_crossover.py --rows 32,64,96,128,192,256 > gpurun_out/r04/linear_crossover_2.txt 2>&1 || exit 1
cat gpurun_out/r04/linear_crossover_2.txt
