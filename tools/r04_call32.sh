#!/bin/bash
# round 4, GPU call 32: the frozen sources (one-launch kernels up to 512 rows): parity suite, rule check, QuantLinear crossover, profiles part a
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final6.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final6.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1100 python tools/check_rules.py > gpurun_out/r04/check_rules_8.txt 2>&1
echo "check_rules rc $?"; grep "^FAIL" gpurun_out/r04/check_rules_8.txt; tail -1 gpurun_out/r04/check_rules_8.txt
timeout -k 10 300 python tools/linear_crossover.py --rows 8,16,32,48,64,96,128,192,256,384,512,640 > gpurun_out/r04/linear_crossover_7.txt 2>&1
echo "crossover rc $?"; cat gpurun_out/r04/linear_crossover_7.txt
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof7 a
echo "profile a rc $?"
