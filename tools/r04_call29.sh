#!/bin/bash
# round 4, GPU call 29: the frozen sources (shared-activation-tile kernel with 32 or 64 rows per workgroup): parity suite, rule check, QuantLinear crossover, profiles part a
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final5.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final5.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1100 python tools/check_rules.py > gpurun_out/r04/check_rules_7.txt 2>&1
echo "check_rules rc $?"; grep "^FAIL" gpurun_out/r04/check_rules_7.txt; tail -1 gpurun_out/r04/check_rules_7.txt
timeout -k 10 300 python tools/linear_crossover.py --rows 8,16,32,48,64,96,128,192,256,320 > gpurun_out/r04/linear_crossover_5.txt 2>&1
echo "crossover rc $?"; cat gpurun_out/r04/linear_crossover_5.txt
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof6 a
echo "profile a rc $?"
