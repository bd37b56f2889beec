#!/bin/bash
# round 4, GPU call 12: the frozen sources (rows beyond 32 in grid.y; no 64-row build): parity suite, rule check, QuantLinear crossover, profiles part a
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final2.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final2.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1000 python tools/check_rules.py > gpurun_out/r04/check_rules_4.txt 2>&1
echo "check_rules rc $?"; tail -1 gpurun_out/r04/check_rules_4.txt
timeout -k 10 300 python tools/linear_crossover.py > gpurun_out/r04/linear_crossover_3.txt 2>&1
echo "crossover rc $?"
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof3 a
echo "profile a rc $?"
