#!/bin/bash
# round 4, GPU call 38: final sources (2-bit on the rows kernel): parity suite, profiles part a
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final8.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final8.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1100 python tools/check_rules.py > gpurun_out/r04/check_rules_9.txt 2>&1
echo "check_rules rc $?"; grep "^FAIL" gpurun_out/r04/check_rules_9.txt; tail -1 gpurun_out/r04/check_rules_9.txt
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof9 a
echo "profile a rc $?"
