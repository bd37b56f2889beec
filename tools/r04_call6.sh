#!/bin/bash
# round 4, GPU call 6: parity suite with the 64-row build; rows 32 / 64 / 128 timed; QuantLinear crossover; rule check
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_5.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04/pytest_gpu_5.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python tools/sweep_wide_shapes.py --shapes 4096x11008,11008x4096,8192x8192,5120x13824 --rows 32,48,64,128 --launches=-1,399 > gpurun_out/r04/sweep_rows_32_to_128.txt 2>&1
rc=$?; echo "rows rc $rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python tools/linear_crossover.py > gpurun_out/r04/linear_crossover_2.txt 2>&1
echo "crossover rc $?"
timeout -k 10 1000 python tools/check_rules.py > gpurun_out/r04/check_rules_3.txt 2>&1
echo "check_rules rc $?"
tail -2 gpurun_out/r04/check_rules_3.txt
cat gpurun_out/r04/sweep_rows_32_to_128.txt | tail -34
