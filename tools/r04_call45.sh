#!/bin/bash
# round 4, GPU call 45: the QuantLinear long-K test added after call 42, with the module's other GPU tests and the bench contract
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_quant_linear.py tests/test_bench_contract.py tests/test_launch_rules.py -m gpu -x -q > gpurun_out/r04/pytest_gpu_quant_linear.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_quant_linear.txt
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/r04/pytest_gpu_quant_linear.txt | cut -c1-300 | tail -20; fi
exit $rc
