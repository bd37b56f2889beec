#!/bin/bash
# round 4, GPU call 47: the GPU parity suite once more, with the full-size 70B down-projection test and the QuantLinear long-K test added since call 42; the driver's command line
mkdir -p gpurun_out/r04
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "down_projection_128_rows" > gpurun_out/r04/pytest_gpu_down_proj.txt 2>&1
rc=$?; echo "new test rc $rc"; tail -2 gpurun_out/r04/pytest_gpu_down_proj.txt
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/r04/pytest_gpu_down_proj.txt | cut -c1-300 | tail -20; exit $rc; fi
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final10.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final10.txt
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/r04/pytest_gpu_final10.txt | cut -c1-300 | tail -20; exit $rc; fi
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command_10.json.log 2> gpurun_out/r04/bench_driver_command_10.err
echo "bench rc $?"; cut -c1-1500 gpurun_out/r04/bench_driver_command_10.json.log | grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"traffic": [0-9.a-z]*\|"output_check": "[a-z ]*"' | head
timeout -k 10 200 python bench.py > gpurun_out/r04/bench_default_10.json.log 2> /dev/null
echo "bench default rc $?"
