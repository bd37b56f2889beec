#!/bin/bash
# round 4, GPU call 50: the committed end state as the driver will run it - smoke(), bench.py with and without flags (traffic attached from the committed profiles)
mkdir -p gpurun_out/r04
timeout -k 10 120 python __graft_entry__.py --smoke > gpurun_out/r04/smoke.txt 2>&1
echo "smoke rc $?"; tail -1 gpurun_out/r04/smoke.txt
timeout -k 10 150 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command_12.json.log 2> gpurun_out/r04/bench_driver_command_12.err
echo "bench rc $?"; grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"traffic": [0-9.a-z]*' gpurun_out/r04/bench_driver_command_12.json.log | head -3
timeout -k 10 150 python bench.py > gpurun_out/r04/bench_default_12.json.log 2> /dev/null
echo "bench default rc $?"
timeout -k 10 200 python -m pytest tests/test_bench_contract.py tests/test_capi_and_host.py -m gpu -x -q 2>&1 | tail -2
