#!/bin/bash
# round 4, GPU call 39: profiles part b (with 64 / 128 rows), the driver's command lines, smoke()
mkdir -p gpurun_out/r04
timeout -k 10 1050 bash tools/profile_all.sh gpurun_out/r04/prof9 b
echo "profile b rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command_8.json.log 2> gpurun_out/r04/bench_driver_command_8.err
echo "bench rc $?"
timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg4_driver_steps_8.json.log 2>/dev/null
echo "bench cfg4 rc $?"
timeout -k 10 120 python __graft_entry__.py --smoke > gpurun_out/r04/smoke.txt 2>&1
echo "smoke rc $?"; tail -1 gpurun_out/r04/smoke.txt
