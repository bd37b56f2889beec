#!/bin/bash
# round 4, GPU call 9: part a of the profiles once more with the final bench.py (floor pattern from the dry run), and the driver's own command line
mkdir -p gpurun_out/r04
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof2 a
echo "profile a rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command.json.log 2> gpurun_out/r04/bench_driver_command.err
echo "bench rc $?"
timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg4_driver_steps.json.log 2>/dev/null
echo "bench cfg4 rc $?"
