#!/bin/bash
# round 4, GPU call 49: a comment in xbit_gemv_rows.h was corrected after call 47 (the library is byte-identical, its src_sha256 is not): every profile directory
# once more so that lib_sha256 AND src_sha256 of the records are the shipped ones; the driver's command lines
mkdir -p gpurun_out/r04
timeout -k 10 200 bash tools/profile_all.sh gpurun_out/r04/prof12 c
echo "profile c rc $?"
timeout -k 10 150 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command_11.json.log 2> gpurun_out/r04/bench_driver_command_11.err
echo "bench rc $?"; grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"traffic": [0-9.a-z]*' gpurun_out/r04/bench_driver_command_11.json.log | head -4
timeout -k 10 150 python bench.py > gpurun_out/r04/bench_default_11.json.log 2> /dev/null
echo "bench default rc $?"
timeout -k 10 150 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg4_driver_steps_11.json.log 2>/dev/null
echo "bench cfg4 rc $?"
timeout -k 10 600 bash tools/profile_all.sh gpurun_out/r04/prof12 ab
echo "profile ab rc $?"
