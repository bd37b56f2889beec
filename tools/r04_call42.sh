#!/bin/bash
# round 4, GPU call 42: final sources (windowed-table builds of the rows kernel): parity suite, profiles of what the change touches + the metric config and cfg4 with the shipped
# library, the driver's command lines, smoke(), the rule check on the long-K points
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final9.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final9.txt
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/r04/pytest_gpu_final9.txt | cut -c1-300 | tail -20; exit $rc; fi
timeout -k 10 500 bash tools/profile_all.sh gpurun_out/r04/prof10 c
echo "profile c rc $?"
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command_9.json.log 2> gpurun_out/r04/bench_driver_command_9.err
echo "bench rc $?"; cut -c1-400 gpurun_out/r04/bench_driver_command_9.json.log
timeout -k 10 200 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg4_driver_steps_9.json.log 2>/dev/null
echo "bench cfg4 rc $?"
timeout -k 10 120 python __graft_entry__.py --smoke > gpurun_out/r04/smoke.txt 2>&1
echo "smoke rc $?"; tail -1 gpurun_out/r04/smoke.txt
timeout -k 10 250 python tools/check_rules.py --points long_k > gpurun_out/r04/check_rules_long_k.txt 2>&1
echo "check_rules rc $?"; grep "^FAIL" gpurun_out/r04/check_rules_long_k.txt | cut -c1-200; tail -1 gpurun_out/r04/check_rules_long_k.txt
