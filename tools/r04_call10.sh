#!/bin/bash
# round 4, GPU call 10: the captured (GEMV + all-gather) path rehearsed with a one-rank RCCL group (the refactored capture-then-agree-then-replay order)
mkdir -p gpurun_out/r04
XBIT_BENCH_FORCE_PG=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg2_one_rank_rccl.json.log 2> gpurun_out/r04/bench_cfg2_one_rank_rccl.err
echo "cfg2 rc $?"
XBIT_BENCH_FORCE_PG=1 timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg4_one_rank_rccl.json.log 2> gpurun_out/r04/bench_cfg4_one_rank_rccl.err
echo "cfg4 rc $?"
XBIT_BENCH_FORCE_PG=1 timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu --no-graph > gpurun_out/r04/bench_cfg4_one_rank_rccl_eager.json.log 2>/dev/null
echo "cfg4 eager rc $?"
tail -c 600 gpurun_out/r04/bench_cfg2_one_rank_rccl.err
