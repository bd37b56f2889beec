#!/bin/bash
# round 4, GPU call 2: parity suite on the new builds (128-column / narrowed tiles, fused launches on the staged-operand kernel, long-K by-id),
# wide-layer sweep with them, fused decoder layers, bench.py with read floors and the output check
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_1.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04/pytest_gpu_1.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python tools/sweep_wide_shapes.py --launches=-1,107,304,340,341,342 > gpurun_out/r04/sweep_wide_2.txt 2>&1
rc=$?; echo "sweep rc $rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for model in 7b 13b; do
  timeout -k 10 120 python tools/decode_layer_bench.py --model $model --fused > gpurun_out/r04/decode_${model}_fused.txt 2>&1 || exit 1
  timeout -k 10 120 python tools/decode_layer_bench.py --model $model > gpurun_out/r04/decode_${model}.txt 2>&1 || exit 1
done
for rows in 4 8 16; do
  timeout -k 10 120 python tools/decode_layer_bench.py --model 7b --fused --rows $rows > gpurun_out/r04/decode_7b_fused_rows$rows.txt 2>&1 || exit 1
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_cfg2_1.json 2> gpurun_out/r04/bench_cfg2_1.err
echo "bench rc $?"
timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r04/bench_cfg4_1.json 2> gpurun_out/r04/bench_cfg4_1.err
echo "bench cfg4 rc $?"
tail -3 gpurun_out/r04/decode_7b_fused.txt
