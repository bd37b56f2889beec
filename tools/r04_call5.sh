#!/bin/bash
# round 4, GPU call 5: parity suite (17-32-row split builds, wide rule from 21504 columns), the rule check on 60 points, fused + unfused decoder layers,
# QuantLinear crossover
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_4.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04/pytest_gpu_4.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 1000 python tools/check_rules.py > gpurun_out/r04/check_rules_2.txt 2>&1
echo "check_rules rc $?"
tail -2 gpurun_out/r04/check_rules_2.txt
timeout -k 10 400 python tools/linear_crossover.py > gpurun_out/r04/linear_crossover.txt 2>&1
echo "crossover rc $?"
for model in 7b 13b 70b; do
  timeout -k 10 120 python tools/decode_layer_bench.py --model $model --fused > gpurun_out/r04/decode_${model}_fused.txt 2>&1 || exit 1
  timeout -k 10 120 python tools/decode_layer_bench.py --model $model > gpurun_out/r04/decode_${model}.txt 2>&1 || exit 1
done
for rows in 4 8 16; do
  timeout -k 10 120 python tools/decode_layer_bench.py --model 7b --fused --rows $rows > gpurun_out/r04/decode_7b_fused_rows$rows.txt 2>&1 || exit 1
  timeout -k 10 120 python tools/decode_layer_bench.py --model 13b --fused --rows $rows > gpurun_out/r04/decode_13b_fused_rows$rows.txt 2>&1 || exit 1
done
grep -h "us per decoder" gpurun_out/r04/decode_*.txt | head -3
