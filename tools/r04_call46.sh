#!/bin/bash
# round 4, GPU call 46: the whole rule check with a second pass over a point that fails (final library); decoder layers on the final library
mkdir -p gpurun_out/r04
timeout -k 10 1000 python tools/check_rules.py > gpurun_out/r04/check_rules_11.txt 2>&1
echo "check_rules rc $?"; grep "^FAIL\|^again" gpurun_out/r04/check_rules_11.txt | cut -c1-220; tail -1 gpurun_out/r04/check_rules_11.txt
for model in 7b 13b 70b; do
  timeout -k 10 120 python tools/decode_layer_bench.py --model $model --fused >> gpurun_out/r04/decode_layer_final.txt 2>&1 || exit 1
done
grep -i "layer\|total" gpurun_out/r04/decode_layer_final.txt | cut -c1-160 | tail -12
