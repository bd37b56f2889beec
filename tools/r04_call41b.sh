#!/bin/bash
# round 4, GPU call 41b: does the rule "2-bit, 65-128 rows, < 192 tiles, K > 4096: 16-row passes" still hold now that the rows kernel keeps long-K tables as a window?
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 6144x4096,8192x3584,5120x5120,11008x4096,8192x1024 --bits 2 --rows 80,96,128 --launches=-1,399,350,351,353 > gpurun_out/r04/sweep_rows_w2_few_tiles_long_k.txt 2>&1 || exit 1
cut -c1-110 gpurun_out/r04/sweep_rows_w2_few_tiles_long_k.txt | grep -v "^#\|amdgpu.ids"
