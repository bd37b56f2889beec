#!/bin/bash
# round 4, GPU call 36: the driver's bench command on the committed tree: does the line pick up the committed PMC traffic (src + lib sha match)?
mkdir -p gpurun_out/r04
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final_check.json.log 2> gpurun_out/r04/bench_final_check.err || exit 1
python - <<'PY'
import json
rec=json.loads([l for l in open("gpurun_out/r04/bench_final_check.json.log").read().splitlines() if l.startswith("{")][-1])
print("value", rec["value"], "frac", rec["roofline"]["frac"], "traffic", rec["roofline"]["traffic"], "offline", rec["offline_profile"], "check", rec["output_check"]["ok"])
PY
timeout -k 10 200 python tools/decode_layer_bench.py > gpurun_out/r04/decode_layer_final.txt 2>&1 || exit 1
tail -12 gpurun_out/r04/decode_layer_final.txt
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 4096x11008,4096x12288,5120x13824,4096x14336 --rows 1 --launches=-1,303,343,342,304 > gpurun_out/r04/sweep_narrowed_tiles_metric_shape.txt 2>&1 || exit 1
cut -c1-130 gpurun_out/r04/sweep_narrowed_tiles_metric_shape.txt
