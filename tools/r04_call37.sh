#!/bin/bash
# round 4, GPU call 37: 2-bit on the shared-activation-tile kernel (units of 256 k): parity, then against 16-row passes
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gemv_dyn.py tests/test_capi_and_host.py -x -q -m gpu -k "shared_activation or thirty_three or five_twelve or spill" > gpurun_out/r04/pytest_rows_w2.txt 2>&1
rc=$?; tail -3 gpurun_out/r04/pytest_rows_w2.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192,4096x4096,8192x28672 --bits 2 --rows 64,128,256,512 --launches=-1,399,351,352,353,354 > gpurun_out/r04/sweep_rows_w2.txt 2>&1 || exit 1
cut -c1-130 gpurun_out/r04/sweep_rows_w2.txt
timeout -k 10 300 python tools/linear_crossover.py --bits 2 --rows 64,128,256,512 > gpurun_out/r04/linear_crossover_w2.txt 2>&1 || exit 1
cat gpurun_out/r04/linear_crossover_w2.txt
