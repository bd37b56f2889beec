#!/bin/bash
# round 4, GPU call 15: the shared-activation-tile kernel (xbit_gemv_rows.h) in the library - parity, then every shape of it against the grid.y launch
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gemv_dyn.py -x -q -m gpu -k "shared_activation or thirty_three or rows_in_grid or seventeen_to_sixty" > gpurun_out/r04/pytest_rows.txt 2>&1
rc=$?; tail -5 gpurun_out/r04/pytest_rows.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/sweep_wide_shapes.py --shapes 4096x11008,4096x12288,8192x8192,4096x4096,11008x4096,5120x13824,8192x28672,4096x22016 --rows 48,64,96,128 \
  --launches=-1,348,351,352,353,354 > gpurun_out/r04/sweep_rows_shared_tile.txt 2>&1 || exit 1
tail -3 gpurun_out/r04/sweep_rows_shared_tile.txt
timeout -k 10 400 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192 --bits 3,8 --rows 64,128 --launches=-1,348,351,352,353,354 > gpurun_out/r04/sweep_rows_shared_tile_w3_w8.txt 2>&1 || exit 1
tail -3 gpurun_out/r04/sweep_rows_shared_tile_w3_w8.txt
