#!/bin/bash
# round 4, GPU call 31: the one-launch kernels up to 512 rows: parity, 320 / 384 / 512 rows against 16-row passes and the forced shapes, QuantLinear crossover
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gemv_dyn.py -x -q -m gpu -k "five_twelve or shared_activation_tile_is_what" > gpurun_out/r04/pytest_rows512.txt 2>&1
rc=$?; tail -3 gpurun_out/r04/pytest_rows512.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192,4096x4096,8192x28672,11008x4096 --rows 320,384,512 --launches=-1,399,352,353,354,355,356 > gpurun_out/r04/sweep_rows_320_512.txt 2>&1 || exit 1
cut -c1-130 gpurun_out/r04/sweep_rows_320_512.txt
timeout -k 10 300 python tools/linear_crossover.py --rows 256,320,384,512 > gpurun_out/r04/linear_crossover_6.txt 2>&1 || exit 1
cat gpurun_out/r04/linear_crossover_6.txt
timeout -k 10 300 python tools/linear_crossover.py --shape 8192,8192 --rows 128,256,384,512 > gpurun_out/r04/linear_crossover_8192.txt 2>&1 || exit 1
cat gpurun_out/r04/linear_crossover_8192.txt
