#!/bin/bash
# round 4, GPU call 41: windowed-table builds of the rows kernel (long K; 64 rows per workgroup beyond K 4.6 K): parity tests of the new builds, then sweeps against
# what these layers ran before (348: 32-row build in grid.y, 399: 16-row passes) and against the other shapes
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "window or long_k or shared_activation or five_twelve" > gpurun_out/r04/pytest_gpu_window.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_window.txt
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/r04/pytest_gpu_window.txt | cut -c1-300 | tail -20; exit $rc; fi
timeout -k 10 500 python tools/sweep_wide_shapes.py --shapes 28672x8192,14336x4096,8192x8192,11008x4096 --rows 64,128,256,512 --launches=-1,348,399,351,353,355,356 > gpurun_out/r04/sweep_rows_long_k.txt 2>&1 || exit 1
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 4096x4096,11008x4096,8192x8192 --bits 2 --group 64 --rows 129,256 --launches=-1,399,351,352,353,354 > gpurun_out/r04/sweep_rows_long_k_w2.txt 2>&1 || exit 1
timeout -k 10 200 python tools/sweep_wide_shapes.py --shapes 8192x8192,8192x3584 --bits 2 --rows 128,256 --launches=-1,351,352,353,354 >> gpurun_out/r04/sweep_rows_long_k_w2.txt 2>&1 || exit 1
cut -c1-110 gpurun_out/r04/sweep_rows_long_k.txt gpurun_out/r04/sweep_rows_long_k_w2.txt | grep -v "^#\|amdgpu.ids"
