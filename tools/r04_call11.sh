#!/bin/bash
# round 4, GPU call 11: more than 32 rows as blocks of 32 rows in grid.y of one launch (variant 348) against the 64-row build (production at 33..64 rows) and launches of 32 (349)
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests/test_gemv_dyn.py -m gpu -x -q > gpurun_out/r04/pytest_gpu_dyn_rows_y.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_dyn_rows_y.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192,5120x13824,4096x28672 --rows 48,64,96,128 --launches=-1,348,349,399 > gpurun_out/r04/sweep_rows_in_grid_y.txt 2>&1
echo "sweep rc $?"
grep -v amdgpu gpurun_out/r04/sweep_rows_in_grid_y.txt | tail -70
