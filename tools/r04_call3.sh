#!/bin/bash
# round 4, GPU call 3: unaligned-load probe, parity suite (ragged K with several rows, 17-32 rows by id, long-K act-order), the former cliffs timed,
# split / 8-wave / deep-lookahead probes of the 128-column tiles, v_dot2 build of the metric kernel in the lab
mkdir -p gpurun_out/r04
timeout -k 10 60 tools/microbench/unaligned_probe > gpurun_out/r04/unaligned_probe.txt 2>&1; echo "probe rc $?"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_2.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04/pytest_gpu_2.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/sweep_ragged.py --bits 4 > gpurun_out/r04/sweep_ragged_shapes.txt 2>&1
rc=$?; echo "ragged rc $rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 8192x28672,4096x28672 --launches=-1,340,344:2,344:4,345:2,345:4,346,347 > gpurun_out/r04/sweep_wide_3.txt 2>&1
rc=$?; echo "wide rc $rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 8192x16384,8192x14336,8192x22016,4096x22016,6656x17920,5120x27648 --launches=-1,107,340,304 > gpurun_out/r04/sweep_wide_4.txt 2>&1
rc=$?; echo "wide4 rc $rc"; if [ $rc -ge 124 ]; then exit $rc; fi
LAB_ONLY=48-col timeout -k 10 200 tools/microbench/gemv_lab 4096 11008 > gpurun_out/r04/lab_2.txt 2>&1
echo "lab rc $?"
cat gpurun_out/r04/unaligned_probe.txt | head -8
