#!/bin/bash
# round 4, GPU call 7: the frozen sources - parity suite, then the profiles of every BASELINE config (part a)
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof a
echo "profile a rc $?"
