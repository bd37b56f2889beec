#!/bin/bash
# round 4, GPU call 34: final sources (comment-only edits since call 32): parity suite, profiles part a
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final7.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04/pytest_gpu_final7.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 bash tools/profile_all.sh gpurun_out/r04/prof8 a
echo "profile a rc $?"
