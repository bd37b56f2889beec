#!/bin/bash
# round 4, GPU call 4: parity suite on the table-driven launch rule + odd K / unaligned operands + 32-row by-id build; the rule check on held-out points;
# the former cliffs again
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_3.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04/pytest_gpu_3.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/sweep_ragged.py --bits 4 > gpurun_out/r04/sweep_ragged_shapes_2.txt 2>&1
rc=$?; echo "ragged rc $rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python tools/check_rules.py > gpurun_out/r04/check_rules_1.txt 2>&1
echo "check_rules rc $?"
tail -3 gpurun_out/r04/check_rules_1.txt
