#!/bin/bash
# round 4, GPU call 17: what bounds a workgroup of the shared-activation-tile kernel - one stage removed at a time (results of ablated builds are wrong by construction)
mkdir -p gpurun_out/r04
for ab in 0 1 2 4 8 16 3 6; do
  echo "## ablate $ab (1 no barrier, 2 no activation fetch, 4 no MFMA, 8 no unpack, 16 no row loads in the loop)"
  timeout -k 10 100 tools/microbench/rows_lab_a$ab 4096 4096 64 2>&1 | grep "shared activation" || exit 1
  timeout -k 10 100 tools/microbench/rows_lab_a$ab 4096 11008 64 2>&1 | grep "shared activation" || exit 1
done > gpurun_out/r04/rows_lab_ablations.txt 2>&1
cut -c1-60,90-110,150-220 gpurun_out/r04/rows_lab_ablations.txt
