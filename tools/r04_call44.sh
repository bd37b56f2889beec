#!/bin/bash
# round 4, GPU call 44: every profile directory re-collected with the shipped library (parts a and b of tools/profile_all.sh; part c was call 42)
mkdir -p gpurun_out/r04
timeout -k 10 1100 bash tools/profile_all.sh gpurun_out/r04/prof11 ab
echo "profile ab rc $?"
