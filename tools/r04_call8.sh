#!/bin/bash
# round 4, GPU call 8: profiles of the remaining configs (part b) and smoke()
mkdir -p gpurun_out/r04
timeout -k 10 1000 bash tools/profile_all.sh gpurun_out/r04/prof b
echo "profile b rc $?"
timeout -k 10 120 python __graft_entry__.py --smoke > gpurun_out/r04/smoke.txt 2>&1
echo "smoke rc $?"; tail -1 gpurun_out/r04/smoke.txt
