#!/bin/bash
# round 4, GPU call 48: SQ counters of the windowed-table build on the 70B down-projection at 128 rows, beside the full-table build on 4096x11008 at 128 rows
mkdir -p gpurun_out/r04
timeout -k 10 300 bash tools/profile_sq_counters.sh gpurun_out/r04/sq_down_proj_70b_m128 --shape 28672,8192,4,128,128 > gpurun_out/r04/sq_counters_down_proj_70b_m128.txt 2>&1
echo "sq 1 rc $?"; tail -12 gpurun_out/r04/sq_counters_down_proj_70b_m128.txt | cut -c1-160
timeout -k 10 300 bash tools/profile_sq_counters.sh gpurun_out/r04/sq_4096x11008_m128 --shape 4096,11008,4,128,128 > gpurun_out/r04/sq_counters_4096x11008_m128.txt 2>&1
echo "sq 2 rc $?"; tail -12 gpurun_out/r04/sq_counters_4096x11008_m128.txt | cut -c1-160
