#!/bin/bash
# round 4, GPU call 20: where a wave of the shared-activation-tile kernel spends its cycles (SQ counters, one group per pass; 4096x4096, 64 rows: 86 workgroups, nothing contended)
mkdir -p gpurun_out/r04/rows_pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/gpurun_out/r04/rows_pmc/p$i -o p -- $R/tools/microbench/rows_lab 4096 4096 64 0x21 > $R/gpurun_out/r04/rows_pmc/p$i.log 2>&1 || { tail -5 $R/gpurun_out/r04/rows_pmc/p$i.log; echo "pass $i failed"; }
done
cd $R
python - > gpurun_out/r04/rows_pmc_summary.txt <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/r04/rows_pmc/p*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemv_rows" not in k: continue
        name="NW16" if "16>" in k.split("(")[0] else "NW8"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name in acc:
        print(f.split("/")[3],name," ".join("%s=%.0f"%(c,sum(v)/len(v)) for c,v in acc[name].items()))
PY
for i in 1 2 3 4 5 6; do tail -3 gpurun_out/r04/rows_pmc/p$i.log > gpurun_out/r04/rows_pmc_p$i.tail; done
rm -rf gpurun_out/r04/rows_pmc
cat gpurun_out/r04/rows_pmc_summary.txt
