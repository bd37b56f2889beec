#!/bin/bash
# round 4, GPU call 28: 64 rows per workgroup in the library: parity of the rows kernel, then the model's choice against every forced shape
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gemv_dyn.py tests/test_quant_linear.py -x -q -m gpu -k "shared_activation or thirty_three or one_twenty_nine or rows_in_grid or quant_linear_forward" > gpurun_out/r04/pytest_rows64.txt 2>&1
rc=$?; tail -3 gpurun_out/r04/pytest_rows64.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192,4096x4096,5120x13824,8192x28672,4096x22016,11008x4096 --rows 96,128,192,256 \
  --launches=-1,351,352,353,354,355,356 > gpurun_out/r04/sweep_rows_shared_tile_v2.txt 2>&1 || exit 1
timeout -k 10 300 python tools/sweep_wide_shapes.py --shapes 4096x11008,8192x8192 --bits 3,8 --rows 128,256 --launches=-1,351,352,353,354,355,356 > gpurun_out/r04/sweep_rows_shared_tile_v2_w3_w8.txt 2>&1 || exit 1
python - <<'PY'
import re,collections
for f in ("gpurun_out/r04/sweep_rows_shared_tile_v2.txt","gpurun_out/r04/sweep_rows_shared_tile_v2_w3_w8.txt"):
    d=collections.OrderedDict()
    for ln in open(f):
        m=re.match(r"w(\d) (\S+)\s+m(\d+)\s+[\d.]+ MB\s+launch (\S+)\s+events med\s+([\d.]+) min\s+([\d.]+)\s+period\s+([\d.]+)",ln)
        if m:
            b,s,mm,l,med,mn,per=m.groups(); d.setdefault((b,s,int(mm)),{})[l]=float(per)
    L=("-1","351","352","353","354","355","356")
    print("%-22s"%"period us"+"".join("%8s"%l for l in L))
    for k,v in d.items():
        best=min((v[l],l) for l in v if l!="-1")
        print("w%s %-12s m%-4d"%k+"".join("%8.2f"%v.get(l,float('nan')) for l in L)+"   best %s  auto/best %.3f"%(best[1],v["-1"]/best[0]))
PY
