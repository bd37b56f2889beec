#!/usr/bin/env python3
"""DESIGN.md section 4's table from the committed profiles:  python tools/design_table.py profiles/r04/bench"""
import json,glob,os,sys
root=sys.argv[1]
names=[("cfg2_w4_4096x11008_m1","**2: A16W4 4096×11008 g128 M=1 (the metric)**"),("cfg3_w2_4096x4096_m1","3: 4096×4096 M=1, 2-bit"),("cfg3_w3_4096x4096_m1","3: … 3-bit"),("cfg3_w4_4096x4096_m1","3: … 4-bit"),("cfg3_w5_4096x4096_m1","3: … 5-bit"),("cfg3_w6_4096x4096_m1","3: … 6-bit"),("cfg3_w7_4096x4096_m1","3: … 7-bit"),("cfg3_w8_4096x4096_m1","3: … 8-bit"),
("cfg4_w4_8192x28672_m1_one_gpu","4: A16W4 8192×28672 M=1 on one GPU"),("cfg4_shard8_w4_8192x3584_m1","4: per-rank shard at 8 GPUs, 8192×3584"),("cfg5_w3_4096x11008_m1","5: A16W3 4096×11008 M=1"),("cfg5_w3_4096x11008_m4","5: … M=4"),("cfg5_w3_4096x11008_m16","5: … M=16"),
("w4_4096x11008_m4","A16W4 4096×11008 M=4"),("w4_4096x11008_m8","… M=8"),("w4_4096x11008_m16","… M=16"),("w4_4096x11008_m32","… M=32"),("w4_4096x11008_m64","… M=64 (`gemv_rows_kernel`, 2 × 48 columns × 32 rows)"),("w4_4096x11008_m128","… M=128 (`gemv_rows_kernel`, 2 × 48 columns × 64 rows)"),("w4_11008x4096_m1_down_proj","A16W4 11008×4096 M=1 (7B down-projection, fill-once split)"),("w5_4096x14336_m1","A16W5 4096×14336 M=1"),
("dequant_w4_4096x11008","dequant w4 4096×11008 (write-bound)"),("dequant_w4_4096x4096","dequant w4 4096×4096"),("dequant_w3_4096x4096","dequant w3 4096×4096")]
print("| Config (BASELINE.json) | events (chain) µs | % of 8 TB/s | events (isolated) µs | rocprof µs | step µs | pure read, same run: flat / tiles µs per step | traffic ÷ algorithmic |")
print("|---|---|---|---|---|---|---|---|")
for d,label in names:
    p=os.path.join(root,d)
    if not os.path.exists(os.path.join(p,'bench.json.log')): continue
    rec=json.loads([l for l in open(os.path.join(p,'bench.json.log')).read().splitlines() if l.startswith('{')][-1])
    r=rec['roofline']
    try: t="%.2f"%json.load(open(os.path.join(p,'pmc_traffic.json')))['traffic_over_algorithmic']
    except Exception: t="–"
    try: tr=[l for l in open(os.path.join(p,'bench_kernel_trace.txt')).read().splitlines() if l.startswith('all launches')][0].split(' avg ')[1].split(' us')[0]; tr="%.2f"%float(tr)
    except Exception: tr="–"
    fl=r.get('read_floor_us') or {}
    fls="%s / %s" % ("%.2f"%fl['flat'] if fl.get('flat') else "–", "%.2f"%fl['tiles'] if fl.get('tiles') else "–") if fl else "–"
    print("| %s | %.2f | %.1f | %.2f | %s | %.2f | %s | %s |" % (label, r['avg_kernel_us'], 100*r['frac'], r['isolated_launch_avg_kernel_us'], tr, rec['end_to_end_us_per_step'], fls, t))
