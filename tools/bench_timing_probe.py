#!/usr/bin/env python3
"""How to time K dependent GEMV launches on the stream without host latency in the span (GPU):
graph replay vs eager launches, each queued behind a blocker kernel; K = 20 and 400."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth

dev = torch.device("cuda", 0)
lib = _lib.load()
k, n, bits, g, m = 4096, 11008, 4, 128, 1
nbytes = synth.algorithmic_bytes(k, n, bits, g, m)
copies = (512 << 20) // nbytes + 2
p = synth.make_problem(k, n, bits, g, m=m, seed=1)
act = torch.from_numpy(p.act).to(dev)
qw0, sc0, qz0 = (torch.from_numpy(x).to(dev) for x in (p.qweight, p.scales, p.qzeros))
qws = [(qw0 ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
outs = torch.empty((4, m, n), dtype=torch.float16, device=dev)
stream = torch.cuda.current_stream(dev)

def step(i, sp):
    st = lib.xbit_gemv_f16(outs[i % 4].data_ptr(), act.data_ptr(), qws[i % copies].data_ptr(), sc0.data_ptr(), qz0.data_ptr(), bits, g, m, k, n, 1, sp)
    assert st == 0

def run(first, count):
    sp = torch.cuda.current_stream(dev).cuda_stream
    for i in range(first, first + count):
        step(i, sp)

run(0, 50)
torch.cuda.synchronize()
# blocker calibration
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 1000000 / (e0.elapsed_time(e1) * 1e3)
print("blocker: %.1f cycles per us" % cyc_per_us)

def span(fn, blocker_us):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if blocker_us:
        torch.cuda._sleep(int(blocker_us * cyc_per_us))
    a.record(stream)
    fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3

for K in (20, 400):
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        run(50, K)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev)
    for name, fn, blk in (
        ("graph, no blocker", g1.replay, 0),
        ("graph behind blocker", g1.replay, 200),
        ("eager, no blocker", lambda: run(50, K), 0),
        ("eager behind blocker", lambda: run(50, K), 12 * K + 100),
    ):
        vals = sorted(span(fn, blk) / K for _ in range(7))
        print("K=%3d %-24s us/step: min %.2f med %.2f max %.2f" % (K, name, vals[0], vals[3], vals[-1]), flush=True)
    # two replays, time the second
    def two():
        pass
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    vals = []
    for _ in range(7):
        torch.cuda.synchronize()
        torch.cuda._sleep(int(300 * cyc_per_us))
        g1.replay(); a.record(stream); g1.replay(); b.record(stream)
        torch.cuda.synchronize()
        vals.append(a.elapsed_time(b) * 1e3 / K)
    vals.sort()
    print("K=%3d %-24s us/step: min %.2f med %.2f max %.2f" % (K, "second of two replays", vals[0], vals[3], vals[-1]), flush=True)
