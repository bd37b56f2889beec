#!/usr/bin/env python3
"""Kernel-duration sweep on the GPU box: GEMV launch-shape variants (4-bit), every bit-width, the
batch edge, and dequant.  Per-launch durations come from hipExtLaunchKernel start/stop events
(xbit_profile_enable).  Prints one line per case:  name  avg/median/min us  GB/s(algorithmic)  %of 8 TB/s.

    python tools/gemv_sweep.py [--variants 0,1,2] [--quick]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth  # noqa: E402

COLD_BYTES = 512 << 20


_CYCLES_PER_US = None


def _blocker(us: float) -> None:
    """Keep the stream busy for ~`us` microseconds so that the launches queued behind it run back to back."""
    global _CYCLES_PER_US
    if _CYCLES_PER_US is None:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(1000000)
        c1.record()
        torch.cuda.synchronize()
        _CYCLES_PER_US = 1000000 / (c0.elapsed_time(c1) * 1e3)
    torch.cuda._sleep(int(us * _CYCLES_PER_US))


def measure(lib, fn, n=200, warm=10):
    """Per-launch kernel durations (hipExtLaunchKernel events) of `n` launches queued back to back behind a blocker -
    the kernel as it runs inside a dependent chain.  (A launch on an idle chip reads 0.3-1 us longer: XCD wake-up skew;
    XBIT_SWEEP_ISOLATED=1 measures that instead, with a host wait after every launch, as rounds 1-2 did.)"""
    ms = ctypes.c_float()
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    vals = []
    if os.environ.get("XBIT_SWEEP_ISOLATED"):
        lib.xbit_profile_enable(1)
        for i in range(n):
            fn(warm + i)
            _lib.check(lib.xbit_profile_last_ms(ctypes.byref(ms)), "profile")
            vals.append(ms.value * 1e3)
        lib.xbit_profile_enable(0)
    else:
        _blocker(100 + 30 * n)
        _lib.check(lib.xbit_profile_series_begin(n), "profile series")
        for i in range(n):
            fn(warm + i)
        buf, cnt = (ctypes.c_float * n)(), ctypes.c_int()
        _lib.check(lib.xbit_profile_series_end(buf, n, ctypes.byref(cnt)), "profile series")
        vals = [buf[i] * 1e3 for i in range(cnt.value)]
    vals.sort()
    return sum(vals) / len(vals), vals[len(vals) // 2], vals[0]


def chain_period(fn, n=200, first=0, reps=3):
    """Microseconds per launch of `n` launches queued back to back behind a blocker, between two events on the stream
    (no per-launch events): kernel + launch boundary, what a decoder pays per layer.  Median of `reps`."""
    out = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        _blocker(100 + 12 * n)
        e0.record()
        for i in range(n):
            fn(first + r * n + i)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    out.sort()
    return out[len(out) // 2]


_PROBLEM = {}


def _problem(k, n, bits, g, m):
    """The synthetic problem of a shape, generated once per process (the last one is kept: sweeps visit a shape's launches in a row)."""
    key = (k, n, bits, g, m)
    if key not in _PROBLEM:
        _PROBLEM.clear()
        _PROBLEM[key] = synth.make_problem(k, n, bits, g, m=m, seed=k + n + bits)
    return _PROBLEM[key]


def gemv_case(lib, dev, k, n, bits, g, m, cold=True, iters=200, variant=-1, split=0, both=False):
    nbytes = synth.algorithmic_bytes(k, n, bits, g, m)
    copies = (COLD_BYTES + nbytes - 1) // nbytes + 1 if cold else 1
    warm_launches = 2 * copies if cold else 10  # every copy touched before the timed launches (a first touch pays page-table walks)
    base = _problem(k, n, bits, g, m)
    act = torch.from_numpy(base.act).to(dev)
    qw = [(torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
    sc = [torch.from_numpy(base.scales).to(dev).clone() for c in range(copies)]
    qz = [torch.from_numpy(base.qzeros).to(dev).clone() for c in range(copies)]
    out = torch.empty((m, n), dtype=torch.float16, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream

    opts = _lib.GemvOptions(split_k=split, variant=variant)

    def fn(i):
        c = i % copies
        st = lib.xbit_gemv_f16_ex(out.data_ptr(), act.data_ptr(), qw[c].data_ptr(), sc[c].data_ptr(), qz[c].data_ptr(), bits, g, m, k, n, 1, sp, opts)
        assert st == 0, st

    avg, med, mn = measure(lib, fn, iters, warm_launches)
    if both:  # events AND chain period of the same buffers
        return nbytes, avg, med, mn, chain_period(fn, iters, warm_launches + iters)
    if os.environ.get("XBIT_SWEEP_PERIOD"):  # report the chain period instead of the kernel's own duration (same tuple layout)
        per = chain_period(fn, iters, warm_launches + iters)
        return nbytes, per, per, mn
    return nbytes, avg, med, mn


def dequant_case(lib, dev, k, n, bits, g, iters=100):
    nbytes = synth.dequant_bytes(k, n, bits, g)
    copies = max(2, (COLD_BYTES + nbytes - 1) // nbytes + 1)
    base = synth.make_problem(k, n, bits, g, m=1, seed=k + n + bits)
    qw = [(torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
    sc = torch.from_numpy(base.scales).to(dev)
    qz = torch.from_numpy(base.qzeros).to(dev)
    outs = [torch.empty((k, n), dtype=torch.float16, device=dev) for _ in range(copies)]
    sp = torch.cuda.current_stream(dev).cuda_stream

    def fn(i):
        c = i % copies
        st = lib.xbit_dequant_f16(outs[c].data_ptr(), qw[c].data_ptr(), sc.data_ptr(), qz.data_ptr(), bits, g, k, n, 1, sp)
        assert st == 0, st

    avg, med, mn = measure(lib, fn, iters)
    return nbytes, avg, med, mn


def show(name, nbytes, avg, med, mn):
    gbps = nbytes / (avg * 1e-6) / 1e9
    print("%-44s avg %8.2f  med %8.2f  min %8.2f us   %8.1f GB/s  %5.1f%% of 8TB/s  (min: %.1f%%)" % (
        name, avg, med, mn, gbps, gbps / 80.0, nbytes / (mn * 1e-6) / 1e9 / 80.0), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="-1", help="-1 = production launch (production library); > 0 = measurement variants (tuning library)")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shape", default="4096,11008,4,128,1", help="K,N,bits,group,M for the variant loop")
    ap.add_argument("--splits", default="0", help="split-K factors to try per variant (0 = automatic, 1 = never split)")
    ap.add_argument("--dequant-only", action="store_true", help="only the dequant cases (XBIT_DEQUANT_VARIANT selects a measurement variant: tuning library)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    print("device:", torch.cuda.get_device_name(0), flush=True)
    if args.dequant_only:
        dl = _lib.load_tuning() if os.environ.get("XBIT_DEQUANT_VARIANT") else lib
        tag = "variant " + os.environ.get("XBIT_DEQUANT_VARIANT", "0")
        for bits in (2, 4, 8):
            show("dequant w%d 4096x4096 %s cold" % (bits, tag), *dequant_case(dl, dev, 4096, 4096, bits, 128))
        for bits in (3, 4):
            show("dequant w%d 4096x11008 %s cold" % (bits, tag), *dequant_case(dl, dev, 4096, 11008, bits, 128))
        return
    sk, sn, sb, sg, sm = (int(x) for x in args.shape.split(","))
    for v in [int(x) for x in args.variants.split(",") if x != ""]:
        vlib = _lib.load_tuning() if v > 0 else lib  # measurement variants live in the tuning build only
        for sp in [int(x) for x in args.splits.split(",")]:
            tag = "variant %d" % v + (" split %d" % sp if sp else "")
            show("gemv w%d %dx%d m%d %s cold" % (sb, sk, sn, sm, tag), *gemv_case(vlib, dev, sk, sn, sb, sg, sm, True, 100 if sk * sn > 1e8 else 200, v, sp))
            show("gemv w%d %dx%d m%d %s warm" % (sb, sk, sn, sm, tag), *gemv_case(vlib, dev, sk, sn, sb, sg, sm, False, 100 if sk * sn > 1e8 else 200, v, sp))
    if args.quick:
        return
    show("gemv w4 8192x28672 m1 cold (cfg4, 1 GPU)", *gemv_case(lib, dev, 8192, 28672, 4, 128, 1, True, 60))
    show("gemv w4 8192x3584 m1 cold (cfg4 shard /8)", *gemv_case(lib, dev, 8192, 3584, 4, 128, 1, True))
    for bits in range(2, 9):
        show("gemv w%d 4096x4096 m1 cold (cfg3)" % bits, *gemv_case(lib, dev, 4096, 4096, bits, 128, 1, True))
    for m in (1, 4, 16):
        show("gemv w3 4096x11008 m%d cold (cfg5)" % m, *gemv_case(lib, dev, 4096, 11008, 3, 128, m, True))
    for m in (4, 16):
        show("gemv w4 4096x11008 m%d cold" % m, *gemv_case(lib, dev, 4096, 11008, 4, 128, m, True))
    for bits in range(2, 9):
        show("dequant w%d 4096x4096 cold" % bits, *dequant_case(lib, dev, 4096, 4096, bits, 128))
    show("dequant w4 4096x11008 cold", *dequant_case(lib, dev, 4096, 11008, 4, 128))


if __name__ == "__main__":
    main()
