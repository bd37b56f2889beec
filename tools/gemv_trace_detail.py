#!/usr/bin/env python3
"""Where the spread in the GEMV's per-wave timeline comes from: the stamped measurement builds (variant 63: the kernel of
xbit_gemv_mfma.h; variant 320: the staged-operand kernel of xbit_gemv_dyn.h; 4-bit, one row) broken down by wave index inside
the workgroup, by workgroup, by XCD and by SIMD.  --chain N: N back-to-back launches before the traced one (no idle gap).  Every weight copy is touched
once before the traced launches (a first touch pays page-table walks), the traced launches rotate over > 512 MB of
copies (cold HBM), and several launches are read back one by one.
    python tools/gemv_trace_detail.py [--shape K,N] [--launches 4]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth  # noqa: E402

NAMES = ["wave start", "loads issued", "act staged", "last MFMA issued", "past barrier", "wave end"]


def pct(x, qs=(0, 10, 50, 90, 100)):
    return "  ".join("%5.2f" % v for v in np.percentile(x, qs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096,11008", help="K,N")
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--variant", type=int, default=63)
    ap.add_argument("--chain", type=int, default=0, help="launches queued right in front of the traced one")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load_tuning()
    k, n = [int(x) for x in args.shape.split(",")]
    bits, g, nw, cn = 4, 128, 8, 48
    nbytes = synth.algorithmic_bytes(k, n, bits, g, 1)
    copies = (512 << 20) // nbytes + 2
    base = synth.make_problem(k, n, bits, g, m=1, seed=1)
    qw = [(torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
    sc = [torch.from_numpy(base.scales).to(dev).clone() for _ in range(copies)]
    qz = [torch.from_numpy(base.qzeros).to(dev).clone() for _ in range(copies)]
    act = torch.from_numpy(base.act).to(dev)
    out = torch.empty((1, n), dtype=torch.float16, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    tiles = (n + cn - 1) // cn
    trace = torch.zeros((tiles * nw, 8), dtype=torch.int64, device=dev)
    opts = _lib.GemvOptions(variant=args.variant, trace=trace.data_ptr())

    def launch(c):
        st = lib.xbit_gemv_f16_ex(out.data_ptr(), act.data_ptr(), qw[c].data_ptr(), sc[c].data_ptr(), qz[c].data_ptr(), bits, g, 1, k, n, 1, sp, opts)
        assert st == 0, st

    for c in range(copies):  # first touch of every copy
        launch(c)
    for c in range(copies):
        launch(c)
    torch.cuda.synchronize()
    names = list(NAMES)
    if args.variant == 320:
        names[1], names[2] = "staged (pre-barrier)", "past staging barrier"
    for it in range(args.launches):
        for j in range(args.chain):
            launch((7 * it + 3 + j) % copies)
        launch((7 * it + 3 + args.chain) % copies)
        torch.cuda.synchronize()
        raw = trace.cpu().numpy()
        t = raw[:, :6].astype(np.float64)
        t = (t - t[:, 0].min()) * 0.01  # 100 MHz ticks -> us
        xcc = ((raw[:, 7] >> 32) & 0xF).astype(int)
        hw = (raw[:, 7] & 0xFFFFFFFF).astype(np.int64)
        simd = ((hw >> 4) & 3).astype(int)
        cu = ((hw >> 8) & 15).astype(int)
        sh = ((hw >> 12) & 1).astype(int)
        se = ((hw >> 13) & 7).astype(int)
        tw = t.reshape(tiles, nw, 6)
        print("== launch %d (variant %d, %dx%d): us after the first wave started; min p10 p50 p90 max" % (it, args.variant, k, n))
        for j, name in enumerate(names):
            print("  %-22s %s" % (name, pct(t[:, j])))
        print("  %-22s %s" % ("first rows requested", pct((raw[:, 6].astype(np.float64) - raw[:, 0].min()) * 0.01)))
        print("  -- by wave index in the workgroup (mean over workgroups): loads issued / last MFMA / own start")
        for w in range(nw):
            print("     wave %d  %5.2f  %5.2f  %5.2f   simd histogram %s" % (w, tw[:, w, 1].mean(), tw[:, w, 3].mean(), tw[:, w, 0].mean(),
                  np.bincount(simd.reshape(tiles, nw)[:, w], minlength=4).tolist()))
        last = tw[:, :, 3]
        print("  -- per workgroup: last-MFMA of its waves: mean %s" % pct(last.mean(axis=1)))
        print("                                              max  %s" % pct(last.max(axis=1)))
        print("                                         max-mean  %s" % pct(last.max(axis=1) - last.mean(axis=1)))
        print("                                          max-min  %s" % pct(last.max(axis=1) - last.min(axis=1)))
        print("     slowest wave index histogram: %s" % np.bincount(last.argmax(axis=1), minlength=nw).tolist())
        print("     fastest wave index histogram: %s" % np.bincount(last.argmin(axis=1), minlength=nw).tolist())
        wg_end = tw[:, :, 5].max(axis=1)
        wg_start = tw[:, :, 0].min(axis=1)
        wg_xcc = xcc.reshape(tiles, nw)[:, 0]
        print("  -- per XCD: workgroups, mean start, mean / max end")
        for x in range(8):
            sel = wg_xcc == x
            if sel.any():
                print("     xcc %d  %3d  %5.2f  %5.2f  %5.2f" % (x, sel.sum(), wg_start[sel].mean(), wg_end[sel].mean(), wg_end[sel].max()))
        print("     corr(workgroup start, workgroup end) = %.2f" % np.corrcoef(wg_start, wg_end)[0, 1])
        cu_key = (xcc * 1024 + se * 64 + sh * 16 + cu).reshape(tiles, nw)[:, 0]
        uniq, cnt = np.unique(cu_key, return_counts=True)
        print("     distinct CUs used: %d (workgroups per CU histogram %s)" % (len(uniq), np.bincount(cnt).tolist()))
        # the two waves of a SIMD: does the later-started one finish later?
        sim = simd.reshape(tiles, nw)
        d_first, d_second = [], []
        for wg in range(tiles):
            for s in range(4):
                idx = np.where(sim[wg] == s)[0]
                if len(idx) == 2:
                    a, b = idx if tw[wg, idx[0], 0] <= tw[wg, idx[1], 0] else idx[::-1]
                    d_first.append(tw[wg, a, 3])
                    d_second.append(tw[wg, b, 3])
        if d_first:
            print("     SIMD partners: last MFMA of the earlier-started wave %.2f, of the later-started %.2f (mean, %d pairs)" % (np.mean(d_first), np.mean(d_second), len(d_first)))


if __name__ == "__main__":
    main()
