#!/usr/bin/env python3
"""Per-wave timeline of the GEMV kernel (stamped measurement builds: variant 63 = 1 row, 193 = 4 rows, 189 = 16 rows): when did each
wave start, finish issuing its loads, finish staging, issue its last MFMA, pass the barrier, end.
    python tools/gemv_trace.py [--shape K,N]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth  # noqa: E402

NAMES = ["wave start", "loads issued", "act staged", "last MFMA issued", "past barrier", "wave end"]
NW = {63: 8, 189: 8, 193: 8}
CN = {63: 48, 189: 48, 193: 48}
ROWS = {63: 1, 189: 16, 193: 4}  # activation rows of the stamped build


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="63")
    ap.add_argument("--shape", default="4096,11008", help="K,N")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load_tuning()  # the stamped kernel lives in the measurement build only
    k, n = [int(x) for x in args.shape.split(",")]
    bits, g = 4, 128
    nbytes = synth.algorithmic_bytes(k, n, bits, g, 1)
    copies = (512 << 20) // nbytes + 2
    base = synth.make_problem(k, n, bits, g, m=1, seed=1)
    qw = [(torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
    sc = torch.from_numpy(base.scales).to(dev)
    qz = torch.from_numpy(base.qzeros).to(dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    for v in [int(x) for x in args.variants.split(",")]:
        m = ROWS[v]
        act = torch.from_numpy(synth.make_problem(k, n, bits, g, m=m, seed=1).act).to(dev)
        out = torch.empty((m, n), dtype=torch.float16, device=dev)
        waves = ((n + CN[v] - 1) // CN[v]) * NW[v]
        trace = torch.zeros((waves, 8), dtype=torch.int64, device=dev)
        opts = _lib.GemvOptions(variant=v, trace=trace.data_ptr())
        for mode in ("cold", "warm"):
            for i in range(12):
                c = i % copies if mode == "cold" else 0
                st = lib.xbit_gemv_f16_ex(out.data_ptr(), act.data_ptr(), qw[c].data_ptr(), sc.data_ptr(), qz.data_ptr(), bits, g, m, k, n, 1, sp, opts)
                assert st == 0
            torch.cuda.synchronize()
            t = trace.cpu().numpy()[:, :6].astype(np.float64)
            t = (t - t[:, 0].min()) * 0.01  # 100 MHz ticks -> us
            print("variant %d %s: %d waves; us after the first wave started" % (v, mode, waves))
            for j, name in enumerate(NAMES):
                d = np.sort(t[:, j])
                print("  %-18s min %5.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" % (name, d[0], d[len(d) // 10], d[len(d) // 2], d[len(d) * 9 // 10], d[-1]))
            dur = t[:, 3] - t[:, 1]
            print("  per-wave (last MFMA - loads issued): p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(dur, [50, 90, 100])))


if __name__ == "__main__":
    main()
