#!/usr/bin/env python3
"""Act-order (g_idx) GEMV against the contiguous-group GEMV on the same shape: period of a dependent chain of launches,
cold weights.  Two kinds of g_idx: a random permutation of equal-sized groups (what GPTQ act-order writes; QuantLinear sorts
such a checkpoint at load time and never calls the general entry for it) and unequal groups (the general entry's own case).
    python tools/gidx_bench.py [--shape K,N] [--bits 4] [--rows 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from xbitops_amd import _lib, synth  # noqa: E402
import gemv_sweep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096,11008")
    ap.add_argument("--bits", default="4")
    ap.add_argument("--rows", default="1")
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--dequant", action="store_true", help="time xbit_dequant_gidx_f16 against xbit_dequant_f16 instead")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    k, n = (int(x) for x in args.shape.split(","))
    g = args.group
    groups = (k + g - 1) // g
    sp = torch.cuda.current_stream(dev).cuda_stream
    rng = np.random.default_rng(3)
    equal = np.repeat(np.arange(groups), g)[:k].astype(np.int32)
    rng.shuffle(equal)
    unequal = rng.integers(0, groups, size=k).astype(np.int32)
    if args.dequant:
        print("# %dx%d group %d: dequant, us per launch, period of a dependent chain" % (k, n, g))
        for bits in (int(b) for b in args.bits.split(",")):
            base = synth.make_problem(k, n, bits, g, m=1, seed=bits)
            qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (base.qweight, base.scales, base.qzeros))
            out = torch.empty((k, n), dtype=torch.float16, device=dev)
            gd = torch.from_numpy(unequal).to(dev)

            def plain(i):
                assert lib.xbit_dequant_f16(out.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), bits, g, k, n, 1, sp) == 0

            def by_id(i):
                assert lib.xbit_dequant_gidx_f16(out.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), gd.data_ptr(), bits, groups, k, n, 1, sp) == 0

            for f in (plain, by_id):
                for i in range(5):
                    f(i)
            torch.cuda.synchronize()
            print("w%d  contiguous groups %7.2f us   g_idx %7.2f us" % (bits, gemv_sweep.chain_period(plain, 50, 0), gemv_sweep.chain_period(by_id, 50, 0)), flush=True)
        return
    print("# %dx%d group %d: us per launch, period of a dependent chain, cold" % (k, n, g))
    for bits in (int(b) for b in args.bits.split(",")):
        for m in (int(r) for r in args.rows.split(",")):
            nbytes = synth.algorithmic_bytes(k, n, bits, g, m)
            copies = (512 << 20) // nbytes + 2
            base = synth.make_problem(k, n, bits, g, m=m, seed=bits)
            act = torch.from_numpy(base.act).to(dev)
            qw = [(torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))).contiguous() for c in range(copies)]
            sc = [torch.from_numpy(base.scales).to(dev).clone() for _ in range(copies)]
            qz = [torch.from_numpy(base.qzeros).to(dev).clone() for _ in range(copies)]
            out = torch.empty((m, n), dtype=torch.float16, device=dev)
            row = []
            for name, gi in (("contiguous groups (xbit_gemv_f16)", None), ("g_idx: shuffled equal groups", equal), ("g_idx: unequal groups", unequal)):
                gd = None if gi is None else torch.from_numpy(gi).to(dev)

                def fn(i, gd=gd):
                    c = i % copies
                    if gd is None:
                        st = lib.xbit_gemv_f16(out.data_ptr(), act.data_ptr(), qw[c].data_ptr(), sc[c].data_ptr(), qz[c].data_ptr(), bits, g, m, k, n, 1, sp)
                    else:
                        st = lib.xbit_gemv_gidx_f16(out.data_ptr(), act.data_ptr(), qw[c].data_ptr(), sc[c].data_ptr(), qz[c].data_ptr(), gd.data_ptr(), bits, groups,
                                                    m, k, n, 1, sp)
                    assert st == 0, st

                for i in range(2 * copies):
                    fn(i)
                torch.cuda.synchronize()
                row.append((name, gemv_sweep.chain_period(fn, 100, 0)))
            print("w%d rows %2d  " % (bits, m) + "   ".join("%s %6.2f" % r for r in row) + "   (ratio %.2f / %.2f)" % (row[1][1] / row[0][1], row[2][1] / row[0][1]), flush=True)


if __name__ == "__main__":
    main()
