#!/usr/bin/env python3
"""Measured tok/s proxy: the quantised linear layers of ONE decoder layer at batch 1 (q/k/v/o,
gate/up/down), replayed from a hipGraph over `--layers` distinct weight sets so that, as in a real
decode step, every launch streams its weights cold from HBM.

    python tools/decode_layer_bench.py [--model 7b|13b|70b] [--bits 4] [--layers 32] [--rows M] [--variant 398]

Prints us per decoder layer (linear layers only: no attention, norms or sampling), the
algorithmic GB/s over the layer, and tokens/s = 1 / (layers x t_layer).  GPU only.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, synth  # noqa: E402

MODELS = {  # hidden, intermediate, kv_dim, layers
    "7b": (4096, 11008, 4096, 32),
    "13b": (5120, 13824, 5120, 40),
    "70b": (8192, 28672, 1024, 80),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=sorted(MODELS))
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--layers", type=int, default=0, help="weight sets to rotate over (default: the model's layer count, capped by memory)")
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--fused", action="store_true", help="q/k/v in one launch and gate/up in one launch (xbit_gemv_multi_f16)")
    ap.add_argument("--rows", type=int, default=1, help="activation rows per call (batch / speculative decode)")
    ap.add_argument("--variant", type=int, default=-1, help="tuning library: launch-rule variant (398 = round 2's rules) for A/B")
    args = ap.parse_args()
    hid, inter, kv, nlayers = MODELS[args.model]
    shapes = [("q", hid, hid), ("k", hid, kv), ("v", hid, kv), ("o", hid, hid), ("gate", hid, inter), ("up", hid, inter), ("down", inter, hid)]
    dev = torch.device("cuda", 0)
    lib = _lib.load() if args.variant <= 0 else _lib.load_tuning()
    opts = _lib.GemvOptions(variant=args.variant) if args.variant > 0 else None
    m = args.rows
    layer_bytes = sum(synth.algorithmic_bytes(k, n, args.bits, args.group, m) for _, k, n in shapes)
    layers = args.layers or nlayers
    layers = max(2, min(layers, int(40e9 // layer_bytes)))
    sets = []
    for name, k, n in shapes:
        base = synth.make_problem(k, n, args.bits, args.group, m=m, seed=sum(map(ord, name)))
        qw0, sc, qz = (torch.from_numpy(x).to(dev) for x in (base.qweight, base.scales, base.qzeros))
        act = torch.from_numpy(base.act).to(dev)
        out = torch.empty((m, n), dtype=torch.float16, device=dev)
        qws = [(qw0 ^ (0x01010101 * (i % 127))).contiguous() for i in range(layers)]
        sets.append((k, n, act, out, qws, sc, qz))
    torch.cuda.synchronize()

    import ctypes

    vp = ctypes.c_void_p

    def fused(group, i, sp):
        c = len(group)
        k, act = group[0][0], group[0][2]
        st = lib.xbit_gemv_multi_f16(c, (vp * c)(*[g[3].data_ptr() for g in group]), act.data_ptr(), (vp * c)(*[g[4][i].data_ptr() for g in group]),
                                     (vp * c)(*[g[5].data_ptr() for g in group]), (vp * c)(*[g[6].data_ptr() for g in group]),
                                     (ctypes.c_uint32 * c)(*[g[1] for g in group]), args.bits, args.group, m, k, 1, sp)
        if st != 0:
            _lib.check(st, "xbit_gemv_multi_f16")

    def run_layer(i, sp):
        if args.fused:  # sets: q k v o gate up down
            fused(sets[0:3], i, sp)
            fused(sets[3:4], i, sp)
            fused(sets[4:6], i, sp)
            fused(sets[6:7], i, sp)
            return
        for k, n, act, out, qws, sc, qz in sets:
            if opts is None:
                st = lib.xbit_gemv_f16(out.data_ptr(), act.data_ptr(), qws[i].data_ptr(), sc.data_ptr(), qz.data_ptr(), args.bits, args.group, m, k, n, 1, sp)
            else:
                st = lib.xbit_gemv_f16_ex(out.data_ptr(), act.data_ptr(), qws[i].data_ptr(), sc.data_ptr(), qz.data_ptr(), args.bits, args.group, m, k, n, 1, sp, opts)
            if st != 0:
                _lib.check(st, "xbit_gemv_f16")

    sp = torch.cuda.current_stream(dev).cuda_stream
    for i in range(layers):  # warm-up (also allocates the split-K workspace outside the capture)
        run_layer(i, sp)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sp = torch.cuda.current_stream(dev).cuda_stream
        for i in range(layers):
            run_layer(i, sp)
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.replays):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (args.replays * layers)
    if args.fused:
        print("FUSED: q/k/v and gate/up share a launch (4 launches per layer)")
    print("model %s w%d g%d, %d row%s%s: %d linear layers per layer, %.1f MB per layer, %d weight sets (%.1f GB)" % (
        args.model, args.bits, args.group, m, "" if m == 1 else "s", "" if args.variant <= 0 else " (variant %d)" % args.variant, len(shapes),
        layer_bytes / 1e6, layers, layers * layer_bytes / 1e9))
    print("  %.2f us per decoder layer (linear layers, hipGraph replay, cold weights)  = %.0f GB/s algorithmic = %.1f %% of 8 TB/s" % (
        dt * 1e6, layer_bytes / dt / 1e9, layer_bytes / dt / 8e12 * 100))
    print("  tok/s proxy = 1 / (%d layers x %.2f us) = %.0f tokens/s" % (nlayers, dt * 1e6, 1.0 / (nlayers * dt)))
    for (name, k, n), (_k, _n, act, out, qws, sc, qz) in zip(shapes, sets):
        lib.xbit_profile_enable(1)
        import ctypes
        ms, vals = ctypes.c_float(), []
        for i in range(3 * layers):
            if opts is None:
                lib.xbit_gemv_f16(out.data_ptr(), act.data_ptr(), qws[i % layers].data_ptr(), sc.data_ptr(), qz.data_ptr(), args.bits, args.group, m, k, n, 1,
                                  torch.cuda.current_stream(dev).cuda_stream)
            else:
                lib.xbit_gemv_f16_ex(out.data_ptr(), act.data_ptr(), qws[i % layers].data_ptr(), sc.data_ptr(), qz.data_ptr(), args.bits, args.group, m, k, n, 1,
                                     torch.cuda.current_stream(dev).cuda_stream, opts)
            lib.xbit_profile_last_ms(ctypes.byref(ms))
            vals.append(ms.value * 1e3)
        lib.xbit_profile_enable(0)
        vals = vals[layers:]
        print("    %-5s %6d x %-6d kernel %.2f us" % (name, k, n, sum(vals) / len(vals)))


if __name__ == "__main__":
    main()
