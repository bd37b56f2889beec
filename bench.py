#!/usr/bin/env python3
"""bench.py — A16W4 GEMV throughput on MI355X, the metric BASELINE.json names.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg4] [--shape K,N,bits,g,M]

One "step" = one call of the hot path (xbit_gemv_f16 through the C ABI) plus, on N > 1 GPUs, the ONE
all-gather of the fp16 output shards that a column-sharded layer needs before its result can be
used (issued per step on the compute stream: no bucketing, a decoder cannot defer it).

Workloads
  cfg2 (default)  BASELINE.json configs[1]: in=4096 out=11008 4-bit g128 batch 1 per GPU (Llama-7B
                  FFN).  N > 1: weak scaling, every rank owns one 4096x11008 column shard of a
                  4096 x (11008*N) layer.
  cfg4            BASELINE.json configs[3]: in=8192 out=28672 (Llama-70B FFN) column-sharded N ways
                  with xbitops_amd.sharded.shard_columns — strong scaling; N = 1 is the whole layer.
  --shape         any other K,N,bits,group,M on one GPU (used to profile configs 3 and 5).
Weights rotate over enough distinct copies (>= 512 MB in total, twice the 256 MB Infinity Cache) that
every timed launch streams from HBM ("cold").

Printed JSON (one line, rank 0)
  value / ms_per_step  algorithmic bytes of all ranks x K / the time of exactly K steps, measured with
                       HIP events recorded ON THE COMPUTE STREAM around the region,
                       max over ranks, between barrier + synchronize pairs.  On one GPU the W warm-up
                       launches and the K timed ones are queued, in that order, behind a blocker kernel that
                       outlasts the host's enqueue time, so the span holds device time only: K x (kernel +
                       launch boundary, `boundary_us_per_step`), and the timed steps start on a chip that is
                       already streaming; `host_clock_ms_per_step` is the host's view (blocker included).
                       N > 1 on RCCL: the K x (GEMV + all-gather) steps are one captured hipGraph (the W warm-up
                       steps another, replayed directly ahead); `collective_us_per_step` = step - kernel;
                       `graph_note` says so when the capture failed and the region ran eagerly.
  roofline             algorithmic bytes per launch / mean KERNEL duration over >= 200 cold launches
                       (whatever --steps says), each bracketed by its own pair of HIP events on its own stream
                       (hipExtLaunchKernel via xbit_profile_series_begin/_end) and queued back to back like the
                       launches of the timed region; `isolated_launch_avg_kernel_us` is the same kernel launched on
                       an idle chip (host wait after every launch).  peak = 8000 GB/s (HBM3E spec).
                       `traffic` = HBM bytes per launch from the rocprofv3 PMC passes under profiles/
                       when they were taken from THESE kernel sources AND this very library file (src_sha256 and
                       lib_sha256 both match), else null.  `frac_isolated` is the same fraction by the timer of rounds
                       1-2 (a host wait after every launch).  `read_floor_us` (one GPU): the period of a dependent chain
                       of PURE READS of the same weight copies, measured in this very run through the measurement
                       build's xbit_tuning_read_floor - `flat` (16-byte stream) and `tiles` (the column tiles the GEMV
                       walks, same lane map, same workgroup order) - next to `kernel_chain_us`, the GEMV's own period
                       measured the same way: what the memory system charges for the bytes alone on this box.
  output_check         (one GPU, with the CPU leg) the output of the LAST timed step against the oracle under the GEMV
                       contract (oracle/contract.py); dequant: bit-exact.
  cpu_baseline         (N = 1) the oracle's C restatement (fused dequant + GEMV, OpenMP) for ~10 s, and
                       the torch-CPU path (int32 shift/mask unpack -> fp16 W -> a @ W) timed as
                       unpack + matmul and matmul alone, thread counts stated.  Reported, not a target.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling
COLD_BYTES = 512 << 20
MIN_ROOFLINE_LAUNCHES = 200
CONFIGS = {
    # name: (K, N, bits, group, M, scaling, workload text)
    "cfg2": (4096, 11008, 4, 128, 1, "weak", "A16W4 GEMV in=4096 out=11008 group=128 batch=1 (Llama-7B FFN, BASELINE.json configs[1]) per GPU"),
    "cfg4": (8192, 28672, 4, 128, 1, "strong", "A16W4 GEMV in=8192 out=28672 group=128 batch=1 (Llama-70B FFN, BASELINE.json configs[3]) column-sharded over the GPUs"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--shape", default="", help="K,N,bits,group,M: another single-GPU workload (profiling configs 3 and 5)")
    ap.add_argument("--op", choices=("gemv", "dequant"), default="gemv", help="dequant: time xbit_dequant_f16 on the --shape instead (write-bound; SURVEY 8f rank 1)")
    ap.add_argument("--graph", action="store_true", help="one GPU: time a hipGraph replay of the K launches instead of queued eager launches")
    ap.add_argument("--no-graph", action="store_true", help="N > 1: time eager launches + eager collectives instead of one captured hipGraph of K x (GEMV + all-gather)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--roofline-launches", type=int, default=MIN_ROOFLINE_LAUNCHES, help="event-bracketed cold launches (at least 200)")
    ap.add_argument("--warm", action="store_true", help="also report the cache-resident kernel time (one weight copy)")
    ap.add_argument("--variant", type=int, default=-1, help="tuning builds only: GEMV launch-shape variant")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, one GPU).  The oracle may be used here and only here (it is the thing timed).
def cpu_baseline_port(seconds: float, k: int, n: int, bits: int, g: int, m: int, nbytes: int):
    import numpy as np

    import oracle
    from xbitops_amd import synth

    oracle.set_num_threads(min(16, os.cpu_count() or 1))  # the GPU box grants a 16-CPU share per GPU
    p = synth.make_problem(k, n, bits, g, m=m, seed=7)
    oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1)  # warm (page in, spin up threads)
    reps, t0 = 0, time.perf_counter()
    while True:
        y = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and reps >= 3:
            break
    assert np.isfinite(y.astype(np.float32)).all()
    return {
        "value": round(nbytes * reps / dt / 1e9, 4), "unit": "GB/s", "cores": oracle.num_threads(), "kind": "port",
        "sample": "%d full-size %dx%d w%d g%d batch-%d GEMVs in %.1f s (oracle/xbit_oracle.c xo_gemv, fused dequant + GEMV, OpenMP; host has %d logical CPUs)"
                  % (reps, k, n, bits, g, m, dt, os.cpu_count()),
        "ms_per_gemv": round(1e3 * dt / reps, 3),
    }, p, y


def torch_cpu_unpack(qweight, scales, qzeros, bits: int, g: int, k: int, add_zero_bias: int):
    """The plumbing path of BASELINE.json config 1 in stock torch on the CPU: int32 shift/mask unpack of
    the LSB-first bit-streams -> fp16 W = q*s - (z + bias)*s (each product rounded to fp16)."""
    import torch

    kw, n = qweight.shape
    mask = (1 << bits) - 1
    kk = torch.arange(k, dtype=torch.int64)
    bit = kk * bits
    wi, sh = bit >> 5, (bit & 31).to(torch.int32)
    qw64 = qweight.to(torch.int64) & 0xFFFFFFFF
    lo = qw64[wi] >> sh.to(torch.int64)[:, None]
    hi_rows = torch.clamp(wi + 1, max=kw - 1)
    hi = qw64[hi_rows] << (32 - sh.to(torch.int64))[:, None]
    q = ((lo | hi) & mask).to(torch.float16)  # [K, N]
    nn = torch.arange(n, dtype=torch.int64)
    zbit = nn * bits
    zi, zs = zbit >> 5, zbit & 31
    qz64 = qzeros.to(torch.int64) & 0xFFFFFFFF
    zlo = qz64[:, zi] >> zs[None, :]
    zhi = qz64[:, torch.clamp(zi + 1, max=qzeros.shape[1] - 1)] << (32 - zs)[None, :]
    z = (((zlo | zhi) & mask) + add_zero_bias).to(torch.float16)  # [G, N]
    zsc = z * scales  # fp16 product, rounded (hmul)
    gi = kk // g
    return q * scales[gi] - zsc[gi]  # two roundings here (mul, sub) where the kernels fuse: a CPU baseline, not the checker


def cpu_baseline_torch(p, y_port, bits: int, g: int, k: int, nbytes: int, budget_s: float):
    import numpy as np
    import torch

    qw, sc, qz = torch.from_numpy(p.qweight), torch.from_numpy(p.scales), torch.from_numpy(p.qzeros)
    a = torch.from_numpy(p.act)
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    w = torch_cpu_unpack(qw, sc, qz, bits, g, k, 1)
    t_unpack = time.perf_counter() - t0
    (a @ w)  # warm
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < min(2.0, budget_s) and reps < 200):
        y = a @ w
        reps += 1
    t_mm = (time.perf_counter() - t0) / reps
    ref = y_port.astype(np.float32)
    err = np.abs(y.float().numpy() - ref)
    assert (err <= 5e-3 * np.abs(ref) + 2e-3).all(), "torch-CPU fallback disagrees with the C restatement"  # fp16 GEMM + unfused W
    return {
        "threads": threads, "unpack_ms": round(1e3 * t_unpack, 2), "matmul_only_ms": round(1e3 * t_mm, 3),
        "unpack_plus_matmul_ms": round(1e3 * (t_unpack + t_mm), 2),
        "gbps_unpack_plus_matmul": round(nbytes / (t_unpack + t_mm) / 1e9, 4),
        "gbps_matmul_only_fp16_weights": round((2 * w.numel() + 2 * a.numel() + 2 * y.numel()) / t_mm / 1e9, 3),
        "what": "stock torch %s on the host: int32 shift/mask unpack -> fp16 W (once), then a.half() @ W averaged over %d calls" % (torch.__version__, reps),
    }


def check_last_output(y_gpu, copy: int, qw_h, sc_h, qz_h, act_h, op: str, bits: int, g: int, k: int):
    """The output of the last timed step (weight copy `copy`) against the oracle - the checker, on the host, after the timed
    region.  Raises if the timed launches did not compute what they are credited for."""
    import numpy as np

    import oracle
    from oracle import contract

    qw = (qw_h.view(np.uint32) ^ np.uint32(0x01010101 * (copy % 127))).view(np.int32)  # the copy's contents (see `qws` in main)
    qz = (qz_h.view(np.uint32) ^ np.uint32(0x11111111 * (copy % 13))).view(np.int32)
    if op == "dequant":
        w_ref = oracle.dequant(qw, sc_h, qz, g, bits, k, 1, oracle.RNE)
        same = np.array_equal(np.ascontiguousarray(y_gpu).view(np.uint16), w_ref.view(np.uint16))
        if not same:
            raise SystemExit("bench.py: the last timed dequant is NOT bit-exact against the oracle")
        return {"checked": "last timed step (weight copy %d), dequant, bit-exact against oracle.dequant (RNE)" % copy, "ok": True}
    y_ref = oracle.gemv(act_h, qw, sc_h, qz, g, bits, k, 1, oracle.RNE)
    why = contract.gemv_mismatch(np.ascontiguousarray(y_gpu), y_ref, k)
    if why is not None:
        raise SystemExit("bench.py: the last timed GEMV does not match the oracle: " + why)
    err = np.abs(y_gpu.astype(np.float64) - y_ref.astype(np.float64))
    return {"checked": "last timed step (weight copy %d) against oracle.gemv (RNE) under oracle/contract.py" % copy, "ok": True,
            "max_abs_err": float(err.max()), "outputs": int(y_ref.size)}


# ------------------------------------------------------------------------------------------------
def lib_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def source_sha256() -> str:
    """Identity of the native code that is running: the kernel sources and the flags they are compiled with.  (The bytes of
    the built library are not one: a box that rebuilds it - tests/conftest.py does when the object files did not travel -
    embeds its own paths.)"""
    from xbitops_amd import build as hip_build

    h = hashlib.sha256()
    for name in sorted(os.listdir(hip_build.CSRC)):
        path = os.path.join(hip_build.CSRC, name)
        if os.path.isfile(path) and name.endswith((".h", ".hip", ".cc")):
            h.update(name.encode())
            with open(path, "rb") as f:
                h.update(f.read())
    with open(os.path.join(ROOT, "include", "xbitops_hip.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(hip_build.CXXFLAGS).encode())
    return h.hexdigest()


def offline_profile(workload_key: str, this_lib_sha: str):
    """rocprofv3 numbers committed under profiles/rNN/ for this workload (collected offline: counters
    cannot be read in-process).  `matches_this_build` says whether they were taken from the kernel sources
    that are running now (source_sha256); only then do they appear in `roofline`."""
    best = None
    for root, _dirs, files in sorted(os.walk(os.path.join(ROOT, "profiles"))):
        if "pmc_traffic.json" not in files:
            continue
        path = os.path.join(root, "pmc_traffic.json")
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("workload") != workload_key:
            continue
        cand = dict(rec, source=os.path.relpath(path, ROOT), matches_this_build=rec.get("src_sha256") == this_lib_sha)
        if best is None or cand["matches_this_build"] or not best["matches_this_build"]:
            best = cand  # later rounds win; a record of this very build wins over everything
    return best


def floor_tile_shape(lib, bits: int, group: int, batch: int, k: int, n: int):
    """The column tiles the library's launch for this call walks, read back from its dry run (xbit_dry_run_begin / _end: the
    launch rule runs, nothing is issued): lane-map width for the pure-read floor, or None when the launch is not one grid of
    whole-K column tiles of 32 / 48 / 64 / 128 columns (split-K plans, 16-column tiles, several launches) - the floor kernels
    have no such pattern, and a floor that does not mirror the launch would say nothing about it."""
    import re

    from xbitops_amd import _lib

    fake = 0x10000000  # never followed in a dry run

    def call():
        _lib.check(lib.xbit_gemv_f16(fake, fake, fake, fake, fake, bits, group, batch, k, n, 1, None), "xbit_gemv_f16 (dry run)")

    recs = _lib.dry_run(lib, call)
    if len(recs) != 1:
        return None
    m = re.search(r"gemv_(dyn|mfma)_kernel<([^>]*)>", recs[0]["kernel"])
    if not m:
        return None
    a = [x.strip() for x in m.group(2).split(",")]
    cpl, nl = int(a[2]), int(a[3])
    split = a[10] == "true"  # SPLIT is the 11th template argument of both kernels
    if split or nl not in (16, 32) or (nl == 32 and cpl != 4):
        return None
    return nl * cpl


def main(argv=None):
    args = parse_args(argv)
    import torch
    import torch.distributed as dist

    from xbitops_amd import _lib, sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU; there is no CPU fallback for the measured path")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev)  # one rank per GPU; the modulo only matters for the 1-GPU rehearsal below
    torch.cuda.set_device(dev)
    class _StdoutToStderr:
        """RCCL prints a version banner on STDOUT when a communicator is created; stdout carries the ONE JSON line."""

        def __enter__(self):
            sys.stdout.flush()
            self._saved = os.dup(1)
            os.dup2(2, 1)

        def __exit__(self, *exc):
            sys.stdout.flush()
            os.dup2(self._saved, 1)
            os.close(self._saved)

    backend = "none"
    force_pg = world == 1 and os.environ.get("XBIT_BENCH_FORCE_PG") == "1"  # rehearsal: a ONE-rank RCCL group, so that the captured (GEMV + all-gather) path runs on a 1-GPU box
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            dist.all_reduce(torch.zeros(1, device=dev))  # the communicator is created at the first collective
            torch.cuda.synchronize()
        backend = "nccl"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # nccl == RCCL on ROCm.  XBIT_BENCH_BACKEND=gloo rehearses the multi-rank control flow with
        # several ranks on ONE GPU (RCCL refuses two ranks per device); never used for reported numbers.
        backend = os.environ.get("XBIT_BENCH_BACKEND", "nccl")
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.all_reduce(torch.zeros(1, device=dev if backend == "nccl" else "cpu"))  # creates the communicator (and its banner) now
            if backend == "nccl":
                torch.cuda.synchronize()

    lib = _lib.load()  # raises if the HIP library is missing
    gemv_opts = None
    if args.variant > 0:
        lib = _lib.load_tuning()  # measurement variants live in the tuning build (python -m xbitops_amd.build --tuning)
        gemv_opts = _lib.GemvOptions(variant=args.variant)

    # ---- workload
    if args.shape:
        if world > 1:
            raise SystemExit("--shape is a single-GPU workload")
        k_in, n_full, bits, group, batch = (int(x) for x in args.shape.split(","))
        scaling, workload = "weak", "A16W%d GEMV in=%d out=%d group=%d batch=%d (--shape)" % (bits, k_in, n_full, group, batch)
        cfg_name = "shape"
    else:
        k_in, n_full, bits, group, batch, scaling, workload = CONFIGS[args.config]
        cfg_name = args.config
    K, W = args.steps, args.warmup

    # ---- synthetic, seeded inputs resident in HBM before any timing
    if cfg_name == "cfg4":
        # strong scaling: ONE 8192x28672 layer (same seed on every rank), this rank keeps its column shard
        full = synth.make_problem(k_in, n_full, bits, group, m=batch, seed=4000)
        qw_h, sc_h, qz_h = sharded.shard_columns(full.qweight, full.scales, full.qzeros, bits, world, rank)
        act_h = full.act
        del full
    else:
        # weak scaling: every rank owns one whole shard of a layer N times as wide (or the only one)
        base = synth.make_problem(k_in, n_full, bits, group, m=batch, seed=1000 + rank)
        qw_h, sc_h, qz_h, act_h = base.qweight, base.scales, base.qzeros, synth.make_problem(k_in, 2, bits, group, m=batch, seed=999).act
        del base
    n_local = qw_h.shape[1]
    nbytes = synth.algorithmic_bytes(k_in, n_local, bits, group, batch)  # per rank and step
    if args.op == "dequant":
        if world > 1 or not args.shape:
            raise SystemExit("--op dequant is a single-GPU --shape workload")
        nbytes = synth.dequant_bytes(k_in, n_local, bits, group)
        workload = "dequant W%d in=%d out=%d group=%d (--op dequant)" % (bits, k_in, n_full, group)
    copies = (COLD_BYTES + nbytes - 1) // nbytes + 1  # >= 512 MB of distinct weights per GPU
    act = torch.from_numpy(act_h).to(dev)
    qw0, sc0, qz0 = torch.from_numpy(qw_h).to(dev), torch.from_numpy(sc_h).to(dev), torch.from_numpy(qz_h).to(dev)
    qws, scs, qzs = [], [], []
    for c in range(copies):
        # distinct contents per copy (xor with a per-copy word) so no two copies can be deduplicated
        qws.append((qw0 ^ (0x01010101 * (c % 127))).contiguous())
        scs.append(sc0.clone())
        qzs.append((qz0 ^ (0x11111111 * (c % 13))).contiguous())
    n_out_bufs = 4
    if args.op == "dequant":  # rotate the outputs too: [K, N] fp16 each, > 512 MB in total so that no write stays cached
        n_out_bufs = max(2, min(copies, (COLD_BYTES + 2 * k_in * n_local - 1) // (2 * k_in * n_local) + 1))
    outs = torch.empty((n_out_bufs, k_in if args.op == "dequant" else batch, n_local), dtype=torch.float16, device=dev)
    collective = world > 1 or force_pg  # every step ends with the all-gather of its output shards
    gathered = torch.empty((n_out_bufs, world, batch, n_local), dtype=torch.float16, device=dev) if collective else None
    torch.cuda.synchronize()

    a_ptr = act.data_ptr()
    ptrs = [(qws[c].data_ptr(), scs[c].data_ptr(), qzs[c].data_ptr()) for c in range(copies)]
    out_ptrs = [outs[i].data_ptr() for i in range(n_out_bufs)]
    gemv = lib.xbit_gemv_f16
    stream = torch.cuda.current_stream(dev)

    def kernel_step(i: int, stream_ptr: int) -> None:
        qw, sc, qz = ptrs[i % copies]
        if args.op == "dequant":
            st = lib.xbit_dequant_f16(out_ptrs[i % n_out_bufs], qw, sc, qz, bits, group, k_in, n_local, 1, stream_ptr)
        elif gemv_opts is None:
            st = gemv(out_ptrs[i % n_out_bufs], a_ptr, qw, sc, qz, bits, group, batch, k_in, n_local, 1, stream_ptr)
        else:
            st = lib.xbit_gemv_f16_ex(out_ptrs[i % n_out_bufs], a_ptr, qw, sc, qz, bits, group, batch, k_in, n_local, 1, stream_ptr, gemv_opts)
        if st != 0:
            _lib.check(st, "xbit_gemv_f16")

    def run_steps(first: int, count: int) -> None:
        """The timed path: GEMV, then (sharded layer) the one all-gather of this step's output shards,
        both ordered on the compute stream."""
        sp = torch.cuda.current_stream(dev).cuda_stream
        for i in range(first, first + count):
            kernel_step(i, sp)
            if collective:
                sharded.all_gather_output(gathered[i % n_out_bufs], outs[i % n_out_bufs])

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- untimed: touch every weight copy once (a first touch pays GPU page-table walks: with W = 5 the
    # timed steps would otherwise be the first to read most copies, +1.2 us per step).  The W warm-up steps
    # themselves are queued directly AHEAD of the timed steps (below), not here: a warm-up that ends in a host
    # wait leaves the chip idle again, and the first timed steps then pay the memory system's wake-up
    # (20 timed steps: 7.03-7.23 us per step against 7.00-7.08 with 5 warm-up steps in the chain and 6.84-6.90 with
    # 100; profiles/r03/bench_k20_probe.txt).
    sp0 = torch.cuda.current_stream(dev).cuda_stream
    for c in range(copies):
        kernel_step(c, sp0)
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, HIP events on the compute stream, host clock beside them.
    # One GPU: the K launches are queued behind a blocker kernel long enough to cover the host's
    # enqueue time, so the event-to-event span holds device time only: K x (kernel + launch boundary).
    # (A hipGraph replay measures the same per step — tools/bench_timing_probe.py — once its first,
    # slower replay is out of the way; --graph times that instead.)
    graph, warm_graph, graph_note = None, None, ""
    if (args.graph and world == 1 and not collective) or (collective and backend == "nccl" and not args.no_graph):
        # One GPU: on request.  Sharded layer over RCCL: by default - K x (GEMV + all-gather) recorded ONCE into a hipGraph
        # and replayed, so that the timed region holds device time (kernel + collective) and not the host's ~5 us of
        # enqueue per GEMV plus torch's per-collective launch path.  If the collective cannot be captured on this stack
        # the region falls back to eager launches and says so.
        # Capture first, replay later: a capture that fails on ONE rank must not leave the others inside K captured all-gathers
        # while that rank waits in the all_reduce below (mismatched collectives hang).  So: both graphs are recorded without
        # being replayed, the ranks agree, and only then does every rank run the untimed first replay - or none does.
        g = gw = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run_steps(W, K)
            if W > 0:  # the W warm-up steps as a graph of their own, replayed directly ahead of the timed one
                gw = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gw):
                    run_steps(0, W)
        except Exception as exc:  # noqa: BLE001 - any capture failure means "time it eagerly"
            g = gw = None
            graph_note = "graph capture failed (%s: %s); eager launches" % (type(exc).__name__, str(exc).splitlines()[0][:160] if str(exc) else "")
            torch.cuda.synchronize()
        ok = torch.tensor([1 if g is not None else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank replays, or none does
        if int(ok.item()) == 1:
            graph, warm_graph = g, gw  # assigned together: the note below always matches what runs
            graph.replay()  # the first replay of a graph pays a one-off ~40 us on the device: untimed
            if warm_graph is not None:
                warm_graph.replay()
            torch.cuda.synchronize()
        elif g is not None:
            graph_note = "graph capture failed on another rank; eager launches"
        del g, gw

    stream = torch.cuda.current_stream(dev)
    blocker_cycles = 0
    if world == 1:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        torch.cuda._sleep(1000000)
        c1.record(stream)
        torch.cuda.synchronize()
        cycles_per_us = 1000000 / (c0.elapsed_time(c1) * 1e3)
        blocker_cycles = int((100 + (0 if graph is not None else 12 * (K + W))) * cycles_per_us)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if blocker_cycles:
        torch.cuda._sleep(blocker_cycles)
    # the W untimed warm-up steps, back to back with the timed ones
    if graph is None:
        run_steps(0, W)
    elif warm_graph is not None:
        warm_graph.replay()
    ev0.record(stream)
    if graph is not None:
        graph.replay()
    else:
        run_steps(W, K)
    ev1.record(stream)
    t_queued = time.perf_counter() - t0
    torch.cuda.synchronize()
    barrier()
    dt_host = time.perf_counter() - t0
    dt = ev0.elapsed_time(ev1) * 1e-3
    last_step = W + K - 1  # its output is checked against the oracle below (before anything overwrites the buffer)
    last_out_host = outs[last_step % n_out_bufs].cpu().numpy() if (world == 1 and not args.no_cpu) else None
    if world > 1:
        t = torch.tensor([dt, dt_host], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_host = float(t[0].item()), float(t[1].item())

    # ---- kernel-only durations: HIP events bracketing each launch, >= 200 cold launches whatever K is.
    # The launches are queued back to back behind a blocker (xbit_profile_series_*: one event pair per launch, no host
    # wait in between), i.e. the kernel is timed as it runs inside the timed region above - in a dependent chain.  A
    # launch that starts on an IDLE chip reads 0.3-1 us longer: its XCDs wake up to a microsecond apart
    # (profiles/r03/trace_detail_v63.txt); that figure is kept beside it as `isolated_launch_avg_kernel_us`.
    def profiled_chain(n: int, rotate: bool):
        sp = torch.cuda.current_stream(dev).cuda_stream
        if blocker_cycles:
            torch.cuda._sleep(int((100 + 30 * n) * cycles_per_us))
        _lib.check(lib.xbit_profile_series_begin(n), "xbit_profile_series_begin")
        for i in range(n):
            kernel_step(W + K + i if rotate else 0, sp)
        buf, cnt = (ctypes.c_float * n)(), ctypes.c_int()
        _lib.check(lib.xbit_profile_series_end(buf, n, ctypes.byref(cnt)), "xbit_profile_series_end")
        return [buf[i] for i in range(cnt.value)]

    def profiled_isolated(n: int):
        lib.xbit_profile_enable(1)
        ms = ctypes.c_float()
        vals = []
        sp = torch.cuda.current_stream(dev).cuda_stream
        for i in range(n):
            kernel_step(W + K + i, sp)
            _lib.check(lib.xbit_profile_last_ms(ctypes.byref(ms)), "xbit_profile_last_ms")
            vals.append(ms.value)
        lib.xbit_profile_enable(0)
        return vals

    # the collective alone (K all-gathers of one step's shards, captured like the timed region when that was): what the
    # xGMI hop costs per step, reported beside the kernel and the end-to-end figure
    collective_us = None
    if collective:
        def gathers(count):
            for i in range(count):
                sharded.all_gather_output(gathered[i % n_out_bufs], outs[i % n_out_bufs])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cg = None
        if graph is not None:
            try:
                cg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cg):
                    gathers(K)
                cg.replay()
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                cg = None
                torch.cuda.synchronize()
        gathers(8)
        barrier()
        torch.cuda.synchronize()
        e0.record(stream)
        if cg is not None:
            cg.replay()
        else:
            gathers(K)
        e1.record(stream)
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / K], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        collective_us = float(t.item())

    # ---- pure-read floors, same run, same buffers, same clock (one GPU): the period of a dependent chain of launches
    def chain_period_us(fn, n=200, reps=3):
        out = []
        for _r in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            if blocker_cycles:
                torch.cuda._sleep(int((100 + 12 * n) * cycles_per_us))
            e0.record(stream)
            for i in range(n):
                fn(W + K + i)
            e1.record(stream)
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / n)
        return sorted(out)[len(out) // 2]

    read_floor = None
    if world == 1 and args.op == "gemv":
        try:
            tun = _lib.load_tuning()
            tun.xbit_tuning_read_floor.restype = ctypes.c_int
            tun.xbit_tuning_read_floor.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
        except (_lib.XbitLibraryError, AttributeError) as exc:
            tun, read_floor = None, {"note": "measurement build unavailable (%s)" % str(exc).splitlines()[0][:120]}
        if tun is not None:
            kw_rows = qws[0].shape[0]
            pattern, tcols = floor_tile_shape(lib if gemv_opts is None else _lib.load(), bits, group, batch, k_in, n_local), 0
            sp = stream.cuda_stream

            def floor_fn(pat, tc):
                def fn(i):
                    st = tun.xbit_tuning_read_floor(ptrs[i % copies][0], kw_rows, n_local, pat, tc, sp)
                    if st != 0:
                        raise RuntimeError("xbit_tuning_read_floor -> %d" % st)
                return fn

            floor_fn(0, 0)(0)
            if pattern:
                floor_fn(pattern, tcols)(0)
            torch.cuda.synchronize()
            read_floor = {
                "flat": round(chain_period_us(floor_fn(0, 0)), 3),
                "tiles": round(chain_period_us(floor_fn(pattern, tcols)), 3) if pattern else None,  # None: the launch is not one grid of whole-K 32..128-column tiles (split-K, 16-column tiles)
                "kernel_chain_us": round(chain_period_us(lambda i: kernel_step(i, sp)), 3),
                "tile_lane_map_columns": pattern, "tile_columns": pattern,
                "bytes_read": int(qws[0].numel() * 4),
                "what": "period of 200 back-to-back launches behind a blocker (median of 3), cold weight copies: pure read of the packed rows - flat 16-byte stream / the GEMV's "
                        "column tiles, all loads up front, no arithmetic - and the GEMV itself by the same clock; scales, zero-points, activations and outputs are not read by the floors",
            }

    profiled_chain(8, True)
    cold = profiled_chain(max(MIN_ROOFLINE_LAUNCHES, args.roofline_launches), True)
    warm = profiled_chain(200, False) if args.warm else None
    isolated = profiled_isolated(50)
    torch.cuda.synchronize()
    cold_avg = sum(cold) / len(cold)
    cold_sorted = sorted(cold)
    if world > 1:
        t = torch.tensor([cold_avg], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest GPU's kernel
        cold_avg = float(t.item())

    if rank == 0:
        achieved = nbytes / (cold_avg * 1e-3) / 1e9
        value = world * nbytes * K / dt / 1e9
        sha = lib_sha256(_lib.TUNING_LIB_PATH if gemv_opts is not None else _lib.LIB_PATH)
        src_sha = source_sha256()
        off = offline_profile(workload if world == 1 else "", src_sha) if cfg_name != "shape" else None
        traffic = off["traffic_bytes_per_launch"] if off and off.get("matches_this_build") and off.get("lib_sha256") == sha else None
        rec = {
            "metric": ("dequant W%d GB/s at %dx%d g%d (bytes read + written / time; cold HBM)" % (bits, k_in, n_full, group)) if args.op == "dequant" else
                      "A16W%d GEMV GB/s at %dx%d g%d batch=%d (algorithmic bytes / time; cold HBM)" % (bits, k_in, n_full, group, batch),
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(1e3 * dt / K, 6),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": workload, "name": cfg_name,
                "bits": bits, "groupsize": group, "batch": batch, "in_features": k_in, "out_features_per_gpu": n_local,
                "algorithmic_bytes_per_step": nbytes, "weight_copies_rotated": copies,
                "timed_region": ("hipGraph replay of K launches" if graph is not None else "K eager launches") + (
                    " + K all-gathers" if collective else " queued behind a blocker kernel") + ", HIP events on the compute stream" + (
                    "; " + graph_note if graph_note else ""),
                "host_enqueue_us_per_step": round(1e6 * t_queued / K, 3),
                "parallelism": ("column shards x%d (sharded.shard_columns) + one %s all_gather_into_tensor per step on the compute stream" % (
                    world, "RCCL" if backend == "nccl" else backend)) if collective else "single GPU",
            },
            "host_clock_ms_per_step": round(1e3 * dt_host / K, 6),
            "end_to_end_us_per_step": round(1e6 * dt / K, 3),
            "boundary_us_per_step": round(1e6 * dt / K - cold_avg * 1e3, 3),  # dependent-launch boundary (+ the all-gather on N > 1)
            "collective_us_per_step": None if collective_us is None else round(collective_us, 3),  # the all-gather alone, same launch mode
            "kernel_gbps_per_gpu": round(achieved, 2),
            "aggregate_kernel_gbps": round(world * achieved, 2),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "algorithmic_bytes": nbytes,
                "kernel": "xbit_gemv_f16 dominant kernel (see profiles/ for its name)", "avg_kernel_us": round(cold_avg * 1e3, 3),
                "median_kernel_us": round(cold_sorted[len(cold_sorted) // 2] * 1e3, 3), "min_kernel_us": round(cold_sorted[0] * 1e3, 3),
                "launches": len(cold), "timer": "hipExtLaunchKernel start/stop events per launch, launches queued back to back (a dependent chain, as in the timed region), cold weights" + (
                    ", max over GPUs" if world > 1 else ""),
                "isolated_launch_avg_kernel_us": round(1e3 * sum(isolated) / len(isolated), 3),
                "frac_isolated": round(nbytes / (sum(isolated) / len(isolated) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),  # rounds 1-2's timer: a host wait after every launch
                "read_floor_us": read_floor,
            },
            "offline_profile": off,
            "tok_per_s_proxy": round(achieved * 1e9 / 3.37e9, 1) if cfg_name == "cfg2" else None,
            "tok_per_s_formula": "roofline.achieved GB/s / 3.37 GB of w4g128 linear weights per token of a 7B model",
            "lib_sha256": sha,
            "src_sha256": src_sha,
        }
        if warm is not None:
            warm_avg = sum(warm) / len(warm)
            rec["roofline"]["warm_cache_avg_kernel_us"] = round(warm_avg * 1e3, 3)
            rec["roofline"]["warm_cache_gbps"] = round(nbytes / (warm_avg * 1e-3) / 1e9, 2)
        if world == 1 and not args.no_cpu:
            port, prob, y_port = cpu_baseline_port(args.cpu_seconds, k_in, n_local, bits, group, batch, nbytes)
            port["torch_cpu"] = cpu_baseline_torch(prob, y_port, bits, group, k_in, nbytes, args.cpu_seconds)
            rec["cpu_baseline"] = port
            rec["output_check"] = check_last_output(last_out_host, last_step % copies, qw_h, sc_h, qz_h, act_h, args.op, bits, group, k_in)
        print(json.dumps(rec), flush=True)
    if world > 1 or force_pg:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
