#!/usr/bin/env python3
"""bench.py — A16W4 GEMV throughput on MI355X, the metric BASELINE.json names.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one call of the hot path (xbit_gemv_f16 through the C ABI) on the Llama-7B FFN
shape of BASELINE.json config 2: in=4096, out=11008, 4-bit, group 128, batch 1, synthetic data
(xbitops_amd.synth).  The step rotates over enough distinct weight copies (>= 512 MB in total,
twice the 256 MB Infinity Cache) that every launch streams its weights from HBM ("cold");
a cache-resident ("warm") figure is reported beside it, never as the headline.

Printed JSON (one line, rank 0):
  value / ms_per_step  whole-job GB/s of algorithmic bytes and wall time per step over exactly K
                       steps between barrier+synchronize pairs.  The K launches are replayed from a
                       hipGraph so the host is not the bottleneck; kernel-to-kernel boundaries ARE
                       included.
  roofline             algorithmic bytes per launch / average KERNEL duration, measured live with
                       HIP events that bracket each launch on its own stream
                       (hipExtLaunchKernel start/stop events via xbit_profile_enable), cold weights.
                       peak = 8000 GB/s (HBM3E spec, MI355X_MICROARCH.md).  `traffic` is filled from
                       rocprofv3 PMC runs offline (profiles/), null here.
  cpu_baseline         the oracle's C restatement (fused dequant + GEMV, OpenMP) timed on this
                       box's host cores on repeated full-size GEMVs for ~10 s.  Reported, not a target.
N > 1 (launched by torch.distributed.run): weak scaling — every rank owns one 4096x11008 column
shard of a 4096 x (11008*N) layer, no data-path collective inside the GEMV; the fp16 output
shards are all-gathered over RCCL on a side stream, bucketed, overlapping the following GEMVs.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
K_IN, N_OUT, BITS, GROUP, BATCH = 4096, 11008, 4, 128, 1
COLD_BYTES = 512 << 20


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--no-graph", action="store_true", help="eager launches in the timed region")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--gather-bucket", type=int, default=32, help="N>1: steps per all-gather bucket")
    ap.add_argument("--variant", type=int, default=-1, help="tuning: GEMV launch-shape variant")
    return ap.parse_args()


def cpu_baseline(seconds: float, nbytes: int):
    """Oracle C restatement (kind 'port'), fused dequant+GEMV, OpenMP over output columns."""
    import numpy as np

    import oracle
    from xbitops_amd import synth

    oracle.set_num_threads(min(16, os.cpu_count() or 1))  # the GPU box grants a 16-CPU share per GPU
    p = synth.make_problem(K_IN, N_OUT, BITS, GROUP, m=BATCH, seed=7)
    oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, GROUP, BITS, K_IN, 1)  # warm (page in, spin up threads)
    reps, t0 = 0, time.perf_counter()
    while True:
        y = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, GROUP, BITS, K_IN, 1)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and reps >= 3:
            break
    assert np.isfinite(y.astype(np.float32)).all()
    return {
        "value": round(nbytes * reps / dt / 1e9, 4), "unit": "GB/s", "cores": oracle.num_threads(), "kind": "port",
        "sample": "%d full-size %dx%d w%d g%d batch-%d GEMVs in %.1f s (oracle/xbit_oracle.c xo_gemv, OpenMP; host has %d logical CPUs)"
                  % (reps, K_IN, N_OUT, BITS, GROUP, BATCH, dt, os.cpu_count()),
        "ms_per_gemv": round(1e3 * dt / reps, 3),
    }


def pmc_traffic(workload_key: str):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE with
    the gfx950 x2 correction + WRITE_SIZE; collected offline — counters cannot be read in-process).
    Returns None when no record matches this workload."""
    best = None
    for root, _dirs, files in os.walk(os.path.join(ROOT, "profiles")):
        if "pmc_traffic.json" in files:
            try:
                with open(os.path.join(root, "pmc_traffic.json")) as f:
                    rec = json.load(f)
                if rec.get("workload") == workload_key:
                    best = (os.path.relpath(os.path.join(root, "pmc_traffic.json"), ROOT), rec)
            except (OSError, ValueError):
                pass
    return best


def rocprof_kernel_avg_us():
    """Average duration of the GEMV kernel in the committed rocprofv3 --kernel-trace --stats summary
    of this same command (profiles/rNN/bench_kernel_stats.csv, newest round), for comparison with
    the live HIP-event figure.  None when there is no summary."""
    import csv
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_kernel_stats.csv")), reverse=True):
        try:
            with open(path) as f:
                for row in csv.DictReader(f):
                    if "gemv_mfma_kernel" in row["Name"]:
                        return os.path.relpath(path, ROOT), round(float(row["AverageNs"]) / 1e3, 3)
        except (OSError, ValueError, KeyError):
            pass
    return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from xbitops_amd import _lib, synth

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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # nccl == RCCL on ROCm.  XBIT_BENCH_BACKEND=gloo rehearses the multi-rank control flow with
        # several ranks on ONE GPU (RCCL refuses two ranks per device); never used for reported numbers.
        backend = os.environ.get("XBIT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    lib = _lib.load()  # raises if the HIP library is missing
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.xbit_debug_set_gemv_variant.argtypes = [ctypes.c_int]
    raw.xbit_debug_set_gemv_variant(args.variant)

    nbytes = synth.algorithmic_bytes(K_IN, N_OUT, BITS, GROUP, BATCH)
    copies = (COLD_BYTES + nbytes - 1) // nbytes + 1  # >= 512 MB of distinct weights
    K, W = args.steps, args.warmup

    # ---- synthetic, seeded inputs resident in HBM before any timing
    base = synth.make_problem(K_IN, N_OUT, BITS, GROUP, m=BATCH, seed=1000 + rank)
    act = torch.from_numpy(base.act).to(dev)
    qws, scs, qzs = [], [], []
    for c in range(copies):
        # distinct contents per copy (xor with a per-copy word) so no two copies can be deduplicated
        qw = torch.from_numpy(base.qweight).to(dev) ^ (0x01010101 * (c % 127))
        qws.append(qw.contiguous())
        scs.append(torch.from_numpy(base.scales).to(dev).clone())
        qzs.append((torch.from_numpy(base.qzeros).to(dev) ^ (0x11111111 * (c % 13))).contiguous())
    n_out_bufs = 2 * max(1, args.gather_bucket)
    outs = torch.empty((n_out_bufs, BATCH, N_OUT), dtype=torch.float16, device=dev)
    gathered = torch.empty((2, world, max(1, args.gather_bucket), BATCH, N_OUT), dtype=torch.float16, device=dev) if world > 1 else None
    torch.cuda.synchronize()

    a_ptr = act.data_ptr()
    ptrs = [(qws[c].data_ptr(), scs[c].data_ptr(), qzs[c].data_ptr()) for c in range(copies)]
    out_ptrs = [outs[i].data_ptr() for i in range(n_out_bufs)]
    gemv = lib.xbit_gemv_f16

    def step(i: int, stream_ptr: int) -> None:
        qw, sc, qz = ptrs[i % copies]
        st = gemv(out_ptrs[i % n_out_bufs], a_ptr, qw, sc, qz, BITS, GROUP, BATCH, K_IN, N_OUT, 1, stream_ptr)
        if st != 0:
            _lib.check(st, "xbit_gemv_f16")

    compute = torch.cuda.current_stream(dev)
    comm = torch.cuda.Stream(dev) if world > 1 else None
    bucket = max(1, args.gather_bucket)

    gathered_ev = [None, None]  # per output half: the all-gather that last read it

    def gather_bucket(b: int) -> None:
        """all-gather the `bucket` outputs of bucket b on the side stream (after they are computed)."""
        half = b % 2
        src = outs[half * bucket:(half + 1) * bucket]
        ev = torch.cuda.Event()
        ev.record(compute)
        with torch.cuda.stream(comm):
            comm.wait_event(ev)
            dist.all_gather_into_tensor(gathered[half].view(-1), src.reshape(-1))
            gathered_ev[half] = torch.cuda.Event()
            gathered_ev[half].record(comm)

    def run_steps(first: int, count: int) -> None:
        sp = torch.cuda.current_stream(dev).cuda_stream
        for i in range(first, first + count):
            if world > 1 and i % bucket == 0 and gathered_ev[(i // bucket) % 2] is not None:
                compute.wait_event(gathered_ev[(i // bucket) % 2])  # do not overwrite outputs still being gathered
            step(i, sp)
            if world > 1 and (i + 1) % bucket == 0:
                gather_bucket(i // bucket)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up (untimed)
    run_steps(0, W)
    if comm is not None:
        compute.wait_stream(comm)
    torch.cuda.synchronize()

    use_graph = (not args.no_graph) and world == 1
    graph = None
    if use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run_steps(W, K)
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        graph.replay()
    else:
        run_steps(W, K)
        if comm is not None:
            # leftover partial bucket, then join the side stream
            if (W + K) % bucket and world > 1:
                gather_bucket((W + K) // bucket)
            compute.wait_stream(comm)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- kernel-only durations: HIP events bracketing each launch (cold rotation, then warm)
    def profiled(n: int, rotate: bool):
        lib.xbit_profile_enable(1)
        ms = ctypes.c_float()
        vals = []
        sp = torch.cuda.current_stream(dev).cuda_stream
        for i in range(n):
            step(W + K + i if rotate else 0, sp)
            _lib.check(lib.xbit_profile_last_ms(ctypes.byref(ms)), "xbit_profile_last_ms")
            vals.append(ms.value)
        lib.xbit_profile_enable(0)
        return vals

    profiled(8, True)
    cold = profiled(min(K, 400), True)
    warm = profiled(min(K, 400), False)
    torch.cuda.synchronize()
    cold_avg = sum(cold) / len(cold)
    warm_avg = sum(warm) / len(warm)
    cold_sorted = sorted(cold)

    if rank == 0:
        achieved = nbytes / (cold_avg * 1e-3) / 1e9
        value = world * nbytes * K / dt / 1e9
        traffic = pmc_traffic("A16W4 GEMV in=%d out=%d group=%d batch=%d" % (K_IN, N_OUT, GROUP, BATCH))
        rec = {
            "metric": "A16W4 GEMV GB/s at 4096x11008 g128 batch=1 (algorithmic bytes / time; cold HBM)",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(1e3 * dt / K, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": "A16W4 GEMV in=4096 out=11008 group=128 batch=1 (Llama-7B FFN, BASELINE.json configs[1]) per GPU",
                "bits": BITS, "groupsize": GROUP, "batch": BATCH, "algorithmic_bytes_per_step": nbytes,
                "weight_copies_rotated": copies, "timed_region": "hipGraph replay of K launches" if graph is not None else "eager launches",
                "parallelism": "column shards x%d + bucketed %s all-gather (bucket %d) on a side stream" % (
                    world, "RCCL" if os.environ.get("XBIT_BENCH_BACKEND", "nccl") == "nccl" else os.environ["XBIT_BENCH_BACKEND"], bucket)
                if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic[1]["traffic_bytes_per_launch"] if traffic else None,
                "rocprof_avg_kernel_us": (rocprof_kernel_avg_us() or (None, None))[1],
                "rocprof_source": ((rocprof_kernel_avg_us() or (None, None))[0] or "") + ": rocprofv3 --kernel-trace --stats of this command; averages cold, graph and cache-warm launches and runs ~5-10 % slower under the profiler",
                "traffic_source": (traffic[0] + ": rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, bytes per launch") if traffic else None,
                "algorithmic_bytes": nbytes,
                "kernel": "gemv_mfma_kernel<4,...>", "avg_kernel_us": round(cold_avg * 1e3, 3),
                "median_kernel_us": round(cold_sorted[len(cold_sorted) // 2] * 1e3, 3), "min_kernel_us": round(cold_sorted[0] * 1e3, 3),
                "launches": len(cold), "timer": "hipExtLaunchKernel start/stop events per launch",
                "warm_cache_avg_kernel_us": round(warm_avg * 1e3, 3), "warm_cache_gbps": round(nbytes / (warm_avg * 1e-3) / 1e9, 2),
            },
            "tok_per_s_proxy": round(achieved * 1e9 / 3.37e9, 1),
            "tok_per_s_formula": "roofline.achieved GB/s / 3.37 GB of w4g128 linear weights per token of a 7B model",
        }
        if world == 1 and not args.no_cpu:
            rec["cpu_baseline"] = cpu_baseline(args.cpu_seconds, nbytes)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
