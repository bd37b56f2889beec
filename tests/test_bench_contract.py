"""bench.py's output contract (one JSON line with the driver's keys plus `roofline` and `cpu_baseline`),
checked on the committed run of the final kernel (CPU) and live on the GPU."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def _check(rec, need_cpu_baseline, scaling="weak"):
    for key in REQUIRED:
        assert key in rec, key
    assert rec["unit"] == "GB/s" and rec["higher_is_better"] is True and rec["scaling"] == scaling and rec["vs_baseline"] is None
    assert rec["dtype"] == "f16" and rec["data"] == "synthetic" and "workload" in rec["config"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for key in ROOFLINE:
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert rec["value"] > 0 and abs(rec["ms_per_step"] * 1e-3 * rec["value"] * 1e9 - rec["n_gpus"] * rec["config"]["algorithmic_bytes_per_step"]) < 1e-2 * rec["config"]["algorithmic_bytes_per_step"]
    if need_cpu_baseline:
        cpu = rec["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in cpu, key
        assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1


def _check_round2(rec):
    """What round 2 added: event-timed region, >= 200 roofline launches whatever
    --steps is, offline counters only when they belong to this build, both torch-CPU legs."""
    roof = rec["roofline"]
    assert roof["launches"] >= 200
    assert "host_clock_ms_per_step" in rec and "boundary_us_per_step" in rec
    assert "offline_profile" in rec and "lib_sha256" in rec and "src_sha256" in rec
    if roof["traffic"] is not None:
        assert rec["offline_profile"]["matches_this_build"] is True


def test_committed_bench_line_follows_the_contract():
    """The bench lines committed with the round's profiles (the metric config and the cfg4 mode)."""
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench", "cfg2_w4_4096x11008_m1", "bench.json.log")))  # the latest round's own run (other boxes sit beside it)
    assert logs, "no committed bench line"
    line = [ln for ln in open(logs[-1]).read().splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    _check(rec, need_cpu_baseline=False)  # the profiled run uses --no-cpu
    _check_round2(rec)
    assert rec["config"]["algorithmic_bytes_per_step"] == 23455232 and rec["value"] >= 0.85 * rec["roofline"]["achieved"]
    cfg4 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench", "cfg4_w4_8192x28672*", "bench.json.log")))
    rec4 = json.loads([ln for ln in open(cfg4[-1]).read().splitlines() if ln.startswith("{")][-1])
    _check(rec4, need_cpu_baseline=False, scaling="strong")
    assert rec4["config"]["algorithmic_bytes_per_step"] == 122101760
    # round 4: the record carries the pure-read floors measured in the same run and the same fraction by
    # rounds 1-2's timer; the cfg4 line agrees with the rocprofv3 summary committed beside it (rocprof reads the same dispatch
    # timestamps 0.3-1 us above the events and never below)
    if "read_floor_us" in rec["roofline"]:
        fl = rec["roofline"]["read_floor_us"]
        assert 0 < fl["flat"] <= fl["tiles"] * 1.02 and fl["tiles"] < fl["kernel_chain_us"] and "frac_isolated" in rec["roofline"]
        for r_, d_ in ((rec, os.path.dirname(logs[-1])), (rec4, os.path.dirname(cfg4[-1]))):
            head = [ln for ln in open(os.path.join(d_, "bench_kernel_trace.txt")).read().splitlines() if ln.startswith("all launches")][0]
            rocprof_avg = float(head.split(" avg ")[1].split(" us")[0])
            assert r_["roofline"]["avg_kernel_us"] * 0.97 <= rocprof_avg <= r_["roofline"]["avg_kernel_us"] * 1.25, (d_, rocprof_avg, r_["roofline"]["avg_kernel_us"])
    # every profiled config carries its PMC traffic, taken from the library that produced the bench line
    for d in glob.glob(os.path.join(ROOT, "profiles", "r02", "bench", "*")):
        t = json.load(open(os.path.join(d, "pmc_traffic.json")))
        b = json.loads([ln for ln in open(os.path.join(d, "bench.json.log")).read().splitlines() if ln.startswith("{")][-1])
        assert t["lib_sha256"] == b["lib_sha256"] and 0.98 <= t["traffic_over_algorithmic"] <= 1.10, d


@pytest.mark.gpu
def test_bench_runs_and_follows_the_contract(cuda_device):
    """The driver's command line at a SMALL step count: the roofline sample must not shrink with it, and
    the event-timed `value` must sit near the kernel rate (the graph's launch boundaries are all that
    separates them), not at the host-clock rate the round-1 line reported."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    _check(rec, need_cpu_baseline=True)
    _check_round2(rec)
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 5
    # a step = the kernel (timed inside the chain by its own events) + the dependent-launch boundary (0.4-0.7 us of a ~6.7 us step)
    assert 0.85 * rec["roofline"]["achieved"] <= rec["value"] <= 1.02 * rec["roofline"]["achieved"], (rec["value"], rec["roofline"]["achieved"])
    tc = rec["cpu_baseline"]["torch_cpu"]
    assert tc["threads"] >= 1 and tc["unpack_plus_matmul_ms"] > tc["matmul_only_ms"] > 0
    # round 4: the floors of this very run, and the last timed step's output checked against the oracle
    fl = rec["roofline"]["read_floor_us"]
    assert 0 < fl["flat"] < fl["kernel_chain_us"] and 0 < fl["tiles"] < fl["kernel_chain_us"] and fl["tile_lane_map_columns"] == 48
    assert rec["output_check"]["ok"] is True and rec["output_check"]["outputs"] == 11008
    assert 0 < rec["roofline"]["frac_isolated"] <= rec["roofline"]["frac"] * 1.05


@pytest.mark.gpu
def test_bench_cfg4_single_gpu_and_two_rank_rehearsal(cuda_device):
    """--config cfg4: the whole 8192x28672 layer on one GPU, and the N = 2 control flow (shard_columns +
    one all-gather per step on the compute stream) rehearsed with two ranks on this one GPU over gloo
    (RCCL needs one GPU per rank; the driver's multi-GPU runs use it)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg4", "--steps", "20", "--warmup", "5", "--no-cpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    _check(rec, need_cpu_baseline=False, scaling="strong")
    _check_round2(rec)
    assert rec["config"]["out_features_per_gpu"] == 28672 and rec["config"]["algorithmic_bytes_per_step"] == 122101760
    env = dict(os.environ, XBIT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg4", "--steps", "10", "--warmup", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    _check(rec, need_cpu_baseline=False, scaling="strong")
    assert rec["n_gpus"] == 2 and rec["config"]["out_features_per_gpu"] == 14336 and rec["config"]["algorithmic_bytes_per_step"] == 61059072
    assert "all_gather_into_tensor per step" in rec["config"]["parallelism"]
