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


def _check(rec, need_cpu_baseline):
    for key in REQUIRED:
        assert key in rec, key
    assert rec["unit"] == "GB/s" and rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None
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


def test_committed_bench_line_follows_the_contract():
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench.json.log")))
    assert logs, "no committed bench line"
    line = [ln for ln in open(logs[-1]).read().splitlines() if ln.startswith("{")][-1]
    _check(json.loads(line), need_cpu_baseline=False)  # the profiled run uses --no-cpu


@pytest.mark.gpu
def test_bench_runs_and_follows_the_contract(cuda_device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "64", "--warmup", "8", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    _check(rec, need_cpu_baseline=True)
    assert rec["n_gpus"] == 1 and rec["steps"] == 64 and rec["warmup"] == 8
