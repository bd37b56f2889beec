"""The reference harness's seven-file data set (main.cc:14-80): write, read back, and (GPU) replay
with the reference's own acceptance tests."""
import numpy as np
import pytest

import oracle
from xbitops_amd import harness, synth


def _sample(k=7167 // 7 * 8, n=96, bits=5, g=128, bias=0):
    p = synth.make_problem(k, n, bits, g, m=1, seed=77)
    w = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, bias)
    y = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, bias)
    return harness.Sample(k, n, bits, g, p.act, p.qweight, p.scales, p.qzeros, w, y)


def test_round_trip(tmp_path):
    s = _sample()
    harness.save_sample(str(tmp_path), s)
    assert sorted(f.name for f in tmp_path.iterdir()) == sorted(harness.FILES)
    assert (tmp_path / "shape.bin").read_text() == "%d,%d,5,128" % (s.k, s.n)
    t = harness.load_sample(str(tmp_path))
    assert (t.k, t.n, t.bits, t.groupsize) == (s.k, s.n, 5, 128)
    for name in ("input", "qweight", "scales", "qzeros", "weight", "out"):
        assert np.array_equal(getattr(t, name).view(np.uint8), np.ascontiguousarray(getattr(s, name)).view(np.uint8)), name


def test_reference_acceptance_rules():
    ref = np.array([[1.0, 2.0, 1e-4, -3.0]], np.float16)
    assert harness.gemv_errors(ref, ref) == 0
    off = ref.astype(np.float32).copy()
    off[0, 1] *= 1.01  # rel > 1e-3 and abs > 4e-4 -> error
    off[0, 2] *= 1.5  # rel > 1e-3 but abs < 4e-4 -> tolerated
    assert harness.gemv_errors(off, ref) == 1
    w = np.zeros((4, 4), np.float16)
    w2 = w.copy()
    w2[1, 1] = 2e-3
    assert harness.dequant_errors(w2, w) == 1


@pytest.mark.gpu
def test_replay_on_gpu(tmp_path, cuda_device):
    s = _sample(k=1024, n=128, bits=4, g=128, bias=1)
    harness.save_sample(str(tmp_path), s)
    r = harness.replay(str(tmp_path), add_zero_bias=1, iters=3)
    assert r["dequant_errors"] == 0 and r["gemv_errors"] == 0 and r["gemv_ms"] > 0


def test_cpp_replay_program_builds_without_torch():
    """examples/replay_dataset.cc: a C++ host over the C ABI alone (the reference's harness executable,
    main.cc).  Builds with hipcc, links libxbitops_hip and nothing of torch, and rejects bad usage."""
    import subprocess

    from xbitops_amd import build

    exe = build.build_example()
    deps = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    names = [line.split()[0] for line in deps.splitlines() if line.strip()]  # library names only (addresses are random hex)
    assert any(n.startswith("libxbitops_hip.so") for n in names)
    assert not any("torch" in n or "c10" in n for n in names), names
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_cpp_replay_program_on_gpu(tmp_path, cuda_device):
    import subprocess

    from xbitops_amd import build

    exe = build.build_example()
    s = _sample(k=2048, n=256, bits=3, g=128, bias=1)
    harness.save_sample(str(tmp_path), s)
    r = subprocess.run([exe, str(tmp_path), "1", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dequant: 0 of" in r.stdout and "gemv:    0 of" in r.stdout
    # a wrong golden output must be reported (exit status 1)
    s.out = (s.out.astype(np.float32) + 0.5).astype(np.float16)
    harness.save_sample(str(tmp_path), s)
    r = subprocess.run([exe, str(tmp_path), "1", "1"], capture_output=True, text=True)
    assert r.returncode == 1, r.stdout + r.stderr
