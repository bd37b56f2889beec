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
    flip = ref.astype(np.float32).copy()
    flip[0, 0] = -0.5   # opposite sign: both ratios negative; the reference takes abs() of them (gemv_w4a16.cu:406)
    flip[0, 3] = 3.0    # exactly -ref: |ratio| == 1, which the reference's rule does NOT flag (faithfully reproduced)
    assert harness.gemv_errors(flip, ref) == 1
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


def test_harness_library_exports_qbitgemv_and_reference_main_links():
    """libxbitops_harness.so exports the reference harness's entry symbol (gemv.cuh:22), and where the
    reference is mounted its own main.cc (which calls QbitGemv at line 102) compiles and links against
    it UNCHANGED - the binary goes to a temporary directory, nothing of the reference enters the repo."""
    import os
    import subprocess
    import tempfile

    from xbitops_amd import build

    lib = build.build_harness()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    assert [line.split()[-1] for line in out.splitlines() if " T " in line] == ["QbitGemv"]
    assert os.path.exists(build.HARNESS_MAIN_BIN)
    r = subprocess.run([build.HARNESS_MAIN_BIN], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    ref_main = "/root/reference/src/main.cc"
    if os.path.exists(ref_main):
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "ref_main")
            r = subprocess.run(["g++", "-O1", "-w", "-I", "/root/reference/src", ref_main, "-o", exe, "-L", build.LIB_DIR, "-lxbitops_harness", "-lxbitops_hip",
                                "-Wl,-rpath," + build.LIB_DIR], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            syms = subprocess.run(["nm", "-D", exe], capture_output=True, text=True).stdout
            assert " U QbitGemv" in syms and " MATRIX_K" in syms  # imports the entry point, exports the shape globals to the library


@pytest.mark.gpu
def test_qbitgemv_through_a_main_cc_shaped_host(tmp_path, cuda_device):
    import subprocess

    from xbitops_amd import build

    build.build_harness()
    s = _sample(k=2048, n=256, bits=5, g=128, bias=0)
    harness.save_sample(str(tmp_path), s)
    r = subprocess.run([build.HARNESS_MAIN_BIN, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "M = 1, N = 256, K = 2048" in r.stdout and "Results verified" in r.stdout and "gemv took" in r.stdout
    s.out = (s.out.astype(np.float32) * -0.5).astype(np.float16)  # wrong sign and size: both |ratios| leave 1 +- 1e-3
    harness.save_sample(str(tmp_path), s)
    r = subprocess.run([build.HARNESS_MAIN_BIN, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "does not agree" in r.stdout
    # the library also works without the MATRIX_* globals (shape from the byte sizes): load it from Python
    import ctypes

    from xbitops_amd import _lib

    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    hl = ctypes.CDLL(build.HARNESS_LIB)

    class Sizes(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int) for n in ("input_size", "qweight_size", "scale_size", "qzero_size", "f16weight_size", "out_size")]

    class SampleData(ctypes.Structure):
        _fields_ = [("input", ctypes.c_void_p), ("qweight", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("qzero", ctypes.c_void_p),
                    ("f16weight", ctypes.c_void_p), ("out", ctypes.c_void_p), ("bits", ctypes.c_int32), ("group_size", ctypes.c_int32), ("size", Sizes)]

    s = _sample(k=1024, n=128, bits=4, g=128, bias=0)
    arrs = [np.ascontiguousarray(a) for a in (s.input, s.qweight, s.scales, s.qzeros, s.weight, s.out)]
    sd = SampleData(*[a.ctypes.data for a in arrs], 4, 128, Sizes(*[a.nbytes for a in arrs]))
    hl.QbitGemv.restype = ctypes.c_int
    assert hl.QbitGemv(ctypes.byref(sd)) == 0
