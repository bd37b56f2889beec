"""Parity of the HIP path against the oracle, through the operator surface (XbitOps.dequant /
XbitOps.gemv -> ctypes -> C ABI -> kernels).  GPU only.

Gates:
* dequant: BIT-EXACT against the oracle in RNE mode (integer unpack + IEEE fp16 hmul/hfma), and
  within the reference's own 1e-3 abs tolerance of the reference-simulator fixtures
  (unpack_weight_2_to_7.cu:570) — the simulator rounds ties up, hardware rounds them to even;
* GEMV: fp16 output within 1e-3 relative of  fp16(sum_k a*W_oracle)  (BASELINE.json), checked as
  |y - y_ref| <= 1e-3*|y_ref| + 1 fp16 ulp slack near zero; since W is rebuilt bit-exactly and
  accumulated in fp32, the observed error is a few fp32 ulps plus the final fp16 rounding.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import contract
from xbitops_amd import packing, synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "dq_b*.npz")))
with open(os.path.join(GOLD, "digests.json")) as _f:
    DIGESTS = json.load(_f)


def _bits(x):
    return np.ascontiguousarray(x).view(np.uint16)


def to_dev(p, dev):
    return (torch.from_numpy(p.act).to(dev), torch.from_numpy(p.qweight).to(dev), torch.from_numpy(p.scales).to(dev),
            torch.from_numpy(p.qzeros).to(dev))


def run_dequant(p, bias, dev):
    import XbitOps

    _a, qw, sc, qz = to_dev(p, dev)
    out = XbitOps.dequant(qw, sc, qz, p.groupsize, p.bits, p.k, bias)
    torch.cuda.synchronize()
    assert out.shape == (p.k, p.n) and out.dtype == torch.float16
    return out.cpu().numpy()


def run_gemv(p, bias, dev):
    import XbitOps

    a, qw, sc, qz = to_dev(p, dev)
    out = XbitOps.gemv(a, qw, sc, qz, p.groupsize, p.bits, p.k, bias)
    torch.cuda.synchronize()
    assert out.shape == (p.m, p.n) and out.dtype == torch.float16
    return out.cpu().numpy()


def assert_gemv_close(y, y_ref, what="", k=None):
    """The one GEMV contract (oracle/contract.py): 1e-3 relative, and equal to the oracle up to the last fp16 bit unless the sum
    cancels (slack k * max|a*W| * 2^-24)."""
    assert k is not None
    contract.assert_gemv_close(y, y_ref, k, what)


def test_native_library_is_what_runs(cuda_device):
    """Fail loudly if anything but the in-tree HIP library would serve the ops."""
    from xbitops_amd import _lib

    _lib.load()
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(_lib.LIB_PATH) in maps


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_dequant_vs_reference_fixture(path, cuda_device):
    f = np.load(path)
    k, n, bits, g, seed = (int(v) for v in f["meta"])
    p = synth.make_problem(k, n, bits, g, m=1, seed=seed)
    w = run_dequant(p, 0, cuda_device)
    assert np.abs(w.astype(np.float32) - f["w_ref"].astype(np.float32)).max() <= 1e-3  # reference tolerance
    w_rne = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, 0, oracle.RNE)
    assert np.array_equal(_bits(w), _bits(w_rne))  # bit-exact vs hardware-rounding oracle


def reference_gemv_errors(y, y_ref):
    """The reference harness's own acceptance rule for the fused GEMV (src/cu/gemv_w4a16.cu:406-407): an element is
    an error iff (ref / v1 > 1.001 or v1 / ref > 1.001) and |ref - v1| > 4e-4.  Written so that NaN / Inf on one side
    only counts as an error (the reference's comparison is false for NaN)."""
    v1, ref = y.astype(np.float64).ravel(), y_ref.astype(np.float64).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (np.abs(ref / v1) > 1.001) | (np.abs(v1 / ref) > 1.001)
        far = np.abs(ref - v1) > 4e-4
    broken = ~np.isfinite(v1) | ~np.isfinite(ref)
    return int(((ratio & far) | (broken & (v1 != ref))).sum())


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
@pytest.mark.parametrize("m", [1, 4])
def test_gemv_vs_reference_fixture_weights(path, m, cuda_device):
    """The direct link: HIP GEMV against an fp32 matmul of the weights the REFERENCE SIMULATOR produced (fixture `w_ref`,
    oracle/gen_golden.py), judged by the reference's own rule - no oracle code between the two.
    The simulator rounds fp32 -> fp16 half-up (cpp_simulate.cc:43-59) where hardware (CUDA and this kernel alike) rounds to
    nearest even, so 4-14 % of its W are one ulp high and a K = 200..384 sum over them is biased by up to ~1.2e-3
    (measured on these fixtures with W rebuilt in hardware rounding).  The reference's rule therefore flags a few outputs
    for ANY hardware kernel; what is asserted is that nothing lies outside that rounding band: every flagged output is
    within 2e-3 absolute, at most 8 % of the outputs are flagged (worst fixture: 4.9 %), and the reference's own pass
    criterion (`errors > 100000` fails, gemv_w4a16.cu:413) holds trivially."""
    f = np.load(path)
    k, n, bits, g, seed = (int(v) for v in f["meta"])
    p = synth.make_problem(k, n, bits, g, m=m, seed=seed)
    assert np.array_equal(p.qweight, f["qweight"]) and np.array_equal(_bits(p.scales), _bits(f["scales"]))
    y = run_gemv(p, 0, cuda_device)
    y_ref = p.act.astype(np.float32) @ f["w_ref"].astype(np.float32)  # fp32 accumulate over the simulator's fp16 W
    flagged = reference_gemv_errors(y, y_ref)
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    assert np.isfinite(y.astype(np.float32)).all()
    assert flagged <= 0.08 * y.size, (flagged, y.size)
    assert err.max() <= 2e-3, err.max()
    assert (err <= 1e-3 * np.abs(y_ref) + 2e-3).all()


DQ_CASES = [
    # k, n, g, bias  — aligned, ragged K, group smaller than a k-block, generic-path shapes
    (256, 128, 128, 0), (256, 128, 128, 1), (4096, 256, 128, 1), (200, 64, 32, 0), (7167 // 7, 72, 64, 1),
    (96, 16, 16, 2), (160, 24, 48, 0), (64, 6, 32, 1), (33, 2, 16, 0), (1024, 8, 1024, 255),
]


@pytest.mark.parametrize("bits", range(2, 9))
def test_dequant_bit_exact_all_bits(bits, cuda_device):
    for k, n, g, bias in DQ_CASES:
        p = synth.make_problem(k, n, bits, g, m=1, seed=bits * 131 + k + n)
        w = run_dequant(p, bias, cuda_device)
        w_ref = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
        assert np.array_equal(_bits(w), _bits(w_ref)), (bits, k, n, g, bias)


@pytest.mark.parametrize("bits", range(2, 9))
def test_integer_unpack_bit_exact_via_unit_scales(bits, cuda_device):
    """scale = 1, zero = 0  =>  W == q exactly: pins the bit-pack / index path on its own."""
    rng = np.random.default_rng(bits)
    k, n = 512, 64
    q = rng.integers(0, 1 << bits, size=(k, n), dtype=np.uint8)
    p = synth.Problem(k, n, bits, 128, 1, packing.pack_qweight(q, bits), np.ones((4, n), np.float16),
                      packing.pack_qzeros(np.zeros((4, n), np.uint8), bits), np.zeros((1, k), np.float16))
    w = run_dequant(p, 0, cuda_device)
    assert np.array_equal(w.astype(np.int32), q.astype(np.int32))
    # zero-points: W == q - z with per-(group, column) z
    z = rng.integers(0, 1 << bits, size=(4, n), dtype=np.uint8)
    p.qzeros = packing.pack_qzeros(z, bits)
    w = run_dequant(p, 0, cuda_device)
    assert np.array_equal(w.astype(np.int32), q.astype(np.int32) - z[np.arange(k) // 128].astype(np.int32))


GEMV_CASES = [
    # k, n, g, m, bias
    (4096, 256, 128, 1, 0), (4096, 256, 128, 1, 1), (512, 64, 128, 1, 0), (8192, 128, 128, 1, 1), (11008, 64, 128, 1, 0),
    (4096, 96, 64, 2, 1), (4096, 32, 32, 3, 0), (1024, 40, 128, 4, 1), (4096, 128, 128, 16, 0), (2048, 36, 256, 5, 1),
    (200, 8, 40, 1, 0), (4096, 6, 128, 2, 1), (1000, 64, 128, 1, 0), (4100, 64, 128, 1, 1), (96, 16, 16, 1, 0),
]


@pytest.mark.parametrize("bits", range(2, 9))
def test_gemv_all_bits(bits, cuda_device):
    for k, n, g, m, bias in GEMV_CASES:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 977 + k + 3 * n + m)
        y = run_gemv(p, bias, cuda_device)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
        assert_gemv_close(y, y_ref, "bits=%d k=%d n=%d g=%d m=%d bias=%d" % (bits, k, n, g, m, bias), k=k)


@pytest.mark.parametrize("bits", [3, 4, 8])
def test_gemv_row_counts_and_tile_widths(bits, cuda_device):
    """Every rows-per-pass variant (1, 4, 8, 16, >16) and every tile width the launcher can pick
    (8/12/16-byte loads), including column counts that end inside a tile."""
    for m in (2, 3, 5, 8, 9, 16, 17, 33):
        p = synth.make_problem(1024, 96, bits, 128, m=m, seed=bits * 31 + m)
        y = run_gemv(p, 1, cuda_device)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, bits, 1024, 1, oracle.RNE)
        assert_gemv_close(y, y_ref, "bits=%d m=%d" % (bits, m), k=p.k)
    # ... and the 128-column tiles of very wide layers (n >= 25600), whole and ragged, 1..8 rows, K sweeps
    for k, n, m in ((256, 8192, 1), (512, 11008, 1), (512, 11008, 4), (256, 4096 + 64, 2), (8192, 50, 1), (8192 + 64, 130, 3),
                    (4096, 25600, 1), (4096 + 256, 25600 + 34, 3), (512, 32000, 8), (256, 28672, 2)):
        p = synth.make_problem(k, n, bits, 128, m=m, seed=bits * 37 + n)
        y = run_gemv(p, 0, cuda_device)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, bits, k, 0, oracle.RNE)
        assert_gemv_close(y, y_ref, "bits=%d k=%d n=%d m=%d" % (bits, k, n, m), k=k)


def test_gemv_w4_llama70b_ffn_shard_and_full(cuda_device):
    """BASELINE.json config 4: the 8-GPU column shard (8192x3584) against the oracle, and the full
    8192x28672 layer through a size-independent property: the result restricted to a column shard
    equals the GEMV of that shard alone (columns are independent)."""
    import XbitOps
    from xbitops_amd import sharded

    p = synth.make_problem(8192, 28672, 4, 128, m=1, seed=4)
    a, qw, sc, qz = to_dev(p, cuda_device)
    y_full = XbitOps.gemv(a, qw, sc, qz, 128, 4, 8192, 1)
    for r in (0, 5, 7):
        qw_r, sc_r, qz_r = sharded.shard_columns(qw, sc, qz, 4, 8, r)
        y_r = XbitOps.gemv(a, qw_r, sc_r, qz_r, 128, 4, 8192, 1)
        # the shard may run a different tile shape than the full layer (different fp32 summation
        # order), so the two agree to the last fp16 bit, not bit for bit
        y_f = y_full[:, r * 3584:(r + 1) * 3584].contiguous()
        d = (y_r.view(torch.int16).int() - y_f.view(torch.int16).int()).abs()
        near_zero = (y_r.float() - y_f.float()).abs() <= contract.cancel_slack(8192)  # fp32 summation noise where the sum cancels (oracle/contract.py)
        assert bool(((d <= 1) | near_zero).all()) and float((d == 0).float().mean()) > 0.98
    qw_r, sc_r, qz_r = sharded.shard_columns(p.qweight, p.scales, p.qzeros, 4, 8, 5)
    y_ref = oracle.gemv(p.act, qw_r, sc_r, qz_r, 128, 4, 8192, 1, oracle.RNE)
    assert_gemv_close(y_full[:, 5 * 3584:6 * 3584].cpu().numpy(), y_ref, "cfg4 shard 5", k=p.k)


def test_gemv_w4_llama70b_down_projection_128_rows_full_size(cuda_device):
    """The 70B down-projection (28672x8192, 224 groups: the rows kernel with its tables as a window) at 128 rows and full size: a column shard against the
    oracle, and two size-independent properties - the result restricted to a column shard equals the GEMV of that shard alone, and a row of the batch equals
    the one-row GEMV of that row (other kernels, other summation orders: to the last fp16 bit)."""
    import XbitOps
    from xbitops_amd import sharded

    k, n, m = 28672, 8192, 128
    p = synth.make_problem(k, n, 4, 128, m=m, seed=70)
    a, qw, sc, qz = to_dev(p, cuda_device)
    y_full = XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
    assert y_full.shape == (m, n)

    def same_to_the_last_bit(y1, y2, what):
        d = (y1.view(torch.int16).int() - y2.view(torch.int16).int()).abs()
        near_zero = (y1.float() - y2.float()).abs() <= contract.cancel_slack(k)
        assert bool(((d <= 1) | near_zero).all()) and float((d == 0).float().mean()) > 0.97, what

    for r in (0, 13):  # 16 shards of 512 columns
        qw_r, sc_r, qz_r = sharded.shard_columns(qw, sc, qz, 4, 16, r)
        y_r = XbitOps.gemv(a, qw_r, sc_r, qz_r, 128, 4, k, 1)
        same_to_the_last_bit(y_r, y_full[:, r * 512:(r + 1) * 512].contiguous(), ("shard", r))
    for row in (0, 77, 127):
        y_1 = XbitOps.gemv(a[row:row + 1].contiguous(), qw, sc, qz, 128, 4, k, 1)
        same_to_the_last_bit(y_1, y_full[row:row + 1].contiguous(), ("row", row))
    qw_r, sc_r, qz_r = sharded.shard_columns(p.qweight, p.scales, p.qzeros, 4, 16, 13)
    y_ref = oracle.gemv(p.act, qw_r, sc_r, qz_r, 128, 4, k, 1, oracle.RNE)
    assert_gemv_close(y_full[:, 13 * 512:14 * 512].cpu().numpy(), y_ref, "70B down-projection, 128 rows, shard 13", k=k)


def test_gemv_w4_llama7b_ffn_full_size(cuda_device):
    """BASELINE.json config 2 at full size."""
    p = synth.make_problem(4096, 11008, 4, 128, m=1, seed=2)
    y = run_gemv(p, 1, cuda_device)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, 4096, 1, oracle.RNE)
    assert_gemv_close(y, y_ref, "cfg2", k=p.k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_gemv_bit_sweep_4096x4096_full_size(bits, cuda_device):
    """BASELINE.json configs 1 and 3 at their stated size: in=4096 out=4096 g128 batch 1, b = 2..8 (256
    tiles of 16 columns / the launch shape the real layer gets, not a narrow stand-in)."""
    p = synth.make_problem(4096, 4096, bits, 128, m=1, seed=300 + bits)
    for bias in (0, 1):
        y = run_gemv(p, bias, cuda_device)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, bits, 4096, bias, oracle.RNE)
        assert_gemv_close(y, y_ref, "cfg3 bits=%d bias=%d" % (bits, bias), k=p.k)


@pytest.mark.parametrize("m", [1, 4, 16])
def test_gemv_w3_skinny_4096x11008_full_size(m, cuda_device):
    """BASELINE.json config 5 at its stated size: A16W3 in=4096 out=11008 g128, batch 1 / 4 / 16
    (the 8-wave whole-K launch with 1, 4 and 16 rows per pass)."""
    p = synth.make_problem(4096, 11008, 3, 128, m=m, seed=500 + m)
    y = run_gemv(p, 1, cuda_device)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 3, 4096, 1, oracle.RNE)
    assert_gemv_close(y, y_ref, "cfg5 m=%d" % m, k=p.k)


@pytest.mark.parametrize("m", [4, 16])
def test_gemv_w4_skinny_4096x11008_full_size(m, cuda_device):
    """The metric shape at batch 4 and 16 (skinny-GEMM edge of the headline bit-width)."""
    p = synth.make_problem(4096, 11008, 4, 128, m=m, seed=600 + m)
    y = run_gemv(p, 1, cuda_device)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, 4096, 1, oracle.RNE)
    assert_gemv_close(y, y_ref, "w4 4096x11008 m=%d" % m, k=p.k)


@pytest.mark.parametrize("name", sorted(DIGESTS))
def test_dequant_full_size_digest_shapes(name, cuda_device):
    """The nine full-size shapes whose reference-simulator output is pinned by SHA-256
    (tests/golden/digests.json, oracle/gen_golden.py): the HIP dequant is bit-exact against the
    oracle in hardware rounding (RNE), the oracle in simulator rounding reproduces the pinned digest
    on the same inputs, and the two stay within the reference's own 1e-3 tolerance
    (unpack_weight_2_to_7.cu:570)."""
    d = DIGESTS[name]
    k, n, bits, g = d["k"], d["n"], d["bits"], d["groupsize"]
    p = synth.make_problem(k, n, bits, g, m=1, seed=d["seed"])
    w = run_dequant(p, 0, cuda_device)
    w_rne = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, 0, oracle.RNE)
    assert np.array_equal(_bits(w), _bits(w_rne)), name
    w_sim = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, 0, oracle.SIM)
    assert hashlib.sha256(np.ascontiguousarray(w_sim).tobytes()).hexdigest() == d["sha256_w_ref"], name
    assert np.abs(w.astype(np.float32) - w_sim.astype(np.float32)).max() <= 1e-3, name


def test_mfma_and_valu_accumulate_builds_agree(cuda_device):
    """DESIGN.md's A/B of the north star's "no MFMA" design: the measurement build that accumulates
    on the VALU (v_dot2_f32_f16 + shuffles, variant 64 of the tuning library) must give the production
    (MFMA) kernel's result up to the fp32 summation order, and both must meet the oracle."""
    from xbitops_amd import _lib, ops

    if not os.path.exists(_lib.TUNING_LIB_PATH):
        pytest.skip("tuning library not built (python -m xbitops_amd.build --tuning)")
    for k, n in ((4096, 11008), (4096, 1008), (2048, 512)):
        p = synth.make_problem(k, n, 4, 128, m=1, seed=64 + n)
        a, qw, sc, qz = to_dev(p, cuda_device)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, k, 1, oracle.RNE)
        y_mfma = ops.gemv(a, qw, sc, qz, 128, 4, k, 1)
        y_valu = ops.gemv_ex(a, qw, sc, qz, 128, 4, k, 1, variant=64)
        torch.cuda.synchronize()
        assert_gemv_close(y_mfma.cpu().numpy(), y_ref, "mfma k=%d n=%d" % (k, n), k=k)
        assert_gemv_close(y_valu.cpu().numpy(), y_ref, "valu k=%d n=%d" % (k, n), k=k)
        d = (y_mfma.view(torch.int16).int() - y_valu.view(torch.int16).int()).abs()
        close = (d <= 1) | ((y_mfma.float() - y_valu.float()).abs() <= contract.cancel_slack(k))  # last fp16 bit, or fp32 noise where the sum cancels
        assert bool(close.all()) and float((d == 0).float().mean()) > 0.98
    with pytest.raises(RuntimeError, match="not supported by this build"):  # the production library has no variants
        st = _lib.load().xbit_gemv_f16_ex(y_mfma.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), 4, 128, 1, k, n, 1,
                                          torch.cuda.current_stream().cuda_stream, _lib.GemvOptions(variant=64))
        _lib.check(st, "xbit_gemv_f16_ex")


def test_gemv_linearity_full_size(cuda_device):
    """Size-independent property: y(a1 + a2) == y(a1) + y(a2) in exact arithmetic; with fp16
    activations chosen so that a1 + a2 is exact, the fp32-accumulated results agree to rounding."""
    import XbitOps

    p = synth.make_problem(4096, 11008, 4, 128, m=1, seed=11)
    a1 = (np.round(p.act.astype(np.float32) * 64) / 64).astype(np.float16)
    a2 = np.roll(a1, 7, axis=1)
    dev = cuda_device
    _a, qw, sc, qz = to_dev(p, dev)
    ys = []
    for a in (a1, a2, (a1.astype(np.float32) + a2.astype(np.float32)).astype(np.float16), np.concatenate([a1, a2], 0), np.concatenate([a2, a1, a2, a1], 0), a1):
        ys.append(XbitOps.gemv(torch.from_numpy(a).to(dev), qw, sc, qz, 128, 4, 4096, 1).cpu().numpy())
    y1, y2, y12, yb, yq, y1_again = ys
    scale = np.abs(y1.astype(np.float32)) + np.abs(y2.astype(np.float32)) + 1e-3
    assert (np.abs(y12.astype(np.float32) - (y1.astype(np.float32) + y2.astype(np.float32))) <= 3e-3 * scale).all()
    assert np.array_equal(_bits(y1), _bits(y1_again))  # deterministic from launch to launch
    # batch rows are independent of their position and of their neighbours (bit for bit within the multi-row kernel) ...
    assert np.array_equal(_bits(yb[0]), _bits(yq[1])) and np.array_equal(_bits(yb[1]), _bits(yq[0])) and np.array_equal(_bits(yq[0]), _bits(yq[2]))
    # ... and equal to the one-row launch, which runs another kernel (xbit_gemv_dyn.h) with another summation order, up to
    # the last bit of the fp16 result (both accumulate the same exact W in fp32; |y| >= 2^-6 so an ulp is meaningful)
    for yrow, ysingle in ((yb[0], y1[0]), (yb[1], y2[0])):
        big = np.abs(ysingle.astype(np.float32)) >= 2.0 ** -6
        ulp = np.abs(_bits(yrow).astype(np.int32) - _bits(ysingle).astype(np.int32))
        assert (ulp[big] <= 1).all() and (np.abs(yrow.astype(np.float32) - ysingle.astype(np.float32))[~big] <= 2.0 ** -16).all()


def test_gemv_3d_input_and_bf16_scales(cuda_device):
    import XbitOps

    p = synth.make_problem(1024, 64, 4, 128, m=6, seed=5)
    a, qw, sc, qz = to_dev(p, cuda_device)
    y2d = XbitOps.gemv(a, qw, sc, qz, 128, 4, 1024, 0)
    y3d = XbitOps.gemv(a.view(2, 3, 1024), qw, sc, qz, 128, 4, 1024, 0)
    assert y3d.shape == (2, 3, 64) and torch.equal(y3d.view(6, 64), y2d)
    sc_bf = sc.to(torch.bfloat16)
    yb = XbitOps.gemv(a, qw, sc_bf, qz, 128, 4, 1024, 0)
    assert yb.dtype == torch.bfloat16
    y_exp = XbitOps.gemv(a, qw, sc_bf.to(torch.float16), qz, 128, 4, 1024, 0).to(torch.bfloat16)
    assert torch.equal(yb, y_exp)
    wb = XbitOps.dequant(qw, sc_bf, qz, 128, 4, 1024, 0)
    assert wb.dtype == torch.bfloat16 and wb.shape == (1024, 64)


def test_ops_validation_on_device(cuda_device):
    import XbitOps

    p = synth.make_problem(256, 64, 4, 128, m=1, seed=1)
    a, qw, sc, qz = to_dev(p, cuda_device)
    with pytest.raises(RuntimeError, match="groupsize must be >= 16"):
        XbitOps.gemv(a, qw, sc, qz, 8, 4, 256, 0)
    with pytest.raises(RuntimeError, match="bits must be >= 1 and <= 8"):
        XbitOps.gemv(a, qw, sc, qz, 128, 9, 256, 0)
    with pytest.raises(RuntimeError, match="in_features must be >= 1"):
        XbitOps.gemv(a, qw, sc, qz, 128, 4, 264, 0)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        XbitOps.dequant(qw.t(), sc, qz, 128, 4, 256, 0)
    with pytest.raises(RuntimeError, match="input_a must be float16"):
        XbitOps.gemv(a.float(), qw, sc, qz, 128, 4, 256, 0)
    qw1 = torch.zeros((8, 64), dtype=torch.int32, device=cuda_device)
    with pytest.raises(RuntimeError, match="bits must be in 2..8"):  # reference: prints "error bits" and aborts
        XbitOps.dequant(qw1, sc, torch.zeros((2, 2), dtype=torch.int32, device=cuda_device), 128, 1, 256, 0)


def test_ops_run_on_current_stream_and_keep_device(cuda_device):
    import XbitOps

    p = synth.make_problem(4096, 256, 4, 128, m=1, seed=8)
    a, qw, sc, qz = to_dev(p, cuda_device)
    y_ref = XbitOps.gemv(a, qw, sc, qz, 128, 4, 4096, 1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a2 = a.clone()  # produced on s; the op must be ordered after it on the same stream
        y = XbitOps.gemv(a2, qw, sc, qz, 128, 4, 4096, 1)
    s.synchronize()
    assert torch.equal(y, y_ref)
    assert torch.cuda.current_device() == cuda_device.index


def test_sharded_wrapper_single_rank_uses_hip_op(cuda_device):
    """world size 1: ShardedGemv is the plain HIP op (the multi-rank logic is covered on gloo)."""
    from xbitops_amd import sharded

    p = synth.make_problem(2048, 128, 4, 128, m=2, seed=21)
    a, qw, sc, qz = to_dev(p, cuda_device)
    op = sharded.ShardedGemv(qw, sc, qz, 128, 4, 2048, 1)
    y = op(a)
    torch.cuda.synchronize()
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, 2048, 1)
    assert_gemv_close(y.cpu().numpy(), y_ref, "sharded world=1", k=p.k)
    # a column shard computed on the GPU equals the matching slice of the full result
    qw0, sc0, qz0 = sharded.shard_columns(qw, sc, qz, 4, 2, 1)
    import XbitOps

    y1 = XbitOps.gemv(a, qw0, sc0, qz0, 128, 4, 2048, 1)
    assert torch.equal(y1, y[:, 64:128])


@pytest.mark.parametrize("bits", [2, 3, 4, 6, 8])
def test_adversarial_scales_and_zero_points(bits, cuda_device):
    """Negative, zero, subnormal and large scales; extreme zero-points and bias: the rebuilt W must
    still be the IEEE result bit for bit (subnormal results included), and the GEMV must follow."""
    k, n, g = 512, 96, 64
    p = synth.make_problem(k, n, bits, g, m=2, seed=500 + bits)
    rng = np.random.default_rng(bits)
    sc = p.scales.astype(np.float32)
    kinds = rng.integers(0, 6, size=sc.shape)
    sc = np.where(kinds == 0, -sc, sc)                      # negative
    sc = np.where(kinds == 1, 0.0, sc)                      # zero
    sc = np.where(kinds == 2, sc * 2.0 ** -12, sc)          # results in the fp16 subnormal range
    sc = np.where(kinds == 3, sc * 64.0, sc)                # large
    sc = np.where(kinds == 4, np.float32(6.0e-8), sc)       # smallest subnormal scale
    p.scales = sc.astype(np.float16)
    z = rng.integers(0, 1 << bits, size=(p.groups, n), dtype=np.uint8)
    z[:, ::3] = (1 << bits) - 1
    z[:, 1::5] = 0
    p.qzeros = packing.pack_qzeros(z, bits)
    for bias in (0, 1, 255):
        w = run_dequant(p, bias, cuda_device)
        w_ref = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
        assert np.array_equal(_bits(w), _bits(w_ref)), (bits, bias)
    y = run_gemv(p, 1, cuda_device)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
    ok = np.isfinite(y_ref.astype(np.float32))
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    assert (err[ok] <= 1e-3 * np.abs(y_ref.astype(np.float64))[ok] + 2.0 ** -14).all()


@pytest.mark.parametrize("bits", range(2, 9))
def test_gemv_rebuilds_w_bit_exact_one_hot(bits, cuda_device):
    """One-hot activations make y[m, :] = W[k_m, :] exactly (1.0 * W summed with zeros, one final
    rounding of an fp16 value), so the GEMV's in-register dequantisation — the subnormal-form fast
    path and, for the scales too large for it, the spliced form — is compared BIT FOR BIT with the
    oracle's W for every bit-width, at M = 1, 4, 16 (every rows-per-pass build)."""
    k, n, g = 1024, 176, 128
    p = synth.make_problem(k, n, bits, g, m=16, seed=900 + bits)
    rng = np.random.default_rng(40 + bits)
    sc = p.scales.astype(np.float32)
    kinds = rng.integers(0, 8, size=sc.shape)
    sc = np.where(kinds == 0, -sc, sc)
    sc = np.where(kinds == 1, sc * 2.0 ** -10, sc)          # W in the fp16 subnormal range
    sc = np.where(kinds == 2, np.float32(6.0e-8), sc)       # smallest subnormal scale
    sc = np.where(kinds == 3, sc * 256.0, sc)               # beyond the fast path's scale range
    sc[:, 64:112] = np.abs(sc[:, 64:112]) * 0.5             # whole tiles inside the fast path's range
    p.scales = sc.astype(np.float16)
    w_ref = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
    rows = rng.choice(k, size=16, replace=False)
    act = np.zeros((16, k), np.float16)
    act[np.arange(16), rows] = 1.0
    import XbitOps

    _a, qw, scd, qz = to_dev(p, cuda_device)
    for m in (1, 4, 16):
        a = torch.from_numpy(act[:m]).to(cuda_device)
        y = XbitOps.gemv(a, qw, scd, qz, g, bits, k, 1).cpu().numpy()
        want = w_ref[rows[:m]]
        # -0.0 + 0.0 accumulates to +0.0: compare values, and bits wherever W is not a zero
        nz = want != 0
        assert np.array_equal(_bits(y)[nz], _bits(want)[nz]), (bits, m)
        assert (y[~nz] == 0).all(), (bits, m)


@pytest.mark.parametrize("bits", [3, 4, 8])
def test_gemv_split_k_shapes(bits, cuda_device):
    """Small N and long K: the launcher splits K over workgroups (fixed-point partial sums in scratch,
    the last arriver converts).  Repeated launches reuse the scratch; a second stream gets its own;
    ragged K / N and M = 3, 4 go through the same path.  The last cases force a split factor
    (options.split_k) on shapes the heuristic would not split."""
    import XbitOps
    from xbitops_amd import ops

    for k, n, m, forced in ((16384, 256, 1, 0), (14336 + 512, 130, 3, 0), (14336, 512, 4, 0), (4096 + 512, 130, 3, 3), (8192, 1024, 1, 5)):
        p = synth.make_problem(k, n, bits, 128, m=m, seed=bits * 41 + n + m)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, bits, k, 1, oracle.RNE)
        a, qw, sc, qz = to_dev(p, cuda_device)
        ys = [ops.gemv_ex(a, qw, sc, qz, 128, bits, k, 1, split_k=forced) for _ in range(3)]
        side = torch.cuda.Stream(device=cuda_device)
        side.wait_stream(torch.cuda.current_stream(cuda_device))
        with torch.cuda.stream(side):
            ys.append(ops.gemv_ex(a, qw, sc, qz, 128, bits, k, 1, split_k=forced))
        torch.cuda.synchronize()
        for y in ys:
            assert_gemv_close(y.cpu().numpy(), y_ref, "bits=%d k=%d n=%d m=%d" % (bits, k, n, m), k=k)
            assert torch.equal(y, ys[0])  # fixed-point accumulation: launches agree bit for bit
        y1 = ops.gemv_ex(a, qw, sc, qz, 128, bits, k, 1, split_k=1)  # never split: same result up to the summation order
        assert_gemv_close(y1.cpu().numpy(), y_ref, "unsplit bits=%d k=%d n=%d m=%d" % (bits, k, n, m), k=k)
        if forced == 0:
            assert torch.equal(XbitOps.gemv(a, qw, sc, qz, 128, bits, k, 1), ys[0])  # the plain operator takes the same path


@pytest.mark.parametrize("bits,k,n", [(4, 11008, 4096), (4, 13824, 2560), (3, 13824, 2560), (2, 11008, 4096)])
def test_gemv_measured_plan_shapes(bits, k, n, cuda_device):
    """Down-projection shapes whose K is not a power of two (round 2 kept a table of measured split plans for them; round 3
    splits them by the fill-the-chip-once rule like every few-column long-K layer): full size against the oracle, bit-stable
    across launches, and equal up to summation order to the unsplit launch and to a forced other plan."""
    import XbitOps
    from xbitops_amd import ops

    p = synth.make_problem(k, n, bits, 128, m=1, seed=k + n + bits)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, bits, k, 1, oracle.RNE)
    a, qw, sc, qz = to_dev(p, cuda_device)
    ys = [XbitOps.gemv(a, qw, sc, qz, 128, bits, k, 1) for _ in range(3)]
    torch.cuda.synchronize()
    for y in ys:
        assert_gemv_close(y.cpu().numpy(), y_ref, "measured plan bits=%d k=%d n=%d" % (bits, k, n), k=k)
        assert torch.equal(y, ys[0])
    for forced in (1, 3):
        y = ops.gemv_ex(a, qw, sc, qz, 128, bits, k, 1, split_k=forced)
        assert_gemv_close(y.cpu().numpy(), y_ref, "split_k=%d bits=%d k=%d n=%d" % (forced, bits, k, n), k=k)


def test_split_k_with_caller_workspace(cuda_device):
    """options.workspace: the caller's zero-filled scratch is used (and left zero); one that is too
    small is never used (the launch runs unsplit instead of touching it)."""
    from xbitops_amd import _lib, ops

    k, n = 16384, 512
    p = synth.make_problem(k, n, 4, 128, m=1, seed=77)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, k, 1, oracle.RNE)
    a, qw, sc, qz = to_dev(p, cuda_device)
    need = int(_lib.load().xbit_gemv_workspace_bytes(1, n))
    assert need == 8 * 512
    ws = torch.zeros(need // 8, dtype=torch.int64, device=cuda_device)
    y = ops.gemv_ex(a, qw, sc, qz, 128, 4, k, 1, workspace=ws)
    y2 = ops.gemv_ex(a, qw, sc, qz, 128, 4, k, 1, workspace=ws)  # reusable at once: the kernel left it zero
    torch.cuda.synchronize()
    assert_gemv_close(y.cpu().numpy(), y_ref, "caller workspace", k=k)
    assert torch.equal(y, y2) and int(ws.abs().sum()) == 0
    assert torch.equal(y, ops.gemv(a, qw, sc, qz, 128, 4, k, 1))  # library-managed scratch: same split, same bits
    poisoned = torch.full((16,), -1, dtype=torch.int64, device=cuda_device)  # too small: must not be touched
    y3 = ops.gemv_ex(a, qw, sc, qz, 128, 4, k, 1, workspace=poisoned)
    torch.cuda.synchronize()
    assert_gemv_close(y3.cpu().numpy(), y_ref, "workspace too small -> unsplit", k=k)
    assert bool((poisoned == -1).all())
    assert _lib.load().xbit_workspace_reset() == 0
    assert torch.equal(ops.gemv(a, qw, sc, qz, 128, 4, k, 1), y)


def test_two_captured_split_k_graphs_replay_concurrently(cuda_device):
    """two split-K launches captured into two hipGraphs (torch captures both on one stream)
    and replayed AT THE SAME TIME on two streams, beside eager split-K launches on a third: each captured
    launch owns a private scratch range, so every replay gives the eager result bit for bit."""
    from xbitops_amd import ops

    cases = []
    for k, n, seed in ((16384, 512, 5), (28672, 256, 6)):
        p = synth.make_problem(k, n, 4, 128, m=1, seed=seed)
        a, qw, sc, qz = to_dev(p, cuda_device)
        want = ops.gemv(a, qw, sc, qz, 128, 4, k, 1)
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, k, 1, oracle.RNE)
        assert_gemv_close(want.cpu().numpy(), y_ref, "eager k=%d" % k, k=k)
        cases.append((k, a, qw, sc, qz, want))
    torch.cuda.synchronize()
    graphs = []
    for k, a, qw, sc, qz, want in cases:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = ops.gemv(a, qw, sc, qz, 128, 4, k, 1)
        graphs.append((g, y, want))
    streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]
    bad = []
    for rep in range(30):
        for (g, y, _want), st in zip(graphs, streams):
            with torch.cuda.stream(st):
                y.zero_()
                g.replay()
        with torch.cuda.stream(streams[2]):
            k, a, qw, sc, qz, want_eager = cases[rep % 2]
            y_eager = ops.gemv(a, qw, sc, qz, 128, 4, k, 1)
        torch.cuda.synchronize()
        for i, (g, y, want) in enumerate(graphs):
            if not torch.equal(y, want):
                bad.append((rep, i))
        if not torch.equal(y_eager, want_eager):
            bad.append((rep, "eager"))
    assert not bad, bad


def test_gemv_random_shapes(cuda_device):
    """Seeded random shapes through every launch path the heuristics can take (whole-K tiles of
    16/32/48/64/128 columns, 1/4/8/16 rows per pass, forced and automatic split-K, the generic
    kernel for K % 8 != 0 or groups that are not a multiple of 32), ragged K and N included."""
    from xbitops_amd import ops

    rng = np.random.default_rng(2024)
    for case in range(48):
        bits = int(rng.integers(2, 9))
        g = int(rng.choice([16, 32, 64, 96, 128, 256]))
        k = int(rng.integers(1, 40)) * g + int(rng.choice([0, 0, 8, 24, g // 2]))
        if rng.random() < 0.15:
            k += 4  # K % 8 != 0: generic kernel
        n = int(rng.choice([2, 34, 96, 130, 256, 1000, 4096 + 6, 11008, 25600 + 62]))
        m = int(rng.choice([1, 1, 1, 2, 4, 5, 8, 13, 16, 21]))
        if k * n > 3e7:
            k = max(g, (int(3e7 // n) // g) * g)
        forced = int(rng.choice([0, 0, 0, 2, 3, 5]))
        bias = int(rng.integers(0, 2))
        p = synth.make_problem(k, n, bits, g, m=m, seed=7000 + case)
        a, qw, sc, qz = to_dev(p, cuda_device)
        y = ops.gemv_ex(a, qw, sc, qz, g, bits, k, bias, split_k=forced).cpu().numpy()
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
        assert_gemv_close(y, y_ref, "case %d bits=%d k=%d n=%d g=%d m=%d bias=%d split=%d" % (case, bits, k, n, g, m, bias, forced), k=k)


def test_dequant_random_shapes_bit_exact(cuda_device):
    """Seeded random shapes through both dequant kernels (tile kernel: n % 8 == 0; generic kernel
    otherwise), ragged K, every group size class: bit-exact against the oracle (RNE)."""
    rng = np.random.default_rng(77)
    for case in range(40):
        bits = int(rng.integers(2, 9))
        g = int(rng.choice([16, 32, 48, 64, 128, 256]))
        k = int(rng.integers(1, 24)) * g + int(rng.choice([0, 0, 1, 7, 8, g // 2]))
        n = int(rng.choice([2, 8, 34, 96, 136, 512, 1000, 4096 + 8]))
        bias = int(rng.integers(0, 2))
        p = synth.make_problem(k, n, bits, g, m=1, seed=8000 + case)
        w = run_dequant(p, bias, cuda_device)
        w_ref = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
        assert np.array_equal(_bits(w), _bits(w_ref)), "case %d bits=%d k=%d n=%d g=%d bias=%d" % (case, bits, k, n, g, bias)


def test_concurrent_streams_and_threads(cuda_device):
    """include/xbitops_hip.h promises thread safety: four host threads, each on its own stream, run
    split-K and plain GEMVs concurrently (split-K scratch is per stream); every result must equal
    the single-threaded one bit for bit."""
    import threading

    import XbitOps

    cases = []
    for k, n, m in ((16384, 256, 1), (4096, 1024, 1), (14336, 512, 3)):
        p = synth.make_problem(k, n, 4, 128, m=m, seed=k + n)
        a, qw, sc, qz = to_dev(p, cuda_device)
        want = XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
        cases.append((k, a, qw, sc, qz, want))
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            stream = torch.cuda.Stream(device=cuda_device)
            with torch.cuda.stream(stream):
                for it in range(40):
                    k, a, qw, sc, qz, want = cases[(it + tid) % len(cases)]
                    y = XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
                    if it % 8 == 7:
                        stream.synchronize()
                        if not torch.equal(y, want):
                            errors.append("thread %d iteration %d: result differs" % (tid, it))
            stream.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (tid, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_split_k_shape_inside_graph_capture(cuda_device):
    """A split-K-eligible GEMV captured into a hipGraph: on a stream that has no scratch buffer yet
    the launch must not allocate inside the capture (it runs unsplit); after a warm-up call the
    captured launch is split.  Both graphs replay to the oracle's result."""
    import XbitOps

    k, n = 16384, 512
    p = synth.make_problem(k, n, 4, 128, m=1, seed=99)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, k, 1, oracle.RNE)
    a, qw, sc, qz = to_dev(p, cuda_device)
    for warm in (False, True):
        stream = torch.cuda.Stream(device=cuda_device)  # a fresh stream: no scratch buffer of its own
        with torch.cuda.stream(stream):
            if warm:
                XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
                stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                y = XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
            for _ in range(3):
                y.zero_()
                graph.replay()
            stream.synchronize()
        assert_gemv_close(y.cpu().numpy(), y_ref, "graph replay, warm=%s" % warm, k=k)


def test_capture_pool_release_hands_the_ranges_out_again(cuda_device):
    """Every captured split-K launch keeps a private range of the per-device pool; xbit_workspace_capture_bytes_free
    reports what is left, xbit_workspace_release hands everything out again, and a graph captured afterwards still
    gives the oracle's result."""
    import XbitOps
    from xbitops_amd import _lib

    lib = _lib.load()
    k, n = 16384, 512
    p = synth.make_problem(k, n, 4, 128, m=1, seed=98)
    y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, k, 1, oracle.RNE)
    a, qw, sc, qz = to_dev(p, cuda_device)
    assert lib.xbit_workspace_reset() == 0  # creates the pool
    assert lib.xbit_workspace_release() == 0
    whole = lib.xbit_workspace_capture_bytes_free()
    assert whole == 8 << 20
    stream = torch.cuda.Stream(device=cuda_device)
    graphs = []
    with torch.cuda.stream(stream):
        XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
        stream.synchronize()
        for _ in range(2):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                y = XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
            graphs.append((graph, y))
        after_two = lib.xbit_workspace_capture_bytes_free()
        assert whole - after_two == 2 * 8 * n  # 8 bytes per output element and captured launch
        for graph, y in graphs:
            graph.replay()
        stream.synchronize()
        for _graph, y in graphs:
            assert_gemv_close(y.cpu().numpy(), y_ref, "captured split-K", k=k)
        del graphs, graph
        assert lib.xbit_workspace_release() == 0
        assert lib.xbit_workspace_capture_bytes_free() == whole
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            y = XbitOps.gemv(a, qw, sc, qz, 128, 4, k, 1)
        assert lib.xbit_workspace_capture_bytes_free() == whole - 8 * n
        y.zero_()
        graph.replay()
        stream.synchronize()
    assert_gemv_close(y.cpu().numpy(), y_ref, "captured after release", k=k)


@pytest.mark.parametrize("bits", [3, 4, 8])
def test_gemv_multi_fused_launch(bits, cuda_device):
    """xbit_gemv_multi_f16: q/k/v-style (equal N), gate/up-style, ragged and single sets in one launch at
    1..4 rows; more rows and group sizes the fused build does not take fall back to one launch per
    set.  Every output is checked against the oracle and against the single-set operator."""
    import XbitOps
    from xbitops_amd import ops

    for k, ns, g, m in ((4096, (1024, 1024, 1024), 128, 1), (2048, (2752, 2752), 128, 1), (1024, (130, 66, 258, 34), 128, 1),
                        (2048, (512,), 128, 1), (1024, (256, 512), 128, 3), (2048, (1024, 1024, 512), 128, 4), (1024, (256, 130), 128, 6),
                        (1024, (256, 256), 32, 1), (8192, (96, 160), 128, 1)):
        sets, refs = [], []
        act = None
        for j, n in enumerate(ns):
            p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 13 + n + 7 * j)
            if act is None:
                act = p.act
            refs.append(oracle.gemv(act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE))
            sets.append(tuple(torch.from_numpy(x).to(cuda_device) for x in (p.qweight, p.scales, p.qzeros)))
        a = torch.from_numpy(act).to(cuda_device)
        ys = ops.gemv_multi(a, sets, g, bits, k, 1)
        torch.cuda.synchronize()
        assert len(ys) == len(ns)
        for y, ref, (qw, sc, qz), n in zip(ys, refs, sets, ns):
            assert y.shape == (m, n)
            assert_gemv_close(y.cpu().numpy(), ref, "multi bits=%d k=%d ns=%s n=%d m=%d" % (bits, k, ns, n, m), k=k)
            single = XbitOps.gemv(a, qw, sc, qz, g, bits, k, 1)
            assert_gemv_close(y.cpu().numpy(), single.cpu().numpy().astype(np.float16), "multi vs single", k=k)
    with pytest.raises(RuntimeError):
        ops.gemv_multi(a, [], 128, bits, k, 1)


def test_split_k_overflow_and_nan_follow_the_unsplit_result(cuda_device):
    """Split-K accumulates in fixed point; partial sums outside its range, Inf and NaN must still give
    what the unsplit kernel stores: +Inf / -Inf on overflow, NaN for NaN (or Inf - Inf)."""
    from xbitops_amd import ops

    k, n = 16384, 256
    p = synth.make_problem(k, n, 4, 128, m=1, seed=321)
    act = p.act.copy()
    sc = p.scales.astype(np.float32)
    sc[:, 0:64] = 200.0     # columns 0-63: huge positive/negative sums -> +-Inf in fp16
    p.scales = sc.astype(np.float16)
    a, qw, scd, qz = (torch.from_numpy(x).to(cuda_device) for x in (act, p.qweight, p.scales, p.qzeros))
    want = ops.gemv_ex(a, qw, scd, qz, 128, 4, k, 1, split_k=1).cpu().numpy()
    got = ops.gemv_ex(a, qw, scd, qz, 128, 4, k, 1, split_k=0).cpu().numpy()
    big = ~np.isfinite(want)
    assert big[:, :64].any()                       # the test does overflow
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[big & ~np.isnan(want)], want[big & ~np.isnan(want)])   # same signed infinities
    assert_gemv_close(got[~big], want[~big], "finite columns", k=k)
    act[0, 7] = np.float16(np.nan)                 # a NaN activation poisons every output
    a = torch.from_numpy(act).to(cuda_device)
    got = ops.gemv(a, qw, scd, qz, 128, 4, k, 1).cpu().numpy()
    assert np.isnan(got).all()


def test_split_k_cancellation_between_k_ranges(cuda_device):
    """partial sums of opposite sign in different K ranges.  Inside the fixed-point range
    (|partial| < 2^19) the split kernel adds them exactly, so a huge +/- pair cancels to the small
    finite total the unsplit fp32 sum gives; beyond it the documented contract is saturation."""
    from xbitops_amd import ops

    k, n, g = 16384, 256, 128
    q = np.full((k, n), 7, np.uint8)
    q[5, :] = 9          # one weight differs: total = W(9) - W(7) ... times its activation
    z = np.full((k // g, n), 3, np.uint8)
    act = np.ones((1, k), np.float16)
    act[0, k // 2:] = -1.0   # first K half sums to +8192 * W, second half to -8192 * W
    for scale, want_finite in ((8.0, True), (32.0, False)):   # partials of 2^18 (in range) and 2^20 (saturate)
        sc = np.full((k // g, n), scale, np.float16)
        qw, qz = packing.pack_qweight(q, 4), packing.pack_qzeros(z, 4)
        y_ref = oracle.gemv(act, qw, sc, qz, g, 4, k, 0, oracle.RNE)
        assert np.all(y_ref.astype(np.float32) == 2 * scale)  # (9 - 3) * s - (7 - 3) * s: everything else cancels
        a, qwd, scd, qzd = (torch.from_numpy(x).to(cuda_device) for x in (act, qw, sc, qz))
        y1 = ops.gemv_ex(a, qwd, scd, qzd, g, 4, k, 0, split_k=1).cpu().numpy()
        y2 = ops.gemv_ex(a, qwd, scd, qzd, g, 4, k, 0, split_k=2).cpu().numpy()
        assert np.array_equal(y1, y_ref)  # unsplit: fp32 accumulation of exactly representable terms
        if want_finite:
            assert np.array_equal(y2, y_ref)
        else:
            assert np.isnan(y2.astype(np.float32)).all()  # +2^20 in one range, -2^20 in the other: both saturate
