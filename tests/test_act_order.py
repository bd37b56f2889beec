"""act-order (``g_idx``) checkpoints and the transposed 4-bit layout.

CPU: the oracle's act-order restatement (quant_cuda_kernel.cu:257-336) reduces to the contiguous-group
oracle for g_idx[k] = k // groupsize, the load-time row sort is exact, and the transposed layout of
gemv_w4a16_nt.cu:64-159 is the three transposes the packer says it is.  GPU: the HIP act-order kernels
and the QuantLinear paths against the oracle."""
import numpy as np
import pytest
import torch

import oracle
from oracle import contract
from xbitops_amd import packing, synth


def _bits(x):
    return np.ascontiguousarray(x).view(np.uint16)


def _act_order_problem(k, n, bits, g, m, seed, equal_groups=True):
    p = synth.make_problem(k, n, bits, g, m=m, seed=seed)
    rng = np.random.default_rng(seed)
    groups = p.groups
    if equal_groups:  # what GPTQ desc_act produces: a permutation of k // g
        order = rng.permutation(k)
        g_idx = np.empty(k, np.int32)
        g_idx[order] = np.arange(k) // g
    else:  # arbitrary group sizes
        g_idx = rng.integers(0, groups, size=k).astype(np.int32)
    return p, g_idx


@pytest.mark.parametrize("bits", range(2, 9))
def test_oracle_act_order_reduces_to_contiguous_groups(bits):
    p = synth.make_problem(320, 24, bits, 64, m=3, seed=40 + bits)
    g_idx = (np.arange(320) // 64).astype(np.int32)
    for bias in (0, 1):
        for mode in (oracle.RNE, oracle.SIM):
            w0 = oracle.dequant(p.qweight, p.scales, p.qzeros, 64, bits, 320, bias, mode)
            w1 = oracle.dequant_gidx(p.qweight, p.scales, p.qzeros, g_idx, bits, 320, bias, mode)
            assert np.array_equal(_bits(w0), _bits(w1))
        y0 = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 64, bits, 320, bias)
        y1 = oracle.gemv_gidx(p.act, p.qweight, p.scales, p.qzeros, g_idx, bits, 320, bias)
        assert np.array_equal(_bits(y0), _bits(y1))


def test_oracle_act_order_known_answer():
    """Hand-built: 4-bit, two groups, rows alternate between them; W[k] = (q - z[g]) * s[g] with g = g_idx[k]
    and the kernel's '+1' as add_zero_bias (quant_cuda_kernel.cu:299-300)."""
    k, n = 8, 2
    q = np.arange(8, dtype=np.uint8).reshape(8, 1).repeat(2, axis=1)
    qw = packing.pack_qweight(q, 4)
    s = np.array([[0.5, 0.25], [2.0, 1.0]], np.float16)
    z = np.array([[1, 2], [3, 0]], np.uint8)
    qz = packing.pack_qzeros(z, 4)
    g_idx = np.array([0, 1, 0, 1, 1, 1, 0, 0], np.int32)
    w = oracle.dequant_gidx(qw, s, qz, g_idx, 4, k, 1).astype(np.float32)
    want = (q.astype(np.float32) - (z[g_idx].astype(np.float32) + 1)) * s[g_idx].astype(np.float32)
    assert np.array_equal(w, want)


@pytest.mark.parametrize("bits", [3, 4, 8])
def test_sort_rows_by_group_is_exact(bits):
    k, n, g = 384, 20, 64
    p, g_idx = _act_order_problem(k, n, bits, g, 2, 7 + bits)
    qs, perm = packing.sort_rows_by_group(p.qweight, g_idx, bits, g)
    assert np.array_equal(np.sort(perm), np.arange(k)) and np.array_equal(g_idx[perm], np.arange(k) // g)
    w_act = oracle.dequant_gidx(p.qweight, p.scales, p.qzeros, g_idx, bits, k, 1)
    w_sorted = oracle.dequant(qs, p.scales, p.qzeros, g, bits, k, 1)
    assert np.array_equal(_bits(w_sorted), _bits(w_act[perm]))  # W_sorted[k'] = W[perm[k']]
    with pytest.raises(ValueError):
        packing.sort_rows_by_group(p.qweight, np.zeros(k, np.int32), bits, g)  # one group holds every row


def test_transposed_w4_layout_is_three_transposes():
    """gemv_w4a16_nt.cu:64-159 indexes qweight_t[n][k/8], scales_t[n][g], qzeros_t[n/8][g]; the oracle's
    direct restatement of that indexing equals the ordinary oracle on the transposed tensors."""
    k, n, g = 512, 64, 128
    p = synth.make_problem(k, n, 4, g, m=2, seed=31)
    qt, st, zt = (np.ascontiguousarray(x.T) for x in (p.qweight, p.scales, p.qzeros))
    for bias in (0, 1):
        y_t = oracle.gemv_w4_transposed(p.act, qt, st, zt, g, k, bias)
        y = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, 4, k, bias)
        assert np.array_equal(_bits(y_t), _bits(y))
    qw, sc, qz = packing.from_transposed_w4(qt, st, zt)
    assert np.array_equal(qw, p.qweight) and np.array_equal(_bits(sc), _bits(p.scales)) and np.array_equal(qz, p.qzeros)


# ---------------------------------------------------------------------------------------------- GPU
def _to_dev(dev, *arrays):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]


def _close(y, y_ref, what, k=None):
    assert k is not None
    contract.assert_gemv_close(y, y_ref, k, what)  # oracle/contract.py: the one GEMV contract


@pytest.mark.gpu
@pytest.mark.parametrize("bits", range(2, 9))
def test_act_order_kernels_vs_oracle(bits, cuda_device):
    from xbitops_amd import ops

    for k, n, g, m, equal in ((4096, 256, 128, 1, True), (1000, 68, 40, 3, False), (8192, 64, 32, 2, True), (4096 + 64, 132, 128, 1, False),
                              (96, 6, 16, 1, False), (4096, 200, 128, 8, False), (8192, 96, 64, 16, True), (2048, 144, 128, 13, False),
                              (4096, 64, 128, 40, True), (1000, 68, 40, 7, False),  # 5..16 rows: the 8- / 16-row by-id builds; 40: 32 + 8
                              (11008, 96, 128, 1, True), (11008, 96, 128, 4, False), (11008, 64, 128, 20, True), (4096, 130, 128, 27, False)):  # K beyond the 1- / 4-row builds' staging: the 8-row build; 17..32 rows: the 32-row build
        p, g_idx = _act_order_problem(k, n, bits, g, m, 100 * bits + n, equal)
        a, qw, sc, qz, gi = _to_dev(cuda_device, p.act, p.qweight, p.scales, p.qzeros, g_idx)
        w = ops.dequant_gidx(qw, sc, qz, gi, bits, k, 1).cpu().numpy()
        w_ref = oracle.dequant_gidx(p.qweight, p.scales, p.qzeros, g_idx, bits, k, 1, oracle.RNE)
        assert np.array_equal(_bits(w), _bits(w_ref)), (bits, k, n)  # bit-exact
        y = ops.gemv_gidx(a, qw, sc, qz, gi, bits, k, 1).cpu().numpy()
        y_ref = oracle.gemv_gidx(p.act, p.qweight, p.scales, p.qzeros, g_idx, bits, k, 1, oracle.RNE)
        assert y.shape == (m, n)
        _close(y, y_ref, "bits=%d k=%d n=%d g=%d m=%d" % (bits, k, n, g, m), k=k)


@pytest.mark.gpu
def test_act_order_full_size_and_many_groups(cuda_device):
    """The metric shape with act-order (32 groups), and K = 28672 (224 groups: the group table takes
    56 KB of LDS) - plus > 512 groups, which goes to the general fallback kernel."""
    from xbitops_amd import ops

    for k, n, g in ((4096, 11008, 128), (28672, 128, 128), (8192 + 4096, 64, 16)):
        p, g_idx = _act_order_problem(k, n, 4, g, 1, k + n)
        a, qw, sc, qz, gi = _to_dev(cuda_device, p.act, p.qweight, p.scales, p.qzeros, g_idx)
        y = ops.gemv_gidx(a, qw, sc, qz, gi, 4, k, 1).cpu().numpy()
        y_ref = oracle.gemv_gidx(p.act, p.qweight, p.scales, p.qzeros, g_idx, 4, k, 1, oracle.RNE)
        _close(y, y_ref, "k=%d n=%d g=%d" % (k, n, g), k=k)


@pytest.mark.gpu
@pytest.mark.parametrize("equal", [True, False])
def test_quant_linear_act_order(equal, cuda_device):
    """QuantLinear.from_act_order: equal-sized groups are sorted at load (ordinary kernels + permuted
    activation), anything else keeps g_idx (general kernels); both match the act-order oracle."""
    from xbitops_amd.linear import QuantLinear

    k, n, g, bits = 2048, 256, 128, 4
    p, g_idx = _act_order_problem(k, n, bits, g, 140, 5, equal)
    layer = QuantLinear.from_act_order(p.qweight, p.scales, p.qzeros, g_idx, bits, g, add_zero_bias=1).to(cuda_device)
    assert (layer.perm is not None) == equal and (layer.g_idx is not None) == (not equal)
    w_ref = oracle.dequant_gidx(p.qweight, p.scales, p.qzeros, g_idx, bits, k, 1, oracle.RNE)
    assert np.array_equal(_bits(layer.dequantize().cpu().numpy()), _bits(w_ref))
    for rows in (1, 4, 20, 40, 140):  # fused kernel (one launch, several launches) and, at 140 rows, dequant + dense GEMM
        x = torch.from_numpy(p.act[:rows]).to(cuda_device)
        y = layer(x).cpu().numpy()
        y_ref = oracle.gemv_gidx(p.act[:rows], p.qweight, p.scales, p.qzeros, g_idx, bits, k, 1, oracle.RNE)
        dense = rows > 128  # fp16 GEMM of the dequantised weight
        tol = 2e-3 if dense else 1e-3
        assert (np.abs(y.astype(np.float64) - y_ref.astype(np.float64)) <= tol * np.abs(y_ref.astype(np.float64)) + 1e-3 * dense + 2.0 ** -14).all(), rows


@pytest.mark.gpu
def test_quant_linear_from_transposed_layout(cuda_device):
    from xbitops_amd.linear import QuantLinear

    k, n, g = 4096, 512, 128
    p = synth.make_problem(k, n, 4, g, m=2, seed=61)
    qt, st, zt = (np.ascontiguousarray(x.T) for x in (p.qweight, p.scales, p.qzeros))
    layer = QuantLinear.from_transposed_w4(qt, st, zt, g, k, add_zero_bias=0).to(cuda_device)
    y = layer(torch.from_numpy(p.act).to(cuda_device)).cpu().numpy()
    y_ref = oracle.gemv_w4_transposed(p.act, qt, st, zt, g, k, 0, oracle.RNE)  # the reference kernel's own indexing
    _close(y, y_ref, "transposed layout", k=k)
