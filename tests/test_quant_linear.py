"""QuantLinear: the packer's quantise->pack round trip (CPU) and the module on the GPU."""
import numpy as np
import pytest
import torch

import oracle
from xbitops_amd import _lib, packing


@pytest.mark.parametrize("bits,bias", [(2, 0), (3, 1), (4, 1), (8, 0)])
def test_quantize_pack_round_trip(bits, bias):
    rng = np.random.default_rng(bits)
    w = rng.normal(0, 0.05, size=(256, 64)).astype(np.float32)
    qweight, scales, qzeros = packing.quantize_rtn(w, bits, 64, add_zero_bias=bias)
    q = oracle.unpack_q(qweight, bits, 256).astype(np.float32)
    z = oracle.unpack_z(qzeros, bits, 64).astype(np.float32) + bias
    gi = np.arange(256) // 64
    recon = (q - z[gi]) * scales.astype(np.float32)[gi]
    step = scales.astype(np.float32)[gi]
    # within half a quantisation step, plus the fp16 rounding of the scale seen through up to maxq steps
    assert (np.abs(recon - w) <= (0.5 + ((1 << bits) - 1) * 2.0 ** -10) * step + 1e-6).all()
    w_deq = oracle.dequant(qweight, scales, qzeros, 64, bits, 256, bias).astype(np.float32)
    assert np.abs(w_deq - recon).max() <= 2e-3 * np.abs(recon).max() + 1e-4  # fp16 rounding of the same value


def _oracle_forward(layer, x):
    """oracle GEMV on the module's own packed buffers (copied to the host): the independent checker."""
    qw, sc, qz = (t.detach().cpu().numpy() for t in (layer.qweight, layer.scales, layer.qzeros))
    a = x.detach().reshape(-1, layer.in_features).cpu().numpy()
    return oracle.gemv(a, qw, sc, qz, layer.groupsize, layer.bits, layer.in_features, layer.add_zero_bias, oracle.RNE)


def _assert_close_to_oracle(y, y_ref, what):
    yf, rf = y.detach().float().cpu().numpy().astype(np.float64), y_ref.astype(np.float64)
    assert yf.shape == rf.shape, what
    assert (np.abs(yf - rf) <= 1e-3 * np.abs(rf) + 2.0 ** -14).all(), what


@pytest.mark.gpu
@pytest.mark.parametrize("bits,bias", [(4, 1), (3, 0), (8, 1)])
def test_quant_linear_forward(bits, bias, cuda_device):
    from xbitops_amd.linear import QuantLinear

    torch.manual_seed(bits)
    w = torch.randn(512, 128) * 0.05
    layer = QuantLinear.from_float(w, bits=bits, groupsize=128, add_zero_bias=bias).to(cuda_device)
    qw, sc, qz = (t.cpu().numpy() for t in (layer.qweight, layer.scales, layer.qzeros))
    w_ref = oracle.dequant(qw, sc, qz, 128, bits, 512, bias, oracle.RNE)
    w_deq = layer.dequantize()
    assert np.array_equal(w_deq.cpu().numpy().view(np.uint16), w_ref.view(np.uint16))  # bit-exact vs the oracle
    assert (w_deq.float().cpu() - w).abs().max() < (0.05 if bits >= 4 else 0.12)
    for rows in (1, 5, 16, 40, 128, 200, 400):  # fused path (up to GEMV_MAX_ROWS rows): the oracle GEMV on the same packed buffers
        x = (torch.randn(rows, 512, device=cuda_device) * 0.5).half()
        y = layer(x)
        assert y.shape == (rows, 128)
        _assert_close_to_oracle(y, _oracle_forward(layer, x), "bits=%d rows=%d" % (bits, rows))
    # dequant + dense GEMM path (more rows): fp16 GEMM of the oracle's W, fp32 accumulation order unknown
    x = (torch.randn(600, 512, device=cuda_device) * 0.5).half()
    y = layer(x)
    y_ref = oracle.matvec_f16(x.cpu().numpy(), w_ref).astype(np.float64)
    assert y.shape == (600, 128)
    assert (np.abs(y.float().cpu().numpy() - y_ref) <= 2e-3 * np.abs(y_ref) + 1e-3).all()
    assert layer(torch.zeros(2, 3, 512, device=cuda_device).half()).shape == (2, 3, 128)


@pytest.mark.gpu
def test_quant_linear_long_k_stays_on_the_fused_path(cuda_device):
    """A down-projection with many groups (12288 / 64 = 192: their tables do not fit in LDS beside the activation images) at prefill-sized row counts: the module keeps the
    fused path up to GEMV_MAX_ROWS (the rows kernel's windowed-table builds; profiles/r04/linear_crossover_long_k.txt) and equals the oracle on its own packed buffers."""
    from xbitops_amd.linear import QuantLinear

    torch.manual_seed(11)
    k, n = 12288, 3072  # (a layer of fewer than 96 tiles of 32 columns takes 16-row passes up to 128 rows: xbit_gemv_rules.h)
    layer = QuantLinear.from_float(torch.randn(k, n) * 0.05, bits=4, groupsize=64, add_zero_bias=1).to(cuda_device)
    for rows in (200, 400):
        x = (torch.randn(rows, k, device=cuda_device) * 0.5).half()
        recs = _lib.dry_run(_lib.load(), lambda: layer(x))
        assert len(recs) == 1 and "gemv_rows_kernel<4," in recs[0]["kernel"] and ", true>" in recs[0]["kernel"], recs
        _assert_close_to_oracle(layer(x), _oracle_forward(layer, x), "long K rows=%d" % rows)


@pytest.mark.gpu
def test_gemv_plan_equals_gemv(cuda_device):
    """ops.GemvPlan (weight-side checks done once) gives the operator's result bit for bit, rejects
    what the operator rejects, and QuantLinear rebuilds its plan when its buffers are replaced."""
    import XbitOps
    from xbitops_amd import ops, synth
    from xbitops_amd.linear import QuantLinear

    p = synth.make_problem(1024, 256, 4, 128, m=3, seed=11)
    a, qw, sc, qz = (torch.from_numpy(x).to(cuda_device) for x in (p.act, p.qweight, p.scales, p.qzeros))
    plan = ops.GemvPlan(qw, sc, qz, 128, 4, 1024, 1)
    assert torch.equal(plan(a), XbitOps.gemv(a, qw, sc, qz, 128, 4, 1024, 1))
    _assert_close_to_oracle(plan(a), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, 4, 1024, 1, oracle.RNE), "GemvPlan")
    a3 = a.reshape(1, 3, 1024)
    assert torch.equal(plan(a3), XbitOps.gemv(a3, qw, sc, qz, 128, 4, 1024, 1))
    with pytest.raises(RuntimeError):
        plan(a.float())
    with pytest.raises(RuntimeError):
        plan(a[:, :512].contiguous())
    with pytest.raises(RuntimeError):
        ops.GemvPlan(qw, sc, qz, 128, 4, 1000, 1)  # in_features does not match qweight
    layer = QuantLinear.from_float(torch.randn(512, 64) * 0.05, bits=4, groupsize=128).to(cuda_device)
    x = (torch.randn(2, 512, device=cuda_device) * 0.5).half()
    y1 = layer(x)
    first = layer._plan
    assert layer(x) is not None and layer._plan is first
    layer.qweight = layer.qweight.clone()  # a replaced buffer: the plan must follow
    assert torch.equal(layer(x), y1) and layer._plan is not first


@pytest.mark.gpu
def test_quant_linear_group(cuda_device):
    """QuantLinearGroup: one token -> one fused launch, outputs equal to the layers' own."""
    from xbitops_amd.linear import QuantLinear, QuantLinearGroup

    torch.manual_seed(1)
    layers = [QuantLinear.from_float(torch.randn(1024, n) * 0.05, bits=4, groupsize=128, add_zero_bias=1).to(cuda_device) for n in (256, 128, 128)]
    group = QuantLinearGroup(layers)
    for rows in (1, 3):
        x = (torch.randn(rows, 1024, device=cuda_device) * 0.5).half()
        ys = group(x)
        assert len(ys) == 3
        for layer, y in zip(layers, ys):
            _assert_close_to_oracle(y, _oracle_forward(layer, x), "group rows=%d n=%d" % (rows, layer.out_features))  # the checker
            want = layer(x)  # and the layer on its own: same kernel family, agreement to the last fp16 bit
            assert y.shape == want.shape
            assert (y.float() - want.float()).abs().max() <= 1e-3 * want.float().abs().max() + 2.0 ** -14
    with pytest.raises(ValueError):
        QuantLinearGroup([layers[0], QuantLinear.from_float(torch.randn(512, 64) * 0.05, bits=4, groupsize=128).to(cuda_device)])


def _act_order_layers():
    from xbitops_amd import synth
    from xbitops_amd.linear import QuantLinear

    k, n, bits, g = 256, 64, 4, 64
    p = synth.make_problem(k, n, bits, g, m=1, seed=3)
    rng = np.random.default_rng(0)
    gi = np.repeat(np.arange(k // g), g)
    rng.shuffle(gi)
    equal = QuantLinear.from_act_order(p.qweight, p.scales, p.qzeros, gi, bits, g, 1)  # rows sorted at load time: `perm`
    gi2 = gi.copy()
    gi2[:5] = 0
    general = QuantLinear.from_act_order(p.qweight, p.scales, p.qzeros, gi2, bits, g, 1)  # unequal groups: `g_idx` kept
    assert equal.perm is not None and equal.g_idx is None and general.g_idx is not None and general.perm is None
    return (k, n, bits, g), equal, general


def test_act_order_state_dict_round_trip():
    """A fresh layer must come out of load_state_dict with the permutation / g_idx of the saved one (strict AND non-strict);
    before, strict loading raised on the extra key and strict=False dropped it silently - wrong results afterwards."""
    from xbitops_amd.linear import QuantLinear

    (k, n, bits, g), equal, general = _act_order_layers()
    for src, name in ((equal, "perm"), (general, "g_idx")):
        sd = src.state_dict()
        assert name in sd
        for strict in (True, False):
            dst = QuantLinear(k, n, bits, g, 1)
            res = dst.load_state_dict(sd, strict=strict)
            assert not res.missing_keys and not res.unexpected_keys
            assert torch.equal(getattr(dst, name), getattr(src, name))
            assert torch.equal(dst.qweight, src.qweight)
            other = "g_idx" if name == "perm" else "perm"
            assert getattr(dst, other) is None
    # a plain checkpoint loaded into a layer that held a permutation drops it (the rows it described are gone)
    equal.load_state_dict(QuantLinear(k, n, bits, g, 1).state_dict())
    assert equal.perm is None and equal.g_idx is None
    # a permutation of the wrong length is an error, not a silent mismatch
    bad = dict(general.state_dict())
    bad["g_idx"] = bad["g_idx"][:-1]
    with pytest.raises(RuntimeError):
        QuantLinear(k, n, bits, g, 1).load_state_dict(bad)


def test_fused_rows_limit_is_shape_aware():
    """layers the multi-row builds do not take (odd group sizes, group 32 at 2 bits) leave the fused path at 32 rows
    instead of 128; general act-order layers whose group table does not fit the by-id builds leave it at 4."""
    from xbitops_amd import linear

    assert linear.fused_rows_limit(4096, 128, 4) == linear.GEMV_MAX_ROWS == 512
    assert linear.fused_rows_limit(4096, 32, 4) == 512 and linear.fused_rows_limit(4096, 64, 2) == 512 and linear.fused_rows_limit(11008, 256, 8) == 512
    assert linear.fused_rows_limit(4096, 48, 4) == 32      # odd group size: by-id builds, 32 rows per launch
    assert linear.fused_rows_limit(7167, 128, 4) == 512    # ragged K: the same builds, masked tail (round 4)
    assert linear.fused_rows_limit(4096, 32, 2) == 32      # 2-bit needs groups of 64 and more
    assert linear.fused_rows_limit(4096, 96, 4) == 32      # 32 x odd
    assert linear.GIDX_GEMV_MAX_ROWS_OTHER < linear.GIDX_GEMV_MAX_ROWS
