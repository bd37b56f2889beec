"""The staged-operand one-row GEMV (xbitops_amd/csrc/xbit_gemv_dyn.h) against the oracle: every bit-width, every tile
shape, every count of staging rounds, group sizes from one k-block up, edge tiles, the spliced-form fallback - through
the production entry point (which picks the kernel for these shapes) and, tile shape by tile shape, through the tuning
library's forced variants.  GPU only."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import contract
from xbitops_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.ascontiguousarray(x).view(np.uint16)


def _close(y, y_ref, what, k=None):
    assert k is not None
    contract.assert_gemv_close(y, y_ref, k, what)  # oracle/contract.py: the one GEMV contract


def _run(lib, p, bias, dev, variant=-1):
    a, qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (p.act, p.qweight, p.scales, p.qzeros))
    out = torch.full((1, p.n), float("nan"), dtype=torch.float16, device=dev)
    st = lib.xbit_gemv_f16_ex(out.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), p.bits, p.groupsize, 1, p.k, p.n, bias,
                              torch.cuda.current_stream().cuda_stream, _lib.GemvOptions(variant=variant))
    _lib.check(st, "xbit_gemv_f16_ex")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _ref(p, bias):
    return oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, p.groupsize, p.bits, p.k, bias, oracle.RNE)


# (k, n, groupsize): tiles of 32 / 48 / 64 / 16 columns by the launcher's rule, 2 / 4 / 8 staging rounds, edge tiles, long K
SHAPES = [(4096, 11008, 128), (4096, 4096, 128), (4096, 8192, 128), (4096, 512, 128), (8192, 7168, 128), (11008, 5504, 128), (2048, 8200, 64),
          (4096, 6150, 256), (1024, 8192, 64), (8192, 1536, 128), (256, 64, 64), (4096, 11008, 64)]


@pytest.mark.parametrize("bits", range(2, 9))
def test_production_one_row_shapes_vs_oracle(bits, cuda_device):
    lib = _lib.load()
    for k, n, g in SHAPES:
        p = synth.make_problem(k, n, bits, g, m=1, seed=bits * 977 + k + n)
        _close(_run(lib, p, 1, cuda_device), _ref(p, 1), (bits, k, n, g), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [301, 302, 303, 304, 311, 313])
def test_forced_tile_shapes_equal_the_oracle_and_each_other(bits, variant, cuda_device):
    """Tuning library: 16-column tiles (8 lanes), 32 / 48 / 64-column tiles, nontemporal and plain row loads."""
    lib = _lib.load_tuning()
    for k, n, g in [(4096, 1030, 128), (2048, 4096, 32 if bits != 2 else 64), (8192, 344, 128)]:
        p = synth.make_problem(k, n, bits, g, m=1, seed=bits * 31 + variant)
        y = _run(lib, p, 0, cuda_device, variant)
        _close(y, _ref(p, 0), (bits, variant, k, n, g), k=k)
        assert np.array_equal(_bits(y), _bits(_run(lib, p, 0, cuda_device, variant)))  # launch to launch: identical


@pytest.mark.parametrize("bits", range(2, 9))
def test_large_scale_takes_the_spliced_form_with_the_same_w(bits, cuda_device):
    """One scale too large for the subnormal form (and one infinite, one NaN in another tile) switch their workgroups to the
    spliced form: the columns of unaffected scales still equal the oracle bit for bit, the affected ones as the oracle says."""
    lib = _lib.load()
    k, n, g = 4096, 2048, 128
    p = synth.make_problem(k, n, bits, g, m=1, seed=bits + 500)
    sc = p.scales.copy()
    sc[3, 5] = np.float16(30.0)      # s * 2^(14 + bits) overflows fp16
    sc[7, 700] = np.float16(np.inf)
    sc[9, 1500] = np.float16(np.nan)
    p2 = synth.Problem(k, n, bits, g, 1, p.qweight, sc, p.qzeros, p.act)
    y, y_ref = _run(lib, p2, 1, cuda_device), _ref(p2, 1)
    finite = np.isfinite(y_ref.astype(np.float32))
    assert np.array_equal(np.isnan(y.astype(np.float32)), np.isnan(y_ref.astype(np.float32)))
    _close(y[finite], y_ref[finite], bits, k=k)


def test_dyn_kernel_is_what_serves_the_metric_shape(cuda_device):
    """rocprof names the kernel; here: the forced variant (tuning library) and the production launch agree bit for bit on
    the metric shape while the round-2 kernel (variant 399) differs in summation order somewhere."""
    lib, tun = _lib.load(), _lib.load_tuning()
    p = synth.make_problem(4096, 11008, 4, 128, m=1, seed=77)
    y_prod, y_dyn, y_old = _run(lib, p, 1, cuda_device), _run(tun, p, 1, cuda_device, 300), _run(tun, p, 1, cuda_device, 399)
    assert np.array_equal(_bits(y_prod), _bits(y_dyn))
    _close(y_old, _ref(p, 1), "round-2 kernel", k=p.k)
    assert not np.array_equal(_bits(y_old), _bits(y_dyn))


# the reference harness's own ragged K (main.cc:21 forces MATRIX_K = 7167) and friends: any k, one row -> still the fast kernel
@pytest.mark.parametrize("bits", range(2, 9))
def test_ragged_k_one_row(bits, cuda_device):
    lib = _lib.load()
    for k, n, g in [(7167, 4096, 128), (4097, 1024, 128), (1001, 520, 64), (63, 64, 64), (4100, 11008, 128)]:
        p = synth.make_problem(k, n, bits, g, m=1, seed=bits * 7 + k)
        _close(_run(lib, p, 1, cuda_device), _ref(p, 1), (bits, k, n, g), k=k)


def _run_rows(lib, p, bias, dev):
    a, qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (p.act, p.qweight, p.scales, p.qzeros))
    out = torch.full((p.m, p.n), float("nan"), dtype=torch.float16, device=dev)
    st = lib.xbit_gemv_f16(out.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), p.bits, p.groupsize, p.m, p.k, p.n, bias,
                           torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "xbit_gemv_f16")
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("bits", range(2, 9))
def test_odd_group_sizes_one_to_sixteen_rows(bits, cuda_device):
    """Group sizes that are not a multiple of 32 (16, 48, 80, 1000, 17: a group may begin anywhere inside a k-block), one
    to sixteen rows: the staged-operand kernel's by-id builds with the ids computed as k / groupsize (ragged K included at one
    row; K = 4096 at group 16 is the largest table it takes: 256 groups)."""
    lib = _lib.load()
    for k, n, g, m in [(4096, 1024, 48, 1), (2048, 130, 16, 1), (7167, 512, 80, 1), (1000, 66, 1000, 3), (4096, 96, 16, 4), (96, 2, 16, 2),
                       (1024, 4100, 17, 1), (8192, 48, 48, 2), (2048, 256, 48, 9), (1000, 66, 1000, 6), (4096, 130, 80, 16)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 13 + k + g)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_column_kernel_shapes(bits, cuda_device):
    """What no fast kernel takes - more than sixteen rows of an odd group size, ragged K with several rows, more groups than
    the by-id table holds - runs the coalesced column kernel: exact W, fp32 sums."""
    lib = _lib.load()
    for k, n, g, m in [(333, 64, 17, 5), (7167, 1024, 128, 4), (2048, 256, 48, 19), (1000, 66, 1000, 33), (8192, 130, 16, 1), (1001, 70, 80, 2)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 13 + k + g)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)


def test_column_kernel_takes_misaligned_tensors(cuda_device):
    """Tensors aligned to their element size only (views at odd element offsets): still served, still exact."""
    lib = _lib.load()
    k, n, bits, g = 1024, 256, 4, 128
    p = synth.make_problem(k, n, bits, g, m=2, seed=5)
    dev = cuda_device

    def shifted(x, elems):  # the same values at an address that is `elems` elements past an aligned one
        buf = torch.empty(x.numel() + elems + 64, dtype=x.dtype, device=dev)
        view = buf[elems:elems + x.numel()].view(x.shape)
        view.copy_(x)
        return buf, view

    keep = []
    a = shifted(torch.from_numpy(p.act).to(dev), 1)       # 2-byte aligned
    qw = shifted(torch.from_numpy(p.qweight).to(dev), 1)  # 4-byte aligned, not 16
    sc = shifted(torch.from_numpy(p.scales).to(dev), 3)   # 2-byte aligned
    qz = shifted(torch.from_numpy(p.qzeros).to(dev), 1)
    keep += [a, qw, sc, qz]
    out = torch.empty((2, n), dtype=torch.float16, device=dev)
    st = lib.xbit_gemv_f16(out.data_ptr(), a[1].data_ptr(), qw[1].data_ptr(), sc[1].data_ptr(), qz[1].data_ptr(), bits, g, 2, k, n, 1,
                           torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "xbit_gemv_f16")
    torch.cuda.synchronize()
    _close(out.cpu().numpy(), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), "misaligned", k=k)


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [330, 332, 334])
def test_split_k_build_vs_oracle(bits, variant, cuda_device):
    """Few column tiles, long K: K cut into ranges that meet in the fixed-point accumulators (tuning library: the planned
    split with the planned / 32-column / 64-column tiles through the staged-operand kernel's split build)."""
    lib = _lib.load_tuning()
    for k, n, g in [(8192, 3584, 128), (11008, 4096, 128), (16384, 1024, 128), (8192, 1030, 64)]:
        p = synth.make_problem(k, n, bits, g, m=1, seed=bits * 17 + variant + k)
        y = _run(lib, p, 1, cuda_device, variant)
        _close(y, _ref(p, 1), (bits, variant, k, n, g), k=k)
        assert np.array_equal(_bits(y), _bits(_run(lib, p, 1, cuda_device, variant)))  # arrival order does not matter: integer sums


@pytest.mark.parametrize("bits", range(2, 9))
def test_fill_once_split_plans_production(bits, cuda_device):
    """What the production library's split rule launches on few-column long-K layers: one round of 64-column tiles (an edge
    tile included), 32-column tiles on a narrow layer, two rounds where one would leave the chip a fifth idle, a layer of 8192
    columns at K >= 12288, K ranges too long for the split build (the same plan on the round-2 split kernel), group sizes 64
    and 256; against the oracle, and bit-stable across launches."""
    lib = _lib.load()
    for k, n, g in [(8192, 1030, 128), (12288, 6150, 128), (16384, 520, 128), (28672, 3072, 128), (12288, 8192, 256), (11008, 4096, 64),
                    (10240, 2560, 128)]:
        p = synth.make_problem(k, n, bits, g, m=1, seed=bits * 19 + k + n)
        y = _run_rows(lib, p, 1, cuda_device)
        _close(y, _ref(p, 1), (bits, k, n, g), k=k)
        assert np.array_equal(_bits(y), _bits(_run_rows(lib, p, 1, cuda_device)))


@pytest.mark.parametrize("bits", range(2, 9))
def test_split_k_with_several_rows(bits, cuda_device):
    """Split-K builds of the 4-, 8-, 16- and 32-row kernels (few-column long-K layers at small batch: down-projections); the
    fill-once plan on 2..32 rows (variant 336 = the production rule), against the oracle and bit-stable across launches."""
    lib = _lib.load_tuning()
    for k, n, g, m in [(11008, 4096, 128, 2), (8192, 3584, 128, 4), (11008, 4096, 128, 8), (16384, 1030, 128, 16), (8192, 2560, 64, 5), (12288, 6144, 128, 13),
                       (11008, 4096, 128, 32), (16384, 1030, 128, 17), (8192, 3584, 128, 24)]:  # 17..32 rows: the split build of the 32-row kernel (from 3 bits up)
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 23 + k + m)
        y = _run_m(lib, p, 1, cuda_device, 336)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, k, n, g, m), k=k)
        assert np.array_equal(_bits(y), _bits(_run_m(lib, p, 1, cuda_device, 336)))


def _run_m(lib, p, bias, dev, variant=-1):
    a, qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (p.act, p.qweight, p.scales, p.qzeros))
    out = torch.full((p.m, p.n), float("nan"), dtype=torch.float16, device=dev)
    st = lib.xbit_gemv_f16_ex(out.data_ptr(), a.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), p.bits, p.groupsize, p.m, p.k, p.n, bias,
                              torch.cuda.current_stream().cuda_stream, _lib.GemvOptions(variant=variant))
    _lib.check(st, "xbit_gemv_f16_ex")
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("bits", range(2, 9))
def test_five_to_sixteen_rows_production(bits, cuda_device):
    """What the production rule sends to the 8- and 16-row builds: layers with many tiles at 5..16 rows, few-column layers
    at 9..16 rows on 32-column tiles, a K whose scale table needs more than the two static staging rounds (8192: 64 groups);
    11008x4096 (long K on few tiles) stays with the round-2 pass and is checked all the same."""
    lib = _lib.load()
    for k, n, g, m in [(4096, 6144, 128, 5), (4096, 6144, 128, 16), (8192, 4096, 128, 16), (11008, 4096, 128, 16), (8192, 6144, 128, 9), (2048, 11008, 64, 8)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 3 + m + k)
        y = _run_rows(lib, p, 1, cuda_device)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [302, 303, 304])
def test_five_to_sixteen_rows_forced_tile_shapes(bits, variant, cuda_device):
    """The 8- and 16-row builds (every wave fetches and permutes the activations of its own next unit): every tile width,
    both staging-round counts, an edge tile, a K that ends inside a unit; each row equals the oracle."""
    lib = _lib.load_tuning()
    for k, n, g, m in [(4096, 1030, 128, 16), (2048, 2048, 64, 8), (8192, 344, 128, 5), (4104, 96, 128, 9), (1024, 64, 128, 12)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 5 + variant + m)
        y = _run_m(lib, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_seventeen_to_sixty_four_rows_production(bits, cuda_device):
    """17..32 rows: one launch of the 32-row build on layers with many tiles; 33..64: two such launches; a few-column layer
    (two 16-row passes of the round-2 kernel) beside them."""
    lib = _lib.load()
    for k, n, g, m in [(4096, 6144, 128, 17), (4096, 6144, 128, 32), (4096, 6144, 128, 33), (2048, 11008, 64, 64), (4096, 1024, 128, 40)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 3 + m + k)
        y = _run_rows(lib, p, 1, cuda_device)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [302, 303, 304])
def test_seventeen_to_thirty_two_rows_forced_tile_shapes(bits, variant, cuda_device):
    """The 32-row build (two A operands - rows 0-15 and 16-31 - per unpacked B fragment): every tile width, an edge tile, a K
    that ends inside a unit; each row equals the oracle."""
    lib = _lib.load_tuning()
    for k, n, g, m in [(4096, 1030, 128, 32), (2048, 2048, 64, 17), (8192, 344, 128, 24), (4104, 96, 128, 31)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 5 + variant + m)
        y = _run_m(lib, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [302, 303, 304])
def test_two_to_four_rows_forced_tile_shapes(bits, variant, cuda_device):
    """The 4-row build (rows 2..4 per launch): every tile width, two staging-round counts, an edge tile; each row equals the
    oracle and does not depend on the rows beside it."""
    lib = _lib.load_tuning()
    for k, n, g, m in [(4096, 1030, 128, 4), (2048, 2048, 64, 2), (8192, 344, 128, 3)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 5 + variant + m)
        y = _run_m(lib, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)
        p1 = synth.Problem(k, n, bits, g, 1, p.qweight, p.scales, p.qzeros, p.act[m - 1:m])
        y1 = _run_m(lib, p1, 1, cuda_device, variant)
        ulp = np.abs(_bits(y[m - 1]).astype(np.int32) - _bits(y1[0]).astype(np.int32))
        assert (ulp <= 1).all() or np.abs(y[m - 1].astype(np.float32) - y1[0].astype(np.float32)).max() <= contract.cancel_slack(k)


# ---- round 4: 128-column tiles (32 lanes x 16 B), tiles narrowed to fill the chip, fused launches ----------------------------
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("variant", [340, 341])
def test_128_column_tiles_forced(bits, variant, cuda_device):
    """Tuning library: the 128-column lane map of the staged-operand kernel (two 16-lane column sets sharing the MFMA's column
    slots), whole (340) and narrowed to the width that makes the tile count a multiple of the CU count (341: 28672 columns ->
    256 tiles of 112), one to four rows, edge tiles, 2 / 4 / 8 staging rounds."""
    lib = _lib.load_tuning()
    for k, n, g, m in [(4096, 28672, 128, 1), (8192, 3500, 128, 1), (2048, 7000, 64, 3), (1024, 130, 128, 4), (8192, 7168, 128, 2), (16384, 520, 128, 1)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 3 + variant + k)
        y = _run_m(lib, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)
        assert np.array_equal(_bits(y), _bits(_run_m(lib, p, 1, cuda_device, variant)))  # launch to launch: identical


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [342, 343])
def test_narrowed_tiles_forced(bits, variant, cuda_device):
    """Tuning library: 64- / 48-column lane maps with tiles narrowed to fill the chip (28672 columns: 512 tiles of 56)."""
    lib = _lib.load_tuning()
    for k, n, g, m in [(2048, 28672, 128, 1), (4096, 7000, 128, 1), (1024, 3000, 64, 4), (4096, 200, 128, 1)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 3 + variant + k)
        y = _run_m(lib, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_fused_launch_every_width_up_to_sixteen_rows(bits, cuda_device):
    """xbit_gemv_multi_f16 on the staged-operand kernel: q/k/v- and gate/up-style sets, mixed and ragged N, 1..16 rows - every
    set against the oracle and bit-equal to the same set launched alone at the same row count... where the same build serves it
    (checked against the oracle either way)."""
    from xbitops_amd import ops

    g = 128
    for k, ns, m in ((4096, (4096, 1024, 1024), 1), (2048, (5504, 5504), 1), (1024, (130, 66, 258, 34), 1), (8192, (2048, 512), 1),
                     (1024, (256, 512), 3), (2048, (1024, 1024, 512), 4), (1024, (256, 130), 6), (2048, (1100, 900), 8), (1024, (512, 512, 512), 13),
                     (4096, (2048, 384), 16)):
        sets, refs, act = [], [], None
        for j, n in enumerate(ns):
            p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 13 + n + 7 * j + m)
            act = p.act if act is None else act
            refs.append(oracle.gemv(act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE))
            sets.append(tuple(torch.from_numpy(x).to(cuda_device) for x in (p.qweight, p.scales, p.qzeros)))
        a = torch.from_numpy(act).to(cuda_device)
        ys = ops.gemv_multi(a, sets, g, bits, k, 1)
        torch.cuda.synchronize()
        for y, ref, n in zip(ys, refs, ns):
            assert y.shape == (m, n)
            _close(y.cpu().numpy(), ref, "fused bits=%d k=%d ns=%s n=%d m=%d" % (bits, k, ns, n, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_ragged_k_with_several_rows(bits, cuda_device):
    """A row length that is no multiple of 8 - even or odd - at 2..40 rows: the 4- / 8- / 16- / 32-row builds mask the piece that
    holds element k - 1 where they stage it (the reference harness's own K = 7167, main.cc:21, at more than one row)."""
    lib = _lib.load()
    for k, n, g, m in [(7166, 1024, 128, 4), (4098, 520, 128, 2), (1002, 130, 64, 8), (7166, 4096, 128, 16), (4100, 2048, 128, 24), (2046, 256, 128, 40),
                       # odd K: every other row begins on an odd halfword and is read unaligned
                       (7167, 4096, 128, 4), (7167, 512, 128, 16), (4097, 256, 128, 32), (1001, 130, 64, 3), (63, 64, 64, 2), (7167, 1024, 128, 37)]:
        if bits == 2 and g < 64:
            continue
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 11 + k + m)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_odd_group_sizes_seventeen_to_forty_rows(bits, cuda_device):
    """17..32 rows of a group size that is no multiple of 32: the 32-row by-id build; more rows: launches of 32."""
    lib = _lib.load()
    for k, n, g, m in [(2048, 256, 48, 19), (4096, 1024, 48, 24), (1000, 66, 1000, 33), (4096, 130, 80, 32), (2048, 512, 16, 40)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 13 + k + g)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_thirty_three_to_one_twenty_eight_rows(bits, cuda_device):
    """33..128 rows: one launch of the 32-row build, the blocks of 32 rows in grid.y (from 3 bits up; 2-bit: 16-row passes); ragged K and
    few-tile layers included."""
    lib = _lib.load()
    for k, n, g, m in [(4096, 11008, 128, 64), (4096, 11008, 128, 33), (2048, 9216, 64, 48), (8192, 12288, 128, 64), (4096, 10000, 128, 100), (1024, 9300, 128, 128),
                       (4098, 9216, 128, 40), (4097, 1024, 128, 57)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 29 + k + m)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", [3, 4, 5, 8])
def test_rows_in_grid_y(bits, cuda_device):
    """More than 32 rows as blocks of 32 rows in grid.y of ONE launch of the 32-row build (production: 33..128 rows; variant 348: any count up to 256),
    ragged K and a ragged last block included."""
    lib, tun = _lib.load(), _lib.load_tuning()
    for k, n, g, m, variant in [(4096, 11008, 128, 128, -1), (2048, 9216, 64, 65, -1), (4098, 9300, 128, 100, -1), (1024, 9216, 128, 40, 348), (4096, 3000, 128, 200, 348),
                                (4097, 9216, 128, 97, 348)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 31 + k + m)
        y = _run_m(tun if variant > 0 else lib, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, k, n, g, m, variant), k=k)


# ---- xbit_gemv_rows.h: 33..128 rows, column groups that share one LDS image of the activations ------------------------------------------------

@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [351, 352, 353, 354, 355, 356])
def test_shared_activation_tile_every_shape(bits, variant, cuda_device):
    """Every (columns per group, groups per workgroup, rows per workgroup: 351-354 32 rows, 355 / 356 64) build of the shared-activation-tile kernel: a last workgroup with idle column groups, an
    edge tile, a ragged last block of rows, K that ends inside a unit / inside an octet / on an odd element, group sizes 32..K, more groups in
    the table than the two k-slices' first units name.  (2-bit: a unit is 256 k and groups start at 64; what a forced shape does not take - group 32,
    64 rows per workgroup - falls back to the round-2 kernel and is checked all the same.)"""
    tun = _lib.load_tuning()
    for k, n, g, m in [(4096, 1030, 128, 64), (2048, 2048, 64, 33), (8192, 344, 128, 100), (4104, 96, 128, 47), (4097, 1024, 128, 57), (1000, 530, 32, 128),
                       (1027, 200, 1024, 40), (96, 4096, 32, 65)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 37 + variant + k)
        y = _run_m(tun, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)
        assert np.array_equal(_bits(y), _bits(_run_m(tun, p, 1, cuda_device, variant)))  # fixed summation order: launch to launch identical


@pytest.mark.parametrize("bits", range(2, 9))
def test_shared_activation_tile_is_what_production_launches(bits, cuda_device):
    """33..128 rows of a layer with many tiles: production and the forced launch of the chosen shape (variant 350) agree bit for bit; the
    32-row build in grid.y (348) agrees within the contract (another summation order); 256 rows through 350 as well."""
    lib, tun = _lib.load(), _lib.load_tuning()
    for k, n, g, m in [(4096, 11008, 128, 128), (4096, 12288, 128, 64), (2048, 9216, 128, 96)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 41 + m)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)
        assert np.array_equal(_bits(y), _bits(_run_m(tun, p, 1, cuda_device, 350)))
        _close(_run_m(tun, p, 1, cuda_device, 348), y_ref, (bits, k, n, g, m, 348), k=k)
    p = synth.make_problem(2048, 3000, bits, 128, m=256, seed=bits + 256)
    _close(_run_m(tun, p, 1, cuda_device, 350), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, 128, bits, 2048, 1, oracle.RNE), (bits, 256), k=2048)


@pytest.mark.parametrize("bits", range(2, 9))
def test_shared_activation_tile_large_scale_takes_the_spliced_form(bits, cuda_device):
    """A scale too large for the subnormal form (one infinite, one NaN elsewhere) switches ITS workgroups - all column groups of them - to the
    spliced form: unaffected columns equal the oracle, affected ones as the oracle says."""
    lib = _lib.load()
    k, n, g, m = 4096, 9216, 128, 48
    p = synth.make_problem(k, n, bits, g, m=m, seed=bits + 900)
    sc = p.scales.copy()
    sc[3, 5] = np.float16(30.0)
    sc[7, 700] = np.float16(np.inf)
    sc[31, 9215] = np.float16(np.nan)
    p2 = synth.Problem(k, n, bits, g, m, p.qweight, sc, p.qzeros, p.act)
    y, y_ref = _run_rows(lib, p2, 1, cuda_device), oracle.gemv(p2.act, p2.qweight, p2.scales, p2.qzeros, g, bits, k, 1, oracle.RNE)
    finite = np.isfinite(y_ref.astype(np.float32))
    assert np.array_equal(np.isnan(y.astype(np.float32)), np.isnan(y_ref.astype(np.float32)))
    _close(y[finite], y_ref[finite], bits, k=k)


@pytest.mark.parametrize("bits", range(2, 9))
def test_one_twenty_nine_to_five_twelve_rows(bits, cuda_device):
    """129..512 rows through the production entry point: one launch of the shared-activation-tile kernel from 3 bits up (ragged K, few tiles,
    group sizes 64 / 32 included), 16-row passes at 2 bits."""
    lib = _lib.load()
    for k, n, g, m in [(4096, 11008, 128, 256), (2048, 9216, 64, 129), (4097, 1024, 128, 200), (1024, 4096, 32, 192), (8192, 2048, 128, 160), (1024, 9216, 128, 512), (4097, 1024, 128, 384)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 43 + k + m)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)


@pytest.mark.parametrize("bits", range(2, 9))
@pytest.mark.parametrize("variant", [351, 353, 355])
def test_shared_activation_tile_tables_as_a_window(bits, variant, cuda_device):
    """The windowed-table builds (xbit_gemv_rows.h, WIN): layers whose tables of all groups do not fit in LDS beside the activation images - a long K at every
    group size the kernel indexes (a window is 1, 2, 4 ... steps), K that ends inside a window / a unit / on an odd element, a ragged last block of rows, an edge
    tile, more windows than two (the ring wraps), 64 rows per workgroup at a K that the full tables no longer allow.  Forced shapes: 2 column groups of 48 / 64
    columns x 32 rows, 2 x 48 x 64 rows; the dry run says the windowed build is what ran."""
    tun = _lib.load_tuning()
    shapes = [(28672, 520, 128, 40), (16384, 200, 32, 70), (20481, 96, 64, 70), (65536, 130, 256, 65)]  # (a ragged K goes by the production rule whatever the variant: beyond 64 rows this kernel)
    if variant == 355:
        shapes.append((8192, 344, 128, 100))  # 64 groups: too many beside the images of 64 rows only
    for k, n, g, m in shapes:
        if bits == 2 and g == 32:
            g = 64  # 2-bit units are 256 k: groups start at 64
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 53 + variant + k)
        if not (bits == 2 and variant == 355):  # 2-bit has no 64-row build (round-2 kernel instead, checked all the same)
            (r,) = _lib.dry_run(tun, lambda: _run_m(tun, p, 1, cuda_device, variant))
            assert "gemv_rows_kernel<%d," % bits in r["kernel"] and ", true>" in r["kernel"], (bits, variant, k, n, g, m, r)
        y = _run_m(tun, p, 1, cuda_device, variant)
        _close(y, oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE), (bits, variant, k, n, g, m), k=k)
        assert np.array_equal(_bits(y), _bits(_run_m(tun, p, 1, cuda_device, variant)))


@pytest.mark.parametrize("bits", range(2, 9))
def test_long_k_rows_production_and_spliced_form(bits, cuda_device):
    """Production entry point on a long-K layer at 48 and 130 rows (windowed tables), and the same layer with a scale too large for the subnormal form in a LATE
    group (found by the opening's pass over the tile's scales, not by the first window): affected columns as the oracle says, the others equal to it."""
    lib = _lib.load()
    k, n, g = 28672, 1030, 128
    for m in (48, 130):
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 59 + m)
        y, y_ref = _run_rows(lib, p, 1, cuda_device), oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, 1, oracle.RNE)
        _close(y, y_ref, (bits, k, n, g, m), k=k)
    sc = p.scales.copy()
    sc[200, 7] = np.float16(30.0)
    sc[223, 1029] = np.float16(np.inf)
    p2 = synth.Problem(k, n, bits, g, p.m, p.qweight, sc, p.qzeros, p.act)
    y, y_ref = _run_rows(lib, p2, 1, cuda_device), oracle.gemv(p2.act, p2.qweight, p2.scales, p2.qzeros, g, bits, k, 1, oracle.RNE)
    finite = np.isfinite(y_ref.astype(np.float32))
    assert np.array_equal(np.isnan(y.astype(np.float32)), np.isnan(y_ref.astype(np.float32)))
    _close(y[finite], y_ref[finite], bits, k=k)


@pytest.mark.parametrize("bits", [3, 4, 8])
def test_shared_activation_tile_operands_on_any_halfword(bits, cuda_device):
    """Activations and scales that begin on an odd halfword (views into larger buffers), 40 and 200 rows: the rows kernel reads them unaligned
    and gives the result of the aligned call bit for bit."""
    lib = _lib.load()
    for k, n, g, m in [(4096, 9216, 128, 40), (2050, 9216, 128, 200)]:
        p = synth.make_problem(k, n, bits, g, m=m, seed=bits * 47 + m)
        want = _run_rows(lib, p, 1, cuda_device)
        a_buf = torch.zeros(p.act.size + 1, dtype=torch.float16, device=cuda_device)
        s_buf = torch.zeros(p.scales.size + 1, dtype=torch.float16, device=cuda_device)
        a_buf[1:] = torch.from_numpy(p.act).to(cuda_device).reshape(-1)
        s_buf[1:] = torch.from_numpy(p.scales).to(cuda_device).reshape(-1)
        qw, qz = torch.from_numpy(p.qweight).to(cuda_device), torch.from_numpy(p.qzeros).to(cuda_device)
        out = torch.full((m, n), float("nan"), dtype=torch.float16, device=cuda_device)
        st = lib.xbit_gemv_f16(out.data_ptr(), a_buf.data_ptr() + 2, qw.data_ptr(), s_buf.data_ptr() + 2, qz.data_ptr(), bits, g, m, k, n, 1, torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "xbit_gemv_f16")
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(want)), (bits, k, n, m)
