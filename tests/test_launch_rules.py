"""Which kernel a call would launch - read back through the library's dry run (xbit_dry_run_begin / _end: the entry points run
their checks and launch rules, the launches are recorded instead of issued).  No GPU: device pointers are never followed.

These tests pin the launch RULES to shapes (the BASELINE configs, the shapes ADVICE / VERDICT r03 named, one shape per rule);
tools/check_rules.py checks on the GPU that the rules still pick the fastest launch."""
import ctypes

import pytest

from xbitops_amd import _lib

A, QW, SC, QZ, OUT, GI = 0x10000000, 0x20000000, 0x30000000, 0x40000000, 0x50000000, 0x60000000  # never followed


def plan(lib, bits, m, k, n, g=128, opts=None):
    def call():
        if opts is None:
            st = lib.xbit_gemv_f16(OUT, A, QW, SC, QZ, bits, g, m, k, n, 1, None)
        else:
            st = lib.xbit_gemv_f16_ex(OUT, A, QW, SC, QZ, bits, g, m, k, n, 1, None, opts)
        _lib.check(st, "xbit_gemv_f16")

    return _lib.dry_run(lib, call)


def test_dry_run_records_instead_of_launching(native_lib):
    recs = plan(native_lib, 4, 1, 4096, 11008)
    assert len(recs) == 1
    r = recs[0]
    assert r["kernel"].startswith("void xbit::dyn::gemv_dyn_kernel<4, 1, 3, 16, 8, true, 2, 4, 4, 0, false, false, 1, false>"), r  # the metric kernel: 48-column tiles
    assert r["grid"] == (230, 1, 1) and r["block"] == (512, 1, 1)
    # outside a dry run nothing is recorded, and ending one that was never begun is an error
    cnt = ctypes.c_int()
    assert native_lib.xbit_dry_run_end((_lib.LaunchRecord * 1)(), 1, ctypes.byref(cnt)) == -4


def test_baseline_configs_take_the_staged_operand_kernel(native_lib):
    for bits in range(2, 9):  # cfg3: 4096 x 4096
        (r,) = plan(native_lib, bits, 1, 4096, 4096)
        assert ("gemv_dyn_kernel<%d," % bits in r["kernel"]) == (bits >= 3), (bits, r)  # 2-bit few-column layers stay on the round-2 kernel (4.1 against 4.5 us)
    for m in (1, 4, 16):  # cfg5
        (r,) = plan(native_lib, 3, m, 4096, 11008)
        assert "gemv_dyn_kernel<3," in r["kernel"] or m == 4, (m, r)


def gidx_plan(lib, bits, m, k, n, groups):
    def call():
        _lib.check(lib.xbit_gemv_gidx_f16(OUT, A, QW, SC, QZ, GI, bits, groups, m, k, n, 1, None), "xbit_gemv_gidx_f16")

    return _lib.dry_run(lib, call)


@pytest.mark.parametrize("m", [1, 4, 17, 20, 36])
def test_act_order_down_projection_stays_on_the_fast_kernel(native_lib, m):
    """K = 11008 is beyond the 1- / 4-row by-id builds' staging rounds; those launches (and the 1..4-row tail
    of a longer batch) now take the 8-row build, and no row is ever computed twice."""
    recs = gidx_plan(native_lib, 4, m, 11008, 4096, 86)
    assert len(recs) in ((m + 31) // 32, (m + 15) // 16), recs  # rows in launches of 32, or of 16 where the 32-row build's LDS image does not fit (86 groups here)
    for r in recs:
        assert "gemv_dyn_kernel<4," in r["kernel"] and ", true, " in r["kernel"].split("gemv_dyn_kernel<")[1], r  # GIDX builds


def test_workspace_bytes_cover_the_build_that_takes_the_rows(native_lib):
    # 9..12 rows run the 16-row split-K build: 16 rows of accumulators
    assert native_lib.xbit_gemv_workspace_bytes(10, 4096) == 16 * 4096 * 8
    assert native_lib.xbit_gemv_workspace_bytes(5, 4096) == 8 * 4096 * 8
    assert native_lib.xbit_gemv_workspace_bytes(3, 4000) == 4 * 4032 * 8
    assert native_lib.xbit_gemv_workspace_bytes(1, 64) == 64 * 8


def test_fused_launch_is_one_launch_of_the_staged_operand_kernel(native_lib):
    vp = ctypes.c_void_p
    for m in (1, 4, 8, 16):
        def call():
            ns = (4096, 1024, 1024)
            st = native_lib.xbit_gemv_multi_f16(3, (vp * 3)(OUT, OUT + (1 << 24), OUT + (2 << 24)), A, (vp * 3)(QW, QW + (1 << 26), QW + (2 << 26)),
                                                (vp * 3)(SC, SC + (1 << 24), SC + (2 << 24)), (vp * 3)(QZ, QZ + (1 << 24), QZ + (2 << 24)),
                                                (ctypes.c_uint32 * 3)(*ns), 4, 128, m, 4096, 1, None)
            _lib.check(st, "xbit_gemv_multi_f16")

        recs = _lib.dry_run(native_lib, call)
        assert len(recs) == 1 and "gemv_dyn_kernel<4," in recs[0]["kernel"], recs
        assert recs[0]["kernel"].split("gemv_dyn_kernel<")[1].split(">")[0].split(", ")[-1] == "true", recs  # MULTI build


def test_ragged_k_and_odd_groups_with_several_rows_take_the_fast_kernel(native_lib):
    """VERDICT r03 'cliffs': a row length that is no multiple of 8 at 2..32 rows, and 17..32 rows of an odd group size, ran the
    column kernel (85 / 93 us); they are builds of the staged-operand kernel now."""
    for k in (7166, 7167):
        for m in (2, 4, 8, 16, 24):
            (r,) = plan(native_lib, 4, m, k, 4096)
            assert "gemv_dyn_kernel<4," in r["kernel"], (k, m, r)
    (r,) = plan(native_lib, 4, 24, 4096, 4096, g=48)
    assert "gemv_dyn_kernel<4, 1, 3, 16, 8, true, 2, 4, 4, 0, false, true, 32" in r["kernel"], r  # the 32-row by-id build
    recs = plan(native_lib, 4, 40, 4096, 11008, g=80)
    assert len(recs) == 2 and all("gemv_dyn_kernel<4," in r["kernel"] for r in recs), recs


def test_launch_rules_grid_is_as_recorded(native_lib):
    """tests/golden/launch_rules.txt: what the production library launches for ~9 K (bits, rows, K, N, group) points.  A change to
    xbit_gemv_rules.h (or to a kernel's eligibility) shows up here as a diff to review; re-record with tools/launch_rules_grid.py
    next to the tools/check_rules.py run that justifies it."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "launch_rules_grid.py"), "--check", os.path.join(root, "tests", "golden", "launch_rules.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_thirty_three_to_five_twelve_rows_are_one_launch(native_lib):
    """33..512 rows: ONE launch over blocks of 32 / 64 rows - of the kernel whose column groups share the activation image
    (xbit_gemv_rows.h; columns per group and groups per workgroup by grid fill), or launches of the staged-operand kernel's 32-row build where that one
    does not take the layer (group sizes it cannot index by k-slot)."""
    # grid: whole runs of column tiles per XCD (8 x ceil(tiles of the workgroup's width / 8)) x blocks of 32 rows, mapped to (tile, block) in the kernel
    # (up to 64 rows: 32 rows per workgroup; beyond: 64 where the model says so - here 2 x 48 columns x 64 rows, 115 column tiles in runs of 15 per XCD)
    for m, shape, grid in ((33, "3, 2, 8, 32", (120 * 2, 1, 1)), (64, "3, 2, 8, 32", (120 * 2, 1, 1)), (65, "3, 2, 8, 64", (120 * 2, 1, 1)), (128, "3, 2, 8, 64", (120 * 2, 1, 1))):
        (r,) = plan(native_lib, 4, m, 4096, 11008)
        assert "gemv_rows_kernel<4, %s, false>" % shape in r["kernel"] and r["grid"] == grid and r["block"] == (512, 1, 1), (m, r)
    (r,) = plan(native_lib, 4, 256, 4096, 11008)  # up to 256 rows (dequantise + dense GEMM takes over beyond: xbitops_amd/linear.py)
    assert "gemv_rows_kernel<4, 3, 4, 8, 64, false>" in r["kernel"] and r["grid"] == (64 * 4, 1, 1), r
    (r,) = plan(native_lib, 4, 512, 4096, 11008)  # ... and up to 512 rows (kGemvRowsMax); beyond: 16-row passes of the round-2 kernel
    assert "gemv_rows_kernel<4, 3, 4, 8, 64, false>" in r["kernel"] and r["grid"] == (64 * 8, 1, 1), r
    assert "gemv_mfma_kernel<4," in plan(native_lib, 4, 513, 4096, 11008)[0]["kernel"]
    (r,) = plan(native_lib, 4, 64, 4096, 4096)  # few tiles, up to 128 rows: the 32-row build on the layer's default tiles (here 32 columns: 256 workgroups)
    assert "gemv_dyn_kernel<4, 1, 2, 16, 8, true, 2, 4, 4, 0, false, false, 32, false>" in r["kernel"] and r["grid"] == (128, 2, 1), r
    (r,) = plan(native_lib, 4, 192, 4096, 4096)
    assert "gemv_rows_kernel<4," in r["kernel"] and r["grid"][0] % 8 == 0 and r["grid"][1] == 1, r
    recs = plan(native_lib, 4, 64, 4096, 11008, g=48)  # groups that begin inside a k-block: two launches of the 32-row by-id build
    assert len(recs) == 2 and all("gemv_dyn_kernel<4," in r["kernel"] for r in recs), recs
    (r,) = plan(native_lib, 2, 64, 4096, 11008)  # 2-bit: the same kernel on units of 256 k (32 rows per workgroup only)
    assert "gemv_rows_kernel<2, " in r["kernel"] and r["kernel"].split(">")[0].endswith(", 8, 32, false"), r
    recs = plan(native_lib, 2, 64, 4096, 11008, g=32)  # ... groups of 64 and more; below: 16-row passes of the round-2 kernel
    assert "gemv_mfma_kernel<2," in recs[0]["kernel"], recs


def test_long_k_layers_take_the_windowed_table_build(native_lib):
    """The tables of ALL groups of a long K do not fit in LDS beside the activation images (K beyond about 14 K at group 128 with 32 rows per workgroup,
    about 4.6 K with 64): such layers used to fall to the 32-row build in grid.y, which re-reads the activations per 48-64 columns.  They now run the builds of
    the shared-activation-tile kernel that keep the tables as a ring of two windows (last template argument true; 16 KB of tables whatever K)."""
    for m in (40, 64, 128, 256, 512):
        (r,) = plan(native_lib, 4, m, 28672, 8192)  # the 70B down-projection
        assert "gemv_rows_kernel<4," in r["kernel"] and r["kernel"].split("gemv_rows_kernel<")[1].split(">")[0].endswith(", true"), (m, r)
        assert r["lds"] <= 156 * 1024, r
    (r,) = plan(native_lib, 4, 128, 16384, 6144, g=64)  # 256 groups
    assert "gemv_rows_kernel<4," in r["kernel"] and ", true>" in r["kernel"], r
    (r,) = plan(native_lib, 4, 128, 4096, 11008)  # short K: the tables of all groups, as before
    assert ", false>" in r["kernel"], r
