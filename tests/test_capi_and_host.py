"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/xbitops_hip.h declares, argument validation (no compute without a GPU), and the Python
operator surface mirrors the reference binding's checks (dq_torch_ops.cc:5-9,25-31,49-57)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "xbitops_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xbit_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_two_launcher_replacements():
    syms = declared_symbols()
    assert "xbit_dequant_f16" in syms and "xbit_gemv_f16" in syms


def test_library_exports_every_declared_symbol(native_lib):
    from xbitops_amd import _lib

    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), name
    assert sorted(_lib.EXPORTS) == declared_symbols()
    assert native_lib.xbit_abi_version() == 2
    # only the declared C ABI is visible: no C++ internals, no process-global tuning setters
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    assert exported == declared_symbols(), exported


def test_library_has_no_torch_dependency(native_lib):
    from xbitops_amd import _lib
    import subprocess

    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    names = [line.split()[0] for line in out.splitlines() if line.strip()]  # names only: the load addresses are random hex
    assert not any("torch" in n or "c10" in n for n in names), names
    assert any("amdhip64" in n for n in names)


def test_status_strings(native_lib):
    from xbitops_amd import _lib

    assert _lib.status_string(0) == "success"
    for code in (-1, -2, -3, -4, -5, -6):
        assert _lib.status_string(code).startswith("xbitops:")
    with pytest.raises(RuntimeError, match="bits must be in 2..8"):
        _lib.check(-2, "x")


def test_capi_rejects_bad_arguments_before_touching_the_gpu(native_lib):
    lib = native_lib
    p = 0x1000  # never dereferenced: every call below must fail validation first
    assert lib.xbit_gemv_f16(None, p, p, p, p, 4, 128, 1, 128, 64, 0, None) == -1
    assert lib.xbit_gemv_f16(p, p, p, p, p, 1, 128, 1, 128, 64, 0, None) == -2
    assert lib.xbit_gemv_f16(p, p, p, p, p, 9, 128, 1, 128, 64, 0, None) == -2
    assert lib.xbit_gemv_f16(p, p, p, p, p, 4, 0, 1, 128, 64, 0, None) == -3
    assert lib.xbit_gemv_f16(p, p, p, p, p, 4, 128, 0, 128, 64, 0, None) == -4
    assert lib.xbit_gemv_f16(p, p, p, p, p, 4, 128, 1, 128, 63, 0, None) == -4  # odd n
    assert lib.xbit_gemv_f16(p, p, p + 2, p, p, 4, 128, 1, 128, 64, 0, None) == -5  # qweight not 4-byte aligned
    assert lib.xbit_dequant_f16(p, None, p, p, 4, 128, 128, 64, 0, None) == -1
    assert lib.xbit_dequant_f16(p, p, p, p, 0, 128, 128, 64, 0, None) == -2
    assert lib.xbit_dequant_f16(p, p, p, p, 4, 128, 0, 64, 0, None) == -4
    assert lib.xbit_profile_last_ms(None) == -1
    from xbitops_amd._lib import GemvOptions

    assert lib.xbit_gemv_f16_ex(p, p, p, p, p, 4, 128, 1, 128, 64, 0, None, GemvOptions(split_k=16)) == -4   # split factor out of range
    assert lib.xbit_gemv_f16_ex(p, p, p, p, p, 4, 128, 1, 128, 64, 0, None, GemvOptions(variant=64)) == -6   # production build: no variants
    assert lib.xbit_gemv_f16_ex(p, p, p, p, p, 4, 128, 1, 128, 64, 0, None, GemvOptions(workspace=p + 4, workspace_bytes=64)) == -5
    assert lib.xbit_gemv_f16_ex(None, p, p, p, p, 4, 128, 1, 128, 64, 0, None, None) == -1
    assert lib.xbit_gemv_workspace_bytes(1, 3584) == 8 * 3584 and lib.xbit_gemv_workspace_bytes(3, 130) == 8 * 4 * 192


def _cpu_args(k=128, n=64, bits=4, g=128):
    qweight = torch.zeros(((k * bits + 31) // 32, n), dtype=torch.int32)
    scales = torch.ones(((k + g - 1) // g, n), dtype=torch.float16)
    qzeros = torch.zeros(((k + g - 1) // g, (n * bits + 31) // 32), dtype=torch.int32)
    return qweight, scales, qzeros


def test_ops_reject_cpu_tensors_like_check_cuda(native_lib):
    import XbitOps

    qweight, scales, qzeros = _cpu_args()
    with pytest.raises(RuntimeError, match="qweight must be a CUDA tensor"):
        XbitOps.dequant(qweight, scales, qzeros, 128, 4, 128, 0)
    with pytest.raises(RuntimeError, match="input_a must be a CUDA tensor"):
        XbitOps.gemv(torch.zeros(1, 128, dtype=torch.float16), qweight, scales, qzeros, 128, 4, 128, 0)


def test_module_surface_matches_reference_registration():
    import inspect

    import XbitOps

    assert sorted(XbitOps.__all__) == ["dequant", "gemv"]
    assert list(inspect.signature(XbitOps.dequant).parameters) == ["qweight", "scales", "qzeros", "groupsize", "bits", "in_features", "add_zero_bias"]
    assert list(inspect.signature(XbitOps.gemv).parameters) == ["input_a", "qweight", "scales", "qzeros", "groupsize", "bits", "in_features", "add_zero_bias"]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; the shipped path must not reach it."""
    for pkg in ("xbitops_amd", "XbitOps"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, pkg)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, fn)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, fn)
                    assert "xbit_oracle" not in text and "libxbit_ref" not in text, os.path.join(dirpath, fn)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from xbitops_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.XbitLibraryError, match="no CPU fallback"):
        _lib.load()


def test_public_header_is_plain_c():
    """include/xbitops_hip.h is the drop-in boundary: it must compile as C99 and as C++ on its own."""
    import subprocess

    hdr = os.path.join(ROOT, "include", "xbitops_hip.h")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], ["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_split_plan_occupancy_table(native_lib):
    """The split-K planner prices a plan by how many of its workgroups fit on the chip at once, from the VGPR
    counts of the split builds (kSplitVgprs in csrc/xbit_gemv_mfma.h).  The table must describe the library that
    ships: read the counts back from the code objects embedded in the .so."""
    import re

    from xbitops_amd import _lib, codeobj

    src = open(os.path.join(os.path.dirname(_lib.__file__), "csrc", "xbit_gemv_mfma.h")).read()
    body = re.search(r"kSplitVgprs\[7\]\[4\] = \{(.*?)\};", src, re.S).group(1)
    table = [[int(x) for x in row.split(",")] for row in re.findall(r"\{([^{}]*)\}", body)]
    assert len(table) == 7 and all(len(r) == 4 for r in table)
    seen = {}
    for name, r in codeobj.kernel_resources(_lib.LIB_PATH).items():
        a = codeobj.gemv_template_args(name)
        if a and a["split"] and a["nl"] == 16 and a["ub"] == (4 if a["bits"] <= 4 else 2):  # the builds the table describes
            seen[(a["bits"], a["cpl"], a["mt"])] = r
    for bits in range(2, 9):
        for col, (cpl, mt) in enumerate(((4, 1), (4, 4), (2, 1), (2, 4))):
            r = seen[(bits, cpl, mt)]
            assert r["spill"] == 0, (bits, cpl, mt, r)
            # the planner only needs the allocation class (512 // round_up(vgpr, 8) waves per SIMD) to be right
            assert 512 // ((r["vgpr"] + 7) & ~7) == 512 // ((table[bits - 2][col] + 7) & ~7), (bits, cpl, mt, r["vgpr"], table[bits - 2][col])


def test_no_kernel_in_the_library_spills(native_lib):
    """Every kernel of the production library keeps its registers: the launchers do not compile the 16-wave
    builds that would exceed 128 VGPRs (fits_16_waves in csrc/xbit_gemv_mfma.h) - measured up to 40 % slower
    than 8 waves over two sweeps (profiles/r02/sweep_16_vs_8_waves.txt)."""
    from xbitops_amd import _lib, codeobj

    res = codeobj.kernel_resources(_lib.LIB_PATH)
    assert len(res) > 100
    bad = {n: r for n, r in res.items() if r["spill"] or r["scratch"]}
    assert not bad, bad
