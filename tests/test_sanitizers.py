"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer ("-fsanitize=address,undefined on the
CPU restatement").  CPU only - sanitizers are never run on the GPU.

* the oracle's C restatement (oracle/xbit_oracle.c) built with both sanitizers; the pinning tests and the harness-format
  tests then run in a child interpreter that loads that build (the sanitizer runtime is pre-loaded into it);
* the harness's host logic (xbitops_amd/csrc/xbit_harness_check.h: fp16 decoding, the reference's acceptance rules incl. the
  NaN cases, shape derivation) through a small driver built with both sanitizers.
* the host half of the C ABI (xbitops_amd/csrc/xbit_capi.hip: argument checks, options parsing, the split-K workspace pool's
  bookkeeping, profiling series, dry run) compiled as plain C++ against a mock HIP runtime and recording launchers
  (tests/sanitize/capi_host_driver.cc), driven through bad arguments, capture / release / exhaustion / failure injection.
Any report from either sanitizer aborts the child (halt_on_error, -fno-sanitize-recover) and fails the test.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_FLAGS = ["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or _runtime("libasan.so") is None, reason="gcc with libasan is needed")


def test_harness_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "harness_check_driver")
    src = os.path.join(ROOT, "tests", "sanitize", "harness_check_driver.cc")
    r = subprocess.run(["g++", "-std=c++17"] + SAN_FLAGS + [src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout + r.stderr


def test_oracle_restatement_under_asan_ubsan(tmp_path):
    lib = str(tmp_path / "libxbit_oracle_san.so")
    src = os.path.join(ROOT, "oracle", "xbit_oracle.c")
    r = subprocess.run(["gcc"] + SAN_FLAGS + ["-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", lib, src, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, XBIT_ORACLE_LIB=lib, LD_PRELOAD=_runtime("libasan.so"), OMP_NUM_THREADS="4",
               ASAN_OPTIONS="halt_on_error=1:detect_leaks=0:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the child must really be running the sanitized build
    chk = subprocess.run([sys.executable, "-c", "import oracle; oracle.lib(); print(open('/proc/self/maps').read().count('libxbit_oracle_san.so') > 0)"],
                         capture_output=True, text=True, env=env, cwd=ROOT)
    assert chk.returncode == 0 and chk.stdout.strip() == "True", chk.stdout + chk.stderr
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "tests/test_oracle_pinning.py",
                        "tests/test_harness_format.py::test_round_trip", "tests/test_harness_format.py::test_reference_acceptance_rules"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_capi_host_half_under_asan_ubsan(tmp_path):
    """xbit_capi.hip holds no kernel: with the HIP runtime mocked (host memory, injectable failures, settable capture status) and the
    other translation units' launchers replaced by recorders, its argument validation and workspace-pool bookkeeping run under both
    sanitizers on the CPU - the code ADVICE r02 found four bugs in."""
    inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers are needed (declarations only)")
    exe = str(tmp_path / "capi_host_driver")
    cmd = ["g++", "-std=c++17"] + SAN_FLAGS + ["-D__HIP_PLATFORM_AMD__", "-I", inc, "-include", os.path.join(ROOT, "tests", "sanitize", "hip_host_shim.h"), "-w", "-x", "c++",
           os.path.join(ROOT, "xbitops_amd", "csrc", "xbit_capi.hip"), os.path.join(ROOT, "tests", "sanitize", "capi_host_driver.cc"), "-o", exe, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert r.returncode == 0 and "all expectations hold" in r.stdout, r.stdout + r.stderr[-3000:]
