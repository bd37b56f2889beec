"""Build libxbitops_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m xbitops_amd.build            # incremental: the production library
    python -m xbitops_amd.build --force    # rebuild everything
    python -m xbitops_amd.build --asm      # also keep .s files under xbitops_amd/csrc/_obj/
    python -m xbitops_amd.build --tuning   # also libxbitops_hip_tuning.so: the same sources with -DXBIT_TUNING=1,
                                           # i.e. plus the measurement variants (launch shapes, ablations, stamped
                                           # and VALU-only kernels) that tools/ and two A/B tests select through
                                           # xbit_gemv_f16_ex(options.variant); the production library has none

hipcc cross-compiles for gfx950 without a GPU present, so this runs in the build container; the
resulting .so travels to the GPU box with the working tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libxbitops_hip.so")
TUNING_LIB = os.path.join(LIB_DIR, "libxbitops_hip_tuning.so")
OBJ_TUNING = os.path.join(CSRC, "_obj_tuning")
SOURCES = (["xbit_capi.hip", "xbit_dequant.hip", "xbit_gidx.hip", "xbit_gemv.hip", "xbit_gemv_mfma.hip", "xbit_gemv_dyn.hip", "xbit_gemv_rows.hip"] +
           ["xbit_gemv_mfma_b%d.hip" % b for b in range(2, 9)] + ["xbit_gemv_dyn_b%d.hip" % b for b in range(2, 9)] + ["xbit_gemv_rows_b%d.hip" % b for b in range(2, 9)])
TUNING_ONLY_SOURCES = ["xbit_floor.hip"]  # measurement build only: pure-read floors for bench.py (include/xbitops_tuning.h)
HEADERS = ["xbit_device.h", "xbit_host.h", "xbit_unpack_gen.h", "xbit_gemv_mfma.h", "xbit_gemv_dyn.h", "xbit_gemv_rows.h", "xbit_harness_check.h", "xbit_launch_all.h", "xbit_gemv_rules.h", "xbit_gemv_tuning.h", os.path.join("..", "..", "include", "xbitops_hip.h"),
           os.path.join("..", "..", "include", "xbitops_tuning.h")]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
            "--offload-arch=" + ARCH]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the xbitops HIP library cannot be built on this machine")
    return exe


def _sources(tuning: bool):
    return SOURCES + (TUNING_ONLY_SOURCES if tuning else [])


def _newest_input(tuning: bool = False) -> float:
    paths = [os.path.join(CSRC, s) for s in _sources(tuning)] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return max(os.path.getmtime(p) for p in paths)


def is_current(tuning: bool = False) -> bool:
    lib = TUNING_LIB if tuning else LIB
    return os.path.exists(lib) and os.path.getmtime(lib) >= _newest_input(tuning)


def _object_is_current(obj: str) -> bool:
    """True when `obj` is newer than its source and every header the compiler saw (the .d file written next to it)."""
    dep = obj + ".d"
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return False
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    try:
        return bool(files) and all(os.path.getmtime(f) <= t for f in files if f.startswith(os.path.dirname(HERE)))  # this repo's files; the toolchain's own headers do not change
    except OSError:
        return False


def _compile(src: str, keep_asm: bool, tuning: bool, force: bool = False) -> str:
    objdir = OBJ_TUNING if tuning else OBJ
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    if not force and not keep_asm and _object_is_current(obj):
        return obj
    cmd = [hipcc()] + CXXFLAGS + (["-DXBIT_TUNING=1"] if tuning else []) + ["-MD", "-MF", obj + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
    if keep_asm:
        cmd += ["-save-temps=obj"]
    started = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=objdir)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    # the object carries the time its compile STARTED: a header edited while the compiler ran (these translation units
    # take minutes) then still counts as newer than the object, and the next build recompiles it
    os.utime(obj, (started, started))
    return obj


def build(force: bool = False, keep_asm: bool = False, verbose: bool = False, tuning: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library (the production one, or with
    ``tuning=True`` the measurement build).  Returns its path."""
    lib = TUNING_LIB if tuning else LIB
    objdir = OBJ_TUNING if tuning else OBJ
    # Fast path: a library newer than every source and header IS the build of these sources, whether or not its object
    # files are around.  (The GPU box receives the shipped .so but not csrc/_obj: without this every GPU session would
    # recompile all the translation units - minutes of lease time - and test a box-built library instead of the one whose
    # sha256 the bench records; a machine with the .so and no hipcc would fail where nothing needs compiling.)
    if not force and not keep_asm and is_current(tuning):
        return lib
    if verbose:
        print("xbitops build: compiling %s (sources newer than the library, or no library)" % os.path.basename(lib), flush=True)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    # every source goes through _compile, which recompiles only what is older than its own inputs (the .d files)
    srcs = _sources(tuning)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(lambda s: _compile(s, keep_asm, tuning, force), srcs))
    if not force and os.path.exists(lib) and all(os.path.getmtime(o) <= os.path.getmtime(lib) for o in objs):
        return lib
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib + ".tmp"] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # a kernel whose host stub was dropped shows up as an undefined symbol only at load time
    chk = subprocess.run([sys.executable, "-c", "import ctypes,sys; ctypes.CDLL(sys.argv[1])", lib + ".tmp"], capture_output=True, text=True)
    if chk.returncode != 0:
        raise RuntimeError("built library does not load:\n" + chk.stderr)
    os.replace(lib + ".tmp", lib)
    if verbose:
        print("built", lib)
    return lib


EXAMPLE_SRC = os.path.normpath(os.path.join(HERE, "..", "examples", "replay_dataset.cc"))
EXAMPLE_BIN = EXAMPLE_SRC[:-3]
HARNESS_SRC = os.path.join(CSRC, "xbit_harness.cc")
HARNESS_LIB = os.path.join(LIB_DIR, "libxbitops_harness.so")
HARNESS_MAIN_SRC = os.path.normpath(os.path.join(HERE, "..", "examples", "harness_main.cc"))
HARNESS_MAIN_BIN = HARNESS_MAIN_SRC[:-3]
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))


def _stale(target: str, *inputs: str) -> bool:
    return not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(p) for p in inputs)


def _run(cmd, what: str) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building %s failed:\n%s\n%s" % (what, r.stdout, r.stderr))


def build_example(force: bool = False) -> str:
    """Build examples/replay_dataset: a C++ host program over the C ABI (no torch), the counterpart
    of the reference's harness executable.  Needs the library (links -lxbitops_hip)."""
    lib = build()
    if force or _stale(EXAMPLE_BIN, EXAMPLE_SRC, lib):
        _run([hipcc(), "-O2", "-w", "-I", INCLUDE, EXAMPLE_SRC, "-o", EXAMPLE_BIN, "-L", LIB_DIR, "-lxbitops_hip", "-Wl,-rpath,$ORIGIN/../xbitops_amd/lib"], EXAMPLE_SRC)
    return EXAMPLE_BIN


def build_harness(force: bool = False) -> str:
    """libxbitops_harness.so — ``extern "C" int QbitGemv(SampleData*)`` of the reference's stand-alone
    harness (include/xbitops_harness.h) over the C ABI — and examples/harness_main, a host shaped like
    the reference's main.cc that calls it."""
    lib = build()
    if force or _stale(HARNESS_LIB, HARNESS_SRC, lib, os.path.join(INCLUDE, "xbitops_harness.h"), os.path.join(CSRC, "xbit_harness_check.h")):
        _run([hipcc(), "-O2", "-w", "-fPIC", "-shared", "-fvisibility=hidden", "-x", "hip", "--offload-arch=" + ARCH, HARNESS_SRC, "-o", HARNESS_LIB,
              "-L", LIB_DIR, "-lxbitops_hip", "-Wl,-rpath,$ORIGIN"], HARNESS_SRC)
    if force or _stale(HARNESS_MAIN_BIN, HARNESS_MAIN_SRC, HARNESS_LIB):
        _run(["g++", "-O2", "-I", INCLUDE, HARNESS_MAIN_SRC, "-o", HARNESS_MAIN_BIN, "-L", LIB_DIR, "-lxbitops_harness", "-lxbitops_hip",
              "-Wl,-rpath,$ORIGIN/../xbitops_amd/lib"], HARNESS_MAIN_SRC)
    return HARNESS_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, keep_asm="--asm" in sys.argv, verbose=True)
    if "--tuning" in sys.argv:
        build(force="--force" in sys.argv, keep_asm="--asm" in sys.argv, verbose=True, tuning=True)
    print("built", build_example(force="--force" in sys.argv))
    print("built", build_harness(force="--force" in sys.argv))
