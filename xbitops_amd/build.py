"""Build libxbitops_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m xbitops_amd.build            # incremental
    python -m xbitops_amd.build --force    # rebuild everything
    python -m xbitops_amd.build --asm      # also keep .s files under xbitops_amd/csrc/_obj/

hipcc cross-compiles for gfx950 without a GPU present, so this runs in the build container; the
resulting .so travels to the GPU box with the working tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libxbitops_hip.so")
SOURCES = ["xbit_capi.hip", "xbit_dequant.hip", "xbit_gemv.hip", "xbit_gemv_mfma.hip"] + ["xbit_gemv_mfma_b%d.hip" % b for b in range(2, 9)]
HEADERS = ["xbit_device.h", "xbit_host.h", "xbit_unpack_gen.h", "xbit_gemv_mfma.h", os.path.join("..", "..", "include", "xbitops_hip.h")]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
            "--offload-arch=" + ARCH]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the xbitops HIP library cannot be built on this machine")
    return exe


def _newest_input() -> float:
    paths = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return max(os.path.getmtime(p) for p in paths)


def is_current() -> bool:
    return os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_input()


def _compile(src: str, keep_asm: bool) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [hipcc()] + CXXFLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    if keep_asm:
        cmd += ["-save-temps=obj"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, keep_asm: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    if not force and not keep_asm and is_current():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(lambda s: _compile(s, keep_asm), SOURCES))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB + ".tmp"] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # a kernel whose host stub was dropped shows up as an undefined symbol only at load time
    chk = subprocess.run([sys.executable, "-c", "import ctypes,sys; ctypes.CDLL(sys.argv[1])", LIB + ".tmp"], capture_output=True, text=True)
    if chk.returncode != 0:
        raise RuntimeError("built library does not load:\n" + chk.stderr)
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        print("built", LIB)
    return LIB


EXAMPLE_SRC = os.path.normpath(os.path.join(HERE, "..", "examples", "replay_dataset.cc"))
EXAMPLE_BIN = EXAMPLE_SRC[:-3]


def build_example(force: bool = False) -> str:
    """Build examples/replay_dataset: a C++ host program over the C ABI (no torch), the counterpart
    of the reference's harness executable.  Needs the library (links -lxbitops_hip)."""
    lib = build()
    if not force and os.path.exists(EXAMPLE_BIN) and os.path.getmtime(EXAMPLE_BIN) >= max(os.path.getmtime(EXAMPLE_SRC), os.path.getmtime(lib)):
        return EXAMPLE_BIN
    inc = os.path.normpath(os.path.join(HERE, "..", "include"))
    cmd = [hipcc(), "-O2", "-w", "-I", inc, EXAMPLE_SRC, "-o", EXAMPLE_BIN, "-L", LIB_DIR, "-lxbitops_hip", "-Wl,-rpath,$ORIGIN/../xbitops_amd/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building %s failed:\n%s\n%s" % (EXAMPLE_SRC, r.stdout, r.stderr))
    return EXAMPLE_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv, keep_asm="--asm" in sys.argv, verbose=True)
    print("built", build_example(force="--force" in sys.argv))
