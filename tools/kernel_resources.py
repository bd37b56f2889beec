#!/usr/bin/env python3
"""VGPR / SGPR / spills / scratch / LDS of every kernel in the built library (read from the code objects
embedded in the .so: no --asm build needed).
    python tools/kernel_resources.py [substring of the demangled name] [--tuning]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib, codeobj  # noqa: E402


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"xbit::\(anonymous namespace\)::", "", o) for o in out]
    except FileNotFoundError:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pattern = args[0] if args else ""
    res = codeobj.kernel_resources(_lib.TUNING_LIB_PATH if "--tuning" in sys.argv else _lib.LIB_PATH)
    names = sorted(res)
    for mangled, n in zip(names, demangle(names)):
        n = re.sub(r"^void ", "", n).split("(")[0]
        if pattern in n:
            r = res[mangled]
            print("%-84s vgpr=%3d sgpr=%3d spill=%d scratch=%d lds=%d" % (n[:84], r["vgpr"], r["sgpr"], r["spill"], r["scratch"], r["lds"]))


if __name__ == "__main__":
    main()
