#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch per kernel from the .s files left by `python -m xbitops_amd.build --asm`."""
import glob
import os
import re
import subprocess
import sys

OBJ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "xbitops_amd", "csrc", "_obj")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"xbit::\(anonymous namespace\)::", "", o) for o in out]
    except FileNotFoundError:
        return names


def main(pattern=""):
    for f in sorted(glob.glob(os.path.join(OBJ, "*gfx950.s"))):
        s = open(f).read()
        rows = []
        for b in s.split("  - .agpr_count:")[1:]:
            g = lambda key: re.search(r"\." + key + r":\s+(\S+)", b).group(1)
            rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                         g("group_segment_fixed_size")))
        names = demangle([r[0] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r"^void ", "", n).split("(")[0]
            if pattern in n:
                print("%-80s vgpr=%3s sgpr=%3s spill=%s scratch=%s lds=%s" % (n[:80], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")
