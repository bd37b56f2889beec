#!/usr/bin/env python3
"""What the library would launch for a grid of (bits, rows, K, N, group size) - through the dry run of the C ABI, no GPU.

    python tools/launch_rules_grid.py > tests/golden/launch_rules.txt        # record
    python tools/launch_rules_grid.py --check tests/golden/launch_rules.txt  # compare (tests/test_launch_rules.py does this)

One line per call: the shape, then per launch the kernel (dyn = gemv_dyn_kernel of xbit_gemv_dyn.h, rows = gemv_rows_kernel of
xbit_gemv_rows.h, mfma = gemv_mfma_kernel of xbit_gemv_mfma.h, with their template arguments), grid x * y, block, dynamic LDS bytes.  The file pins
the launch RULES: an edit to them shows up as a diff to review next to the sweep that justifies it (tools/check_rules.py
measures, on the GPU, that the rules still pick the fastest launch)."""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xbitops_amd import _lib  # noqa: E402

A, QW, SC, QZ, OUT = 0x10000000, 0x20000000, 0x30000000, 0x40000000, 0x50000000  # never followed in a dry run

SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 14336), (14336, 4096), (5120, 5120), (5120, 13824), (13824, 5120), (8192, 8192), (8192, 1024),
          (8192, 3584), (8192, 28672), (28672, 8192), (4096, 32000), (8192, 22016), (4096, 6144), (12288, 3072), (16384, 6144), (28672, 6144), (12288, 8192),
          (20480, 2560), (32768, 1024), (4096, 512), (2048, 8192), (256, 64), (7168, 4096), (3072, 9216), (6656, 6656)]
GROUPS = [128, 64, 32, 96]
ROWS = [1, 2, 4, 5, 8, 9, 16, 17, 32, 33, 128, 129, 256, 257, 512, 513]


def short(kernel: str) -> str:
    m = re.search(r"(\w+_kernel)<([^>]*)>", kernel)
    return "%s<%s>" % (m.group(1).replace("gemv_", "").replace("_kernel", ""), m.group(2).replace(" ", "").replace("false", "f").replace("true", "t")) if m else kernel


def lines(lib):
    for bits in range(2, 9):
        for k, n in SHAPES:
            for g in GROUPS:
                if k % g:
                    continue
                for m in ROWS:
                    def call():
                        _lib.check(lib.xbit_gemv_f16(OUT, A, QW, SC, QZ, bits, g, m, k, n, 1, None), "xbit_gemv_f16")

                    recs = _lib.dry_run(lib, call)
                    yield "w%d m%-3d %5dx%-5d g%-3d : %s" % (bits, m, k, n, g, " + ".join(
                        "%s %dx%d*%d lds %d" % (short(r["kernel"]), r["grid"][0], r["grid"][1], r["block"][0], r["lds"]) for r in recs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", default="")
    args = ap.parse_args()
    lib = _lib.load()
    got = list(lines(lib))
    if not args.check:
        print("\n".join(got))
        return 0
    want = open(args.check).read().split("\n")
    want = [w for w in want if w]
    diff = [(w, g) for w, g in zip(want, got) if w != g]
    if len(want) != len(got) or diff:
        print("%d of %d launches differ from %s" % (len(diff) + abs(len(want) - len(got)), len(want), args.check))
        for w, g in diff[:40]:
            print("- " + w + "\n+ " + g)
        return 1
    print("%d launches as recorded" % len(got))
    return 0


if __name__ == "__main__":
    sys.exit(main())
