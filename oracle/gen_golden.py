#!/usr/bin/env python3
"""Generate tests/golden/ from the REFERENCE simulator (oracle/_ref).  TEST INFRASTRUCTURE ONLY.

Run in the build container only (needs /root/reference to have been compiled by
``make -C oracle``):   python oracle/gen_golden.py

Outputs (data only — inputs and expected outputs, no reference source):

* ``tests/golden/dq_b{bits}_{tag}.npz`` — small dequant cases per bit-width: the packed inputs
  (from xbitops_amd.synth, seeded) and ``w_ref`` = the reference simulator's fp16 W
  (cpu::DequantizeAndUnpackWeight3567_v2<ushort,bits>, /root/reference/src/cpp_simulate.cc:568-691);
* ``tests/golden/digests.json`` — SHA-256 of the reference simulator's W for the full-size
  BASELINE.json configs (inputs are regenerated from the seed, W is too big to commit).

The reference simulator has no ``add_zero_bias`` argument, so the fixtures pin bias=0.  The
bias path is pinned indirectly: tests re-pack ``z+1`` (where it fits) and compare bias=1 on z
against the reference on z+1.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from xbitops_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# (tag, K, N, groupsize): aligned, ragged-K (K % 32 != 0, K*b % 32 != 0 for most b), multi-group
SMALL = [("k256n128g128", 256, 128, 128), ("k200n64g32", 200, 64, 32), ("k384n96g64", 384, 96, 64)]
# full-size configs of BASELINE.json that have a dequant: (name, K, N, bits, g)
FULL = [("cfg2_w4_4096x11008", 4096, 11008, 4, 128), ("cfg5_w3_4096x11008", 4096, 11008, 3, 128)] + [
    ("cfg3_w%d_4096x4096" % b, 4096, 4096, b, 128) for b in range(2, 9)
]


def seed_for(k, n, bits, g):
    return (k * 1000003 + n * 10007 + bits * 101 + g) & 0x7FFFFFFF


def main():
    if not oracle.ref_available():
        raise SystemExit("oracle/_ref/libxbit_ref.so missing: run `make -C oracle` where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)
    for bits in range(2, 9):
        for tag, k, n, g in SMALL:
            p = synth.make_problem(k, n, bits, g, m=1, seed=seed_for(k, n, bits, g))
            w_ref = oracle.ref_dequant(p.qweight, p.scales, p.qzeros, g, bits, k)
            np.savez_compressed(
                os.path.join(GOLD, "dq_b%d_%s.npz" % (bits, tag)),
                qweight=p.qweight, scales=p.scales, qzeros=p.qzeros, w_ref=w_ref,
                meta=np.array([k, n, bits, g, seed_for(k, n, bits, g)], np.int64),
            )
    digests = {}
    for name, k, n, bits, g in FULL:
        p = synth.make_problem(k, n, bits, g, m=1, seed=seed_for(k, n, bits, g))
        w_ref = oracle.ref_dequant(p.qweight, p.scales, p.qzeros, g, bits, k)
        digests[name] = {
            "k": k, "n": n, "bits": bits, "groupsize": g, "seed": seed_for(k, n, bits, g),
            "sha256_w_ref": hashlib.sha256(np.ascontiguousarray(w_ref).tobytes()).hexdigest(),
            "sha256_inputs": hashlib.sha256(p.qweight.tobytes() + p.scales.tobytes() + p.qzeros.tobytes()).hexdigest(),
        }
        print(name, digests[name]["sha256_w_ref"][:16])
    with open(os.path.join(GOLD, "digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
