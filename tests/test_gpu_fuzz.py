"""Randomised parity on the GPU: XbitOps.gemv / dequant against the oracle over random (bits, K, N, M, group,
add_zero_bias) - ragged K / N, the shapes that take split-K plans (scored and measured), the 8/16-wave rule, the
16-column and 128-column tile paths and the generic kernel.  XBIT_FUZZ_CASES scales the number of cases (default 80)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import contract
from xbitops_amd import synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.ascontiguousarray(x).view(np.uint16)


def _gemv_mismatch(y, y_ref, k):
    return contract.gemv_mismatch(y, y_ref, k)  # oracle/contract.py: the one GEMV contract


def test_random_shapes_against_oracle():
    import XbitOps

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda", 0)
    cases = int(os.environ.get("XBIT_FUZZ_CASES", "80"))
    rng = np.random.default_rng(int(os.environ.get("XBIT_FUZZ_SEED", "20260401")))
    ks = [512, 1024, 2048, 4096, 5120, 8192, 11008, 13824, 14336, 16384, 4096 + 32, 8192 - 64, 2976]
    ns = [64, 130, 256, 1000, 1024, 2048, 2560, 3584, 4096, 5120, 6144, 8192, 11008, 14336, 4096 + 6, 12288, 28672]
    failures = []
    for i in range(cases):
        bits = int(rng.integers(2, 9))
        k, n = int(rng.choice(ks)), int(rng.choice(ns))
        m = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8, 9, 16]))
        g = int(rng.choice([128, 128, 128, 64, 32, 256]))
        bias = int(rng.integers(0, 2))
        if k * n > (48 << 20):
            n = 1024  # keeps the oracle to well under a second per case
        p = synth.make_problem(k, n, bits, g, m=m, seed=int(rng.integers(1, 1 << 30)))
        y_ref = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
        a = torch.from_numpy(p.act).to(dev)
        qw, sc, qz = (torch.from_numpy(x).to(dev) for x in (p.qweight, p.scales, p.qzeros))
        why = _gemv_mismatch(XbitOps.gemv(a, qw, sc, qz, g, bits, k, bias).cpu().numpy(), y_ref, k=k)
        if why is None and i % 4 == 0 and k * n <= (16 << 20):
            w_ref = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, bias, oracle.RNE)
            if not np.array_equal(_bits(XbitOps.dequant(qw, sc, qz, g, bits, k, bias).cpu().numpy()), _bits(w_ref)):
                why = "dequant not bit-exact"
        if why:
            failures.append("bits=%d k=%d n=%d m=%d g=%d bias=%d: %s" % (bits, k, n, m, g, bias, why))
    assert not failures, "\n".join(failures)
