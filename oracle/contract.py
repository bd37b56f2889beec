"""The GEMV acceptance contract, in ONE place (tests, __graft_entry__.smoke() and bench.py's output check use it).

HIP output y against the oracle's y_ref = fp16(sum_k a*W) with W rebuilt bit-exactly and the sum taken in double:

1. north star (BASELINE.json): |y - y_ref| <= 1e-3 * |y_ref| + 2^-14.  The absolute term is one fp16 unit in the last place at
   |y| ~ 0.06 - outputs that cancel to nearly nothing have no meaningful relative error.
2. much tighter, because the kernels rebuild the SAME fp16 W and only the order of an fp32 sum differs: y equals y_ref up to
   the last fp16 bit (<= 1 ulp, same sign), or - where the sum cancels and its own rounding noise exceeds an fp16 ulp -
   |y - y_ref| <= cancel_slack(k).

cancel_slack(k) = k * max|a*W| * 2^-24: an fp32 sum of k terms whose partial sums stay below sum|a*W| <= k * max|a*W| carries
an absolute error of the order of one fp32 rounding (2^-24 relative) of that magnitude.  For synth.make_problem operands
|a| < 1 and |W| = |q - z - bias| * s <= 2^bits * 0.01 * 16 / 2^bits = 0.16, so max|a*W| <= 0.16 whatever the bit-width:
4096 * 0.16 * 2^-24 = 3.9e-5, 8192: 7.8e-5.  (Observed: a few 1e-6; rounds 2-3 used hand-picked 4e-6 / 3e-5.)
Tests on other operands pass their own `term`.
"""
from __future__ import annotations

import numpy as np

SYNTH_TERM = 0.16  # max |a * W| of synth.make_problem operands (see above)


def cancel_slack(k: int, term: float = SYNTH_TERM) -> float:
    return float(k) * term * 2.0 ** -24


def _bits(x):
    return np.ascontiguousarray(x).view(np.uint16)


def gemv_mismatch(y, y_ref, k: int, term: float = SYNTH_TERM):
    """None when `y` meets the contract against `y_ref` (both float16 arrays of one shape), else a description."""
    y, y_ref = np.asarray(y), np.asarray(y_ref)
    if y.shape != y_ref.shape or y.dtype != np.float16 or y_ref.dtype != np.float16:
        return "shape / dtype: %s %s against %s %s" % (y.shape, y.dtype, y_ref.shape, y_ref.dtype)
    yf, rf = y.astype(np.float64), y_ref.astype(np.float64)
    err = np.abs(yf - rf)
    bad = ~(err <= 1e-3 * np.abs(rf) + 2.0 ** -14)
    if bad.any():
        return "%d of %d outside 1e-3 relative; worst error %.3e at reference %.4f" % (bad.sum(), bad.size, np.nanmax(err), rf.flat[np.nanargmax(err)])
    ulp = np.abs(_bits(y).astype(np.int32) - _bits(y_ref).astype(np.int32))
    same_sign = (_bits(y) >> 15) == (_bits(y_ref) >> 15)
    loose = ~(((ulp <= 1) & same_sign) | (err <= cancel_slack(k, term)))
    if loose.any():
        return "%d of %d more than one fp16 ulp off and beyond the cancellation slack %.2e; worst error %.3e" % (loose.sum(), loose.size, cancel_slack(k, term), err[loose].max())
    return None


def assert_gemv_close(y, y_ref, k: int, what="", term: float = SYNTH_TERM) -> None:
    why = gemv_mismatch(y, y_ref, k, term)
    assert why is None, "%s: %s" % (what, why)
