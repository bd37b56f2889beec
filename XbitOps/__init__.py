"""XbitOps — drop-in module name of the reference extension (setup.py:100, TORCH_EXTENSION_NAME).

Exposes exactly the two callables the reference registers (/root/reference/src/dq_torch_ops.cc:80-85),
with identical positional arguments, implemented for MI355X in ``xbitops_amd``::

    XbitOps.dequant(qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias) -> Tensor[K, N]
    XbitOps.gemv(input_a, qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias) -> Tensor[B, (S,) N]
"""
from xbitops_amd.ops import dequant, gemv  # noqa: F401

__all__ = ["dequant", "gemv"]
