"""QuantLinear — the caller the reference expects but does not ship.

The reference registers two ops for external quantised-Linear modules to call
(/root/reference/src/dq_torch_ops.cc:80-85).  This module is such a caller: it holds the packed
tensors in the reference layout and routes ``forward`` to the fused GEMV for decode-sized inputs
and to dequant + dense matmul for prefill-sized ones.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, packing

GEMV_MAX_ROWS = 16  # rows handled by one pass of the fused kernel; above this dequant + GEMM wins


class QuantLinear(torch.nn.Module):
    """y = x @ W (+ bias) with W stored as 2..8-bit group-quantised ``qweight/scales/qzeros``."""

    def __init__(self, in_features: int, out_features: int, bits: int, groupsize: int, add_zero_bias: int = 0, bias: bool = False):
        super().__init__()
        if out_features % 2:
            raise ValueError("out_features must be even (the packed layout stores column pairs)")
        groups = (in_features + groupsize - 1) // groupsize
        self.in_features, self.out_features, self.bits, self.groupsize, self.add_zero_bias = in_features, out_features, bits, groupsize, add_zero_bias
        self.register_buffer("qweight", torch.zeros((packing.packed_rows(in_features, bits), out_features), dtype=torch.int32))
        self.register_buffer("scales", torch.zeros((groups, out_features), dtype=torch.float16))
        self.register_buffer("qzeros", torch.zeros((groups, packing.packed_rows(out_features, bits)), dtype=torch.int32))
        self.bias = torch.nn.Parameter(torch.zeros(out_features, dtype=torch.float16)) if bias else None
        self._plan = None  # ops.GemvPlan for the current buffers (rebuilt when .to() / load_state_dict replaces them)

    @classmethod
    def from_float(cls, weight_kn, bits: int, groupsize: int = 128, add_zero_bias: int = 0) -> "QuantLinear":
        """Round-to-nearest group quantisation of ``weight_kn[K, N]`` (torch or numpy, float)."""
        w = weight_kn.detach().float().cpu().numpy() if isinstance(weight_kn, torch.Tensor) else np.asarray(weight_kn, np.float32)
        k, n = w.shape
        qweight, scales, qzeros = packing.quantize_rtn(w, bits, groupsize, add_zero_bias)
        m = cls(k, n, bits, groupsize, add_zero_bias)
        m.qweight.copy_(torch.from_numpy(qweight))
        m.scales.copy_(torch.from_numpy(scales))
        m.qzeros.copy_(torch.from_numpy(qzeros))
        return m

    def dequantize(self) -> torch.Tensor:
        return ops.dequant(self.qweight, self.scales, self.qzeros, self.groupsize, self.bits, self.in_features, self.add_zero_bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.in_features).to(torch.float16).contiguous()
        if x2.shape[0] <= GEMV_MAX_ROWS:
            plan = self._plan
            if plan is None or not plan.matches(self.qweight, self.scales, self.qzeros):
                plan = self._plan = ops.GemvPlan(self.qweight, self.scales, self.qzeros, self.groupsize, self.bits, self.in_features, self.add_zero_bias)
            y = plan(x2)
        else:
            y = x2 @ self.dequantize()
        if self.bias is not None:
            y = y + self.bias
        return y.reshape(*lead, self.out_features)


class QuantLinearGroup(torch.nn.Module):
    """Several :class:`QuantLinear` layers that read the SAME input (q/k/v, gate/up).

    ``forward(x)`` returns the list of their outputs.  Up to four rows (a decode step, a short speculative batch) go through
    ``ops.gemv_multi`` - one kernel launch for the whole group instead of one per layer, which is what
    the small layers of a decoder need (they sit on the launch floor); anything else runs the layers
    one by one.  The layers must share ``in_features``, ``bits``, ``groupsize`` and ``add_zero_bias``.
    """

    def __init__(self, layers):
        super().__init__()
        layers = list(layers)
        if not 1 <= len(layers) <= 4:
            raise ValueError("a group holds 1 to 4 layers")
        first = layers[0]
        for layer in layers[1:]:
            if (layer.in_features, layer.bits, layer.groupsize, layer.add_zero_bias) != (first.in_features, first.bits, first.groupsize, first.add_zero_bias):
                raise ValueError("grouped layers must share in_features, bits, groupsize and add_zero_bias")
        self.layers = torch.nn.ModuleList(layers)

    def forward(self, x: torch.Tensor):
        first = self.layers[0]
        lead = x.shape[:-1]
        x2 = x.reshape(-1, first.in_features)
        if not 1 <= x2.shape[0] <= 4:  # the fused launch takes up to 4 rows
            return [layer(x) for layer in self.layers]
        x2 = x2.to(torch.float16).contiguous()
        ys = ops.gemv_multi(x2, [(layer.qweight, layer.scales, layer.qzeros) for layer in self.layers], first.groupsize, first.bits,
                            first.in_features, first.add_zero_bias)
        out = []
        for layer, y in zip(self.layers, ys):
            if layer.bias is not None:
                y = y + layer.bias
            out.append(y.reshape(*lead, layer.out_features))
        return out
