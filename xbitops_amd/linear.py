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

# Rows served by the fused kernel.  Up to 32 rows are one launch that unpacks every weight once; beyond that ONE launch takes
# the rows in blocks of 32 or 64 (column groups that share the activation tile, csrc/xbit_gemv_rows.h, up to 512 rows).  Measured against dequantise + torch's dense fp16 GEMM, cold weights
# (profiles/r04/linear_crossover.txt, us per call): 4096x11008 w4 11.8 against 45.5 at 32 rows, 18.0 / 49.6 at 64,
# 24.3 / 60.5 at 128, 40.1 / 75.7 at 256, 69.3 / 89.8 at 384, 79.1 / 94.2 at 512 (8192x8192: 36.8 / 107.5 at 128, 129.9 / 131.9 at
# 512: level).  The dense path starts near 40 us whatever the row count (19 us of dequant + a small-M GEMM) and gains 0.1 us per
# row; the fused one gains 0.15 from 64 rows on - they meet near 512.  Long K (the rows kernel keeps the tables as a window there): 28672x8192 w4 96 against 310 us at 64 rows,
# 126 / 309 at 128, 237 / 392 at 256, 402 / 486 at 512; 14336x4096 111 / 164 at 512 (profiles/r04/linear_crossover_long_k.txt).
GEMV_MAX_ROWS = 512
GEMV_MAX_ROWS_2BIT = 512  # (profiles/r04/linear_crossover_w2.txt: 4096x11008 w2 at 256 rows 54.7 against 70.8, at 512 rows 92.0 / 94.7)
# ... for layers the multi-row builds take: a group size that is a power-of-two multiple of 32 (64 at 2 bits), ANY in_features (round 4:
# a row length that is no multiple of 8 is masked where it is staged; 7166x4096 w4 at 32 rows 17.2 us).  Other group sizes (48, 80, 96 ...)
# run the by-id builds, 32 rows per launch (4096x4096 w4 at group 48: 16.6 us at 24 rows, 29.6 at 40 - profiles/r04/sweep_ragged_shapes.txt),
# so beyond 32 rows they go to dequantise + dense GEMM.
GEMV_MAX_ROWS_OTHER = 32
# General act-order layers (any g_idx): the library takes rows sixteen to a launch (4096x11008 w4: 10.4 us at one row, 11.9
# at sixteen - profiles/r03/gidx_rows_8_16.txt); three launches (36 us) still beat the act-order dequant (25 us) plus the
# dense GEMM (23 us and more), four do not.
GIDX_GEMV_MAX_ROWS = 48
# ... where the by-id builds take the layer (its table of every group fits in LDS: 512 B per group, about 290 groups); a
# layer with more groups runs the round-2 act-order kernel, one launch per row - worth it for a decode step only.
GIDX_GEMV_MAX_ROWS_OTHER = 4
GIDX_TABLE_MAX_GROUPS = 290


def fused_rows_limit(in_features: int, groupsize: int, bits: int) -> int:
    """Rows up to which a contiguous-group layer of this shape should take the fused kernel rather than dequantise + dense GEMM."""
    unit = 64 if bits == 2 else 32
    q = groupsize // unit
    regular = groupsize % unit == 0 and q >= 1 and (q & (q - 1)) == 0
    if not regular:
        return GEMV_MAX_ROWS_OTHER
    return GEMV_MAX_ROWS_2BIT if bits == 2 else GEMV_MAX_ROWS


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
        # act-order checkpoints: either the rows were sorted by group at load time and `perm` re-orders the
        # activation to match (fast: the ordinary kernels), or `g_idx` is kept and the general entry points run
        self.register_buffer("perm", None)
        self.register_buffer("g_idx", None)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """A state dict saved from an act-order layer carries `perm` or `g_idx`; a freshly constructed layer has both as None
        buffers, which torch would report as unexpected keys (strict) or drop without a word (strict=False) - leaving sorted
        rows with no permutation, i.e. silently wrong results.  Adopt them here, and drop a local one the incoming dict lacks."""
        for name, length in (("perm", self.in_features), ("g_idx", self.in_features)):
            key = prefix + name
            if key in state_dict:
                t = state_dict[key]
                if t.dim() != 1 or t.shape[0] != length:
                    error_msgs.append("size mismatch for %s: expected a vector of %d row indices, got %s" % (key, length, tuple(t.shape)))
                    state_dict = dict(state_dict)
                    del state_dict[key]
                    continue
                if getattr(self, name) is None or getattr(self, name).shape != t.shape or getattr(self, name).dtype != t.dtype:
                    setattr(self, name, torch.empty_like(t, device=self.qweight.device))
            elif getattr(self, name) is not None:
                setattr(self, name, None)  # the checkpoint is not (this kind of) act-order
        self._plan = None
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

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

    @classmethod
    def from_act_order(cls, qweight, scales, qzeros, g_idx, bits: int, groupsize: int, add_zero_bias: int = 0) -> "QuantLinear":
        """An act-order (GPTQ ``desc_act``) checkpoint: ``g_idx[k]`` is the group of row k (numpy or torch, host).
        Equal-sized groups are sorted into the contiguous layout once, here (``packing.sort_rows_by_group``);
        any other ``g_idx`` is kept and served by the general ``*_gidx`` kernels."""
        to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        qweight, scales, qzeros, g_idx = to_np(qweight), to_np(scales), to_np(qzeros), to_np(g_idx).astype(np.int32)
        k, n = g_idx.shape[0], qweight.shape[1]
        m = cls(k, n, bits, groupsize, add_zero_bias)
        if scales.shape[0] != m.scales.shape[0]:
            raise ValueError("scales must have ceil(in_features / groupsize) rows")
        try:
            qweight, perm = packing.sort_rows_by_group(qweight, g_idx, bits, groupsize)
            m.perm = torch.from_numpy(perm)
        except ValueError:
            m.g_idx = torch.from_numpy(g_idx)
        m.qweight.copy_(torch.from_numpy(np.ascontiguousarray(qweight).view(np.int32)))
        m.scales.copy_(torch.from_numpy(np.ascontiguousarray(scales, dtype=np.float16)))
        m.qzeros.copy_(torch.from_numpy(np.ascontiguousarray(qzeros).view(np.int32)))
        return m

    @classmethod
    def from_transposed_w4(cls, qweight_t, scales_t, qzeros_t, groupsize: int, in_features: int, add_zero_bias: int = 0) -> "QuantLinear":
        """The transposed 4-bit layout of the reference's experimental gemv_NT (gemv_w4a16_nt.cu:64-159)."""
        to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        qweight, scales, qzeros = packing.from_transposed_w4(to_np(qweight_t), to_np(scales_t), to_np(qzeros_t))
        m = cls(in_features, qweight.shape[1], 4, groupsize, add_zero_bias)
        m.qweight.copy_(torch.from_numpy(qweight.view(np.int32)))
        m.scales.copy_(torch.from_numpy(scales.astype(np.float16)))
        m.qzeros.copy_(torch.from_numpy(qzeros.view(np.int32)))
        return m

    def dequantize(self) -> torch.Tensor:
        """``W[K, N]`` in the ORIGINAL row order (what ``x @ W`` multiplies), whatever the storage order."""
        if self.g_idx is not None:
            return ops.dequant_gidx(self.qweight, self.scales, self.qzeros, self.g_idx, self.bits, self.in_features, self.add_zero_bias)
        w = ops.dequant(self.qweight, self.scales, self.qzeros, self.groupsize, self.bits, self.in_features, self.add_zero_bias)
        if self.perm is not None:
            out = torch.empty_like(w)
            out[self.perm] = w  # W_sorted[k'] = W[perm[k']]
            return out
        return w

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.in_features).to(torch.float16).contiguous()
        if self.g_idx is not None:  # general act-order: the group of row k is g_idx[k]
            if x2.shape[0] <= (GIDX_GEMV_MAX_ROWS if self.scales.shape[0] <= GIDX_TABLE_MAX_GROUPS else GIDX_GEMV_MAX_ROWS_OTHER):
                y = ops.gemv_gidx(x2, self.qweight, self.scales, self.qzeros, self.g_idx, self.bits, self.in_features, self.add_zero_bias)
            else:
                y = x2 @ self.dequantize()
            if self.bias is not None:
                y = y + self.bias
            return y.reshape(*lead, self.out_features)
        if self.perm is not None:  # rows sorted by group at load time: the same sum over k in that order
            x2 = x2.index_select(1, self.perm)
        if x2.shape[0] <= fused_rows_limit(self.in_features, self.groupsize, self.bits):
            plan = self._plan
            if plan is None or not plan.matches(self.qweight, self.scales, self.qzeros):
                plan = self._plan = ops.GemvPlan(self.qweight, self.scales, self.qzeros, self.groupsize, self.bits, self.in_features, self.add_zero_bias)
            y = plan(x2)
        else:
            y = x2 @ ops.dequant(self.qweight, self.scales, self.qzeros, self.groupsize, self.bits, self.in_features, self.add_zero_bias)
        if self.bias is not None:
            y = y + self.bias
        return y.reshape(*lead, self.out_features)


FUSED_GROUP_MAX_ROWS = 16  # rows one fused launch takes (xbit_gemv_multi_f16 on the staged-operand kernel)


class QuantLinearGroup(torch.nn.Module):
    """Several :class:`QuantLinear` layers that read the SAME input (q/k/v, gate/up).

    ``forward(x)`` returns the list of their outputs.  Up to sixteen rows (a decode step, a speculative or small batch) go
    through ``ops.gemv_multi`` - one kernel launch for the whole group instead of one per layer, which is what
    the small layers of a decoder need (they sit on the launch floor); anything else runs the layers
    one by one.  The layers must share ``in_features``, ``bits``, ``groupsize`` and ``add_zero_bias``; act-order layers
    (``perm`` / ``g_idx``) are run one by one.
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
        if not 1 <= x2.shape[0] <= FUSED_GROUP_MAX_ROWS or any(layer.perm is not None or layer.g_idx is not None for layer in self.layers):
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
