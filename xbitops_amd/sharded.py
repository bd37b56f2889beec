"""Column-sharded A16Wx GEMV across GPUs (one process per GPU, torch.distributed / RCCL).

The reference has no multi-GPU code at all.  The path shards naturally along
the OUTPUT columns: ``y[:, n]`` depends only on column ``n`` of ``qweight`` / ``scales`` /
``qzeros`` and on the whole activation — the reference kernel itself tiles N in independent
64-column blocks (/root/reference/src/cu/gemv_w4a16_pt.cu:50,158).  So:

* rank ``r`` of ``P`` owns columns ``[r*N/P, (r+1)*N/P)``: ``qweight[:, n0:n1]``,
  ``scales[:, n0:n1]`` and the matching bit range of every ``qzeros`` row, re-packed contiguous
  once at load time (``shard_columns``);
* the activation (a few KB) is replicated;
* there is NO collective inside the GEMV; the only exchange is one all-gather of the fp16 output
  shards (``N/P * M * 2`` bytes per rank — 7 KB at 8 GPUs for the Llama-70B FFN), issued on the
  compute stream right behind the kernel.  K is never split across GPUs (that would need a
  reduce-scatter of fp32 partials and breaks group alignment).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .packing import packed_rows

__all__ = ["shard_bounds", "shard_columns", "all_gather_output", "full_from_gathered", "ShardedGemv"]


def shard_bounds(n: int, bits: int, world: int, rank: int):
    """Column range of ``rank``.  Boundaries must fall on a qzeros word boundary and on an even column."""
    if n % world:
        raise ValueError("out_features %d is not divisible by world size %d" % (n, world))
    per = n // world
    unit = 32 // np.gcd(bits, 32)  # columns per whole number of qzeros words
    if per % unit or per % 2:
        raise ValueError("shard width %d must be a multiple of %d columns for %d-bit zero-points" % (per, max(unit, 2), bits))
    return rank * per, (rank + 1) * per


def shard_columns(qweight, scales, qzeros, bits: int, world: int, rank: int):
    """Return this rank's contiguous ``(qweight, scales, qzeros)`` shard (numpy or torch in, same kind out)."""
    n = qweight.shape[1]
    n0, n1 = shard_bounds(n, bits, world, rank)
    z0, z1 = n0 * bits // 32, n0 * bits // 32 + packed_rows(n1 - n0, bits)
    qw, sc, qz = qweight[:, n0:n1], scales[:, n0:n1], qzeros[:, z0:z1]
    if isinstance(qweight, torch.Tensor):
        return qw.contiguous(), sc.contiguous(), qz.contiguous()
    return np.ascontiguousarray(qw), np.ascontiguousarray(sc), np.ascontiguousarray(qz)


def all_gather_output(gathered: torch.Tensor, y_local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """The ONE collective of a column-sharded layer: ``gathered[r] = rank r's [M, N/P] output shard``.
    Issued on the current stream right behind the kernel (``bench.py`` times exactly this per step)."""
    dist.all_gather_into_tensor(gathered.view(-1), y_local.view(-1), group=group)


def full_from_gathered(gathered: torch.Tensor) -> torch.Tensor:
    """``[P, M, N/P]`` rank-major shards -> ``[M, N]``.  One row: rank-major IS column-major, no copy."""
    world, m, n_local = gathered.shape
    if m == 1:
        return gathered.view(1, world * n_local)
    return gathered.permute(1, 0, 2).reshape(m, world * n_local)


class ShardedGemv:
    """``y = gemv(a, W)`` with W column-sharded over the ranks of ``group``.

    ``gemv_fn`` defaults to :func:`xbitops_amd.ops.gemv` (the HIP path).  It is a parameter so that
    the sharding / gather logic can be exercised on CPU ranks (gloo) in tests with a checker
    function; the product never substitutes anything for the HIP op.
    """

    def __init__(self, qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, add_zero_bias: int,
                 group: Optional[dist.ProcessGroup] = None, gemv_fn: Optional[Callable] = None, already_sharded: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if already_sharded or self.world == 1:
            self.qweight, self.scales, self.qzeros = qweight, scales, qzeros
        else:
            self.qweight, self.scales, self.qzeros = shard_columns(qweight, scales, qzeros, bits, self.world, self.rank)
        self.groupsize, self.bits, self.in_features, self.add_zero_bias = groupsize, bits, in_features, add_zero_bias
        self.n_local = self.qweight.shape[1]
        if gemv_fn is None:
            from .ops import gemv as gemv_fn  # the HIP operator; raises if the native library is missing
        self._gemv = gemv_fn

    def local(self, a: torch.Tensor) -> torch.Tensor:
        """This rank's output columns ``[M, N/P]`` (no communication)."""
        return self._gemv(a, self.qweight, self.scales, self.qzeros, self.groupsize, self.bits, self.in_features, self.add_zero_bias)

    def __call__(self, a: torch.Tensor) -> torch.Tensor:
        """Full ``[..., N]`` output on every rank: local GEMV + one all-gather of the output shards.
        ``a`` is ``[M, K]`` or ``[B, S, K]`` as for the operator (dq_torch_ops.cc:59-63)."""
        y_local = self.local(a)
        if self.world == 1:
            return y_local
        lead = y_local.shape[:-1]
        y2 = y_local.reshape(-1, self.n_local)  # leading dimensions flattened: the gather works on [M, N/P]
        gathered = torch.empty((self.world, y2.shape[0], self.n_local), dtype=y2.dtype, device=y2.device)
        all_gather_output(gathered, y2.contiguous(), self.group)
        return full_from_gathered(gathered).reshape(*lead, self.world * self.n_local)
