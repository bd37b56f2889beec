"""Multi-GPU form of the path on CPU ranks: world_size 2 and 4 over gloo (127.0.0.1).

Checks the column-shard repacking (qzeros bit ranges included), that per-rank GEMVs on shards
concatenate to the unsharded result, and the all-gather reassembly for M = 1 and M > 1.
The compute function injected here is the oracle (tests may use it as the checker); the product
default is the HIP op and is covered by the -m gpu suite.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from xbitops_amd import sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_gemv(a, qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias):
    y = oracle.gemv(a.numpy(), qweight.numpy(), scales.numpy(), qzeros.numpy(), groupsize, bits, in_features, add_zero_bias)
    return torch.from_numpy(y).reshape(*a.shape[:-1], qweight.shape[1])  # leading dimensions as the operator returns them


def _worker(rank, world, port, bits, k, n, g, m, bias, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle.set_num_threads(2)
        p = synth.make_problem(k, n, bits, g, m=m, seed=1234 + bits)
        t = lambda x: torch.from_numpy(x)
        op = sharded.ShardedGemv(t(p.qweight), t(p.scales), t(p.qzeros), g, bits, k, bias, gemv_fn=_oracle_gemv)
        assert op.n_local == n // world
        y = op(t(p.act))
        y_full = oracle.gemv(p.act, p.qweight, p.scales, p.qzeros, g, bits, k, bias)
        ok = y.shape == (m, n) and np.array_equal(y.numpy().view(np.uint16), y_full.view(np.uint16))
        # the shard itself dequantises to the matching column slice (pins the qzeros repack)
        w_shard = oracle.dequant(op.qweight.numpy(), op.scales.numpy(), op.qzeros.numpy(), g, bits, k, bias)
        w_full = oracle.dequant(p.qweight, p.scales, p.qzeros, g, bits, k, bias)
        n0, n1 = sharded.shard_bounds(n, bits, world, rank)
        ok = ok and np.array_equal(w_shard.view(np.uint16), w_full[:, n0:n1].view(np.uint16))
        # [B, S, K] activations: leading dimensions survive the gather (the operator accepts them, dq_torch_ops.cc:59-63)
        if m % 2 == 0:
            y3 = op(t(p.act).reshape(2, m // 2, k))
            ok = ok and y3.shape == (2, m // 2, n) and np.array_equal(y3.reshape(m, n).numpy().view(np.uint16), y_full.view(np.uint16))
        # bench.py's per-step sequence (--config cfg4): local GEMV into a preallocated shard, then the ONE
        # all-gather into a preallocated [P, M, N/P] buffer on the same stream
        out_local = op.local(t(p.act)).contiguous()
        gathered = torch.empty((world, m, op.n_local), dtype=out_local.dtype)
        sharded.all_gather_output(gathered, out_local)
        ok = ok and np.array_equal(sharded.full_from_gathered(gathered).numpy().view(np.uint16), y_full.view(np.uint16))
        ret[rank] = bool(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bits,k,n,g,m,bias", [
    (2, 4, 512, 256, 128, 1, 1),
    (2, 3, 256, 128, 64, 3, 0),
    (4, 5, 256, 512, 128, 1, 1),
    (2, 8, 128, 64, 32, 2, 0),
    (2, 4, 256, 7168 // 4, 128, 4, 1),  # the cfg4 split pattern (N/P on a qzeros word boundary), batch 4 as [2, 2, K]
])
def test_column_sharded_gemv_over_gloo(world, bits, k, n, g, m, bias):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bits, k, n, g, m, bias, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def test_shard_bounds_respect_zero_point_words():
    assert sharded.shard_bounds(28672, 4, 8, 3) == (3 * 3584, 4 * 3584)
    with pytest.raises(ValueError):
        sharded.shard_bounds(100, 4, 8, 0)  # not divisible
    with pytest.raises(ValueError):
        sharded.shard_bounds(8 * 12, 3, 8, 0)  # 12 columns * 3 bits is not a whole number of words
    assert sharded.shard_bounds(8 * 32, 3, 8, 1) == (32, 64)
