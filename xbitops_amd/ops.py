"""The two XbitOps operators, host side (PyTorch-ROCm tensors in, HIP kernels underneath).

Mirror of the reference's operator binding, /root/reference/src/dq_torch_ops.cc:

* ``dequant(qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias)`` -> ``[K, N]``
  (``dequant_any_bit``, dq_torch_ops.cc:23-44)
* ``gemv(input_a, qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias)``
  -> ``[B, N]`` or ``[B, S, N]``   (``op_gemv``, dq_torch_ops.cc:46-78)

Same positional argument order, same shape/dtype rules, same validation (each check cites its
line below).  Differences, all deliberate:

* GEMV accepts bits 2..8 and any group size (the reference aborts unless bits==4 and
  groupsize==128, gemv_w4a16_pt.cu:152-155);
* both ops launch on torch's current stream of the tensors' device and restore the caller's
  current device (the reference leaves ``cudaSetDevice`` sticky, dq_torch_ops.cc:32,58, and runs
  GEMV on the legacy default stream, gemv_w4a16_pt.cu:162);
* the output is ``torch.empty`` — the kernels write every element (reference: ``at::zeros``,
  one extra memset per call, dq_torch_ops.cc:38,71);
* errors raise ``RuntimeError``; nothing aborts the process.

There is no CPU path: CPU tensors are rejected exactly as the reference's CHECK_CUDA does, and a
missing native library is an import-time error (xbitops_amd._lib).
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["dequant", "gemv", "gemv_ex", "gemv_gidx", "dequant_gidx", "gemv_multi", "GemvPlan"]


def _check_input(t: torch.Tensor, name: str) -> None:
    # CHECK_INPUT = CHECK_CUDA + CHECK_CONTIGUOUS (dq_torch_ops.cc:5-9)
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _check_common(qweight, scales, qzeros, groupsize: int, bits: int, in_features: int):
    _check_input(qweight, "qweight")
    _check_input(scales, "scales")
    _check_input(qzeros, "qzeros")
    if qweight.dim() != 2:  # dq_torch_ops.cc:28,53
        raise RuntimeError("qweight must be 2-dimensional")
    if groupsize < 16:  # :29,54
        raise RuntimeError("groupsize must be >= 16")
    if not 1 <= bits <= 8:  # :30,55
        raise RuntimeError("bits must be >= 1 and <= 8")
    if (in_features * bits + 31) // 32 != qweight.size(0):  # :31,56 (the reference's text is "in_features must be >= 1")
        raise RuntimeError(
            "in_features must be >= 1 and match qweight: ceil(in_features*bits/32) = %d but qweight has %d rows"
            % ((in_features * bits + 31) // 32, qweight.size(0))
        )
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:  # .data_ptr<int32_t>() (unpack_weight_2_to_7.cu:432)
        raise RuntimeError("qweight and qzeros must be int32")
    if scales.dtype not in (torch.float16, torch.bfloat16):  # bf16 is converted, :33-37; anything else fails data_ptr<half>
        raise RuntimeError("scales must be float16 or bfloat16")
    n = qweight.size(1)
    groups = (in_features + groupsize - 1) // groupsize
    # layout preconditions the reference indexes by but never checks
    if scales.dim() != 2 or scales.size(0) < groups or scales.size(1) != n:
        raise RuntimeError("scales must have shape [ceil(in_features/groupsize)=%d, N=%d], got %s" % (groups, n, tuple(scales.shape)))
    zw = (n * bits + 31) // 32
    if qzeros.dim() != 2 or qzeros.size(0) < groups or qzeros.size(1) != zw:
        raise RuntimeError("qzeros must have shape [%d, ceil(N*bits/32)=%d], got %s" % (groups, zw, tuple(qzeros.shape)))
    if qzeros.device != qweight.device or scales.device != qweight.device:
        raise RuntimeError("qweight, scales and qzeros must be on the same device")


def add_zero_bias_check(v: int) -> int:
    v = int(v)
    if not 0 <= v <= 255:  # uint8_t parameter (dq_torch_ops.cc:24,48)
        raise RuntimeError("add_zero_bias must fit in uint8")
    return v


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the handle without building a Stream object


def _stream_ptr(index: int) -> int:
    """Raw hipStream_t of torch's current stream on device ``index``."""
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


class _on_device:
    """``with _on_device(i):`` makes device ``i`` current and restores the caller's device afterwards;
    free when ``i`` is already current (the usual case) - ``torch.cuda.device`` costs microseconds."""

    __slots__ = ("idx", "prev")

    def __init__(self, idx: int):
        self.idx = idx
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def dequant(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, groupsize: int, bits: int, in_features: int,
            add_zero_bias: int) -> torch.Tensor:
    """Dequantise ``qweight`` to ``W[in_features, N]`` (fp16, or bf16 when ``scales`` is bf16)."""
    groupsize, bits, in_features = int(groupsize), int(bits), int(in_features)
    bias = add_zero_bias_check(add_zero_bias)
    _check_common(qweight, scales, qzeros, groupsize, bits, in_features)
    lib = _lib.load()
    ori_dtype = scales.dtype
    f16_scale = scales.to(torch.float16) if ori_dtype == torch.bfloat16 else scales  # dq_torch_ops.cc:33-37
    n = qweight.size(1)
    idx = qweight.device.index
    with _on_device(idx):
        out = torch.empty((in_features, n), dtype=torch.float16, device=qweight.device)
        st = lib.xbit_dequant_f16(out.data_ptr(), qweight.data_ptr(), f16_scale.data_ptr(), qzeros.data_ptr(), bits, groupsize,
                                  in_features, n, bias, _stream_ptr(idx))
    _lib.check(st, "XbitOps.dequant")
    return out.to(torch.bfloat16) if ori_dtype == torch.bfloat16 else out  # :40-42


def gemv(input_a: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, groupsize: int, bits: int,
         in_features: int, add_zero_bias: int) -> torch.Tensor:
    """``input_a[B, K]`` or ``[B, S, K]`` (fp16) times the dequantised weight -> ``[B, (S,) N]``."""
    return gemv_ex(input_a, qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias)


def gemv_ex(input_a: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, groupsize: int, bits: int,
            in_features: int, add_zero_bias: int, *, split_k: int = 0, workspace: torch.Tensor = None, variant: int = -1) -> torch.Tensor:
    """:func:`gemv` with the per-call options of ``xbit_gemv_f16_ex`` (include/xbitops_hip.h):

    * ``split_k``: 0 = the library chooses, 1 = never split K, 2..15 = split K that many ways where the
      shape allows (same result up to the fp32 summation order);
    * ``workspace``: a zero-filled device tensor of at least ``xbit_gemv_workspace_bytes(M, N)`` bytes that
      this launch alone uses as split-K scratch (it is zero again afterwards); default: library-managed;
    * ``variant`` > 0: a measurement variant — served by the TUNING build of the library
      (``python -m xbitops_amd.build --tuning``), never by the production one.
    """
    groupsize, bits, in_features = int(groupsize), int(bits), int(in_features)
    bias = add_zero_bias_check(add_zero_bias)
    _check_input(input_a, "input_a")
    _check_common(qweight, scales, qzeros, groupsize, bits, in_features)
    if qweight.device != input_a.device:  # dq_torch_ops.cc:57
        raise RuntimeError("input and weight must be on the same device")
    if input_a.dtype != torch.float16:  # a_fp16.data_ptr<half>() throws otherwise (gemv_w4a16_pt.cu:163)
        raise RuntimeError("input_a must be float16")
    if input_a.dim() not in (2, 3) or input_a.size(-1) != in_features:
        raise RuntimeError("input_a must be [B, in_features] or [B, S, in_features], got %s" % (tuple(input_a.shape),))
    lib = _lib.load_tuning() if variant > 0 else _lib.load()
    n = qweight.size(1)
    shape = [input_a.size(0), n]  # dq_torch_ops.cc:59-64
    mat_m = input_a.size(0)
    if input_a.dim() > 2:
        shape.insert(1, input_a.size(1))
        mat_m *= input_a.size(1)
    ori_dtype = scales.dtype
    f16_scale = scales.to(torch.float16) if ori_dtype == torch.bfloat16 else scales  # :65-69
    idx = qweight.device.index
    opts = None
    if split_k or workspace is not None or variant > 0:
        if workspace is not None and (not workspace.is_cuda or workspace.device != qweight.device or not workspace.is_contiguous()):
            raise RuntimeError("workspace must be a contiguous tensor on the weights' device")
        opts = _lib.GemvOptions(split_k=int(split_k), workspace=workspace.data_ptr() if workspace is not None else 0,
                                workspace_bytes=workspace.numel() * workspace.element_size() if workspace is not None else 0, variant=int(variant))
    with _on_device(idx):
        out = torch.empty(shape, dtype=torch.float16, device=qweight.device)
        if mat_m > 0:
            if opts is None:
                st = lib.xbit_gemv_f16(out.data_ptr(), input_a.data_ptr(), qweight.data_ptr(), f16_scale.data_ptr(), qzeros.data_ptr(),
                                       bits, groupsize, mat_m, in_features, n, bias, _stream_ptr(idx))
            else:
                st = lib.xbit_gemv_f16_ex(out.data_ptr(), input_a.data_ptr(), qweight.data_ptr(), f16_scale.data_ptr(), qzeros.data_ptr(),
                                          bits, groupsize, mat_m, in_features, n, bias, _stream_ptr(idx), opts)
            _lib.check(st, "XbitOps.gemv")
    return out.to(torch.bfloat16) if ori_dtype == torch.bfloat16 else out  # :74-76


def _check_gidx(g_idx: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, in_features: int) -> None:
    _check_input(g_idx, "g_idx")
    if g_idx.dtype != torch.int32 or g_idx.dim() != 1 or g_idx.numel() != in_features:
        raise RuntimeError("g_idx must be int32 [in_features]")
    if g_idx.device != qweight.device:
        raise RuntimeError("g_idx and weight must be on the same device")


def _check_common_gidx(qweight, scales, qzeros, bits: int, in_features: int) -> int:
    """The operator checks with the group count taken from ``scales`` (no groupsize with act-order)."""
    groups = scales.size(0) if scales.dim() == 2 else 0
    if groups < 1:
        raise RuntimeError("scales must be [groups, N]")
    # groupsize = in_features makes ceil(in_features / groupsize) == 1 <= groups: the row checks of _check_common hold for any G
    _check_common(qweight, scales, qzeros, max(16, in_features), bits, in_features)
    if qzeros.size(0) < groups:
        raise RuntimeError("qzeros must have one row per group")
    return groups


def gemv_gidx(input_a: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, g_idx: torch.Tensor, bits: int,
              in_features: int, add_zero_bias: int) -> torch.Tensor:
    """Act-order GEMV (``xbit_gemv_gidx_f16``): the group of row k is ``g_idx[k]``.  Counterpart of the
    reference's ``vecquant4matmul_g_cuda`` (quant_cuda_kernel.cu:221-336), bits 2..8.  Same shapes and
    dtypes as :func:`gemv`; ``scales`` / ``qzeros`` have one row per group."""
    bits, in_features = int(bits), int(in_features)
    bias = add_zero_bias_check(add_zero_bias)
    _check_input(input_a, "input_a")
    groups = _check_common_gidx(qweight, scales, qzeros, bits, in_features)
    _check_gidx(g_idx, qweight, scales, in_features)
    if qweight.device != input_a.device:
        raise RuntimeError("input and weight must be on the same device")
    if input_a.dtype != torch.float16:
        raise RuntimeError("input_a must be float16")
    if input_a.dim() not in (2, 3) or input_a.size(-1) != in_features:
        raise RuntimeError("input_a must be [B, in_features] or [B, S, in_features], got %s" % (tuple(input_a.shape),))
    lib = _lib.load()
    n = qweight.size(1)
    lead = tuple(input_a.shape[:-1])
    mat_m = 1
    for d in lead:
        mat_m *= d
    ori_dtype = scales.dtype
    f16_scale = scales.to(torch.float16) if ori_dtype == torch.bfloat16 else scales
    idx = qweight.device.index
    with _on_device(idx):
        out = torch.empty(lead + (n,), dtype=torch.float16, device=qweight.device)
        if mat_m > 0:
            st = lib.xbit_gemv_gidx_f16(out.data_ptr(), input_a.data_ptr(), qweight.data_ptr(), f16_scale.data_ptr(), qzeros.data_ptr(),
                                        g_idx.data_ptr(), bits, groups, mat_m, in_features, n, bias, _stream_ptr(idx))
            _lib.check(st, "XbitOps.gemv_gidx")
    return out.to(torch.bfloat16) if ori_dtype == torch.bfloat16 else out


def dequant_gidx(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, g_idx: torch.Tensor, bits: int, in_features: int,
                 add_zero_bias: int) -> torch.Tensor:
    """Act-order dequant (``xbit_dequant_gidx_f16``): ``W[in_features, N]`` with the group of row k = ``g_idx[k]``."""
    bits, in_features = int(bits), int(in_features)
    bias = add_zero_bias_check(add_zero_bias)
    groups = _check_common_gidx(qweight, scales, qzeros, bits, in_features)
    _check_gidx(g_idx, qweight, scales, in_features)
    lib = _lib.load()
    ori_dtype = scales.dtype
    f16_scale = scales.to(torch.float16) if ori_dtype == torch.bfloat16 else scales
    n = qweight.size(1)
    idx = qweight.device.index
    with _on_device(idx):
        out = torch.empty((in_features, n), dtype=torch.float16, device=qweight.device)
        st = lib.xbit_dequant_gidx_f16(out.data_ptr(), qweight.data_ptr(), f16_scale.data_ptr(), qzeros.data_ptr(), g_idx.data_ptr(), bits,
                                       groups, in_features, n, bias, _stream_ptr(idx))
    _lib.check(st, "XbitOps.dequant_gidx")
    return out.to(torch.bfloat16) if ori_dtype == torch.bfloat16 else out


def gemv_multi(input_a: torch.Tensor, weights, groupsize: int, bits: int, in_features: int, add_zero_bias: int):
    """Extension (not in the reference): ``[gemv(input_a, qw, sc, qz, ...) for (qw, sc, qz) in weights]`` for
    1..4 weight sets that share ``in_features``, ``bits`` and ``groupsize`` (q/k/v, gate/up) - in ONE
    kernel launch for up to 16 rows.  Same checks and results as :func:`gemv` per set; returns a list."""
    import ctypes

    groupsize, bits, in_features = int(groupsize), int(bits), int(in_features)
    bias = add_zero_bias_check(add_zero_bias)
    weights = list(weights)
    if not 1 <= len(weights) <= 4:
        raise RuntimeError("gemv_multi takes 1 to 4 weight sets")
    _check_input(input_a, "input_a")
    if input_a.dtype != torch.float16:
        raise RuntimeError("input_a must be float16")
    if input_a.dim() not in (2, 3) or input_a.size(-1) != in_features:
        raise RuntimeError("input_a must be [B, in_features] or [B, S, in_features], got %s" % (tuple(input_a.shape),))
    bf16 = []
    for qw, sc, qz in weights:
        _check_common(qw, sc, qz, groupsize, bits, in_features)
        if qw.device != input_a.device:
            raise RuntimeError("input and weight must be on the same device")
        bf16.append(sc.dtype == torch.bfloat16)
    lib = _lib.load()
    lead = tuple(input_a.shape[:-1])
    mat_m = 1
    for d in lead:
        mat_m *= d
    f16_scales = [sc.to(torch.float16) if b else sc for (_qw, sc, _qz), b in zip(weights, bf16)]
    idx = input_a.device.index
    count = len(weights)
    vp = ctypes.c_void_p
    with _on_device(idx):
        outs = [torch.empty(lead + (qw.size(1),), dtype=torch.float16, device=input_a.device) for qw, _sc, _qz in weights]
        if mat_m > 0:
            st = lib.xbit_gemv_multi_f16(
                count, (vp * count)(*[o.data_ptr() for o in outs]), input_a.data_ptr(),
                (vp * count)(*[qw.data_ptr() for qw, _s, _z in weights]), (vp * count)(*[s.data_ptr() for s in f16_scales]),
                (vp * count)(*[qz.data_ptr() for _q, _s, qz in weights]), (ctypes.c_uint32 * count)(*[qw.size(1) for qw, _s, _z in weights]),
                bits, groupsize, mat_m, in_features, bias, _stream_ptr(idx))
            _lib.check(st, "XbitOps.gemv_multi")
    return [o.to(torch.bfloat16) if b else o for o, b in zip(outs, bf16)]


class GemvPlan:
    """The checks of :func:`gemv` done once for a fixed set of weight tensors (a module's buffers).

    ``plan = GemvPlan(qweight, scales, qzeros, groupsize, bits, in_features, add_zero_bias)`` validates
    exactly as ``gemv`` does; ``plan(input_a)`` then only checks the activation and launches.  Saves
    the per-call validation of the weight-side tensors (about a quarter of the host cost of a call);
    the result is identical to ``gemv(input_a, ...)``.  The plan holds references to the tensors;
    build a new one if a buffer is replaced (``plan.matches`` tells).
    """

    __slots__ = ("qweight", "scales", "qzeros", "f16_scale", "groupsize", "bits", "in_features", "bias", "n", "idx", "ptrs", "bf16", "lib")

    def __init__(self, qweight, scales, qzeros, groupsize: int, bits: int, in_features: int, add_zero_bias: int):
        self.groupsize, self.bits, self.in_features = int(groupsize), int(bits), int(in_features)
        self.bias = add_zero_bias_check(add_zero_bias)
        _check_common(qweight, scales, qzeros, self.groupsize, self.bits, self.in_features)
        self.qweight, self.scales, self.qzeros = qweight, scales, qzeros
        self.bf16 = scales.dtype == torch.bfloat16
        self.f16_scale = scales.to(torch.float16) if self.bf16 else scales
        self.n = qweight.size(1)
        self.idx = qweight.device.index
        self.ptrs = (qweight.data_ptr(), self.f16_scale.data_ptr(), qzeros.data_ptr())
        self.lib = _lib.load()

    def matches(self, qweight, scales, qzeros) -> bool:
        return qweight is self.qweight and scales is self.scales and qzeros is self.qzeros

    def __call__(self, input_a: torch.Tensor) -> torch.Tensor:
        if not input_a.is_cuda or not input_a.is_contiguous():
            _check_input(input_a, "input_a")
        if input_a.device.index != self.idx:
            raise RuntimeError("input and weight must be on the same device")
        if input_a.dtype != torch.float16:
            raise RuntimeError("input_a must be float16")
        dim = input_a.dim()
        if dim not in (2, 3) or input_a.size(-1) != self.in_features:
            raise RuntimeError("input_a must be [B, in_features] or [B, S, in_features], got %s" % (tuple(input_a.shape),))
        if dim == 2:
            mat_m = input_a.size(0)
            shape = (mat_m, self.n)
        else:
            shape = (input_a.size(0), input_a.size(1), self.n)
            mat_m = shape[0] * shape[1]
        with _on_device(self.idx):
            out = torch.empty(shape, dtype=torch.float16, device=input_a.device)
            if mat_m > 0:
                st = self.lib.xbit_gemv_f16(out.data_ptr(), input_a.data_ptr(), self.ptrs[0], self.ptrs[1], self.ptrs[2], self.bits,
                                            self.groupsize, mat_m, self.in_features, self.n, self.bias, _stream_ptr(self.idx))
                if st != 0:
                    _lib.check(st, "XbitOps.gemv")
        return out.to(torch.bfloat16) if self.bf16 else out
