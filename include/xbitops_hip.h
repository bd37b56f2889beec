/*
 * xbitops_hip.h — C ABI of libxbitops_hip.so, the MI355X (gfx950) implementation of the
 * XbitOps low-bit dequant / A16Wx GEMV hot path.
 *
 * This is the drop-in boundary: the two entry points below replace, one for one, the two native
 * launchers that the reference's operator binding calls
 *
 *   void lauch_deqantize_cuda_pt_kernel(Tensor& b_fp16, const Tensor& qweight_i32,
 *            const Tensor& scale_fp16, const Tensor& qzeros_i32, int bits, int groupsize,
 *            uint32_t mat_k, uint32_t mat_n, uint8_t add_zero_bias);
 *                 declared /root/reference/src/dq_torch_ops.cc:11-13,
 *                 defined  /root/reference/src/cu/unpack_weight_2_to_7.cu:426-441
 *
 *   void lauch_Gemv_kernel(Tensor& out_fp16, const Tensor& a_fp16, const Tensor& qweight_i32,
 *            const Tensor& scale_fp16, const Tensor& qzeros_i32, int bits, int groupsize,
 *            uint32_t mat_m, uint32_t mat_k, uint32_t mat_n, uint8_t add_zero_bias);
 *                 declared /root/reference/src/dq_torch_ops.cc:15-17,
 *                 defined  /root/reference/src/cu/gemv_w4a16_pt.cu:149-173
 *
 * with the same parameter order, torch::Tensor replaced by raw DEVICE pointers, and one extra
 * trailing argument: the HIP stream to launch on (the reference takes torch's current stream for
 * dequant, unpack_weight_2_to_7.cu:364, and the legacy default stream for GEMV,
 * gemv_w4a16_pt.cu:162).  Nothing here depends on torch.
 *
 * Tensor layout (identical to the reference's wire format):
 *   qweight  int32  [ceil(K*bits/32), N] row-major — per column an LSB-first bit-stream along K
 *   scales   fp16   [G = ceil(K/groupsize), N]
 *   qzeros   int32  [G, ceil(N*bits/32)] — per group an LSB-first bit-stream along N
 *   a        fp16   [M, K]          (GEMV activation rows, contiguous)
 *   out      fp16   [K, N] (dequant)  /  [M, N] (GEMV); every element is written, no pre-zeroing
 * Semantics:  W[k,n] = fp16( q*s - fp16((z + add_zero_bias)*s) ),  y[m,n] = fp16( sum_k a[m,k]*W[k,n] )
 * with the fp16 rounding of W kept exactly as the reference's hmul2/hfma2 produce it
 * (unpack_weight_2_to_7.cu:60,75,285,305) and the sum over k accumulated in fp32.
 *
 * Error convention: the reference aborts the process on any launcher error
 * (gemv_w4a16_pt.cu:152-155,168-172; unpack_weight_2_to_7.cu:436-440).  These functions never
 * abort: they return 0 on success, a negative XBIT_ERR_* for a rejected argument, or a positive
 * hipError_t if the HIP runtime refused the launch.  Launches are asynchronous on `hip_stream`.
 * The library is safe to call from several host threads.
 *
 * State.  A GEMV with few output columns and a long K is split over K (several workgroups per
 * column tile); the partial sums meet in 64-bit accumulators that must be zero when the launch
 * starts and are zero again when it ends.  That scratch is either
 *   - the caller's: xbit_gemv_f16_ex(..., &options) with options.workspace set (one launch at a time
 *     per buffer; xbit_gemv_workspace_bytes says how much), or
 *   - the library's: one 2 MiB buffer per (device, stream) for eager launches (launches on a stream
 *     run in order, so they can share it) and, for a launch recorded by a stream capture, a private
 *     range of a per-device pool that is not handed out again until xbit_workspace_release() - a captured
 *     graph may replay on any stream, concurrently with anything, so it cannot share.  Nothing is allocated inside a capture:
 *     the pool is created by the first eager split-K launch on the device or by xbit_workspace_reset().
 *     If no scratch can be had (no pool yet, allocation failure, pool exhausted) the launch simply
 *     is not split; the result differs from the split one only in the fp32 summation order.
 * xbit_workspace_reset() creates the pool if needed and re-zeroes the library's buffers; the library
 * re-zeroes by itself after a launch error, a host that killed work on the device by other means
 * calls it.
 */
#ifndef XBITOPS_HIP_H_
#define XBITOPS_HIP_H_

#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: only this header's functions are exported */
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define XBIT_OK 0
#define XBIT_ERR_NULL_POINTER (-1) /* a required device pointer is NULL */
#define XBIT_ERR_BITS (-2)         /* bits outside 2..8 (the reference prints "error bits" and aborts,
                                      unpack_weight_2_to_7.cu:418-420) */
#define XBIT_ERR_GROUPSIZE (-3)    /* groupsize < 1 */
#define XBIT_ERR_SHAPE (-4)        /* m, k or n is 0, n is odd (the reference stores column pairs,
                                      unpack_weight_2_to_7.cu:48,62), or a size overflows 32 bits */
#define XBIT_ERR_ALIGNMENT (-5)    /* a pointer is not aligned to its element size */
#define XBIT_ERR_UNSUPPORTED (-6)  /* an option this build of the library does not implement (measurement
                                      variants exist only in libxbitops_hip_tuning.so) */

/* ABI version of this header; bumped on any signature change. */
#define XBIT_ABI_VERSION 2
int xbit_abi_version(void);

/* Human-readable text for a status returned by this library (negative XBIT_ERR_*, 0, or a
 * positive hipError_t).  The returned string is static. */
const char* xbit_status_string(int status);

/* Replaces lauch_deqantize_cuda_pt_kernel (dq_torch_ops.cc:11-13).
 *   out      [k, n]  fp16, device            (reference: b_fp16)
 *   qweight, scales, qzeros as above, device
 *   bits 2..8; groupsize >= 1; k = in_features (mat_k); n = out_features (mat_n)
 *   add_zero_bias is added to every unpacked zero-point before use
 *   hip_stream: hipStream_t (NULL = default stream) */
int xbit_dequant_f16(void* out, const void* qweight, const void* scales, const void* qzeros, int bits, int groupsize,
                     uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream);

/* Replaces lauch_Gemv_kernel (dq_torch_ops.cc:15-17), generalised from the reference's
 * "bits == 4 && groupsize == 128 or abort" (gemv_w4a16_pt.cu:152-155) to bits 2..8 and any
 * groupsize.
 *   out  [m, n] fp16, device      a  [m, k] fp16, device
 *   m = rows of a (batch, or batch*seq flattened as dq_torch_ops.cc:60-64 does) */
int xbit_gemv_f16(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, int bits,
                  int groupsize, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream);

/* Per-call options of xbit_gemv_f16_ex.  Zero-initialise, set struct_bytes = sizeof(xbit_gemv_options),
 * then the fields you need; fields beyond struct_bytes take their defaults (the struct may grow). */
typedef struct xbit_gemv_options {
  uint32_t struct_bytes;
  int32_t split_k;          /* 0 = choose (default); 1 = never split K; 2..15 = split K that many ways when the
                               shape has a fast path (otherwise ignored).  Results differ only in summation order,
                               with ONE exception, the numeric contract of a split launch: the K ranges meet in 64-bit
                               fixed-point accumulators (2^-28 units).  While every range's partial sum stays inside
                               +-2^19 (8 x the largest finite fp16) the result is the exact sum of the partials rounded
                               once - cancellation between ranges included.  A partial sum beyond that saturates: the
                               output is +-Inf (NaN if both signs occur) even where the unsplit fp32 sum would have
                               cancelled back into range - it takes |W * a| ~ 1e2 per term over thousands of terms with an
                               fp16-representable total.  The automatic choice (0) splits only few-column, long-K shapes
                               (n <= ~5600 and k >= 8192 at one to four rows); pass 1 for the unsplit contract everywhere. */
  void* workspace;          /* caller-owned split-K scratch: device memory, zero-filled, >= xbit_gemv_workspace_bytes(m, n),
                               used by ONE launch at a time (it is zero again when that launch has finished).  NULL
                               (default) = library-managed, see "State" above. */
  uint64_t workspace_bytes; /* size of `workspace` */
  int32_t variant;          /* -1 or 0 = production launch (default).  > 0: measurement variant, tuning build only */
  int32_t reserved;
  void* trace;              /* tuning build only: device buffer for the stamped measurement kernel */
} xbit_gemv_options;

/* xbit_gemv_f16 with options (NULL options = xbit_gemv_f16). */
int xbit_gemv_f16_ex(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, int bits,
                     int groupsize, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream,
                     const xbit_gemv_options* options);

/* Bytes of zero-filled scratch that a split-K launch of an [m, k] x [k, n] GEMV may need at most
 * (8 bytes per output element, rounded up to whole column tiles and row passes). */
uint64_t xbit_gemv_workspace_bytes(uint32_t m, uint32_t n);

/* Create the capture pool if it does not exist yet and re-zero every library-managed scratch buffer of
 * the current device (synchronises the device; not to be called while a stream is being captured).
 * Returns 0 or a positive hipError_t. */
int xbit_workspace_reset(void);

/* Hand the whole capture pool of the current device out again (synchronises the device and re-zeroes it first).
 * A captured split-K launch keeps its range for the life of the process otherwise (the pool is 8 MiB: about 250
 * captured launches of a 4096-column layer); a host that re-captures its graphs - per batch size, per sequence
 * bucket - calls this once the old graphs are destroyed.  Graphs captured BEFORE the call must not be replayed
 * after it: their ranges may now belong to another launch.  Returns 0 or a positive hipError_t. */
int xbit_workspace_release(void);

/* Bytes of the current device's capture pool that are still free (0 before the pool exists).  A capture that
 * needs more than this runs its launch unsplit - same result up to the fp32 summation order, slower on long-K
 * layers - so a host can check before capturing instead of finding out from the timing. */
uint64_t xbit_workspace_capture_bytes_free(void);

/* Act-order ("desc_act") checkpoints: the group of row k is g_idx[k] instead of k / groupsize.
 * Counterpart of the reference's vecquant4matmul_g_cuda / VecQuant4MatMulKernel_G
 * (/root/reference/src/cu/quant_cuda_kernel.cu:221-336, compiled out there by `#if 0`), for bits 2..8
 * and with add_zero_bias in place of that kernel's hard-wired "+ 1" (:299-300):
 *     W[k,n] = fp16( q[k,n]*s[g,n] - fp16((z[g,n] + add_zero_bias)*s[g,n]) ),  g = g_idx[k]
 *   g_idx   int32 [k], device, values in [0, groups)  (larger values are clamped to groups - 1)
 *   groups  rows of `scales` [groups, n] and `qzeros` [groups, ceil(n*bits/32)]
 * These are the GENERAL entries (any g_idx).  When every group has the same number of rows — what
 * GPTQ act-order produces — sort the packed rows by group once at load time and call the ordinary
 * entry points on the sorted weight with the activation permuted alike (xbitops_amd.packing.
 * sort_rows_by_group; xbitops_amd.linear.QuantLinear does it): that runs at the ordinary kernels' speed. */
int xbit_gemv_gidx_f16(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, const void* g_idx,
                       int bits, uint32_t groups, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream);
int xbit_dequant_gidx_f16(void* out, const void* qweight, const void* scales, const void* qzeros, const void* g_idx, int bits,
                          uint32_t groups, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream);

/* Extension (not in the reference): one activation against `count` (1..4) weight sets that share k,
 * bits and groupsize - q/k/v or gate/up of a decoder layer - in ONE launch when the shape allows
 * (m <= 16, k % 8 == 0, a group size the staged-operand kernel indexes: a power-of-two multiple of 32, 64 at 2 bits),
 * otherwise as `count` xbit_gemv_f16 launches; results as xbit_gemv_f16 per set.  outs / qweights / scales / qzeros / ns
 * are HOST arrays of `count` device pointers / column counts.  Small layers sit on the launch floor, which only fewer
 * launches remove: the seven linears of a Llama-7B decoder layer at batch 1 take 37.2 us as seven launches and 31.7 us
 * with q/k/v and gate/up fused (13B: 54.7 -> 43.8; profiles/r04/decode_layer.txt). */
int xbit_gemv_multi_f16(int count, void* const* outs, const void* a, const void* const* qweights, const void* const* scales,
                        const void* const* qzeros, const uint32_t* ns, int bits, int groupsize, uint32_t m, uint32_t k,
                        uint8_t add_zero_bias, void* hip_stream);

/* Measurement aid (not part of the reference's surface; used by bench.py only).
 * While enabled on the calling host thread, every launch made by that thread is bracketed by a
 * pair of HIP events that record the kernel's own begin/end on the device (hipExtLaunchKernel),
 * so the launch duration can be read without inter-launch gaps.  xbit_profile_last_ms waits for
 * the most recent profiled launch of this thread and returns its duration in milliseconds.
 * Returns 0, a negative XBIT_ERR_*, or a positive hipError_t. */
int xbit_profile_enable(int on);
int xbit_profile_last_ms(float* ms);
/* The same for a SERIES of launches queued back to back: after xbit_profile_series_begin(capacity) every launch of the
 * calling thread gets its own event pair (up to `capacity` launches; further ones run untimed) and nothing waits on
 * the host, so the device sees the launches as a dependent chain - a kernel that starts on an idle chip pays up to a
 * microsecond of XCD wake-up skew that a kernel inside a chain does not (profiles/r03/trace_detail_v63.txt).
 * xbit_profile_series_end waits for the last launch, writes the durations (milliseconds, launch order) of the first
 * min(launches, capacity, max_count) launches to `ms`, their number to `*count`, and leaves profiling disabled. */
int xbit_profile_series_begin(int capacity);
int xbit_profile_series_end(float* ms, int max_count, int* count);

/* What WOULD be launched (not part of the reference's surface).  Between xbit_dry_run_begin() and xbit_dry_run_end()
 * every entry point called by this host thread goes through its argument checks and its launch rule as usual, but each
 * kernel launch is RECORDED instead of issued: nothing touches the device (it works on a machine without one; device
 * pointers are never followed, so any non-null, suitably aligned values do), split-K scratch is not allocated.
 * A record names the kernel by the offset of its host-side handle inside this library (`nm -C` maps it to the template
 * instantiation), and carries grid, block and dynamic LDS bytes.  Used by tests/ and tools/check_rules.py to pin the
 * launch rules to a table of shapes, and by callers who want to know which path a layer will take.
 * xbit_dry_run_end writes the first min(records, max_count) records, their number to `*count`, and ends the dry run. */
typedef struct xbit_launch_record {
  uint64_t kernel_offset;     /* address of the kernel's host-side handle minus the load address of this library */
  uint32_t grid[3], block[3];
  uint32_t dynamic_lds_bytes;
  uint32_t reserved;
} xbit_launch_record;
int xbit_dry_run_begin(void);
int xbit_dry_run_end(xbit_launch_record* records, int max_count, int* count);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#endif /* XBITOPS_HIP_H_ */
