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
 * The library is safe to call from several host threads.  Its only state is a 2 MiB scratch buffer
 * per (device, stream), allocated the first time a GEMV with few output columns (N <= 4608) and
 * K >= 8192 is split over K on that stream (never inside a stream capture).
 */
#ifndef XBITOPS_HIP_H_
#define XBITOPS_HIP_H_

#include <stdint.h>

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

/* ABI version of this header; bumped on any signature change. */
#define XBIT_ABI_VERSION 1
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

/* Extension (not in the reference): one activation against `count` (1..4) weight sets that share k,
 * bits and groupsize - q/k/v or gate/up of a decoder layer - in ONE launch when the shape allows
 * (m <= 4 and the fast path's alignment rules), otherwise as `count` xbit_gemv_f16 launches; results
 * as xbit_gemv_f16 per set.  outs / qweights / scales / qzeros / ns are HOST arrays of `count`
 * device pointers / column counts.  Small layers sit on the ~4 us launch floor, which only fewer
 * launches remove: three 4096x4096 w4 GEMVs take 3 x 4.3 us, fused 6.9 us. */
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

#ifdef __cplusplus
}
#endif

#endif /* XBITOPS_HIP_H_ */
