/*
 * xbitops_harness.h — the benchmark ABI of the reference's stand-alone harness, served by
 * libxbitops_harness.so on top of the C ABI of include/xbitops_hip.h.
 *
 * The reference's executable (/root/reference/src/main.cc:92-105) fills a SampleData with HOST
 * buffers read from a seven-file data set and calls
 *
 *     extern "C" int QbitGemv(SampleData *data);        declared /root/reference/src/gemv.cuh:3-22,
 *                                                        called   /root/reference/src/main.cc:102
 *
 * which each of its experimental .cu files implements for one kernel (copy to the device, run,
 * compare with the golden tensors, print time and "gflops": gemv_w4a16.cu:300-440,
 * unpack_weight_2_to_7.cu:455-610).  This header declares the same struct, field for field, and
 * the same entry point, so main.cc links against libxbitops_harness.so unchanged.
 *
 * The matrix shape travels in three globals that main.cc defines (main.cc:7-9) and the .cu files
 * import (`extern int MATRIX_M, MATRIX_K, MATRIX_N;` unpack_weight_2_to_7.cu:30-32).  The library
 * reads them when the executable defines them and otherwise derives M, K and N from the byte sizes
 * in SampleData (K*N = f16weight_size/2, M*K = input_size/2, M*N = out_size/2).
 */
#ifndef XBITOPS_HARNESS_H_
#define XBITOPS_HARNESS_H_

#include <stdint.h>

/* /root/reference/src/gemv.cuh:3-20 — host pointers; sizes in BYTES */
struct SampleData {
  short *input;      /* fp16 [M, K] */
  int32_t *qweight;  /* [ceil(K*bits/32), N] */
  short *scale;      /* fp16 [ceil(K/group), N] */
  int32_t *qzero;    /* [ceil(K/group), ceil(N*bits/32)] */
  short *f16weight;  /* fp16 [K, N]: golden dequantised weight */
  short *out;        /* fp16 [M, N]: golden GEMV output */
  int32_t bits;
  int32_t group_size;
  struct {
    int input_size;
    int qweight_size;
    int scale_size;
    int qzero_size;
    int f16weight_size;
    int out_size;
  } size;
};

#ifdef __cplusplus
extern "C" {
#endif

/* Runs xbit_dequant_f16 and xbit_gemv_f16 on the data set, applies the reference's acceptance rules
 * (dequant: |diff| <= 1e-3, unpack_weight_2_to_7.cu:570; GEMV: error iff relative > 1e-3 AND
 * absolute > 4e-4, gemv_w4a16.cu:406-407), times 10 GEMV launches between two events as the reference
 * does (gemv_w4a16.cu:352-357) and prints the reference's report lines.  Returns 0 when both checks
 * pass, 1 when one fails, 2 on a usage / HIP error (the reference aborts there).
 * XBIT_HARNESS_ADD_ZERO_BIAS=1 in the environment applies the GPTQ "stored zero - 1" convention
 * (the harness ABI has no field for it). */
int QbitGemv(struct SampleData *data);

#ifdef __cplusplus
}
#endif

#endif /* XBITOPS_HARNESS_H_ */
