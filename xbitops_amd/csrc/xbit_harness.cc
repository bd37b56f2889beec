// xbit_harness.cc — libxbitops_harness.so: QbitGemv(SampleData*) of the reference's stand-alone
// harness (include/xbitops_harness.h) on top of the C ABI.  Host code only; the kernels are in
// libxbitops_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "../../include/xbitops_harness.h"
#include "../../include/xbitops_hip.h"
#include "xbit_harness_check.h"

// defined by the reference's main.cc (lines 7-9); absent when another host loads the library
extern int MATRIX_M __attribute__((weak));
extern int MATRIX_K __attribute__((weak));
extern int MATRIX_N __attribute__((weak));

namespace {

using xbit_harness::half_to_float;

struct DeviceBuffer {
  void* p = nullptr;
  ~DeviceBuffer() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  hipError_t upload(const void* host, size_t bytes) {
    hipError_t e = alloc(bytes);
    return e != hipSuccess ? e : hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
  }
};

#define HARNESS_HIP(expr)                                                              \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "QbitGemv: %s: %s\n", #expr, hipGetErrorString(e_));             \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

}  // namespace

extern "C" __attribute__((visibility("default"))) int QbitGemv(SampleData* data) {
  if (!data || !data->input || !data->qweight || !data->scale || !data->qzero) {
    fprintf(stderr, "QbitGemv: null data\n");
    return 2;
  }
  // shape: the executable's globals if it has them, else from the byte sizes
  uint64_t m = (&MATRIX_M && MATRIX_M > 0) ? (uint64_t)MATRIX_M : 0, k = (&MATRIX_K && MATRIX_K > 0) ? (uint64_t)MATRIX_K : 0,
           n = (&MATRIX_N && MATRIX_N > 0) ? (uint64_t)MATRIX_N : 0;
  if (!m || !k || !n) {
    if (!xbit_harness::derive_shape(data->size.input_size, data->size.f16weight_size, data->size.out_size, &m, &k, &n)) {
      fprintf(stderr, "QbitGemv: cannot derive the shape (sizes %d %d %d)\n", data->size.input_size, data->size.f16weight_size, data->size.out_size);
      return 2;
    }
  }
  const int bits = data->bits, group = data->group_size;
  const uint64_t kw = (k * bits + 31) / 32, groups = (k + group - 1) / group, zw = (n * bits + 31) / 32;
  if (bits < 2 || bits > 8 || group < 1 || (uint64_t)data->size.qweight_size < kw * n * 4 || (uint64_t)data->size.scale_size < groups * n * 2 ||
      (uint64_t)data->size.qzero_size < groups * zw * 4 || (uint64_t)data->size.input_size < m * k * 2) {
    fprintf(stderr, "QbitGemv: buffers too small for M=%llu K=%llu N=%llu bits=%d group=%d\n", (unsigned long long)m, (unsigned long long)k,
            (unsigned long long)n, bits, group);
    return 2;
  }
  const char* bias_env = getenv("XBIT_HARNESS_ADD_ZERO_BIAS");
  const uint8_t bias = (uint8_t)(bias_env ? atoi(bias_env) : 0);
  printf("\nM = %llu, N = %llu, K = %llu. bits = %d, group = %d, add_zero_bias = %d\n\n", (unsigned long long)m, (unsigned long long)n,
         (unsigned long long)k, bits, group, (int)bias);

  DeviceBuffer d_in, d_qw, d_sc, d_qz, d_w, d_y;
  HARNESS_HIP(d_in.upload(data->input, m * k * 2));
  HARNESS_HIP(d_qw.upload(data->qweight, kw * n * 4));
  HARNESS_HIP(d_sc.upload(data->scale, groups * n * 2));
  HARNESS_HIP(d_qz.upload(data->qzero, groups * zw * 4));
  HARNESS_HIP(d_w.alloc(k * n * 2));
  HARNESS_HIP(d_y.alloc(m * n * 2));

  printf("Running with xbit_dequant_f16 / xbit_gemv_f16...\n");
  int st = xbit_dequant_f16(d_w.p, d_qw.p, d_sc.p, d_qz.p, bits, group, (uint32_t)k, (uint32_t)n, bias, nullptr);
  if (st == 0) st = xbit_gemv_f16(d_y.p, d_in.p, d_qw.p, d_sc.p, d_qz.p, bits, group, (uint32_t)m, (uint32_t)k, (uint32_t)n, bias, nullptr);
  if (st != 0) {
    fprintf(stderr, "QbitGemv: %s\n", xbit_status_string(st));
    return 2;
  }
  HARNESS_HIP(hipDeviceSynchronize());
  const int n_iter = 10;  // as the reference (gemv_w4a16.cu:306, unpack_weight_2_to_7.cu:520)
  hipEvent_t start, stop;
  HARNESS_HIP(hipEventCreate(&start));
  HARNESS_HIP(hipEventCreate(&stop));
  HARNESS_HIP(hipEventRecord(start, nullptr));
  for (int i = 0; i < n_iter && st == 0; ++i)
    st = xbit_gemv_f16(d_y.p, d_in.p, d_qw.p, d_sc.p, d_qz.p, bits, group, (uint32_t)m, (uint32_t)k, (uint32_t)n, bias, nullptr);
  HARNESS_HIP(hipEventRecord(stop, nullptr));
  HARNESS_HIP(hipEventSynchronize(stop));
  float gemv_ms = 0.f;
  HARNESS_HIP(hipEventElapsedTime(&gemv_ms, start, stop));
  (void)hipEventDestroy(start);
  (void)hipEventDestroy(stop);
  if (st != 0) {
    fprintf(stderr, "QbitGemv: %s\n", xbit_status_string(st));
    return 2;
  }
  gemv_ms /= n_iter;

  printf("\nChecking results...\n");
  int dq_errors = 0, gemv_errors = 0;
  bool dq_checked = false, gemv_checked = false;
  if (data->f16weight && (uint64_t)data->size.f16weight_size >= k * n * 2) {
    std::vector<uint16_t> w(k * n);
    HARNESS_HIP(hipMemcpy(w.data(), d_w.p, w.size() * 2, hipMemcpyDeviceToHost));
    dq_errors = xbit_harness::count_dequant_errors(w.data(), reinterpret_cast<const uint16_t*>(data->f16weight), w.size(), 9);  // unpack_weight_2_to_7.cu:570
    dq_checked = true;
  }
  if (data->out && (uint64_t)data->size.out_size >= m * n * 2) {
    std::vector<uint16_t> y(m * n);
    HARNESS_HIP(hipMemcpy(y.data(), d_y.p, y.size() * 2, hipMemcpyDeviceToHost));
    gemv_errors = xbit_harness::count_gemv_errors(y.data(), reinterpret_cast<const uint16_t*>(data->out), y.size(), 9);  // gemv_w4a16.cu:406-407
    gemv_checked = true;
  }
  if (dq_errors || gemv_errors) {
    printf("xbitops does not agree with the golden data! %d dequant errors, %d gemv errors\n", dq_errors, gemv_errors);
    return 1;
  }
  if (!dq_checked && !gemv_checked) {
    // neither golden buffer is present (or large enough): nothing was compared - say so and do not report success
    printf("Results NOT checked: the data set holds no golden weight / output of the expected size.\n\n");
    printf("gemv took %fms, gflops:%f\n", gemv_ms, (double)m * (double)n * (double)k * 2 / gemv_ms / 1e6);
    return 3;
  }
  printf("Results verified: golden data and xbitops agree (%s%s%s).\n\n", dq_checked ? "dequant" : "", dq_checked && gemv_checked ? ", " : "",
         gemv_checked ? "gemv" : "");
  printf("gemv took %fms, gflops:%f\n", gemv_ms, (double)m * (double)n * (double)k * 2 / gemv_ms / 1e6);
  return 0;
}
