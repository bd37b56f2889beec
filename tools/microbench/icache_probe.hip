// icache_probe.hip — does straight-line code size cost a streaming kernel time on MI355X?
// The GEMV is ~11 KB of fully unrolled code; a pure read of the same bytes in the same pattern is
// ~0.3 us faster than the GEMV's loads-only build.  Here a whole-K 48-column read (all loads up front)
// executes N dummy VALU instructions between issuing its loads and consuming them - once as a rolled
// loop (a few hundred bytes of code) and once fully unrolled (4 bytes per instruction: N = 2048 -> 8 KB)
// - so both run the same instruction count and only the instruction-fetch footprint differs.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ uint32_t xcd_run(uint32_t b, uint32_t n) {
  const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int N, bool UNROLL>
__global__ void __launch_bounds__(512) k_probe(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n_words) {
  constexpr int LPR = 12, ROWS = 512, NW = 8, CHUNKS = ROWS * LPR / 64, CL = CHUNKS / NW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t tile = xcd_run(blockIdx.x, gridDim.x);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(w), 0, (int)((uint32_t)ROWS * n_words * 4u), 0x00020000);
  u32x4 v[CL];
#pragma unroll
  for (int i = 0; i < CL; ++i) {
    const uint32_t c = (uint32_t)(i * NW + wv);
    const uint32_t p = c * 64u + (uint32_t)lane, row = p / LPR, j = p % LPR;
    v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (row * n_words + tile * (LPR * 4u) + j * 4u) * 4u, 0, 0);
  }
  uint32_t x = threadIdx.x;
  if constexpr (N > 0) {
    if constexpr (UNROLL) {
#pragma unroll
      for (int t = 0; t < N; ++t) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(x));
    } else {
#pragma unroll 1
      for (int t = 0; t < N / 8; ++t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(x));
      }
    }
  }
  uint32_t acc = x;
#pragma unroll
  for (int i = 0; i < CL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <typename F>
void bench(const char* name, double bytes, int copies, F launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  std::vector<float> us;
  for (int i = 0; i < 10; ++i) launch(i % copies, nullptr, nullptr);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 200; ++i) {
    launch(i % copies, a, b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  double avg = 0;
  for (float x : us) avg += x;
  avg /= us.size();
  printf("%-58s %s avg %7.2f med %7.2f min %7.2f us\n", name, copies > 1 ? "cold" : "warm", avg, us[us.size() / 2], us[0]);
  fflush(stdout);
}

int main() {
  const uint32_t K_ROWS = 512, N = 11008;
  const size_t words = (size_t)K_ROWS * N;
  const double bytes = words * 4.0;
  const int copies = (int)((600e6 + bytes - 1) / bytes) + 1;
  uint32_t *w, *out;
  (void)hipMalloc(&w, words * 4 * copies + (1 << 20));
  (void)hipMalloc(&out, 4 << 20);
  (void)hipMemset(w, 1, words * 4 * copies);
  (void)hipDeviceSynchronize();
#define CASE(name, NI, UNR)                                                                         \
  for (int cp : {copies, 1})                                                                        \
    bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                                 \
      const uint32_t* ptr = w + (size_t)c * words;                                                  \
      uint32_t nw = N;                                                                              \
      void* args[] = {&ptr, &out, &nw};                                                             \
      if (s) (void)hipExtLaunchKernel((const void*)k_probe<NI, UNR>, dim3(230), dim3(512), args, 0, 0, s, e, 0); \
      else (void)hipLaunchKernel((const void*)k_probe<NI, UNR>, dim3(230), dim3(512), args, 0, 0);  \
    });
  for (int rep = 0; rep < 2; ++rep) {
    CASE("no dummy code", 0, false)
    CASE("512 VALU, rolled loop", 512, false)
    CASE("512 VALU, unrolled (2 KB)", 512, true)
    CASE("1024 VALU, rolled loop", 1024, false)
    CASE("1024 VALU, unrolled (4 KB)", 1024, true)
    CASE("2048 VALU, rolled loop", 2048, false)
    CASE("2048 VALU, unrolled (8 KB)", 2048, true)
    CASE("4096 VALU, rolled loop", 4096, false)
    CASE("4096 VALU, unrolled (16 KB)", 4096, true)
  }
  return 0;
}
