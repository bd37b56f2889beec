// arrival_trace.hip — WHEN does the data of each of a wave's outstanding loads arrive?
// Every wave issues LOADS 16-byte row loads back to back (same tile shape as the GEMV), then waits
// for them one at a time (s_waitcnt vmcnt(LOADS-1-j)) and stamps s_memrealtime (100 MHz) after each
// wait.  Output: per load index j, the distribution over all waves of (t_j - t_kernel_start).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int LOADS = 8;

template <int NW>
__global__ void __launch_bounds__(64 * NW) k_trace(const uint32_t* __restrict__ w, uint64_t* stamps, uint32_t* sink, uint32_t n_words,
                                                  uint32_t rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane >> 4, nl = lane & 15;
  const uint32_t col = (blockIdx.x * 16 + nl) * 4;
  const uint32_t r0 = (wv * 4 + kl) * LOADS;
  uint64_t t[LOADS + 2];
  t[0] = __builtin_amdgcn_s_memrealtime();
  u32x4 v[LOADS];
  const uint32_t* base = w + (size_t)r0 * n_words + col;
#define LD(i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(base + (size_t)(i) * n_words) : "memory");
  LD(0) LD(1) LD(2) LD(3) LD(4) LD(5) LD(6) LD(7)
#undef LD
  t[1] = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#define WAITN(n, j)                                                          \
  asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(v[j]) : : "memory");         \
  t[2 + j] = __builtin_amdgcn_s_memrealtime();                               \
  acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  WAITN(7, 0) WAITN(6, 1) WAITN(5, 2) WAITN(4, 3) WAITN(3, 4) WAITN(2, 5) WAITN(1, 6) WAITN(0, 7)
  sink[blockIdx.x * 64 * NW + threadIdx.x] = acc;
  if (lane == 0) {
    uint64_t* o = stamps + (size_t)(blockIdx.x * NW + wv) * (LOADS + 2);
    for (int i = 0; i < LOADS + 2; ++i) o[i] = t[i];
  }
}

int main() {
  const uint32_t K_ROWS = 512, N = 11008;
  const size_t words = (size_t)K_ROWS * N;
  const int copies = 26, NW = 16;
  const int blocks = N / 64;
  uint32_t *w, *sink;
  uint64_t* st;
  hipMalloc(&w, words * 4 * copies);
  hipMalloc(&sink, 4 << 20);
  hipMalloc(&st, sizeof(uint64_t) * blocks * NW * (LOADS + 2));
  hipMemset(w, 1, words * 4 * copies);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<uint64_t> h((size_t)blocks * NW * (LOADS + 2));
    for (int it = 0; it < 30; ++it) {
      const uint32_t* ptr = w + (size_t)(mode == 0 ? it % copies : 0) * words;
      k_trace<NW><<<blocks, 64 * NW>>>(ptr, st, sink, N, K_ROWS);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    uint64_t t0 = ~0ull;
    for (size_t i = 0; i < h.size(); i += LOADS + 2) t0 = std::min(t0, h[i]);
    printf("%s: per-stamp distribution over %d waves, in us after the first wave started (10 ns ticks)\n", mode == 0 ? "COLD" : "WARM",
           blocks * NW);
    const char* names[LOADS + 2] = {"wave start", "loads issued", "load0 back", "load1 back", "load2 back", "load3 back",
                                    "load4 back", "load5 back", "load6 back", "load7 back"};
    for (int j = 0; j < LOADS + 2; ++j) {
      std::vector<double> d;
      for (size_t i = 0; i < h.size(); i += LOADS + 2) d.push_back((double)(h[i + j] - t0) * 0.01);
      std::sort(d.begin(), d.end());
      printf("  %-12s min %5.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f\n", names[j], d[0], d[d.size() / 10], d[d.size() / 2],
             d[d.size() * 9 / 10], d.back());
    }
  }
  return 0;
}
