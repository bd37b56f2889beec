// stream_floor.hip — what a 23 MB read costs on MI355X in the launch shapes the GEMV uses, with none
// of the GEMV's arithmetic: the floor the real kernel is measured against.
//   k_empty        : launch + drain only
//   k_tile<W,...>  : grid of column tiles; each lane issues LOADS row loads of W bytes (rows strided by
//                    the matrix row pitch, like the packed weight rows), xors them, writes one dword
//   k_linear       : the same bytes as one flat contiguous stream (lane-contiguous 16-byte loads)
// Timing: hipExtLaunchKernel start/stop events (kernel-only), cold (rotating over > 512 MB) and warm.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(uint32_t* out) {
  if (out == nullptr) out[0] = 1;
}

// tile kernel: NL lanes along N (each W bytes), 64/NL k-slots per wave, NW waves; each lane reads
// LOADS consecutive rows starting at row (wave*LK + kl) * LOADS
// ADJ: the 64/NL k-slots of a wave read ADJACENT rows in each instruction (row = base + i*LK + kl)
// instead of rows LOADS apart (row = base + kl*LOADS + i)
template <int W, int NL, int NW, int LOADS, bool NT, bool ADJ = false>
__global__ void __launch_bounds__(64 * NW) k_tile(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n_words, uint32_t rows) {
  constexpr int LK = 64 / NL, WW = W / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane / NL, nl = lane % NL;
  const uint32_t col = (blockIdx.x * NL + nl) * WW;
  uint32_t acc = 0;
  for (uint32_t r0 = ADJ ? wv * LK * LOADS + kl : (wv * LK + kl) * LOADS; r0 < rows; r0 += NW * LK * LOADS) {
    uint32_t v[LOADS][WW];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const uint32_t* src = w + (size_t)(r0 + (ADJ ? i * LK : i)) * n_words + col;
      if constexpr (W == 8) {
        u32x2 t = NT ? __builtin_nontemporal_load((const u32x2*)src) : *(const u32x2*)src;
        v[i][0] = t.x; v[i][1] = t.y;
      } else if constexpr (W == 12) {
        u32x3 t = NT ? __builtin_nontemporal_load((const u32x3*)src) : *(const u32x3*)src;
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z;
      } else if constexpr (W == 16) {
        u32x4 t = NT ? __builtin_nontemporal_load((const u32x4*)src) : *(const u32x4*)src;
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      } else {
        v[i][0] = NT ? __builtin_nontemporal_load(src) : *src;
      }
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
#pragma unroll
      for (int j = 0; j < WW; ++j) acc ^= v[i][j];
  }
  out[blockIdx.x * 64 * NW + threadIdx.x] = acc;
}

template <bool NT>
__global__ void __launch_bounds__(256) k_linear(const u32x4* __restrict__ w, uint32_t* out, uint32_t n16, int per_thread) {
  // block handles a contiguous chunk of 256*per_thread 16-byte elements
  const size_t base = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
  uint32_t acc = 0;
  u32x4 v[8];
  for (int i0 = 0; i0 < per_thread; i0 += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t idx = base + (size_t)(i0 + i) * 256;
      v[i] = idx < n16 ? (NT ? __builtin_nontemporal_load(w + idx) : w[idx]) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { hipEventCreate(&a); hipEventCreate(&b); }
};

template <typename F>
void bench(const char* name, double bytes, int copies, F launch) {
  Timer t;
  std::vector<float> us;
  for (int i = 0; i < 10; ++i) launch(i % copies, nullptr, nullptr);
  hipDeviceSynchronize();
  for (int i = 0; i < 200; ++i) {
    launch(i % copies, t.a, t.b);
    hipEventSynchronize(t.b);
    float ms;
    hipEventElapsedTime(&ms, t.a, t.b);
    us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  double avg = 0;
  for (float x : us) avg += x;
  avg /= us.size();
  printf("%-52s %s avg %7.2f med %7.2f min %7.2f us  %8.1f GB/s (avg)\n", name, copies > 1 ? "cold" : "warm", avg, us[us.size() / 2], us[0],
         bytes / avg * 1e-3);
}

int main(int argc, char** argv) {
  // default: 4-bit 4096 x 11008 = 512 packed rows of 11008 words; `stream_floor 1024 28672` = 8192 x 28672
  const uint32_t K_ROWS = argc > 2 ? (uint32_t)atoi(argv[1]) : 512, N = argc > 2 ? (uint32_t)atoi(argv[2]) : 11008;
  const size_t words = (size_t)K_ROWS * N;
  const double bytes = words * 4.0;
  const int copies = (int)((600e6 + bytes - 1) / bytes) + 1;  // > 512 MB
  printf("# %u packed rows x %u words = %.1f MB per launch, %d copies\n", K_ROWS, N, bytes * 1e-6, copies);
  uint32_t* w;
  uint32_t* out;
  hipMalloc(&w, words * 4 * copies);
  hipMalloc(&out, 4 << 20);
  hipMemset(w, 1, words * 4 * copies);
  hipDeviceSynchronize();

  auto ext = [](const void* f, dim3 g, dim3 b, void** args, hipEvent_t s, hipEvent_t e) {
    if (s) hipExtLaunchKernel(f, g, b, args, 0, 0, s, e, 0);
    else hipLaunchKernel(f, g, b, args, 0, 0);
  };
#define TILE_CASE(name, W, NL, NW, LOADS, NT) TILE_CASE_A(name, W, NL, NW, LOADS, NT, false)
#define TILE_CASE_A(name, W, NL, NW, LOADS, NT, ADJ)                                                     \
  for (int cp : {copies, 1})                                                                              \
    bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                                       \
      const uint32_t* ptr = w + (size_t)c * words;                                                        \
      uint32_t nw = N, rows = K_ROWS;                                                                     \
      void* args[] = {&ptr, &out, &nw, &rows};                                                            \
      ext((const void*)k_tile<W, NL, NW, LOADS, NT, ADJ>, dim3(N / (NL * (W / 4))), dim3(64 * NW), args, s, e); \
    });

  for (int cp : {1})
    bench("empty 344 x 512", 0, cp, [&](int, hipEvent_t s, hipEvent_t e) {
      void* args[] = {&out};
      ext((const void*)k_empty, dim3(344), dim3(512), args, s, e);
    });
  TILE_CASE("tile 8B x16lanes(32col) NW8 16 loads nt", 8, 16, 8, 16, true)
  TILE_CASE("tile 8B x16lanes(32col) NW8 16 loads", 8, 16, 8, 16, false)
  TILE_CASE("tile 8B x16lanes(32col) NW16 8 loads nt", 8, 16, 16, 8, true)
  TILE_CASE("tile 12B x16lanes(48col) NW8 16 loads nt", 12, 16, 8, 16, true)
  TILE_CASE("tile 12B x16lanes(48col) NW8 16 loads", 12, 16, 8, 16, false)
  TILE_CASE_A("tile 12B x16lanes(48col) NW8 16 loads, adjacent rows", 12, 16, 8, 16, false, true)
  TILE_CASE_A("tile 16B x16lanes(64col) NW8 16 loads, adjacent rows", 16, 16, 8, 16, false, true)
  TILE_CASE_A("tile 8B x16lanes(32col) NW8 16 loads, adjacent rows", 8, 16, 8, 16, false, true)
  TILE_CASE("tile 16B x16lanes(64col) NW8 16 loads nt", 16, 16, 8, 16, true)
  TILE_CASE("tile 16B x16lanes(64col) NW8 16 loads", 16, 16, 8, 16, false)
  TILE_CASE("tile 16B x16lanes(64col) NW16 8 loads nt", 16, 16, 16, 8, true)
  TILE_CASE("tile 16B x8lanes(32col) NW4 16 loads nt", 16, 8, 4, 16, true)
  TILE_CASE("tile 16B x8lanes(32col) NW8 8 loads nt", 16, 8, 8, 8, true)
  TILE_CASE("tile 8B x8lanes(16col) NW4 16 loads nt", 8, 8, 4, 16, true)
  TILE_CASE("tile 8B x8lanes(16col) NW8 8 loads nt", 8, 8, 8, 8, true)
  TILE_CASE("tile 16B x32lanes(128col) NW16 16 loads nt", 16, 32, 16, 16, true)
  TILE_CASE("tile 16B x64lanes(256col) NW16 16 loads nt(2 sweeps)", 16, 64, 16, 16, true)
  for (int per : {8, 16, 32})
    for (int cp : {copies, 1}) {
      char nm[96];
      snprintf(nm, sizeof nm, "linear 16B, %d loads/thread, 256 thr/block nt", per);
      bench(nm, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {
        const u32x4* ptr = (const u32x4*)(w + (size_t)c * words);
        uint32_t n16 = words / 4;
        int p = per;
        void* args[] = {&ptr, &out, &n16, &p};
        ext((const void*)k_linear<true>, dim3((n16 + 256 * per - 1) / (256 * per)), dim3(256), args, s, e);
      });
    }
  return 0;
}
