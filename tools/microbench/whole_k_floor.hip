// whole_k_floor.hip — pure-read floors of WHOLE-K column tiles (no cross-workgroup reduction) for
// lane mappings the round-1 kernel did not try.  No arithmetic: loads, xor, one store.
//
// Round 1 read 48-column tiles as 16 lanes x 12 bytes x 4 rows per wave-instruction (768 B, dwordx3)
// and measured that pattern's pure read at the GEMV's own time.  Candidates here keep every load a
// 16-byte dwordx4 and put more bytes into each wave-instruction:
//   rows<LPR>   LPR lanes x 16 B side by side (tile = 4*LPR columns), floor(64/LPR) rows per instruction
//               (LPR = 12: 48-column tiles, 5 rows, 60 lanes, 960 B; LPR = 11: 44 columns, 251 tiles)
//   chunks<LPR> the tile's rows seen as one sequence of 16-byte chunks, 64 consecutive chunks per
//               instruction (1 KB, every lane busy; a lane's column group changes from load to load)
// Timing: hipExtLaunchKernel start/stop events, cold (rotating over > 512 MB) and warm.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcd_run(uint32_t b, uint32_t n) {
  const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;

// MODE 0: rows<LPR>, MODE 1: chunks<LPR>, MODE 2: round-1 pattern (16 lanes x 12 B x 4 rows; LPR ignored)
// DEPTH: loads in flight per wave before the first is consumed (0 = all)
template <int LPR, int NW, int LOADS, int MODE, bool XCD, bool NT>
__global__ void __launch_bounds__(64 * NW) k_whole(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n_words, uint32_t rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t tile = XCD ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(w), 0, (int)(rows * n_words * 4u), 0x00020000);
  constexpr int aux = NT ? 2 : 0;
  uint32_t acc = 0;
  if constexpr (MODE == 2) {
    const int kl = lane >> 4, nl = lane & 15;
    u32x3 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const uint32_t row = (uint32_t)((wv * 4 + kl) * LOADS + i);
      const uint32_t off = row < rows ? (row * n_words + tile * 48u + nl * 3u) * 4u : 0x80000000u;
      v[i] = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, aux);
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z;
  } else {
    constexpr int R = 64 / LPR;
    u32x4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const uint32_t g = (uint32_t)(i * NW + wv);  // wave-instruction index within the tile
      uint32_t row, j;
      bool on;
      if constexpr (MODE == 0) {
        const uint32_t seg = lane / LPR;
        j = lane % LPR;
        row = g * R + seg;
        on = seg < (uint32_t)R && row < rows;
      } else {
        const uint32_t c = g * 64u + lane;
        row = c / LPR;
        j = c % LPR;
        on = row < rows;
      }
      const uint32_t off = on ? (row * n_words + tile * (LPR * 4u) + j * 4u) * 4u : 0x80000000u;
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, aux);
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  out[blockIdx.x * 64 * NW + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_linear(const u32x4* __restrict__ w, uint32_t* out, uint32_t n16, int per_thread) {
  const size_t base = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
  uint32_t acc = 0;
  u32x4 v[8];
  for (int i0 = 0; i0 < per_thread; i0 += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t idx = base + (size_t)(i0 + i) * 256;
      v[i] = idx < n16 ? w[idx] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
};

template <typename F>
void bench(const char* name, double bytes, int copies, F launch) {
  Timer t;
  std::vector<float> us;
  for (int i = 0; i < 10; ++i) launch(i % copies, nullptr, nullptr);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 200; ++i) {
    launch(i % copies, t.a, t.b);
    (void)hipEventSynchronize(t.b);
    float ms;
    (void)hipEventElapsedTime(&ms, t.a, t.b);
    us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  double avg = 0;
  for (float x : us) avg += x;
  avg /= us.size();
  printf("%-62s %s avg %7.2f med %7.2f min %7.2f us  %8.1f GB/s (avg)\n", name, copies > 1 ? "cold" : "warm", avg, us[us.size() / 2], us[0],
         bytes / avg * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t K_ROWS = argc > 2 ? (uint32_t)atoi(argv[1]) : 512, N = argc > 2 ? (uint32_t)atoi(argv[2]) : 11008;
  const size_t words = (size_t)K_ROWS * N;
  const double bytes = words * 4.0;
  const int copies = (int)((600e6 + bytes - 1) / bytes) + 1;
  printf("# %u packed rows x %u words = %.1f MB per launch, %d copies\n", K_ROWS, N, bytes * 1e-6, copies);
  uint32_t* w;
  uint32_t* out;
  (void)hipMalloc(&w, words * 4 * copies);
  (void)hipMalloc(&out, 4 << 20);
  (void)hipMemset(w, 1, words * 4 * copies);
  (void)hipDeviceSynchronize();

  auto ext = [](const void* f, dim3 g, dim3 b, void** args, hipEvent_t s, hipEvent_t e) {
    if (s) (void)hipExtLaunchKernel(f, g, b, args, 0, 0, s, e, 0);
    else (void)hipLaunchKernel(f, g, b, args, 0, 0);
  };
#define CASE(name, LPR, NW, MODE, XCD, NT, COLS)                                                              \
  {                                                                                                            \
    constexpr int R_ = MODE == 2 ? 4 : 64 / LPR;                                                               \
    constexpr int PER_ = MODE == 1 ? 64 : R_ * LPR;                                                            \
    (void)PER_;                                                                                                \
    {                                                                                                          \
      const uint32_t tiles = (N + COLS - 1) / COLS;                                                            \
      for (int cp : {copies, 1})                                                                               \
        bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                                        \
          const uint32_t* ptr = w + (size_t)c * words;                                                         \
          uint32_t nw = N, rows = K_ROWS;                                                                      \
          void* args[] = {&ptr, &out, &nw, &rows};                                                             \
          constexpr int LOADS_ = MODE == 2 ? 512 / (NW * 4) : (MODE == 1 ? (512 * LPR + 64 * NW - 1) / (64 * NW) : (512 + R_ * NW - 1) / (R_ * NW)); \
          ext((const void*)k_whole<LPR, NW, LOADS_, MODE, XCD, NT>, dim3(tiles), dim3(64 * NW), args, s, e);   \
        });                                                                                                    \
    }                                                                                                          \
  }
  if (K_ROWS != 512) {
    printf("this microbenchmark is sized for 512 packed rows\n");
    return 1;
  }
  for (int cp : {copies, 1})
    bench("linear 16B, 8 loads/thread, 256 thr/block", bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {
      const u32x4* ptr = (const u32x4*)(w + (size_t)c * words);
      uint32_t n16 = words / 4;
      int pp = 8;
      void* args[] = {&ptr, &out, &n16, &pp};
      ext((const void*)k_linear, dim3((n16 + 256 * 8 - 1) / (256 * 8)), dim3(256), args, s, e);
    });
  CASE("r01: 48-col, 16 lanes x 12B x 4 rows (768 B/instr) NW8 xcd", 16, 8, 2, true, false, 48)
  CASE("64-col, 16 lanes x 16B x 4 rows (1 KB/instr) NW8 xcd", 16, 8, 0, true, false, 64)
  CASE("48-col, 12 lanes x 16B x 5 rows (960 B/instr) NW8 xcd", 12, 8, 0, true, false, 48)
  CASE("48-col, 12 lanes x 16B x 5 rows (960 B/instr) NW8 no-xcd", 12, 8, 0, false, false, 48)
  CASE("48-col, 12 lanes x 16B x 5 rows (960 B/instr) NW8 xcd nt", 12, 8, 0, true, true, 48)
  CASE("48-col, 12 lanes x 16B x 5 rows NW16 xcd", 12, 16, 0, true, false, 48)
  CASE("48-col, 12 lanes x 16B x 5 rows NW4 xcd", 12, 4, 0, true, false, 48)
  CASE("48-col, chunks of 16B, 64 per instr (1 KB/instr) NW8 xcd", 12, 8, 1, true, false, 48)
  CASE("48-col, chunks of 16B, 64 per instr (1 KB/instr) NW16 xcd", 12, 16, 1, true, false, 48)
  CASE("44-col, 11 lanes x 16B x 5 rows (880 B/instr) NW8 xcd", 11, 8, 0, true, false, 44)
  CASE("44-col, chunks of 16B, 64 per instr NW8 xcd", 11, 8, 1, true, false, 44)
  CASE("40-col, 10 lanes x 16B x 6 rows (960 B/instr) NW8 xcd", 10, 8, 0, true, false, 40)
  CASE("32-col, 8 lanes x 16B x 8 rows (1 KB/instr) NW4 xcd", 8, 4, 0, true, false, 32)
  CASE("32-col, 8 lanes x 16B x 8 rows (1 KB/instr) NW8 xcd", 8, 8, 0, true, false, 32)
  CASE("56-col, 14 lanes x 16B x 4 rows NW8 xcd", 14, 8, 0, true, false, 56)
  return 0;
}
