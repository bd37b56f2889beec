// equal_work_floor.hip — the read floor of an EQUAL-WORK GEMV launch on MI355X, with none of the
// GEMV's arithmetic, next to the whole-K column-tile pattern of the round-1 kernel and a flat read.
//
// Question: whole-K column tiles either leave CUs idle (64-column tiles on
// 4096x11008: 172 tiles on 256 CUs) or read short, unaligned row segments (48-column tiles: 192 B).
// An equal-work launch cuts the (tile, k-step) space into W equal runs, one per workgroup, over
// ALIGNED tiles of 64/128/256 columns; every wave-instruction then reads 1 KB in 256/512/1024-byte
// row segments and every CU streams the same number of bytes.  The price is a cross-workgroup
// reduction for every tile; MODE 1 pays it with one returning 64-bit atomic per column and segment
// (the production split-K protocol), MODE 0 leaves it out (floor of the streaming part alone).
//
//   item  = (tile, step): one wave-instruction row group: (64/NL) k-slots x 4 rows x NL lanes x 16 B = 4 KB
//   workgroup w owns items { it : it*W/T == w }; wave v of it takes items first+v, first+v+NW, ...
//
// Timing: hipExtLaunchKernel start/stop events, cold (rotating over > 512 MB) and warm; run the
// binary under `rocprofv3 --kernel-trace --stats` for the profiler's durations of the same launches.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(uint32_t* out) {
  if (out == nullptr) out[0] = 1;
}

__device__ __forceinline__ uint32_t xcd_run(uint32_t b, uint32_t n) {  // blocks b, b+8, .. share an XCD: give each a contiguous run
  const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// whole-K column tiles (the round-1 pattern): NL lanes x W bytes along N, 64/NL k-slots, LOADS rows per lane
template <int W, int NL, int NW, int LOADS>
__global__ void __launch_bounds__(64 * NW) k_tile(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n_words, uint32_t rows) {
  constexpr int LK = 64 / NL, WW = W / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane / NL, nl = lane % NL;
  const uint32_t col = (xcd_run(blockIdx.x, gridDim.x) * NL + nl) * WW;
  uint32_t acc = 0;
  for (uint32_t r0 = (wv * LK + kl) * LOADS; r0 < rows; r0 += NW * LK * LOADS) {
    uint32_t v[LOADS][WW];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const uint32_t* src = w + (size_t)(r0 + i) * n_words + col;
      if constexpr (W == 12) {
        u32x3 t = *(const u32x3*)src;
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z;
      } else {
        u32x4 t = *(const u32x4*)src;
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      }
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
#pragma unroll
      for (int j = 0; j < WW; ++j) acc ^= v[i][j];
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

struct EwParams {
  const uint32_t* w;
  uint32_t* out;
  unsigned long long* acc;  // one 64-bit slot per column (zero between launches)
  uint32_t n_words, rows, tiles, steps_per_tile, items, wgs;
};

// MODE 0: no cross-workgroup reduction (every segment stores, results meaningless)
// MODE 1: one returning 64-bit atomic per column and segment, last arriver stores and clears
// MAXI: items per wave (loads of all of them are issued before the first is consumed)
// Epilogue as a real kernel would do it: a wave folds its items per tile segment in registers, the k-slots
// of a wave are folded by shuffles (the MFMA does that for free), one LDS row per (wave, segment), one
// barrier, then one thread per (segment, column) adds the waves' rows and does the atomic.
template <int NL, int NW, int MAXI, int MODE>
__global__ void __launch_bounds__(64 * NW) k_equal(const EwParams p) {
  constexpr int LK = 64 / NL, CN = NL * 4, MAXSEG = 3;
  __shared__ float s_red[NW][MAXSEG][CN];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane / NL, nl = lane % NL;
  const uint32_t wg = xcd_run(blockIdx.x, p.wgs);
  // items of this workgroup: { it : it * wgs / items == wg } = [ceil(wg*T/W), ceil((wg+1)*T/W))
  const uint32_t first = (uint32_t)(((uint64_t)wg * p.items + p.wgs - 1) / p.wgs);
  const uint32_t last = (uint32_t)(((uint64_t)(wg + 1) * p.items + p.wgs - 1) / p.wgs);
  if (first >= last) return;
  const uint32_t t_first = first / p.steps_per_tile, t_last = (last - 1) / p.steps_per_tile;
  // all loads first (buffer loads: an item past the end reads zeros through the range check, no branch)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.w), 0, (int)(p.rows * p.n_words * 4u), 0x00020000);
  u32x4 v[MAXI][4];
  uint32_t seg_of[MAXI];
  const uint32_t it0w = first + wv;
  uint32_t tile = it0w / p.steps_per_tile, step = it0w % p.steps_per_tile;  // one division per wave, then incremental
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
    const bool on = first + wv + j * NW < last;
    const uint32_t row0 = (step * LK + kl) * 4, col = tile * CN + nl * 4;
    seg_of[j] = on ? tile - t_first : 0xFFu;
    const uint32_t off = on ? (row0 * p.n_words + col) * 4u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, (uint32_t)i * p.n_words * 4u, 0);
    step += NW;
    if (step >= p.steps_per_tile) { step -= p.steps_per_tile; ++tile; }
  }
  float acc[MAXSEG][4];
#pragma unroll
  for (int sg = 0; sg < MAXSEG; ++sg)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[sg][c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = (float)((v[j][0][c] ^ v[j][1][c] ^ v[j][2][c] ^ v[j][3][c]) & 0xFFFFu);
#pragma unroll
      for (int sg = 0; sg < MAXSEG; ++sg) acc[sg][c] += seg_of[j] == (uint32_t)sg ? x : 0.f;
    }
  }
#pragma unroll
  for (int sg = 0; sg < MAXSEG; ++sg)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = acc[sg][c];
      for (int off = NL; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);  // fold the k-slots of the wave
      if (kl == 0) s_red[wv][sg][nl * 4 + c] = x;
    }
  __syncthreads();
  const uint32_t nseg = t_last - t_first + 1;
  for (uint32_t e = threadIdx.x; e < nseg * CN; e += 64 * NW) {
    const uint32_t sg = e / CN, c = e % CN, t = t_first + sg;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += s_red[w][sg][c];
    if constexpr (MODE == 0) {
      p.out[t * CN + c] = __builtin_bit_cast(uint32_t, sum);
    } else {
      // contributors of tile t: workgroups of its first and last item
      const uint32_t it0 = t * p.steps_per_tile, it1 = it0 + p.steps_per_tile - 1;
      const uint32_t w0 = (uint32_t)((uint64_t)it0 * p.wgs / p.items), w1 = (uint32_t)((uint64_t)it1 * p.wgs / p.items);
      const uint32_t expect = w1 - w0 + 1;
      unsigned long long* slot = p.acc + (size_t)t * CN + c;
      const unsigned long long add = (unsigned long long)(long long)(sum * 16.0f) + (1ull << 52);
      const unsigned long long old = __hip_atomic_fetch_add(slot, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (((old >> 52) & 0xFFFull) + 1ull == expect) {
        p.out[t * CN + c] = (uint32_t)(old + add);
        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
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
  printf("%-58s %s avg %7.2f med %7.2f min %7.2f us  %8.1f GB/s (avg)\n", name, copies > 1 ? "cold" : "warm", avg, us[us.size() / 2], us[0],
         bytes / avg * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t K_ROWS = argc > 2 ? (uint32_t)atoi(argv[1]) : 512, N = argc > 2 ? (uint32_t)atoi(argv[2]) : 11008;
  const size_t words = (size_t)K_ROWS * N;
  const double bytes = words * 4.0;
  const int copies = (int)((600e6 + bytes - 1) / bytes) + 1;  // > 512 MB
  printf("# %u packed rows x %u words = %.1f MB per launch, %d copies\n", K_ROWS, N, bytes * 1e-6, copies);
  uint32_t* w;
  uint32_t* out;
  unsigned long long* acc;
  hipMalloc(&w, words * 4 * copies);
  hipMalloc(&out, 4 << 20);
  hipMalloc(&acc, 1 << 20);
  hipMemset(w, 1, words * 4 * copies);
  hipMemset(acc, 0, 1 << 20);
  hipDeviceSynchronize();

  auto ext = [](const void* f, dim3 g, dim3 b, void** args, hipEvent_t s, hipEvent_t e) {
    if (s) hipExtLaunchKernel(f, g, b, args, 0, 0, s, e, 0);
    else hipLaunchKernel(f, g, b, args, 0, 0);
  };
  for (int cp : {1})
    bench("empty 256 x 512", 0, cp, [&](int, hipEvent_t s, hipEvent_t e) {
      void* args[] = {&out};
      ext((const void*)k_empty, dim3(256), dim3(512), args, s, e);
    });
#define TILE_CASE(name, W, NL, NW, LOADS)                                                            \
  if (N % (NL * (W / 4)) == 0 && K_ROWS % (NW * (64 / NL) * LOADS) == 0)                              \
    for (int cp : {copies, 1})                                                                        \
      bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                                 \
        const uint32_t* ptr = w + (size_t)c * words;                                                  \
        uint32_t nw = N, rows = K_ROWS;                                                               \
        void* args[] = {&ptr, &out, &nw, &rows};                                                      \
        ext((const void*)k_tile<W, NL, NW, LOADS>, dim3(N / (NL * (W / 4))), dim3(64 * NW), args, s, e); \
      });
#define EQ_CASE(name, NL, NW, MAXI, MODE, WGS)                                                        \
  if (N % (NL * 4) == 0 && K_ROWS % ((64 / NL) * 4) == 0) {                                           \
    EwParams p;                                                                                       \
    p.out = out; p.acc = acc; p.n_words = N; p.rows = K_ROWS;                                         \
    p.tiles = N / (NL * 4); p.steps_per_tile = K_ROWS / ((64 / NL) * 4);                              \
    p.items = p.tiles * p.steps_per_tile; p.wgs = WGS;                                                \
    const uint32_t per = (p.items + p.wgs - 1) / p.wgs;                                               \
    if (per <= (uint32_t)(NW * MAXI)) {                                                               \
      for (int cp : {copies, 1})                                                                      \
        bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                               \
          p.w = w + (size_t)c * words;                                                                \
          void* args[] = {&p};                                                                        \
          ext((const void*)k_equal<NL, NW, MAXI, MODE>, dim3(WGS), dim3(64 * NW), args, s, e);        \
        });                                                                                           \
    } else {                                                                                          \
      printf("%-58s skipped: %u items per workgroup > %d\n", name, per, NW * MAXI);                   \
    }                                                                                                 \
  }

  TILE_CASE("whole-K 48-col tiles (12B x16) NW8 16 loads [r01 pattern]", 12, 16, 8, 16)
  TILE_CASE("whole-K 64-col tiles (16B x16) NW8 16 loads", 16, 16, 8, 16)
  for (int per : {8})
    for (int cp : {copies, 1}) {
      char nm[96];
      snprintf(nm, sizeof nm, "linear 16B, %d loads/thread, 256 thr/block", per);
      bench(nm, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {
        const u32x4* ptr = (const u32x4*)(w + (size_t)c * words);
        uint32_t n16 = words / 4;
        int pp = per;
        void* args[] = {&ptr, &out, &n16, &pp};
        ext((const void*)k_linear, dim3((n16 + 256 * per - 1) / (256 * per)), dim3(256), args, s, e);
      });
    }
  // equal work over 256 workgroups (one per CU)
  EQ_CASE("equal 64-col  NW8  x3 items, 256 wgs, no reduction", 16, 8, 3, 0, 256)
  EQ_CASE("equal 64-col  NW8  x3 items, 256 wgs, atomic", 16, 8, 3, 1, 256)
  EQ_CASE("equal 128-col NW8  x3 items, 256 wgs, no reduction", 32, 8, 3, 0, 256)
  EQ_CASE("equal 128-col NW8  x3 items, 256 wgs, atomic", 32, 8, 3, 1, 256)
  EQ_CASE("equal 256-col NW8  x3 items, 256 wgs, no reduction", 64, 8, 3, 0, 256)
  EQ_CASE("equal 256-col NW8  x3 items, 256 wgs, atomic", 64, 8, 3, 1, 256)
  EQ_CASE("equal 64-col  NW16 x2 items, 256 wgs, no reduction", 16, 16, 2, 0, 256)
  EQ_CASE("equal 64-col  NW16 x2 items, 256 wgs, atomic", 16, 16, 2, 1, 256)
  EQ_CASE("equal 128-col NW16 x2 items, 256 wgs, atomic", 32, 16, 2, 1, 256)
  EQ_CASE("equal 64-col  NW12 x2 items, 256 wgs, atomic", 16, 12, 2, 1, 256)
  // two / four smaller workgroups per CU
  EQ_CASE("equal 64-col  NW4  x3 items, 512 wgs, no reduction", 16, 4, 3, 0, 512)
  EQ_CASE("equal 64-col  NW4  x3 items, 512 wgs, atomic", 16, 4, 3, 1, 512)
  EQ_CASE("equal 128-col NW4  x3 items, 512 wgs, atomic", 32, 4, 3, 1, 512)
  EQ_CASE("equal 64-col  NW4  x2 items, 768 wgs, atomic", 16, 4, 2, 1, 768)
  EQ_CASE("equal 64-col  NW2  x3 items, 1024 wgs, atomic", 16, 2, 3, 1, 1024)
  return 0;
}
