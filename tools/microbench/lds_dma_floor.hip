// lds_dma_floor.hip — pure-read floor of a whole-K column tile streamed into LDS by dedicated loader
// waves (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction), read back by consumer waves.
//
// Why: in the register-path GEMV the wave that issues a load is the wave that computes on it.  The
// CU's vector-memory queue saturates, a wave blocks at its next load instruction and cannot touch the
// rows that have already arrived (profiles/r01/wave_timeline_upfront_loads.txt: "loads issued" as
// late as 3.8 us, last MFMA 2 us later).  With loader waves the block is harmless and the consumers
// chase the arrivals.  The whole tile fits in LDS (48 columns x 512 rows = 96 KB; 64 columns = 128 KB),
// so there is no ring and no flow control: loader L issues chunks L, L+NL, ... up front, then
// publishes "my first i chunks have landed" (counted vmcnt) in an LDS word; a consumer spins on the
// words of the chunks it needs.
//   MODE 0: bulk - loaders wait for everything, one barrier, then all waves read (floor of the DMA path)
//   MODE 1: flags - consumers chase arrivals chunk by chunk
//   MODE 2: flags + a dummy VALU load per chunk about the size of the 4-bit unpack (26 cycles x 4 words)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ uint32_t xcd_run(uint32_t b, uint32_t n) {
  const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_void*)p; }

// loader: after all CL loads are out, publish "my first i+1 chunks have landed" as each one lands
template <int CL, int I = 0>
__device__ __forceinline__ void publish_landed(uint32_t flag_addr) {
  if constexpr (I < CL) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CL - 1 - I) : "memory");
    const uint32_t val = I + 1;
    asm volatile("ds_write_b32 %0, %1" ::"v"(flag_addr), "v"(val) : "memory");
    publish_landed<CL, I + 1>(flag_addr);
  }
}

// LPR: 16-byte pieces per tile row (12 = 48 columns, 16 = 64 columns); ROWS packed rows; NL loader waves, NC consumer waves
template <int LPR, int ROWS, int NL, int NC, int MODE, bool NT>
__global__ void __launch_bounds__(64 * (NL + NC)) k_dma(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n_words) {
  constexpr int CHUNKS = ROWS * LPR / 64;  // 1 KB each
  constexpr int CL = CHUNKS / NL;          // per loader
  static_assert(CHUNKS % NL == 0 && CL <= 63, "chunks per loader");
  __shared__ __attribute__((aligned(16))) uint32_t s_w[CHUNKS * 256];
  __shared__ __attribute__((aligned(16))) uint32_t s_flag[8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t tile = xcd_run(blockIdx.x, gridDim.x);
  if (MODE != 0 && threadIdx.x < 8) s_flag[threadIdx.x] = 0;
  if (MODE != 0) __syncthreads();
  uint32_t acc = 0;
  if (wv < NL) {
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(w), 0, (int)((uint32_t)ROWS * n_words * 4u), 0x00020000);
    static_for<CL>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const uint32_t c = (uint32_t)(i * NL + wv);
      const uint32_t p = c * 64u + (uint32_t)lane, row = p / LPR, j = p % LPR;
      const uint32_t off = (row * n_words + tile * (LPR * 4u) + j * 4u) * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(s_w + c * 256u), 16, off, 0, 0, NT ? 2 : 0);
    });
    if constexpr (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      publish_landed<CL>(lds_addr(&s_flag[wv]));
    }
  }
  if constexpr (MODE == 0) {
    __syncthreads();
    for (int c = wv; c < CHUNKS; c += NL + NC) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(s_w + c * 256 + lane * 4);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  } else if (wv >= NL) {
    const int cw = wv - NL;
    for (int c = cw; c < CHUNKS; c += NC) {
      const uint32_t need = (uint32_t)(c / NL) + 1u;
      volatile uint32_t* f = &s_flag[c % NL];
      for (uint32_t spin = 0; *f < need && spin < (1u << 18); ++spin) __builtin_amdgcn_s_sleep(1);  // bounded: a lost DMA must not hang the GPU
      const u32x4 v = *reinterpret_cast<const u32x4*>(s_w + c * 256 + lane * 4);
      uint32_t x = v.x ^ v.y ^ v.z ^ v.w;
      if constexpr (MODE == 2) {
        // ~100 dependent-free VALU ops: the size of unpacking 4 words
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          x = (x >> 3) ^ (v.x & (0x0F0F0F0Fu << (t & 3)));
          x += (v.y >> t) & 0x03C003C0u;
          x ^= (v.z << (t & 7)) & 0x03C003C0u;
          x += (v.w >> (t + 1)) | 0x64006400u;
        }
      }
      acc ^= x;
    }
  }
  out[blockIdx.x * 64 * (NL + NC) + threadIdx.x] = acc;
}

// register-path reference (same tile, same chunking, all loads up front)
template <int LPR, int ROWS, int NW>
__global__ void __launch_bounds__(64 * NW) k_reg(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n_words) {
  constexpr int CHUNKS = ROWS * LPR / 64, CL = CHUNKS / NW;
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
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < CL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  out[blockIdx.x * 64 * NW + threadIdx.x] = acc;
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
  printf("%-60s %s avg %7.2f med %7.2f min %7.2f us  %8.1f GB/s (avg)\n", name, copies > 1 ? "cold" : "warm", avg, us[us.size() / 2], us[0],
         bytes / avg * 1e-3);
  fflush(stdout);
}

int main() {
  const uint32_t K_ROWS = 512, N = 11008;
  const size_t words = (size_t)K_ROWS * N;
  const double bytes = words * 4.0;
  const int copies = (int)((600e6 + bytes - 1) / bytes) + 1;
  printf("# %u packed rows x %u words = %.1f MB per launch, %d copies\n", K_ROWS, N, bytes * 1e-6, copies);
  uint32_t* w;
  uint32_t* out;
  (void)hipMalloc(&w, words * 4 * copies + (1 << 20));
  (void)hipMalloc(&out, 4 << 20);
  (void)hipMemset(w, 1, words * 4 * copies);
  (void)hipDeviceSynchronize();
  auto ext = [](const void* f, dim3 g, dim3 b, void** args, hipEvent_t s, hipEvent_t e) {
    hipError_t err = s ? hipExtLaunchKernel(f, g, b, args, 0, 0, s, e, 0) : hipLaunchKernel(f, g, b, args, 0, 0);
    if (err != hipSuccess) {
      printf("launch failed: %s\n", hipGetErrorString(err));
      exit(1);
    }
  };
#define DMA_CASE(name, LPR, NL, NC, MODE, NT)                                                             \
  for (int cp : {copies, 1})                                                                              \
    bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                                       \
      const uint32_t* ptr = w + (size_t)c * words;                                                        \
      uint32_t nw = N;                                                                                    \
      void* args[] = {&ptr, &out, &nw};                                                                   \
      ext((const void*)k_dma<LPR, 512, NL, NC, MODE, NT>, dim3(N / (LPR * 4) + (N % (LPR * 4) ? 1 : 0)), dim3(64 * (NL + NC)), args, s, e); \
    });
#define REG_CASE(name, LPR, NW)                                                                           \
  for (int cp : {copies, 1})                                                                              \
    bench(name, bytes, cp, [&](int c, hipEvent_t s, hipEvent_t e) {                                       \
      const uint32_t* ptr = w + (size_t)c * words;                                                        \
      uint32_t nw = N;                                                                                    \
      void* args[] = {&ptr, &out, &nw};                                                                   \
      ext((const void*)k_reg<LPR, 512, NW>, dim3(N / (LPR * 4) + (N % (LPR * 4) ? 1 : 0)), dim3(64 * NW), args, s, e); \
    });
  REG_CASE("registers: 48-col chunks, NW8, all loads up front", 12, 8)
  REG_CASE("registers: 64-col chunks, NW8, all loads up front", 16, 8)
  DMA_CASE("lds-dma bulk: 48-col, 4 loaders + 8 readers", 12, 4, 8, 0, false)
  DMA_CASE("lds-dma bulk: 48-col, 4 loaders + 8 readers, nt", 12, 4, 8, 0, true)
  DMA_CASE("lds-dma bulk: 48-col, 2 loaders + 8 readers", 12, 2, 8, 0, false)
  DMA_CASE("lds-dma bulk: 48-col, 8 loaders + 8 readers", 12, 8, 8, 0, false)
  DMA_CASE("lds-dma bulk: 64-col, 4 loaders + 8 readers", 16, 4, 8, 0, false)
  DMA_CASE("lds-dma flags: 48-col, 4 loaders + 8 consumers", 12, 4, 8, 1, false)
  DMA_CASE("lds-dma flags: 48-col, 4 loaders + 8 consumers, nt", 12, 4, 8, 1, true)
  DMA_CASE("lds-dma flags: 48-col, 2 loaders + 8 consumers", 12, 2, 8, 1, false)
  DMA_CASE("lds-dma flags: 48-col, 4 loaders + 4 consumers", 12, 4, 4, 1, false)
  DMA_CASE("lds-dma flags: 48-col, 4 loaders + 12 consumers", 12, 4, 12, 1, false)
  DMA_CASE("lds-dma flags: 64-col, 4 loaders + 8 consumers", 16, 4, 8, 1, false)
  DMA_CASE("lds-dma flags+valu: 48-col, 4 loaders + 8 consumers", 12, 4, 8, 2, false)
  DMA_CASE("lds-dma flags+valu: 48-col, 4 loaders + 12 consumers", 12, 4, 12, 2, false)
  DMA_CASE("lds-dma flags+valu: 48-col, 2 loaders + 8 consumers", 12, 2, 8, 2, false)
  DMA_CASE("lds-dma flags+valu: 64-col, 4 loaders + 8 consumers", 16, 4, 8, 2, false)
  return 0;
}
