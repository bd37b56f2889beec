// line_floor.hip — pure-read floors that test one hypothesis about the 4096x11008 GEMV: a CU's time is set by the number
// of 128-byte LINES its vector L1 has to request (about one per ten clocks), not by bytes or by load instructions.
// 48-column whole-K tiles ask for 2 lines per packed row and use 1.5 of them on 230 of 256 CUs (1024 line requests per
// CU); 64-column tiles use both lines but only 172 CUs (1024 per CU again); a flat read needs 704 per CU.
//   P+S   every workgroup reads one 32-column tile (ONE aligned line per row) over all of K plus an equal share of the
//         11 tiles per XCD that are left over (K split): 688 line requests per CU on all 256 CUs, XCD stripes contiguous.
// No arithmetic: loads, xor, one store.  Clocks: per-launch events (isolated launches) and the period of 400 launches
// queued back to back; cold (rotating over > 512 MB) unless the name says warm.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/line_floor.hip -o tools/microbench/line_floor
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t kOob = 0x80000000u;

__device__ __forceinline__ uint32_t xcd_run(uint32_t b, uint32_t n) {
  const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// whole-K tiles of 16 * CPL columns: 16 lanes x CPL words x 4 k-slots per instruction, 8 waves, 16 loads per wave (512 rows)
template <int CPL, bool NT>
__global__ void __launch_bounds__(512) k_whole(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n, uint32_t rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane >> 4, nl = lane & 15;
  const uint32_t tile = xcd_run(blockIdx.x, gridDim.x);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(w), 0, (int)(rows * n * 4u), 0x00020000);
  const uint32_t col = tile * (16 * CPL) + nl * CPL;
  uint32_t acc = 0;
  uint32_t v[16][CPL];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t row = (uint32_t)((wv * 4 + kl) * 16 + i);
    const uint32_t off = (col < n && row < rows) ? (row * n + col) * 4u : kOob;
    if constexpr (CPL == 3) {
      const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, NT ? 2 : 0);
      v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z;
    } else {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, NT ? 2 : 0);
      v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc ^= v[i][c];
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

// P+S: 256 workgroups x 8 waves; lane = (k-block kl of the unit: lane >> 3, 16-byte chunk of the 128-byte line: lane & 7);
// a unit = 8 k-blocks = 32 packed rows of one 32-column tile = 4 instructions of 1 KB.  Needs n = 8 * cx * 32 columns with
// 32 <= cx <= 64 (4096x11008: cx = 43) and rows = 512 (16 units per tile).
template <bool NT>
__global__ void __launch_bounds__(512) k_ps(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n, uint32_t rows, uint32_t cx) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane >> 3, nl = lane & 7;
  const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;  // XCD (as dealt), workgroup within it: 0..31
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(w), 0, (int)(rows * n * 4u), 0x00020000);
  const uint32_t units_per_tile = rows / 32u;             // 16
  const uint32_t sec_tiles = cx - 32u, sec_units = sec_tiles * units_per_tile;
  const uint32_t u0 = (sec_units * i) / 32u, u1 = (sec_units * (i + 1u)) / 32u;  // this workgroup's share of the left-over tiles
  uint32_t acc = 0;
  u32x4 v[12];
  auto unit_loads = [&](int slot, uint32_t linecol, uint32_t unit, bool on) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const uint32_t row = (unit * 8u + (uint32_t)kl) * 4u + (uint32_t)rr;
      const uint32_t off = on ? (row * n + linecol * 32u + (uint32_t)nl * 4u) * 4u : kOob;
      v[slot * 4 + rr] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, NT ? 2 : 0);
    }
  };
  // left-over share first (its partial sums would have the longest way to go), then the own tile
  const uint32_t su = u0 + (uint32_t)wv;
  unit_loads(0, x * cx + 32u + su / units_per_tile, su % units_per_tile, su < u1);
  unit_loads(1, x * cx + i, (uint32_t)wv, true);
  unit_loads(2, x * cx + i, (uint32_t)wv + 8u, true);
#pragma unroll
  for (int j = 0; j < 12; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_linear(const u32x4* __restrict__ w, uint32_t* out, uint32_t n16) {
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  uint32_t acc = 0;
  u32x4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t idx = base + (size_t)i * 256;
    v[i] = idx < n16 ? w[idx] : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static void bench(const char* name, double bytes, int copies, const std::function<void(int, hipEvent_t, hipEvent_t)>& launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 2 * copies; ++i) launch(i % copies, nullptr, nullptr);
  (void)hipDeviceSynchronize();
  std::vector<float> us;
  for (int i = 0; i < 200; ++i) {
    launch((i * 7 + 1) % copies, a, b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  std::vector<float> chain;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < 400; ++i) launch((i + 3 * rep) % copies, nullptr, nullptr);
    (void)hipEventRecord(b, nullptr);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    chain.push_back(ms * 1e3f / 400.f);
  }
  std::sort(chain.begin(), chain.end());
  printf("%-58s events med %6.2f min %6.2f us | chain med %6.2f min %6.2f us/launch = %6.1f GB/s\n", name, us[us.size() / 2], us[0], chain[2], chain[0],
         bytes / chain[2] * 1e-3);
  fflush(stdout);
}

int main() {
  const uint32_t rows = 512, n = 11008;
  const size_t words = (size_t)rows * n;
  const double bytes = words * 4.0;
  const int copies = (int)((600e6 + bytes - 1) / bytes) + 1;
  printf("# %u packed rows x %u words = %.1f MB per launch, %d copies\n", rows, n, bytes * 1e-6, copies);
  uint32_t *w, *out;
  (void)hipMalloc(&w, words * 4 * copies);
  (void)hipMalloc(&out, 4 << 20);
  (void)hipMemset(w, 1, words * 4 * copies);
  (void)hipDeviceSynchronize();
  auto ext = [](const void* f, dim3 g, dim3 b, void** args, hipEvent_t s, hipEvent_t e) {
    if (s) (void)hipExtLaunchKernel(f, g, b, args, 0, 0, s, e, 0);
    else (void)hipLaunchKernel(f, g, b, args, 0, 0);
  };
  uint32_t nn = n, rr = rows, cx = n / 32 / 8;
  for (int pass = 0; pass < 2; ++pass) {
    bench("flat 16-byte read, 8 loads per thread", bytes, copies, [&](int c, hipEvent_t s, hipEvent_t e) {
      const u32x4* ptr = (const u32x4*)(w + (size_t)c * words);
      uint32_t n16 = (uint32_t)(words / 4);
      void* args[] = {&ptr, &out, &n16};
      ext((const void*)k_linear, dim3((n16 + 2047) / 2048), dim3(256), args, s, e);
    });
    bench("whole-K 48-column tiles, 230 workgroups (1024 lines per CU)", bytes, copies, [&](int c, hipEvent_t s, hipEvent_t e) {
      const uint32_t* ptr = w + (size_t)c * words;
      void* args[] = {&ptr, &out, &nn, &rr};
      ext((const void*)k_whole<3, false>, dim3((n + 47) / 48), dim3(512), args, s, e);
    });
    bench("whole-K 48-column tiles, nontemporal", bytes, copies, [&](int c, hipEvent_t s, hipEvent_t e) {
      const uint32_t* ptr = w + (size_t)c * words;
      void* args[] = {&ptr, &out, &nn, &rr};
      ext((const void*)k_whole<3, true>, dim3((n + 47) / 48), dim3(512), args, s, e);
    });
    bench("whole-K 64-column tiles, 172 workgroups (1024 lines per CU)", bytes, copies, [&](int c, hipEvent_t s, hipEvent_t e) {
      const uint32_t* ptr = w + (size_t)c * words;
      void* args[] = {&ptr, &out, &nn, &rr};
      ext((const void*)k_whole<4, false>, dim3((n + 63) / 64), dim3(512), args, s, e);
    });
    bench("P+S: 32-column tile + share of the left-over tiles (688 lines)", bytes, copies, [&](int c, hipEvent_t s, hipEvent_t e) {
      const uint32_t* ptr = w + (size_t)c * words;
      void* args[] = {&ptr, &out, &nn, &rr, &cx};
      ext((const void*)k_ps<false>, dim3(256), dim3(512), args, s, e);
    });
    bench("P+S, nontemporal", bytes, copies, [&](int c, hipEvent_t s, hipEvent_t e) {
      const uint32_t* ptr = w + (size_t)c * words;
      void* args[] = {&ptr, &out, &nn, &rr, &cx};
      ext((const void*)k_ps<true>, dim3(256), dim3(512), args, s, e);
    });
  }
  return 0;
}
