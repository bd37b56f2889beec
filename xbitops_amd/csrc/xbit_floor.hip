// xbit_floor.hip — pure-read floors for bench.py's roofline record (MEASUREMENT BUILD ONLY: compiled into
// libxbitops_hip_tuning.so, never into the production library).  A kernel here reads the packed weight rows of one GEMV
// call in a given access shape and does nothing with them (xor, one store per thread): what the memory system charges
// for the bytes alone, through the same launch path and on the same buffers as the GEMV next to which it is timed.
//   pattern 0           flat: every thread reads 16-byte pieces of the buffer, side by side (the friendliest stream there is)
//   pattern 32 / 48 / 64 / 128   whole-K column tiles of that width exactly as the GEMV kernels walk them: 16 lanes x 8 / 12 /
//                       16 B x 4 k-slots per instruction (128: 32 lanes x 16 B x 2 k-slots), a lane owning runs of 4 packed
//                       rows, workgroups in XCD-contiguous tile order, `tcols` columns owned per tile, every load up front
//                       in batches of 16 per lane
// Declared in include/xbitops_tuning.h.
#include "xbit_device.h"
#include "xbit_host.h"
#include "../../include/xbitops_tuning.h"

namespace xbit {
namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
constexpr uint32_t kOob = 0x80000000u;

template <int CPL, int NL, int NW>
__global__ void __launch_bounds__(64 * NW) floor_tiles_kernel(const uint32_t* __restrict__ w, uint32_t* out, uint32_t n, uint32_t rows, uint32_t tiles, uint32_t tcols) {
  constexpr int LK = 64 / NL;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kl = lane / NL, nl = lane % NL;
  const uint32_t tile = xcd_tile(blockIdx.x, tiles);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(w), 0, (int)(rows * n * 4u), 0x00020000);
  const uint32_t col = tile * tcols + (uint32_t)(nl * CPL);
  const bool col_ok = (uint32_t)(nl * CPL) < tcols && col < n;
  uint32_t acc = 0;
  // a lane owns runs of 4 consecutive packed rows (a 4-bit k-block); a wave-instruction covers LK runs; 16 loads = 4 runs per lane and batch
  for (uint32_t run0 = (uint32_t)(wv * LK + kl); run0 * 4u < rows; run0 += (uint32_t)(NW * LK * 4)) {
    uint32_t v[16][CPL];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t row = (run0 + (uint32_t)((i >> 2) * NW * LK)) * 4u + (uint32_t)(i & 3);
      const uint32_t off = (col_ok && row < rows) ? (row * n + col) * 4u : kOob;
      if constexpr (CPL == 2) {
        const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 2);
        v[i][0] = t.x; v[i][1] = t.y;
      } else if constexpr (CPL == 3) {
        const u32x3_t t = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 2);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z;
      } else {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc ^= v[i][c];
  }
  out[(blockIdx.x * (uint32_t)(64 * NW) + threadIdx.x) & 0xFFFFFu] = acc;
}

__global__ void __launch_bounds__(256) floor_flat_kernel(const u32x4_t* __restrict__ w, uint32_t* out, uint32_t n16) {
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  uint32_t acc = 0;
  u32x4_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t idx = base + (size_t)i * 256;
    v[i] = idx < n16 ? __builtin_nontemporal_load(&w[idx]) : u32x4_t{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  out[(blockIdx.x * 256u + threadIdx.x) & 0xFFFFFu] = acc;
}

uint32_t* floor_sink() {  // 4 MiB of device memory the floors store their xor into; lives as long as the process
  static uint32_t* sink = nullptr;
  if (sink == nullptr && hipMalloc(reinterpret_cast<void**>(&sink), (size_t)4 << 20) != hipSuccess) {
    (void)hipGetLastError();
    sink = nullptr;
  }
  return sink;
}

}  // namespace
}  // namespace xbit

extern "C" int xbit_tuning_read_floor(const void* qweight, uint32_t packed_rows, uint32_t n, int pattern, uint32_t tcols, void* hip_stream) {
  using namespace xbit;
  if (!qweight) return XBIT_ERR_NULL_POINTER;
  if (packed_rows == 0 || n == 0 || (uint64_t)packed_rows * n * 4u >= (1ull << 31) || ((uintptr_t)qweight & 15u)) return XBIT_ERR_SHAPE;
  uint32_t* out = floor_sink();
  if (!out) return (int)hipErrorOutOfMemory;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const uint32_t* w = static_cast<const uint32_t*>(qweight);
  if (pattern == 0) {
    const u32x4_t* w16 = static_cast<const u32x4_t*>(qweight);
    uint32_t n16 = (uint32_t)(((uint64_t)packed_rows * n) / 4u);
    return launch(floor_flat_kernel, dim3((n16 + 2047u) / 2048u), dim3(256), stream, w16, out, n16);
  }
  const uint32_t cn = (uint32_t)pattern;
  if (tcols == 0) tcols = cn;
  if (tcols > cn) return XBIT_ERR_SHAPE;
  const uint32_t tiles = (n + tcols - 1u) / tcols;
  switch (pattern) {
    case 32: return launch(floor_tiles_kernel<2, 16, 8>, dim3(tiles), dim3(512), stream, w, out, n, packed_rows, tiles, tcols);
    case 48: return launch(floor_tiles_kernel<3, 16, 8>, dim3(tiles), dim3(512), stream, w, out, n, packed_rows, tiles, tcols);
    case 64: return launch(floor_tiles_kernel<4, 16, 8>, dim3(tiles), dim3(512), stream, w, out, n, packed_rows, tiles, tcols);
    case 128: return launch(floor_tiles_kernel<4, 32, 16>, dim3(tiles), dim3(1024), stream, w, out, n, packed_rows, tiles, tcols);
    default: return XBIT_ERR_SHAPE;
  }
}
