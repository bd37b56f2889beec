// xbit_dequant.hip — packed 2..8-bit weights -> fp16 W[K,N] for MI355X (gfx950).
//
// Replaces DequantizeAndUnpackWeight248 / DequantizeAndUnpackWeight3567_v2 and their launchers
// (/root/reference/src/cu/unpack_weight_2_to_7.cu:44-85, 219-330, 353-441).  One kernel template
// covers every bit-width because the k-block (32 k = BITS packed rows) is the unit for all of
// them; the reference's 6-bit row-stride quirk is not reproduced — the
// layout is the contiguous bit-stream that its own simulator and its 3/5/7-bit path use.
//
// The op is write-bound (2*K*N bytes out vs ~b/8*K*N in).  A lane owns 8 adjacent columns of one
// k-block: 2*BITS 16-byte loads in, 32 16-byte stores out; 16 lanes side by side write 256
// contiguous bytes of every output row.  The arithmetic is the reference's, bit for bit:
//   zs = hmul(half(z + bias), s);  W = hfma(half(q), s, -zs)      (:60,75 and :285,305)
// packed over column pairs (c, c+1) so that each result is already the dword to store.
#include <stdlib.h>

#include "xbit_device.h"
#include "xbit_host.h"

namespace xbit {

namespace {

constexpr int kDqLanesN = 64;  // lanes along N (x 8 columns = 512 columns: a wave stores 1 KB of one output row)
constexpr int kDqLanesK = 4;   // k-blocks per workgroup (one per wave)
constexpr int kDqThreads = kDqLanesN * kDqLanesK;

// Packed weights are read with PLAIN loads, W is written with nontemporal stores: measured on MI355X
// (profiles/r02/sweep_dequant_variants.txt) nontemporal weight loads cost 7-8 % (4096x11008 w4: 20.5 us
// against 19.05), plain stores cost 2-15 %.
// VAR (measurement builds, XBIT_DEQUANT_VARIANT): 0 = production; 1 = plain (not nontemporal) stores;
// 2 = the 4 waves of a workgroup side by side along N (one k-block, 2048 columns: 4 KB per output row);
// 3 = nontemporal packed-weight loads (the round-1 kernel); 5 / 6 / 7 = 8 / 2 / 1 waves per workgroup
// BY_ID (act-order, xbit_dequant_gidx_f16): the group of row k is g_idx[k] (clamped to [0, groups)); a lane re-reads its 8
// scales and zero-points whenever the group changes from one row to the next - every row for a shuffled g_idx, which makes
// the op read about as many bytes from L2 as it writes (4096x11008 w4: 17.7 us contiguous, profiles/r03/gidx_dequant.txt).
template <int BITS, bool GROUP_PER_BLOCK, int VAR = 0, int WAVES = kDqLanesK, bool BY_ID = false>
__global__ void __launch_bounds__(kDqLanesN * WAVES) dequant_tile_kernel(const DequantParams p, const int32_t* g_idx) {
  static_assert(!BY_ID || !GROUP_PER_BLOCK, "act-order: the group may change at every row");
  const int nl = threadIdx.x % kDqLanesN, kl = threadIdx.x / kDqLanesN;
  const uint32_t col = VAR == 2 ? ((blockIdx.x * WAVES + kl) * kDqLanesN + nl) * 8u : (blockIdx.x * kDqLanesN + nl) * 8u;
  const uint32_t kb = VAR == 2 ? blockIdx.y : blockIdx.y * WAVES + kl;
  if (col >= p.n || kb >= p.kblocks) return;  // n % 8 == 0 on this path
  const uint32_t k0 = kb * 32u;

  // packed rows of this k-block: BITS rows x 8 columns
  uint4 wlo[BITS], whi[BITS];
#pragma unroll
  for (int i = 0; i < BITS; ++i) {
    const uint32_t row = kb * BITS + i;
    if (row < p.kw) {
      const uint32_t* src = p.qweight + (size_t)row * p.n + col;
      wlo[i] = VAR == 3 ? ld16_nt(src) : ld16(src);
      whi[i] = VAR == 3 ? ld16_nt(src + 4) : ld16(src + 4);
    } else {
      wlo[i] = whi[i] = make_uint4(0, 0, 0, 0);
    }
  }

  constexpr uint32_t mask = (1u << BITS) - 1u;
  half2_t s2[4], nzs2[4];  // column pairs (0,1) (2,3) (4,5) (6,7)
  auto load_group = [&](uint32_t gi) {
    const uint4 sraw = *reinterpret_cast<const uint4*>(p.scales + (size_t)gi * p.n + col);
    const uint32_t* zrow = p.qzeros + (size_t)gi * p.zw;
    const uint32_t z03 = zero_window4<BITS>(zrow, col, p.zw);
    const uint32_t z47 = zero_window4<BITS>(zrow, col + 4, p.zw);
    const uint32_t sr[4] = {sraw.x, sraw.y, sraw.z, sraw.w};
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      const uint32_t zwin = c2 < 2 ? z03 : z47;
      const int c = (2 * c2) & 3;
      const uint32_t za = ((zwin >> (c * BITS)) & mask) + p.bias;
      const uint32_t zb = ((zwin >> ((c + 1) * BITS)) & mask) + p.bias;
      const _Float16 sa = as_h((uint16_t)(sr[c2] & 0xFFFFu)), sb = as_h((uint16_t)(sr[c2] >> 16));
      s2[c2] = half2_t{sa, sb};
      nzs2[c2] = half2_t{neg_zero_times_scale(za, sa), neg_zero_times_scale(zb, sb)};
    }
  };

  auto group_of = [&](uint32_t k) -> uint32_t {
    if constexpr (BY_ID) {
      const uint32_t g = k < p.k ? (uint32_t)g_idx[k] : 0u;
      return g < p.groups ? g : p.groups - 1u;  // out-of-range (or negative) ids are clamped, never followed
    } else {
      return k / p.groupsize;
    }
  };
  uint32_t cur_g = group_of(k0);
  load_group(cur_g);

  uint32_t wcol[8][BITS];
#pragma unroll
  for (int i = 0; i < BITS; ++i) {
    wcol[0][i] = wlo[i].x; wcol[1][i] = wlo[i].y; wcol[2][i] = wlo[i].z; wcol[3][i] = wlo[i].w;
    wcol[4][i] = whi[i].x; wcol[5][i] = whi[i].y; wcol[6][i] = whi[i].z; wcol[7][i] = whi[i].w;
  }

  uint16_t* orow = p.out + (size_t)k0 * p.n + col;
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) {
    const bool row_ok = k0 + kk < p.k;  // ragged K: only rows < in_features exist (:77-84, :316-329)
    if constexpr (!GROUP_PER_BLOCK) {
      const uint32_t g = group_of(k0 + kk);
      if (row_ok && g != cur_g) {
        cur_g = g;
        load_group(g);
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      const uint32_t Q = field<BITS>(wcol[2 * c2], kk) | (field<BITS>(wcol[2 * c2 + 1], kk) << 16) | kMagic1024x2;
      o[c2] = as_u32(dequant_pair(Q, s2[c2], nzs2[c2]));
    }
    if (row_ok) {
      if constexpr (VAR == 1) *reinterpret_cast<u32x4_t*>(orow + (size_t)kk * p.n) = u32x4_t{o[0], o[1], o[2], o[3]};
      else st16_nt(orow + (size_t)kk * p.n, make_uint4(o[0], o[1], o[2], o[3]));
    }
  }
}

// General fallback: one thread per output element; any even n, any alignment.
__global__ void __launch_bounds__(256) dequant_generic_kernel(const DequantParams p, int bits) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (idx >= (uint64_t)p.k * p.n) return;
  const uint32_t k = (uint32_t)(idx / p.n), col = (uint32_t)(idx % p.n);
  const uint32_t g = k / p.groupsize;
  const _Float16 s = as_h(p.scales[(size_t)g * p.n + col]);
  const uint32_t z = stream_field(p.qzeros + (size_t)g * p.zw, 1, col, bits, p.zw) + p.bias;
  const uint32_t q = stream_field(p.qweight + col, p.n, k, bits, p.kw);
  const _Float16 w = __builtin_elementwise_fma((_Float16)(float)(int)q, s, neg_zero_times_scale(z, s));
  p.out[idx] = as_u16(w);
}

template <int BITS>
int launch_bits(const DequantParams& p, hipStream_t stream) {
  dim3 grid((p.n / 8 + kDqLanesN - 1) / kDqLanesN, (p.kblocks + kDqLanesK - 1) / kDqLanesK), block(kDqThreads);
#if XBIT_TUNING
  static const int var = getenv("XBIT_DEQUANT_VARIANT") ? atoi(getenv("XBIT_DEQUANT_VARIANT")) : 0;
  if (p.groupsize % 32u == 0 && var == 1) return launch(dequant_tile_kernel<BITS, true, 1>, grid, block, stream, p, static_cast<const int32_t*>(nullptr));
  if (p.groupsize % 32u == 0 && var == 3) return launch(dequant_tile_kernel<BITS, true, 3>, grid, block, stream, p, static_cast<const int32_t*>(nullptr));
  if (p.groupsize % 32u == 0 && (var == 5 || var == 6 || var == 7)) {  // 8 / 2 / 1 waves (k-blocks) per workgroup
    const uint32_t wv = var == 5 ? 8u : var == 6 ? 2u : 1u;
    dim3 gridw((p.n / 8 + kDqLanesN - 1) / kDqLanesN, (p.kblocks + wv - 1) / wv), blockw(kDqLanesN * wv);
    if (var == 5) return launch(dequant_tile_kernel<BITS, true, 0, 8>, gridw, blockw, stream, p, static_cast<const int32_t*>(nullptr));
    if (var == 6) return launch(dequant_tile_kernel<BITS, true, 0, 2>, gridw, blockw, stream, p, static_cast<const int32_t*>(nullptr));
    return launch(dequant_tile_kernel<BITS, true, 0, 1>, gridw, blockw, stream, p, static_cast<const int32_t*>(nullptr));
  }
  if (p.groupsize % 32u == 0 && var == 2) {
    dim3 grid2((p.n / 8 + kDqLanesN * kDqLanesK - 1) / (kDqLanesN * kDqLanesK), p.kblocks);
    return launch(dequant_tile_kernel<BITS, true, 2>, grid2, block, stream, p, static_cast<const int32_t*>(nullptr));
  }
#endif
  if (p.groupsize % 32u == 0) return launch(dequant_tile_kernel<BITS, true>, grid, block, stream, p, static_cast<const int32_t*>(nullptr));
  return launch(dequant_tile_kernel<BITS, false>, grid, block, stream, p, static_cast<const int32_t*>(nullptr));
}

}  // namespace

// act-order: the tile kernel with per-row group ids; the caller has checked the tile kernel's alignment rules
int launch_dequant_by_id(const DequantParams& p, const int32_t* g_idx, int bits, hipStream_t stream) {
  dim3 grid((p.n / 8 + kDqLanesN - 1) / kDqLanesN, (p.kblocks + kDqLanesK - 1) / kDqLanesK), block(kDqThreads);
  switch (bits) {
    case 2: return launch(dequant_tile_kernel<2, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    case 3: return launch(dequant_tile_kernel<3, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    case 4: return launch(dequant_tile_kernel<4, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    case 5: return launch(dequant_tile_kernel<5, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    case 6: return launch(dequant_tile_kernel<6, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    case 7: return launch(dequant_tile_kernel<7, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    case 8: return launch(dequant_tile_kernel<8, false, 0, kDqLanesK, true>, grid, block, stream, p, g_idx);
    default: return XBIT_ERR_BITS;
  }
}

int launch_dequant(const DequantParams& p, int bits, hipStream_t stream) {
  const bool aligned = ((uintptr_t)p.qweight % 16 == 0) && ((uintptr_t)p.scales % 16 == 0) && ((uintptr_t)p.out % 16 == 0);
  const uint32_t grid_y = (p.kblocks + kDqLanesK - 1) / kDqLanesK;
  if (p.n % 8u != 0 || !aligned || grid_y > 65535u) {
    const uint64_t total = (uint64_t)p.k * p.n;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    return launch(dequant_generic_kernel, grid, block, stream, p, bits);
  }
  switch (bits) {
    case 2: return launch_bits<2>(p, stream);
    case 3: return launch_bits<3>(p, stream);
    case 4: return launch_bits<4>(p, stream);
    case 5: return launch_bits<5>(p, stream);
    case 6: return launch_bits<6>(p, stream);
    case 7: return launch_bits<7>(p, stream);
    case 8: return launch_bits<8>(p, stream);
    default: return XBIT_ERR_BITS;
  }
}

}  // namespace xbit
