// xbit_gidx.hip — act-order ("desc_act") checkpoints: the group of row k is g_idx[k], not k / groupsize.
//
// Counterpart of the reference's VecQuant4MatMulKernel_G (/root/reference/src/cu/quant_cuda_kernel.cu:
// 257-336, compiled out there by `#if 0` at :220), generalised from 4-bit / "+1" to bits 2..8 and
// add_zero_bias, with the arithmetic of the two registered ops:
//     zs = hmul(half(z[g,n] + bias), s[g,n]);  W[k,n] = hfma(half(q[k,n]), s[g,n], -zs),  g = g_idx[k]
// (the reference forms the same single-rounded product-sum, :299-300 and :306-309) and fp32
// accumulation (the reference accumulates in fp16 and adds with an fp16 atomic, :306-315).
//
// This is the GENERAL entry: any g_idx.  A checkpoint whose groups all have `groupsize` rows — what
// GPTQ act-order produces — is better served by sorting the rows by group once at load time
// (xbitops_amd.packing.sort_rows_by_group) and running the ordinary kernels on the sorted weight with
// the activation permuted the same way (QuantLinear does that); this kernel is for everything else.
//
// gemv: a workgroup of 4 waves owns 64 columns over all of K.  Lane (slot, nl): nl = 16 lanes x 4
// adjacent columns (one 16-byte load per packed row), slot = 16 k-slots taking k-blocks slot, slot+16, ...
// Staged once per workgroup in LDS: the activation row, g_idx (as 16-bit), and for every group the
// tile's (scale, -zero*scale) pairs, so the per-k lookups are LDS reads, not memory instructions.
#include "xbit_device.h"
#include "xbit_host.h"
#include "xbit_launch_all.h"

namespace xbit {

namespace {

constexpr int kGxCols = 64;      // columns per workgroup
constexpr int kGxSlots = 16;     // k-slots per workgroup (4 waves x 4)
constexpr int kGxThreads = 256;
constexpr uint32_t kGxChunkK = 4096;   // activations / g_idx staged per pass over K
constexpr uint32_t kGxMaxGroups = 512; // the group table must fit LDS: groups x 64 columns x 4 bytes = 128 KB of the 160 KB

template <int BITS>
__global__ void __launch_bounds__(kGxThreads) gemv_gidx_kernel(const GemvGidxParams p) {
  __shared__ __attribute__((aligned(16))) uint16_t s_a[kGxChunkK];
  __shared__ __attribute__((aligned(16))) uint16_t s_g[kGxChunkK];
  extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];  // [groups][64] (scale | nzs << 16)
  const int tid = threadIdx.x, nl = tid & 15, slot = tid >> 4;
  const uint32_t col0 = blockIdx.x * kGxCols, col = col0 + (uint32_t)nl * 4u, row = blockIdx.y;
  // group table of this tile: every thread converts a few (group, column) entries
  for (uint32_t e = tid; e < p.groups * kGxCols; e += kGxThreads) {
    const uint32_t g = e / kGxCols, c = col0 + e % kGxCols;
    uint32_t v = 0;
    if (c < p.n) {
      const _Float16 s = as_h(p.scales[(size_t)g * p.n + c]);
      const uint32_t z = stream_field(p.qzeros + (size_t)g * p.zw, 1, c, BITS, p.zw) + p.bias;
      v = (uint32_t)as_u16(s) | ((uint32_t)as_u16(neg_zero_times_scale(z, s)) << 16);
    }
    s_tab[e] = v;
  }

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool col_ok = col < p.n;  // n % 4 == 0 on this path
  for (uint32_t kc = 0; kc < p.k; kc += kGxChunkK) {
    __syncthreads();  // the previous chunk's readers are done (and, first time, the table is complete)
    for (uint32_t i = tid; i < kGxChunkK; i += kGxThreads) {
      const uint32_t k = kc + i;
      const bool in = k < p.k;
      s_a[i] = in ? p.a[(size_t)row * p.k + k] : (uint16_t)0;
      const uint32_t g = in ? (uint32_t)p.g_idx[k] : 0u;
      s_g[i] = (uint16_t)(g < p.groups ? g : p.groups - 1u);  // out-of-range indices are clamped, never followed
    }
    __syncthreads();
    const uint32_t kb_end = ((p.k - kc < kGxChunkK ? p.k - kc : kGxChunkK) + 31u) / 32u;
    for (uint32_t kb = slot; kb < kb_end; kb += kGxSlots) {
      uint32_t w[4][BITS];
#pragma unroll
      for (int i = 0; i < BITS; ++i) {
        const uint32_t prow = (kc / 32u + kb) * BITS + i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (col_ok && prow < p.kw) v = ld16(p.qweight + (size_t)prow * p.n + col);
        w[0][i] = v.x; w[1][i] = v.y; w[2][i] = v.z; w[3][i] = v.w;
      }
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        const uint32_t ki = kb * 32u + (uint32_t)kk;  // < kGxChunkK; rows past K carry a zero activation
        const float av = (float)as_h(s_a[ki]);
        const uint4 t = *reinterpret_cast<const uint4*>(s_tab + (uint32_t)s_g[ki] * kGxCols + nl * 4);
        const uint32_t tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const uint32_t Q = field<BITS>(w[2 * c2], kk) | (field<BITS>(w[2 * c2 + 1], kk) << 16) | kMagic1024x2;
          const half2_t s2 = half2_t{as_h((uint16_t)(tt[2 * c2] & 0xFFFFu)), as_h((uint16_t)(tt[2 * c2 + 1] & 0xFFFFu))};
          const half2_t nzs2 = half2_t{as_h((uint16_t)(tt[2 * c2] >> 16)), as_h((uint16_t)(tt[2 * c2 + 1] >> 16))};
          const half2_t wv = dequant_pair(Q, s2, nzs2);
          acc[2 * c2] += av * (float)wv.x;
          acc[2 * c2 + 1] += av * (float)wv.y;
        }
      }
    }
  }
  // sum over the 16 k-slots: through LDS (reusing the activation buffer: 16 x 64 floats = 4 KB)
  __syncthreads();
  float* red = reinterpret_cast<float*>(s_a);  // 8 KB available
  *reinterpret_cast<float4*>(red + (slot * kGxCols + nl * 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (tid < kGxCols) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < kGxSlots; ++s) sum += red[s * kGxCols + tid];
    if (col0 + tid < p.n) p.out[(size_t)row * p.n + col0 + tid] = as_u16((_Float16)sum);
  }
}

// exact for any shape / alignment / number of groups: one workgroup per output element
__global__ void __launch_bounds__(256) gemv_gidx_generic_kernel(const GemvGidxParams p, int bits) {
  __shared__ float s_part[256];
  const uint32_t col = blockIdx.x, row = blockIdx.y;
  float acc = 0.f;
  for (uint32_t k = threadIdx.x; k < p.k; k += 256) {
    uint32_t g = (uint32_t)p.g_idx[k];
    g = g < p.groups ? g : p.groups - 1u;
    const _Float16 s = as_h(p.scales[(size_t)g * p.n + col]);
    const uint32_t z = stream_field(p.qzeros + (size_t)g * p.zw, 1, col, bits, p.zw) + p.bias;
    const uint32_t q = stream_field(p.qweight + col, p.n, k, bits, p.kw);
    const _Float16 w = __builtin_elementwise_fma((_Float16)(float)(int)q, s, neg_zero_times_scale(z, s));
    acc += (float)as_h(p.a[(size_t)row * p.k + k]) * (float)w;
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_part[threadIdx.x] += s_part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.out[(size_t)row * p.n + col] = as_u16((_Float16)s_part[0]);
}

// dequant: one thread per (k, column pair); coalesced fp16x2 stores.  The op is write-bound and the
// scale / zero lookups hit L2, so the simple form is within reach of the contiguous-group kernel.
__global__ void __launch_bounds__(256) dequant_gidx_kernel(const GemvGidxParams p, int bits) {
  const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t half_n = p.n / 2u;  // n is even
  if (idx >= (uint64_t)p.k * half_n) return;
  const uint32_t k = (uint32_t)(idx / half_n), col = (uint32_t)(idx % half_n) * 2u;
  uint32_t g = (uint32_t)p.g_idx[k];
  g = g < p.groups ? g : p.groups - 1u;
  uint16_t o[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const _Float16 s = as_h(p.scales[(size_t)g * p.n + col + c]);
    const uint32_t z = stream_field(p.qzeros + (size_t)g * p.zw, 1, col + c, bits, p.zw) + p.bias;
    const uint32_t q = stream_field(p.qweight + col + c, p.n, k, bits, p.kw);
    o[c] = as_u16(__builtin_elementwise_fma((_Float16)(float)(int)q, s, neg_zero_times_scale(z, s)));
  }
  reinterpret_cast<uint32_t*>(p.out)[(size_t)k * half_n + col / 2u] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
}

template <int BITS>
int launch_gemv_gidx_bits(const GemvGidxParams& p, hipStream_t stream) {
  dim3 grid((p.n + kGxCols - 1) / kGxCols, p.m), block(kGxThreads);
  const size_t dyn = (size_t)p.groups * kGxCols * sizeof(uint32_t);
  void* args[] = {const_cast<GemvGidxParams*>(&p)};
  return launch_raw_dynamic(reinterpret_cast<const void*>(gemv_gidx_kernel<BITS>), grid, block, dyn, stream, args);
}

}  // namespace

int launch_gemv_gidx(const GemvGidxParams& p, int bits, hipStream_t stream) {
  if (bits < 2 || bits > 8) return XBIT_ERR_BITS;
  if (p.m > 65535u) return XBIT_ERR_SHAPE;
  // Up to 64 rows: the act-order builds of the staged-operand kernel (xbit_gemv_dyn.h), up to 32 rows per launch - the
  // weight stream and the arithmetic of the contiguous-group kernel plus a per-pair table lookup (4096x11008 w4, one row:
  // 29.8 -> 10.0 us, 1.5x the contiguous kernel; profiles/r03/gidx_before.txt, gidx_after.txt, gidx_rows_8_16.txt).  More
  // rows, very many groups or a K beyond its staging: the kernels of this file.
  if (p.m <= 64u) {
    GemvParams q{};
    q.qweight = p.qweight;
    q.scales = p.scales;
    q.qzeros = p.qzeros;
    q.k = p.k;
    q.n = p.n;
    q.groupsize = 128;  // not used by the act-order build
    q.kw = p.kw;
    q.groups = p.groups;
    q.zw = p.zw;
    q.bias = p.bias;
    // rows in launches of 32 where the 32-row build takes the layer (its LDS image grows with the groups), else of 16; every
    // chunk is checked before the first is launched (a 1..4-row tail takes another build than the full chunks)
    int st = kGemvDynNotEligible;
    for (uint32_t chunk = 32u; chunk >= 16u && st == kGemvDynNotEligible; chunk /= 2u) {
      st = launch_all_or_nothing([&]() -> int {
        int s = kGemvDynNotEligible;
        for (uint32_t r = 0; r < p.m; r += chunk) {
          q.m = p.m - r < chunk ? p.m - r : chunk;
          q.a = p.a + (size_t)r * p.k;
          q.out = p.out + (size_t)r * p.n;
          s = launch_gemv_dyn(q, bits, stream, 3, 16, true, false, nullptr, 1, nullptr, p.g_idx);
          if (s != XBIT_OK) break;
        }
        return s;
      });
    }
    if (st != kGemvDynNotEligible) return st;
  }
  const bool fast = p.groups <= kGxMaxGroups && p.n % 4u == 0 && ((uintptr_t)p.qweight % 16 == 0) &&
                    (uint64_t)p.kw * p.n < (1ull << 32) && (p.n + kGxCols - 1) / kGxCols <= 0x7FFFFFFFu;
  if (fast) {
    switch (bits) {
      case 2: return launch_gemv_gidx_bits<2>(p, stream);
      case 3: return launch_gemv_gidx_bits<3>(p, stream);
      case 4: return launch_gemv_gidx_bits<4>(p, stream);
      case 5: return launch_gemv_gidx_bits<5>(p, stream);
      case 6: return launch_gemv_gidx_bits<6>(p, stream);
      case 7: return launch_gemv_gidx_bits<7>(p, stream);
      default: return launch_gemv_gidx_bits<8>(p, stream);
    }
  }
  dim3 grid(p.n, p.m), block(256);
  return launch(gemv_gidx_generic_kernel, grid, block, stream, p, bits);
}

int launch_dequant_gidx(const GemvGidxParams& p, int bits, hipStream_t stream) {
  if (bits < 2 || bits > 8) return XBIT_ERR_BITS;
  {
    // the contiguous-group tile kernel with per-row group ids (4096x11008 w4, shuffled ids: 76 -> see profiles/r03/gidx_dequant.txt)
    const uint32_t kblocks = (p.k + 31u) / 32u;
    const bool aligned = ((uintptr_t)p.qweight % 16 == 0) && ((uintptr_t)p.scales % 16 == 0) && ((uintptr_t)p.out % 16 == 0) && ((uintptr_t)p.g_idx % 4 == 0);
    if (p.n % 8u == 0 && aligned && (kblocks + 3u) / 4u <= 65535u && p.groups >= 1u) {
      DequantParams q{};
      q.out = p.out;
      q.qweight = p.qweight;
      q.scales = p.scales;
      q.qzeros = p.qzeros;
      q.k = p.k;
      q.n = p.n;
      q.groupsize = 1u;  // not used by the by-id build
      q.kw = p.kw;
      q.groups = p.groups;
      q.zw = p.zw;
      q.bias = p.bias;
      q.kblocks = kblocks;
      return launch_dequant_by_id(q, p.g_idx, bits, stream);
    }
  }
  const uint64_t total = (uint64_t)p.k * (p.n / 2u);
  if ((total + 255) / 256 > 0x7FFFFFFFull) return XBIT_ERR_SHAPE;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  return launch(dequant_gidx_kernel, grid, block, stream, p, bits);
}

}  // namespace xbit
