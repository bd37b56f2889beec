// xbit_device.h — device-side building blocks shared by the dequant and GEMV kernels (gfx950).
//
// The unit of work everywhere is the "k-block": 32 consecutive k-values of one column, which the
// packed layout stores in exactly BITS consecutive int32 words of that column for every
// BITS in 2..8 (reference layout: /root/reference/src/cu/unpack_weight_2_to_7.cu:228-239 for
// 3/5/6/7 bit, :64-72 for 2/4/8 bit).  A lane always owns whole k-blocks, so no field it needs
// ever lives in another lane's registers.
//
// Dequantisation keeps the reference's fp16 arithmetic bit for bit
// (unpack_weight_2_to_7.cu:60,75 / :285,305):   zs = hmul(half(z + bias), s);  w = hfma(half(q), s, -zs)
// done two k-values at a time in packed fp16 (v_pk_add_f16 / v_pk_fma_f16).  The integer->half
// conversion is the exponent-splice trick: (q | 0x6400) read as fp16 is exactly 1024 + q for
// q < 1024, and 1024 + q - 1024 is exact in fp16, so half(q) costs one packed add per pair.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "xbit_unpack_gen.h"

namespace xbit {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half2_t as_h2(uint32_t x) { return __builtin_bit_cast(half2_t, x); }
__device__ __forceinline__ uint32_t as_u32(half2_t x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ _Float16 as_h(uint16_t x) { return __builtin_bit_cast(_Float16, x); }
__device__ __forceinline__ uint16_t as_u16(_Float16 x) { return __builtin_bit_cast(uint16_t, x); }

constexpr uint32_t kMagic1024x2 = 0x64006400u;  // (1024.0h, 1024.0h)

__device__ __forceinline__ half2_t splat(_Float16 x) { return half2_t{x, x}; }

// 16-byte global accesses.  The nontemporal forms mark read-once / write-once streams (packed
// weights, dequantised output) so they do not displace the small re-read operands in cache.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const void* p) {
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16_nt(void* p, uint4 v) {
  __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p));
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// component c (0..3) of a uint4 (c is a compile-time constant after unrolling)
__device__ __forceinline__ uint32_t comp_rt(const uint4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// Field `idx` (0..31, compile-time after unrolling) of the 32-field LSB-first stream w[0..BITS).
template <int BITS>
__device__ __forceinline__ uint32_t field(const uint32_t (&w)[BITS], int idx) {
  const int bit = idx * BITS, wi = bit >> 5, sh = bit & 31;
  constexpr uint32_t mask = (1u << BITS) - 1u;
  if (sh + BITS <= 32) return (w[wi] >> sh) & mask;                       // v_bfe_u32
  return __builtin_amdgcn_alignbit(w[wi + 1 < BITS ? wi + 1 : wi], w[wi], sh) & mask;  // straddles a word
}

// (x & mask) | magic must become ONE v_and_or_b32.  gfx950 VOP3 takes no literal operand, so with
// two literal constants hipcc emits v_and + v_or; handing it the constants as opaque registers
// (mask in an SGPR, magic in a VGPR) lets instruction selection pick the fused form.  (Writing the
// v_and_or_b32 itself in inline asm costs an s_nop pad per use.)
__device__ __forceinline__ uint32_t opaque_sgpr(uint32_t v) {
  uint32_t r;
  asm("s_mov_b32 %0, %1" : "=s"(r) : "i"(v));
  return r;
}
__device__ __forceinline__ uint32_t opaque_vgpr(uint32_t v) {
  uint32_t r;
  asm("v_mov_b32 %0, %1" : "=v"(r) : "i"(v));
  return r;
}

// How the 32 k-values of a k-block are paired into 16 packed-fp16 operands, and how a pair is
// extracted with ONE v_and_or_b32:  (X & (mask << e)) | 0x64006400 = (1024 + q_lo*2^e, 1024 + q_hi*2^e)
// whenever the two fields sit at the same bit offset e <= 10 - BITS of the two 16-bit halves of X.
// The conversion back is exact: fma(1024 + q*2^e, 2^-e, -1024*2^-e) = q  (dequant_pair<E>).
//   2/4/8 bit: X is the packed word itself (fields 16 bits apart pair up), shifted once per word to
//              bring the upper fields into the mantissa range;
//   3/5/6/7:   fields k and k+8 start exactly BITS bytes apart, so X is one v_perm_b32 of two
//              byte-aligned 16-bit windows (generated plans: xbit_unpack_gen.h, tools/gen_unpack.py).
// `pair_k(p, h)` is the k index (0..31) in half h (0 = low) of pair p; pairs never leave a chunk of
// kChunkK consecutive k, and the activation vector is staged into LDS in exactly this pair order.
template <int BITS>
struct Unpack {
  static constexpr bool kPow2 = (BITS == 2 || BITS == 4 || BITS == 8);
  static constexpr int kPerWord = kPow2 ? 32 / BITS : 0;  // fields per word
  static constexpr int kPerHalf = kPerWord / 2;           // fields per 16-bit half word
  static constexpr uint32_t kMask = (1u << BITS) - 1u;
  static constexpr int kFree = !kPow2 ? 1 : (10 / BITS < kPerHalf ? 10 / BITS : kPerHalf);  // fields usable per view
  static constexpr int kShiftStep = kFree * BITS;
  static constexpr int kChunkK = (kPow2 && BITS != 2) ? 8 : 16;  // span of k inside which pairs are formed
  static constexpr int kMaskN = 11 - BITS;                       // admissible field offsets e = 0 .. 10 - BITS

  __host__ __device__ static constexpr int pair_k(int p, int h) {
    if (kPow2) return (p / kPerHalf) * kPerWord + (p % kPerHalf) + h * kPerHalf;
    return 16 * (p / 8) + (p % 8) + 8 * h;
  }
  __host__ __device__ static constexpr int pair_exp(int p) {
    if constexpr (kPow2) return ((p % kPerHalf) % kFree) * BITS;
    else return OddUnpack<BITS>::kExp[p];
  }
  // packed rows (of the k-block's BITS) that hold every field of MFMA octet `oct` (pairs 4*oct..4*oct+3)
  __host__ __device__ static constexpr int rows_for_octet(int oct) {
    if (kPow2) return ((oct + 1) * 8 * BITS + 31) / 32;
    return ((oct / 2 + 1) * 16 * BITS + 31) / 32;
  }

  // Loop-invariant constants: build once per kernel.
  struct Consts {
    uint32_t magic;         // VGPR: 0x64006400
    uint32_t mask[kMaskN];  // SGPRs: (field mask << e) in both halves, e = index
  };
  __device__ __forceinline__ static Consts consts() {
    Consts k;
    k.magic = opaque_vgpr(kMagic1024x2);
    static_for<kMaskN>([&](auto e) { k.mask[decltype(e)::value] = opaque_sgpr((kMask << decltype(e)::value) * 0x00010001u); });
    return k;
  }

  // fp16x2 bit pattern (1024 + q_lo * 2^e, 1024 + q_hi * 2^e) of pair P, e = pair_exp(P)
  template <int P>
  __device__ __forceinline__ static uint32_t pair(const uint32_t (&w)[BITS], const Consts& k) {
    if constexpr (kPow2) {
      constexpr int c = P / kPerHalf, i = P % kPerHalf, view = i / kFree;
      const uint32_t x = view == 0 ? w[c] : (w[c] >> (view * kShiftStep));  // CSE'd across the pairs of a word
      return (x & k.mask[(i % kFree) * BITS]) | k.magic;                    // v_and_or_b32
    } else {
      return OddUnpack<BITS>::template pair<P>(w, k);
    }
  }

  // --- subnormal form (the fast path) ---------------------------------------------------------
  // A field moved to mantissa bits [kSubE, 10) of an otherwise zero fp16 IS the subnormal
  // q * 2^(kSubE - 24): no exponent to splice in and no 1024 to subtract afterwards, so
  //   w = fma(Qsub, s * 2^(24 - kSubE), -zs)        (product exact inside the fma = q * s)
  // is the reference's single-rounded hfma in ONE packed op per pair.  Extraction is one shift and
  // one v_and_b32 per pair, both full-rate VOP2 (1.45 cycles per wave-instruction against 3.5 for
  // v_and_or_b32 / v_pk_*: profiles/r01/valu_issue_rates.txt).  Valid while the pre-scaled scale is
  // finite in fp16, |s| <= 65504 * 2^-(14 + BITS) (a quantisation range of about 4.0 for every
  // BITS); the caller checks that per wave and otherwise takes the spliced form above.
  static constexpr int kSubE = 10 - BITS;
  static constexpr int kSubScaleLog2 = 24 - kSubE;  // 14 + BITS
  static constexpr uint32_t kSubMask2 = (kMask << kSubE) * 0x00010001u;

  template <int P>
  __device__ __forceinline__ static uint32_t sub_pair(const uint32_t (&w)[BITS], uint32_t mask) {
    if constexpr (kPow2) {
      constexpr int c = P / kPerHalf, pos = (P % kPerHalf) * BITS, t = pos - kSubE;
      if constexpr (t > 0) return (w[c] >> t) & mask;
      else if constexpr (t < 0) return (w[c] << (-t)) & mask;
      else return w[c] & mask;
    } else {
      return OddUnpack<BITS>::template sub_pair<P>(w, mask);
    }
  }
};

// Pre-scaled scale for the subnormal form: fp16(s * 2^(14 + BITS)), exact (a pure exponent shift,
// subnormal s included) whenever it is finite; `ok` is false when it is not (or s is NaN/Inf).
template <int BITS>
__device__ __forceinline__ _Float16 sub_scale(_Float16 s, bool& ok) {
  const float sf = (float)s * (float)(1u << Unpack<BITS>::kSubScaleLog2);
  ok = __builtin_fabsf(sf) <= 65504.0f;
  return (_Float16)sf;
}

// w2 = hfma2(half2(q), s2, nzs2) from the spliced pattern Q = (1024 + q * 2^E) x2 (see Unpack)
template <int E = 0>
__device__ __forceinline__ half2_t dequant_pair(uint32_t Q, half2_t s2, half2_t nzs2) {
  half2_t q;
  if constexpr (E == 0) {
    q = as_h2(Q) - splat((_Float16)1024.0f);
  } else {
    constexpr float inv = 1.0f / (float)(1 << E);
    q = __builtin_elementwise_fma(as_h2(Q), splat((_Float16)inv), splat((_Float16)(-1024.0f * inv)));
  }
  return __builtin_elementwise_fma(q, s2, nzs2);
}

// (zero-point + bias) * scale in fp16, negated, exactly as the reference's hmul2 then unary minus
__device__ __forceinline__ _Float16 neg_zero_times_scale(uint32_t z_plus_bias, _Float16 s) {
  const _Float16 zh = (_Float16)(float)(int)z_plus_bias;  // <= 510: exact
  return -(zh * s);
}

// 32-bit window holding the 4 zero-points of columns col..col+3 (col % 4 == 0) of one group row.
// The row is an LSB-first stream of BITS-wide fields along N
// (reference: unpack_weight_2_to_7.cu:56-59 and :258-281).
template <int BITS>
__device__ __forceinline__ uint32_t zero_window4(const uint32_t* __restrict__ zrow, uint32_t col, uint32_t zw) {
  const uint32_t bit = col * BITS, wi = bit >> 5, sh = bit & 31u;
  const uint32_t lo = zrow[wi];
  if constexpr (BITS == 2 || BITS == 4 || BITS == 8) return lo >> sh;  // 4*BITS divides 32: never straddles
  const uint32_t hi = zrow[wi + 1 < zw ? wi + 1 : wi];
  return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// Runtime-indexed field of a stream in global memory (generic, slow paths only).
__device__ __forceinline__ uint32_t stream_field(const uint32_t* __restrict__ base, size_t stride, uint64_t idx, int bits,
                                                 uint64_t nwords) {
  const uint64_t bit = idx * (uint64_t)bits, w = bit >> 5;
  const uint32_t sh = (uint32_t)(bit & 31u);
  const uint64_t lo = base[w * stride];
  const uint64_t hi = (sh + (uint32_t)bits > 32u && w + 1 < nwords) ? base[(w + 1) * stride] : 0;
  return (uint32_t)(((lo | (hi << 32)) >> sh) & ((1u << bits) - 1u));
}

// Workgroups b and b+8 are observed to share an XCD (and its L2); give each XCD a contiguous run
// of tiles so neighbouring column tiles — which share scale / zero-point cache lines — meet in one
// L2.  Pure speed: any placement gives the same result.  Bijective for every tile count.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t tiles) {
  const uint32_t q = tiles >> 3, r = tiles & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

}  // namespace xbit
