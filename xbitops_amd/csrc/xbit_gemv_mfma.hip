// xbit_gemv_mfma.hip — the production A16Wx GEMV / skinny-GEMM kernel (M = 1..16 per pass) for gfx950.
//
// Same decomposition as xbit_gemv.hip (a workgroup owns a column tile over all of K; lanes own
// whole 32-k blocks; all packed-weight loads of a batch are issued before any is consumed; the
// activation slice is staged per wave in LDS in the unpacker's pair order; W is rebuilt bit-exactly
// in packed fp16), but the multiply-accumulate runs on the matrix pipe:
//
//   v_mfma_f32_16x16x32_f16:  D[16 x 16] += A[16 x 32] * B[32 x 16],  fp32 accumulate
//     lane l supplies  B[k = 8*(l>>4) + j][col = l&15]  and  A[row = l&15][k = 8*(l>>4) + j], j < 8.
//
// Why a matrix instruction in a bandwidth-bound GEMV: measured on MI355X (tools/microbench/
// valu_rates.hip, profiles/) every VALU op this kernel needs — v_and_or_b32, v_pk_add_f16,
// v_pk_fma_f16 and also v_dot2c_f32_f16 — retires one wave-instruction per ~3.5 cycles per SIMD at
// best (7 with one wave per SIMD).  With the dot product on the VALU (xbit_gemv.hip: 4.3 ops per
// pair of weights) the unpack+MAC stream needs 2-3x the time HBM needs to deliver the weights.
// Moving the MACs to the otherwise idle matrix pipe removes a quarter of the VALU stream, removes
// the cross-lane shuffle reduction (the MFMA sums the four k-lane groups), and makes M = 2..16
// rows cost the same as M = 1.  Nothing is reshaped: the "k" of the MFMA is simply 4 lane groups x
// 8 consecutive fields of the k-block each lane already owns.
//
// Lane map (wave64): kq = lane >> 4 picks the k-batch (UB k-blocks) of this lane, nc = lane & 15 the
// column group; a lane loads CPL adjacent columns (8 or 16 bytes) of each packed row of its
// k-blocks.  MFMA step (block j, octet i, column c): B fragment = fields 8i..8i+7 (pair order) of
// block j, column c, from every lane; A fragment = the matching 8 activations of row (nc % MT),
// read from LDS as one broadcast ds_read_b128.  Output element: D[row][nc] of accumulator c is
// y[m0 + row][tile*CN + nc*CPL + c].
#include "xbit_device.h"
#include "xbit_host.h"

namespace xbit {

namespace {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// CPL adjacent packed words of one row: 8-byte or 16-byte streaming load
template <int CPL, bool NT>
struct RowLoad;
template <bool NT>
struct RowLoad<2, NT> {
  uint32_t v[2];
  __device__ __forceinline__ void load(const uint32_t* p) {
    u32x2_t t;
    if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    else t = *reinterpret_cast<const u32x2_t*>(p);
    v[0] = t.x;
    v[1] = t.y;
  }
};
template <bool NT>
struct RowLoad<4, NT> {
  uint32_t v[4];
  __device__ __forceinline__ void load(const uint32_t* p) {
    const uint4 t = NT ? ld16_nt(p) : ld16(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};

// Template parameters
//   BITS  weight width 2..8          UB   k-blocks per lane and sweep (<= 16 row loads in flight)
//   CPL   adjacent columns per lane (2: 8-byte loads, 4: 16-byte loads)
//   NL    lanes side by side along N (16 or 8).  The MFMA tile has 16 column slots (lane & 15):
//         with NL = 16 they are 16 column groups; with NL = 8 they are 8 column groups x 2 k-slots,
//         and the two k-slots are kept apart by feeding their activations to row blocks 0-7 and
//         8-15 of the A operand: D[ks*8 + m][ks*8 + nl] is the sum over k-slot ks, the off-diagonal
//         quadrants are don't-care.  NL = 8 halves the tile width (finer load balance over the 256
//         CUs) and needs MT <= 8.
//   NW    waves per workgroup        MT   activation rows per pass (<= 16 * NL / 16)
//   NT    nontemporal weight loads   GUARD  false when the shape needs no edge handling at all
template <int BITS, int UB, int CPL, int NL, int NW, int MT, bool NT, bool GUARD>
__global__ void __launch_bounds__(64 * NW) gemv_mfma_kernel(const GemvParams p) {
  static_assert(CPL == 2 || CPL == 4, "columns per lane");
  static_assert(NL == 16 || NL == 8, "lanes along N");
  constexpr int LK = 64 / NL;            // k-slots per wave
  constexpr int RS = 16 / NL;            // k-slots sharing one MFMA k-quad (row split of the A tile)
  constexpr int RB = 16 / RS;            // rows per row block
  static_assert(MT == 1 || MT == 2 || MT == 4 || MT == 8 || MT == 16, "rows per pass");
  static_assert(MT <= RB, "rows per pass exceed the row block");
  constexpr int T = LK * NW;             // k-batches per workgroup sweep
  constexpr int CN = NL * CPL;           // columns per workgroup
  constexpr int KLS = UB * 16 + 4;       // dwords per (row, k-slot) activation slot (+4: bank spread)
  constexpr int ACT_DW = MT * LK * KLS;  // staged activation dwords per wave
  constexpr int NTASK = MT * LK * UB;    // 32-k staging tasks per wave and sweep
  constexpr int ROUNDS = (NTASK + 63) / 64;

  __shared__ __attribute__((aligned(16))) uint32_t s_act[NW * ACT_DW];
  __shared__ __attribute__((aligned(16))) float s_red[NW * RS * MT * CN];

  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, kl = lane / NL, nl = lane % NL;
  const uint32_t tile = xcd_tile(blockIdx.x, p.tiles);
  const uint32_t col = tile * CN + (uint32_t)(CPL * nl);
  const bool col_ok = GUARD ? col < p.n : true;  // n % CPL == 0 on this path
  const uint32_t col_c = col_ok ? col : 0u;
  const uint32_t m0 = blockIdx.y * MT;
  uint32_t* const my_act = s_act + wv * ACT_DW;
  const int arow = nl % MT;  // activation row this lane feeds to its row block of the A tile
  const typename Unpack<BITS>::Consts uk = Unpack<BITS>::consts();
  const uint32_t row_bytes = p.n * 4u;  // one packed row

  f4_t acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = f4_t{0.f, 0.f, 0.f, 0.f};

  for (uint32_t b0 = 0; b0 < p.batches; b0 += T) {
    const uint32_t wb = b0 + wv * LK;  // first batch of this wave in this sweep
    const uint32_t bt = wb + kl;       // this lane's batch
    const bool active = GUARD ? (col_ok && bt < p.batches) : true;
    const uint32_t bt_c = (!GUARD || bt < p.batches) ? bt : p.batches - 1;

    // (1) activation slice of this wave (oldest loads in the vmcnt queue).  Clamped addresses;
    //     out-of-range chunks are zeroed when staged (k >= K must contribute 0).
    uint4 hraw[ROUNDS][4];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int s = (lane + 64 * r) % NTASK;
      const int mi = s / (LK * UB), blk = s % (LK * UB);
      const uint32_t k0 = (wb * UB + blk) * 32u;
      const uint32_t row = (!GUARD || m0 + mi < p.m) ? m0 + mi : p.m - 1;
      const char* src = reinterpret_cast<const char*>(p.a) + (size_t)row * p.k * 2u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t kk = k0 + 8u * q;
        hraw[r][q] = ld16(src + 2u * ((!GUARD || kk < p.k) ? kk : 0u));  // k % 8 == 0 on this path
      }
    }

    // (2) scale and zero-point of the batch's group (one group per batch on this path)
    uint32_t gi = (bt_c * (uint32_t)(UB * 32)) / p.groupsize;
    if (GUARD) gi = gi < p.groups ? gi : p.groups - 1;
    uint32_t sraw[CPL / 2];
#pragma unroll
    for (int c2 = 0; c2 < CPL / 2; ++c2)
      sraw[c2] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.scales) + 2u * (gi * p.n + col_c + 2 * c2));
    const uint32_t zwin = zero_window4<BITS>(p.qzeros + gi * p.zw, col_c, p.zw);

    // (3) all packed-weight loads of this lane's batch.  The lane part of the address is one 32-bit
    //     byte offset; the row step is wave-uniform and rides in the scalar base, so a load costs no
    //     VALU instruction.  (Rows past the end are clamped under GUARD: their k >= K.)
    RowLoad<CPL, NT> w[UB][BITS];
    const uint32_t lane_off = (bt_c * (uint32_t)(UB * BITS)) * row_bytes + col_c * 4u;
#pragma unroll
    for (int j = 0; j < UB; ++j)
#pragma unroll
      for (int i = 0; i < BITS; ++i) {
        if constexpr (GUARD) {
          const uint32_t row = (bt_c * UB + j) * BITS + i;
          w[j][i].load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.qweight) +
                                                         (size_t)(row < p.kw ? row : p.kw - 1) * row_bytes + col_c * 4u));
        } else {
          const char* rowbase = reinterpret_cast<const char*>(p.qweight) + (size_t)(j * BITS + i) * row_bytes;  // uniform
          w[j][i].load(reinterpret_cast<const uint32_t*>(rowbase + lane_off));
        }
      }
    // Pin the issue order act -> scale/zero -> weights: each wait below then leaves every younger
    // load in flight (vmcnt retires in order).
    __builtin_amdgcn_sched_barrier(0);

    // (4) permute the activation slice into pair order and park it in this wave's LDS region
    //     (every lane stores; lanes beyond NTASK repeat a task with identical data)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int s = (lane + 64 * r) % NTASK;
      const int mi = s / (LK * UB), blk = s % (LK * UB);
      uint32_t h[16];
      if constexpr (GUARD) {
        const uint32_t k0 = (wb * UB + blk) * 32u;
        const bool row_ok = m0 + mi < p.m;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = row_ok && k0 + 8u * q < p.k;
          h[4 * q + 0] = ok ? hraw[r][q].x : 0u;
          h[4 * q + 1] = ok ? hraw[r][q].y : 0u;
          h[4 * q + 2] = ok ? hraw[r][q].z : 0u;
          h[4 * q + 3] = ok ? hraw[r][q].w : 0u;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[4 * q + 0] = hraw[r][q].x;
          h[4 * q + 1] = hraw[r][q].y;
          h[4 * q + 2] = hraw[r][q].z;
          h[4 * q + 3] = hraw[r][q].w;
        }
      }
      uint32_t* dst = my_act + (mi * LK + blk / UB) * KLS + (blk % UB) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pr = 4 * q + e;
          const int klo = Unpack<BITS>::pair_k(pr, 0), khi = Unpack<BITS>::pair_k(pr, 1);
          const uint32_t sel = (uint32_t)(2 * (klo & 1)) | ((uint32_t)(2 * (klo & 1) + 1) << 8) |
                               ((uint32_t)(4 + 2 * (khi & 1)) << 16) | ((uint32_t)(5 + 2 * (khi & 1)) << 24);
          o[e] = __builtin_amdgcn_perm(h[khi >> 1], h[klo >> 1], sel);
        }
        *reinterpret_cast<u32x4_t*>(dst + 4 * q) = u32x4_t{o[0], o[1], o[2], o[3]};
      }
    }
    // produced and consumed by the same wave: in-order LDS + a wave-level fence suffice
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // (5) per-column dequant constants; an inactive lane gets scale 0 => its weights are exactly 0
    half2_t s2[CPL], nzs2[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t sr = (sraw[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
      const _Float16 s = active ? as_h((uint16_t)sr) : (_Float16)0.0f;
      const uint32_t z = ((zwin >> (c * BITS)) & Unpack<BITS>::kMask) + p.bias;
      s2[c] = splat(s);
      nzs2[c] = splat(neg_zero_times_scale(z, s));
    }

    // (6) unpack + dequantise on the VALU, multiply-accumulate on the matrix pipe
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      uint32_t wc[CPL][BITS];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int i = 0; i < BITS; ++i) wc[c][i] = w[j][i].v[c];

      static_for<4>([&](auto oct_c) {
        constexpr int oct = decltype(oct_c)::value;
        const u32x4_t araw = *reinterpret_cast<const u32x4_t*>(my_act + (arow * LK + kl) * KLS + j * 16 + 4 * oct);
        const h8_t a_frag = __builtin_bit_cast(h8_t, araw);
        u32x4_t braw[CPL];
        static_for<4>([&](auto e_c) {
          constexpr int e = decltype(e_c)::value;
          constexpr int pr = 4 * oct + e;
          uint32_t Q[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) Q[c] = Unpack<BITS>::template pair<pr>(wc[c], uk);
#pragma unroll
          for (int c = 0; c < CPL; ++c) braw[c][e] = as_u32(dequant_pair<Unpack<BITS>::pair_exp(pr)>(Q[c], s2[c], nzs2[c]));
        });
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_frag, __builtin_bit_cast(h8_t, braw[c]), acc[c], 0, 0, 0);
      });
    }
    __builtin_amdgcn_wave_barrier();  // keep the next sweep's LDS writes behind these reads
  }

  // D[drow][slot]: drow = 4*(lane>>4) + r lives in register r of lane (lane>>4, slot = lane&15).
  // Keep the quadrant where the row block equals the slot's k-slot, then sum over k-slots and waves.
  {
    const int slot = lane & 15, ks = slot / NL, snl = slot % NL;
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int drow = 4 * (lane >> 4) + r;
        const int m = drow % RB;
        if (drow / RB == ks && m < MT) s_red[((wv * RS + ks) * MT + m) * CN + snl * CPL + c] = acc[c][r];
      }
  }
  __syncthreads();
  for (int e = tid; e < MT * CN; e += 64 * NW) {
    const int m = e / CN, c = e % CN;
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NW * RS; ++v) sum += s_red[(v * MT + m) * CN + c];
    const uint32_t n_out = tile * CN + c, row = m0 + m;
    if (!GUARD || (n_out < p.n && row < p.m)) p.out[(size_t)row * p.n + n_out] = as_u16((_Float16)sum);
  }
}

template <int BITS, int UB, int CPL, int NL, int NW, int MT, bool NT>
int launch_one(GemvParams p, hipStream_t stream) {
  constexpr int CN = NL * CPL, T = (64 / NL) * NW;
  p.tiles = (p.n + CN - 1) / CN;
  const uint32_t kblocks = (p.k + 31) / 32;
  p.batches = (kblocks + UB - 1) / UB;
  dim3 grid(p.tiles, (p.m + MT - 1) / MT), block(64 * NW);
  const bool exact = (p.k % (32u * UB) == 0) && (p.batches % T == 0) && (p.n % CN == 0) && (p.m % MT == 0) &&
                     ((uint64_t)p.kw * p.n * 4u < (1ull << 32));
  if (exact) return launch(gemv_mfma_kernel<BITS, UB, CPL, NL, NW, MT, NT, false>, grid, block, stream, p);
  return launch(gemv_mfma_kernel<BITS, UB, CPL, NL, NW, MT, NT, true>, grid, block, stream, p);
}

// rows per pass: 1, 4, 8 or 16 (fewer staged rows = less LDS traffic); NL = 8 tiles take <= 8 rows
template <int BITS, int UB, int CPL, int NL, int NW, bool NT>
int launch_rows(const GemvParams& p, hipStream_t stream) {
  if (p.m == 1) return launch_one<BITS, UB, CPL, NL, NW, 1, NT>(p, stream);
  if (p.m <= 4) return launch_one<BITS, UB, CPL, NL, NW, 4, NT>(p, stream);
  // 8-16 rows: the staged activation tile is much larger, so fewer waves and smaller batches
  constexpr int UBR = UB > 2 ? 2 : UB;
  if (p.m <= 8) return launch_one<BITS, UBR, CPL, NL, 4, 8, NT>(p, stream);
  return launch_one<BITS, UBR, CPL, 16, 4, 16, NT>(p, stream);
}

template <int BITS>
constexpr int max_ub() {
  return BITS <= 4 ? 4 : 2;  // <= 16 outstanding row loads per lane
}

template <int BITS>
int launch_bits(const GemvParams& p, hipStream_t stream, int variant) {
  constexpr int UBMAX = max_ub<BITS>();
  const uint32_t kblocks = (p.k + 31) / 32;
  if (p.groupsize % (32u * UBMAX) == 0) {
    if constexpr (BITS == 4) {  // tuning variants exist for the headline bit-width only
      switch (variant) {
        case 11: return launch_rows<BITS, 4, 2, 16, 8, false>(p, stream);
        case 12: return launch_rows<BITS, 2, 2, 16, 16, true>(p, stream);
        case 13: return launch_rows<BITS, 4, 4, 16, 8, true>(p, stream);
        case 14: return launch_rows<BITS, 2, 4, 16, 16, true>(p, stream);
        case 15: return launch_rows<BITS, 4, 2, 8, 4, true>(p, stream);
        case 16: return launch_rows<BITS, 2, 2, 8, 8, true>(p, stream);
        case 17: return launch_rows<BITS, 4, 4, 8, 4, true>(p, stream);
        case 18: return launch_rows<BITS, 2, 4, 8, 8, true>(p, stream);
        case 19: return launch_rows<BITS, 1, 2, 8, 16, true>(p, stream);
        case 20: return launch_rows<BITS, 2, 2, 8, 4, true>(p, stream);
        default: break;
      }
    }
    // enough waves to cover K in one sweep when possible (T = 4*NW batches per sweep)
    if (kblocks / UBMAX >= 64) return launch_rows<BITS, UBMAX, 2, 16, 16, true>(p, stream);
    return launch_rows<BITS, UBMAX, 2, 16, 8, true>(p, stream);
  }
  if constexpr (UBMAX > 2) {
    if (p.groupsize % 64u == 0) return launch_rows<BITS, 2, 2, 16, 8, true>(p, stream);
  }
  return launch_rows<BITS, 1, 2, 16, 8, true>(p, stream);
}

}  // namespace

int launch_gemv_mfma(const GemvParams& p, int bits, hipStream_t stream, int variant) {
  switch (bits) {
    case 2: return launch_bits<2>(p, stream, variant);
    case 3: return launch_bits<3>(p, stream, variant);
    case 4: return launch_bits<4>(p, stream, variant);
    case 5: return launch_bits<5>(p, stream, variant);
    case 6: return launch_bits<6>(p, stream, variant);
    case 7: return launch_bits<7>(p, stream, variant);
    case 8: return launch_bits<8>(p, stream, variant);
    default: return XBIT_ERR_BITS;
  }
}

}  // namespace xbit
