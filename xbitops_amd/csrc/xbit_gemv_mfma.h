// xbit_gemv_mfma.h — the A16Wx GEMV / skinny-GEMM kernel of rounds 1-2 (M = 1..16 per pass) for gfx950, and the interpreter of the launch
// rule (xbit_gemv_rules.h says WHICH launch serves a call - a table of rows, each with its evidence; launch_by_rules below says HOW).
// What this kernel still serves, with the measurement that keeps it there, is the table's last rows and DESIGN.md section 3.2:
// 128-column tiles of very wide 4- / 8-bit layers, 16-column tiles of few-column layers up to 8 rows, 160-tile layers and 2-bit
// few-column layers at one row, group sizes 32 x odd, split plans whose K ranges exceed the staged-operand kernel's staging, 16-row
// passes where no 32-row launch applies (few tiles, 2-bit, more than 128 rows).  Measurement variants: xbit_gemv_tuning.h (tuning builds).
//
// A workgroup owns a column tile over all of K (or one K range of it: split-K); lanes own whole
// 32-k blocks of CPL adjacent columns; the packed rows of a lane are loaded a few rows ahead of the
// step that consumes them (LA); the activation slice of a wave is staged in LDS in the unpacker's
// pair order; W is rebuilt bit-exactly in packed fp16 (xbit_device.h) and multiplied on the matrix
// pipe:
//
//   v_mfma_f32_16x16x32_f16:  D[16 x 16] += A[16 x 32] * B[32 x 16],  fp32 accumulate
//     lane l supplies  B[k = 8*(l>>4) + j][col = l&15]  and  A[row = l&15][k = 8*(l>>4) + j], j < 8.
//
// Why a matrix instruction in a bandwidth-bound GEMV: measured on MI355X (tools/microbench/
// valu_rates.hip, profiles/) the packed-fp16 ops and v_dot2c_f32_f16 retire one wave-instruction
// per ~3.5 cycles per SIMD at best.  With the dot product on the VALU (xbit_gemv.hip: 4.3 ops per
// pair of weights) the kernel was VALU-bound at 8.8-9.3 us on the 23 MB metric shape.  Moving the
// MACs to the otherwise idle matrix pipe removes them from the VALU stream, removes the cross-lane
// shuffle reduction (the MFMA sums the four k-lane groups), and makes M = 2..16 rows cost about the
// same as M = 1.  Nothing is reshaped: the "k" of the MFMA is simply 4 lane groups x 8 consecutive
// fields of the k-block each lane already owns.
//
// Lane map (wave64, NL = 16): kl = lane >> 4 picks the k-batch (UB k-blocks) of this lane, nl =
// lane & 15 the column group; a lane loads CPL adjacent columns (8, 12 or 16 bytes) of each packed row
// of its k-blocks.  MFMA step (block j, octet i, column c): B fragment = fields 8i..8i+7 (pair
// order) of block j, column c, from every lane; A fragment = the matching 8 activations of row
// (nl % MT), read from LDS as one ds_read_b128.  Output element: D[row][nl] of accumulator c is
// y[m0 + row][tile*CN + nl*CPL + c].  NL = 8 and NL = 32 re-map lanes to narrower / wider tiles (see
// the kernel).
#pragma once

#include "xbit_device.h"
#include "xbit_host.h"
#include "xbit_launch_all.h"
#include "xbit_gemv_rules.h"

namespace xbit {

namespace {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));

// Buffer descriptors give (a) a 32-bit per-lane byte offset plus a wave-uniform scalar offset, so
// stepping to the next packed row costs no VALU instruction, (b) hardware range checking: a load
// past the end of the tensor returns 0, which is exactly the neutral value wanted for edge rows.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr uint32_t kOob = 0x80000000u;  // a lane offset that is out of range for every tensor (< 2 GiB each)

// Which row loads must be out before MFMA step s = (k-block s/4, octet s%4) of a sweep.
template <int BITS, int UB, int LA>
struct LoadSchedule {
  static constexpr int kTotal = UB * BITS;  // row loads per lane and sweep
  // loads step s consumes: all rows of earlier blocks plus the rows holding the octet's pairs
  static constexpr int need(int s) { return (s / 4) * BITS + Unpack<BITS>::rows_for_octet(s % 4); }
  // loads issued once step s's prefetch is out (LA == 0: everything up front)
  // LA >= 100 (measurement builds): a ramp - lookahead la0 = (LA-100)/10 before the first step, one more per
  // step up to lamax = (LA-100)%10: every wave starts with a short burst (the CU serves waves' bursts one
  // after the other), the window then widens while the wave computes
  static constexpr int la_at(int s) {
    if (LA < 100) return LA;
    const int la0 = (LA - 100) / 10, lamax = (LA - 100) % 10;
    return la0 + s < lamax ? la0 + s : lamax;
  }
  static constexpr int issued_before(int s) {
    if (LA == 0) return kTotal;
    return need(s) + la_at(s) < kTotal ? need(s) + la_at(s) : kTotal;
  }
};

// CPL adjacent packed words of one row (8 / 12 / 16 bytes per lane)
template <int CPL, bool NT>
struct RowLoad {
  uint32_t v[CPL];
  __device__ __forceinline__ void load(rsrc_t r, uint32_t voff, uint32_t soff) {
    constexpr int aux = NT ? 2 : 0;  // nt: read-once stream
    if constexpr (CPL == 2) {
      const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux);
      v[0] = t.x; v[1] = t.y;
    } else if constexpr (CPL == 3) {
      const u32x3_t t = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, aux);
      v[0] = t.x; v[1] = t.y; v[2] = t.z;
    } else {
      const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
  }
};

// Template parameters
//   BITS  weight width 2..8          UB   k-blocks per lane and sweep (UB*BITS row loads in flight)
//   CPL   adjacent columns per lane: 2, 3 or 4 (8/12/16-byte loads; tile = NL*CPL columns).  Picked
//         by the launcher so that the number of tiles sits just under a multiple of the CU count.
//   NL    lanes side by side along N (8, 16 or 32).  The MFMA tile has 16 column slots (lane & 15):
//         with NL = 16 they are 16 column groups; with NL = 8 they are 8 column groups x 2 k-slots,
//         kept apart by feeding their activations to row blocks 0-7 and 8-15 of the A operand:
//         D[ks*8 + m][ks*8 + nl] is the sum over k-slot ks, the off-diagonal quadrants are don't-care;
//         with NL = 32 two column sets share the slots in k-octets of different parity and each row
//         block carries zeros in the other set's octets (512 contiguous bytes per row and instruction).
//   NW    waves per workgroup        MT   activation rows per pass (<= NL)
//   NT    nontemporal weight loads   GUARD  false when the shape needs no edge handling at all
//   ABLATE (measurement builds only; every one of them computes the production results):
//          4 = always the spliced (pre-subnormal) unpack, for A/B timing;
//          8 = per-wave s_memrealtime stamps written to p.trace;
//          16 = (M = 1) multiply-accumulate on the VALU (v_dot2_f32_f16 + shuffles) instead of the MFMA;
//          1024 = first weight rows ahead of the small operands;
//          8192 / 16384 = activation slice of an 8-16 row pass staged in 4 / 2 parts;
//          32768 / 65536 = other forms of the reduction reads;
//          131072 = wait for a step's rows before issuing the next.
//          XBIT_TUNING builds only, each with its measurement under profiles/ (DESIGN.md section 4).  The probes that
//          did NOT compute the GEMV (loads-only and no-operand ablations, cache-policy bits, tile-contiguous address
//          streams, next-call prefetch workgroups, helper waves, the integer-domain instruction mix) lived here in
//          rounds 1-2; their results are in profiles/r01 and profiles/r02, their code in the history (tools/README.md),
//          and such experiments now go to tools/microbench/gemv_lab.hip, outside the library.
//   LA   load lookahead in row loads (0 = issue the whole batch up front).  A wave issues in order,
//        and the CU's memory pipeline back-pressures: a wave that queues all its loads first sits at
//        the last load instruction until the queue drains and only then starts on data that came
//        back long ago.  With LA > 0 the row loads are woven into the compute stream: before MFMA
//        step s the loads needed by steps <= s + LA are issued, so a wave computes while its later
//        rows are in flight and never holds more than LA + 1 queue slots.
//   SPLIT  split-K: workgroup (tile, split) sums k-batches [split*bps, (split+1)*bps) only; the
//        partial sums meet in per-element 64-bit fixed-point accumulators (one returning atomic
//        per element and workgroup, see the epilogue).  For few column tiles and long K
//        (down-projections), where whole-K tiles leave CUs idle.
//        (Fused launches over several weight sets: the MULTI builds of the staged-operand kernel, xbit_gemv_dyn.h, since round 4.)
template <int BITS, int UB, int CPL, int NL, int NW, int MT, bool NT, bool GUARD, int ABLATE = 0, int LA = 0, bool SPLIT = false>
__global__ void __launch_bounds__(64 * NW) gemv_mfma_kernel(const GemvParams p) {
  static_assert(!SPLIT || GUARD, "split-K launches use the guarded build");
  static_assert(CPL >= 2 && CPL <= 4, "columns per lane");
  static_assert(NL == 32 || NL == 16 || NL == 8, "lanes along N");
  constexpr bool WAITFIRST = (ABLATE & 131072) != 0;
  constexpr bool WFIRST = (ABLATE & 1024) != 0;  // measurement: first weight rows issued ahead of the activation slice
  constexpr int LK = 64 / NL;                // k-slots per wave
  constexpr int RS = NL < 16 ? 16 / NL : 1;  // NL = 8: k-slots sharing one MFMA k-octet (in different column slots)
  constexpr int CS = NL > 16 ? NL / 16 : 1;  // NL = 32: column sets sharing the 16 column slots (in different k-octets)
  constexpr int RB = 16 / (RS * CS);         // rows per row block of the A tile
  static_assert(MT == 1 || MT == 2 || MT == 4 || MT == 8 || MT == 16, "rows per pass");
  static_assert(MT <= RB, "rows per pass exceed the row block");
  // NW is the LARGEST workgroup this build is launched with (it sizes LDS and the register budget);
  // the launcher picks the actual wave count so that K fills whole sweeps (K = 5120: 10 waves)
  constexpr int CN = NL * CPL;           // columns per workgroup
  constexpr int KLS = UB * 16 + 4;       // dwords per (row, k-slot) activation slot (+4: bank spread)
  constexpr int ZERO_DW = CS > 1 ? KLS : 0;         // NL = 32: an all-zero slot for the lanes that feed the other column set
  constexpr int ACT_DW = MT * LK * KLS + ZERO_DW;  // staged activation dwords per wave
  // Activation staging: a task is one chunk of CHK consecutive k (the span inside which the
  // unpacker pairs fields: a packed word for 2/4/8 bit), 16*CPT bytes, one lane per task.  Every
  // VMEM instruction costs the CU's address pipeline the same whatever its width, so the slice is
  // fetched with as few wave-instructions as possible (one 16-byte load per wave at M = 1).
  constexpr int CHK = Unpack<BITS>::kChunkK;  // k per chunk
  constexpr int CPT = CHK / 8;               // 16-byte loads / LDS stores per task
  constexpr int NTASK = MT * LK * UB * (32 / CHK);
  // 8-16 rows per pass: the slice is 8-16 loads per lane and the CU's vector-memory path moves 64 B a clock, so
  // fetching it whole in front of the first weight rows holds the HBM stream back by ~0.85 us (per-wave timelines,
  // tools/gemv_trace.py --variants 189: first weight row out at 2.1 us against 0.7 at one row).  Measurement builds
  // cut the slice into parts of JP k-blocks of every k-slot, part 0 in front of the first weight rows, part 1
  // behind them, part p staged in front of its first step: 16384 = JP 2 (128 contiguous bytes per row and k-slot),
  // 8192 = JP 1 (64 B).  Both measured SLOWER than the whole slice up front (4096x11008 M=16, one box: 9.16 and
  // 9.8 us against 8.92; profiles/r02/sweep_16_row_pass_*_staging.txt): every part boundary is one more full wait
  // on the in-order load queue, and that costs more than the later start of the stream.
  constexpr int JP = (MT >= 8 && UB == 4) ? ((ABLATE & 8192) ? 1 : (ABLATE & 16384) ? 2 : UB) : UB;
  constexpr bool PIPE_ACT = JP < UB;
  constexpr int PARTS = UB / JP;
  constexpr int TB = NTASK / PARTS;        // tasks per part
  constexpr int ROUNDS = (TB + 63) / 64;   // rounds per part

  // 16-row passes: the reduction scratch reuses the activation region (one extra barrier), which
  // is what lets 4 k-blocks per lane fit in the 160 KB of LDS (139 KB of activations at NW = 8)
  constexpr bool ALIAS = MT >= 16;
  constexpr int RED_DW = NW * RS * MT * CN;
  constexpr int LDS_DW = ALIAS ? (NW * ACT_DW > RED_DW ? NW * ACT_DW : RED_DW) : NW * ACT_DW + RED_DW;
  __shared__ __attribute__((aligned(16))) uint32_t s_lds[LDS_DW];
  uint32_t* const s_act = s_lds;
  float* const s_red = reinterpret_cast<float*>(s_lds + (ALIAS ? 0 : NW * ACT_DW));

  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, kl = lane / NL, nl = lane % NL;
  const int nwaves = (int)(blockDim.x >> 6);  // <= NW
  const uint32_t T = (uint32_t)(LK * nwaves);  // k-batches per workgroup sweep
  const uint32_t split = SPLIT ? blockIdx.x % p.splits : 0u;
  const uint32_t tile_g = SPLIT ? blockIdx.x / p.splits : xcd_tile(blockIdx.x, p.tiles);  // tile of the whole grid
  const uint32_t tile = tile_g, n_ = p.n, zw_ = p.zw;
  const uint32_t* const qweight_ = p.qweight;
  const uint16_t* const scales_ = p.scales;
  const uint32_t* const qzeros_ = p.qzeros;
  uint16_t* const out_ = p.out;
  const uint32_t b_begin = SPLIT ? split * p.bps : 0u;
  const uint32_t b_end = SPLIT ? (b_begin + p.bps < p.batches ? b_begin + p.bps : p.batches) : p.batches;
  const uint32_t k_end = SPLIT ? (b_end * (uint32_t)(UB * 32) < p.k ? b_end * (uint32_t)(UB * 32) : p.k) : p.k;  // activations beyond: 0
  const uint32_t col = tile * CN + (uint32_t)(CPL * nl);
  const bool col_ok = GUARD ? col < n_ : true;  // n % CPL == 0 is NOT required: stores are per column
  const uint32_t m0 = blockIdx.y * MT;
  uint32_t* const my_act = s_act + wv * ACT_DW;
  // A fragment of this lane = row (lane & 15), k-octet (lane >> 4) of the MFMA's A tile.
  //   NL = 16: row = nl, octet = kl: the activations of row nl % MT for k-slot kl.
  //   NL = 8:  row = (kl & 1)*8 + nl, octet = kl >> 1: k-slots 2g and 2g+1 share octet g but sit in column
  //            slots 0-7 / 8-15; feeding them to row blocks 0-7 / 8-15 keeps their sums apart
  //            (D[ks*8 + m][ks*8 + nl]; the off-diagonal quadrants are don't-care).
  //   NL = 32: row = nl & 15, octet = 2*kl + (nl >> 4): column sets nl < 16 and nl >= 16 share the column
  //            slots but sit in octets of different parity; row block rb must see ZERO in the octets of
  //            the other set, so D[rb*8 + m][slot] sums column set rb only.
  const int arow = (CS > 1 ? (nl & 7) : nl) % MT;  // activation row this lane feeds to its row block
  int a_off = (kl * MT + arow) * KLS;  // slot order [k-slot][row]: the 16 lanes one ds_read_b128 pass serves then hit 16 different slots (no bank conflict at 8-16 rows)
  if constexpr (CS > 1) {
    if ((nl >> 4) != ((nl >> 3) & 1)) a_off = MT * LK * KLS;  // the zero slot
#pragma unroll
    for (int i = lane; i < ZERO_DW; i += 64) my_act[MT * LK * KLS + i] = 0u;
  }
  const typename Unpack<BITS>::Consts uk = Unpack<BITS>::consts();
  const uint32_t sub_mask = opaque_sgpr(Unpack<BITS>::kSubMask2);
  const uint32_t row_bytes = n_ * 4u;  // one packed row

  const rsrc_t r_w = make_rsrc(qweight_, p.kw * row_bytes);
  const rsrc_t r_s = make_rsrc(scales_, p.groups * n_ * 2u);
  const rsrc_t r_z = make_rsrc(qzeros_, p.groups * zw_ * 4u);
  const rsrc_t r_a = make_rsrc(p.a, p.m * p.k * 2u);

  f4_t acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = f4_t{0.f, 0.f, 0.f, 0.f};
  uint64_t stamp[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (ABLATE & 8) stamp[0] = __builtin_amdgcn_s_memrealtime();

  for (uint32_t b0 = b_begin; b0 < b_end; b0 += T) {
    const uint32_t wb = b0 + wv * LK;  // first batch of this wave in this sweep
    const uint32_t bt = wb + kl;       // this lane's batch
    const bool active = GUARD ? (col_ok && bt < b_end) : true;

    // (3) all packed-weight loads of this lane's batch: one 32-bit lane offset, the row step rides in
    //     the scalar offset.  Rows past the end of qweight read 0 (their activations are 0 as well).
    RowLoad<CPL, NT> w[UB][BITS];
    uint32_t lane_off = (bt * (uint32_t)(UB * BITS)) * row_bytes + col * 4u;
    if (GUARD && !active) lane_off = kOob;
    using LS = LoadSchedule<BITS, UB, LA>;
    auto issue_range = [&](auto from_c, auto to_c) {
      constexpr int from = decltype(from_c)::value, to = decltype(to_c)::value;
      static_for<(to > from ? to - from : 0)>([&](auto d) {
        constexpr int idx = from + decltype(d)::value, j = idx / BITS, i = idx % BITS;
        w[j][i].load(r_w, lane_off, (uint32_t)idx * row_bytes);
      });
    };
    if constexpr (WFIRST) {
      // weights first: with 8-16 rows per pass the activation slice alone is 8-16 loads per lane, and
      // issued in front of the weights it holds back the start of the HBM stream by about a microsecond
      issue_range(std::integral_constant<int, 0>{}, std::integral_constant<int, LS::issued_before(0)>{});
      __builtin_amdgcn_sched_barrier(0);
    }

    // (1) activation slice of this wave (oldest loads in the vmcnt queue).  Rows >= M fall off the
    //     end of the buffer and read 0; k >= K must be forced out of range (it would alias row m+1).
    u32x4_t hraw[PARTS][ROUNDS][CPT];
    // task t of part `part`: chunk `cib` of k-block `blk` (of the wave's LK * UB) of row `mi`
    auto act_task = [&](int part, int r, int& mi, int& blk, int& cib) {
      const int t = (lane + 64 * r) % TB;
      constexpr int per = LK * JP * (32 / CHK);  // tasks of one row: LK k-slots x JP k-blocks x chunks (JP == UB: the whole slice)
      mi = t / per;
      blk = ((t % per) / (JP * (32 / CHK))) * UB + part * JP + (t % (JP * (32 / CHK))) / (32 / CHK);
      cib = t % (32 / CHK);
    };
    auto act_issue = [&](auto part_c) {
      constexpr int part = decltype(part_c)::value;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        int mi, blk, cib;
        act_task(part, r, mi, blk, cib);
        const uint32_t k0 = wb * (uint32_t)(UB * 32) + (uint32_t)(blk * 32 + cib * CHK);
        uint32_t aoff = ((m0 + mi) * p.k + k0) * 2u;
        if (GUARD && (k0 >= k_end || m0 + mi >= p.m)) aoff = kOob;  // k % 8 == 0: chunks are all-in or all-out
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          uint32_t o = aoff + 16u * q;
          if (GUARD && CPT > 1 && k0 + 8u * q >= k_end) o = kOob;
          hraw[part][r][q] = __builtin_amdgcn_raw_buffer_load_b128(r_a, o, 0, 0);
        }
      }
    };
    act_issue(std::integral_constant<int, 0>{});

    // (2) scale and zero-point of the batch's group (one group per batch on this path); a group
    //     index past the end reads scale 0, which makes every weight of that lane exactly 0
    const uint32_t gi = (bt * (uint32_t)(UB * 32)) / p.groupsize;
    uint32_t sbits[CPL];
    {
      // CPL fp16 scales starting at half index h0: one 4- or 8-byte load from the enclosing dword(s)
      const uint32_t h0 = gi * n_ + col;
      uint32_t soff = (h0 & ~1u) * 2u;
      if (GUARD && !active) soff = kOob;
      if constexpr (CPL == 2) {  // col and n are even: h0 is even
        const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r_s, soff, 0, 0);
        sbits[0] = v & 0xFFFFu;
        sbits[1] = v >> 16;
      } else {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r_s, soff, 0, 0);
        const uint32_t sh = (h0 & 1u) * 16u;  // CPL == 4: col % 4 == 0 and n even, so h0 is even and sh == 0
        const uint32_t lo = __builtin_amdgcn_alignbit(v.y, v.x, sh), hi = v.y >> sh;
        sbits[0] = lo & 0xFFFFu;
        sbits[1] = lo >> 16;
        sbits[2] = hi & 0xFFFFu;
        if constexpr (CPL == 4) sbits[3] = hi >> 16;
      }
    }
    uint32_t zwin;
    {
      const uint32_t bit = col * BITS, sh = bit & 31u;
      const uint32_t zoff = (gi * zw_ + (bit >> 5)) * 4u;
      if constexpr (CPL * BITS <= 32 && (32 % (CPL * BITS)) == 0) {
        zwin = __builtin_amdgcn_raw_buffer_load_b32(r_z, zoff, 0, 0) >> sh;  // the CPL fields never straddle a word
      } else {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r_z, zoff, 0, 0);  // a read past the row end is harmless
        zwin = __builtin_amdgcn_alignbit(v.y, v.x, sh);
      }
    }

    if constexpr (!WFIRST) issue_range(std::integral_constant<int, 0>{}, std::integral_constant<int, LS::issued_before(0)>{});
    if constexpr (PARTS > 1) act_issue(std::integral_constant<int, 1>{});  // behind the first weight rows: the HBM stream starts first
    // Pin the issue order act -> scale/zero -> weights (WFIRST: first weights -> act -> scale/zero): each
    // wait below then leaves every younger load in flight (vmcnt retires in order).
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABLATE & 8) {
      if (b0 == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();  // all loads of the first sweep issued
      __builtin_amdgcn_sched_barrier(0);
    }

    // (4) permute each chunk into pair order and park it in this wave's LDS region (every lane
    //     stores; lanes beyond the part's tasks repeat a task with identical data)
    auto act_stage = [&](auto part_c) {
      constexpr int part = decltype(part_c)::value;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        int mi, blk, cib;
        act_task(part, r, mi, blk, cib);
        uint32_t h[CHK / 2];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          h[4 * q + 0] = hraw[part][r][q].x;
          h[4 * q + 1] = hraw[part][r][q].y;
          h[4 * q + 2] = hraw[part][r][q].z;
          h[4 * q + 3] = hraw[part][r][q].w;
        }
        uint32_t* dst = my_act + ((blk / UB) * MT + mi) * KLS + (blk % UB) * 16 + cib * (CHK / 2);
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          uint32_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // pair (4q+e) of this chunk; pair_k is periodic in the chunk, so chunk 0's table serves all
            const int pr = 4 * q + e;
            const int klo = Unpack<BITS>::pair_k(pr, 0), khi = Unpack<BITS>::pair_k(pr, 1);
            static_assert(Unpack<BITS>::pair_k(CHK / 2 - 1, 1) < CHK, "pairs stay inside a chunk");
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
    };
    act_stage(std::integral_constant<int, 0>{});
    if constexpr (ABLATE & 8) {
      if (b0 == 0) stamp[2] = __builtin_amdgcn_s_memrealtime();  // activations staged
      __builtin_amdgcn_sched_barrier(0);
    }

    // (5) per-column dequant constants.  The subnormal form (xbit_device.h) needs the scale times
    //     2^(14 + BITS) to be finite in fp16; if any lane of the wave holds a scale too large for
    //     that (or NaN/Inf) the whole wave takes the spliced form for this batch - same W, bit for bit.
    half2_t s2[CPL], nzs2[CPL], ss2[CPL];
    bool sub_ok = true;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const _Float16 s = as_h((uint16_t)sbits[c]);
      const uint32_t z = ((zwin >> (c * BITS)) & Unpack<BITS>::kMask) + p.bias;
      s2[c] = splat(s);
      nzs2[c] = splat(neg_zero_times_scale(z, s));
      bool ok;
      ss2[c] = splat(sub_scale<BITS>(s, ok));
      sub_ok = sub_ok && ok;
    }
    const bool fast = (ABLATE & 4) ? false : __builtin_amdgcn_ballot_w64(!sub_ok) == 0;  // wave-uniform

    // (6) unpack + dequantise on the VALU, multiply-accumulate on the matrix pipe; the loads of
    //     the steps LA ahead are issued in front of each step
    auto run_steps = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      static_for<UB * 4>([&](auto step_c) {
        constexpr int step = decltype(step_c)::value, j = step / 4, oct = step % 4;
        if constexpr (PIPE_ACT && oct == 0 && j >= JP && j % JP == 0 && j / JP + 1 < PARTS) {
          act_issue(std::integral_constant<int, (j / JP + 1 < PARTS ? j / JP + 1 : 0)>{});  // the part after this one goes out one part ahead of its use
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LA > 0 && step > 0) {
          if constexpr (WAITFIRST && LS::issued_before(step) > LS::issued_before(step - 1)) {
            // wait for the rows this step consumes BEFORE issuing the next ones: the new loads then go out the moment
            // the data lands instead of one step's arithmetic later (a touch of the last needed row makes the
            // compiler place its s_waitcnt here)
            constexpr int last = LS::need(step) - 1;
            asm volatile("" ::"v"(w[last / BITS][last % BITS].v[0]));
            __builtin_amdgcn_sched_barrier(0);
          }
          issue_range(std::integral_constant<int, LS::issued_before(step - 1)>{}, std::integral_constant<int, LS::issued_before(step)>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PIPE_ACT && oct == 0 && j >= JP && j % JP == 0) {
          act_stage(std::integral_constant<int, (j / JP < PARTS ? j / JP : 0)>{});  // waits for this part only: the younger weight rows stay in flight
          __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t wc[CPL][BITS];
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int i = 0; i < BITS; ++i) wc[c][i] = w[j][i].v[c];

        const u32x4_t araw = *reinterpret_cast<const u32x4_t*>(my_act + a_off + j * 16 + 4 * oct);
        const h8_t a_frag = __builtin_bit_cast(h8_t, araw);
        u32x4_t braw[CPL];
        static_for<4>([&](auto e_c) {
          constexpr int e = decltype(e_c)::value;
          constexpr int pr = 4 * oct + e;
          uint32_t Q[CPL];
          if constexpr (FAST) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) Q[c] = Unpack<BITS>::template sub_pair<pr>(wc[c], sub_mask);
#pragma unroll
            for (int c = 0; c < CPL; ++c) braw[c][e] = as_u32(__builtin_elementwise_fma(as_h2(Q[c]), ss2[c], nzs2[c]));
          } else {
#pragma unroll
            for (int c = 0; c < CPL; ++c) Q[c] = Unpack<BITS>::template pair<pr>(wc[c], uk);
#pragma unroll
            for (int c = 0; c < CPL; ++c) braw[c][e] = as_u32(dequant_pair<Unpack<BITS>::pair_exp(pr)>(Q[c], s2[c], nzs2[c]));
          }
        });
        if constexpr (ABLATE & 16) {  // measurement: the same W pairs accumulated on the VALU (v_dot2_f32_f16), M = 1
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c][0] = __builtin_amdgcn_fdot2(as_h2(braw[c][e]), as_h2(araw[e]), acc[c][0], false);
        } else {
#pragma unroll
          for (int c = 0; c < CPL; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_frag, __builtin_bit_cast(h8_t, braw[c]), acc[c], 0, 0, 0);
        }
        if constexpr (LA > 0) __builtin_amdgcn_sched_barrier(0);
      });
    };
    if (fast) run_steps(std::true_type{});
    else run_steps(std::false_type{});
    __builtin_amdgcn_wave_barrier();  // keep the next sweep's LDS writes behind these reads
  }
  if constexpr (ABLATE & 8) {
    asm volatile("" : "+v"(acc[0]));
    __builtin_amdgcn_sched_barrier(0);
    stamp[3] = __builtin_amdgcn_s_memrealtime();  // last MFMA issued (its operands were ready)
    __builtin_amdgcn_sched_barrier(0);
  }

  if constexpr (ABLATE & 16) {
    // VALU build: sum the four k-slot lane groups by shuffles, then present the result where the
    // MFMA build has it (row 0 of the D tile = register 0 of lanes 0-15)
    static_assert(!(ABLATE & 16) || (NL == 16 && MT == 1), "VALU measurement build");
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      float v = acc[c][0];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[c][0] = v;
    }
  }
  if constexpr (ALIAS) __syncthreads();  // every wave is done reading its activations
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
        if constexpr (CS > 1) {
          if (m < MT) s_red[(wv * MT + m) * CN + ((drow / RB) * 16 + slot) * CPL + c] = acc[c][r];
        } else {
          if (drow / RB == ks && m < MT) s_red[((wv * RS + ks) * MT + m) * CN + snl * CPL + c] = acc[c][r];
        }
      }
  }
  __syncthreads();
  if constexpr (ABLATE & 8) stamp[4] = __builtin_amdgcn_s_memrealtime();  // past the workgroup barrier
  for (int e = tid; e < MT * CN; e += 64 * nwaves) {
    const int m = e / CN, c = e % CN;
    float sum = 0.f;
    if constexpr (ABLATE & 32768) {  // measurement: compile-time bound, predicated - every read goes out at once.  Measured SLOWER
#pragma unroll                         // (4096x11008: 6.80 against 6.66 us at 1 row, 9.45 against 8.92 at 16; sweep_16_row_pass_two_part_staging.txt)
      for (int v = 0; v < NW * RS; ++v)
        if (v < nwaves * RS) sum += s_red[(v * MT + m) * CN + c];
    } else if constexpr (ABLATE & 65536) {  // measurement: every slot read unconditionally (inside the scratch), absent waves' values deselected
#pragma unroll
      for (int v = 0; v < NW * RS; ++v) {
        const float x = s_red[(v * MT + m) * CN + c];
        sum += v < nwaves * RS ? x : 0.f;
      }
    } else {
      for (int v = 0; v < nwaves * RS; ++v) sum += s_red[(v * MT + m) * CN + c];
    }
    const uint32_t n_out = tile * CN + c, row = m0 + m;
    const bool valid = !GUARD || (n_out < n_ && row < p.m);
    if constexpr (SPLIT) {
      // Cross-workgroup reduction in ONE trip to the device coherence point: the partial sum is
      // added to the element's 64-bit accumulator as a fixed-point integer (2^-28 units, biased
      // to stay positive) by a returning atomic whose upper bits count the arrivals.  Integer
      // addition is associative, so the total does not depend on the arrival order; 2^-28 is below
      // half the smallest fp16 spacing.  The thread that completes the count converts and stores,
      // then clears the slot for the next launch.  Layout: bits 0-51 sum of (v + 2^47), 52-55
      // arrivals, 56-59 arrivals with a partial sum >= 2^19 or +Inf, 60-63 with one <= -2^19 or -Inf
      // (a NaN counts in both): the result is then +Inf, -Inf or, with both kinds present, NaN.
      // Contract (tests: test_split_k_overflow..., test_split_k_cancellation...): while every K-range
      // partial sum stays inside +-2^19 (8 x the largest finite fp16) the split result is the exact
      // sum of the partials rounded once - cancellation between ranges included.  A partial outside
      // that range saturates: the output is +-Inf (NaN if both signs saturate) even where the unsplit
      // fp32 sum would have cancelled back into range - a case that needs |W * a| ~ 10^2 per term over
      // thousands of terms with an fp16-representable total.  Partials are quantised to 2^-28
      // (<= 15 * 2^-29 of error in total: below half an ulp of every normal fp16, a quarter of the
      // smallest subnormal's spacing).
      if (valid) {
        unsigned long long* slot = p.acc + ((size_t)blockIdx.y * p.tiles + tile_g) * (size_t)(MT * CN) + e;
        const bool fin = __builtin_fabsf(sum) < 524288.0f;  // 2^19; false for NaN / Inf too
        const long long v = fin ? (long long)__builtin_rintf(sum * 268435456.0f) : 0ll;  // |v| < 2^47
        const unsigned long long pos = (!fin && !(sum < 0.f)) ? (1ull << 56) : 0ull;  // +huge, +Inf, NaN
        const unsigned long long neg = (!fin && !(sum > 0.f)) ? (1ull << 60) : 0ull;  // -huge, -Inf, NaN
        const unsigned long long add = (unsigned long long)(v + (1ll << 47)) + (1ull << 52) + pos + neg;
        const unsigned long long old = __hip_atomic_fetch_add(slot, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((old >> 52) & 15ull) + 1ull == p.splits) {
          const unsigned long long tot = old + add;
          const long long fixed = (long long)(tot & ((1ull << 52) - 1ull)) - (long long)p.splits * (1ll << 47);
          float y = (float)fixed * (1.0f / 268435456.0f);
          const bool any_pos = ((tot >> 56) & 15ull) != 0, any_neg = (tot >> 60) != 0;
          if (any_pos || any_neg) y = (any_pos && any_neg) ? __builtin_nanf("") : (any_pos ? __builtin_inff() : -__builtin_inff());
          out_[(size_t)row * n_ + n_out] = as_u16((_Float16)y);
          __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else {
      if (valid) out_[(size_t)row * n_ + n_out] = as_u16((_Float16)sum);
    }
  }
  if constexpr (ABLATE & 8) {
    stamp[5] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && p.trace != nullptr) {
      uint64_t* o = p.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wv) * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i) o[i] = stamp[i];
      // where the wave ran: XCC_ID (hwreg 20) above HW_ID (hwreg 4: wave, simd, cu, sh, se ids); slot 6 as in xbit_gemv_dyn.h
      o[6] = stamp[1];
      o[7] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
  }
}

// row loads in flight ahead of the MFMA step being computed (see LA); measured per width on
// 4096x11008 (profiles/r01/sweep_lookahead_by_width.txt): 4 for 2/3/4/8 bit, 6 for 6 bit
constexpr int kLookahead = 4;
template <int BITS>
constexpr int lookahead_for() {
  return (BITS >= 5 && BITS <= 7) ? 6 : kLookahead;
}

// Waves per workgroup for `batches` k-batches at `lk` k-slots per wave: one sweep if `cap` waves
// cover K, otherwise the fewest equal sweeps - so K = 5120 (40 batches) runs 10 waves x 1 sweep or
// 5 x 2 instead of 8 waves with a second sweep that is three-quarters idle.  At least 4 waves.
inline uint32_t sweep_waves(uint32_t batches, uint32_t lk, uint32_t cap) {
  const uint32_t w = (batches + lk - 1) / lk;
  if (w <= cap) return w < 4u ? 4u : w;
  const uint32_t sweeps = (w + cap - 1) / cap;
  return (w + sweeps - 1) / sweeps;
}

template <int BITS, int UB, int CPL, int NL, int NW, int MT, bool NT, int LA = kLookahead>
int launch_one(GemvParams p, hipStream_t stream, uint32_t waves = NW) {
  constexpr int CN = NL * CPL;
  p.tiles = (p.n + CN - 1) / CN;
  const uint32_t kblocks = (p.k + 31) / 32;
  p.batches = (kblocks + UB - 1) / UB;
  dim3 grid(p.tiles, (p.m + MT - 1) / MT), block(64 * (waves < (uint32_t)NW ? waves : (uint32_t)NW));
  // One build per launch shape, edge handling always on: the unguarded build measured the same
  // (6.7 vs 6.8 us on 4096x11008) and doubled the number of kernels to compile.
  return launch(gemv_mfma_kernel<BITS, UB, CPL, NL, NW, MT, NT, true, 0, LA>, grid, block, stream, p);
}

// tuning variants are M = 1 launches of one shape (tools/gemv_sweep.py)
template <int BITS, int UB, int CPL, int NL, int NW, bool NT, int LA = lookahead_for<BITS>()>
int launch_m1(const GemvParams& p, hipStream_t stream) {
  return launch_one<BITS, UB, CPL, NL, NW, 1, NT, LA>(p, stream);
}

// split-K launch (see SPLIT): grid.x = tiles * splits
template <int BITS, int UB, int CPL, int NL, int NW, int MT>
int launch_split(GemvParams p, hipStream_t stream, uint32_t splits, const GemvWorkspace& ws, uint32_t waves) {
  constexpr int CN = NL * CPL;
  p.tiles = (p.n + CN - 1) / CN;
  p.batches = ((p.k + 31) / 32 + UB - 1) / UB;
  p.splits = splits;
  p.bps = (p.batches + splits - 1) / splits;
  p.acc = ws.acc;
  dim3 grid(p.tiles * splits, (p.m + MT - 1) / MT), block(64 * (waves < (uint32_t)NW ? waves : (uint32_t)NW));
  return launch(gemv_mfma_kernel<BITS, UB, CPL, NL, NW, MT, false, true, 0, lookahead_for<BITS>(), true>, grid, block, stream, p);
}

template <int BITS, int UB, int CPL, int NL, int NW>
int launch_split_rows(const GemvParams& p, hipStream_t stream, uint32_t splits, const GemvWorkspace& ws) {
  const uint32_t batches = ((p.k + 31) / 32 + UB - 1) / UB, bps = (batches + splits - 1) / splits;
  const uint32_t waves = sweep_waves(bps, 64 / NL, NW);
  if (p.m == 1) return launch_split<BITS, UB, CPL, NL, NW, 1>(p, stream, splits, ws, waves);
  return launch_split<BITS, UB, CPL, NL, NW, 4>(p, stream, splits, ws, waves);
}

// Split-K plan.  Measured on MI355X (profiles/r01/sweep_split_k_v2.txt): with the one-atomic reduction
// the tail costs ~1.5 us (the earlier store -> ticket -> load protocol cost ~3 us), so splitting pays
// when N is small (<= 144 tiles of 32 columns) and K >= 8192: 11008x4096 8.6 -> 7.9 us, 14336x4096
// 10.4 -> 8.7, 28672x4096 17.6 -> 13.9, 8192x3584 6.8 -> 6.1; not at K = 4096 (4096x4096 4.5 -> 5.7)
// nor at N >= 8192 (8192x8192 9.3 -> 9.5).  Choice of tile width and split count: fill ~256
// workgroups of 4-8 waves with k-ranges that fill whole sweeps (sweep_waves); score = wave
// utilisation x sqrt(CU fill), 32-column tiles discounted by 0.85 against 64-column ones (256-byte row
// segments): at equal split counts the wide tiles win whenever they still give >= ~190 workgroups
// (11008x4096 s3: 7.9 vs 8.5 us), the narrow ones when they do not (8192x4096 s2: 6.3 vs 7.5).
// splits == 0: do not split.
struct SplitPlan {
  int cpl = 0;
  uint32_t splits = 0;
};
// Waves of a split-K workgroup that fit on one CU: 4 SIMDs x (512 VGPRs / the build's allocation).  The wide-word
// unpackers are register-hungry (7-bit, 64-column tiles: 170 VGPRs = two waves per SIMD), and a plan whose workgroups
// do not all fit on the chip at once runs in rounds: 5-bit 11008x4096 took 14.6 us with nine K ranges of 5-wave
// workgroups (two per CU, 576 of them: two rounds) and 9.3 with eleven of 4 waves (three per CU: one round).
// VGPR counts of the SPLIT builds (tools/kernel_resources.py; rows: bits 2..8; columns: 64-column tiles at 1 / 4 rows,
// 32-column tiles at 1 / 4 rows); tests/test_capi_and_host.py::test_split_plan_occupancy_table checks them against the library.
constexpr uint16_t kSplitVgprs[7][4] = {{122, 124, 76, 80}, {145, 151, 89, 92}, {124, 134, 78, 84}, {158, 160, 92, 92},
                                        {144, 144, 84, 84}, {170, 172, 100, 100}, {132, 132, 78, 82}};
inline uint32_t split_waves_per_cu(int bits, int cpl, uint32_t m) {
  if (bits < 2 || bits > 8) return 16u;
  const uint32_t v = kSplitVgprs[bits - 2][(cpl == 4 ? 0 : 2) + (m > 1 ? 1 : 0)];
  return 4u * (512u / ((v + 7u) & ~7u));
}

inline SplitPlan plan_split(const GemvParams& p, int ub, int forced_splits, int bits = 0) {
  SplitPlan plan;
  if (p.m > 4 || forced_splits == 1) return plan;
  const uint32_t kblocks = (p.k + 31) / 32, batches = (kblocks + ub - 1) / ub;
  const uint32_t tiles32 = (p.n + 31) / 32;
  if (forced_splits > 0) {
    plan.cpl = 4;
    plan.splits = (uint32_t)forced_splits < 15u ? (uint32_t)forced_splits : 15u;
    if (plan.splits > batches / 8u) plan.splits = batches / 8u;
    if (plan.splits >= 2u) {  // no empty K ranges: 121 batches in 15 ranges of 9 leave the last one without work - 14 ranges then
      const uint32_t bps = (batches + plan.splits - 1u) / plan.splits;
      plan.splits = (batches + bps - 1u) / bps;
    }
    if (plan.splits < 2u) plan.splits = 0;
    return plan;
  }
  if (kblocks < 256u || tiles32 > 176u || (tiles32 > 144u && kblocks < 384u)) return plan;  // 8192x5120 loses (7.7 -> 8.3 us), 13824x5120 gains (13.0 -> 12.4)
  double best = 0.0;
  for (int cpl = 4; cpl >= 2; cpl -= 2) {
    const uint32_t tiles = (p.n + 16u * cpl - 1) / (16u * cpl);
    for (uint32_t sp = 2; sp <= 15u; ++sp) {
      const uint32_t bps = (batches + sp - 1) / sp;
      if (bps < 16u) break;
      const uint32_t waves = sweep_waves(bps, 4, 8), sweeps = (bps + 4u * waves - 1) / (4u * waves), wgs = tiles * sp;
      const double fill = wgs < 256u ? wgs / 256.0 : 1.0;
      const uint32_t per_cu = split_waves_per_cu(bits, cpl, p.m) / waves;  // workgroups resident per CU
      const uint32_t rounds = (wgs + 256u * (per_cu ? per_cu : 1u) - 1) / (256u * (per_cu ? per_cu : 1u));
      // 5-wave workgroups put two waves on one of the four SIMDs; the builds with more vector work per byte than
      // 4-bit at 4 k-blocks per lane run at that SIMD's pace (4-bit g64 11008x4096: 10.5-10.7 us with 5, 9 or 10 K
      // ranges - all 5 waves - against 7.9-8.8 with 3, 6, 7, 8 or 11; 5-bit and 3-bit alike; sweep_group_sizes.txt)
      const double simd = (waves == 5u && !(bits == 4 && ub == 4)) ? 0.75 : 1.0;
      const double score = (double)bps / (4.0 * waves * sweeps) * __builtin_sqrt(fill) * (cpl == 4 ? 1.0 : 0.85) * simd / rounds;
      if (score > best + 1e-9) {
        best = score;
        plan.cpl = cpl;
        plan.splits = sp;
      }
    }
  }
  return plan;
}

template <int BITS, int UB>
int launch_plan(const GemvParams& p, hipStream_t stream, const SplitPlan& plan, const GemvWorkspace& ws) {
  if (plan.cpl == 4) return launch_split_rows<BITS, UB, 4, 16, 8>(p, stream, plan.splits, ws);
  return launch_split_rows<BITS, UB, 2, 16, 8>(p, stream, plan.splits, ws);
}

// measurement-only launches (M = 1, plain loads): ablated or stamped builds, see ABLATE
template <int BITS, int UB, int CPL, int NL, int NW, int ABLATE, int LA>
int launch_la(GemvParams p, hipStream_t stream) {
  constexpr int CN = NL * CPL;
  p.tiles = (p.n + CN - 1) / CN;
  p.batches = ((p.k + 31) / 32 + UB - 1) / UB;
  dim3 grid(p.tiles, p.m), block(64 * NW);
  return launch(gemv_mfma_kernel<BITS, UB, CPL, NL, NW, 1, false, true, ABLATE, LA>, grid, block, stream, p);
}

// measurement-only launch of the 16-row pass (NL = 16, 8 waves) with ABLATE bits
template <int BITS, int UB, int CPL, int ABLATE, int MT = 16>
int launch_rows16_ablate(GemvParams p, hipStream_t stream) {
  constexpr int CN = 16 * CPL;
  p.tiles = (p.n + CN - 1) / CN;
  p.batches = ((p.k + 31) / 32 + UB - 1) / UB;
  dim3 grid(p.tiles, (p.m + MT - 1) / MT), block(64 * sweep_waves(p.batches, 4, 8));
  return launch(gemv_mfma_kernel<BITS, UB, CPL, 16, 8, MT, false, true, ABLATE, lookahead_for<BITS>()>, grid, block, stream, p);
}

// rows per pass: 1, 4, 8 or 16 (fewer staged rows = less LDS traffic); NL = 8 / 32 tiles take <= 8 rows.
// Workgroup size: up to 16 waves when there is at most one workgroup per CU (<= 240 tiles), up to 8
// otherwise (several 8-wave workgroups per CU sweeping K measured better: 8192x28672 26.2 -> 25.3 us,
// 8192x8192 9.8 -> 9.2); within that cap, as many waves as fill whole sweeps (sweep_waves).
// A 16-wave workgroup leaves each wave 128 VGPRs.  The builds that do not fit in that (they would spill: 3/5/6/7-bit
// with 64-column tiles, 7-bit with 48, and the 4-row passes of 4- and 8-bit with 64) are not compiled; those shapes
// run 8 waves over two sweeps - 7-bit 4096x11008: 9.8 us against 11.8 with the spilling 16-wave build
// (profiles/r02/sweep_wide_word_bits.txt).  tests/test_capi_and_host.py checks that nothing in the library spills.
template <int BITS, int CPL, int NL, int MT>
constexpr bool fits_16_waves() {
  if (XBIT_TUNING) return true;  // measurement builds keep every build (variant 120 caps the default path at 8 waves, 121 does not)
  if (NL != 16) return true;
  if (CPL == 4) return MT == 1 && (BITS == 2 || BITS == 4 || BITS == 8);
  if (CPL == 3) return BITS != 7;
  return true;
}

template <int BITS, int UB, int CPL, int NL, bool NT, int LA = lookahead_for<BITS>()>
int launch_rows(const GemvParams& p, hipStream_t stream, uint32_t wave_cap = 16u) {
  constexpr uint32_t LK = 64 / NL, CN = NL * CPL;
  const uint32_t batches = ((p.k + 31) / 32 + UB - 1) / UB, tiles = (p.n + CN - 1) / CN;
  if (p.m == 1) {
    constexpr bool fit = fits_16_waves<BITS, CPL, NL, 1>();
    const uint32_t waves = sweep_waves(batches, LK, (fit && tiles <= 240u && wave_cap >= 16u) ? 16u : 8u);
    if constexpr (fit) {
      if (waves > 8u) return launch_one<BITS, UB, CPL, NL, 16, 1, NT, LA>(p, stream, waves);
    }
    return launch_one<BITS, UB, CPL, NL, 8, 1, NT, LA>(p, stream, waves);
  }
  if (p.m <= 4) {
    constexpr bool fit = fits_16_waves<BITS, CPL, NL, 4>();
    const uint32_t waves = sweep_waves(batches, LK, (fit && tiles <= 240u && wave_cap >= 16u) ? 16u : 8u);
    if constexpr (fit) {
      if (waves > 8u) return launch_one<BITS, UB, CPL, NL, 16, 4, NT, LA>(p, stream, waves);
    }
    return launch_one<BITS, UB, CPL, NL, 8, 4, NT, LA>(p, stream, waves);
  }
  // 8-16 rows: the staged activation tile is 8-16x larger (139 KB of LDS at UB = 4, 8 waves, 16 rows).
  // Measured on 4096x11008 w4 M=16: 8 waves 9.0-9.3 us, 4 waves 11.0 us.
  if (p.m <= 8) return launch_one<BITS, UB, CPL, NL, 8, 8, NT, LA>(p, stream, sweep_waves(batches, LK, 8));
  return launch_one<BITS, UB, CPL, 16, 8, 16, NT, LA>(p, stream, sweep_waves(batches, 4, 8));
}

template <int BITS>
constexpr int max_ub() {
  return BITS <= 4 ? 4 : 2;  // <= 16 outstanding row loads per lane
}

template <int BITS, int UB, bool NT = false>
int launch_cpl(const GemvParams& p, hipStream_t stream, int cpl, uint32_t wave_cap = 16u) {
  switch (cpl) {
    case 4: return launch_rows<BITS, UB, 4, 16, NT>(p, stream, wave_cap);
    case 3: return launch_rows<BITS, UB, 3, 16, NT>(p, stream, wave_cap);
    default: return launch_rows<BITS, UB, 2, 16, NT>(p, stream, wave_cap);
  }
}

// ---- the launch rule: xbit_gemv_rules.h says WHICH launch, the functions below HOW --------------------------------------------------

// What a measurement build may override (xbit_gemv_tuning.h); the production library always passes the defaults.
struct RuleHooks {
  int cpl = 0;               // default tile width / 16 (0: pick_cpl(n))
  uint32_t wave_cap = 16u;   // waves per workgroup of the round-2 default path
  bool force_nt = false, force_plain = false;
  bool no_dyn = false;       // never the staged-operand kernel
  bool no_split = false;     // never split K
  bool no_fill_once = false; // never the fill-once split plan
  bool scored_on_dyn = false;  // the scored split plan always on the staged-operand kernel's split build
  int scored_cpl = 0;          // tile width / 16 of the scored plan (0: as planned)
};

template <int BITS>
rules::Call make_call(const GemvParams& p, const RuleHooks& hooks) {
  constexpr int UBMAX = max_ub<BITS>();
  rules::Call c;
  c.bits = BITS;
  c.m = p.m;
  c.k = p.k;
  c.n = p.n;
  c.groupsize = p.groupsize;
  c.kblocks = (p.k + 31u) / 32u;
  c.t32 = (p.n + 31u) / 32u;
  c.cpl = hooks.cpl ? hooks.cpl : pick_cpl(p.n);
  c.tiles = (p.n + 16u * (uint32_t)c.cpl - 1u) / (16u * (uint32_t)c.cpl);
  c.ubmax = UBMAX;
  c.flags = p.groupsize % (32u * UBMAX) == 0u ? rules::kGroupA : ((UBMAX > 2 && p.groupsize % 64u == 0u) ? rules::kGroupB : rules::kGroupC);
  if (c.cpl == 2 && c.t32 < 160u && c.kblocks >= 128u) c.flags |= rules::kSmallN;
  {
    const uint32_t t128 = (p.n + 127u) / 128u, filled = ((t128 + 255u) / 256u) * 256u;  // 128-column tiles, and the CU slots their rounds occupy
    if ((BITS == 4 || BITS == 8) && p.n >= 21504u && t128 * 3u >= filled * 2u && c.kblocks / (uint32_t)UBMAX >= 32u) c.flags |= rules::kWide;
  }
  if ((uint64_t)p.kw * p.n * 4u >= (96ull << 20) && p.n >= 16384u) c.flags |= rules::kBigWeights;
  return c;
}

// Act::FillOnceSplit.  FILL THE CHIP ONCE: 64-column tiles of the staged-operand kernel's split-K build times as many K ranges as give at
// most 256 workgroups (32-column tiles where that would take more than 15 ranges: N around 1024); where one round of workgroups would
// leave more than a fifth of the chip idle (N = 6144: 96 tiles x 2 = 192) two rounds are filled instead (x 5 = 480), with the
// one-round plan as the fallback when the K ranges of the two-round one are too short.  2..32 rows take the same plan through the
// split-K builds of the 4-, 8-, 16- and 32-row kernels (17..32 rows: from 3 bits up).  One row whose K ranges exceed the split build's staging rounds (K / splits > 4096 at
// group 128): the same plan on this file's split kernel (UB_ k-blocks per lane).
template <int BITS, int UB_>
int act_fill_once(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts, const rules::Call& c) {
  const uint32_t t64 = (p.n + 63u) / 64u;
  int fcpl = 4;
  uint32_t tiles = t64, fsplits = 256u / t64;
  if (fsplits > 15u) {  // narrow layers: 32-column tiles
    fcpl = 2;
    tiles = c.t32;
    fsplits = 256u / c.t32 > 15u ? 15u : 256u / c.t32;
  }
  uint32_t alt = 0;  // the one-round plan, tried if the two-round one is not eligible (K ranges too short)
  if (tiles * fsplits < 205u && 512u / tiles <= 15u && tiles * (512u / tiles) >= 410u) {
    alt = fsplits;
    fsplits = 512u / tiles;
  }
  for (int attempt = 0; attempt < 2 && fsplits >= 2u; ++attempt) {
    const uint32_t mt_rows = p.m > 16u ? 32u : (p.m > 8u ? 16u : (p.m > 4u ? 8u : (p.m > 1u ? 4u : 1u)));  // rows of the build that takes p.m
    const size_t slots = (size_t)tiles * (size_t)(16 * fcpl) * mt_rows;
    GemvWorkspace ws;
    if (gemv_workspace(stream, slots, opts, &ws)) {
      const int st = launch_gemv_dyn(p, BITS, stream, fcpl, 16, true, false, nullptr, fsplits, ws.acc);
      if (st != kGemvDynNotEligible) return st;
      if (p.m > 1u) break;  // several rows: no other kernel takes this plan
      const uint32_t batches = (c.kblocks + (uint32_t)UB_ - 1u) / (uint32_t)UB_;
      if (attempt == 0 && fsplits <= batches / 8u) {
        SplitPlan fp;
        fp.cpl = fcpl;
        fp.splits = fsplits;
        return launch_plan<BITS, UB_>(p, stream, fp, ws);
      }
    }
    fsplits = alt;
    alt = 0;
  }
  return kGemvDynNotEligible;
}

// Act::ScoredSplit (and a split the caller forces through options.split_k): plan_split() picks tile width and split count.  One row whose
// K ranges give every one of the staged-operand kernel's eight waves the same number of units, or many of them (units per range == 16 or
// >= 24: 8192x3584 w4 6.5 -> 6.1 us, 14336x4096 8.8 -> 8.1; profiles/r03/sweep_split_dyn.txt), runs that kernel's split build; the ranges
// of 15-18 units that the plans for K = 11008 / 13824 use run this file's (11008x4096 7.8 against 8.6).  Group size 64 (UB_ == 2 with
// max_ub == 4): this file's split kernel only.
template <int BITS, int UB_>
int act_scored_split(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts, const rules::Call& c, const RuleHooks& hooks) {
  SplitPlan plan = plan_split(p, UB_, opts.split, BITS);
  if (plan.splits < 2u) return kGemvDynNotEligible;
  if (hooks.scored_cpl) plan.cpl = hooks.scored_cpl;
  const uint32_t tiles = (p.n + 16u * plan.cpl - 1) / (16u * plan.cpl), mblocks = p.m == 1 ? 1u : (p.m + 3u) / 4u;
  const size_t slots = (size_t)mblocks * tiles * (size_t)((p.m == 1 ? 1 : 4) * 16 * plan.cpl);
  GemvWorkspace ws;
  if (!gemv_workspace(stream, slots, opts, &ws)) return kGemvDynNotEligible;
  if (UB_ == max_ub<BITS>() && p.m == 1u && !hooks.no_dyn) {
    const uint32_t dyn_units = (c.kblocks + (BITS == 2 ? 8u : 4u) - 1u) / (BITS == 2 ? 8u : 4u), dyn_ups = (dyn_units + plan.splits - 1u) / plan.splits;
    if (hooks.scored_on_dyn || dyn_ups == 16u || dyn_ups >= 24u) {
      const int st = launch_gemv_dyn(p, BITS, stream, plan.cpl, 16, true, false, nullptr, plan.splits, ws.acc);
      if (st != kGemvDynNotEligible) return st;
    }
  }
  return launch_plan<BITS, UB_>(p, stream, plan, ws);
}

// Act::DynChunks32: 33..512 rows in ONE launch over blocks of rows (xbit_gemv_rows.hip: which kernel).  The kernel whose column groups share one
// LDS image of the activations (xbit_gemv_rows.h: 3..8 bits, groups of a power-of-two multiple of 32): 4096x11008 w4 at 128 rows 40.4 -> 28.6 us, at 64
// rows 21.7 -> 18.6 (profiles/r04/sweep_rows_shared_tile_v1.txt).  The 32-row build of the staged-operand kernel, whose row blocks of a column tile run side by
// side and share its packed rows through L2: ~10 % ahead of one launch per 32 rows, and ahead of 16-row passes and of dequantise + dense GEMM up to 128
// rows (profiles/r04/sweep_rows_in_grid_y.txt, linear_crossover.txt).
template <int BITS>
int act_dyn_chunks(const GemvParams& p, hipStream_t stream, int cpl) {
  return launch_gemv_rows_or_dyn(p, BITS, stream, cpl);
}

// Act::MfmaDefault: this file's kernel on tiles of the default width.  Weight loads are plain by default: equal to nontemporal when the
// weights come from HBM and ~8 % faster when they are still in the Infinity Cache (4096x11008); very large, wide layers cannot stay cached
// anyway and stream ~9 % faster nontemporal (8192x28672: 25.8 -> 23.4 us; profiles/r01/sweep_nt_large_shapes.txt).
template <int BITS>
int act_mfma_default(const GemvParams& p, hipStream_t stream, const rules::Call& c, const RuleHooks& hooks) {
  constexpr int UBMAX = max_ub<BITS>();
  if (c.flags & rules::kGroupA) {
    if (hooks.force_nt || (!hooks.force_plain && (c.flags & rules::kBigWeights))) return launch_cpl<BITS, UBMAX, true>(p, stream, c.cpl, hooks.wave_cap);
    return launch_cpl<BITS, UBMAX>(p, stream, c.cpl, hooks.wave_cap);
  }
  if constexpr (UBMAX > 2) {
    if (c.flags & rules::kGroupB) return launch_cpl<BITS, 2>(p, stream, c.cpl);
  }
  return launch_cpl<BITS, 1>(p, stream, c.cpl);
}

template <int BITS>
int launch_by_rules(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts, const RuleHooks& hooks = RuleHooks()) {
  constexpr int UBMAX = max_ub<BITS>();
  const rules::Call c = make_call<BITS>(p, hooks);
  const bool group_a = (c.flags & rules::kGroupA) != 0u;
  uint32_t state = 0;
  if (opts.split >= 2 && !hooks.no_split && !(c.flags & rules::kGroupC)) {  // a split the caller asked for: no bucket decides
    const int st = group_a ? act_scored_split<BITS, UBMAX>(p, stream, opts, c, hooks) : act_scored_split<BITS, 2>(p, stream, opts, c, hooks);
    if (st != kGemvDynNotEligible) return st;
  }
  for (int i = 0; i < rules::kRuleCount; ++i) {
    const rules::Rule& r = rules::kRules[i];
    if (!rules::matches(r, c, state)) continue;
    int st = kGemvDynNotEligible;
    switch (r.act) {
      case rules::Act::MarkUnsplit:
        state |= rules::kNoFillOnce | rules::kNoScored;
        break;
      case rules::Act::MarkNoFillOnce:
        state |= rules::kNoFillOnce;
        break;
      case rules::Act::FillOnceSplit:
        if (opts.split != 0 || hooks.no_split || hooks.no_fill_once || hooks.no_dyn) break;
        st = group_a ? act_fill_once<BITS, UBMAX>(p, stream, opts, c) : act_fill_once<BITS, 2>(p, stream, opts, c);
        break;
      case rules::Act::ScoredSplit:
        if (opts.split != 0 || hooks.no_split) break;  // (a forced split was tried above)
        st = group_a ? act_scored_split<BITS, UBMAX>(p, stream, opts, c, hooks) : act_scored_split<BITS, 2>(p, stream, opts, c, hooks);
        break;
      case rules::Act::DynChunks32:
        if (!hooks.no_dyn) st = act_dyn_chunks<BITS>(p, stream, c.cpl);
        break;
      case rules::Act::DynWhole:
        if (!hooks.no_dyn) st = launch_gemv_dyn(p, BITS, stream, c.cpl, 16, true, false, nullptr);
        break;
      case rules::Act::DynNarrow:
        if (!hooks.no_dyn) st = launch_gemv_dyn(p, BITS, stream, 2, 8, true, false, nullptr);
        break;
      case rules::Act::Mfma16Col:
        // halve the tile width (NL = 8: 16 columns, the MFMA's column slots split into two k-slots) so twice as many CUs get work
        st = launch_rows<BITS, (UBMAX > 2 ? 2 : UBMAX), 2, 8, false>(p, stream);
        break;
      case rules::Act::Mfma128Col:
        // 32 lanes side by side, 16 bytes each: a wave-instruction reads 512 contiguous bytes of two packed rows (NL = 32: two column
        // sets share the MFMA's column slots and are kept apart by zeroed A operands, see the kernel); nontemporal loads
        if constexpr (BITS == 4 || BITS == 8) st = launch_rows<BITS, UBMAX, 4, 32, true>(p, stream);
        break;
      case rules::Act::MfmaDefault:
        st = act_mfma_default<BITS>(p, stream, c, hooks);
        break;
    }
    if (st != kGemvDynNotEligible) return st;
  }
  return XBIT_ERR_SHAPE;  // not reached: the last row takes every call
}

}  // namespace
}  // namespace xbit

#if XBIT_TUNING
#include "xbit_gemv_tuning.h"  // measurement variants (options.variant): libxbitops_hip_tuning.so only
#endif

namespace xbit {
namespace {

template <int BITS>
int launch_bits(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) {
#if XBIT_TUNING
  if (opts.variant > 0) return tuning::launch_variant<BITS>(p, stream, opts);
#endif
  return launch_by_rules<BITS>(p, stream, opts);
}

}  // namespace

}  // namespace xbit
