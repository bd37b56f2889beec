// xbit_gemv_dyn.h — A16Wx GEMV for 1 to 32 activation rows (decode, speculative and small-batch decode), gfx950, bits 2..8:
// whole-K column tiles, small operands staged once per workgroup by HALF of its waves while the other half already streams
// weights; builds for 1 / 4 / 8 / 16 / 32 rows, split-K, act-order and odd group sizes (template parameters below).
//
// Same packed layout, same arithmetic and the same lane map as xbit_gemv_mfma.h (NL lanes x CPL adjacent columns,
// 64 / NL k-slots per wave, a lane owns whole k-blocks, W rebuilt bit for bit in packed fp16, multiply-accumulate on
// v_mfma_f32_16x16x32_f16, row loads woven into the compute stream a few rows ahead).  What differs is the opening of
// the kernel, which is where a launch of this size loses most (profiles/r03/):
//
//   * A CU's vector-memory pipeline accepts one wave-level load every ~35-40 clocks whatever its width.  In
//     xbit_gemv_mfma.h every wave opens with three small loads (activation slice, scales, zero-points) and five row
//     loads: 64 instructions per CU, ~1 us before the last wave's first row is even requested
//     (trace_detail_v63.txt: waves 4-7 trail waves 0-3 by 0.5 us at "loads issued" and by 0.9 us at the end).
//   * Here K is cut into units of (64 / NL) * UB k-blocks that ANY wave can work on, because the small operands are
//     staged ONCE per workgroup: the whole activation vector (pair order, 2 B per k) and one (scale, -zero*scale)
//     entry per group and column go to LDS.  STW of the NW waves do that staging (the same three small loads per
//     thread, SI rounds of them); the other waves request their first rows at once.  One workgroup barrier separates
//     staging from use; every wave's first rows are in flight before it.  With the OLDER half staging (the
//     memory pipeline serves older waves first): 4096x11008 w4 6.98 -> 6.61 us per launch in a dependent chain, 6.40
//     with nontemporal row loads; any other choice of staging waves (younger half, 1, 2, 3, 5, 6 of 8) is slower
//     than all eight (lab_3.txt, lab_4.txt).
//
// Tried on this structure and not kept (tools/microbench/gemv_lab.hip at the commit that added this file,
// profiles/r03/lab_1.txt, lab_2.txt): handing the units to the waves at run time from an LDS counter (the waves of a
// workgroup then finish within 0.3 us of each other instead of 0.5, and the launch takes 0.5 us LONGER), a static
// 5 : 3 split between the older and the younger wave of each SIMD (spread 0.13 us, launch +0.1 us), s_setprio for the
// younger half (no change): a CU's time is set by its memory pipeline, not by its slowest wave.
//
// Arithmetic is exactly that of xbit_gemv_mfma.h: zs = fp16((z + bias) * s), W = fp16(fma(q, s, -zs)) through the
// subnormal form when every scale of the tile allows it, the spliced form otherwise (decided per workgroup), fp32
// accumulation in a fixed order - results are identical from launch to launch.
#pragma once

#include "xbit_device.h"
#include "xbit_host.h"

namespace xbit {

namespace dyn {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int CPL, bool NT>
struct Row {
  uint32_t v[CPL];
  __device__ __forceinline__ void load(rsrc_t r, uint32_t voff, uint32_t soff) {
    constexpr int aux = NT ? 2 : 0;
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

// Which rows of a unit (UB k-blocks of BITS rows per lane) an MFMA step touches, and when rows are requested.
// Step s = (k-block s / 4, octet s % 4).  Rows are numbered across the wave's consecutive units ("flat": 0..R-1 this
// unit, R..2R-1 the next, 2R.. the one after); after step s the rows [need(s) + LA, need(s + 1) + LA) go out, so LA + 1
// row loads are in flight when a step's last row is waited for.
template <int BITS, int UB, int LA>
struct Plan {
  static constexpr int R = UB * BITS;  // row loads per lane and unit
  static constexpr int S = UB * 4;     // MFMA steps per unit
  static constexpr int need(int s) {   // rows (of this unit onward) that must have landed for step s; s == S: first step of the next unit
    return s >= S ? R + need(s - S) : (s / 4) * BITS + Unpack<BITS>::rows_for_octet(s % 4);
  }
  static constexpr int first_row(int s) {  // lowest row step s reads (rows below it are dead once step s starts)
    if (s >= S) return R;
    const int j = s / 4, oct = s % 4;
    if (Unpack<BITS>::kPow2) return j * BITS + (oct * 8 * BITS) / 32;
    return j * BITS + (oct < 2 ? 0 : (16 * BITS) / 32);
  }
  static constexpr int open = need(0) + LA;  // rows requested before the first step
  static_assert(open <= 2 * R, "the opening burst stays inside two units");
  static_assert(need(S) + LA <= 3 * R, "requests run at most two units ahead");
};

struct DynParams {
  const uint32_t* qweight;  // [kw, n]
  uint32_t n, tiles, kw, units;  // units = ceil(k-blocks / (LK * UB)): wave w works on units w, w + NW, ...
  uint32_t tcols;           // columns a tile OWNS (<= NL * CPL, a multiple of CPL): tile t starts at column t * tcols; lanes whose columns lie beyond
                            // load nothing.  Narrower than the lane map when that makes the tile count fill the chip (28672 columns: 256 tiles of 112)
  const uint16_t* a;        // fp16 [m, k]
  const uint16_t* scales;   // fp16 [groups, n]
  const uint32_t* qzeros;   // [groups, zw]
  uint16_t* out;            // fp16 [m, n]
  uint32_t k, groups, zw, bias;
  uint32_t m;               // rows (<= MT)
  uint32_t tab_groups;      // rows of the LDS table: every group a k-slot of [0, LK * units) can name (>= groups; the extra rows are zeros)
  uint32_t gshift;          // log2(groupsize / (32 * UB)): k-slot batch t = unit * LK + kl belongs to group t >> gshift
  uint32_t act_bytes;       // LDS bytes of the staged activation vector: (units of one K range) * LK * UB * 64
  uint32_t splits, ups;     // split-K builds: K ranges per tile and units per range (whole-K: 1, units)
  unsigned long long* acc;  // split-K builds: one 64-bit accumulator per output element, zero between launches
  const int32_t* g_idx;     // act-order builds: the group of row k (k entries; values are clamped to [0, groups)); null there = the
                            // ids are computed, row k belongs to group k / gshift (gshift holds the GROUP SIZE in those launches):
                            // group sizes that are no multiple of 32 (16, 48, 80 ...), where a group may begin inside a k-block
  uint64_t* trace;          // stamped build only
  bool by_id;               // host side only: launch the act-order build
  uint32_t id_bytes;        // host side only: LDS bytes of the staged group ids (2 B per k of the range), 0 without by_id
  const GemvSets* sets;     // host side only: the weight sets of a fused launch (MULTI builds), else null
  uint32_t mt;              // host side only: rows of the build that takes this launch (1, 4, 8, 16, 32; >= m)
  uint32_t wide_probe;      // host side only, tuning builds: which measurement build of the 128-column tiles (0: the production one)
  uint32_t row_blocks;      // host side only: grid.y (blocks of 32 rows of the 32-row build), else 1
};

// Template parameters
//   BITS   weight width 2..8        UB   k-blocks per lane and unit (groupsize % (32 * UB) == 0)
//   CPL    adjacent columns per lane (2, 3, 4)      NL   lanes side by side along N (16: tiles of 16 * CPL columns; 8: the
//          MFMA's 16 column slots are 8 columns x 2 k-slots, kept apart by feeding their activations to row blocks 0-7
//          and 8-15 of the A operand, as in xbit_gemv_mfma.h)                       NW   waves per workgroup
//   SI     staging rounds (each staging thread handles SI activation chunks and SI table entries; their loads all go
//          out up front): the launcher picks the smallest of 2, 3, 4, 8 that covers the shape; the by-id and the 8- / 16- /
//          32-row builds exist for SI = 2 (4) only and stage what is left in a loop
//   STW    waves that stage the small operands (the first STW); the other NW - STW go straight to their weight rows
//   LA     row lookahead (see Plan)
//   FLAGS  2 = per-wave stamps to p.trace (measurement builds).  Ablations, compiled by tools/microbench/gemv_lab.hip ONLY (they do
//          not compute the GEMV; the library never instantiates them): 4 = the staging waves load no small operand (constants go
//          through the same staging code), 8 = no cross-wave sum (every wave stores its own partial sums), 16 = no unpack / MFMA
//          (the rows are loaded and waited for, nothing is computed).  32 = (MT = 1, NL = 16; tuning library, results equal) the
//          multiply-accumulate on the VALU - v_dot2_f32_f16 per weight pair, the four k-slots of a column summed by two
//          __shfl_xor at the end - instead of the MFMA: the north star's "no MFMA" design, for the A/B of DESIGN.md section 3
//   SPLIT  split-K: workgroup (tile, split) sums units [split * ups, (split + 1) * ups) only and stages only that range's
//          activations and groups; the partial sums meet in per-element 64-bit fixed-point accumulators exactly as in
//          xbit_gemv_mfma.h (one returning atomic per element and workgroup; the last arriver converts, stores and clears)
//   GIDX   groups by id: act-order ("desc_act", quant_cuda_kernel.cu:257-336 in the reference: the group of row k is g_idx[k]),
//          or - g_idx null - contiguous groups of a size that is no multiple of 32, ids computed as k / groupsize.  The ids
//          (16-bit) are staged in LDS next to the activations, the table holds EVERY group, and each weight pair looks
//          its two (scale, -zero*scale) entries up from its own rows' ids - 8 LDS reads and 2 v_perm per pair and
//          column more than the contiguous build, no change to the weight stream.
//   MT     activation rows per workgroup (1, 4, 8, 16 or 32; 32: two A operands - rows 0-15 and 16-31 - per unpacked B fragment, so
//          every weight is unpacked once for all 32 rows; more than 32 rows: blocks of 32 rows in grid.y of one launch.  A 64-row
//          build - four A operands, four waves per workgroup - measured the same as two row blocks, profiles/r04/sweep_rows_in_grid_y.txt,
//          and was dropped: above 32 rows a launch is bound by its activation stream, 0.31 us per row on 4096x11008, not by its weights): the MFMA's A operand carries row (lane & 15) % MT, the staged
//          activation image holds MT rows per octet, D rows 0..MT-1 are the results (the rest are copies).  1 and 4: the
//          staging waves put the whole K range of every row into LDS up front.  8 and 16 (2 / 4 KB of activations per unit -
//          the whole K range would be 64 / 128 KB at K = 4096): every wave fetches the activations of ITS next unit itself,
//          one unit ahead - the loads go out when a unit starts, in front of every row request that unit makes for the
//          next one, and are permuted into the wave's own 2-4 KB of LDS when the next unit starts; by then they are the
//          oldest loads in flight, so the wait costs nothing the unit's first rows would not have cost anyway.  No
//          workgroup barrier besides the one behind the scale table.
//   MULTI  fused launch: the grid covers the column tiles of up to kMaxMulti weight sets that share K, the bit-width, the group size
//          and the activation (q / k / v, gate / up); a workgroup looks its set up from its tile number (`sets`, passed by value)
//          and then runs exactly the code of the plain build on that set's tensors.
template <int BITS, int UB, int CPL, int NL, int NW, bool NT, int SI, int STW, int LA, int FLAGS = 0, bool SPLIT = false, bool GIDX = false, int MT = 1, bool MULTI = false>
__global__ void __launch_bounds__(64 * NW)
    gemv_dyn_kernel(const uint32_t* qweight, uint32_t n, uint32_t tiles, uint32_t kw, uint32_t units, const uint16_t* a, const uint16_t* scales,
                    const uint32_t* qzeros, uint16_t* out, uint32_t k, uint32_t groups, uint32_t zw, uint32_t bias, uint32_t m, uint32_t tab_groups, uint32_t gshift,
                    uint32_t act_bytes, uint32_t splits, uint32_t ups, unsigned long long* split_acc, const int32_t* g_idx, uint64_t* trace, uint32_t tcols,
                    const GemvSets sets) {
  // scalar arguments in order of urgency (not one struct), so that a build with kernel-argument preload could hand the
  // first ones over in SGPRs (measured: profiles/r03/lab_5_kernarg_preload.txt)
  DynParams p = {qweight, n, tiles, kw, units, tcols, a, scales, qzeros, out, k, groups, zw, bias, m, tab_groups, gshift, act_bytes, splits, ups, split_acc, g_idx, trace, GIDX, 0u, nullptr, (uint32_t)MT, 0u, 1u};
  using P = Plan<BITS, UB, LA>;
  constexpr int R = P::R, S = P::S;
  constexpr int LK = 64 / NL;   // k-slots per wave
  constexpr int CN = NL * CPL;  // columns per workgroup
  constexpr int HS = NL >= 16 ? 1 : 16 / NL;  // NL = 8: the two k-slot parities land in different rows / column slots of D
  constexpr bool WIDE = NL == 32;  // 32 lanes side by side (512 contiguous bytes per packed row and instruction at CPL = 4): the two 16-lane column
                                   // sets share the MFMA's column slots in k-octets of different parity; row block rb of the A operand carries
                                   // zeros in the other set's octets, so D[8 * rb + m][slot] sums column set rb only (as in xbit_gemv_mfma.h)
  constexpr bool STAMP = (FLAGS & 2) != 0;
  constexpr int ST = 64 * STW;  // staging threads
  constexpr int CHK = Unpack<BITS>::kChunkK;  // k per activation chunk (the span inside which the unpacker pairs fields)
  constexpr int CPT = CHK / 8;                // 16-byte loads per chunk
  constexpr uint32_t kOob = 0x80000000u;
  static_assert(NL == 32 || NL == 16 || NL == 8, "lanes along N");
  static_assert(!WIDE || (MT <= 4 && !GIDX), "128-column tiles: up to four rows, contiguous groups");
  static_assert(STW >= 1 && STW <= NW, "staging waves");
  if constexpr (MT == 32 && !SPLIT) {  // more than 32 rows in ONE launch: grid.y counts blocks of 32 rows; the row blocks of a column tile run side by side
                                     // on the chip and share the tile's packed rows through L2 (the weights leave HBM once)
    const uint32_t rb = blockIdx.y;
    p.a += (size_t)rb * 32u * p.k;
    p.out += (size_t)rb * 32u * p.n;
    p.m = p.m - rb * 32u < 32u ? p.m - rb * 32u : 32u;
  }
  static_assert(!GIDX || !SPLIT, "act-order launches keep K whole");
  static_assert(!MULTI || (!GIDX && !SPLIT && FLAGS == 0), "fused launches: contiguous groups, whole K");
  static_assert(MT == 1 || MT == 4 || MT == 8 || MT == 16 || MT == 32, "rows per launch");
  constexpr int RG = MT > 16 ? MT / 16 : 1;  // row groups of 16: one A operand each per unpacked B fragment (32 rows: 2)
  constexpr bool WAVEACT = MT >= 8;  // activations per wave and unit, pipelined (see MT)
  static_assert(!WAVEACT || (NL == 16 && FLAGS == 0), "8 / 16 / 32 rows: tiles of 16 lanes");
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  // LDS: [activations: act_bytes][act-order: group ids, 2 B per k: act_bytes][table: tab_groups * NL lanes * 16 B]
  //      [act-order: the same table with plain scales, for the spliced form][reduction: NW * HS * MT * CN floats][bad flags: 8 dwords]
  //      [128-column tiles: zero activations, UB * 16 * MT dwords]
  uint32_t* const s_act = s_dyn;
  uint32_t* const s_gid = s_act + p.act_bytes / 4u;
  const uint32_t id_dwords = GIDX ? p.ups * (uint32_t)((64 / NL) * UB * 16) : 0u;  // ids: 2 B per k of the range, whatever MT
  uint32_t* const s_tab = s_gid + id_dwords;
  uint32_t* const s_tab_plain = s_tab + (GIDX ? p.tab_groups * (uint32_t)(NL * 4) : 0u);
  // 32 rows: the reduction scratch (NW * 32 * CN floats: 48-64 KB) REUSES the activation region, which is larger and idle by then - one
  // more workgroup barrier in front of the epilogue buys the LDS that lets the by-id build take layers with many groups
  constexpr bool ALIAS = MT >= 32;
  float* const s_red_own = reinterpret_cast<float*>(s_tab_plain + p.tab_groups * (uint32_t)(NL * 4));
  float* const s_red = ALIAS ? reinterpret_cast<float*>(s_act) : s_red_own;
  uint32_t* const s_bad = reinterpret_cast<uint32_t*>(s_red_own + (ALIAS ? 0 : ((NW * HS * MT * CN + 3) & ~3)));
  uint32_t* const s_zero = s_bad + 8;  // 128-column tiles only: one unit's worth of zero activations (UB * 16 * MT dwords)

  const int tid = threadIdx.x, lane = tid & 63, kl = lane / NL, nl = lane % NL;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: unit numbers ride in scalar offsets
  uint64_t stamp[7] = {0, 0, 0, 0, 0, 0, 0};
  if constexpr (STAMP) stamp[0] = __builtin_amdgcn_s_memrealtime();
  const uint32_t split = SPLIT ? blockIdx.x % p.splits : 0u;
  const uint32_t gtile = SPLIT ? blockIdx.x / p.splits : xcd_tile(blockIdx.x, p.tiles);  // tile of the whole grid
  uint32_t tile = gtile;  // ... of this workgroup's weight set
  if constexpr (MULTI) {
    const uint32_t j = (gtile >= sets.tile_end[0] ? 1u : 0u) + (gtile >= sets.tile_end[1] ? 1u : 0u) + (gtile >= sets.tile_end[2] ? 1u : 0u);
    tile = gtile - (j ? sets.tile_end[j - 1] : 0u);
    p.n = sets.n[j];
    p.zw = sets.zw[j];
    p.qweight = sets.qweight[j];
    p.scales = sets.scales[j];
    p.qzeros = sets.qzeros[j];
    p.out = sets.out[j];
  }
  // this workgroup's units [ubeg, uend) and the first activation chunk / group that its LDS images start at
  const uint32_t ubeg = SPLIT ? split * p.ups : 0u;
  const uint32_t uend = SPLIT ? (ubeg + p.ups < p.units ? ubeg + p.ups : p.units) : p.units;
  const uint32_t cbase = ubeg * (uint32_t)(LK * UB * (32 / CHK)), gbase = (ubeg * (uint32_t)LK) >> p.gshift;
  const uint32_t row_bytes = p.n * 4u;
  const uint32_t col = tile * p.tcols + (uint32_t)(CPL * nl);
  const bool col_ok = (uint32_t)(CPL * nl) < p.tcols && col < p.n;
  const rsrc_t r_w = make_rsrc(p.qweight, p.kw * row_bytes);
  // Odd k: the dword that holds the last element must be in range (its other half is masked where it is staged).  With several
  // rows of odd length every other row begins on an odd halfword; its dwords are read unaligned (the chip serves them:
  // profiles/r04/unaligned_probe.txt) and the last one of the last row may reach two bytes further - m * k * 2 is then no
  // multiple of 4096 (m <= 32), so those two bytes lie in the tensor's own last page.
  const rsrc_t r_a = make_rsrc(p.a, ((p.m * p.k * 2u + 3u) & ~3u) + (((p.k & 1u) != 0u && p.m > 1u) ? 4u : 0u));
  const rsrc_t r_s = make_rsrc(p.scales, p.groups * p.n * 2u);
  const rsrc_t r_z = make_rsrc(p.qzeros, p.groups * p.zw * 4u);
  const bool stager = wv < STW;  // wave-uniform

  // ---- small operands, loads only (oldest in the queue: they return first) ----
  // (a) activations: chunk c = k CHK*c .. CHK*c + CHK - 1; (b) per group and column: scale and zero-point of staging
  // thread e's (group e / NL, lane slot e % NL).  Rounds past the end fall off the end of their tensors (the descriptors
  // return zeros, no traffic); lane slots past the last column read some other column's values, which are never stored.
  u32x4_t act_raw[WAVEACT ? 1 : SI][WAVEACT ? 1 : MT][CPT];
  uint32_t s_raw[SI][2], z_raw[SI][2];
  u32x4_t g_raw[GIDX ? SI : 1][2];  // act-order: the ids of 8 consecutive rows per round
  const rsrc_t r_g = make_rsrc(GIDX ? static_cast<const void*>(p.g_idx) : static_cast<const void*>(p.a), GIDX ? p.k * 4u : 0u);
  // table entry e = (group gbase + e / NL, lane slot e % NL): its scale word(s) and zero-point word(s)
  auto tab_load = [&](uint32_t e, uint32_t(&sr)[2], uint32_t(&zr)[2]) {
    const uint32_t g = gbase + e / NL, snl = e % NL;
    const uint32_t scol = tile * p.tcols + (uint32_t)CPL * snl;
    const uint32_t h0 = g * p.n + scol;  // first scale (half index)
    if constexpr (CPL == 2) {
      sr[0] = __builtin_amdgcn_raw_buffer_load_b32(r_s, h0 * 2u, 0, 0);  // n and col even
      sr[1] = 0;
    } else {
      const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r_s, (h0 & ~1u) * 2u, 0, 0);
      sr[0] = v.x;
      sr[1] = v.y;
    }
    const uint32_t bit = scol * BITS;
    const uint32_t zoff = (g * p.zw + (bit >> 5)) * 4u;
    if constexpr (CPL * BITS <= 32 && (32 % (CPL * BITS)) == 0) {
      zr[0] = __builtin_amdgcn_raw_buffer_load_b32(r_z, zoff, 0, 0);  // the CPL fields never straddle a word
      zr[1] = 0;
    } else {
      const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r_z, zoff, 0, 0);  // a read past the row end is harmless
      zr[0] = v.x;
      zr[1] = v.y;
    }
  };
  constexpr bool NO_SMALL = (FLAGS & 4) != 0, NO_SUM = (FLAGS & 8) != 0, NO_MATH = (FLAGS & 16) != 0;  // lab ablations (see FLAGS)
  constexpr bool VALU_ACC = (FLAGS & 32) != 0;  // v_dot2 accumulate instead of the MFMA (measurement build)
  static_assert(!VALU_ACC || (MT == 1 && NL == 16 && !GIDX), "VALU accumulate: one row, tiles of 16 lanes");
  if constexpr (NO_SMALL) {
#pragma unroll
    for (int r = 0; r < SI; ++r) {
#pragma unroll
      for (int q = 0; q < CPT; ++q) act_raw[r][0][q] = u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
      s_raw[r][0] = s_raw[r][1] = 0x20002000u;
      z_raw[r][0] = z_raw[r][1] = 0x12345678u + (uint32_t)tid;
    }
  }
  if (stager && !NO_SMALL) {
    if constexpr (GIDX) {
      if (p.g_idx != nullptr) {  // uniform
#pragma unroll
        for (int r = 0; r < SI; ++r) {
          const uint32_t c = (uint32_t)(r * ST + tid);
          g_raw[r][0] = __builtin_amdgcn_raw_buffer_load_b128(r_g, c * 32u, 0, 0);  // rows past k read id 0: their activations are 0
          g_raw[r][1] = __builtin_amdgcn_raw_buffer_load_b128(r_g, c * 32u + 16u, 0, 0);
        }
      } else {  // contiguous groups of ANY size: id = row / group size, one division per 8 rows
#pragma unroll
        for (int r = 0; r < SI; ++r) {
          const uint32_t k0 = (uint32_t)(r * ST + tid) * 8u;
          uint32_t q = k0 / p.gshift, rem = k0 - q * p.gshift;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            g_raw[r][j >> 2][j & 3] = q;
            if (++rem == p.gshift) rem = 0u, ++q;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < (WAVEACT ? 0 : SI); ++r) {
      const uint32_t c = (uint32_t)(r * ST + tid);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int q = 0; q < CPT; ++q) {  // dwords past the end of the tensor read 0; a last dword that holds ONE element is masked when staged
          // rows beyond m fall off the end of the tensor; chunks beyond this range's k must not read on into the next row (MT > 1) or range (split)
          const bool out_of_range = ((SPLIT || MT > 1) && c * (uint32_t)(CHK * 2 * MT) >= p.act_bytes) || (MT > 1 && (cbase + c) * (uint32_t)CHK + 8u * q >= p.k);
          act_raw[r][mi][q] = __builtin_amdgcn_raw_buffer_load_b128(r_a, out_of_range ? kOob : ((uint32_t)mi * p.k + (cbase + c) * (uint32_t)CHK) * 2u + 16u * q, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < SI; ++r) tab_load((uint32_t)(r * ST + tid), s_raw[r], z_raw[r]);
  }

  // ---- the first weight rows of this wave ----
  // lane part of every row address: k-blocks kl * UB .. of the unit (BITS rows each), this lane's columns
  const uint32_t lane_off = col_ok ? (uint32_t)(kl * R) * row_bytes + col * 4u : kOob;
  const uint32_t unit_bytes = (uint32_t)(LK * R) * row_bytes;
  Row<CPL, NT> ra[R], rb[R];
  uint32_t cur = ubeg + (uint32_t)wv, nxt = cur + (uint32_t)NW, nn = cur + (uint32_t)(2 * NW);
  // Request flat rows [from, to) as seen from the unit in X: X = this unit, Y = the next, X again = the one after.
  // A unit past the end of this workgroup's range reads from the descriptor's end on: zeros, no traffic.
  const uint32_t past_end = p.kw * row_bytes;
  auto unit_off = [&](uint32_t u) -> uint32_t { return u < uend ? u * unit_bytes : past_end; };
  auto request = [&](Row<CPL, NT>(&X)[R], Row<CPL, NT>(&Y)[R], auto from_c, auto to_c) {
    constexpr int from = decltype(from_c)::value, to = decltype(to_c)::value;
    static_for<(to > from ? to - from : 0)>([&](auto d) {
      constexpr int f = from + decltype(d)::value;
      if constexpr (f < R) X[f].load(r_w, lane_off, unit_off(cur) + (uint32_t)f * row_bytes);
      else if constexpr (f < 2 * R) Y[f - R].load(r_w, lane_off, unit_off(nxt) + (uint32_t)(f - R) * row_bytes);
      else X[f - 2 * R].load(r_w, lane_off, unit_off(nn) + (uint32_t)(f - 2 * R) * row_bytes);
    });
  };
  // ---- 8 / 16 rows: this wave's activations, one unit at a time (see MT) ----
  constexpr int TPR = LK * UB * (32 / CHK);           // chunks per row and unit
  constexpr int WTASKS = WAVEACT ? MT * TPR : 64;     // chunks per unit
  constexpr int AR = WAVEACT ? WTASKS / 64 : 1;       // rounds: one chunk per lane and round
  static_assert(WTASKS % 64 == 0, "a unit's chunks fill whole rounds");
  u32x4_t wa_raw[AR][CPT];
  // dwords per octet of the per-wave image: MT rows of 16 B plus one empty slot, so that the 16 lanes that store one row's
  // 16 octets (272 B apart, not 256) and the 16 lanes that read one octet's 16 rows both touch 64 different banks
  constexpr int OST = WAVEACT ? (MT + 1) * 4 : MT * 4;
  uint32_t* const my_act = s_act + (uint32_t)wv * (uint32_t)(LK * UB * 4 * OST);
  // chunk t = lane + 64 * round: row t / TPR, chunk t % TPR of the unit - a row's chunks sit in neighbouring lanes (coalesced)
  auto wact_issue = [&](uint32_t u) {
    static_for<AR>([&](auto r_c) {  // (compile-time r: sixteen rounds at 64 rows are not unrolled by a pragma, and wa_raw must stay in registers)
      constexpr int r = decltype(r_c)::value;
      const uint32_t t = (uint32_t)(lane + 64 * r), mi = t / (uint32_t)TPR, cu = t % (uint32_t)TPR;
      const uint32_t k0 = u * (uint32_t)(LK * UB * 32) + cu * (uint32_t)CHK;
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        // rows beyond m fall off the end of the tensor (zeros); k beyond the row would read on into the next row: forced out of
        // range.  A unit beyond the wave's last one reads something real and is never stored.
        const uint32_t kq = k0 + 8u * (uint32_t)q;
        wa_raw[r][q] = __builtin_amdgcn_raw_buffer_load_b128(r_a, (mi * p.k + kq) * 2u | (kq >= p.k ? kOob : 0u), 0, 0);
      }
    });
  };
  auto wact_store = [&](uint32_t u) {
    __builtin_amdgcn_wave_barrier();  // the previous unit's reads of this region come first
    static_for<AR>([&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      const uint32_t t = (uint32_t)(lane + 64 * r), mi = t / (uint32_t)TPR, cu = t % (uint32_t)TPR;
      uint32_t h[CHK / 2];
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        h[4 * q + 0] = wa_raw[r][q].x;
        h[4 * q + 1] = wa_raw[r][q].y;
        h[4 * q + 2] = wa_raw[r][q].z;
        h[4 * q + 3] = wa_raw[r][q].w;
      }
      if (p.k & 7u) {  // wave-uniform.  A row length that is no multiple of 8: the piece that holds element k - 1 read on into the next row
        const int left = (int)p.k - (int)(u * (uint32_t)(LK * UB * 32) + cu * (uint32_t)CHK);  // elements of this chunk inside the row
#pragma unroll
        for (int d = 0; d < CHK / 2; ++d) h[d] = left >= 2 * d + 2 ? h[d] : (left == 2 * d + 1 ? (h[d] & 0xFFFFu) : 0u);
      }
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pr = 4 * q + e;
          const int klo = Unpack<BITS>::pair_k(pr, 0), khi = Unpack<BITS>::pair_k(pr, 1);
          const uint32_t sel = (uint32_t)(2 * (klo & 1)) | ((uint32_t)(2 * (klo & 1) + 1) << 8) | ((uint32_t)(4 + 2 * (khi & 1)) << 16) |
                               ((uint32_t)(5 + 2 * (khi & 1)) << 24);
          o[e] = __builtin_amdgcn_perm(h[khi >> 1], h[klo >> 1], sel);
        }
        *reinterpret_cast<u32x4_t*>(my_act + (cu * (uint32_t)CPT + (uint32_t)q) * (uint32_t)OST + mi * 4u) = o;
      }
    });
    // produced and consumed by the same wave: in-order LDS + a wave-level fence suffice
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  if constexpr (WAVEACT) wact_issue(cur);  // ahead of the first rows: the oldest loads of this wave
  request(ra, rb, std::integral_constant<int, 0>{}, std::integral_constant<int, P::open>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (STAMP) stamp[6] = __builtin_amdgcn_s_memrealtime();  // first rows requested

  // ---- stage the small operands (waits for them only: the weight rows stay in flight) ----
  bool bad = false;
  if (stager) {
#pragma unroll
    for (int r = 0; r < (WAVEACT ? 0 : SI); ++r) {
      const uint32_t c = (uint32_t)(r * ST + tid);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        uint32_t h[CHK / 2];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          h[4 * q + 0] = act_raw[r][mi][q].x;
          h[4 * q + 1] = act_raw[r][mi][q].y;
          h[4 * q + 2] = act_raw[r][mi][q].z;
          h[4 * q + 3] = act_raw[r][mi][q].w;
        }
        {
          // k is any length (one row): elements at k and beyond count as 0 (the packed rows hold padding fields there).
          // Only an odd k leaves something to clear - the upper half of the dword that holds element k - 1.
          const int left = (int)p.k - (int)((cbase + c) * (uint32_t)CHK);  // elements of this chunk inside the row
#pragma unroll
          for (int d = 0; d < CHK / 2; ++d) h[d] = left >= 2 * d + 2 ? h[d] : (left == 2 * d + 1 ? (h[d] & 0xFFFFu) : 0u);
        }
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          u32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // pair (4q + e) of this chunk in the unpacker's order; pair_k is periodic in the chunk
            const int pr = 4 * q + e;
            const int klo = Unpack<BITS>::pair_k(pr, 0), khi = Unpack<BITS>::pair_k(pr, 1);
            static_assert(Unpack<BITS>::pair_k(CHK / 2 - 1, 1) < CHK, "pairs stay inside a chunk");
            const uint32_t sel = (uint32_t)(2 * (klo & 1)) | ((uint32_t)(2 * (klo & 1) + 1) << 8) | ((uint32_t)(4 + 2 * (khi & 1)) << 16) |
                                 ((uint32_t)(5 + 2 * (khi & 1)) << 24);
            o[e] = __builtin_amdgcn_perm(h[khi >> 1], h[klo >> 1], sel);
          }
          // octet (c * CPT + q) of the range, row mi: 16 bytes at ((octet * MT) + mi) * 16
          if (c * (uint32_t)(CHK * 2 * MT) < p.act_bytes) *reinterpret_cast<u32x4_t*>(s_act + ((c * (uint32_t)CPT + (uint32_t)q) * (uint32_t)MT + (uint32_t)mi) * 4u) = o;
        }
      }
    }
    auto tab_store = [&](uint32_t e, const uint32_t(&sr)[2], const uint32_t(&zr)[2]) {
      const uint32_t g = gbase + e / NL, snl = e % NL;
      const uint32_t scol = tile * p.tcols + (uint32_t)CPL * snl;
      uint32_t sb[CPL];
      if constexpr (CPL == 2) {
        sb[0] = sr[0] & 0xFFFFu;
        sb[1] = sr[0] >> 16;
      } else {
        const uint32_t sh = ((g * p.n + scol) & 1u) * 16u;
        const uint32_t lo = __builtin_amdgcn_alignbit(sr[1], sr[0], sh), hi = sr[1] >> sh;
        sb[0] = lo & 0xFFFFu;
        sb[1] = lo >> 16;
        sb[2] = hi & 0xFFFFu;
        if constexpr (CPL == 4) sb[3] = hi >> 16;
      }
      const uint32_t zsh = (scol * BITS) & 31u;
      uint32_t zwin;
      if constexpr (CPL * BITS <= 32 && (32 % (CPL * BITS)) == 0) zwin = zr[0] >> zsh;
      else zwin = __builtin_amdgcn_alignbit(zr[1], zr[0], zsh);
      u32x4_t ent = {0u, 0u, 0u, 0u};  // (scale | -(zero + bias) * scale << 16) per column
      u32x4_t ent_sub = {0u, 0u, 0u, 0u};  // act-order: the scale pre-multiplied for the subnormal form (a pair mixes two groups)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const _Float16 s = as_h((uint16_t)sb[c]);
        const uint32_t z = ((zwin >> (c * BITS)) & Unpack<BITS>::kMask) + p.bias;
        bool ok;
        const _Float16 ssub = sub_scale<BITS>(s, ok);
        bad = bad || !ok;
        const uint32_t nz = (uint32_t)as_u16(neg_zero_times_scale(z, s)) << 16;
        ent[c] = (uint32_t)as_u16(s) | nz;
        ent_sub[c] = (uint32_t)as_u16(ssub) | nz;
      }
      if (e / NL < p.tab_groups) {
        if constexpr (GIDX) {
          *reinterpret_cast<u32x4_t*>(s_tab + e * 4u) = ent_sub;
          *reinterpret_cast<u32x4_t*>(s_tab_plain + e * 4u) = ent;
        } else {
          *reinterpret_cast<u32x4_t*>(s_tab + e * 4u) = ent;
        }
      }
    };
#pragma unroll
    for (int r = 0; r < SI; ++r) tab_store((uint32_t)(r * ST + tid), s_raw[r], z_raw[r]);
    if constexpr (GIDX || MT >= 8) {
      // more groups than SI rounds of table entries (small or odd group sizes: K = 4096 at group 16 has 256): the rest one
      // round at a time, load -> store.  Only the by-id build (its table holds every group of the layer) and the 8- / 16-row
      // builds (one build for every K: 2 static rounds cover K = 4096 at group 128) pay for the loop.
      for (uint32_t e = (uint32_t)(SI * ST + tid); e < p.tab_groups * (uint32_t)NL; e += (uint32_t)ST) {
        uint32_t sr[2], zr[2];
        tab_load(e, sr, zr);
        tab_store(e, sr, zr);
      }
    }
    if constexpr (GIDX) {
#pragma unroll
      for (int r = 0; r < SI; ++r) {
        const uint32_t c = (uint32_t)(r * ST + tid);
        const uint32_t top = p.groups - 1u;
        u32x4_t o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t lo = d < 2 ? g_raw[r][0][2 * d] : g_raw[r][1][2 * d - 4], hi = d < 2 ? g_raw[r][0][2 * d + 1] : g_raw[r][1][2 * d - 3];
          o[d] = (lo < top ? lo : top) | ((hi < top ? hi : top) << 16);  // out-of-range (or negative) ids are clamped, never followed
        }
        if (c * 4u < id_dwords) *reinterpret_cast<u32x4_t*>(s_gid + c * 4u) = o;
      }
      // a K beyond the static rounds (8 / 16-row builds: one build for every K): the rest one round at a time
      for (uint32_t c = (uint32_t)(SI * ST + tid); c * 4u < id_dwords; c += (uint32_t)ST) {
        const uint32_t top = p.groups - 1u;
        uint32_t id[8];
        if (p.g_idx != nullptr) {
          const u32x4_t a0 = __builtin_amdgcn_raw_buffer_load_b128(r_g, c * 32u, 0, 0), a1 = __builtin_amdgcn_raw_buffer_load_b128(r_g, c * 32u + 16u, 0, 0);
          id[0] = a0.x; id[1] = a0.y; id[2] = a0.z; id[3] = a0.w; id[4] = a1.x; id[5] = a1.y; id[6] = a1.z; id[7] = a1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) id[j] = (c * 8u + (uint32_t)j) / p.gshift;
        }
        u32x4_t o;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = (id[2 * d] < top ? id[2 * d] : top) | ((id[2 * d + 1] < top ? id[2 * d + 1] : top) << 16);
        *reinterpret_cast<u32x4_t*>(s_gid + c * 4u) = o;
      }
    }
  }
  if constexpr (WIDE) {
    for (int i = tid; i < UB * 16 * MT; i += 64 * NW) s_zero[i] = 0u;
  }
  {
    const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (lane < 8 && (lane == wv || (wv == 0 && lane >= NW))) s_bad[lane] = (lane == wv && wave_bad) ? 1u : 0u;  // wave 0 also clears the unused slots
  }
  if constexpr (STAMP) stamp[1] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if constexpr (STAMP) stamp[2] = __builtin_amdgcn_s_memrealtime();
  const u32x4_t b0 = *reinterpret_cast<const u32x4_t*>(s_bad), b1 = *reinterpret_cast<const u32x4_t*>(s_bad + 4);
  const bool fast = __builtin_amdgcn_readfirstlane((int)(b0.x | b0.y | b0.z | b0.w | b1.x | b1.y | b1.z | b1.w)) == 0;

  f4_t acc[CPL], acc_hi[RG > 1 ? RG - 1 : 1][CPL];  // acc_hi[g - 1]: rows 16 g .. 16 g + 15 of the 32- and 64-row builds
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < (RG > 1 ? RG - 1 : 1); ++g)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc_hi[g][c] = f4_t{0.f, 0.f, 0.f, 0.f};
  const uint32_t sub_mask = opaque_sgpr(Unpack<BITS>::kSubMask2);
  const typename Unpack<BITS>::Consts uk = Unpack<BITS>::consts();
  // dwords: this lane's k-blocks of the unit (4 octets each, MT rows of 4 dwords per octet), its row of the A operand
  constexpr int OSTEP = MT >= 8 ? (MT + 1) * 4 : MT * 4;  // dwords from one octet to the next (8 / 16 rows: padded, see OST)
  const uint32_t a_lane = (uint32_t)(kl * UB) * (uint32_t)(4 * OSTEP) + (uint32_t)((WIDE ? (nl & 7) : nl) % (MT > 16 ? 16 : MT)) * 4u;
  // 128-column tiles: a lane whose row block belongs to the OTHER column set feeds zeros (a unit's worth of them sits behind the reduction scratch)
  const bool zero_lane = WIDE && (nl >> 4) != ((nl >> 3) & 1);

  // per-unit dequant constants of this lane's columns
  auto unit_consts = [&](uint32_t u, half2_t(&s2)[CPL], half2_t(&nzs2)[CPL], half2_t(&ss2)[CPL]) {
    const uint32_t gi = ((u * (uint32_t)LK + (uint32_t)kl) >> p.gshift) - gbase;  // < tab_groups
    const u32x4_t ent = *reinterpret_cast<const u32x4_t*>(s_tab + (gi * (uint32_t)NL + (uint32_t)nl) * 4u);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      s2[c] = as_h2(__builtin_amdgcn_perm(ent[c], ent[c], 0x01000100u));
      nzs2[c] = as_h2(__builtin_amdgcn_perm(ent[c], ent[c], 0x03020302u));
      // scale * 2^(14 + BITS) in two exact steps (the factor itself is not an fp16); finite for every scale of a tile that takes the subnormal form
      constexpr int L = Unpack<BITS>::kSubScaleLog2;
      ss2[c] = (s2[c] * splat((_Float16)(float)(1 << 8))) * splat((_Float16)(float)(1 << (L - 8)));
    }
  };
  // one MFMA step: octet `oct` of k-block `j` of the unit whose rows are in X.  `gq`: act-order only - the 32 group ids of
  // each of the lane's k-blocks, two per dword.
  auto step = [&](const Row<CPL, NT>(&X)[R], const uint32_t* a_unit, auto s_c, auto fast_c, const half2_t(&s2)[CPL], const half2_t(&nzs2)[CPL],
                  const half2_t(&ss2)[CPL], const uint32_t(&gq)[GIDX ? UB : 1][16]) {
    constexpr int s = decltype(s_c)::value, j = s / 4, oct = s % 4;
    constexpr bool FAST = decltype(fast_c)::value;
    if constexpr (NO_MATH) {  // lab ablation: wait for the rows this step reads, compute nothing
      static_for<(P::need(s) - (s == 0 ? 0 : P::need(s - 1)))>([&](auto d) {
        constexpr int row = (s == 0 ? 0 : P::need(s - 1)) + decltype(d)::value;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const uint32_t t = X[row].v[c];
          asm volatile("" ::"v"(t));
        }
      });
      return;
    }
    const u32x4_t araw = *reinterpret_cast<const u32x4_t*>(a_unit + (j * 4 + oct) * OSTEP);
    const h8_t a_frag = __builtin_bit_cast(h8_t, araw);
    uint32_t wc[CPL][BITS];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
      for (int i = 0; i < BITS; ++i) wc[c][i] = X[j * BITS + i].v[c];
    u32x4_t braw[CPL];
    static_for<4>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value, pr = 4 * oct + e;
      half2_t ps[CPL], pnz[CPL];  // this pair's scale (pre-multiplied on the fast path) and -zero*scale, low half = lower k
      if constexpr (GIDX) {
        constexpr int klo = Unpack<BITS>::pair_k(pr, 0), khi = Unpack<BITS>::pair_k(pr, 1);
        const uint32_t id_lo = (gq[j][klo >> 1] >> (16 * (klo & 1))) & 0xFFFFu, id_hi = (gq[j][khi >> 1] >> (16 * (khi & 1))) & 0xFFFFu;
        const uint32_t* const tab = FAST ? s_tab : s_tab_plain;
        const u32x4_t elo = *reinterpret_cast<const u32x4_t*>(tab + (id_lo * (uint32_t)NL + (uint32_t)nl) * 4u);
        const u32x4_t ehi = *reinterpret_cast<const u32x4_t*>(tab + (id_hi * (uint32_t)NL + (uint32_t)nl) * 4u);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          ps[c] = as_h2(__builtin_amdgcn_perm(ehi[c], elo[c], 0x05040100u));
          pnz[c] = as_h2(__builtin_amdgcn_perm(ehi[c], elo[c], 0x07060302u));
        }
      } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          ps[c] = FAST ? ss2[c] : s2[c];
          pnz[c] = nzs2[c];
        }
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if constexpr (FAST) {
          const uint32_t q = Unpack<BITS>::template sub_pair<pr>(wc[c], sub_mask);
          braw[c][e] = as_u32(__builtin_elementwise_fma(as_h2(q), ps[c], pnz[c]));
        } else {
          const uint32_t q = Unpack<BITS>::template pair<pr>(wc[c], uk);
          braw[c][e] = as_u32(dequant_pair<Unpack<BITS>::pair_exp(pr)>(q, ps[c], pnz[c]));
        }
      }
    });
    if constexpr (VALU_ACC) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c][0] = __builtin_amdgcn_fdot2(as_h2(braw[c][e]), as_h2(araw[e]), acc[c][0], false);
    } else {
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_frag, __builtin_bit_cast(h8_t, braw[c]), acc[c], 0, 0, 0);
    }
    if constexpr (RG > 1) {  // rows 16 and up: the same B fragments once more per group of 16 rows
#pragma unroll
      for (int g = 1; g < RG; ++g) {
        const u32x4_t araw_hi = *reinterpret_cast<const u32x4_t*>(a_unit + (j * 4 + oct) * OSTEP + 64 * g);
        const h8_t a_hi = __builtin_bit_cast(h8_t, araw_hi);
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          acc_hi[g - 1][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, __builtin_bit_cast(h8_t, braw[c]), acc_hi[g - 1][c], 0, 0, 0);
      }
    }
  };
  // act-order: the ids of the lane's k-blocks of unit u (64 bytes per k-block)
  auto unit_ids = [&](uint32_t u, uint32_t(&gq)[GIDX ? UB : 1][16]) {
    if constexpr (GIDX) {
      const uint32_t* const g_unit = s_gid + ((u - ubeg) * (uint32_t)LK + (uint32_t)kl) * (uint32_t)(UB * 16);
#pragma unroll
      for (int j = 0; j < UB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4_t v = *reinterpret_cast<const u32x4_t*>(g_unit + j * 16 + 4 * q);
          gq[j][4 * q + 0] = v.x; gq[j][4 * q + 1] = v.y; gq[j][4 * q + 2] = v.z; gq[j][4 * q + 3] = v.w;
        }
    }
  };

  // one unit: its rows are in X (in flight or landed); behind each step the rows Plan schedules go out, into Y (next
  // unit) and, once dead, back into X (the unit after that).  A unit past the end computes nothing and loads nothing real.
  auto process = [&](Row<CPL, NT>(&X)[R], Row<CPL, NT>(&Y)[R]) {
    const bool live = cur < uend;  // wave-uniform
    half2_t s2[CPL], nzs2[CPL], ss2[CPL];
    const uint32_t* const a_unit = WAVEACT ? my_act + a_lane : (zero_lane ? s_zero : s_act + (cur - ubeg) * (uint32_t)(LK * UB * 16 * MT) + a_lane);
    uint32_t gq[GIDX ? UB : 1][16];
    if constexpr (WAVEACT) {
      if (live) wact_store(cur);  // this unit's activations: requested one unit ago, the oldest loads in flight
      wact_issue(nxt);         // the next unit's: in front of every row this unit will request for it
      __builtin_amdgcn_sched_barrier(0);
    }
    if (live) {
      if constexpr (GIDX) unit_ids(cur, gq);
      else unit_consts(cur, s2, nzs2, ss2);
    }
    static_for<S>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if (live) step(X, a_unit, s_c, std::true_type{}, s2, nzs2, ss2, gq);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int from = P::need(s) + LA, to = P::need(s + 1) + LA;
      // a row of the unit after next reuses a register of this unit: it must be dead by now
      static_assert(to <= 2 * R || to - 1 - 2 * R < P::first_row(s + 1), "register of a live row reused");
      request(X, Y, std::integral_constant<int, from>{}, std::integral_constant<int, to>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    cur = nxt;
    nxt = nn;
    nn += (uint32_t)NW;
  };

  if (fast) {
    // units are taken two at a time so that the register sets keep their roles across the back edge
    do {
      process(ra, rb);
      process(rb, ra);
    } while (cur < uend);
  } else {
    // some scale of the tile is too large for the subnormal form (or not finite): the spliced form, same W bit for bit.
    // Not a path real checkpoints take: plain loop, own loads, units w, w + NW, ...
    for (uint32_t u = ubeg + (uint32_t)wv; u < uend; u += (uint32_t)NW) {
      half2_t s2[CPL], nzs2[CPL], ss2[CPL];
      uint32_t gq[GIDX ? UB : 1][16];
      if constexpr (GIDX) unit_ids(u, gq);
      else unit_consts(u, s2, nzs2, ss2);
      Row<CPL, NT> rs[R];
#pragma unroll
      for (int i = 0; i < R; ++i) rs[i].load(r_w, lane_off, u * unit_bytes + (uint32_t)i * row_bytes);
      if constexpr (WAVEACT) {
        wact_issue(u);
        wact_store(u);
      }
      const uint32_t* const a_unit = WAVEACT ? my_act + a_lane : (zero_lane ? s_zero : s_act + (u - ubeg) * (uint32_t)(LK * UB * 16 * MT) + a_lane);
      static_for<S>([&](auto s_c) { step(rs, a_unit, s_c, std::false_type{}, s2, nzs2, ss2, gq); });
    }
  }
  if constexpr (STAMP) {
    asm volatile("" : "+v"(acc[0]));
    stamp[3] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- cross-wave sum.  D[row][slot]: row 4 * (lane >> 4) + r lives in register r of lane (lane >> 4, slot = lane & 15).
  // NL = 16: rows 0..MT-1 (registers 0..MT-1 of lanes 0-15) are the sums over the wave's four k-slots for column slot = nl.
  // NL = 8: even k-slots sum into D[m][0..7] (lanes 0-7), odd ones into D[8 + m][8..15] (lanes 40-47). ----
  if constexpr (VALU_ACC) {  // the four k-slots of a column, then where the MFMA build has the sum: register 0 of lanes 0-15
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      float v = acc[c][0];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[c][0] = v;
    }
  }
  if constexpr (NO_SUM) {  // lab ablation: every wave stores its own partial sums, no LDS, no barrier
    if (lane < 16 && col_ok) {
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        if (col + (uint32_t)c < p.n) p.out[col + (uint32_t)c] = as_u16((_Float16)acc[c][0]);
    }
    return;
  }
  {
    if constexpr (ALIAS) __syncthreads();  // every wave is done with its activations
    // NL = 32: column set 0 in rows 0.. (lanes 0-15), column set 1 in rows 8.. (lanes 32-47)
    const int h = NL >= 16 ? 0 : (lane >= 32 ? 1 : 0);
    const bool holder = NL == 16 ? lane < 16 : (NL == 32 ? (lane < 16 || (lane >= 32 && lane < 48)) : (lane < 8 || (lane >= 40 && lane < 48)));
    const int hcol = NL == 32 ? (lane >= 32 ? 16 : 0) + (lane & 15) : lane % NL;  // lane slot of the holder's columns
    if constexpr (WAVEACT) {  // rows 4 * (lane >> 4) + r, all four lane groups up to MT
      if (4 * (lane >> 4) < MT) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) s_red[(wv * MT + 4 * (lane >> 4) + r) * CN + nl * CPL + c] = acc[c][r];
        if constexpr (RG > 1) {
#pragma unroll
          for (int g = 1; g < RG; ++g)
#pragma unroll
            for (int c = 0; c < CPL; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) s_red[(wv * MT + 16 * g + 4 * (lane >> 4) + r) * CN + nl * CPL + c] = acc_hi[g - 1][c][r];
        }
      }
    } else {
      if (holder) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) s_red[((wv * HS + h) * MT + mi) * CN + hcol * CPL + c] = acc[c][mi];
      }
    }
  }
  __syncthreads();
  if constexpr (STAMP) stamp[4] = __builtin_amdgcn_s_memrealtime();
  for (int e0 = tid; e0 < MT * CN; e0 += 64 * NW) {
    const int mi = e0 / CN, cc = e0 % CN;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW * HS; ++w) sum += s_red[(w * MT + mi) * CN + cc];
    const uint32_t n_out = tile * p.tcols + (uint32_t)cc;
    const bool valid = (uint32_t)cc < p.tcols && n_out < p.n && (uint32_t)mi < p.m;
    if constexpr (SPLIT) {
      // One trip to the device coherence point, as in xbit_gemv_mfma.h: the partial sum goes into the element's 64-bit
      // accumulator as a fixed-point integer (2^-28 units, biased positive) by ONE returning atomic whose upper bits count
      // arrivals (bits 0-51 sum of (v + 2^47), 52-55 arrivals, 56-59 / 60-63 arrivals with a partial beyond +-2^19, Inf
      // or NaN); integer addition is associative, so the result does not depend on the arrival order.  The thread that
      // completes the count converts, stores and clears the slot.  Contract: include/xbitops_hip.h (split_k).
      if (valid) {
        unsigned long long* slot = p.acc + (size_t)gtile * (size_t)(MT * CN) + (size_t)e0;
        const bool fin = __builtin_fabsf(sum) < 524288.0f;
        const long long v = fin ? (long long)__builtin_rintf(sum * 268435456.0f) : 0ll;
        const unsigned long long pos = (!fin && !(sum < 0.f)) ? (1ull << 56) : 0ull;
        const unsigned long long neg = (!fin && !(sum > 0.f)) ? (1ull << 60) : 0ull;
        const unsigned long long add = (unsigned long long)(v + (1ll << 47)) + (1ull << 52) + pos + neg;
        const unsigned long long old = __hip_atomic_fetch_add(slot, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((old >> 52) & 15ull) + 1ull == p.splits) {
          const unsigned long long tot = old + add;
          const long long fixed = (long long)(tot & ((1ull << 52) - 1ull)) - (long long)p.splits * (1ll << 47);
          float y = (float)fixed * (1.0f / 268435456.0f);
          const bool any_pos = ((tot >> 56) & 15ull) != 0, any_neg = (tot >> 60) != 0;
          if (any_pos || any_neg) y = (any_pos && any_neg) ? __builtin_nanf("") : (any_pos ? __builtin_inff() : -__builtin_inff());
          p.out[(size_t)mi * p.n + n_out] = as_u16((_Float16)y);
          __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else {
      if (valid) p.out[(size_t)mi * p.n + n_out] = as_u16((_Float16)sum);
    }
  }
  if constexpr (STAMP) {
    stamp[5] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && p.trace != nullptr) {
      uint64_t* o = p.trace + ((size_t)blockIdx.x * NW + wv) * 8;
#pragma unroll
      for (int i = 0; i < 7; ++i) o[i] = stamp[i];
      // where the wave ran: XCC_ID (hwreg 20) above HW_ID (hwreg 4: wave, simd, cu, sh, se ids)
      o[7] = ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------

// LDS bytes of one workgroup (dynamic)
inline size_t dyn_lds_bytes(const DynParams& p, int nl, int nw, int cpl, int mt) {
  const int hs = nl >= 16 ? 1 : 16 / nl, cn = nl * cpl;
  // act-order: group ids (2 B per k) beside the activations and a second table for the spliced form
  const size_t ids = p.by_id ? (size_t)p.id_bytes : 0u, tables = p.by_id ? 2u : 1u;
  const size_t red = (size_t)((nw * hs * mt * cn + 3) & ~3) * 4u;
  const size_t act_and_red = mt >= 32 ? ((size_t)p.act_bytes > red ? (size_t)p.act_bytes : red) : (size_t)p.act_bytes + red;  // 32 / 64 rows: the scratch reuses the activation region
  return act_and_red + ids + tables * (size_t)p.tab_groups * (size_t)(nl * 16) + 32u + (nl == 32 ? (size_t)(2 * 16 * mt * 4) : 0u);
}

constexpr int kDynNotEligible = kGemvDynNotEligible;  // not a status: "this shape is not for this kernel", the caller launches another
constexpr int kDynWaves = 8, kDynStagers = 4, kDynLookahead = 4;
constexpr size_t kDynMaxLds = 96u << 10;  // larger staged operands (K beyond ~24 K at g128) stay with xbit_gemv_mfma.h
constexpr size_t kDynMaxLdsById = 156u << 10;  // act-order / odd group sizes: two table rows of 256 B per group (K = 4096 at group 16: 128 KB of them)

template <int BITS>
constexpr int dyn_ub() {
  return BITS == 2 ? 2 : 1;  // >= 3 row loads per unit
}

// NW waves per workgroup, the older half of them staging (8 / 4 everywhere but the 128-column tiles, which run 16 / 8: one workgroup per CU there)
template <int BITS, int CPL, int NL, bool NT, int SI, int FLAGS = 0, bool SPLIT = false, bool GIDX = false, int MT = 1, int NW = kDynWaves, bool MULTI = false, int LA = kDynLookahead>
int launch_dyn_one(const DynParams& p, hipStream_t stream) {
  constexpr int UB = dyn_ub<BITS>();
  auto kern = gemv_dyn_kernel<BITS, UB, CPL, NL, NW, NT, SI, NW / 2, LA, FLAGS, SPLIT, GIDX, MT, MULTI>;
  static_assert(kDynStagers == kDynWaves / 2, "the older half stages");
  DynParams q = p;
  GemvSets sets = {};
  if (MULTI && p.sets != nullptr) sets = *p.sets;
  void* args[] = {&q.qweight, &q.n, &q.tiles, &q.kw, &q.units, &q.a, &q.scales, &q.qzeros, &q.out, &q.k, &q.groups, &q.zw, &q.bias, &q.m,
                  &q.tab_groups, &q.gshift, &q.act_bytes, &q.splits, &q.ups, &q.acc, &q.g_idx, &q.trace, &q.tcols, &sets};
  return launch_raw_dynamic(reinterpret_cast<const void*>(kern), dim3(p.tiles * (SPLIT ? p.splits : 1u), (MT == 32 && !SPLIT && p.row_blocks > 1u) ? p.row_blocks : 1u), dim3(64 * NW),
                            dyn_lds_bytes(p, NL, NW, CPL, MT), stream, args);
}

// 128-column tiles (32 lanes x 16 B): 4- and 8-bit, one or up to four rows, whole K, sixteen waves
constexpr int kDynWideWaves = 16;
template <int BITS, bool NT>
int launch_dyn_wide(const DynParams& p, hipStream_t stream, uint32_t rounds) {
  if constexpr (BITS == 4 || BITS == 8) {
#if XBIT_TUNING
    if (p.wide_probe != 0u && p.mt == 1u && !p.by_id && rounds <= 4u) {  // measurement builds (variants 344-347): K split over workgroups, eight waves, deeper lookahead
      if (p.splits > 1u) return p.wide_probe == 2u ? launch_dyn_one<BITS, 4, 32, NT, 4, 0, true, false, 1, 8>(p, stream) : launch_dyn_one<BITS, 4, 32, NT, 4, 0, true, false, 1, kDynWideWaves>(p, stream);
      if (p.wide_probe == 2u) return launch_dyn_one<BITS, 4, 32, NT, 4, 0, false, false, 1, 8>(p, stream);
      if (p.wide_probe == 3u) return launch_dyn_one<BITS, 4, 32, NT, 4, 0, false, false, 1, kDynWideWaves, false, 6>(p, stream);
    }
#endif
    if (p.by_id || p.splits > 1u || p.mt > 4u) return kDynNotEligible;
    if (p.mt > 1u) {
      if (rounds <= 2u) return launch_dyn_one<BITS, 4, 32, NT, 2, 0, false, false, 4, kDynWideWaves>(p, stream);
      if (rounds <= 4u) return launch_dyn_one<BITS, 4, 32, NT, 4, 0, false, false, 4, kDynWideWaves>(p, stream);
      return kDynNotEligible;
    }
    if (rounds <= 2u) return launch_dyn_one<BITS, 4, 32, NT, 2, 0, false, false, 1, kDynWideWaves>(p, stream);
    if (rounds <= 4u) return launch_dyn_one<BITS, 4, 32, NT, 4, 0, false, false, 1, kDynWideWaves>(p, stream);
    if (rounds <= 8u) return launch_dyn_one<BITS, 4, 32, NT, 8, 0, false, false, 1, kDynWideWaves>(p, stream);
  }
  return kDynNotEligible;
}

// fused launches (p.sets): tiles of 16 lanes, 1..16 rows, whole K
template <int BITS, int CPL>
int launch_dyn_multi_si(const DynParams& p, hipStream_t stream, uint32_t rounds) {
  constexpr int W = kDynWaves;
  if (p.mt > 8u) return launch_dyn_one<BITS, CPL, 16, true, 2, 0, false, false, 16, W, true>(p, stream);  // table rounds beyond 2: staged in a loop
  if (p.mt > 4u) return launch_dyn_one<BITS, CPL, 16, true, 2, 0, false, false, 8, W, true>(p, stream);
  if (p.mt > 1u) {
    if (rounds <= 2u) return launch_dyn_one<BITS, CPL, 16, true, 2, 0, false, false, 4, W, true>(p, stream);
    if (rounds <= 4u) return launch_dyn_one<BITS, CPL, 16, true, 4, 0, false, false, 4, W, true>(p, stream);
    return kDynNotEligible;
  }
  if (rounds <= 2u) return launch_dyn_one<BITS, CPL, 16, true, 2, 0, false, false, 1, W, true>(p, stream);
  if (rounds <= 3u) return launch_dyn_one<BITS, CPL, 16, true, 3, 0, false, false, 1, W, true>(p, stream);
  if (rounds <= 4u) return launch_dyn_one<BITS, CPL, 16, true, 4, 0, false, false, 1, W, true>(p, stream);
  if (rounds <= 8u) return launch_dyn_one<BITS, CPL, 16, true, 8, 0, false, false, 1, W, true>(p, stream);
  return kDynNotEligible;
}

template <int BITS, int CPL, int NL, bool NT>
int launch_dyn_si(const DynParams& p, hipStream_t stream, uint32_t rounds, bool stamped) {
#if XBIT_TUNING
  if (stamped) {
    if (rounds <= 2u && p.splits <= 1u && p.mt == 1u && !p.by_id) {
      if (p.trace != nullptr) return launch_dyn_one<BITS, CPL, NL, NT, 2, 2>(p, stream);
      if constexpr (NL == 16) return launch_dyn_one<BITS, CPL, NL, NT, 2, 32>(p, stream);  // "stamped" without a trace buffer: the VALU-accumulate build (variant 321)
    }
    return kDynNotEligible;
  }
#else
  (void)stamped;
#endif
  if (p.by_id) {  // act-order: 48-column tiles; 1 or up to 4 rows: K up to 4096 (2 rounds) or 8192 (4 rounds); 5..16 rows: any K
    if constexpr (NL == 16 && CPL == 3) {
      if (p.mt > 16u) {  // 2-bit (2 k-blocks per unit) does not fit the 32-row builds into 256 registers: not compiled, 16 rows per launch there
        if constexpr (dyn_ub<BITS>() == 1) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, true, 32>(p, stream);
        return kDynNotEligible;
      }
      if (p.mt > 8u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, true, 16>(p, stream);
      if (p.mt > 4u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, true, 8>(p, stream);
      if (p.mt > 1u) {
        if (rounds <= 2u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, true, 4>(p, stream);
        if (rounds <= 4u) return launch_dyn_one<BITS, CPL, NL, NT, 4, 0, false, true, 4>(p, stream);
        return kDynNotEligible;
      }
      if (rounds <= 2u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, true>(p, stream);
      if (rounds <= 4u) return launch_dyn_one<BITS, CPL, NL, NT, 4, 0, false, true>(p, stream);
    }
    return kDynNotEligible;
  }
  if (p.mt > 4u) {  // 5..16 rows (or fewer rows of a K beyond the 1- / 4-row builds' staging rounds): the per-wave activation pipeline (8- and 16-row builds); the rounds stage the table only
    if constexpr (NL == 16) {
      if (p.splits > 1u) {  // split-K (few-column long-K layers: down-projections at small batch), 64-column tiles
        if constexpr (CPL == 4) {
          if (p.mt <= 8u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, true, false, 8>(p, stream);
          if (p.mt <= 16u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, true, false, 16>(p, stream);
          if constexpr (dyn_ub<BITS>() == 1) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, true, false, 32>(p, stream);  // 17..32 rows (round 4)
        }
        return kDynNotEligible;
      }
      if (p.mt <= 8u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, false, 8>(p, stream);  // table rounds beyond 2: staged in a loop
      if (p.mt <= 16u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, false, 16>(p, stream);
      if constexpr (dyn_ub<BITS>() == 1) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, false, 32>(p, stream);  // (p.row_blocks of them in grid.y)
      return kDynNotEligible;  // 2-bit: no 32-row build (it would spill)
    }
    return kDynNotEligible;
  }
  if (p.mt > 1u) {  // 2..4 rows: whole-K tiles of 16 lanes, K up to 4096 (2 rounds) or 8192 (4 rounds)
    if constexpr (NL == 16) {
      if (p.splits > 1u) {
        if constexpr (CPL == 4) {
          if (rounds <= 2u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, true, false, 4>(p, stream);
          if (rounds <= 4u) return launch_dyn_one<BITS, CPL, NL, NT, 4, 0, true, false, 4>(p, stream);
        }
        return kDynNotEligible;
      }
      if (rounds <= 2u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, false, false, 4>(p, stream);
      if (rounds <= 4u) return launch_dyn_one<BITS, CPL, NL, NT, 4, 0, false, false, 4>(p, stream);
    }
    return kDynNotEligible;
  }
  if (p.splits > 1u) {  // split-K: 32- and 64-column tiles, K ranges of up to 4096 (2 rounds) or 8192 (4 rounds)
    if constexpr (NL == 16 && CPL != 3) {
      if (rounds <= 2u) return launch_dyn_one<BITS, CPL, NL, NT, 2, 0, true>(p, stream);
      if (rounds <= 4u) return launch_dyn_one<BITS, CPL, NL, NT, 4, 0, true>(p, stream);
    }
    return kDynNotEligible;
  }
  // one build per count of staging rounds (2: K <= 4096 at group 128; 3: K = 5120; 4: K <= 8192; 8: up to 16384).  A build
  // that skips unneeded rounds at run time instead took 0.9 us longer per launch (profiles/r03/lab_9.txt, lab_10.txt).
  if (rounds <= 2u) return launch_dyn_one<BITS, CPL, NL, NT, 2>(p, stream);
  if (rounds <= 3u) return launch_dyn_one<BITS, CPL, NL, NT, 3>(p, stream);
  if (rounds <= 4u) return launch_dyn_one<BITS, CPL, NL, NT, 4>(p, stream);
  if (rounds <= 8u) return launch_dyn_one<BITS, CPL, NL, NT, 8>(p, stream);
  return kDynNotEligible;
}

// GEMV of one to four rows through the staged-operand kernel, or kDynNotEligible.  `cpl` / `nl`: tile shape (16 * cpl
// columns, or nl = 8: 8 * cpl); the caller picks them.  Eligible: one activation row of any length k, or up to four with
// k % 8 == 0; any group size (power-of-two multiples of 32 * UB index the table by k-slot, the rest by computed id);
// 4-byte aligned tensors, staged operands within kDynMaxLds and 8 staging rounds, every byte offset below 2^31.
// `splits` >= 2 with `acc` (zeroed accumulators, one per output element of whole tiles): K is cut into that many ranges.
template <int BITS>
int launch_dyn_bits(const GemvParams& g, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits = 1,
                    unsigned long long* acc = nullptr, const int32_t* g_idx = nullptr, uint32_t tcols = 0, const GemvSets* sets = nullptr) {
  constexpr int UB = dyn_ub<BITS>();
  if (g.m < 1u || g.m > 256u || (g.m > 1u && stamped)) return kDynNotEligible;  // (a row length that is no multiple of 8 is masked where it is staged)
  if (g.m > 4u && nl != 16) return kDynNotEligible;                              // 5..32 rows: tiles of 16 lanes
  if (sets != nullptr && (nl != 16 || (cpl != 3 && cpl != 4) || g.m > 16u || g_idx != nullptr || g.groupsize % 32u != 0u || splits > 1u || stamped || tcols != 0u ||
                          sets->count < 1u || sets->count > (uint32_t)kMaxMulti))
    return kDynNotEligible;  // fused launches: 48- or 64-column tiles, whole K, contiguous groups
  const uint32_t wide_probe = XBIT_TUNING ? tcols >> 16 : 0u;  // tuning builds: measurement build of the 128-column tiles, carried in the upper half of tcols
  tcols &= 0xFFFFu;
  if (nl == 32 && (cpl != 4 || g_idx != nullptr || g.groupsize % 32u != 0u || (splits > 1u && wide_probe == 0u) || stamped)) return kDynNotEligible;  // 128-column tiles: 16-byte lanes, whole K
  // 17..32 rows by id (act-order, odd group sizes): the 32-row by-id build
  // Groups by id: act-order (g_idx given), or contiguous groups that may begin inside a k-block (size no multiple of 32:
  // 16, 48, 80 ... - no other fast kernel takes those) - the same build with the ids computed in the kernel.  48-column
  // tiles, whole K, every group in the LDS table.
  const uint32_t dq = g.groupsize / (32u * UB);
  const bool by_id = g_idx != nullptr || g.groupsize % 32u != 0u;
  if (by_id) {
    if (g_idx == nullptr && g.groupsize == 0u) return kDynNotEligible;
    if (splits > 1u || g.groups > 65535u || (uintptr_t)g_idx % 4u != 0u) return kDynNotEligible;
    cpl = 3;
    nl = 16;
  } else if (g.groupsize % (32u * UB) != 0u || (dq & (dq - 1u)) != 0u) {
    return kDynNotEligible;  // multiples of 32 the table-per-k-slot scheme cannot index (96, or 32 at 2 bits): xbit_gemv_mfma.h
  }
  // activations and scales may sit on any halfword (unaligned dword loads are served: profiles/r04/unaligned_probe.txt); the packed words are int32
  if (((uintptr_t)g.qweight | (uintptr_t)g.qzeros) % 4u != 0u || ((uintptr_t)g.a | (uintptr_t)g.scales) % 2u != 0u) return kDynNotEligible;
  const uint32_t d = by_id ? 1u : dq;
  if (nl != 16 && !(nl == 8 && cpl == 2) && !(nl == 32 && cpl == 4)) return kDynNotEligible;
  // columns a tile owns: the whole lane map unless the caller narrows it (a multiple of cpl: every lane is all in or all out)
  const uint32_t cn = (uint32_t)(nl * cpl);
  if (tcols == 0u) tcols = cn;
  if (tcols > cn || tcols % (uint32_t)cpl != 0u || (by_id && tcols != cn)) return kDynNotEligible;
  DynParams p;
  p.qweight = g.qweight;
  p.n = g.n;
  p.kw = g.kw;
  p.a = g.a;
  p.scales = g.scales;
  p.qzeros = g.qzeros;
  p.out = g.out;
  p.k = g.k;
  p.groups = g.groups;
  p.zw = g.zw;
  p.bias = g.bias;
  p.m = g.m;
  p.trace = trace;
  p.g_idx = g_idx;
  p.by_id = by_id;
  p.gshift = 0;
  while ((1u << p.gshift) < d) ++p.gshift;
  if (by_id) p.gshift = g.groupsize;  // read only when g_idx is null: the size of the contiguous groups
  const uint32_t lk = 64u / (uint32_t)nl, kblocks = (g.k + 31u) / 32u;
  p.tcols = tcols;
  p.tiles = (g.n + tcols - 1u) / tcols;
  p.wide_probe = wide_probe;
  p.sets = nullptr;
  GemvSets fused;
  if (sets != nullptr) {  // the grid: every set's tiles, one after the other
    fused = *sets;
    uint32_t end = 0;
    for (uint32_t j = 0; j < (uint32_t)kMaxMulti; ++j) {
      if (j < fused.count) {
        if (((uintptr_t)fused.qweight[j] | (uintptr_t)fused.qzeros[j]) % 4u != 0u || (uintptr_t)fused.scales[j] % 2u != 0u || fused.n[j] == 0u || fused.n[j] > g.n) return kDynNotEligible;
        end += (fused.n[j] + tcols - 1u) / tcols;
      }
      fused.tile_end[j] = j < fused.count ? end : 0xFFFFFFFFu;
    }
    p.tiles = end;
    p.sets = &fused;
  }
  p.units = (kblocks + lk * UB - 1u) / (lk * UB);
  p.splits = splits >= 2u && acc != nullptr ? splits : 1u;
  p.ups = (p.units + p.splits - 1u) / p.splits;
  p.acc = acc;
  if (p.splits > 1u && (p.ups < 8u || (p.splits - 1u) * p.ups >= p.units)) return kDynNotEligible;  // every range holds a unit for every wave
  // rows of the build.  The 1- and 4-row builds stage the whole K range of every row up front, a fixed number of rounds per
  // build (act-order: 4 = K 8192; contiguous: 8 / 4); a longer K takes the 8-row build instead, whose waves fetch their
  // activations unit by unit and whose table is staged in a loop - rows beyond m read zeros.  (Round 3 fell to the
  // round-2 kernels there: an act-order down-projection with K = 11008 at batch 1.)
  uint32_t mt = g.m > 16u ? 32u : (g.m > 8u ? 16u : (g.m > 4u ? 8u : (g.m > 1u ? 4u : 1u)));
  p.row_blocks = 1u;
  if (g.m > 32u) {  // 33..256 rows: blocks of 32 rows in grid.y of ONE launch of the 32-row build (contiguous groups, whole K, one weight set)
    p.row_blocks = (g.m + 31u) / 32u;
    if (by_id || splits > 1u || sets != nullptr || nl != 16 || UB != 1) return kDynNotEligible;
  }
  const int waves = nl == 32 ? kDynWideWaves : kDynWaves;
  const uint32_t st = 64u * (uint32_t)(waves / 2);  // staging threads
  uint32_t rounds = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    p.mt = mt;
    p.act_bytes = p.ups * lk * UB * 64u * mt;
    if (mt >= 8u) p.act_bytes = (uint32_t)waves * lk * UB * 64u * (mt + 1u);  // one unit per wave (octets padded by one row slot), refilled as the wave moves on
    if (mt >= 32u) {  // the reduction scratch reuses this region: it must hold waves * mt * (nl * cpl) floats (it does: 66 KB against <= 64 KB; kept as a guard)
      const uint32_t red = (uint32_t)waves * mt * (uint32_t)(nl * cpl) * 4u;
      if (p.act_bytes < red) p.act_bytes = red;
    }
    p.id_bytes = by_id ? p.ups * lk * UB * 64u : 0u;
    // rows of the LDS table: every group a k-slot of one K range can name (a range may begin inside a group); whole-K
    // launches cover at least `groups` rows so that the zero rows past the last group exist
    p.tab_groups = ((p.ups * lk - 1u + ((1u << p.gshift) - 1u)) >> p.gshift) + 1u;
    if (p.splits == 1u) {
      p.tab_groups = ((p.units * lk - 1u) >> p.gshift) + 1u;
      if (p.tab_groups < p.groups) p.tab_groups = p.groups;
    }
    if (by_id) p.tab_groups = p.groups;  // every group, whatever K
    const uint32_t chunks = mt >= 8u ? 0u : p.act_bytes / mt / (uint32_t)(Unpack<BITS>::kChunkK * 2);  // a staging thread takes one chunk of every row per round
    const uint32_t r_act = (chunks + st - 1u) / st, r_tab = (p.tab_groups * (uint32_t)nl + st - 1u) / st;
    const uint32_t r_gid = by_id && mt < 8u ? (p.id_bytes / 16u + st - 1u) / st : 0u;  // 8 ids per staging thread and round (8 / 16 rows: beyond 2 rounds in a loop)
    const uint32_t r_tab_static = by_id ? 0u : r_tab;  // the by-id build stages table entries beyond its static rounds in a loop
    rounds = r_act > r_tab_static ? (r_act > r_gid ? r_act : r_gid) : (r_tab_static > r_gid ? r_tab_static : r_gid);
    const uint32_t max_rounds = (by_id || mt == 4u || p.splits > 1u) ? 4u : 8u;
    if (attempt == 0 && mt < 8u && rounds > max_rounds && nl == 16 && !stamped && (by_id || mt == 4u) && (p.splits == 1u || cpl == 4)) {
      mt = 8u;  // 8-row build: any K
      continue;
    }
    break;
  }
  if (dyn_lds_bytes(p, nl, waves, cpl, (int)mt) > (by_id || mt >= 32u || nl == 32 ? kDynMaxLdsById : kDynMaxLds)) return kDynNotEligible;
  const uint64_t unit_bytes = (uint64_t)lk * UB * BITS * g.n * 4u;
  if ((uint64_t)p.units * unit_bytes >= (1ull << 31) || (uint64_t)g.kw * g.n * 4u >= (1ull << 31) || (uint64_t)g.groups * g.n * 2u >= (1ull << 31)) return kDynNotEligible;
  if (sets != nullptr) return cpl == 4 ? launch_dyn_multi_si<BITS, 4>(p, stream, rounds) : launch_dyn_multi_si<BITS, 3>(p, stream, rounds);
#define XBIT_DYN(CPL_, NL_)                                                                      \
  return nt ? launch_dyn_si<BITS, CPL_, NL_, true>(p, stream, rounds, stamped)                   \
            : launch_dyn_si<BITS, CPL_, NL_, XBIT_TUNING ? false : true>(p, stream, rounds, stamped)
  if (nl == 32) return nt ? launch_dyn_wide<BITS, true>(p, stream, rounds) : launch_dyn_wide<BITS, XBIT_TUNING ? false : true>(p, stream, rounds);
  if (nl == 8) { XBIT_DYN(2, 8); }
  if (cpl == 4) { XBIT_DYN(4, 16); }
  if (cpl == 3) { XBIT_DYN(3, 16); }
  XBIT_DYN(2, 16);
#undef XBIT_DYN
}

}  // namespace dyn

}  // namespace xbit
