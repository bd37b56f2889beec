// xbit_gemv_rows.h — 33..512 activation rows (small-batch prefill, wide speculative decode): the staged-operand GEMV's column groups with the
// activation tile SHARED inside a workgroup.  gfx950, bits 2..8, contiguous groups.
//
// Why a second kernel shape.  Above 32 rows gemv_dyn_kernel (xbit_gemv_dyn.h) takes 0.31 us per row on 4096x11008 whichever way the rows are cut
// (blocks of 32 rows in grid.y, a 64-row build: profiles/r04/sweep_rows_in_grid_y.txt): every workgroup owns 48-64 columns and pulls ALL
// M x K x 2 bytes of activations through its CU's vector-memory path - 1 MB at 128 rows for 98 KB of packed weights.  Here a workgroup of eight
// waves owns NCG column groups of 16 * CPL columns (NCG = 2 or 4) and one block of MT = 32 or 64 rows.  Its waves are (column group cg, k-slice ks);
// the NCG waves of one k-slice work on the SAME unit of K at the same time and share ONE LDS image of that unit's activations, which they fetch
// together - 1 / NCG of it each, one unit ahead, double-buffered - with one workgroup barrier per unit.  Activation bytes per weight byte drop by
// NCG, weight bytes and unpack instructions per output by MT / 32.  Everything else is gemv_dyn_kernel's: the packed layout and lane map (16 lanes x
// CPL adjacent columns, 4 k-slots per wave, a lane owns whole k-blocks), the row loads a few rows ahead of the MFMA step that needs them (Plan /
// Row), the bit-exact packed-fp16 W (subnormal form; spliced form for a tile with a scale too large for it), v_mfma_f32_16x16x32_f16 with MT / 16 A
// operands (rows 0-15, 16-31, ...) per unpacked B fragment, fp32 sums in a fixed order.
// Measured, chain period, w4 (profiles/r04/sweep_rows_shared_tile_v1.txt, _v2.txt, sweep_rows_320_512.txt; tools/microbench/rows_lab.hip is the harness
// the kernel was developed in): 4096x11008 at 64 / 128 / 256 / 512 rows 21.7 / 40.4 / 75.3 / 172.6 us (grid.y launch; 16-row passes at 512) ->
// 17.7 / 23.1 / 38.1 / 69.1; 8192x8192 at 128 rows 60.3 -> 34.1.  Long K (tables as a window, WIN below; profiles/r04/sweep_rows_long_k.txt, sweep_rows_long_k_w2.txt): 28672x8192 at 64 / 128 /
// 256 / 512 rows 101 / 192 / 380 / 826 us (32-row build in grid.y; 16-row passes at 512) -> 84 / 105 / 203 / 361; 2-bit at group 64: 11008x4096 at 256 rows 73 -> 45.5.
// What a workgroup's time follows (profiles/r04/rows_lab_ablations.txt, rows_pmc_summary.txt, rows_lab_64_rows.txt): no single stage - but across the six
// shapes it is 28-38 ns per wave-level load instruction the workgroup issues (K = 4096: 2 x 48 columns x 32 rows 512 loads, 16.7 us; 4 x 48 x 32: 768,
// 24.6; 2 x 48 x 64: 768, 21.4; 4 x 48 x 64: 1024, 33.4), whatever else is removed or added: a CU's vector-memory pipeline is the serial resource.
#pragma once

#include "xbit_gemv_dyn.h"

namespace xbit {
namespace rows {

using dyn::f4_t;
using dyn::h8_t;
using dyn::make_rsrc;
using dyn::Plan;
using dyn::Row;
using dyn::rsrc_t;
using dyn::u32x2_t;

constexpr int kRowsWaves = 8, kRowsBlock = 32;  // waves per workgroup, rows per workgroup (or 64)

// Template parameters: BITS weight width (2..8; one k-block per lane and unit, two at 2 bits), CPL adjacent columns per lane (3 or 4), NCG column groups per
// workgroup (2 or 4; NW / NCG k-slices), NW waves, MT rows per workgroup (32 or 64: MT / 16 A operands per unpacked B fragment), WIN the tables as a window.
//
// WIN (long K, or 64 rows per workgroup beyond K of about 4.6 K: the (scale | -(zero + bias) * scale) tables of ALL groups do not fit in LDS beside the
// images).  The tables are a ring of two windows of WG = 4 * NK groups per column group (16 KB in all).  A window is what the workgroup's waves cross in
// 2^gshift steps; every thread owns ONE entry of a window (NCG * WG * 16 lanes = 64 * NW), requests it from HBM one step before the window opens - behind
// that step's activation chunks, in front of its weight rows, so waiting for it costs nothing the activation image does not already wait for - and puts it
// in place in front of the barrier of the window's first step.  The other half of the ring is what slower waves may still be reading: its last reader
// passed the previous step's barrier before any wave can reach this store.  Whether some scale of the tile needs the spliced form is settled by one pass
// over the tile's scales at the opening.
template <int BITS, int CPL, int NCG, int NW, int MT, bool WIN = false>
__global__ void __launch_bounds__(64 * NW)
    gemv_rows_kernel(const uint32_t* qweight, uint32_t n, uint32_t xtiles, uint32_t kw, uint32_t units, const uint16_t* a, const uint16_t* scales, const uint32_t* qzeros,
                     uint16_t* out, uint32_t k, uint32_t groups, uint32_t zw, uint32_t bias, uint32_t m, uint32_t tab_groups, uint32_t gshift, uint32_t red_dwords) {
  constexpr int UB = dyn::dyn_ub<BITS>(), NL = 16, LK = 4, NK = NW / NCG, LA = dyn::kDynLookahead, MG = MT / 16;  // MG: A operands (blocks of 16 rows) per unpacked B fragment
  using P = Plan<BITS, UB, LA>;
  constexpr int R = P::R, S = P::S;
  constexpr int CN = NL * CPL;
  constexpr int CHK = Unpack<BITS>::kChunkK, CPT = CHK / 8;
  constexpr int OST = (MT + 1) * 4;          // dwords per octet of an activation image (one spare row slot: the 16 lanes that store a row's octets and the
                                             // 16 that read an octet's rows both touch 64 different banks)
  constexpr int IMG = LK * UB * 4 * OST;     // dwords of one unit's image
  constexpr int TPR = LK * UB * (32 / CHK);  // chunks (CHK k of one row) per row and unit
  constexpr int TASKS = MT * TPR;            // chunks per unit
  constexpr int AR = TASKS / (64 * NCG);     // rounds per wave: the NCG sibling waves fetch a unit together
  static_assert(NCG == 2 || NCG == 4, "column groups per workgroup");
  static_assert(MT == 32 || MT == 64, "rows per workgroup");
  static_assert(TASKS % (64 * NCG) == 0, "a unit's chunks split evenly over the sibling waves");
  constexpr uint32_t WG = 4 * NK, RING = 2 * WG;  // WIN: groups per window, groups in the ring
  static_assert(NCG * WG * NL == 64 * NW, "one table entry of a window per thread");
  constexpr uint32_t kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];
  // LDS: [activation images: NK k-slices x 2 buffers x IMG][tables: NCG x tab_groups x NL lanes x 16 B][bad flags: NW dwords]; the reduction scratch (NW x MT x CN
  // floats) REUSES the front of it once every wave is done with images and tables
  uint32_t* const s_img = s_rows;
  uint32_t* const s_tab = s_img + NK * 2 * IMG;
  uint32_t* const s_bad = s_rows + red_dwords;  // behind whichever is larger: images + tables, or the reduction scratch
  float* const s_red = reinterpret_cast<float*>(s_rows);

  const int tid = threadIdx.x, lane = tid & 63, kl = lane / NL, nl = lane % NL;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wv % NCG, ks = wv / NCG;  // column group, k-slice: the sibling waves of a k-slice are neighbours
  // Workgroup id -> (column tile xt, row block rb).  Ids go round the 8 XCDs; XCD x owns a contiguous run of column tiles (xcd_tile's split) and ALL row
  // blocks of each, one after the other: the row blocks of a tile then share its packed rows through ONE L2 instead of fetching them once per XCD
  // (4096x11008 w4 at 128 rows with the row blocks in grid.y: 3.9x the algorithmic bytes over the fabric, profiles/r04/bench_before_xcd_rows/).  The grid
  // is padded to whole runs; a workgroup whose slot is past its XCD's run leaves at once.
  const uint32_t row_blocks = (m + (uint32_t)MT - 1u) / (uint32_t)MT;
  const uint32_t xq = xtiles >> 3, xr = xtiles & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t xl = slot / row_blocks, rb = slot % row_blocks;
  if (xl >= xq + (xcd < xr ? 1u : 0u)) return;
  const uint32_t xt = (xcd < xr ? xcd * (xq + 1u) : xr * (xq + 1u) + (xcd - xr) * xq) + xl, tile = xt * (uint32_t)NCG + (uint32_t)cg;
  a += (size_t)rb * MT * k;
  out += (size_t)rb * MT * n;
  const uint32_t rows = m - rb * MT < (uint32_t)MT ? m - rb * MT : (uint32_t)MT;
  const uint32_t row_bytes = n * 4u;
  const uint32_t col = tile * CN + (uint32_t)(CPL * nl);
  const bool col_ok = col < n;
  const rsrc_t r_w = make_rsrc(qweight, kw * row_bytes);
  // odd k: rows that begin on an odd halfword are read unaligned, and the last dword of the last row may reach two bytes past the tensor (xbit_gemv_dyn.h)
  const rsrc_t r_a = make_rsrc(a, ((rows * k * 2u + 3u) & ~3u) + (((k & 1u) != 0u && rows > 1u) ? 4u : 0u));
  const rsrc_t r_s = make_rsrc(scales, groups * n * 2u);
  const rsrc_t r_z = make_rsrc(qzeros, groups * zw * 4u);

  // ---- this wave's share of a unit's activations: chunk t = (cg * AR + r) * 64 + lane: row t / TPR, chunk t % TPR (a row's chunks sit in neighbouring lanes)
  u32x4_t wa_raw[AR][CPT];
  uint32_t* const my_img = s_img + (uint32_t)ks * (uint32_t)(2 * IMG);
  auto act_issue = [&](uint32_t u) {
    static_for<AR>([&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      const uint32_t t = (uint32_t)((cg * AR + r) * 64 + lane), mi = t / (uint32_t)TPR, cu = t % (uint32_t)TPR;
      const uint32_t k0 = u * (uint32_t)(LK * UB * 32) + cu * (uint32_t)CHK;
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        const uint32_t kq = k0 + 8u * (uint32_t)q;  // rows beyond `rows` fall off the end of the tensor (zeros); k beyond the row must not read on into the next row
        wa_raw[r][q] = __builtin_amdgcn_raw_buffer_load_b128(r_a, (mi * k + kq) * 2u | (kq >= k ? kOob : 0u), 0, 0);
      }
    });
  };
  auto act_store = [&](uint32_t u, int buf) {
    static_for<AR>([&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      const uint32_t t = (uint32_t)((cg * AR + r) * 64 + lane), mi = t / (uint32_t)TPR, cu = t % (uint32_t)TPR;
      uint32_t h[CHK / 2];
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        h[4 * q + 0] = wa_raw[r][q].x;
        h[4 * q + 1] = wa_raw[r][q].y;
        h[4 * q + 2] = wa_raw[r][q].z;
        h[4 * q + 3] = wa_raw[r][q].w;
      }
      if (k & 7u) {  // wave-uniform: a row length that is no multiple of 8 - the piece that holds element k - 1 read on into the next row
        const int left = (int)k - (int)(u * (uint32_t)(LK * UB * 32) + cu * (uint32_t)CHK);
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
          const uint32_t sel = (uint32_t)(2 * (klo & 1)) | ((uint32_t)(2 * (klo & 1) + 1) << 8) | ((uint32_t)(4 + 2 * (khi & 1)) << 16) | ((uint32_t)(5 + 2 * (khi & 1)) << 24);
          o[e] = __builtin_amdgcn_perm(h[khi >> 1], h[klo >> 1], sel);
        }
        *reinterpret_cast<u32x4_t*>(my_img + buf * IMG + (cu * (uint32_t)CPT + (uint32_t)q) * (uint32_t)OST + mi * 4u) = o;
      }
    });
  };

  // ---- weight rows: the request pipeline of gemv_dyn_kernel over units ks, ks + NK, ... (two whole units ahead instead of LA + 1 rows gained nothing:
  // profiles/r04/rows_lab_two_units_ahead.txt)
  const uint32_t lane_off = col_ok ? (uint32_t)(kl * R) * row_bytes + col * 4u : kOob;
  const uint32_t unit_bytes = (uint32_t)(LK * R) * row_bytes;
  Row<CPL, true> ra[R], rbuf[R];
  uint32_t cur = (uint32_t)ks, nxt = cur + (uint32_t)NK, nn = cur + (uint32_t)(2 * NK);
  const uint32_t past_end = kw * row_bytes;
  auto unit_off = [&](uint32_t u) -> uint32_t { return u < units ? u * unit_bytes : past_end; };
  auto request = [&](Row<CPL, true>(&X)[R], Row<CPL, true>(&Y)[R], auto from_c, auto to_c) {
    constexpr int from = decltype(from_c)::value, to = decltype(to_c)::value;
    static_for<(to > from ? to - from : 0)>([&](auto d) {
      constexpr int f = from + decltype(d)::value;
      if constexpr (f < R) X[f].load(r_w, lane_off, unit_off(cur) + (uint32_t)f * row_bytes);
      else if constexpr (f < 2 * R) Y[f - R].load(r_w, lane_off, unit_off(nxt) + (uint32_t)(f - R) * row_bytes);
      else X[f - 2 * R].load(r_w, lane_off, unit_off(nn) + (uint32_t)(f - 2 * R) * row_bytes);
    });
  };
  act_issue(cur);
  request(ra, rbuf, std::integral_constant<int, 0>{}, std::integral_constant<int, P::open>{});
  __builtin_amdgcn_sched_barrier(0);

  // ---- the (scale | -(zero + bias) * scale) table of every column group, staged by all threads in a loop (a launch of this size is not bound by its opening)
  bool bad = false;
  // one entry: CPL columns of one group.  sv / zv: the 8 bytes that hold the scales from halfword (h0 & ~1) on, the zero-points from dword (scol * BITS) / 32 on
  auto tab_entry = [&](const u32x2_t sv, const u32x2_t zv, uint32_t scol) -> u32x4_t {
    const uint32_t sh = (scol & 1u) * 16u;  // n is even: the parity of g * n + scol is scol's
    const uint32_t lo = __builtin_amdgcn_alignbit(sv.y, sv.x, sh), hi = sv.y >> sh;
    const uint32_t sb[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
    const uint32_t zwin = __builtin_amdgcn_alignbit(zv.y, zv.x, (scol * BITS) & 31u);
    u32x4_t ent = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const _Float16 s = as_h((uint16_t)sb[c]);
      const uint32_t z = ((zwin >> (c * BITS)) & Unpack<BITS>::kMask) + bias;
      ent[c] = (uint32_t)as_u16(s) | ((uint32_t)as_u16(neg_zero_times_scale(z, s)) << 16);
    }
    return ent;
  };
  // WIN: this thread's entry of a window - column group t_cge, group t_g of the window, lane t_snl's columns
  const uint32_t t_cge = (uint32_t)tid / (WG * NL), t_g = ((uint32_t)tid % (WG * NL)) / NL, t_snl = (uint32_t)tid % NL;
  const uint32_t t_scol = (xt * (uint32_t)NCG + t_cge) * CN + (uint32_t)CPL * t_snl;
  u32x2_t tw_s = {0u, 0u}, tw_z = {0u, 0u};
  auto tab_issue = [&](uint32_t w) {
    const uint32_t g = w * WG + t_g;
    const bool in = g < groups;  // a window may reach past the last group (zeros: nothing reads them)
    tw_s = __builtin_amdgcn_raw_buffer_load_b64(r_s, in ? ((g * n + t_scol) & ~1u) * 2u : kOob, 0, 0);
    tw_z = __builtin_amdgcn_raw_buffer_load_b64(r_z, in ? (g * zw + ((t_scol * BITS) >> 5)) * 4u : kOob, 0, 0);
  };
  auto tab_store = [&](uint32_t w) {
    const uint32_t g = w * WG + t_g;
    *reinterpret_cast<u32x4_t*>(s_tab + ((t_cge * RING + (g & (RING - 1u))) * (uint32_t)NL + t_snl) * 4u) = tab_entry(tw_s, tw_z, t_scol);
  };
  if constexpr (!WIN) {
    for (uint32_t e = (uint32_t)tid; e < (uint32_t)NCG * tab_groups * (uint32_t)NL; e += (uint32_t)(64 * NW)) {
      const uint32_t cge = e / (tab_groups * (uint32_t)NL), ee = e % (tab_groups * (uint32_t)NL), g = ee / NL, snl = ee % NL;
      const uint32_t scol = (xt * (uint32_t)NCG + cge) * CN + (uint32_t)CPL * snl;
      const uint32_t h0 = g * n + scol;  // first scale (half index); n is even
      const u32x2_t sv = __builtin_amdgcn_raw_buffer_load_b64(r_s, (h0 & ~1u) * 2u, 0, 0);
      const uint32_t sh = (h0 & 1u) * 16u;
      const uint32_t lo = __builtin_amdgcn_alignbit(sv.y, sv.x, sh), hi = sv.y >> sh;
      uint32_t sb[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
      const uint32_t bit = scol * BITS, zsh = bit & 31u;
      const u32x2_t zv = __builtin_amdgcn_raw_buffer_load_b64(r_z, (g * zw + (bit >> 5)) * 4u, 0, 0);  // a read past the row end is harmless
      const uint32_t zwin = __builtin_amdgcn_alignbit(zv.y, zv.x, zsh);
      u32x4_t ent = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const _Float16 s = as_h((uint16_t)sb[c]);
        const uint32_t z = ((zwin >> (c * BITS)) & Unpack<BITS>::kMask) + bias;
        bool ok;
        (void)sub_scale<BITS>(s, ok);
        bad = bad || (!ok && g < groups && scol + (uint32_t)c < n);
        ent[c] = (uint32_t)as_u16(s) | ((uint32_t)as_u16(neg_zero_times_scale(z, s)) << 16);
      }
      *reinterpret_cast<u32x4_t*>(s_tab + (cge * tab_groups * (uint32_t)NL + ee) * 4u) = ent;
    }
  } else {
    tab_issue(0u);
    for (uint32_t e = (uint32_t)tid; e < (uint32_t)NCG * groups * (uint32_t)NL; e += (uint32_t)(64 * NW)) {  // the scales of the whole tile: do all take the subnormal form?
      const uint32_t cge = e / (groups * (uint32_t)NL), ee = e % (groups * (uint32_t)NL), g = ee / NL, snl = ee % NL;
      const uint32_t scol = (xt * (uint32_t)NCG + cge) * CN + (uint32_t)CPL * snl, h0 = g * n + scol;
      const u32x2_t sv = __builtin_amdgcn_raw_buffer_load_b64(r_s, (h0 & ~1u) * 2u, 0, 0);
      const uint32_t sh = (h0 & 1u) * 16u;
      const uint32_t lo = __builtin_amdgcn_alignbit(sv.y, sv.x, sh), hi = sv.y >> sh;
      const uint32_t sb[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        bool ok;
        (void)sub_scale<BITS>(as_h((uint16_t)sb[c]), ok);
        bad = bad || (!ok && scol + (uint32_t)c < n);
      }
    }
    tab_store(0u);
  }
  {
    const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (lane == 0) s_bad[wv] = wave_bad ? 1u : 0u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tables and flags are in place (the row and activation loads stay in flight)
  uint32_t any_bad = 0;
#pragma unroll
  for (int q = 0; q < NW / 4; ++q) {
    const u32x4_t b = *reinterpret_cast<const u32x4_t*>(s_bad + 4 * q);
    any_bad |= b.x | b.y | b.z | b.w;
  }
  const bool fast = __builtin_amdgcn_readfirstlane((int)any_bad) == 0;

  f4_t acc[MG][CPL];
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[g][c] = f4_t{0.f, 0.f, 0.f, 0.f};
  const uint32_t sub_mask = opaque_sgpr(Unpack<BITS>::kSubMask2);
  const typename Unpack<BITS>::Consts uk = Unpack<BITS>::consts();
  const uint32_t a_lane = (uint32_t)(kl * UB) * (uint32_t)(4 * OST) + (uint32_t)nl * 4u;
  const uint32_t* const my_tab = s_tab + (uint32_t)cg * (WIN ? RING : tab_groups) * (uint32_t)(NL * 4);
  const uint32_t ws_mask = (1u << gshift) - 1u;  // WIN: a window is 2^gshift steps
  uint32_t stepi = 0;

  auto step = [&](const Row<CPL, true>(&X)[R], const uint32_t* a_unit, auto s_c, auto fast_c, const half2_t(&s2)[CPL], const half2_t(&nzs2)[CPL], const half2_t(&ss2)[CPL]) {
    constexpr int s = decltype(s_c)::value, j = s / 4, oct = s % 4;
    constexpr bool FAST = decltype(fast_c)::value;
    uint32_t wc[CPL][BITS];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
      for (int i = 0; i < BITS; ++i) wc[c][i] = X[j * BITS + i].v[c];
    u32x4_t braw[CPL];
    static_for<4>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value, pr = 4 * oct + e;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if constexpr (FAST) {
          const uint32_t q = Unpack<BITS>::template sub_pair<pr>(wc[c], sub_mask);
          braw[c][e] = as_u32(__builtin_elementwise_fma(as_h2(q), ss2[c], nzs2[c]));
        } else {
          const uint32_t q = Unpack<BITS>::template pair<pr>(wc[c], uk);
          braw[c][e] = as_u32(dequant_pair<Unpack<BITS>::pair_exp(pr)>(q, s2[c], nzs2[c]));
        }
      }
    });
#pragma unroll
    for (int g = 0; g < MG; ++g) {  // rows 0-15, 16-31, ...: the same B fragments
      const u32x4_t araw = *reinterpret_cast<const u32x4_t*>(a_unit + (j * 4 + oct) * OST + 64 * g);
      const h8_t a_frag = __builtin_bit_cast(h8_t, araw);
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_frag, __builtin_bit_cast(h8_t, braw[c]), acc[g][c], 0, 0, 0);
    }
  };

  // One unit per wave and step; all waves keep step (one workgroup barrier per step): behind it the siblings' shares of this unit's image are complete, and
  // the other buffer - read during the previous step - is free for the next unit's image.  (Reading the unit's A fragments and table entry ahead of the first
  // unpack gains nothing: profiles/r04/rows_lab_prefetch.txt.)
  auto process = [&](Row<CPL, true>(&X)[R], Row<CPL, true>(&Y)[R], int buf, auto fast_c) {
    const bool live = cur < units;  // wave-uniform
    if (live) act_store(cur, buf);  // this unit's activations: requested one step ago, the oldest loads in flight
    if constexpr (WIN) {
      if (stepi != 0u && (stepi & ws_mask) == 0u) tab_store(stepi >> gshift);  // a window opens with this step: its entries were requested one step ago
    }
    act_issue(nxt);  // the next unit's: in front of every row this step requests for it
    if constexpr (WIN) {
      if (((stepi + 1u) & ws_mask) == 0u) tab_issue((stepi + 1u) >> gshift);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS stores only: the row loads stay in flight
    half2_t s2[CPL], nzs2[CPL], ss2[CPL];
    if (live) {
      const uint32_t gi = ((cur * (uint32_t)LK + (uint32_t)kl) >> gshift) & (WIN ? RING - 1u : 0xFFFFFFFFu);
      const u32x4_t ent = *reinterpret_cast<const u32x4_t*>(my_tab + (gi * (uint32_t)NL + (uint32_t)nl) * 4u);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        s2[c] = as_h2(__builtin_amdgcn_perm(ent[c], ent[c], 0x01000100u));
        nzs2[c] = as_h2(__builtin_amdgcn_perm(ent[c], ent[c], 0x03020302u));
        constexpr int L = Unpack<BITS>::kSubScaleLog2;  // scale * 2^(14 + BITS) in two exact steps
        ss2[c] = (s2[c] * splat((_Float16)(float)(1 << 8))) * splat((_Float16)(float)(1 << (L - 8)));
      }
    }
    const uint32_t* const a_unit = my_img + buf * IMG + a_lane;
    static_for<S>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if (live) step(X, a_unit, s_c, fast_c, s2, nzs2, ss2);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int from = P::need(s) + LA, to = P::need(s + 1) + LA;
      static_assert(to <= 2 * R || to - 1 - 2 * R < P::first_row(s + 1), "register of a live row reused");
      request(X, Y, std::integral_constant<int, from>{}, std::integral_constant<int, to>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    cur = nxt;
    nxt = nn;
    nn += (uint32_t)NK;
    ++stepi;
  };
  const uint32_t nsteps = (units + (uint32_t)NK - 1u) / (uint32_t)NK;  // the same for every wave: every wave passes every barrier
  if (fast) {
    for (uint32_t i = 0; i < nsteps; i += 2u) {
      process(ra, rbuf, 0, std::true_type{});
      process(rbuf, ra, 1, std::true_type{});
    }
  } else {  // some scale of the tile is too large for the subnormal form (or not finite): the spliced form, same W bit for bit
    for (uint32_t i = 0; i < nsteps; i += 2u) {
      process(ra, rbuf, 0, std::false_type{});
      process(rbuf, ra, 1, std::false_type{});
    }
  }

  // ---- sum over the k-slices of every column group (the scratch reuses images and tables: every wave is done with them behind this barrier), store
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int g = 0; g < MG; ++g)
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_red[(wv * MT + 16 * g + 4 * (lane >> 4) + r) * CN + nl * CPL + c] = acc[g][c][r];
  __syncthreads();
  for (int e0 = tid; e0 < NCG * MT * CN; e0 += 64 * NW) {
    const int cge = e0 / (MT * CN), mi = (e0 / CN) % MT, cc = e0 % CN;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NK; ++q) sum += s_red[((q * NCG + cge) * MT + mi) * CN + cc];
    const uint32_t n_out = (xt * (uint32_t)NCG + (uint32_t)cge) * CN + (uint32_t)cc;
    if (n_out < n && (uint32_t)mi < rows) out[(size_t)mi * n + n_out] = as_u16((_Float16)sum);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------

template <int BITS, int CPL, int NCG, int NW, int MT, bool WIN = false>
int launch_rows_one(const GemvParams& g, hipStream_t stream, uint32_t units, uint32_t tab_groups, uint32_t gshift) {
  constexpr int CN = 16 * CPL, NK = NW / NCG, IMG = 4 * dyn::dyn_ub<BITS>() * 4 * (MT + 1) * 4;
  const uint32_t tiles = (g.n + CN - 1u) / CN, xtiles = (tiles + NCG - 1u) / NCG, row_blocks = (g.m + MT - 1u) / MT;
  if (WIN) tab_groups = 2u * 4u * (uint32_t)NK;  // the ring of two windows
  const uint32_t front = (uint32_t)(NK * 2 * IMG) + (uint32_t)NCG * tab_groups * 64u, red = (uint32_t)(NW * MT * CN);  // dwords
  uint32_t red_dwords = front > red ? front : red;
  const size_t lds = (size_t)red_dwords * 4u + (size_t)NW * 4u;
  if (lds > dyn::kDynMaxLdsById) return kGemvDynNotEligible;
  auto kern = gemv_rows_kernel<BITS, CPL, NCG, NW, MT, WIN>;
  GemvParams q = g;
  uint32_t xt = xtiles, un = units, tg = tab_groups, gs = gshift;
  void* args[] = {&q.qweight, &q.n, &xt, &q.kw, &un, &q.a, &q.scales, &q.qzeros, &q.out, &q.k, &q.groups, &q.zw, &q.bias, &q.m, &tg, &gs, &red_dwords};
  return launch_raw_dynamic(reinterpret_cast<const void*>(kern), dim3(8u * ((xtiles + 7u) / 8u) * row_blocks), dim3(64 * NW), lds, stream, args);  // whole runs of column tiles per XCD x row blocks
}

// What a launch costs, in us: rounds of 256 workgroups (one per CU: the LDS of one fills more than half a CU's) x the time of one workgroup.  A workgroup's
// time is linear in K, and the slope depends on its shape only (profiles/r04/sweep_rows_shared_tile_v1.txt, rows_lab_64_rows.txt: 4096 / 8192 / 11008 rows of K on
// 86..460 workgroups; nothing in it - weight stream, activation fetch, barrier, unpack, MFMA - dominates: profiles/r04/rows_lab_ablations.txt,
// rows_pmc_summary.txt); 7 % more when the round is full (shared L2 and fabric); 5..8-bit rows are 12-byte / 16-byte loads of more rows per k-block, the
// 48-column (12-byte) ones relatively dearer.  64 rows per workgroup cost 1.3x the time of 32.  The model picks the shape; tools/check_rules.py holds it to
// 1.08x of the best of the forced ones on held-out layers.
inline double rows_cost_us(int bits, uint32_t k, uint32_t n, uint32_t m, int cpl, int ncg, int mt) {
  const uint32_t tiles = (n + 16u * (uint32_t)cpl - 1u) / (16u * (uint32_t)cpl), row_blocks = (m + (uint32_t)mt - 1u) / (uint32_t)mt;
  const uint32_t wgs = ((tiles + (uint32_t)ncg - 1u) / (uint32_t)ncg) * row_blocks, rounds = (wgs + 255u) / 256u;
  double c0 = ncg == 2 ? (cpl == 3 ? 3.0 : 4.0) : (cpl == 3 ? 5.8 : 6.2), slope = ncg == 2 ? (cpl == 3 ? 13.7 : 14.0) : (cpl == 3 ? 18.8 : 23.4);
  if (mt == 64) c0 = ncg == 2 ? 4.0 : 8.0, slope = ncg == 2 ? 17.4 : 25.4;  // 48-column groups only
  double t = c0 + slope * (double)k / 4096.0;
  if (bits > 4) t *= 1.0 + (double)(bits - 4) / 4.0 * (cpl == 3 ? 0.32 : 0.15);
  const double per_round = (double)wgs / (double)rounds / 256.0;
  return (double)rounds * t * (1.0 + 0.07 * (per_round < 1.0 ? per_round : 1.0));
}

// 33..512 (kGemvRowsMax) rows of a layer with contiguous groups whose size is a power-of-two multiple of 32 (64 at 2 bits), any K; kGemvDynNotEligible otherwise.  Columns per
// group (48 / 64), groups per workgroup (2 / 4) and rows per workgroup (32; 64 with 48-column groups) by rows_cost_us, cheapest first; a shape whose tables do
// not fit in LDS is passed over.  `cpl` / `ncg` / `mt` other than 0 force the shape (measurement: tools/check_rules.py compares the choice with the others).
template <int BITS>
int launch_rows_bits(const GemvParams& g, hipStream_t stream, int force_cpl = 0, int force_ncg = 0, int force_mt = 0) {
  {
    constexpr uint32_t UB = (uint32_t)dyn::dyn_ub<BITS>();  // k-blocks per lane and unit
    if (g.m <= (uint32_t)kRowsBlock || g.m > kGemvRowsMax) return kGemvDynNotEligible;
    const uint32_t dq = g.groupsize / (32u * UB);
    if (g.groupsize % (32u * UB) != 0u || (dq & (dq - 1u)) != 0u) return kGemvDynNotEligible;
    if (((uintptr_t)g.qweight | (uintptr_t)g.qzeros) % 4u != 0u || ((uintptr_t)g.a | (uintptr_t)g.scales) % 2u != 0u) return kGemvDynNotEligible;
    uint32_t gshift = 0;
    while ((1u << gshift) < dq) ++gshift;
    const uint32_t kblocks = (g.k + 31u) / 32u, units = (kblocks + 4u * UB - 1u) / (4u * UB);
    uint32_t tab_groups = ((units * 4u - 1u) >> gshift) + 1u;
    if (tab_groups < g.groups) tab_groups = g.groups;
    const uint64_t unit_bytes = (uint64_t)4 * UB * BITS * g.n * 4u;
    if ((uint64_t)units * unit_bytes >= (1ull << 31) || (uint64_t)g.kw * g.n * 4u >= (1ull << 31) || (uint64_t)g.groups * g.n * 2u >= (1ull << 31) ||
        (uint64_t)g.m * g.k * 2u >= (1ull << 31))
      return kGemvDynNotEligible;
    // every shape with the tables of all groups in LDS first; where they do not fit beside the images (K beyond about 14 K at 32 rows per workgroup, about 4.6 K at 64), the
    // two-groups-of-columns shapes with the tables as a window (WIN)
    auto launch = [&](int cpl, int ncg, int mt) -> int {
      int st = kGemvDynNotEligible;
      if (mt == 64) {  // four A operands per unpacked B fragment: 48-column groups only (64-column ones never won: profiles/r04/rows_lab_64_rows.txt)
        if constexpr (UB == 1) {
          if (cpl != 3) return kGemvDynNotEligible;
          st = ncg == 4 ? launch_rows_one<BITS, 3, 4, kRowsWaves, 64>(g, stream, units, tab_groups, gshift) : launch_rows_one<BITS, 3, 2, kRowsWaves, 64>(g, stream, units, tab_groups, gshift);
          if (st == kGemvDynNotEligible && ncg == 2) st = launch_rows_one<BITS, 3, 2, kRowsWaves, 64, true>(g, stream, units, tab_groups, gshift);
        }
        return st;  // 2-bit: a unit is 256 k - the images of 64 rows do not fit in LDS
      }
      if (cpl == 4) {
        st = ncg == 4 ? launch_rows_one<BITS, 4, 4, kRowsWaves, 32>(g, stream, units, tab_groups, gshift) : launch_rows_one<BITS, 4, 2, kRowsWaves, 32>(g, stream, units, tab_groups, gshift);
        if (st == kGemvDynNotEligible && ncg == 2) st = launch_rows_one<BITS, 4, 2, kRowsWaves, 32, true>(g, stream, units, tab_groups, gshift);
        return st;
      }
      st = ncg == 4 ? launch_rows_one<BITS, 3, 4, kRowsWaves, 32>(g, stream, units, tab_groups, gshift) : launch_rows_one<BITS, 3, 2, kRowsWaves, 32>(g, stream, units, tab_groups, gshift);
      if (st == kGemvDynNotEligible && ncg == 2) st = launch_rows_one<BITS, 3, 2, kRowsWaves, 32, true>(g, stream, units, tab_groups, gshift);
      return st;
    };
    if (force_cpl != 0 || force_ncg != 0 || force_mt != 0) {
      if ((force_cpl != 3 && force_cpl != 4) || (force_ncg != 2 && force_ncg != 4) || (force_mt != 32 && force_mt != 64)) return kGemvDynNotEligible;
      return launch(force_cpl, force_ncg, force_mt);
    }
    struct Shape {
      int cpl, ncg, mt;
      double cost;
    } shapes[6] = {{3, 2, 32, 0.0}, {4, 2, 32, 0.0}, {3, 4, 32, 0.0}, {4, 4, 32, 0.0}, {3, 2, 64, 0.0}, {3, 4, 64, 0.0}};
    const int count = g.m > 64u && UB == 1u ? 6 : 4;  // up to 64 rows a 64-row workgroup has no second block to amortise over: measured slower
    for (int i = 0; i < count; ++i) shapes[i].cost = rows_cost_us(BITS, g.k, g.n, g.m, shapes[i].cpl, shapes[i].ncg, shapes[i].mt);
    for (int i = 0; i < count; ++i) {  // cheapest first
      int best = i;
      for (int j = i + 1; j < count; ++j)
        if (shapes[j].cost < shapes[best].cost) best = j;
      const Shape sh = shapes[best];
      shapes[best] = shapes[i];
      shapes[i] = sh;
      const int st = launch(sh.cpl, sh.ncg, sh.mt);
      if (st != kGemvDynNotEligible) return st;
    }
    return kGemvDynNotEligible;
  }
}

}  // namespace rows
}  // namespace xbit
