// xbit_gemv_tuning.h — the measurement variants (xbit_gemv_options.variant) of libxbitops_hip_tuning.so.  Included by xbit_gemv_mfma.h
// only under -DXBIT_TUNING=1: the production library contains none of this and rejects options.variant.  Every variant computes the
// production result; tools/ (gemv_sweep.py, sweep_*.py, check_rules.py, gemv_trace*.py) and a few A/B tests select them per call.
// tools/README.md lists what each number is for.
//
//   hooks on the production rule (launch_by_rules with RuleHooks):
//     112-114  default tile width forced to 32 / 48 / 64 columns        120 / 121  at most 8 / 16 waves per workgroup (round-2 default path)
//     122 / 123  nontemporal / plain weight loads (round-2 default path)  398  round 2's rules: never the staged-operand kernel, no fill-once plan
//     399  never the staged-operand kernel, never split                  202 / 205  scored split plan on 32- / 64-column tiles
//     330 / 332 / 334  scored split plan on the staged-operand kernel's split build (planned / 32- / 64-column tiles)      336 = the production rule
//   forced launches of the staged-operand kernel (xbit_gemv_dyn.h):
//     300  automatic tile   301  16-column tiles   302-304  32 / 48 / 64 columns   310-314  the same with plain loads   320  stamped   321  v_dot2 accumulate
//     340  128-column tiles   341-343  128- / 64- / 48-column lane maps narrowed to fill the chip   344-347  probes of the 128-column tiles
//     348  more than 32 rows as blocks of 32 in grid.y of ONE launch   349  ... as launches of 32 rows
//     350  more than 32 rows: column groups sharing the activation image (xbit_gemv_rows.h), shape chosen; 351-354 forced: 2 x 48, 4 x 48, 2 x 64, 4 x 64 columns x 32 rows;
//          355 / 356: 2 x 48 / 4 x 48 columns x 64 rows
//   forced launches of the round-2 kernel (M = 1 unless noted): 11-25, 28, 39, 51-56, 63, 64, 105-109, 135-140, 164, 170-177, 183, 190-192, 206,
//     214, 229-243, 272, 274; 16- / 4-row passes: 180, 181, 188, 189, 193, 197-200
#pragma once

namespace xbit {
namespace tuning {

template <int BITS>
int forced_dyn(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts, int variant) {
  const RuleHooks none;
  const rules::Call c = make_call<BITS>(p, none);
  const bool small_n = (c.flags & rules::kSmallN) != 0u;
  if (variant >= 344 && variant <= 347) {  // probes of the 128-column tiles: 344 K split opts.split ways (16 waves), 345 the same with 8 waves, 346 whole K with 8 waves, 347 lookahead 6
    const uint32_t probe = variant == 344 ? 1u : (variant <= 346 ? 2u : 3u);
    const uint32_t ways = variant <= 345 ? (uint32_t)(opts.split >= 2 ? opts.split : 2) : 1u;
    GemvWorkspace ws;
    if (ways > 1u && !gemv_workspace(stream, (size_t)((p.n + 127u) / 128u) * 128u, opts, &ws)) return kGemvDynNotEligible;
    return launch_gemv_dyn(p, BITS, stream, 4, 32, true, false, nullptr, ways, ways > 1u ? ws.acc : nullptr, nullptr, probe << 16);
  }
  if (variant >= 340 && variant <= 343) {  // 128-column tiles (340), tiles narrowed so that their count fills the chip (341: 128-column lane map, 342 / 343: 64 / 48)
    const int dcpl = variant == 343 ? 3 : 4, dnl = variant <= 341 ? 32 : 16;
    const uint32_t tcols = variant != 340 ? fill_tcols(p.n, (uint32_t)(dnl * dcpl), (uint32_t)dcpl) : 0u;
    return launch_gemv_dyn(p, BITS, stream, dcpl, dnl, true, false, nullptr, 1, nullptr, nullptr, tcols);
  }
  if (variant >= 350 && variant <= 356) {
    if (p.m <= 32u) return kGemvDynNotEligible;
    if (variant >= 355) return launch_gemv_rows(p, BITS, stream, 3, variant == 355 ? 2 : 4, 64);
    return variant == 350 ? launch_gemv_rows(p, BITS, stream) : launch_gemv_rows(p, BITS, stream, variant >= 353 ? 4 : 3, (variant & 1) ? 2 : 4, 32);
  }
  if (variant == 348 && p.m > 32u) return launch_gemv_dyn(p, BITS, stream, c.cpl, 16, true, false, nullptr);  // 33..256 rows: blocks of 32 rows in grid.y of the 32-row build (= production up to 128)
  if (variant == 349 && p.m > 32u) {  // launches of 32 rows (round 3's chunks)
    GemvParams q = p;
    int s_ = XBIT_OK;
    for (uint32_t r0 = 0; r0 < p.m && s_ == XBIT_OK; r0 += 32u) {
      q.m = p.m - r0 < 32u ? p.m - r0 : 32u;
      q.a = p.a + (size_t)r0 * p.k;
      q.out = p.out + (size_t)r0 * p.n;
      s_ = launch_gemv_dyn(q, BITS, stream, c.cpl, 16, true, false, nullptr);
    }
    return s_;
  }
  if (p.m > 32u && p.m <= kGemvRowsMax) return act_dyn_chunks<BITS>(p, stream, c.cpl);
  if (p.m > 32u) return kGemvDynNotEligible;
  const bool auto_small_n = small_n && p.m <= 8u;
  if (p.m > 4u && auto_small_n && !(variant >= 302 && variant <= 304)) return kGemvDynNotEligible;  // 16-column tiles take one row
  int dcpl = auto_small_n ? 2 : c.cpl, dnl = auto_small_n ? 8 : 16;
  bool nt = true, stamped = false;
  if (variant >= 310 && variant <= 314) nt = false;
  const int v = variant >= 310 && variant <= 314 ? variant - 10 : variant;
  if (v == 301) dcpl = 2, dnl = 8;
  if (v >= 302 && v <= 304) dcpl = v - 300, dnl = 16;
  if (variant == 320 || variant == 321) stamped = true;  // 321: no trace buffer -> the v_dot2 (VALU accumulate) build, 48-column tiles
  if (variant == 321) dcpl = 3, dnl = 16;
  return launch_gemv_dyn(p, BITS, stream, dcpl, dnl, nt, stamped, variant == 321 ? nullptr : opts.trace);
}

template <int BITS>
int launch_variant(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) {
  constexpr int UBMAX = max_ub<BITS>();
  const int variant = opts.variant;
  RuleHooks hooks;
  // ---- hooks on the production rule
  if (variant >= 112 && variant <= 114) {
    hooks.cpl = variant - 110;
    return launch_by_rules<BITS>(p, stream, opts, hooks);
  }
  if (variant == 120 || variant == 121) {
    hooks.wave_cap = variant == 120 ? 8u : 16u;
    return launch_by_rules<BITS>(p, stream, opts, hooks);
  }
  if (variant == 122 || variant == 123) {
    hooks.force_nt = variant == 122;
    hooks.force_plain = variant == 123;
    return launch_by_rules<BITS>(p, stream, opts, hooks);
  }
  if (variant == 336) return launch_by_rules<BITS>(p, stream, opts, hooks);
  if (variant == 398 || variant == 399) {
    hooks.no_dyn = true;
    hooks.no_fill_once = true;
    hooks.no_split = variant == 399;
    return launch_by_rules<BITS>(p, stream, opts, hooks);
  }
  if (variant == 202 || variant == 205 || variant == 330 || variant == 332 || variant == 334) {
    hooks.no_fill_once = true;
    hooks.scored_on_dyn = variant >= 330;
    hooks.scored_cpl = (variant == 202 || variant == 332) ? 2 : ((variant == 205 || variant == 334) ? 4 : 0);
    const rules::Call c = make_call<BITS>(p, hooks);
    if (opts.split != 1 && !(c.flags & rules::kGroupC)) {  // the scored plan (or the split the caller forces), whatever the buckets say
      const int st = (c.flags & rules::kGroupA) ? act_scored_split<BITS, UBMAX>(p, stream, opts, c, hooks) : act_scored_split<BITS, 2>(p, stream, opts, c, hooks);
      if (st != kGemvDynNotEligible) return st;
    }
    hooks.no_split = true;
    return launch_by_rules<BITS>(p, stream, opts, hooks);
  }
  // ---- forced launches of the staged-operand kernel; a shape it does not take falls back to the round-2 kernels, unsplit
  if (variant >= 300 && variant <= 356) {
    const int st = forced_dyn<BITS>(p, stream, opts, variant);
    if (st != kGemvDynNotEligible) return st;
    hooks.no_dyn = hooks.no_split = true;
    return launch_by_rules<BITS>(p, stream, opts, hooks);
  }
  // ---- forced launches of the round-2 kernel
  if (p.groupsize % (32u * UBMAX) == 0) {
    if (variant == 180 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 1024>(p, stream);  // 16-row pass, weights first
    if (variant == 181 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 0>(p, stream);     // same launch, production order
    if (variant == 189 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 8>(p, stream);             // stamped (tools/gemv_trace.py --m 16)
    if (variant == 193 && p.m > 1 && p.m <= 4) return launch_rows16_ablate<BITS, UBMAX, 3, 8, 4>(p, stream);  // 4-row pass, stamped
    if (variant == 198 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 16384 | 32768>(p, stream);  // ... and the reduction reads unrolled and predicated
    if (variant == 199 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 32768>(p, stream);          // whole slice, reduction reads unrolled and predicated
    if (variant == 200 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 65536>(p, stream);         // whole slice, reduction reads unconditional + select
    if (variant == 197 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 16384>(p, stream);         // activation slice in two parts (k-blocks 0-1, 2-3), the second behind the first weight rows
    if (variant == 188 && p.m > 8) return launch_rows16_ablate<BITS, UBMAX, 3, 8192>(p, stream);          // activation slice fetched and staged k-block by k-block inside the step loop
    if (variant == 206 && p.m == 1) {  // 32-column tiles read as 8 lanes x 16 B (NL = 8), split as planned
      const SplitPlan plan = plan_split(p, UBMAX, opts.split, BITS);
      GemvWorkspace ws;
      if (plan.splits >= 2 && gemv_workspace(stream, (size_t)((p.n + 31u) / 32u) * 32u, opts, &ws)) return launch_split_rows<BITS, 2, 4, 8, 8>(p, stream, plan.splits, ws);
    }
    if (p.m == 1) {  // measurement variants (M = 1): tools/gemv_sweep.py, tools/README.md
      switch (variant) {
        case 105: return launch_la<BITS, UBMAX, 3, 16, 8, 4, lookahead_for<BITS>()>(p, stream);  // spliced unpack forced
        case 106: return launch_la<BITS, UBMAX, 3, 16, 8, 0, lookahead_for<BITS>()>(p, stream);  // same launch, default unpack
        case 107: return launch_m1<BITS, UBMAX, 4, 32, 16, true>(p, stream);   // 128-column tiles, nt
        case 108: return launch_m1<BITS, UBMAX, 4, 16, 8, true>(p, stream);    // 64-column tiles, nt
        case 109: return launch_m1<BITS, UBMAX, 3, 16, 8, false>(p, stream);   // 48-column tiles
        default: break;
      }
      if constexpr (BITS == 4) {  // launch-shape variants exist for the headline bit-width only
        switch (variant) {
          case 21: return launch_m1<BITS, 4, 2, 16, 8, false>(p, stream);
          case 15: return launch_m1<BITS, 4, 3, 16, 8, false>(p, stream);
          case 22: return launch_m1<BITS, 4, 4, 16, 8, false>(p, stream);
          case 23: return launch_m1<BITS, 4, 2, 16, 16, false>(p, stream);
          case 24: return launch_m1<BITS, 4, 3, 16, 16, false>(p, stream);
          case 25: return launch_m1<BITS, 4, 4, 16, 16, false>(p, stream);
          case 11: return launch_m1<BITS, 4, 2, 16, 8, true>(p, stream);
          case 12: return launch_m1<BITS, 4, 3, 16, 8, true>(p, stream);
          case 13: return launch_m1<BITS, 4, 4, 16, 8, true>(p, stream);
          case 51: return launch_m1<BITS, 4, 4, 32, 16, false>(p, stream);  // 128-column tiles
          case 52: return launch_m1<BITS, 4, 4, 32, 16, true>(p, stream);
          case 53: return launch_m1<BITS, 4, 4, 32, 8, false>(p, stream);
          case 54: return launch_m1<BITS, 4, 4, 32, 8, true>(p, stream);
          case 55: return launch_m1<BITS, 4, 3, 32, 16, true>(p, stream);   // 96-column tiles
          case 56: return launch_m1<BITS, 4, 2, 32, 16, true>(p, stream);   // 64-column tiles, 8-byte loads
          case 28: return launch_m1<BITS, 4, 2, 8, 8, false>(p, stream);
          case 39: return launch_m1<BITS, 2, 2, 8, 8, false>(p, stream);
          case 16: return launch_m1<BITS, 4, 3, 16, 8, false, 0>(p, stream);  // lookahead sweep
          case 17: return launch_m1<BITS, 4, 3, 16, 8, false, 2>(p, stream);
          case 18: return launch_m1<BITS, 4, 3, 16, 8, false, 6>(p, stream);
          case 19: return launch_m1<BITS, 4, 3, 16, 8, false, 8>(p, stream);
          case 63: return launch_la<BITS, 4, 3, 16, 8, 8, kLookahead>(p, stream);  // stamped (tools/gemv_trace.py)
          case 64: return launch_la<BITS, 4, 3, 16, 8, 16, kLookahead>(p, stream);  // VALU accumulate (v_dot2) instead of MFMA
          case 183: return launch_la<BITS, 4, 3, 16, 8, 1024, 4>(p, stream);     // M = 1, first weight rows ahead of the small operands
          case 190: return launch_m1<BITS, 2, 4, 8, 8, false>(p, stream);   // 32-column tiles as 8 lanes x 16 B x 8 k-slots (1 KB per instruction), 2 k-blocks per lane
          case 191: return launch_m1<BITS, 1, 4, 8, 8, false>(p, stream);   // ... 1 k-block per lane
          case 192: return launch_m1<BITS, 2, 4, 8, 16, false>(p, stream);  // ... 16 waves
          case 170: return launch_m1<BITS, 4, 3, 16, 8, false, 138>(p, stream);   // lookahead ramp 3 -> 8
          case 171: return launch_m1<BITS, 4, 3, 16, 8, false, 148>(p, stream);   // 4 -> 8
          case 172: return launch_m1<BITS, 4, 3, 16, 8, false, 126>(p, stream);   // 2 -> 6
          case 173: return launch_m1<BITS, 4, 3, 16, 8, false, 146>(p, stream);   // 4 -> 6
          case 174: return launch_m1<BITS, 4, 3, 16, 8, false, 139>(p, stream);   // 3 -> 9
          case 175: return launch_m1<BITS, 4, 3, 16, 8, false, 128>(p, stream);   // 2 -> 8
          case 176: return launch_m1<BITS, 4, 4, 16, 8, false, 138>(p, stream);   // 64-column tiles, 3 -> 8
          case 177: return launch_m1<BITS, 4, 4, 16, 8, false, 126>(p, stream);   // 64-column tiles, 2 -> 6
          case 164: return launch_m1<BITS, 4, 3, 16, 8, true, 8>(p, stream);        // nt, lookahead 8
          // 128-column tiles (the 8192x28672 path: 32 lanes x 16 B, nontemporal): lookahead and workgroup size
          case 236: return launch_m1<BITS, 4, 4, 32, 16, true, 2>(p, stream);
          case 237: return launch_m1<BITS, 4, 4, 32, 16, true, 6>(p, stream);
          case 238: return launch_m1<BITS, 4, 4, 32, 16, true, 8>(p, stream);
          case 239: return launch_m1<BITS, 4, 4, 32, 8, true, 4>(p, stream);
          case 240: return launch_m1<BITS, 4, 4, 32, 8, true, 6>(p, stream);
          case 241: return launch_m1<BITS, 4, 4, 32, 8, true, 8>(p, stream);
          case 242: return launch_m1<BITS, 4, 4, 16, 8, true, 6>(p, stream);   // 64-column tiles, nontemporal, lookahead 6
          case 243: return launch_m1<BITS, 4, 4, 16, 8, true, 8>(p, stream);
          case 272: return launch_la<BITS, 4, 4, 16, 8, 0, 4>(p, stream);         // 64-column tiles, 8 waves, lookahead 4
          case 274: return launch_la<BITS, 4, 4, 32, 16, 0, 4>(p, stream);        // 128-column tiles, 16 waves, plain loads
          case 231: return launch_la<BITS, 4, 3, 16, 8, 131072, 4>(p, stream);  // wait for a step's rows before issuing the next loads, lookahead 4
          case 232: return launch_la<BITS, 4, 3, 16, 8, 131072, 5>(p, stream);  // ... lookahead 5 (the same 5 rows outstanding at most)
          case 233: return launch_la<BITS, 4, 3, 16, 8, 131072, 6>(p, stream);
          case 234: return launch_la<BITS, 4, 3, 16, 8, 131072, 8>(p, stream);
          case 235: return launch_la<BITS, 4, 3, 16, 8, 0, 5>(p, stream);       // lookahead 5, production order
          case 230: return launch_la<BITS, 4, 3, 16, 8, 65536, 4>(p, stream);   // reduction reads unconditional + select
          case 229: return launch_la<BITS, 4, 3, 16, 8, 32768, 4>(p, stream);   // reduction reads unrolled and predicated
          case 214: return launch_la<BITS, 2, 2, 8, 8, 0, 4>(p, stream);   // 16-column tiles (8 lanes x 8 B x 8 k-slots), whole K
          case 135: return launch_m1<BITS, 4, 3, 16, 8, false, 5>(p, stream);
          case 136: return launch_m1<BITS, 4, 4, 16, 8, false, 6>(p, stream);
          case 137: return launch_m1<BITS, 4, 4, 16, 8, false, 2>(p, stream);
          case 138: return launch_m1<BITS, 2, 3, 16, 16, false, 4>(p, stream);   // 16 waves x 2 k-blocks
          case 139: return launch_m1<BITS, 2, 3, 16, 16, false, 7>(p, stream);   // ... all 8 loads of a wave up front
          case 140: return launch_m1<BITS, 2, 3, 16, 16, false, 2>(p, stream);
          default: break;
        }
      }
    }
  }
  // a variant that does not apply to this call (another bit-width, row count or group size): the round-2 kernels, unsplit, as a reference launch
  hooks.no_dyn = hooks.no_split = true;
  return launch_by_rules<BITS>(p, stream, opts, hooks);
}

}  // namespace tuning
}  // namespace xbit
