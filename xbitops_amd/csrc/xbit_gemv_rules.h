// xbit_gemv_rules.h — WHICH launch serves a GEMV call whose operands meet the fast kernels' alignment rules (k % 8 == 0, group
// size a multiple of 32, 16-byte aligned activations; everything else: xbit_gemv.hip).  Host logic only, no kernel here.
//
// The rule is DATA: kRules below is scanned top to bottom; the first row whose buckets hold the call and whose launch is
// eligible serves it (a launcher may still decline - staged operands beyond LDS, a K range beyond a build's staging rounds -
// and the scan goes on).  Every row names the measurement that put it there (files under profiles/).  What a row launches is
// an Act; how an Act is launched is in xbit_gemv_mfma.h (launch_by_rules).  tools/launch_rules_grid.py prints the launch of
// 13.5 K shapes through the library's dry run (tests/golden/launch_rules.txt pins it: a rule edit shows up as a diff), and
// tools/check_rules.py re-measures on the GPU that the row which fires is within 8 % of the best forced launch.
#pragma once

#include <stdint.h>

namespace xbit {
namespace rules {

// ---- the buckets a call falls into ---------------------------------------------------------------------------------------
struct Call {
  int bits;
  uint32_t m, k, n, groupsize;
  // derived
  uint32_t kblocks;     // ceil(k / 32)
  uint32_t t32;         // ceil(n / 32): 32-column tiles
  uint32_t tiles;       // column tiles at the default width (16 * cpl columns, cpl = pick_cpl(n))
  int cpl;              // default tile width / 16
  int ubmax;            // k-blocks per lane of the round-2 kernel (4 up to 4 bits, else 2)
  uint32_t flags;       // kSmallN | kWide | kBigWeights | group class
};

enum : uint32_t {
  kSmallN = 1u << 0,      // fewer than 160 tiles of 32 columns (N < 5120) with K >= 4096: 16-column tiles (8 lanes side by side) put twice the workgroups on the chip
  kWide = 1u << 1,        // 4- or 8-bit, N >= 21504 with 128-column tiles filling at least two thirds of their rounds of 256 CUs (168+ tiles), K >= 32 sweeps:
                          // 128-column tiles (32 lanes x 16 B: 512 contiguous bytes per packed row and instruction)
  kBigWeights = 1u << 2,  // >= 96 MiB of packed weights and N >= 16384: nontemporal loads on the round-2 kernel
  kGroupA = 1u << 3,      // group size a multiple of 32 * ubmax (128 up to 4 bits, else 64): ubmax k-blocks per lane share one scale
  kGroupB = 1u << 4,      // up to 4 bits, group size a multiple of 64 but not of 128: 2 k-blocks per lane
  kGroupC = 1u << 5,      // any other multiple of 32: 1 k-block per lane
  kNoFillOnce = 1u << 6,  // state: an earlier row ruled the fill-once split out for this call
  kNoScored = 1u << 7,    // state: ... and the scored split
};

// ---- what a row launches -------------------------------------------------------------------------------------------------
enum class Act : uint8_t {
  MarkUnsplit,        // not a launch: this call keeps K whole (sets kNoFillOnce, and kNoScored at one row); the scan goes on
  MarkNoFillOnce,     // not a launch: sets kNoFillOnce only
  FillOnceSplit,      // staged-operand kernel, split-K: 64-column tiles x floor(256 / tiles) K ranges (32-column tiles where that needs more
                      // than 15 ranges; two rounds of workgroups where one would leave a fifth of the chip idle); one row with K ranges
                      // beyond the build's staging rounds: the same plan on the round-2 split kernel
  ScoredSplit,        // plan_split(): tile width and split count by score; one row with 16 or >= 24 units per range: staged-operand split
                      // build, else the round-2 split kernel (up to 4 rows)
  DynChunks32,        // 33..512 rows: the shared-activation-tile kernel or the staged-operand kernel's 32-row build (3 bits and more), one launch over blocks of rows
  DynWhole,           // staged-operand kernel, whole-K tiles of the default width
  DynNarrow,          // staged-operand kernel, whole-K tiles of 16 columns (8 lanes side by side)
  Mfma16Col,          // round-2 kernel, 16-column tiles
  Mfma128Col,         // round-2 kernel, 128-column tiles, nontemporal loads
  MfmaDefault,        // round-2 kernel, tiles of the default width (1..16 rows per pass; nontemporal loads with kBigWeights)
};

constexpr uint32_t kAny = 0xFFFFFFFFu;
constexpr uint16_t bit_set(int lo, int hi) { return (uint16_t)(((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u)); }
constexpr uint16_t kAllBits = bit_set(2, 8), kBits48 = (uint16_t)((1u << 4) | (1u << 8));

struct Rule {
  const char* name;
  uint16_t bits;                  // bit b set: the row applies to b-bit weights
  uint32_t m_lo, m_hi;            // activation rows
  uint32_t kb_lo, kb_hi;          // k-blocks (K / 32)
  uint32_t n_lo, n_hi;            // columns
  uint32_t t_lo, t_hi;            // tiles of the default width
  uint32_t t32_lo, t32_hi;        // tiles of 32 columns
  uint32_t need, never;           // flags that must / must not be set
  Act act;
  const char* evidence;           // the measurement behind the row (profiles/...)
};

// The rows.  Order matters: first eligible match wins.
constexpr Rule kRules[] = {
    // ---- few-column, long-K layers (down-projections): split K over workgroups --------------------------------------------------
    {"2/3-bit, K <= 8192, >= 3072 columns, one row: unsplit", bit_set(2, 3), 1, 1, 0, 256, 3072, kAny, 0, kAny, 0, kAny, kGroupA, 0, Act::MarkUnsplit,
     "profiles/r03/sweep_split_dyn.txt (8192x3584 w2 5.5 us unsplit on 16-column tiles against 6.1 split; 8192x4096 w3 6.0 / 6.5)"},
    {"... group size 64: no fill-once split", bit_set(2, 3), 1, 1, 0, 256, 3072, kAny, 0, kAny, 0, kAny, kGroupB, 0, Act::MarkNoFillOnce,
     "profiles/r03/sweep_split_dyn.txt (as above; the scored plan keeps its say at group size 64)"},
    {"2-bit, K <= 8192, >= 3072 columns, 2-4 rows: no fill-once split", bit_set(2, 2), 2, 4, 0, 256, 3072, kAny, 0, kAny, 0, kAny, 0, kGroupC, Act::MarkNoFillOnce,
     "profiles/r03/sweep_rows_split_k.txt (8192x3584 w2 at 2 rows: 6.1 us against 6.8)"},
    {"fill the chip once: K >= 8192, <= 176 tiles of 32 columns, 1-32 rows", kAllBits, 1, 32, 256, kAny, 0, kAny, 0, kAny, 0, 176, 0, kGroupC | kNoFillOnce, Act::FillOnceSplit,
     "profiles/r03/sweep_split_unseen_shapes_after.txt, sweep_split_known_shapes_after.txt, sweep_split_other_bits_after.txt, sweep_rows_split_k_after.txt"},
    {"fill the chip once: K >= 12288, <= 256 tiles of 32 columns, 1-32 rows", kAllBits, 1, 32, 384, kAny, 0, kAny, 0, kAny, 177, 256, 0, kGroupC | kNoFillOnce, Act::FillOnceSplit,
     "profiles/r03/sweep_split_wide_shapes_after.txt (12288x8192 12.6 -> 11.9 us, 28672x6144 24.8 -> 20.2)"},
    {"scored split plan, 1-4 rows (what the fill-once plan does not take)", kAllBits, 1, 4, 256, kAny, 0, kAny, 0, kAny, 0, 176, 0, kGroupC | kNoScored, Act::ScoredSplit,
     "profiles/r02/sweep_split_plans_fine.txt, sweep_split_plans_other_bits.txt; profiles/r03/sweep_split_dyn.txt (which kernel runs the plan)"},
    // ---- more than 32 rows ---------------------------------------------------------------------------------------------------------
    {"2-bit, 33-64 rows, < 320 tiles of 32 columns: 16-row passes", bit_set(2, 2), 33, 64, 0, kAny, 0, kAny, 0, kAny, 0, 319, 0, 0, Act::MfmaDefault,
     "profiles/r04/sweep_rows_w2.txt (2-bit units are 256 k: two row blocks of a narrow layer do not fill the chip - 4096x4096 at 64 rows 11.0 us against 15.0, 8192x8192 35.2 / 42.4; 4096x11008 wins: 16.4 against 25.6)"},
    {"33-512 rows, >= 192 tiles of 32 columns: one launch over blocks of 32 / 64 rows", kAllBits, 33, 512, 0, kAny, 0, kAny, 0, kAny, 192, kAny, 0, 0, Act::DynChunks32,
     "profiles/r04/sweep_rows_shared_tile_v1.txt, sweep_rows_group64_32.txt (group 64 / 32: 4096x11008 w4 at 128 rows 40.0 us against 45.7 in 16-row passes), sweep_rows_192_256.txt (256 rows: 54.7 us against 88.5; dequantise + dense GEMM 72.9: linear_crossover_2.txt), sweep_rows_shared_tile_v2.txt (64 rows per workgroup: 38.1), sweep_rows_in_grid_y.txt"},
    {"129-512 rows, fewer tiles", kAllBits, 129, 512, 0, kAny, 0, kAny, 0, kAny, 0, 191, 0, 0, Act::DynChunks32,
     "profiles/r04/sweep_rows_192_256.txt (4096x4096 w4 at 192 / 256 rows: 17.9 / 18.7 us against 30.0 / 39.8 in 16-row passes)"},
    {"33-128 rows, 96-191 tiles of 32 columns, groups of 128 and more", kAllBits, 33, 128, 0, kAny, 0, kAny, 0, kAny, 96, 191, kGroupA, 0, Act::DynChunks32,
     "profiles/r04/sweep_rows_shared_tile_v1.txt (4096x4096 w4 at 64 / 128 rows: 9.4 / 17.4 us against 12.1 / 21.4; 11008x4096: 21.9 / 40.9 against 27.8 / 51.3), sweep_rows_group64_32.txt (group 32 loses there: 21.6 against 18.4), check_rules_3.txt (6144x2048 at 64 rows loses: 12.2 against 10.9), "
     "sweep_rows_w2_few_tiles_long_k.txt (2-bit at K > 4096 had a row of its own here - 16-row passes, the tables did not fit in LDS - until the tables became a window: 6144x4096 w2 at 96 / 128 rows 24.8 / 32.2 us -> 20.5 / 20.8, 5120x5120 30.8 / 38.3 -> 18.1 / 18.6, 8192x3584 25.2 / 32.3 -> 25.8 / 25.8)"},
    // ---- the staged-operand kernel, whole K ----------------------------------------------------------------------------------------
    {"17-32 rows, >= 192 tiles: 32-row build", kAllBits, 17, 32, 0, kAny, 0, kAny, 192, kAny, 0, kAny, kGroupA, 0, Act::DynWhole,
     "profiles/r03/sweep_rows_17_32.txt (4096x11008 w4 15.2 -> 11.5 us; few tiles lose: 4096x4096 7.4 -> 8.6)"},
    {"9-16 rows, >= 224 tiles: 16-row build", kAllBits, 9, 16, 0, kAny, 0, kAny, 224, kAny, 0, kAny, kGroupA, 0, Act::DynWhole,
     "profiles/r03/sweep_rows_8_16.txt, sweep_rows_unseen_shapes.txt (192 tiles with a long K lose 4-7 %: hence 224)"},
    {"9-16 rows, K <= 8192: 16-row build", kAllBits, 9, 16, 0, 256, 0, kAny, 0, kAny, 0, kAny, kGroupA, 0, Act::DynWhole,
     "profiles/r03/sweep_rows_8_16_small_n.txt (4096x4096 w4 7.3 -> 6.5 us on 32-column tiles), sweep_rows_5_16_rule_check_before_long_k_exclusion.txt (13824x5120 loses)"},
    {"5-8 rows, >= 224 tiles: 8-row build", kAllBits, 5, 8, 0, kAny, 0, kAny, 224, kAny, 0, kAny, kGroupA, kSmallN | kWide, Act::DynWhole,
     "profiles/r03/sweep_rows_8_16.txt (4096x11008 w4 8.3 -> 7.6 us); 128-column tiles keep up to 8 rows (8192x28672: 25.4 against 27.8)"},
    {"5-8 rows, K <= 8192: 8-row build", kAllBits, 5, 8, 0, 256, 0, kAny, 0, kAny, 0, kAny, kGroupA, kSmallN | kWide, Act::DynWhole,
     "profiles/r03/sweep_rows_8_16_more.txt"},
    {"2-4 rows, 4- and 8-bit, >= 224 tiles: 4-row build", kBits48, 2, 4, 0, kAny, 0, kAny, 224, kAny, 0, kAny, kGroupA, kSmallN | kWide, Act::DynWhole,
     "profiles/r03/sweep_rows_2_to_4.txt (4096x11008 M=4: w4 7.75 -> 7.52 us, w8 11.7 -> 10.9; 3-bit loses), sweep_rows_unseen_shapes.txt (224)"},
    {"2-4 rows, 4- and 8-bit, 192-223 tiles, K <= 4096: 4-row build", kBits48, 2, 4, 0, 128, 0, kAny, 192, 223, 0, kAny, kGroupA, kSmallN | kWide, Act::DynWhole,
     "profiles/r04/check_rules_1.txt (3072x9216 w4 at 2 rows: 6.20 us against 6.99; 192 tiles lose only with a long K, profiles/r03/sweep_rows_unseen_shapes.txt)"},
    {"one row, few columns, from 3 bits up: 16-column tiles", bit_set(3, 8), 1, 1, 0, kAny, 0, kAny, 0, kAny, 0, kAny, kGroupA | kSmallN, 0, Act::DynNarrow,
     "profiles/r03/sweep_dyn_vs_mfma_period.txt (8192x3584 w3/w4 6.5 -> 5.9 / 6.1 us)"},
    {"one row, few columns, 2-bit, K >= 8192: 16-column tiles", bit_set(2, 2), 1, 1, 256, kAny, 0, kAny, 0, kAny, 0, kAny, kGroupA | kSmallN, 0, Act::DynNarrow,
     "profiles/r03/sweep_dyn_vs_mfma_period.txt (8192x3584 w2 6.0 -> 5.5 us; 4096x4096 w2 stays: 4.1 against 4.5)"},
    {"one row, >= 192 tiles", kAllBits, 1, 1, 0, kAny, 0, kAny, 192, kAny, 0, kAny, kGroupA, kSmallN | kWide, Act::DynWhole,
     "profiles/r03/sweep_dyn_vs_mfma_period.txt, sweep_tiles_unseen_shapes.txt (160-tile layers stay: 5120x5120 w4 6.4 against 7.4 us)"},
    {"one row, 5/6/7-bit, any tile count", bit_set(5, 7), 1, 1, 0, kAny, 0, kAny, 0, kAny, 0, kAny, kGroupA, kSmallN | kWide, Act::DynWhole,
     "profiles/r03/sweep_dyn_vs_mfma_period.txt (4096x14336 w5 17.8 -> 9.0 us)"},
    // group size 64 up to 4 bits / group size 32 (or 32 x odd): the staged-operand kernel where it was measured to win
    {"group 64, 4-bit, one row, >= 192 tiles", bit_set(4, 4), 1, 1, 0, kAny, 0, kAny, 192, kAny, 0, kAny, kGroupB, 0, Act::DynWhole,
     "profiles/r03/sweep_group_sizes_after.txt (4096x11008 7.5 -> 7.3 us; 3-bit and few tiles lose)"},
    {"group 64 / 32, 4- and 8-bit, 2-4 rows", kBits48, 2, 4, 0, kAny, 0, kAny, 0, kAny, 0, kAny, 0, kGroupA, Act::DynWhole,
     "profiles/r03/sweep_group_sizes_after.txt"},
    {"group 32, 5 bits and more, one row", bit_set(5, 8), 1, 1, 0, kAny, 0, kAny, 0, kAny, 0, kAny, kGroupC, 0, Act::DynWhole,
     "profiles/r03/sweep_group_sizes_after.txt (3- and 4-bit lose 4-14 % at group 32: 4096x11008 w3 7.6 against 8.7 us)"},
    // ---- the round-2 kernel ------------------------------------------------------------------------------------------------------------
    {"few columns, up to 8 rows: 16-column tiles", kAllBits, 1, 8, 0, kAny, 0, kAny, 0, kAny, 0, kAny, kGroupA | kSmallN, 0, Act::Mfma16Col,
     "profiles/r01/sweep_k_heavy_shapes.txt (4096x4096 5.4 -> 4.5 us, 11008x4096 11.2 -> 8.5)"},
    {"very wide layers, 4- and 8-bit, up to 8 rows: 128-column tiles", kBits48, 1, 8, 0, kAny, 0, kAny, 0, kAny, 0, kAny, kGroupA | kWide, 0, Act::Mfma128Col,
     "profiles/r01/sweep_wide_tiles.txt (8192x28672 23.4 -> 21.5 us); profiles/r04/sweep_wide_2.txt, sweep_wide_3.txt (staged-operand builds of the same tiles, narrowed tiles, split K: none faster), "
     "sweep_wide_4.txt + check_rules_1.txt (N = 22016, 172 tiles: 8192x22016 w4 22.8 -> 19.2 us, 4096x22016 w8 22.7 -> 17.3; N = 16384, 128 tiles: loses)"},
    {"everything else", kAllBits, 1, kAny, 0, kAny, 0, kAny, 0, kAny, 0, kAny, 0, 0, Act::MfmaDefault,
     "profiles/r01/sweep_all_configs.txt, sweep_nt_large_shapes.txt; profiles/r02/sweep_16_vs_8_waves.txt"},
};
constexpr int kRuleCount = (int)(sizeof(kRules) / sizeof(kRules[0]));

inline bool matches(const Rule& r, const Call& c, uint32_t state) {
  const uint32_t f = c.flags | state;
  return ((r.bits >> c.bits) & 1u) && c.m >= r.m_lo && c.m <= r.m_hi && c.kblocks >= r.kb_lo && c.kblocks <= r.kb_hi && c.n >= r.n_lo && c.n <= r.n_hi &&
         c.tiles >= r.t_lo && c.tiles <= r.t_hi && c.t32 >= r.t32_lo && c.t32 <= r.t32_hi && (f & r.need) == r.need && (f & r.never) == 0u;
}

}  // namespace rules
}  // namespace xbit
