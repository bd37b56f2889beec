// xbit_host.h — host-side glue shared by the launchers: parameter blocks, status codes, and the
// one function every launch goes through (so the optional per-launch event bracketing of
// xbit_profile_enable applies uniformly).
#pragma once

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/xbitops_hip.h"

namespace xbit {

struct DequantParams {
  uint16_t* out;            // fp16 [k, n]
  const uint32_t* qweight;  // [kw, n]
  const uint16_t* scales;   // fp16 [groups, n]
  const uint32_t* qzeros;   // [groups, zw]
  uint32_t k, n, groupsize, kw, groups, zw, bias, kblocks;
};

constexpr int kMaxMulti = 4;  // weight sets one fused launch can take (xbit_gemv_multi_f16)

struct GemvParams {
  uint16_t* out;            // fp16 [m, n]
  const uint16_t* a;        // fp16 [m, k]
  const uint32_t* qweight;  // [kw, n]
  const uint16_t* scales;   // fp16 [groups, n]
  const uint32_t* qzeros;   // [groups, zw]
  uint32_t m, k, n, groupsize, kw, groups, zw, bias;
  uint32_t tiles;    // column tiles (grid.x)
  uint32_t batches;  // k-batches of UB k-blocks
  uint64_t* trace;   // measurement builds only: per-wave s_memrealtime stamps (else nullptr)
  // split-K launches only (few column tiles, long K): K is cut into `splits` ranges of `bps`
  // batches, one workgroup each.  Every output element has a 64-bit accumulator in `acc`; each
  // workgroup adds its partial sum there as a fixed-point integer with ONE returning atomic that
  // also counts arrivals, and the thread that sees the last arrival writes the fp16 result.
  uint32_t splits, bps;
  unsigned long long* acc;  // [grid.y][tiles][MT * CN], zero between launches (the last arriver resets its slot)
};

// Fused launches of the staged-operand kernel (xbit_gemv_dyn.h, MULTI builds): the weight sets of one launch.  Passed to the
// kernel by value; tile t of the grid belongs to set j with tile_end[j-1] <= t < tile_end[j] (filled in by the launcher,
// which knows the tile width; unused entries: tile_end = 0xFFFFFFFF).
struct GemvSets {
  uint32_t count;
  uint32_t tile_end[kMaxMulti], n[kMaxMulti], zw[kMaxMulti];
  const uint32_t* qweight[kMaxMulti];
  const uint16_t* scales[kMaxMulti];
  const uint32_t* qzeros[kMaxMulti];
  uint16_t* out[kMaxMulti];
};

// Act-order launches (xbit_gidx.hip): the group of row k is g_idx[k].
struct GemvGidxParams {
  uint16_t* out;            // fp16 [m, n] (gemv) or [k, n] (dequant)
  const uint16_t* a;        // fp16 [m, k] (gemv only)
  const uint32_t* qweight;  // [kw, n]
  const uint16_t* scales;   // fp16 [groups, n]
  const uint32_t* qzeros;   // [groups, zw]
  const int32_t* g_idx;     // [k], values in [0, groups)
  uint32_t m, k, n, kw, groups, zw, bias;
};

// Split-K scratch (64-bit accumulators, zero between launches; see include/xbitops_hip.h "State").
struct GemvWorkspace {
  unsigned long long* acc = nullptr;
  size_t slots = 0;
};

// Host-side choices of one GEMV call (never passed to a kernel).
struct GemvLaunchOpts {
  int split = 0;                        // 0 automatic, 1 never, 2..15 forced
  int variant = -1;                     // measurement variant (XBIT_TUNING builds), <= 0: production launch
  unsigned long long* user_ws = nullptr;  // caller-owned scratch, or nullptr
  size_t user_ws_slots = 0;
  uint64_t* trace = nullptr;            // stamped measurement build only
};

// Scratch for a split-K launch of `slots` accumulators on `stream`: the caller's if `opts` has one
// that is large enough, else the library's (per-stream buffer for eager launches, a private pool
// range inside a stream capture).  false = none available: launch unsplit.
bool gemv_workspace(hipStream_t stream, size_t slots, const GemvLaunchOpts& opts, GemvWorkspace* ws);
constexpr size_t kMaxSplitSlots = (size_t)1 << 18;  // largest launch that may be split (2 MiB of accumulators)
// A failed launch may have left library scratch dirty: re-zero it before the next split-K launch.
void mark_workspaces_dirty();

// thread-local profiling state (xbit_profile_enable / xbit_profile_last_ms)
struct ProfileState {
  bool enabled = false;
  bool pending = false;
  hipEvent_t start = nullptr, stop = nullptr;
  // series mode (xbit_profile_series_begin / _end): one event pair per launch, no host wait in between
  bool series = false;
  int series_used = 0, series_capacity = 0;
  std::vector<hipEvent_t> series_events;  // start, stop, start, stop, ...
  // dry run (xbit_dry_run_begin / _end): launches are recorded, not issued
  bool dry_run = false;
  std::vector<xbit_launch_record> records;
};
ProfileState& profile_state();

// Every launch goes through here; returns 0 or a positive hipError_t (never aborts).
int launch_raw(const void* kernel, dim3 grid, dim3 block, hipStream_t stream, void** args);
// ... with `dynamic_lds` bytes of dynamic LDS (the 64 KB default limit is raised for the kernel when needed)
int launch_raw_dynamic(const void* kernel, dim3 grid, dim3 block, size_t dynamic_lds, hipStream_t stream, void** args);

// Launch `kernel<<<grid, block, 0, stream>>>(args...)`.
template <typename... KArgs, typename... Args>
int launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t stream, Args... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
  void* ptrs[] = {static_cast<void*>(&args)...};
  return launch_raw(reinterpret_cast<const void*>(kernel), grid, block, stream, ptrs);
}

// Tiles are 16*CPL columns wide.  Prefer the width whose tile count wastes the fewest CU slots
// (tiles / (ceil(tiles/256)*256)); ties go to the wider load (longer contiguous row segments).
inline int pick_cpl(uint32_t n) {
  int best = 2;
  double best_eff = -1.0;
  for (int cpl = 2; cpl <= 4; ++cpl) {
    const uint32_t tiles = (n + 16u * cpl - 1) / (16u * cpl);
    const double eff = (double)n / (16.0 * cpl) / (double)(((tiles + 255u) / 256u) * 256u);
    if (eff >= best_eff - 1e-9) {
      best_eff = eff;
      best = cpl;
    }
  }
  return best;
}

// Columns a tile should own so that the tile count fills whole rounds of the chip's 256 CUs: the lane map's `cn` columns
// (cn = lanes x cpl) unless a narrower tile - a multiple of cpl - brings the count up to the next multiple of 256
// (28672 columns, cn 128: 224 tiles of 128 -> 256 tiles of 112; cn 64: 448 -> 512 tiles of 56).
inline uint32_t fill_tcols(uint32_t n, uint32_t cn, uint32_t cpl) {
  const uint32_t tiles = (n + cn - 1u) / cn, target = ((tiles + 255u) / 256u) * 256u;
  const uint32_t per = (n + target - 1u) / target, t = ((per + cpl - 1u) / cpl) * cpl;
  return t < cn ? t : cn;
}

int launch_dequant(const DequantParams& p, int bits, hipStream_t stream);
// act-order dequant on the tile kernel (n % 8 == 0, 16-byte aligned qweight / scales / out, kblocks within the grid limit)
int launch_dequant_by_id(const DequantParams& p, const int32_t* g_idx, int bits, hipStream_t stream);
int launch_gemv(const GemvParams& p, int bits, hipStream_t stream, const GemvLaunchOpts& opts);          // dispatcher (xbit_gemv.hip)
int launch_gemv_mfma(const GemvParams& p, int bits, hipStream_t stream, const GemvLaunchOpts& opts);     // xbit_gemv_mfma.hip
// One activation row through the staged-operand kernel (xbit_gemv_dyn.h): tiles of 16 * cpl columns (nl = 16) or 16 columns
// (nl = 8, cpl = 2).  Returns kGemvDynNotEligible when the shape is not for that kernel (the caller launches another).
constexpr int kGemvDynNotEligible = 1 << 30;
// splits >= 2 with acc (zeroed 64-bit accumulators, one per output element of whole tiles): K cut into that many ranges
int launch_gemv_dyn(const GemvParams& p, int bits, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits = 1,
                    unsigned long long* acc = nullptr, const int32_t* g_idx = nullptr,  // xbit_gemv_dyn.hip; g_idx: act-order (group of row k), 48-column tiles, whole K
                    uint32_t tcols = 0);  // columns a tile owns (0: the whole lane map, nl * cpl); nl = 32: 128-column tiles (4- and 8-bit, up to four rows)
// Fused launch through the staged-operand kernel: `p` describes the shared operands (a, m, k, group size, bias; n = the widest set),
// `sets` the weight sets (count, n, zw, pointers; tile_end is filled in here).  1..16 rows, whole K, tiles of 16 * cpl columns
// (cpl 3 or 4).  kGemvDynNotEligible when the shape is not for that kernel.
int launch_gemv_dyn_multi(const GemvParams& p, const GemvSets& sets, int bits, hipStream_t stream, int cpl);  // xbit_gemv_dyn.hip
constexpr uint32_t kGemvRowsMax = 512;  // rows the one-launch kernels for more than 32 rows take (beyond: 16-row passes; QuantLinear goes dense there)
// 33..kGemvRowsMax rows, contiguous groups of a power-of-two multiple of 32 (64 at 2 bits): column groups that share one LDS image of the activations
// (xbit_gemv_rows.h).  cpl / ncg / mt (rows per workgroup) = 0: chosen by a cost model; kGemvDynNotEligible when the shape is not for that kernel.
int launch_gemv_rows(const GemvParams& p, int bits, hipStream_t stream, int cpl = 0, int ncg = 0, int mt = 0);  // xbit_gemv_rows.hip
// 33..kGemvRowsMax rows in one launch: that kernel or the staged-operand kernel's 32-row build with the blocks of 32 rows in grid.y, by tile and row count
int launch_gemv_rows_or_dyn(const GemvParams& p, int bits, hipStream_t stream, int dyn_cpl);  // xbit_gemv_rows.hip
int launch_gemv_gidx(const GemvGidxParams& p, int bits, hipStream_t stream);     // xbit_gidx.hip
int launch_dequant_gidx(const GemvGidxParams& p, int bits, hipStream_t stream);

// XBIT_TUNING builds (libxbitops_hip_tuning.so) additionally hold the measurement variants: launch
// shapes, ablations, the stamped kernel and the first-generation VALU-only kernel.  The production
// library contains none of them.
#ifndef XBIT_TUNING
#define XBIT_TUNING 0
#endif

}  // namespace xbit
