// xbit_host.h — host-side glue shared by the launchers: parameter blocks, status codes, and the
// one function every launch goes through (so the optional per-launch event bracketing of
// xbit_profile_enable applies uniformly).
#pragma once

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Fused launches (MULTI builds of the kernel): several weight sets with the same K / bits / group size
// and ONE activation (q/k/v, gate/up) share a grid; tile t belongs to set j with
// multi_tile_end[j-1] <= t < multi_tile_end[j].  A separate kernel-argument type: the plain kernels keep
// their argument block (and their code) unchanged.
struct GemvMultiParams : GemvParams {
  uint32_t multi_count;
  uint32_t multi_tile_end[kMaxMulti], multi_n[kMaxMulti], multi_zw[kMaxMulti];
  const uint32_t* multi_qweight[kMaxMulti];
  const uint16_t* multi_scales[kMaxMulti];
  const uint32_t* multi_qzeros[kMaxMulti];
  uint16_t* multi_out[kMaxMulti];
};

// Per-(device, stream) scratch for split-K launches: launches on one stream run in order, so they
// can share it; different streams get their own.  Allocated on first use and kept.
struct GemvWorkspace {
  unsigned long long* acc = nullptr;
  size_t slots = 0;
};
// false: no workspace for this stream (allocation failed, or the first use would fall inside a
// stream capture, where allocation is not allowed) - the caller then launches without split-K.
bool gemv_workspace(hipStream_t stream, GemvWorkspace* ws);

// thread-local profiling state (xbit_profile_enable / xbit_profile_last_ms)
struct ProfileState {
  bool enabled = false;
  bool pending = false;
  hipEvent_t start = nullptr, stop = nullptr;
};
ProfileState& profile_state();

// Every launch goes through here; returns 0 or a positive hipError_t (never aborts).
int launch_raw(const void* kernel, dim3 grid, dim3 block, hipStream_t stream, void** args);

// Launch `kernel<<<grid, block, 0, stream>>>(args...)`.
template <typename... KArgs, typename... Args>
int launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t stream, Args... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
  void* ptrs[] = {static_cast<void*>(&args)...};
  return launch_raw(reinterpret_cast<const void*>(kernel), grid, block, stream, ptrs);
}

int launch_dequant(const DequantParams& p, int bits, hipStream_t stream);
int launch_gemv(const GemvParams& p, int bits, hipStream_t stream);                        // dispatcher (xbit_gemv.hip)
int launch_gemv_mfma(const GemvParams& p, int bits, hipStream_t stream, int variant);     // xbit_gemv_mfma.hip
// fused launch over p.multi_* (M = 1); XBIT_ERR_SHAPE when the shape is not eligible (caller then launches one by one)
int launch_gemv_multi(const GemvMultiParams& p, int bits, hipStream_t stream);

// tuning hooks (tools/ only): variant < 0 = automatic; trace buffer for the stamped build
int gemv_variant_override();
int gemv_split_override();  // <= 0: automatic
uint64_t* gemv_trace_buffer();

}  // namespace xbit
