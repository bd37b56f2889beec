// xbit_capi.hip — the extern "C" surface declared in include/xbitops_hip.h: argument checks,
// status codes, the common launch function and the per-launch profiling aid.  No kernels here.
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "xbit_host.h"

namespace xbit {

ProfileState& profile_state() {
  static thread_local ProfileState state;
  return state;
}

int launch_raw(const void* kernel, dim3 grid, dim3 block, hipStream_t stream, void** args) {
  ProfileState& ps = profile_state();
  hipError_t err;
  if (ps.enabled) {
    err = hipExtLaunchKernel(kernel, grid, block, args, 0, stream, ps.start, ps.stop, 0);
    ps.pending = (err == hipSuccess);
  } else {
    err = hipLaunchKernel(kernel, grid, block, args, 0, stream);
  }
  if (err != hipSuccess) {
    (void)hipGetLastError();  // clear the sticky error; the caller gets the code
    return (int)err;
  }
  return XBIT_OK;
}

static std::atomic<int> g_gemv_variant{-1};
static std::atomic<int> g_gemv_split{0};
int gemv_split_override() { return g_gemv_split.load(std::memory_order_relaxed); }

namespace {
constexpr size_t kAccSlots = (size_t)1 << 18;  // 2 MiB of 64-bit accumulators per stream
std::mutex g_ws_mutex;
std::map<std::pair<int, hipStream_t>, GemvWorkspace> g_ws;  // never freed: lives as long as the process
}  // namespace

bool gemv_workspace(hipStream_t stream, GemvWorkspace* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  auto it = g_ws.find({dev, stream});
  if (it != g_ws.end()) {
    *out = it->second;
    return out->acc != nullptr;
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return false;  // not remembered: a later call outside the capture allocates
  }
  GemvWorkspace ws;
  bool ok = hipMalloc(reinterpret_cast<void**>(&ws.acc), kAccSlots * sizeof(unsigned long long)) == hipSuccess &&
            hipMemsetAsync(ws.acc, 0, kAccSlots * sizeof(unsigned long long), stream) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    if (ws.acc) (void)hipFree(ws.acc);
    ws = GemvWorkspace{};
  } else {
    ws.slots = kAccSlots;
  }
  g_ws[{dev, stream}] = ws;  // a failed allocation is remembered too: no retry storm
  *out = ws;
  return ok;
}

static std::atomic<uint64_t*> g_gemv_trace{nullptr};
int gemv_variant_override() { return g_gemv_variant.load(std::memory_order_relaxed); }
uint64_t* gemv_trace_buffer() { return g_gemv_trace.load(std::memory_order_relaxed); }

static inline uint32_t packed_rows(uint32_t count, int bits) { return (uint32_t)(((uint64_t)count * (uint64_t)bits + 31u) / 32u); }

static int check_common(const void* qweight, const void* scales, const void* qzeros, int bits, int groupsize, uint32_t k,
                        uint32_t n) {
  if (!qweight || !scales || !qzeros) return XBIT_ERR_NULL_POINTER;
  if (bits < 2 || bits > 8) return XBIT_ERR_BITS;
  if (groupsize < 1) return XBIT_ERR_GROUPSIZE;
  if (k == 0 || n == 0 || (n & 1u)) return XBIT_ERR_SHAPE;
  if ((uint64_t)packed_rows(k, bits) * n >= (1ull << 32) || (uint64_t)k * n >= (1ull << 40)) return XBIT_ERR_SHAPE;
  if (((uintptr_t)qweight & 3u) || ((uintptr_t)qzeros & 3u) || ((uintptr_t)scales & 1u)) return XBIT_ERR_ALIGNMENT;
  return XBIT_OK;
}

}  // namespace xbit

extern "C" {

int xbit_abi_version(void) { return XBIT_ABI_VERSION; }

const char* xbit_status_string(int status) {
  switch (status) {
    case XBIT_OK: return "success";
    case XBIT_ERR_NULL_POINTER: return "xbitops: a required device pointer is NULL";
    case XBIT_ERR_BITS: return "xbitops: bits must be in 2..8";
    case XBIT_ERR_GROUPSIZE: return "xbitops: groupsize must be >= 1";
    case XBIT_ERR_SHAPE: return "xbitops: invalid shape (m, k, n must be > 0, n even, sizes within 32-bit packed-word range)";
    case XBIT_ERR_ALIGNMENT: return "xbitops: a device pointer is not aligned to its element size";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "xbitops: unknown status";
}

int xbit_dequant_f16(void* out, const void* qweight, const void* scales, const void* qzeros, int bits, int groupsize,
                     uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream) {
  if (!out) return XBIT_ERR_NULL_POINTER;
  const int st = xbit::check_common(qweight, scales, qzeros, bits, groupsize, k, n);
  if (st != XBIT_OK) return st;
  if ((uintptr_t)out & 1u) return XBIT_ERR_ALIGNMENT;
  xbit::DequantParams p;
  p.out = static_cast<uint16_t*>(out);
  p.qweight = static_cast<const uint32_t*>(qweight);
  p.scales = static_cast<const uint16_t*>(scales);
  p.qzeros = static_cast<const uint32_t*>(qzeros);
  p.k = k;
  p.n = n;
  p.groupsize = (uint32_t)groupsize;
  p.kw = xbit::packed_rows(k, bits);
  p.groups = (k + (uint32_t)groupsize - 1) / (uint32_t)groupsize;
  p.zw = xbit::packed_rows(n, bits);
  p.bias = add_zero_bias;
  p.kblocks = (k + 31) / 32;
  return xbit::launch_dequant(p, bits, static_cast<hipStream_t>(hip_stream));
}

int xbit_gemv_f16(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, int bits,
                  int groupsize, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream) {
  if (!out || !a) return XBIT_ERR_NULL_POINTER;
  const int st = xbit::check_common(qweight, scales, qzeros, bits, groupsize, k, n);
  if (st != XBIT_OK) return st;
  if (m == 0 || (uint64_t)m * k >= (1ull << 40) || (uint64_t)m * n >= (1ull << 40)) return XBIT_ERR_SHAPE;
  if (((uintptr_t)out & 1u) || ((uintptr_t)a & 1u)) return XBIT_ERR_ALIGNMENT;
  xbit::GemvParams p;
  p.out = static_cast<uint16_t*>(out);
  p.a = static_cast<const uint16_t*>(a);
  p.qweight = static_cast<const uint32_t*>(qweight);
  p.scales = static_cast<const uint16_t*>(scales);
  p.qzeros = static_cast<const uint32_t*>(qzeros);
  p.m = m;
  p.k = k;
  p.n = n;
  p.groupsize = (uint32_t)groupsize;
  p.kw = xbit::packed_rows(k, bits);
  p.groups = (k + (uint32_t)groupsize - 1) / (uint32_t)groupsize;
  p.zw = xbit::packed_rows(n, bits);
  p.bias = add_zero_bias;
  p.tiles = 0;
  p.batches = 0;
  p.trace = xbit::gemv_trace_buffer();
  p.splits = 1;
  p.bps = 0;
  p.acc = nullptr;
  return xbit::launch_gemv(p, bits, static_cast<hipStream_t>(hip_stream));
}

int xbit_gemv_multi_f16(int count, void* const* outs, const void* a, const void* const* qweights, const void* const* scales,
                        const void* const* qzeros, const uint32_t* ns, int bits, int groupsize, uint32_t m, uint32_t k,
                        uint8_t add_zero_bias, void* hip_stream) {
  if (count < 1 || count > xbit::kMaxMulti) return XBIT_ERR_SHAPE;
  if (!outs || !a || !qweights || !scales || !qzeros || !ns) return XBIT_ERR_NULL_POINTER;
  for (int j = 0; j < count; ++j) {
    if (!outs[j]) return XBIT_ERR_NULL_POINTER;
    const int st = xbit::check_common(qweights[j], scales[j], qzeros[j], bits, groupsize, k, ns[j]);
    if (st != XBIT_OK) return st;
    if ((uintptr_t)outs[j] & 1u) return XBIT_ERR_ALIGNMENT;
  }
  if (m == 0 || (uint64_t)m * k >= (1ull << 40)) return XBIT_ERR_SHAPE;
  if ((uintptr_t)a & 1u) return XBIT_ERR_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  // one fused launch when the shape allows it (up to 4 rows, the fast path's alignment and size limits) ...
  bool fused = count >= 2 && m <= 4 && (k % 8u == 0) && (groupsize % 32 == 0) && ((uintptr_t)a % 16 == 0) && xbit::gemv_variant_override() < 0;
  for (int j = 0; j < count && fused; ++j)
    fused = ((uint64_t)xbit::packed_rows(k, bits) * ns[j] * 4u < (1ull << 31)) && ((uintptr_t)qweights[j] % 4 == 0);
  if (fused) {
    xbit::GemvMultiParams p = {};
    p.a = static_cast<const uint16_t*>(a);
    p.m = m;
    p.k = k;
    p.groupsize = (uint32_t)groupsize;
    p.kw = xbit::packed_rows(k, bits);
    p.groups = (k + (uint32_t)groupsize - 1) / (uint32_t)groupsize;
    p.bias = add_zero_bias;
    p.splits = 1;
    p.multi_count = (uint32_t)count;
    for (int j = 0; j < count; ++j) {
      p.multi_n[j] = ns[j];
      p.multi_zw[j] = xbit::packed_rows(ns[j], bits);
      p.multi_qweight[j] = static_cast<const uint32_t*>(qweights[j]);
      p.multi_scales[j] = static_cast<const uint16_t*>(scales[j]);
      p.multi_qzeros[j] = static_cast<const uint32_t*>(qzeros[j]);
      p.multi_out[j] = static_cast<uint16_t*>(outs[j]);
    }
    p.out = p.multi_out[0];
    p.qweight = p.multi_qweight[0];
    p.scales = p.multi_scales[0];
    p.qzeros = p.multi_qzeros[0];
    p.n = ns[0];
    p.zw = xbit::packed_rows(ns[0], bits);
    const int st = xbit::launch_gemv_multi(p, bits, stream);
    if (st != XBIT_ERR_SHAPE) return st;
  }
  // ... otherwise one launch per weight set
  for (int j = 0; j < count; ++j) {
    const int st = xbit_gemv_f16(outs[j], a, qweights[j], scales[j], qzeros[j], bits, groupsize, m, k, ns[j], add_zero_bias, hip_stream);
    if (st != XBIT_OK) return st;
  }
  return XBIT_OK;
}

int xbit_profile_enable(int on) {
  xbit::ProfileState& ps = xbit::profile_state();
  if (on && !ps.start) {
    hipError_t err = hipEventCreate(&ps.start);
    if (err == hipSuccess) err = hipEventCreate(&ps.stop);
    if (err != hipSuccess) {
      (void)hipGetLastError();
      return (int)err;
    }
  }
  ps.enabled = on != 0;
  ps.pending = false;
  return XBIT_OK;
}

int xbit_profile_last_ms(float* ms) {
  if (!ms) return XBIT_ERR_NULL_POINTER;
  xbit::ProfileState& ps = xbit::profile_state();
  if (!ps.pending) return XBIT_ERR_SHAPE;
  hipError_t err = hipEventSynchronize(ps.stop);
  if (err == hipSuccess) err = hipEventElapsedTime(ms, ps.start, ps.stop);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return (int)err;
  }
  return XBIT_OK;
}

/* Tuning hook, deliberately NOT in the public header: selects a GEMV launch-shape variant for
 * sweeps (tools/gemv_sweep.py); -1 restores automatic selection. */
void xbit_debug_set_gemv_variant(int v) { xbit::g_gemv_variant.store(v, std::memory_order_relaxed); }
/* Forces the split-K factor of eligible launches (<= 0: automatic; 1: never split). */
void xbit_debug_set_gemv_split(int splits) { xbit::g_gemv_split.store(splits, std::memory_order_relaxed); }
/* Device buffer (8 x uint64 per wave) that the stamped measurement build (variants 41+) fills. */
void xbit_debug_set_gemv_trace(void* device_buffer) { xbit::g_gemv_trace.store(static_cast<uint64_t*>(device_buffer), std::memory_order_relaxed); }

}  // extern "C"
