// xbit_capi.hip — the extern "C" surface declared in include/xbitops_hip.h: argument checks,
// status codes, the common launch function and the per-launch profiling aid.  No kernels here.
#include <dlfcn.h>
#include <stddef.h>
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
  return launch_raw_dynamic(kernel, grid, block, 0, stream, args);
}

namespace {
// Kernels that may ask for more than the 64 KB of LDS a launch gets by default are raised ONCE per (kernel, device) to
// the chip's whole 160 KB, never per call to the call's own size: two host threads launching the same kernel with
// different sizes would otherwise lower each other's limit between the attribute call and the launch.
constexpr int kMaxLds = 160 << 10;
std::mutex g_lds_mutex;
std::map<std::pair<const void*, int>, bool> g_lds_raised;
int raise_lds_limit(const void* kernel) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return (int)hipErrorInvalidDevice;
  }
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  bool& done = g_lds_raised[std::make_pair(kernel, dev)];
  if (done) return XBIT_OK;
  hipFuncAttributes fa;
  hipError_t err = hipFuncGetAttributes(&fa, kernel);  // the limit covers dynamic LDS only: leave room for the kernel's static part
  if (err == hipSuccess) err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - (int)fa.sharedSizeBytes);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return (int)err;
  }
  done = true;
  return XBIT_OK;
}
}  // namespace

int launch_raw_dynamic(const void* kernel, dim3 grid, dim3 block, size_t dynamic_lds, hipStream_t stream, void** args) {
  ProfileState& ps = profile_state();
  hipError_t err = hipSuccess;
  if (dynamic_lds > (size_t)kMaxLds) return XBIT_ERR_SHAPE;
  if (ps.dry_run) {  // recorded, not issued: no HIP call at all
    static const uintptr_t base = [] {
      Dl_info info;
      return dladdr(reinterpret_cast<const void*>(&xbit_abi_version), &info) != 0 ? reinterpret_cast<uintptr_t>(info.dli_fbase) : (uintptr_t)0;
    }();
    xbit_launch_record rec = {};
    rec.kernel_offset = (uint64_t)(reinterpret_cast<uintptr_t>(kernel) - base);
    rec.grid[0] = grid.x; rec.grid[1] = grid.y; rec.grid[2] = grid.z;
    rec.block[0] = block.x; rec.block[1] = block.y; rec.block[2] = block.z;
    rec.dynamic_lds_bytes = (uint32_t)dynamic_lds;
    ps.records.push_back(rec);
    return XBIT_OK;
  }
  if (dynamic_lds > (size_t)(32 << 10)) {  // statics + dynamic may pass the 64 KB default
    const int st = raise_lds_limit(kernel);
    if (st != XBIT_OK) return st;
  }
  if (ps.series && ps.series_used < ps.series_capacity) {
    err = hipExtLaunchKernel(kernel, grid, block, args, dynamic_lds, stream, ps.series_events[2 * ps.series_used], ps.series_events[2 * ps.series_used + 1], 0);
    if (err == hipSuccess) ++ps.series_used;
  } else if (ps.enabled) {
    err = hipExtLaunchKernel(kernel, grid, block, args, dynamic_lds, stream, ps.start, ps.stop, 0);
    ps.pending = (err == hipSuccess);
  } else {
    err = hipLaunchKernel(kernel, grid, block, args, dynamic_lds, stream);
  }
  if (err != hipSuccess) {
    (void)hipGetLastError();  // clear the sticky error; the caller gets the code
    mark_workspaces_dirty();  // whatever ran before may have died with scratch in use
    return (int)err;
  }
  return XBIT_OK;
}

// ---- split-K scratch -------------------------------------------------------------------------------
namespace {
constexpr size_t kStreamSlots = kMaxSplitSlots;       // per (device, stream) buffer for eager launches: 2 MiB
constexpr size_t kPoolSlots = (size_t)1 << 20;        // per-device pool for captured launches: 8 MiB; ranges go back only through xbit_workspace_release
struct CapturePool {
  unsigned long long* base = nullptr;
  size_t used = 0;
};
std::mutex g_ws_mutex;
std::map<std::pair<int, hipStream_t>, GemvWorkspace> g_ws;  // lives as long as the process
std::map<int, CapturePool> g_pool;
std::atomic<uint64_t> g_ws_dirty{0};  // bit d: a launch failed on device d, its accumulators may hold partial sums (devices >= 64 share bit 63)
inline uint64_t dirty_bit(int dev) { return 1ull << (dev < 0 ? 0 : (dev > 63 ? 63 : dev)); }

// hipMalloc + zero-fill; never called for a capturing stream (allocation inside a capture is not legal)
unsigned long long* alloc_zeroed(size_t slots) {
  unsigned long long* ptr = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&ptr), slots * sizeof(unsigned long long)) == hipSuccess &&
            hipMemset(ptr, 0, slots * sizeof(unsigned long long)) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
  }
  return ptr;
}

int reset_workspaces_locked(int dev) {
  hipError_t err = hipDeviceSynchronize();
  for (auto& kv : g_ws)
    if (err == hipSuccess && kv.first.first == dev && kv.second.acc) err = hipMemset(kv.second.acc, 0, kv.second.slots * sizeof(unsigned long long));
  auto it = g_pool.find(dev);
  if (err == hipSuccess && it != g_pool.end() && it->second.base) err = hipMemset(it->second.base, 0, kPoolSlots * sizeof(unsigned long long));
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return (int)err;
  }
  return XBIT_OK;
}
}  // namespace

void mark_workspaces_dirty() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    g_ws_dirty.store(~0ull, std::memory_order_relaxed);  // which device is unknown: every one re-zeroes before its next split launch
    return;
  }
  g_ws_dirty.fetch_or(dirty_bit(dev), std::memory_order_relaxed);
}

bool gemv_workspace(hipStream_t stream, size_t slots, const GemvLaunchOpts& opts, GemvWorkspace* out) {
  if (slots == 0 || slots > kMaxSplitSlots) return false;
  if (profile_state().dry_run && !opts.user_ws) {  // nothing is launched: a placeholder that is never followed
    out->acc = reinterpret_cast<unsigned long long*>((uintptr_t)64);
    out->slots = slots;
    return true;
  }
  if (opts.user_ws) {  // the caller's scratch, or nothing: never mix the two silently
    if (opts.user_ws_slots < slots) return false;
    out->acc = opts.user_ws;
    out->slots = opts.user_ws_slots;
    return true;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (cap != hipStreamCaptureStatusNone) {
    // A captured launch keeps its scratch for as long as the graph may be replayed - on any stream,
    // beside anything - so it gets a range of its own that is never handed out again.
    // The pool is created by the first eager split-K launch on the device (or xbit_workspace_reset):
    // nothing is allocated inside a capture, a capture that comes first runs its launches unsplit.
    CapturePool& pool = g_pool[dev];
    if (!pool.base || pool.used + slots > kPoolSlots) return false;
    out->acc = pool.base + pool.used;
    out->slots = slots;
    pool.used += (slots + 15) & ~(size_t)15;  // 128-byte granules: ranges never share a cache line
    return true;
  }
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;  // another thread may be capturing in global mode: the sync, memset and malloc below must not break its capture
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (g_ws_dirty.load(std::memory_order_relaxed) & dirty_bit(dev)) {  // this device only: another device's flag stays up until a launch there
    g_ws_dirty.fetch_and(~dirty_bit(dev), std::memory_order_relaxed);
    if (reset_workspaces_locked(dev) != XBIT_OK) {
      g_ws_dirty.fetch_or(dirty_bit(dev), std::memory_order_relaxed);
      (void)hipThreadExchangeStreamCaptureMode(&mode);
      return false;
    }
  }
  CapturePool& pool = g_pool[dev];
  if (!pool.base) pool.base = alloc_zeroed(kPoolSlots);  // a failure is not remembered: the next launch tries again
  GemvWorkspace& ws = g_ws[{dev, stream}];
  if (!ws.acc) {
    ws.acc = alloc_zeroed(kStreamSlots);
    ws.slots = ws.acc ? kStreamSlots : 0;
  }
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (!ws.acc) return false;
  *out = ws;
  return true;
}

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
    case XBIT_ERR_UNSUPPORTED: return "xbitops: option not supported by this build of the library (measurement variants need libxbitops_hip_tuning.so)";
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

int xbit_gemv_f16_ex(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, int bits,
                     int groupsize, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream,
                     const xbit_gemv_options* options) {
  if (!out || !a) return XBIT_ERR_NULL_POINTER;
  const int st = xbit::check_common(qweight, scales, qzeros, bits, groupsize, k, n);
  if (st != XBIT_OK) return st;
  if (m == 0 || (uint64_t)m * k >= (1ull << 40) || (uint64_t)m * n >= (1ull << 40)) return XBIT_ERR_SHAPE;
  if (((uintptr_t)out & 1u) || ((uintptr_t)a & 1u)) return XBIT_ERR_ALIGNMENT;
  xbit::GemvLaunchOpts opts;
  if (options) {
    // fields beyond the caller's struct_bytes keep their defaults
    const uint32_t have = options->struct_bytes;
    auto has = [&](size_t offset, size_t size) { return have >= offset + size; };
    if (has(offsetof(xbit_gemv_options, split_k), sizeof(int32_t))) opts.split = options->split_k;
    if (has(offsetof(xbit_gemv_options, workspace_bytes), sizeof(uint64_t)) && options->workspace) {
      if ((uintptr_t)options->workspace & 7u) return XBIT_ERR_ALIGNMENT;
      opts.user_ws = static_cast<unsigned long long*>(options->workspace);
      opts.user_ws_slots = (size_t)(options->workspace_bytes / sizeof(unsigned long long));
    }
    if (has(offsetof(xbit_gemv_options, variant), sizeof(int32_t))) opts.variant = options->variant;
    if (has(offsetof(xbit_gemv_options, trace), sizeof(void*))) opts.trace = static_cast<uint64_t*>(options->trace);
    if (opts.split < 0 || opts.split > 15) return XBIT_ERR_SHAPE;
#if !XBIT_TUNING
    if (opts.variant > 0 || opts.trace) return XBIT_ERR_UNSUPPORTED;
#endif
  }
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
  p.trace = opts.trace;
  p.splits = 1;
  p.bps = 0;
  p.acc = nullptr;
  return xbit::launch_gemv(p, bits, static_cast<hipStream_t>(hip_stream), opts);
}

int xbit_gemv_f16(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, int bits,
                  int groupsize, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream) {
  return xbit_gemv_f16_ex(out, a, qweight, scales, qzeros, bits, groupsize, m, k, n, add_zero_bias, hip_stream, nullptr);
}

uint64_t xbit_gemv_workspace_bytes(uint32_t m, uint32_t n) {
  // split launches run 1, 4, 8, 16 or 32 rows per pass (the build that takes m: 9..12 rows need the 16-row build's 16) over
  // tiles of 32 or 64 columns: round both up; beyond 32 rows, passes of 4
  const uint64_t rows = m <= 1 ? 1 : (m <= 4 ? 4 : (m <= 8 ? 8 : (m <= 16 ? 16 : (m <= 32 ? 32 : (((uint64_t)m + 3) / 4) * 4))));
  const uint64_t cols = (((uint64_t)n + 63) / 64) * 64;
  return rows * cols * sizeof(unsigned long long);
}

int xbit_workspace_reset(void) {
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return (int)err;
  }
  std::lock_guard<std::mutex> lock(xbit::g_ws_mutex);
  xbit::CapturePool& pool = xbit::g_pool[dev];
  if (!pool.base) pool.base = xbit::alloc_zeroed(xbit::kPoolSlots);  // so that a capture may come before any eager launch
  xbit::g_ws_dirty.fetch_and(~xbit::dirty_bit(dev), std::memory_order_relaxed);
  const int st = xbit::reset_workspaces_locked(dev);
  if (st != XBIT_OK) xbit::g_ws_dirty.fetch_or(xbit::dirty_bit(dev), std::memory_order_relaxed);
  return st;
}

int xbit_workspace_release(void) {
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return (int)err;
  }
  std::lock_guard<std::mutex> lock(xbit::g_ws_mutex);
  auto it = xbit::g_pool.find(dev);
  if (it == xbit::g_pool.end() || !it->second.base) return XBIT_OK;
  // waits for everything in flight on the device, re-zeroes, then hands the whole pool out again
  const int st = xbit::reset_workspaces_locked(dev);
  if (st == XBIT_OK) it->second.used = 0;
  return st;
}

uint64_t xbit_workspace_capture_bytes_free(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  std::lock_guard<std::mutex> lock(xbit::g_ws_mutex);
  auto it = xbit::g_pool.find(dev);
  if (it == xbit::g_pool.end() || !it->second.base) return 0;
  return (uint64_t)(xbit::kPoolSlots - it->second.used) * sizeof(unsigned long long);
}

static int gidx_params(xbit::GemvGidxParams* p, void* out, const void* a, const void* qweight, const void* scales, const void* qzeros,
                       const void* g_idx, int bits, uint32_t groups, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias) {
  if (!out || !g_idx) return XBIT_ERR_NULL_POINTER;
  const int st = xbit::check_common(qweight, scales, qzeros, bits, 1, k, n);
  if (st != XBIT_OK) return st;
  if (groups == 0 || groups > 65535u) return XBIT_ERR_GROUPSIZE;
  if (((uintptr_t)out & 1u) || ((uintptr_t)g_idx & 3u) || ((uintptr_t)a & 1u)) return XBIT_ERR_ALIGNMENT;
  p->out = static_cast<uint16_t*>(out);
  p->a = static_cast<const uint16_t*>(a);
  p->qweight = static_cast<const uint32_t*>(qweight);
  p->scales = static_cast<const uint16_t*>(scales);
  p->qzeros = static_cast<const uint32_t*>(qzeros);
  p->g_idx = static_cast<const int32_t*>(g_idx);
  p->m = m;
  p->k = k;
  p->n = n;
  p->kw = xbit::packed_rows(k, bits);
  p->groups = groups;
  p->zw = xbit::packed_rows(n, bits);
  p->bias = add_zero_bias;
  return XBIT_OK;
}

int xbit_gemv_gidx_f16(void* out, const void* a, const void* qweight, const void* scales, const void* qzeros, const void* g_idx,
                       int bits, uint32_t groups, uint32_t m, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream) {
  if (!a) return XBIT_ERR_NULL_POINTER;
  if (m == 0 || (uint64_t)m * k >= (1ull << 40) || (uint64_t)m * n >= (1ull << 40)) return XBIT_ERR_SHAPE;
  xbit::GemvGidxParams p;
  const int st = gidx_params(&p, out, a, qweight, scales, qzeros, g_idx, bits, groups, m, k, n, add_zero_bias);
  if (st != XBIT_OK) return st;
  return xbit::launch_gemv_gidx(p, bits, static_cast<hipStream_t>(hip_stream));
}

int xbit_dequant_gidx_f16(void* out, const void* qweight, const void* scales, const void* qzeros, const void* g_idx, int bits,
                          uint32_t groups, uint32_t k, uint32_t n, uint8_t add_zero_bias, void* hip_stream) {
  xbit::GemvGidxParams p;
  const int st = gidx_params(&p, out, nullptr, qweight, scales, qzeros, g_idx, bits, groups, 0, k, n, add_zero_bias);
  if (st != XBIT_OK) return st;
  return xbit::launch_dequant_gidx(p, bits, static_cast<hipStream_t>(hip_stream));
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
  // One fused launch of the staged-operand kernel when the shape allows it: up to 16 rows, whole K, 48- or 64-column tiles -
  // whichever count sits closer under a multiple of the 256 CUs for the concatenated layer.
  if (count >= 2 && m <= 16 && (k % 8u == 0) && (groupsize % 32 == 0) && ((uintptr_t)a % 4 == 0)) {
    xbit::GemvSets sets = {};
    sets.count = (uint32_t)count;
    uint32_t widest = 0, t48 = 0, t64 = 0;
    for (int j = 0; j < count; ++j) {
      sets.n[j] = ns[j];
      sets.zw[j] = xbit::packed_rows(ns[j], bits);
      sets.qweight[j] = static_cast<const uint32_t*>(qweights[j]);
      sets.scales[j] = static_cast<const uint16_t*>(scales[j]);
      sets.qzeros[j] = static_cast<const uint32_t*>(qzeros[j]);
      sets.out[j] = static_cast<uint16_t*>(outs[j]);
      widest = ns[j] > widest ? ns[j] : widest;
      t48 += (ns[j] + 47u) / 48u;
      t64 += (ns[j] + 63u) / 64u;
    }
    auto fill = [](uint32_t tiles) { return (double)tiles / (double)(((tiles + 255u) / 256u) * 256u); };
    const int cpl = fill(t64) >= fill(t48) - 1e-9 ? 4 : 3;
    xbit::GemvParams p = {};
    p.out = sets.out[0];
    p.a = static_cast<const uint16_t*>(a);
    p.qweight = sets.qweight[0];
    p.scales = sets.scales[0];
    p.qzeros = sets.qzeros[0];
    p.m = m;
    p.k = k;
    p.n = widest;
    p.groupsize = (uint32_t)groupsize;
    p.kw = xbit::packed_rows(k, bits);
    p.groups = (k + (uint32_t)groupsize - 1) / (uint32_t)groupsize;
    p.zw = xbit::packed_rows(widest, bits);
    p.bias = add_zero_bias;
    p.splits = 1;
    const int st = xbit::launch_gemv_dyn_multi(p, sets, bits, stream, cpl);
    if (st != xbit::kGemvDynNotEligible) return st;
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

int xbit_profile_series_begin(int capacity) {
  if (capacity < 1 || capacity > (1 << 20)) return XBIT_ERR_SHAPE;
  xbit::ProfileState& ps = xbit::profile_state();
  while ((int)ps.series_events.size() < 2 * capacity) {
    hipEvent_t e = nullptr;
    const hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) {
      (void)hipGetLastError();
      return (int)err;
    }
    ps.series_events.push_back(e);
  }
  ps.series = true;
  ps.series_used = 0;
  ps.series_capacity = capacity;
  return XBIT_OK;
}

int xbit_profile_series_end(float* ms, int max_count, int* count) {
  if (!ms || !count) return XBIT_ERR_NULL_POINTER;
  xbit::ProfileState& ps = xbit::profile_state();
  if (!ps.series) return XBIT_ERR_SHAPE;
  ps.series = false;
  const int n = ps.series_used < max_count ? ps.series_used : (max_count > 0 ? max_count : 0);
  *count = 0;
  if (ps.series_used > 0) {
    hipError_t err = hipEventSynchronize(ps.series_events[2 * (ps.series_used - 1) + 1]);  // the stream runs them in order
    for (int i = 0; i < n && err == hipSuccess; ++i) err = hipEventElapsedTime(&ms[i], ps.series_events[2 * i], ps.series_events[2 * i + 1]);
    if (err != hipSuccess) {
      (void)hipGetLastError();
      return (int)err;
    }
  }
  *count = n;
  return XBIT_OK;
}

int xbit_dry_run_begin(void) {
  xbit::ProfileState& ps = xbit::profile_state();
  ps.dry_run = true;
  ps.records.clear();
  return XBIT_OK;
}

int xbit_dry_run_end(xbit_launch_record* records, int max_count, int* count) {
  xbit::ProfileState& ps = xbit::profile_state();
  const bool was_on = ps.dry_run;
  ps.dry_run = false;
  if (!count || (max_count > 0 && !records)) return XBIT_ERR_NULL_POINTER;
  if (!was_on) return XBIT_ERR_SHAPE;
  const int n = (int)ps.records.size() < max_count ? (int)ps.records.size() : (max_count > 0 ? max_count : 0);
  for (int i = 0; i < n; ++i) records[i] = ps.records[i];
  *count = n;
  ps.records.clear();
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

}  // extern "C"
