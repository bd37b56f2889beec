// capi_host_driver.cc — the HOST half of the C ABI (xbitops_amd/csrc/xbit_capi.hip: argument checks, status codes, options
// parsing, split-K workspace pool bookkeeping, profiling series, dry run) compiled as plain C++ with AddressSanitizer +
// UndefinedBehaviorSanitizer and driven on the CPU.  xbit_capi.hip holds no kernel; what it calls is replaced here:
//   * the HIP runtime entry points it uses, by a mock backed by host memory (allocation failures and stream-capture status
//     can be injected),
//   * the launchers of the other translation units (launch_gemv, launch_dequant, ...), by recorders that return a status the
//     test chooses - and, for the split-K scratch tests, ask for workspaces the way the real launch rules do.
// tests/test_sanitizers.py builds and runs it:  g++ -std=c++17 -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -x c++ ...
// Any sanitizer report aborts; every expectation below that fails exits non-zero.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <set>
#include <vector>

#include "../../xbitops_amd/csrc/xbit_host.h"

#define EXPECT(cond)                                                    \
  do {                                                                  \
    if (!(cond)) {                                                      \
      fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

// ---- mock HIP runtime --------------------------------------------------------------------------------------------------
namespace mock {
int device = 0;
bool fail_get_device = false, fail_malloc = false, fail_launch = false;
hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
std::map<void*, size_t> allocations;
int memsets = 0, launches = 0, ext_launches = 0, syncs = 0, events = 0;
size_t last_lds = 0;
}  // namespace mock

extern "C" {
hipError_t hipGetDevice(int* d) {
  if (mock::fail_get_device) return hipErrorInvalidDevice;
  *d = mock::device;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock hip error"; }
hipError_t hipMalloc(void** p, size_t bytes) {
  if (mock::fail_malloc) return hipErrorOutOfMemory;
  *p = malloc(bytes);
  memset(*p, 0xAB, bytes);  // device memory is not zero: the library must clear what it hands out
  mock::allocations[*p] = bytes;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  mock::allocations.erase(p);
  free(p);
  return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t bytes) {
  // the range must lie inside one allocation (ASan would also catch an overrun of the malloc'd block)
  bool inside = false;
  for (auto& kv : mock::allocations)
    if ((char*)p >= (char*)kv.first && (char*)p + bytes <= (char*)kv.first + kv.second) inside = true;
  if (!inside) {
    fprintf(stderr, "hipMemset outside every allocation\n");
    exit(1);
  }
  memset(p, v, bytes);
  ++mock::memsets;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
  ++mock::syncs;
  return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) {
  *s = mock::capture;
  return hipSuccess;
}
hipError_t hipThreadExchangeStreamCaptureMode(hipStreamCaptureMode*) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes* fa, const void*) {
  memset(fa, 0, sizeof(*fa));
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t lds, hipStream_t) {
  if (mock::fail_launch) return hipErrorLaunchFailure;
  ++mock::launches;
  mock::last_lds = lds;
  return hipSuccess;
}
hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t lds, hipStream_t, hipEvent_t, hipEvent_t, int) {
  if (mock::fail_launch) return hipErrorLaunchFailure;
  ++mock::ext_launches;
  mock::last_lds = lds;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = reinterpret_cast<hipEvent_t>(malloc(8));  // never freed by the library (events live as long as the process): excluded from the leak check below
  ++mock::events;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.007f;
  return hipSuccess;
}
const char* __asan_default_options() { return "detect_leaks=0"; }  // the library's process-lifetime buffers and the mock's events are not leaks of interest
}

// ---- recorders for the launchers of the other translation units ----------------------------------------------------------
namespace rec {
int gemv = 0, dequant = 0, gidx = 0, dequant_gidx = 0, dyn_multi = 0;
int status = XBIT_OK;
int dyn_multi_status = xbit::kGemvDynNotEligible;
xbit::GemvParams last;
xbit::GemvLaunchOpts last_opts;
size_t want_slots = 0;  // launch_gemv asks for a split-K workspace of this many accumulators and launches once through the common path
xbit::GemvWorkspace got_ws;
bool got = false;
void dummy_kernel() {}
}  // namespace rec

namespace xbit {
int launch_gemv(const GemvParams& p, int, hipStream_t stream, const GemvLaunchOpts& opts) {
  ++rec::gemv;
  rec::last = p;
  rec::last_opts = opts;
  if (rec::want_slots) {
    rec::got = gemv_workspace(stream, rec::want_slots, opts, &rec::got_ws);
    void* args[1] = {nullptr};
    const int st = launch_raw_dynamic(reinterpret_cast<const void*>(&rec::dummy_kernel), dim3(4), dim3(64), 40000, stream, args);
    if (st != XBIT_OK) return st;
  }
  return rec::status;
}
int launch_dequant(const DequantParams&, int, hipStream_t) { return ++rec::dequant, rec::status; }
int launch_gemv_gidx(const GemvGidxParams&, int, hipStream_t) { return ++rec::gidx, rec::status; }
int launch_dequant_gidx(const GemvGidxParams&, int, hipStream_t) { return ++rec::dequant_gidx, rec::status; }
int launch_gemv_dyn_multi(const GemvParams&, const GemvSets&, int, hipStream_t, int) { return ++rec::dyn_multi, rec::dyn_multi_status; }
}  // namespace xbit

// ---- the tests ---------------------------------------------------------------------------------------------------------------
static void* P(uintptr_t v) { return reinterpret_cast<void*>(v); }

static void test_argument_checks() {
  void *out = P(0x1000), *a = P(0x2000), *qw = P(0x3000), *sc = P(0x4000), *qz = P(0x5000);
  EXPECT(xbit_abi_version() == XBIT_ABI_VERSION);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_OK && rec::gemv == 1);
  EXPECT(rec::last.kw == 512 && rec::last.groups == 32 && rec::last.zw == 512 && rec::last.bias == 1);
  EXPECT(xbit_gemv_f16(nullptr, a, qw, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_gemv_f16(out, nullptr, qw, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_gemv_f16(out, a, nullptr, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 1, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_BITS);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 9, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_BITS);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 0, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_GROUPSIZE);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 0, 4096, 4096, 1, nullptr) == XBIT_ERR_SHAPE);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 0, 4096, 1, nullptr) == XBIT_ERR_SHAPE);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 4096, 4095, 1, nullptr) == XBIT_ERR_SHAPE);       // n odd
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 8, 128, 1, 0xFFFFFFF0u, 0xFFFFFFF0u, 1, nullptr) == XBIT_ERR_SHAPE);  // sizes overflow
  EXPECT(xbit_gemv_f16(out, a, P(0x3002), sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_ALIGNMENT);
  EXPECT(xbit_gemv_f16(out, a, qw, P(0x4001), qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_ALIGNMENT);
  EXPECT(xbit_gemv_f16(P(0x1001), a, qw, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_ALIGNMENT);
  EXPECT(rec::gemv == 1);  // none of the rejected calls reached a launcher
  EXPECT(xbit_dequant_f16(out, qw, sc, qz, 3, 128, 4096, 4096, 0, nullptr) == XBIT_OK && rec::dequant == 1);
  EXPECT(xbit_dequant_f16(nullptr, qw, sc, qz, 3, 128, 4096, 4096, 0, nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_dequant_f16(out, qw, sc, qz, 3, 128, 4096, 2, 0, nullptr) == XBIT_OK);
  // a launcher's status comes back unchanged (a positive hipError_t included)
  rec::status = 719;
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr) == 719);
  rec::status = XBIT_OK;
  for (int s : {0, -1, -2, -3, -4, -5, -6, 719, -99}) EXPECT(xbit_status_string(s) != nullptr && strlen(xbit_status_string(s)) > 0);
}

static void test_options() {
  void *out = P(0x1000), *a = P(0x2000), *qw = P(0x3000), *sc = P(0x4000), *qz = P(0x5000);
  xbit_gemv_options o;
  memset(&o, 0, sizeof(o));
  o.struct_bytes = sizeof(o);
  o.split_k = 3;
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 16384, 1024, 1, nullptr, &o) == XBIT_OK && rec::last_opts.split == 3);
  o.split_k = 16;
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 16384, 1024, 1, nullptr, &o) == XBIT_ERR_SHAPE);
  o.split_k = -1;
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 16384, 1024, 1, nullptr, &o) == XBIT_ERR_SHAPE);
  o.split_k = 0;
  o.variant = 64;  // measurement variants: not in the production build
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 4096, 4096, 1, nullptr, &o) == (XBIT_TUNING ? XBIT_OK : XBIT_ERR_UNSUPPORTED));
  o.variant = 0;
  o.workspace = P(0x7004);  // 8-byte alignment required
  o.workspace_bytes = 1 << 20;
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 16384, 1024, 1, nullptr, &o) == XBIT_ERR_ALIGNMENT);
  o.workspace = P(0x7000);
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 16384, 1024, 1, nullptr, &o) == XBIT_OK);
  EXPECT(rec::last_opts.user_ws == P(0x7000) && rec::last_opts.user_ws_slots == (1u << 20) / 8);
  // an older caller's shorter struct: fields beyond struct_bytes keep their defaults, nothing beyond it is read
  struct Short {
    uint32_t struct_bytes;
    int32_t split_k;
  } sh = {sizeof(Short), 2};
  char* guarded = static_cast<char*>(malloc(sizeof(Short)));  // ASan flags a read past the 8 bytes
  memcpy(guarded, &sh, sizeof(sh));
  EXPECT(xbit_gemv_f16_ex(out, a, qw, sc, qz, 4, 128, 1, 16384, 1024, 1, nullptr, reinterpret_cast<xbit_gemv_options*>(guarded)) == XBIT_OK);
  EXPECT(rec::last_opts.split == 2 && rec::last_opts.user_ws == nullptr && rec::last_opts.variant <= 0);
  free(guarded);
  EXPECT(xbit_gemv_workspace_bytes(1, 64) == 64 * 8 && xbit_gemv_workspace_bytes(3, 4000) == 4 * 4032 * 8);
  EXPECT(xbit_gemv_workspace_bytes(5, 4096) == 8 * 4096 * 8 && xbit_gemv_workspace_bytes(10, 4096) == 16 * 4096 * 8 && xbit_gemv_workspace_bytes(17, 64) == 32 * 64 * 8 && xbit_gemv_workspace_bytes(33, 64) == 36 * 64 * 8);
}

static void test_workspace_pool() {
  using namespace xbit;
  GemvLaunchOpts none;
  GemvWorkspace w1, w2, w3;
  hipStream_t s1 = reinterpret_cast<hipStream_t>(0x10), s2 = reinterpret_cast<hipStream_t>(0x20);
  // a capture that comes before any eager launch finds no pool: unsplit
  mock::capture = hipStreamCaptureStatusActive;
  EXPECT(!gemv_workspace(s1, 1024, none, &w1));
  EXPECT(xbit_workspace_capture_bytes_free() == 0);
  mock::capture = hipStreamCaptureStatusNone;
  // an allocation failure is not remembered
  mock::fail_malloc = true;
  EXPECT(!gemv_workspace(s1, 1024, none, &w1));
  mock::fail_malloc = false;
  const int memsets0 = mock::memsets;
  EXPECT(gemv_workspace(s1, 1024, none, &w1) && w1.acc != nullptr && w1.slots == kMaxSplitSlots);
  EXPECT(mock::memsets >= memsets0 + 2);  // the stream buffer and the capture pool were zeroed
  for (size_t i = 0; i < w1.slots; ++i) EXPECT(w1.acc[i] == 0ull);
  EXPECT(gemv_workspace(s1, 4096, none, &w2) && w2.acc == w1.acc);   // one buffer per (device, stream)
  EXPECT(gemv_workspace(s2, 4096, none, &w3) && w3.acc != w1.acc);
  EXPECT(!gemv_workspace(s1, 0, none, &w2) && !gemv_workspace(s1, kMaxSplitSlots + 1, none, &w2));
  // caller-owned scratch: used when large enough, never mixed with the library's
  GemvLaunchOpts user;
  std::vector<unsigned long long> mine(2048, 0ull);
  user.user_ws = mine.data();
  user.user_ws_slots = mine.size();
  EXPECT(gemv_workspace(s1, 2048, user, &w2) && w2.acc == mine.data());
  EXPECT(!gemv_workspace(s1, 2049, user, &w2));
  // captured launches: private ranges of the pool, 128-byte granules, never handed out twice, exhaustion -> unsplit
  mock::capture = hipStreamCaptureStatusActive;
  const uint64_t free0 = xbit_workspace_capture_bytes_free();
  EXPECT(free0 == ((uint64_t)1 << 20) * 8);
  std::set<unsigned long long*> seen;
  GemvWorkspace c;
  EXPECT(gemv_workspace(s1, 100, none, &c) && c.slots == 100);
  seen.insert(c.acc);
  unsigned long long* first = c.acc;
  EXPECT(gemv_workspace(s2, 100, none, &c) && c.acc == first + 112);  // 100 -> 112 slots (16-slot granules)
  EXPECT(xbit_workspace_capture_bytes_free() == free0 - 2 * 112 * 8);
  size_t handed = 224;
  while (gemv_workspace(s1, kMaxSplitSlots, none, &c)) {
    EXPECT(seen.insert(c.acc).second);
    c.acc[0] = 1;  // in bounds (ASan): first and last slot of the range
    c.acc[c.slots - 1] = 1;
    handed += kMaxSplitSlots;
  }
  EXPECT(handed + kMaxSplitSlots > ((size_t)1 << 20) && handed <= ((size_t)1 << 20));
  mock::capture = hipStreamCaptureStatusNone;
  // release: waits, re-zeroes, hands the whole pool out again
  EXPECT(xbit_workspace_release() == XBIT_OK && xbit_workspace_capture_bytes_free() == free0);
  mock::capture = hipStreamCaptureStatusActive;
  EXPECT(gemv_workspace(s1, 100, none, &c) && c.acc == first && c.acc[0] == 0ull);
  mock::capture = hipStreamCaptureStatusNone;
  // a failed launch marks the device's scratch dirty: re-zeroed before the next split launch
  w1.acc[5] = 77;
  void *out = P(0x1000), *a = P(0x2000), *qw = P(0x3000), *sc = P(0x4000), *qz = P(0x5000);
  rec::want_slots = 512;
  mock::fail_launch = true;
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 16384, 512, 1, s1) == (int)hipErrorLaunchFailure);
  mock::fail_launch = false;
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 16384, 512, 1, s1) == XBIT_OK && rec::got && rec::got_ws.acc == w1.acc);
  EXPECT(w1.acc[5] == 0ull);  // cleaned
  EXPECT(mock::last_lds == 40000);
  rec::want_slots = 0;
  EXPECT(xbit_workspace_reset() == XBIT_OK);
  // another device has its own pool and buffers; hipGetDevice failing is reported, not guessed around
  mock::device = 3;
  EXPECT(xbit_workspace_capture_bytes_free() == 0);
  EXPECT(gemv_workspace(s1, 64, none, &c) && c.acc != w1.acc);
  mock::device = 0;
  mock::fail_get_device = true;
  EXPECT(!gemv_workspace(s1, 64, none, &c));
  EXPECT(xbit_workspace_reset() != XBIT_OK && xbit_workspace_release() != XBIT_OK && xbit_workspace_capture_bytes_free() == 0);
  mock::fail_get_device = false;
}

static void test_profile_and_dry_run() {
  void *out = P(0x1000), *a = P(0x2000), *qw = P(0x3000), *sc = P(0x4000), *qz = P(0x5000);
  float ms[8];
  int count = -1;
  EXPECT(xbit_profile_series_begin(0) == XBIT_ERR_SHAPE && xbit_profile_series_end(ms, 8, &count) == XBIT_ERR_SHAPE);
  EXPECT(xbit_profile_series_begin(3) == XBIT_OK);
  rec::want_slots = 64;
  const int ext0 = mock::ext_launches, plain0 = mock::launches;
  for (int i = 0; i < 5; ++i) EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 16384, 512, 1, nullptr) == XBIT_OK);
  EXPECT(mock::ext_launches == ext0 + 3 && mock::launches == plain0 + 2);  // capacity 3: the rest run untimed
  EXPECT(xbit_profile_series_end(nullptr, 8, &count) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_profile_series_begin(3) == XBIT_OK);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 16384, 512, 1, nullptr) == XBIT_OK);
  EXPECT(xbit_profile_series_end(ms, 8, &count) == XBIT_OK && count == 1 && ms[0] > 0.f);
  float one = 0;
  EXPECT(xbit_profile_last_ms(&one) == XBIT_ERR_SHAPE);  // nothing pending
  EXPECT(xbit_profile_enable(1) == XBIT_OK);
  EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 16384, 512, 1, nullptr) == XBIT_OK);
  EXPECT(xbit_profile_last_ms(&one) == XBIT_OK && one > 0.f && xbit_profile_last_ms(nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_profile_enable(0) == XBIT_OK);
  // dry run: recorded, not issued; no workspace is allocated; ending twice is an error
  xbit_launch_record r[4];
  const int launched = mock::launches + mock::ext_launches;
  const size_t allocs = mock::allocations.size();
  EXPECT(xbit_dry_run_begin() == XBIT_OK);
  for (int i = 0; i < 3; ++i) EXPECT(xbit_gemv_f16(out, a, qw, sc, qz, 4, 128, 1, 16384, 512, 1, reinterpret_cast<void*>(0x99)) == XBIT_OK);
  EXPECT(xbit_dry_run_end(r, 2, &count) == XBIT_OK && count == 2);
  EXPECT(r[0].grid[0] == 4 && r[0].block[0] == 64 && r[0].dynamic_lds_bytes == 40000 && r[1].kernel_offset == r[0].kernel_offset);
  EXPECT(mock::launches + mock::ext_launches == launched && mock::allocations.size() == allocs);
  EXPECT(xbit_dry_run_end(r, 2, &count) == XBIT_ERR_SHAPE);
  EXPECT(xbit_dry_run_begin() == XBIT_OK && xbit_dry_run_end(nullptr, 0, &count) == XBIT_OK && count == 0);
  EXPECT(xbit_dry_run_begin() == XBIT_OK && xbit_dry_run_end(nullptr, 2, &count) == XBIT_ERR_NULL_POINTER);
  rec::want_slots = 0;
}

static void test_multi_and_gidx() {
  void *a = P(0x2000), *gi = P(0x6000);
  void* outs[4] = {P(0x1000), P(0x1100), P(0x1200), P(0x1300)};
  const void* qws[4] = {P(0x3000), P(0x3100), P(0x3200), P(0x3300)};
  const void* scs[4] = {P(0x4000), P(0x4100), P(0x4200), P(0x4300)};
  const void* qzs[4] = {P(0x5000), P(0x5100), P(0x5200), P(0x5300)};
  uint32_t ns[4] = {4096, 1024, 1024, 512};
  EXPECT(xbit_gemv_multi_f16(0, outs, a, qws, scs, qzs, ns, 4, 128, 1, 4096, 1, nullptr) == XBIT_ERR_SHAPE);
  EXPECT(xbit_gemv_multi_f16(5, outs, a, qws, scs, qzs, ns, 4, 128, 1, 4096, 1, nullptr) == XBIT_ERR_SHAPE);
  EXPECT(xbit_gemv_multi_f16(3, nullptr, a, qws, scs, qzs, ns, 4, 128, 1, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  void* holes[3] = {P(0x1000), nullptr, P(0x1200)};
  EXPECT(xbit_gemv_multi_f16(3, holes, a, qws, scs, qzs, ns, 4, 128, 1, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  uint32_t odd[3] = {4096, 1023, 1024};
  EXPECT(xbit_gemv_multi_f16(3, outs, a, qws, scs, qzs, odd, 4, 128, 1, 4096, 1, nullptr) == XBIT_ERR_SHAPE);
  // the staged-operand kernel's fused build is asked first; not eligible -> one launch per set
  rec::dyn_multi = rec::gemv = 0;
  rec::dyn_multi_status = XBIT_OK;
  EXPECT(xbit_gemv_multi_f16(4, outs, a, qws, scs, qzs, ns, 4, 128, 16, 4096, 1, nullptr) == XBIT_OK && rec::dyn_multi == 1 && rec::gemv == 0);
  rec::dyn_multi_status = xbit::kGemvDynNotEligible;
  EXPECT(xbit_gemv_multi_f16(4, outs, a, qws, scs, qzs, ns, 4, 128, 3, 4096, 1, nullptr) == XBIT_OK && rec::dyn_multi == 2 && rec::gemv == 4);
  rec::status = XBIT_ERR_SHAPE;
  EXPECT(xbit_gemv_multi_f16(4, outs, a, qws, scs, qzs, ns, 4, 128, 3, 4096, 1, nullptr) == XBIT_ERR_SHAPE && rec::gemv == 5);  // the per-set launches report their own status
  rec::status = XBIT_OK;
  EXPECT(xbit_gemv_multi_f16(4, outs, a, qws, scs, qzs, ns, 4, 128, 40, 4096, 1, nullptr) == XBIT_OK && rec::gemv == 9);  // 40 rows: one launch per set
  EXPECT(xbit_gemv_multi_f16(1, outs, a, qws, scs, qzs, ns, 4, 128, 1, 4096, 1, nullptr) == XBIT_OK && rec::gemv == 10);
  // act-order entries
  EXPECT(xbit_gemv_gidx_f16(outs[0], a, qws[0], scs[0], qzs[0], gi, 4, 32, 1, 4096, 4096, 1, nullptr) == XBIT_OK && rec::gidx == 1);
  EXPECT(xbit_gemv_gidx_f16(outs[0], a, qws[0], scs[0], qzs[0], nullptr, 4, 32, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_gemv_gidx_f16(outs[0], a, qws[0], scs[0], qzs[0], gi, 4, 0, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_GROUPSIZE);
  EXPECT(xbit_gemv_gidx_f16(outs[0], a, qws[0], scs[0], qzs[0], gi, 4, 70000, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_GROUPSIZE);
  EXPECT(xbit_gemv_gidx_f16(outs[0], a, qws[0], scs[0], qzs[0], P(0x6002), 4, 32, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_ALIGNMENT);
  EXPECT(xbit_gemv_gidx_f16(outs[0], nullptr, qws[0], scs[0], qzs[0], gi, 4, 32, 1, 4096, 4096, 1, nullptr) == XBIT_ERR_NULL_POINTER);
  EXPECT(xbit_gemv_gidx_f16(outs[0], a, qws[0], scs[0], qzs[0], gi, 4, 32, 0, 4096, 4096, 1, nullptr) == XBIT_ERR_SHAPE);
  EXPECT(xbit_dequant_gidx_f16(outs[0], qws[0], scs[0], qzs[0], gi, 4, 32, 4096, 4096, 1, nullptr) == XBIT_OK && rec::dequant_gidx == 1);
}

int main() {
  test_argument_checks();
  test_options();
  test_workspace_pool();
  test_profile_and_dry_run();
  test_multi_and_gidx();
  printf("capi host driver: all expectations hold\n");
  return 0;
}
