// empty_launch.hip — what does a launch cost by the kernel-only timer (hipExtLaunchKernel start/stop events) as a
// function of grid and workgroup size, for a kernel that does nothing, one that reads its arguments and stores one
// dword per workgroup, and one that also allocates LDS?  (the fixed part of every short launch.)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__global__ void k_empty() {}
__global__ void k_store(uint32_t* out, uint32_t v) {
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
template <int LDS_DW>
__global__ void k_lds(uint32_t* out, uint32_t v) {
  __shared__ uint32_t s[LDS_DW];
  s[threadIdx.x] = v + threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[(v + 1) % LDS_DW];
}

template <typename F>
static void timeit(const char* name, int grid, int block, F launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  std::vector<float> us;
  for (int i = 0; i < 220; ++i) {
    launch(a, b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (i >= 20) us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  float avg = 0;
  for (float x : us) avg += x;
  printf("%-28s grid %5d x %4d  avg %5.2f  med %5.2f  min %5.2f us\n", name, grid, block, avg / us.size(), us[us.size() / 2], us[0]);
  fflush(stdout);
}

// period of back-to-back launches on one stream (what a step of a dependent chain costs): 400 launches queued
// behind a long blocker kernel so that the host's enqueue time is not in the span, events around the 400
__global__ void k_spin(uint64_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
template <typename F>
static void chain(const char* name, int grid, int block, F launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int n = 400;
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, (uint64_t)400000);  // 4 ms at 100 MHz
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    best = std::min(best, ms * 1e3f / n);
  }
  printf("%-28s grid %5d x %4d  period of back-to-back launches %5.2f us\n", name, grid, block, best);
  fflush(stdout);
}

int main() {
  uint32_t* out;
  (void)hipMalloc(&out, 1 << 20);
  for (int block : {64, 512})
    for (int grid : {1, 256, 2048}) chain("empty", grid, block, [&]() { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, 0); });
  chain("args + 1 store / workgroup", 256, 512, [&]() { hipLaunchKernelGGL(k_store, dim3(256), dim3(512), 0, 0, out, 7u); });
  chain("... + 10 KB LDS + barrier", 256, 512, [&]() { hipLaunchKernelGGL(k_lds<2560>, dim3(256), dim3(512), 0, 0, out, 7u); });
  for (int block : {64, 256, 512, 1024})
    for (int grid : {1, 32, 230, 256, 512, 1024, 2048}) {
      timeit("empty", grid, block, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, 0, a, b, 0); });
    }
  for (int block : {256, 512})
    for (int grid : {230, 256, 1024}) {
      timeit("args + 1 store / workgroup", grid, block, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_store, dim3(grid), dim3(block), 0, 0, a, b, 0, out, 7u); });
      timeit("... + 10 KB LDS + barrier", grid, block, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_lds<2560>, dim3(grid), dim3(block), 0, 0, a, b, 0, out, 7u); });
      timeit("... + 139 KB LDS + barrier", grid, block, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_lds<35584>, dim3(grid), dim3(block), 0, 0, a, b, 0, out, 7u); });
    }
  return 0;
}
