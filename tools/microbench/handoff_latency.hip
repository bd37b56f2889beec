// handoff_latency.hip — what does it cost, on MI355X, to get 256 bytes of partial sums from one
// workgroup to another INSIDE a launch, while every CU streams from HBM?  (an
// equal-work GEMV tiling stands or falls with the price of its cross-workgroup sum.)
//
// 256 workgroups (one per CU) each stream `kb` KB from HBM; then producers (even pair members) publish
// 64 x 8 bytes + a flag and stamp s_memrealtime; consumers (odd members) finish their own streaming,
// stamp, poll the flag, read the payload, stamp again.  Reported: consumer-side wait = t_seen -
// max(t_published, t_consumer_ready), median over pairs, for pairs on the SAME XCD (blocks b, b+8) and on
// DIFFERENT XCDs (b, b+1), and the protocols
//   0  sc1 (write-through) payload + sc1 flag, sc1-load poll + sc1 payload loads   (the portable form)
//   1  plain payload + plain flag, sc1-load poll: valid only when both sit on one XCD (L2 is the meeting point)
//   2  one returning 64-bit atomic add per element on a shared accumulator (the split-K protocol): time of the atomic
// The XCC id of every workgroup is recorded so that "same XCD" is checked, not assumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

struct Params {
  const u32x4* w;
  unsigned long long* payload;  // [pairs][64]
  unsigned int* flag;           // [pairs][32] (one 128-B line per pair)
  unsigned long long* acc;      // [pairs][64]
  unsigned long long* stamps;   // [wgs][4]: ready, published/seen, done, xcc
  uint32_t per_wg16;            // 16-byte elements streamed per workgroup
  uint32_t stride;              // partner distance: 8 = same XCD (observed), 1 = neighbouring XCD
  uint32_t epoch;
  int proto;
};

__global__ void __launch_bounds__(256) k_handoff(const Params p) {
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  // pairs: (b, b + stride) with b in the lower half of each 2*stride group
  const uint32_t grp = b / (2 * p.stride), in = b % (2 * p.stride);
  const bool producer = in < p.stride;
  const uint32_t pair = grp * p.stride + (in % p.stride);
  // stream: every workgroup reads its slice (the consumer reads 25 % more so that it is usually the later one)
  const uint32_t n16 = producer ? p.per_wg16 : p.per_wg16 + p.per_wg16 / 4;
  const u32x4* src = p.w + (size_t)b * (p.per_wg16 * 2);
  uint32_t x = 0;
  for (uint32_t i = tid; i < n16; i += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (i + u * 256 < n16) ? src[i + u * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; ++u) x ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  __shared__ uint32_t s_x[256];
  s_x[tid] = x;
  __syncthreads();
  unsigned long long t_ready = 0, t_mid = 0, t_done = 0;
  gu64* pay = (gu64*)(p.payload + (size_t)pair * 64);
  gu32* flag = (gu32*)(p.flag + (size_t)pair * 32);
  gu64* acc = (gu64*)(p.acc + (size_t)pair * 64);
  if (tid < 64) {  // one wave does the hand-off
    t_ready = __builtin_amdgcn_s_memrealtime();
    const unsigned long long val = ((unsigned long long)p.epoch << 32) | (s_x[tid] & 0xFFFFu) | 1u;
    if (p.proto == 2) {
      const unsigned long long old = __hip_atomic_fetch_add(acc + tid, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old));
      t_mid = __builtin_amdgcn_s_memrealtime();
      if ((old >> 32) == p.epoch) p.payload[(size_t)pair * 64 + tid] = old + val;  // the later one "finalises"
      t_done = __builtin_amdgcn_s_memrealtime();
    } else if (producer) {
      if (p.proto == 0) {
        __hip_atomic_store(pay + tid, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 store
      } else {
        p.payload[(size_t)pair * 64 + tid] = val;  // plain store: stays in this XCD's L2
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) {
        if (p.proto == 0) __hip_atomic_store(flag, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *((volatile unsigned int*)p.flag + (size_t)pair * 32) = p.epoch;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t_mid = __builtin_amdgcn_s_memrealtime();  // published (flag store acknowledged)
      t_done = t_mid;
    } else {
      uint32_t spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch && spins < (1u << 16)) {  // sc1-load poll, bounded
        __builtin_amdgcn_s_sleep(1);
        ++spins;
      }
      t_mid = __builtin_amdgcn_s_memrealtime();  // flag seen
      const unsigned long long got = __hip_atomic_load(pay + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(got));
      t_done = __builtin_amdgcn_s_memrealtime();  // payload in registers
      if ((got >> 32) != p.epoch || spins >= (1u << 16)) t_done = 0;  // stale or timed out: reported
    }
    if (tid == 0) {
      unsigned long long* o = p.stamps + (size_t)b * 4;
      o[0] = t_ready; o[1] = t_mid; o[2] = t_done; o[3] = xcc_id() | (producer ? 0x100u : 0u);
    }
  }
}

int main() {
  const int wgs = 256;
  const uint32_t kb = 92;  // per workgroup, like the metric GEMV
  Params p;
  p.per_wg16 = kb * 1024 / 16;
  const size_t copy16 = (size_t)wgs * p.per_wg16 * 2;
  const int copies = 16;
  u32x4* w;
  (void)hipMalloc(&w, copy16 * 16 * copies);
  (void)hipMemset(w, 1, copy16 * 16 * copies);
  (void)hipMalloc(&p.payload, 128 * 64 * 8);
  (void)hipMalloc(&p.flag, 128 * 32 * 4);
  (void)hipMalloc(&p.acc, 128 * 64 * 8);
  (void)hipMalloc(&p.stamps, wgs * 4 * 8);
  (void)hipMemset(p.payload, 0, 128 * 64 * 8);
  (void)hipMemset(p.flag, 0, 128 * 32 * 4);
  (void)hipMemset(p.acc, 0, 128 * 64 * 8);
  std::vector<unsigned long long> st(wgs * 4);
  uint32_t epoch = 1;
  for (int proto = 0; proto < 3; ++proto)
    for (uint32_t stride : {8u, 1u}) {
      std::vector<double> wait, total, t_atomic;
      int stale = 0, same = 0, pairs_n = 0;
      for (int it = 0; it < 60; ++it) {
        p.w = w + (size_t)(it % copies) * copy16;
        p.stride = stride;
        p.proto = proto;
        p.epoch = epoch++;
        if (proto == 2) (void)hipMemset(p.acc, 0, 128 * 64 * 8);
        hipLaunchKernelGGL(k_handoff, dim3(wgs), dim3(256), 0, 0, p);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(st.data(), p.stamps, wgs * 4 * 8, hipMemcpyDeviceToHost);
        if (it < 10) continue;
        for (int b = 0; b < wgs; ++b) {
          const uint32_t in = b % (2 * stride);
          if (in >= stride) continue;  // b = producer, b + stride = consumer
          const unsigned long long* a = &st[(size_t)b * 4];
          const unsigned long long* c = &st[(size_t)(b + stride) * 4];
          ++pairs_n;
          if ((a[3] & 0xF) == (c[3] & 0xF)) ++same;
          if (proto == 2) {
            t_atomic.push_back((double)(c[1] - c[0]) * 0.01);
            t_atomic.push_back((double)(a[1] - a[0]) * 0.01);
            continue;
          }
          if (c[2] == 0) { ++stale; continue; }
          const unsigned long long start = std::max(a[1], c[0]);
          wait.push_back(c[1] > start ? (double)(c[1] - start) * 0.01 : 0.0);   // flag seen after both were ready
          total.push_back((double)(c[2] - std::max(a[1], c[0])) * 0.01);        // ... payload in registers
        }
      }
      auto med = [](std::vector<double>& v, double q) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
      if (proto == 2)
        printf("proto 2 (returning 64-bit atomic), partner distance %u: same-XCD pairs %d/%d; atomic round trip p50 %.2f p90 %.2f us\n", stride, same, pairs_n,
               med(t_atomic, 0.5), med(t_atomic, 0.9));
      else
        printf("proto %d (%s), partner distance %u: same-XCD pairs %d/%d, stale/timeouts %d; flag seen after both ready p50 %.2f p90 %.2f us; payload in registers p50 %.2f p90 %.2f us\n",
               proto, proto == 0 ? "sc1 stores + sc1 loads" : "plain stores + sc1 loads", stride, same, pairs_n, stale, med(wait, 0.5), med(wait, 0.9), med(total, 0.5),
               med(total, 0.9));
      fflush(stdout);
    }
  return 0;
}
