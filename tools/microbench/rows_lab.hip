// rows_lab.hip — lab harness (NOT part of the library) of the shared-activation-tile kernel for 33..256 rows, xbit_gemv_rows.h: launches
// rows::gemv_rows_kernel directly in each of its four shapes (columns per group x groups per workgroup), checks the result against a host
// reference and times the chain period next to the production library's call at the same row count.  The kernel began here as a prototype
// (profiles/r04/rows_lab_m128.txt); the harness now compiles the library's header, so a kernel edit can be measured without rebuilding the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I xbitops_amd/csrc -I include tools/microbench/rows_lab.hip -o tools/microbench/rows_lab -ldl
//   tools/microbench/rows_lab [K N M [candidate mask, e.g. 0x21: the first and the sixth only, no library call]]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "xbit_gemv_rows.h"

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

// ---- host ------------------------------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double urand() { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }
static uint16_t h_bits(_Float16 h) {
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static _Float16 h_from(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return h;
}

typedef int (*gemv_fn)(void*, const void*, const void*, const void*, const void*, int, int, uint32_t, uint32_t, uint32_t, uint8_t, void*);

template <int CPL, int NCG, int NW, int MT>
static void run_candidate(const char* name, uint32_t k, uint32_t n, uint32_t m, const std::vector<uint32_t*>& qw, const std::vector<uint16_t*>& sc, const std::vector<uint32_t*>& qz,
                          uint16_t* act, uint16_t* out, const std::vector<double>& y_ref, hipStream_t stream) {
  constexpr int BITS = 4, CN = 16 * CPL;
  const uint32_t kw = k / 8, groups = k / 128, zw = (n * 4 + 31) / 32, units = (k / 32 + 3) / 4;
  const uint32_t tiles = (n + CN - 1) / CN, xtiles = (tiles + NCG - 1) / NCG, row_blocks = (m + MT - 1) / MT;
  uint32_t gshift = 0;
  while ((32u << gshift) < 128u) ++gshift;
  const uint32_t tab_groups = ((units * 4 - 1u) >> gshift) + 1u;
  constexpr int NK = NW / NCG, IMG = 4 * 4 * (MT + 1) * 4;
  const uint32_t front = (uint32_t)(NK * 2 * IMG) + (uint32_t)NCG * tab_groups * 64u, red = (uint32_t)NW * MT * CN;
  const uint32_t red_dwords = front > red ? front : red;
  const size_t lds = (size_t)red_dwords * 4 + NW * 4;
  if (lds > (160u << 10)) {
    printf("%-64s n/a: %zu B of LDS\n", name, lds);
    return;
  }
  auto kern = xbit::rows::gemv_rows_kernel<BITS, CPL, NCG, NW, MT>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto launch = [&](int c) {
    hipLaunchKernelGGL(kern, dim3(8u * ((xtiles + 7u) / 8u) * row_blocks), dim3(64 * NW), lds, stream, qw[c], n, xtiles, kw, units, act, sc[c], qz[c], out, k, groups, zw, 1u, m, tab_groups, gshift, red_dwords);
  };
  CK(hipMemsetAsync(out, 0xFF, (size_t)m * n * 2, stream));
  launch(0);
  CK(hipStreamSynchronize(stream));
  std::vector<uint16_t> y((size_t)m * n);
  CK(hipMemcpy(y.data(), out, y.size() * 2, hipMemcpyDeviceToHost));
  size_t bad = 0;
  double worst = 0;
  for (size_t i = 0; i < y.size(); ++i) {
    const double v = (double)(float)h_from(y[i]), r = y_ref[i], err = fabs(v - r);
    if (!(err <= 1e-3 * fabs(r) + 6.2e-5)) ++bad;
    worst = std::max(worst, err);
  }
  const int copies = (int)qw.size();
  for (int c = 0; c < 2 * copies; ++c) launch(c % copies);
  CK(hipStreamSynchronize(stream));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ch;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipStreamSynchronize(stream));
    CK(hipEventRecord(e0, stream));
    for (int i = 0; i < 200; ++i) launch((i + 3 * rep) % copies);
    CK(hipEventRecord(e1, stream));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ch.push_back(ms * 1e3f / 200.f);
  }
  std::sort(ch.begin(), ch.end());
  printf("%-58s %s bad %zu worst %.2e | workgroups %u x %u, LDS %zu B | chain med %6.2f min %6.2f us/launch\n", name, bad == 0 ? "ok " : "BAD", bad, worst, xtiles, row_blocks, lds, ch[2], ch[0]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t k = argc > 3 ? (uint32_t)atoi(argv[1]) : 4096, n = argc > 3 ? (uint32_t)atoi(argv[2]) : 11008, m = argc > 3 ? (uint32_t)atoi(argv[3]) : 128;
  const uint32_t kw = k / 8, groups = k / 128, zw = (n * 4 + 31) / 32;
  std::vector<uint32_t> qweight((size_t)kw * n), qzeros((size_t)groups * zw);
  std::vector<uint16_t> scales((size_t)groups * n), act((size_t)m * k);
  for (auto& w : qweight) w = (uint32_t)rnd();
  for (auto& z : qzeros) z = (uint32_t)rnd();
  for (auto& s : scales) s = h_bits((_Float16)(float)(0.002 + 0.008 * urand()));
  for (auto& a : act) a = h_bits((_Float16)(float)(2.0 * urand() - 1.0));
  // host reference: W = fp16(fma(q, s, -fp16((z + 1) * s))), fp64 sums; W once, then M dot products per column
  std::vector<double> y_ref((size_t)m * n, 0.0);
  {
    std::vector<float> wcol(k);
    for (uint32_t c = 0; c < n; ++c) {
      for (uint32_t kk = 0; kk < k; ++kk) {
        const uint32_t q = (qweight[(size_t)(kk / 8) * n + c] >> (4 * (kk % 8))) & 15u, gi = kk / 128;
        const uint32_t z = ((qzeros[(size_t)gi * zw + c / 8] >> (4 * (c % 8))) & 15u) + 1u;
        const _Float16 s = h_from(scales[(size_t)gi * n + c]);
        const _Float16 zs = (_Float16)((float)z * (float)s);
        wcol[kk] = (float)(_Float16)((float)q * (float)s - (float)zs);
      }
      for (uint32_t r = 0; r < m; ++r) {
        double sum = 0;
        for (uint32_t kk = 0; kk < k; ++kk) sum += (double)wcol[kk] * (double)(float)h_from(act[(size_t)r * k + kk]);
        y_ref[(size_t)r * n + c] = sum;
      }
    }
  }
  const size_t wbytes = qweight.size() * 4, sbytes = scales.size() * 2, zbytes = qzeros.size() * 4;
  const int copies = (int)((512ull << 20) / (wbytes + sbytes + zbytes)) + 2;
  std::vector<uint32_t*> dqw, dqz;
  std::vector<uint16_t*> dsc;
  for (int c = 0; c < copies; ++c) {
    uint32_t *w, *z;
    uint16_t* s;
    CK(hipMalloc(&w, wbytes));
    CK(hipMalloc(&s, sbytes));
    CK(hipMalloc(&z, zbytes));
    CK(hipMemcpy(w, qweight.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(s, scales.data(), sbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(z, qzeros.data(), zbytes, hipMemcpyHostToDevice));
    dqw.push_back(w);
    dsc.push_back(s);
    dqz.push_back(z);
  }
  uint16_t *dact, *dout;
  CK(hipMalloc(&dact, act.size() * 2));
  CK(hipMemcpy(dact, act.data(), act.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, (size_t)m * n * 2));
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  printf("# %ux%u w4 g128, %u rows, %d weight copies\n", k, n, m, copies);
  const unsigned mask = argc > 4 ? (unsigned)strtoul(argv[4], nullptr, 0) : 0xFFu;  // (the 4-wave builds - bits 8..10 - only on request)
  void* lib = argc > 4 ? nullptr : dlopen("xbitops_amd/lib/libxbitops_hip.so", RTLD_NOW);
  if (lib != nullptr) {
    gemv_fn gemv = (gemv_fn)dlsym(lib, "xbit_gemv_f16");
    for (int c = 0; c < 2 * copies; ++c) gemv(dout, dact, dqw[c % copies], dsc[c % copies], dqz[c % copies], 4, 128, m, k, n, 1, stream);
    CK(hipStreamSynchronize(stream));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ch;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipStreamSynchronize(stream));
      CK(hipEventRecord(e0, stream));
      for (int i = 0; i < 200; ++i) gemv(dout, dact, dqw[(i + 3 * rep) % copies], dsc[(i + 3 * rep) % copies], dqz[(i + 3 * rep) % copies], 4, 128, m, k, n, 1, stream);
      CK(hipEventRecord(e1, stream));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ch.push_back(ms * 1e3f / 200.f);
    }
    std::sort(ch.begin(), ch.end());
    printf("%-64s                                                        chain med %6.2f min %6.2f us/call\n", "production library (xbit_gemv_f16: blocks of 32 rows in grid.y)", ch[2], ch[0]);
  }
  if (mask & 1u) run_candidate<3, 2, 8, 32>("32 rows: 2 column groups x 4 k-slices, 48 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 2u) run_candidate<3, 4, 8, 32>("32 rows: 4 column groups x 2 k-slices, 48 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 4u) run_candidate<4, 2, 8, 32>("32 rows: 2 column groups x 4 k-slices, 64 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 8u) run_candidate<4, 4, 8, 32>("32 rows: 4 column groups x 2 k-slices, 64 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 16u) run_candidate<3, 2, 8, 64>("64 rows: 2 column groups x 4 k-slices, 48 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 32u) run_candidate<3, 4, 8, 64>("64 rows: 4 column groups x 2 k-slices, 48 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 64u) run_candidate<4, 2, 8, 64>("64 rows: 2 column groups x 4 k-slices, 64 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 256u) run_candidate<3, 2, 4, 32>("4 waves, 32 rows: 2 column groups x 2 k-slices, 48 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 512u) run_candidate<4, 2, 4, 32>("4 waves, 32 rows: 2 column groups x 2 k-slices, 64 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 1024u) run_candidate<3, 2, 4, 64>("4 waves, 64 rows: 2 column groups x 2 k-slices, 48 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  if (mask & 128u) run_candidate<4, 4, 8, 64>("64 rows: 4 column groups x 2 k-slices, 64 columns each", k, n, m, dqw, dsc, dqz, dact, dout, y_ref, stream);
  return 0;
}
