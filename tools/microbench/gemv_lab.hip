// gemv_lab.hip — stand-alone bench for GEMV kernel candidates (correct results required), next to the production
// library's kernel called through the C ABI in the same process.  For every candidate: results against a CPU
// restatement of the reference arithmetic (W = fp16(fma(q, s, -fp16((z + bias) * s))), fp64 sum), then three clocks,
// all cold (every launch streams a weight copy that left the caches long ago; > 512 MB of copies, each touched first):
//   events   per-launch hipExtLaunchKernel start/stop events
//   chain    microseconds per launch of 400 back-to-back launches between two events on the stream
//   span     (stamped builds) first wave start -> last wave end, s_memrealtime
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I xbitops_amd/csrc tools/microbench/gemv_lab.hip -o tools/microbench/gemv_lab -ldl
//   tools/microbench/gemv_lab [K N [path/to/libxbitops_hip.so]]
#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "xbit_gemv_dyn.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

using xbit::dyn::DynParams;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float g_last_ms = 0.f;  // the library's own per-launch event bracket, handed back to the timing loop
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

struct Problem {
  uint32_t k, n, g, bits, kw, groups, zw, bias;
  std::vector<uint32_t> qweight, qzeros;
  std::vector<uint16_t> scales, act;
  std::vector<double> y_ref;
};

static void make_problem(Problem& p, uint32_t k, uint32_t n) {
  p.k = k; p.n = n; p.g = 128; p.bits = 4; p.bias = 1;
  p.kw = (k * 4 + 31) / 32;
  p.groups = (k + p.g - 1) / p.g;
  p.zw = (n * 4 + 31) / 32;
  p.qweight.resize((size_t)p.kw * n);
  for (auto& w : p.qweight) w = (uint32_t)rnd();
  p.qzeros.resize((size_t)p.groups * p.zw);
  for (auto& z : p.qzeros) z = (uint32_t)rnd();
  p.scales.resize((size_t)p.groups * n);
  for (auto& s : p.scales) s = h_bits((_Float16)(float)(0.002 + 0.008 * urand()));
  p.act.resize(k);
  for (auto& a : p.act) a = h_bits((_Float16)(float)(2.0 * urand() - 1.0));
  p.y_ref.assign(n, 0.0);
  for (uint32_t c = 0; c < n; ++c) {
    double sum = 0.0;
    for (uint32_t kk = 0; kk < k; ++kk) {
      const uint32_t q = (p.qweight[(size_t)(kk / 8) * n + c] >> (4 * (kk % 8))) & 15u;
      const uint32_t gi = kk / p.g;
      const uint32_t z = ((p.qzeros[(size_t)gi * p.zw + c / 8] >> (4 * (c % 8))) & 15u) + p.bias;
      const _Float16 s = h_from(p.scales[(size_t)gi * n + c]);
      const _Float16 zs = (_Float16)((float)z * (float)s);                 // hmul, one rounding
      const _Float16 w = (_Float16)((float)q * (float)s - (float)zs);      // hfma: exact in fp32, one rounding
      sum += (double)(float)w * (double)(float)h_from(p.act[kk]);
    }
    p.y_ref[c] = sum;
  }
}

struct Device {
  std::vector<uint32_t*> qw;
  std::vector<uint16_t*> sc;
  std::vector<uint32_t*> qz;
  uint16_t *act, *out;
  uint64_t* trace;
  int copies;
};

typedef int (*gemv_fn)(void*, const void*, const void*, const void*, const void*, int, int, uint32_t, uint32_t, uint32_t, uint8_t, void*);
typedef int (*prof_enable_fn)(int);
typedef int (*prof_last_fn)(float*);

struct Candidate {
  std::string name;
  int nw;       // waves per workgroup (stamped builds: size of the trace)
  bool stamped;
  // launch copy c on `stream`; start/stop may be null
  std::function<void(int c, hipStream_t s, hipEvent_t start, hipEvent_t stop)> launch;
};

template <int CPL, int NW, bool NT, int STW, int FLAGS = 0, int NL = 16, int LA = 4>
static Candidate dyn_candidate(const char* name, const Problem& pr, Device& d) {
  Candidate cd;
  cd.name = name;
  cd.nw = NW;
  cd.stamped = (FLAGS & 2) != 0;
  DynParams p;
  memset(&p, 0, sizeof(p));
  p.n = pr.n;
  p.tcols = NL * CPL;
  p.kw = pr.kw;
  constexpr int LK = 64 / NL, UB = 1;
  p.tiles = (pr.n + NL * CPL - 1) / (NL * CPL);
  p.units = ((pr.k + 31) / 32 + LK * UB - 1) / (LK * UB);
  p.k = pr.k;
  p.groups = pr.groups;
  p.zw = pr.zw;
  p.bias = pr.bias;
  p.gshift = 0;
  while ((32u * UB << p.gshift) < pr.g) ++p.gshift;
  p.act_bytes = p.units * LK * UB * 64u;
  p.m = 1;
  p.splits = 1;
  p.ups = p.units;
  p.tab_groups = ((p.units * LK - 1u) >> p.gshift) + 1u;
  if (p.tab_groups < p.groups) p.tab_groups = p.groups;
  constexpr uint32_t st = 64u * STW;  // staging threads
  constexpr int SI = (int)((512u + st - 1) / st);  // rounds: K <= 4096 and <= 32 groups in this bench
  if ((p.act_bytes / 16u + st - 1) / st > (uint32_t)SI || (p.tab_groups * NL + st - 1) / st > (uint32_t)SI) {
    fprintf(stderr, "%s: SI too small for this shape\n", name);
    exit(1);
  }
  const size_t lds = xbit::dyn::dyn_lds_bytes(p, NL, NW, CPL, 1);
  auto kern = xbit::dyn::gemv_dyn_kernel<4, UB, CPL, NL, NW, NT, SI, STW, LA, FLAGS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  cd.launch = [=, &d](int c, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    DynParams q = p;
    q.qweight = d.qw[c];
    q.scales = d.sc[c];
    q.qzeros = d.qz[c];
    q.a = d.act;
    q.out = d.out;
    q.trace = d.trace;
    hipExtLaunchKernelGGL(kern, dim3(q.tiles), dim3(64 * NW), lds, s, start, stop, 0, q.qweight, q.n, q.tiles, q.kw, q.units, q.a, q.scales, q.qzeros, q.out,
                          q.k, q.groups, q.zw, q.bias, q.m, q.tab_groups, q.gshift, q.act_bytes, q.splits, q.ups, q.acc, q.g_idx, q.trace, q.tcols, xbit::GemvSets{});
  };
  return cd;
}

static void stats(std::vector<float>& v, float& avg, float& med, float& mn) {
  std::sort(v.begin(), v.end());
  double s = 0;
  for (float x : v) s += x;
  avg = (float)(s / v.size());
  med = v[v.size() / 2];
  mn = v[0];
}

int main(int argc, char** argv) {
  const uint32_t k = argc > 2 ? (uint32_t)atoi(argv[1]) : 4096, n = argc > 2 ? (uint32_t)atoi(argv[2]) : 11008;
  const char* libpath = argc > 3 ? argv[3] : "xbitops_amd/lib/libxbitops_hip.so";
  const char* only = getenv("LAB_ONLY");
  Problem pr;
  make_problem(pr, k, n);
  const size_t wbytes = pr.qweight.size() * 4, sbytes = pr.scales.size() * 2, zbytes = pr.qzeros.size() * 4;
  const double alg_bytes = (double)wbytes + sbytes + zbytes + 2.0 * k + 2.0 * n;
  Device d;
  d.copies = (int)((512ull << 20) / (wbytes + sbytes + zbytes)) + 2;
  for (int c = 0; c < d.copies; ++c) {
    uint32_t *w, *z;
    uint16_t* s;
    CK(hipMalloc(&w, wbytes));
    CK(hipMalloc(&s, sbytes));
    CK(hipMalloc(&z, zbytes));
    CK(hipMemcpy(w, pr.qweight.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(s, pr.scales.data(), sbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(z, pr.qzeros.data(), zbytes, hipMemcpyHostToDevice));
    d.qw.push_back(w);
    d.sc.push_back(s);
    d.qz.push_back(z);
  }
  CK(hipMalloc(&d.act, k * 2));
  CK(hipMemcpy(d.act, pr.act.data(), k * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d.out, n * 2));
  CK(hipMalloc(&d.trace, (size_t)4096 * 16 * 8 * 8));
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# %ux%u w4 g128 m1: %.1f MB algorithmic per launch, %d copies\n", k, n, alg_bytes / 1e6, d.copies);

  std::vector<Candidate> cands;
  void* lib = dlopen(libpath, RTLD_NOW);
  if (lib != nullptr) {
    gemv_fn gemv = (gemv_fn)dlsym(lib, "xbit_gemv_f16");
    Candidate cd;
    cd.name = "production library (xbit_gemv_f16)";
    cd.nw = 8;
    cd.stamped = false;
    prof_enable_fn pen = (prof_enable_fn)dlsym(lib, "xbit_profile_enable");
    prof_last_fn plast = (prof_last_fn)dlsym(lib, "xbit_profile_last_ms");
    // events: the library's own per-launch bracket (xbit_profile_enable) - the same hipExtLaunchKernel events
    cd.launch = [=, &d, &pr](int c, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
      if (start != nullptr) pen(1);
      const int st = gemv(d.out, d.act, d.qw[c], d.sc[c], d.qz[c], 4, 128, 1, pr.k, pr.n, (uint8_t)pr.bias, s);
      if (st != 0) {
        fprintf(stderr, "xbit_gemv_f16 -> %d\n", st);
        exit(1);
      }
      if (start != nullptr) {
        float ms = 0;
        plast(&ms);
        pen(0);
        g_last_ms = ms;
      }
    };
    cands.push_back(cd);
  } else {
    printf("# (no production library at %s: %s)\n", libpath, dlerror());
  }
  cands.push_back(dyn_candidate<3, 8, true, 4>("48-col, waves 0-3 stage, nt", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 2>("48-col, waves 0-3 stage, nt, stamped", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 0, 16, 5>("48-col, nt, lookahead 5", pr, d));
  // round 4: what the small operands, the cross-wave sum and the arithmetic cost this kernel (ablations: results are wrong by design)
  cands.push_back(dyn_candidate<3, 8, true, 4, 4>("ablation: no small-operand loads", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 8>("ablation: no cross-wave sum", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 12>("ablation: no small loads, no sum", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 16>("ablation: no unpack / MFMA", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 28>("ablation: row loads only", pr, d));
  cands.push_back(dyn_candidate<3, 8, true, 4, 32>("48-col, v_dot2 accumulate instead of the MFMA", pr, d));
  // sixteen waves per workgroup (two units each): twice the rows in flight per CU
  cands.push_back(dyn_candidate<3, 16, true, 4>("48-col, 16 waves, waves 0-3 stage", pr, d));
  cands.push_back(dyn_candidate<3, 16, true, 8>("48-col, 16 waves, waves 0-7 stage", pr, d));
  cands.push_back(dyn_candidate<3, 16, true, 4, 0, 16, 3>("48-col, 16 waves, lookahead 3", pr, d));
  cands.push_back(dyn_candidate<3, 16, true, 4, 0, 16, 2>("48-col, 16 waves, lookahead 2", pr, d));
  if (getenv("LAB_OLD")) {
  cands.push_back(dyn_candidate<4, 8, true, 4, 0, 8>("32-col as 8 lanes x 16 B x 8 k-slots, nt", pr, d));
  cands.push_back(dyn_candidate<2, 8, true, 4, 0, 8>("16-col as 8 lanes x 8 B x 8 k-slots, nt", pr, d));
  }
  for (Candidate& cd : cands) {
    if (only != nullptr && cd.name.find(only) == std::string::npos) continue;
    const bool is_lib = cd.name.rfind("production", 0) == 0;
    // correctness
    CK(hipMemsetAsync(d.out, 0xFF, n * 2, stream));
    cd.launch(0, stream, nullptr, nullptr);
    CK(hipStreamSynchronize(stream));
    std::vector<uint16_t> y(n);
    CK(hipMemcpy(y.data(), d.out, n * 2, hipMemcpyDeviceToHost));
    double max_rel = 0;
    int bad = 0;
    for (uint32_t c = 0; c < n; ++c) {
      const double v = (double)(float)h_from(y[c]), r = pr.y_ref[c], err = fabs(v - r);
      if (!(err <= 1e-3 * fabs(r) + 6.2e-5)) ++bad;
      if (fabs(r) > 0.05) max_rel = std::max(max_rel, err / fabs(r));
    }
    // first touch of every copy, then warm-up
    for (int c = 0; c < d.copies; ++c) cd.launch(c, stream, nullptr, nullptr);
    for (int c = 0; c < d.copies; ++c) cd.launch(c, stream, nullptr, nullptr);
    CK(hipStreamSynchronize(stream));
    // events
    std::vector<float> ev;
    for (int i = 0; i < 200; ++i) {
      const int c = (i * 7 + 1) % d.copies;
      float ms = 0;
      if (is_lib) {
        cd.launch(c, stream, e0, e1);
        ms = g_last_ms;
      } else {
        cd.launch(c, stream, e0, e1);
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      ev.push_back(ms * 1e3f);
    }
    float avg, med, mn;
    stats(ev, avg, med, mn);
    // chain
    std::vector<float> ch;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipStreamSynchronize(stream));
      CK(hipEventRecord(e0, stream));
      for (int i = 0; i < 400; ++i) cd.launch((i + 3 * rep) % d.copies, stream, nullptr, nullptr);
      CK(hipEventRecord(e1, stream));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ch.push_back(ms * 1e3f / 400.f);
    }
    float cavg, cmed, cmn;
    stats(ch, cavg, cmed, cmn);
    printf("%-48s %s rel %.1e bad %d | events avg %5.2f med %5.2f min %5.2f us = %.3f of 8 TB/s | chain med %5.2f min %5.2f us/launch",
           cd.name.c_str(), bad == 0 ? "ok " : "BAD", max_rel, bad, avg, med, mn, alg_bytes / (med * 1e-6) / 8e12, cmed, cmn);
    if (cd.stamped) {
      const size_t waves = (size_t)((n + 47) / 48) * cd.nw;  // stamped builds here are 48-column
      std::vector<uint64_t> tr(waves * 8);
      std::vector<double> span, first_issue, last_spread;
      for (int rep = 0; rep < 5; ++rep) {
        cd.launch((11 * rep + 5) % d.copies, stream, nullptr, nullptr);
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(tr.data(), d.trace, waves * 64, hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull, t1 = 0;
        for (size_t w = 0; w < waves; ++w) {
          t0 = std::min(t0, tr[w * 8 + 0]);
          t1 = std::max(t1, tr[w * 8 + 5]);
        }
        span.push_back((double)(t1 - t0) * 0.01);
        // mean over workgroups of (max - mean) of the waves' last-MFMA stamps
        double acc = 0;
        const size_t wgs = waves / cd.nw;
        for (size_t g = 0; g < wgs; ++g) {
          double mx = 0, mean = 0;
          for (int w = 0; w < cd.nw; ++w) {
            const double t = (double)(tr[(g * cd.nw + w) * 8 + 3] - t0) * 0.01;
            mx = std::max(mx, t);
            mean += t / cd.nw;
          }
          acc += (mx - mean) / wgs;
        }
        last_spread.push_back(acc);
        double fi = 0;
        for (size_t w = 0; w < waves; ++w) fi += (double)(tr[w * 8 + 6] - tr[w * 8 + 0]) * 0.01 / waves;
        first_issue.push_back(fi);
      }
      std::sort(span.begin(), span.end());
      std::sort(last_spread.begin(), last_spread.end());
      std::sort(first_issue.begin(), first_issue.end());
      printf(" | span med %.2f us, waves' last MFMA max-mean per workgroup %.2f us, wave start -> first rows issued %.3f us", span[2], last_spread[2], first_issue[2]);
      // inside a chain of back-to-back launches: when does each XCD start and end (trace of the chain's last launch)
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(stream));
        for (int i = 0; i < 40; ++i) cd.launch((i + 5 * rep) % d.copies, stream, nullptr, nullptr);
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(tr.data(), d.trace, waves * 64, hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull, t1 = 0;
        double st[8] = {0}, en[8] = {0}, fe[8] = {0};
        int cnt[8] = {0};
        for (size_t w = 0; w < waves; ++w) t0 = std::min(t0, tr[w * 8 + 0]), t1 = std::max(t1, tr[w * 8 + 5]);
        for (size_t w = 0; w < waves; ++w) {
          const int x = (int)((tr[w * 8 + 7] >> 32) & 7);
          st[x] += (double)(tr[w * 8 + 0] - t0) * 0.01;
          en[x] = std::max(en[x], (double)(tr[w * 8 + 5] - t0) * 0.01);
          fe[x] += (double)(tr[w * 8 + 3] - t0) * 0.01;
          ++cnt[x];
        }
        printf("\n      in a chain (rep %d): span %.2f us; per XCC mean wave start / mean last MFMA / last wave end:", rep, (double)(t1 - t0) * 0.01);
        for (int x = 0; x < 8; ++x)
          if (cnt[x]) printf("  [%d] %.2f %.2f %.2f", x, st[x] / cnt[x], fe[x] / cnt[x], en[x]);
      }
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
