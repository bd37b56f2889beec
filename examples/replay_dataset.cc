// replay_dataset.cc — C++ host program over the C ABI (include/xbitops_hip.h), no torch.
//
// Counterpart of the reference's stand-alone harness executable (/root/reference/src/main.cc:14-118,
// QbitGemv in src/cu/gemv.cuh:22): load the seven-file data set of one quantised layer
//     shape.bin "K,N,bits,group" | input.bin fp16[1,K] | qweight.bin i32 | scales.bin fp16 | qzeros.bin i32
//     | weight.bin fp16[K,N] (golden dequant) | out.bin fp16[1,N] (golden GEMV)
// run xbit_dequant_f16 and xbit_gemv_f16 on the GPU, apply the reference's own acceptance rules
// (dequant |diff| <= 1e-3, unpack_weight_2_to_7.cu:570; GEMV error iff relative > 1e-3 AND absolute >
// 4e-4, gemv_w4a16.cu:406-407) and time the GEMV with the library's per-launch events.
//
//   hipcc -O2 -I include examples/replay_dataset.cc -o examples/replay_dataset \
//         -L xbitops_amd/lib -lxbitops_hip -Wl,-rpath,'$ORIGIN/../xbitops_amd/lib'
//   examples/replay_dataset <directory> [add_zero_bias=0] [iters=50]
// Exit status: 0 = both checks pass, 1 = mismatch, 2 = usage / IO / HIP error.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <string>
#include <vector>

#include "xbitops_hip.h"

namespace {

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {
      int e = -1;
      uint32_t m = man;
      do {
        ++e;
        m <<= 1;
      } while (!(m & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3FFu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

template <typename T>
bool read_file(const std::string& path, size_t count, std::vector<T>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path.c_str());
    return false;
  }
  out->resize(count);
  const size_t got = fread(out->data(), sizeof(T), count, f);
  fclose(f);
  if (got != count) {
    fprintf(stderr, "%s: expected %zu elements, read %zu\n", path.c_str(), count, got);
    return false;
  }
  return true;
}

#define HIP_OK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                      \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

template <typename T>
hipError_t to_device(const std::vector<T>& host, void** dev) {
  hipError_t e = hipMalloc(dev, host.size() * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <directory> [add_zero_bias=0] [iters=50]\n", argv[0]);
    return 2;
  }
  const std::string dir = std::string(argv[1]) + "/";
  const int bias = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 50;

  // shape.bin: four integers with arbitrary single-character separators (main.cc:18-19)
  long dims[4] = {0, 0, 0, 0};
  {
    FILE* f = fopen((dir + "shape.bin").c_str(), "r");
    if (!f) {
      fprintf(stderr, "cannot open %sshape.bin\n", dir.c_str());
      return 2;
    }
    char text[256] = {0};
    const size_t len = fread(text, 1, sizeof(text) - 1, f);
    fclose(f);
    int found = 0;
    for (size_t i = 0; i < len && found < 4;) {
      if (text[i] >= '0' && text[i] <= '9') {
        char* end = nullptr;
        dims[found++] = strtol(text + i, &end, 10);
        i = (size_t)(end - text);
      } else {
        ++i;
      }
    }
    if (found < 4) {
      fprintf(stderr, "shape.bin must hold K,N,bits,group\n");
      return 2;
    }
  }
  const uint32_t k = (uint32_t)dims[0], n = (uint32_t)dims[1];
  const int bits = (int)dims[2], group = (int)dims[3];
  if (bits < 2 || bits > 8 || group < 1 || k == 0 || n == 0) {
    fprintf(stderr, "shape.bin: unsupported K=%u N=%u bits=%d group=%d\n", k, n, bits, group);
    return 2;
  }
  const size_t kw = ((size_t)k * bits + 31) / 32, groups = ((size_t)k + group - 1) / group, zw = ((size_t)n * bits + 31) / 32;

  std::vector<uint16_t> input, scales, weight, out_gold;
  std::vector<uint32_t> qweight, qzeros;
  if (!read_file(dir + "input.bin", k, &input) || !read_file(dir + "qweight.bin", kw * n, &qweight) ||
      !read_file(dir + "scales.bin", groups * n, &scales) || !read_file(dir + "qzeros.bin", groups * zw, &qzeros) ||
      !read_file(dir + "weight.bin", (size_t)k * n, &weight) || !read_file(dir + "out.bin", n, &out_gold))
    return 2;

  void *d_in = nullptr, *d_qw = nullptr, *d_sc = nullptr, *d_qz = nullptr, *d_w = nullptr, *d_y = nullptr;
  HIP_OK(to_device(input, &d_in));
  HIP_OK(to_device(qweight, &d_qw));
  HIP_OK(to_device(scales, &d_sc));
  HIP_OK(to_device(qzeros, &d_qz));
  HIP_OK(hipMalloc(&d_w, (size_t)k * n * 2));
  HIP_OK(hipMalloc(&d_y, (size_t)n * 2));

  int st = xbit_dequant_f16(d_w, d_qw, d_sc, d_qz, bits, group, k, n, (uint8_t)bias, nullptr);
  if (st != 0) {
    fprintf(stderr, "xbit_dequant_f16: %s\n", xbit_status_string(st));
    return 2;
  }
  st = xbit_gemv_f16(d_y, d_in, d_qw, d_sc, d_qz, bits, group, 1, k, n, (uint8_t)bias, nullptr);
  if (st != 0) {
    fprintf(stderr, "xbit_gemv_f16: %s\n", xbit_status_string(st));
    return 2;
  }
  HIP_OK(hipDeviceSynchronize());
  std::vector<uint16_t> w((size_t)k * n), y(n);
  HIP_OK(hipMemcpy(w.data(), d_w, w.size() * 2, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(y.data(), d_y, y.size() * 2, hipMemcpyDeviceToHost));

  size_t dq_bad = 0, gemv_bad = 0;
  double dq_max = 0.0, gemv_max = 0.0;
  for (size_t i = 0; i < w.size(); ++i) {
    const double d = std::fabs((double)half_to_float(w[i]) - (double)half_to_float(weight[i]));
    if (d > dq_max) dq_max = d;
    if (!(d <= 1e-3)) ++dq_bad;
  }
  for (size_t i = 0; i < y.size(); ++i) {
    const double g = half_to_float(out_gold[i]), d = std::fabs((double)half_to_float(y[i]) - g);
    if (d > gemv_max) gemv_max = d;
    // gemv_w4a16.cu:406-407: (abs(ref / v1) > 1.001 || abs(v1 / ref) > 1.001) && abs(ref - v1) > 4e-4
    const double v1 = half_to_float(y[i]);
    if ((std::fabs(g / v1) > 1.001 || std::fabs(v1 / g) > 1.001) && d > 4e-4) ++gemv_bad;
  }

  // kernel time by the library's own per-launch events
  double us = 0.0;
  if (xbit_profile_enable(1) == 0) {
    for (int i = 0; i < iters; ++i) {
      st = xbit_gemv_f16(d_y, d_in, d_qw, d_sc, d_qz, bits, group, 1, k, n, (uint8_t)bias, nullptr);
      float ms = 0.f;
      if (st != 0 || xbit_profile_last_ms(&ms) != 0) {
        fprintf(stderr, "timed launch failed: %s\n", xbit_status_string(st));
        return 2;
      }
      us += ms * 1e3;
    }
    xbit_profile_enable(0);
    us /= iters > 0 ? iters : 1;
  }
  const double bytes = 4.0 * kw * n + 2.0 * groups * n + 4.0 * groups * zw + 2.0 * k + 2.0 * n;
  printf("K=%u N=%u bits=%d group=%d add_zero_bias=%d\n", k, n, bits, group, bias);
  printf("dequant: %zu of %zu elements off by more than 1e-3 (max |diff| %.3e)\n", dq_bad, w.size(), dq_max);
  printf("gemv:    %zu of %zu outputs fail rel 1e-3 AND abs 4e-4 (max |diff| %.3e)\n", gemv_bad, y.size(), gemv_max);
  if (us > 0) printf("gemv kernel: %.2f us (cache-warm, %d launches) = %.0f GB/s algorithmic\n", us, iters, bytes / us * 1e-3);
  (void)hipFree(d_in); (void)hipFree(d_qw); (void)hipFree(d_sc); (void)hipFree(d_qz); (void)hipFree(d_w); (void)hipFree(d_y);
  return (dq_bad == 0 && gemv_bad == 0) ? 0 : 1;
}
