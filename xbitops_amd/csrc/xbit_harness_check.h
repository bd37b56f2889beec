// xbit_harness_check.h — host-only pieces of the harness (no HIP): fp16 decoding, the reference's acceptance rules and
// the derivation of (M, K, N) from a data set's byte sizes.  Kept apart from xbit_harness.cc so that they build and run
// without a GPU - tests/test_sanitizers.py compiles them with -fsanitize=address,undefined.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <cmath>

namespace xbit_harness {

inline float half_to_float(uint16_t h) {
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

// Dequant: the reference counts |v1 - v3| > 1e-3 (unpack_weight_2_to_7.cu:570).  Written so that a NaN on either side
// counts as an error (the reference's comparison is false for NaN).
inline bool dequant_mismatch(float v1, float gold) { return !(std::fabs(v1 - gold) <= 1e-3f); }

// GEMV: the reference counts (|ref / v1| > 1.001 || |v1 / ref| > 1.001) && |ref - v1| > 4e-4 (gemv_w4a16.cu:406-407).
// As written that is false whenever a NaN is involved - an all-NaN output would pass - so a value that is not finite on
// exactly one side, or infinities of different sign, count as errors here; two NaNs or equal infinities agree.
inline bool gemv_mismatch(float v1, float ref) {
  const bool f1 = std::isfinite(v1), fr = std::isfinite(ref);
  if (!f1 || !fr) {
    if (std::isnan(v1) && std::isnan(ref)) return false;
    return !(v1 == ref);
  }
  return (std::fabs(ref / v1) > 1.001f || std::fabs(v1 / ref) > 1.001f) && std::fabs(ref - v1) > 4e-4f;
}

inline int count_dequant_errors(const uint16_t* w, const uint16_t* gold, size_t count, int print_first) {
  int errors = 0;
  for (size_t i = 0; i < count; ++i) {
    const float v1 = half_to_float(w[i]), v3 = half_to_float(gold[i]);
    if (dequant_mismatch(v1, v3) && ++errors <= print_first) printf("dequant %zu: %f %f\n", i, v1, v3);
  }
  return errors;
}

inline int count_gemv_errors(const uint16_t* y, const uint16_t* gold, size_t count, int print_first) {
  int errors = 0;
  for (size_t i = 0; i < count; ++i) {
    const float v1 = half_to_float(y[i]), ref = half_to_float(gold[i]);
    if (gemv_mismatch(v1, ref) && ++errors <= print_first) printf("gemv %zu: r%f %f\n", i, ref, v1);
  }
  return errors;
}

// (M, K, N) from the byte sizes of input [M, K], f16weight [K, N] and out [M, N] (fp16 each); false if they do not fit.
inline bool derive_shape(int64_t input_bytes, int64_t weight_bytes, int64_t out_bytes, uint64_t* m, uint64_t* k, uint64_t* n) {
  const double mk = input_bytes / 2.0, kn = weight_bytes / 2.0, mn = out_bytes / 2.0;
  if (mk <= 0 || kn <= 0 || mn <= 0) return false;
  uint64_t mm = (uint64_t)llround(std::sqrt(mk * mn / kn));
  if (mm == 0) mm = 1;
  *m = mm;
  *k = (uint64_t)llround(mk / (double)mm);
  *n = (uint64_t)llround(mn / (double)mm);
  return *k > 0 && *n > 0 && (double)*m * (double)*k == mk && (double)*k * (double)*n == kn && (double)*m * (double)*n == mn;
}

}  // namespace xbit_harness
