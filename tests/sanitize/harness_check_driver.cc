// Host-only checks of the harness's acceptance rules and shape derivation (xbitops_amd/csrc/xbit_harness_check.h),
// built by tests/test_sanitizers.py with -fsanitize=address,undefined.  Exit code 0 = every check held.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../xbitops_amd/csrc/xbit_harness_check.h"

using namespace xbit_harness;

static int failures = 0;
#define EXPECT(cond)                                                 \
  do {                                                               \
    if (!(cond)) {                                                   \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond);     \
      ++failures;                                                    \
    }                                                                \
  } while (0)

int main() {
  // fp16 decoding: every bit pattern, against the definition
  for (uint32_t h = 0; h < 65536; ++h) {
    const float f = half_to_float((uint16_t)h);
    const int e = (h >> 10) & 31, m = h & 1023;
    const double mag = e == 0 ? ldexp((double)m, -24) : ldexp(1.0 + m / 1024.0, e - 15);
    if (e == 31) {
      EXPECT(m ? std::isnan(f) : (std::isinf(f) && ((h >> 15) ? f < 0 : f > 0)));
    } else {
      EXPECT((double)f == ((h >> 15) ? -mag : mag));
    }
  }
  // the reference's GEMV rule on finite values (gemv_w4a16.cu:406-407) ...
  EXPECT(!gemv_mismatch(1.0f, 1.0005f));          // inside 1.001
  EXPECT(!gemv_mismatch(1e-4f, 3e-4f));           // far apart relatively, but within 4e-4 absolute
  EXPECT(gemv_mismatch(1.0f, 1.002f));
  EXPECT(gemv_mismatch(0.0f, 1e-3f));             // ref / 0 = inf > 1.001 and 1e-3 > 4e-4
  EXPECT(!gemv_mismatch(0.0f, 0.0f));
  // ... and what it misses: NaN / Inf on one side only
  const float nan = std::nanf(""), inf = INFINITY;
  EXPECT(gemv_mismatch(nan, 1.0f));
  EXPECT(gemv_mismatch(1.0f, nan));
  EXPECT(!gemv_mismatch(nan, nan));
  EXPECT(gemv_mismatch(inf, -inf));
  EXPECT(!gemv_mismatch(inf, inf));
  EXPECT(gemv_mismatch(inf, 65504.0f));
  EXPECT(dequant_mismatch(nan, 0.0f));
  EXPECT(dequant_mismatch(0.0f, 2e-3f));
  EXPECT(!dequant_mismatch(0.5f, 0.5009f));
  // counting over buffers that end exactly at the allocation (the sanitizer sees any over-read)
  {
    std::vector<uint16_t> y(1000, 0x3C00), g(1000, 0x3C00);
    y[7] = 0x7E00;      // NaN where the golden value is 1.0
    g[9] = 0x4000;      // 2.0 where the output is 1.0
    EXPECT(count_gemv_errors(y.data(), g.data(), y.size(), 0) == 2);
    EXPECT(count_dequant_errors(y.data(), g.data(), y.size(), 0) == 2);
    EXPECT(count_gemv_errors(y.data(), g.data(), 0, 0) == 0);
  }
  // shape derivation from byte sizes
  {
    uint64_t m = 0, k = 0, n = 0;
    EXPECT(derive_shape(1 * 4096 * 2, 4096ll * 11008 * 2, 1 * 11008 * 2, &m, &k, &n) && m == 1 && k == 4096 && n == 11008);
    EXPECT(derive_shape(16 * 7168 * 2, 7168ll * 4096 * 2, 16 * 4096 * 2, &m, &k, &n) && m == 16 && k == 7168 && n == 4096);
    EXPECT(!derive_shape(0, 10, 10, &m, &k, &n));
    EXPECT(!derive_shape(-4, 10, 10, &m, &k, &n));
    EXPECT(!derive_shape(2 * 100, 2 * 101 * 7, 2 * 7, &m, &k, &n));  // sizes that no (M, K, N) explains
  }
  if (failures) fprintf(stderr, "%d check(s) failed\n", failures);
  return failures ? 1 : 0;
}
