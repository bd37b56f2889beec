/*
 * xbit_oracle.c — CPU restatement of the XbitOps dequant / A16Wx GEMV path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under xbitops_amd/ (the product) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py do, and only as the checker / the CPU timing
 * reference.  The product path is the HIP library behind include/xbitops_hip.h.
 *
 * Parity status: PINNED.  In XO_SIM_HALF_UP mode every function here is checked
 * bit-for-bit against the reference's own CPU simulator
 * (/root/reference/src/cpp_simulate.cc compiled in place by oracle/Makefile into
 * oracle/_ref/) and against the committed fixtures in tests/golden/ that were
 * produced from it (oracle/gen_golden.py).  XO_RNE mode is the same algorithm with
 * IEEE round-to-nearest-even fp16 arithmetic, i.e. what the reference's CUDA
 * kernels compute on hardware (`__hmul2`, `__hfma2`, `__ushort2half_rn`).
 *
 * Algorithm restated (all citations are paths under /root/reference):
 *   packed layout ........ src/cu/unpack_weight_2_to_7.cu:56-59,64,72 (2/4/8 bit),
 *                          :196-217,228-239,258-281 (3/5/6/7 bit, word straddle),
 *                          src/cu/gemv_w4a16_pt.cu:53-58,73,90-94 (4-bit GEMV view)
 *   W = q*s - (z+bias)*s .. src/cu/unpack_weight_2_to_7.cu:60,75 and :285,305
 *                          (sz = hmul2(half(z+bias), s); w = hfma2(half(q), s, -sz))
 *   y = a . W ............. src/cu/gemv_w4a16_pt.cu:116-124,139-143 (the oracle
 *                          accumulates in double; see DESIGN.md "numerics")
 *   simulator rounding .... src/cpp_simulate.cc:43-59 (float_to_half adds 0x1000 and
 *                          truncates => ties round up; saturates to 0x7FFF),
 *                          :226-243 (__hfma2/__hmul2 evaluate in fp32, then round)
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { XO_RNE = 0, XO_SIM_HALF_UP = 1 };

/* ------------------------------------------------------------------ fp16 <-> fp32/fp64 */

static inline uint32_t f32_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float bits_f32(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* exact fp16 -> fp32 (normals, subnormals, inf, nan) */
static inline float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  if (e == 0) {
    if (m == 0) return bits_f32(sign);
    float v = (float)m * 5.9604644775390625e-08f; /* m * 2^-24, exact */
    return sign ? -v : v;
  }
  if (e == 31) return bits_f32(sign | 0x7F800000u | (m << 13));
  return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

/* exact double -> fp16, IEEE round-to-nearest-even, single rounding */
static inline uint16_t d2h_rne(double d) {
  uint16_t sign = signbit(d) ? 0x8000u : 0;
  double a = fabs(d);
  if (isnan(d)) return (uint16_t)(sign | 0x7E00u);
  if (a >= 65520.0) return (uint16_t)(sign | 0x7C00u); /* rounds to inf (incl. inf itself) */
  if (a == 0.0) return sign;
  int e = ilogb(a); /* floor(log2 a) */
  if (e < -14) {    /* subnormal range: quantum 2^-24 */
    double q = rint(ldexp(a, 24));
    return (uint16_t)(sign | (uint16_t)q); /* q == 1024 lands on the smallest normal */
  }
  double q = rint(ldexp(a, 10 - e)); /* in [1024, 2048] */
  if (q >= 2048.0) {
    q = 1024.0;
    e += 1;
  }
  return (uint16_t)(sign | (uint16_t)((e + 15) << 10) | ((uint16_t)q - 1024u));
}

/* fp32 -> fp16 as the reference simulator does it (cpp_simulate.cc:43-59):
 * add half of the 13 dropped mantissa bits and truncate => ties go AWAY from zero
 * ("half-up"), subnormals through a shifted add, overflow saturates to 0x7FFF. */
static inline uint16_t f2h_sim(float x) {
  uint32_t b = f32_bits(x) + 0x00001000u;
  uint32_t e = (b >> 23) & 0xFFu, m = b & 0x007FFFFFu;
  uint32_t r = (b >> 16) & 0x8000u;
  if (e > 112) r |= (((e - 112) << 10) & 0x7C00u) | (m >> 13);
  if (e < 113 && e > 101) r |= (((0x007FF000u + m) >> (125 - e)) + 1) >> 1;
  if (e > 143) r |= 0x7FFFu;
  return (uint16_t)r;
}

/* fp16 -> fp32 as the simulator does it (cpp_simulate.cc:28-41): no inf/nan class,
 * exponent 31 is read as an ordinary binade.  Identical to h2f for finite inputs. */
static inline float h2f_sim(uint16_t h) {
  uint32_t e = (h >> 10) & 0x1Fu;
  if (e == 31) return bits_f32(((uint32_t)(h & 0x8000u) << 16) | ((e + 112u) << 23) | ((uint32_t)(h & 0x3FFu) << 13));
  return h2f(h);
}

static inline uint16_t int_to_half(int v, int mode) {
  return mode == XO_RNE ? d2h_rne((double)v) : f2h_sim((float)v);
}

/* hmul: fp16 * fp16 -> fp16 */
static inline uint16_t hmul(uint16_t a, uint16_t b, int mode) {
  if (mode == XO_RNE) return d2h_rne((double)h2f(a) * (double)h2f(b)); /* product exact in double */
  volatile float p = h2f_sim(a) * h2f_sim(b);
  return f2h_sim(p);
}

/* hfma(a, b, -c): fp16 fused multiply-add with the addend negated.
 * RNE: a*b is exact in double (11x11 bits) and so is the sum for the operand ranges
 *      of this path, so one d2h_rne is the single rounding of a hardware fma.
 * SIM: fp32 product, fp32 sum, then the simulator's rounding (cpp_simulate.cc:226-237). */
static inline uint16_t hfma_negc(uint16_t a, uint16_t b, uint16_t c, int mode) {
  if (mode == XO_RNE) return d2h_rne((double)h2f(a) * (double)h2f(b) + (-(double)h2f(c)));
  volatile float p = h2f_sim(a) * h2f_sim(b);
  volatile float s = p + (-1.0f) * h2f_sim(c);
  return f2h_sim(s);
}

/* ------------------------------------------------------------------ bit-stream access */

/* value `idx` of an LSB-first stream of `bits`-wide fields stored in 32-bit words that
 * are `stride` words apart (unpack_weight_2_to_7.cu:196-217 — same field, read as one
 * 64-bit window instead of the two-branch form). */
static inline uint32_t stream_get(const uint32_t* base, size_t stride, uint64_t idx, int bits, uint64_t nwords) {
  uint64_t bit = idx * (uint64_t)bits;
  uint64_t w = bit >> 5;
  unsigned sh = (unsigned)(bit & 31u);
  uint64_t lo = base[w * stride];
  uint64_t hi = (sh + (unsigned)bits > 32u && w + 1 < nwords) ? base[(w + 1) * stride] : 0;
  return (uint32_t)(((lo | (hi << 32)) >> sh) & ((1u << bits) - 1u));
}

static inline uint32_t kwords(uint32_t k, int bits) { return (uint32_t)(((uint64_t)k * bits + 31) / 32); }

/* ------------------------------------------------------------------ exported API */

/* integer weights q[k,n] (uint8) from qweight int32 [ceil(K*b/32), N]; the k-stream of a
 * column runs down the rows. */
void xo_unpack_q(uint8_t* q, const uint32_t* qweight, int bits, uint32_t k, uint32_t n) {
  const uint32_t kw = kwords(k, bits);
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col)
    for (uint32_t r = 0; r < k; ++r) q[(size_t)r * n + col] = (uint8_t)stream_get(qweight + col, n, r, bits, kw);
}

/* integer zero-points z[g,n] (uint8) from qzeros int32 [G, ceil(N*b/32)]; the n-stream of a
 * group runs along the row. */
void xo_unpack_z(uint8_t* z, const uint32_t* qzeros, int bits, uint32_t groups, uint32_t n) {
  const uint32_t zw = kwords(n, bits);
  for (uint32_t g = 0; g < groups; ++g)
    for (uint32_t col = 0; col < n; ++col) z[(size_t)g * n + col] = (uint8_t)stream_get(qzeros + (size_t)g * zw, 1, col, bits, zw);
}

/* W[k,n] (fp16 bits) = hfma(half(q), s, -hmul(half(z + bias), s)),  group = k / groupsize */
void xo_dequant(uint16_t* out, const uint32_t* qweight, const uint16_t* scales, const uint32_t* qzeros, int bits,
                int groupsize, uint32_t k, uint32_t n, int add_zero_bias, int mode) {
  const uint32_t kw = kwords(k, bits), zw = kwords(n, bits);
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col) {
    uint32_t cur_g = UINT32_MAX;
    uint16_t s = 0, zs = 0;
    for (uint32_t r = 0; r < k; ++r) {
      uint32_t g = r / (uint32_t)groupsize;
      if (g != cur_g) {
        cur_g = g;
        s = scales[(size_t)g * n + col];
        int z = (int)stream_get(qzeros + (size_t)g * zw, 1, (uint64_t)col, bits, zw) + add_zero_bias;
        zs = hmul(int_to_half(z, mode), s, mode);
      }
      uint32_t q = stream_get(qweight + col, n, r, bits, kw);
      out[(size_t)r * n + col] = hfma_negc(int_to_half((int)q, mode), s, zs, mode);
    }
  }
}

/* y[m,n] (fp16 bits) = fp16_rne( sum_k a[m,k] * W[k,n] ), W as xo_dequant(mode), sum in double.
 * This is the reference op's semantics "dequantise, then multiply"  with an
 * accumulator wide enough that the only rounding left is the final fp16 store. */
void xo_gemv(uint16_t* out, const uint16_t* a, const uint32_t* qweight, const uint16_t* scales, const uint32_t* qzeros,
             int bits, int groupsize, uint32_t m, uint32_t k, uint32_t n, int add_zero_bias, int mode) {
  const uint32_t kw = kwords(k, bits), zw = kwords(n, bits);
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col) {
    double acc[64];
    for (uint32_t m0 = 0; m0 < m; m0 += 64) {
      uint32_t mb = m - m0 < 64 ? m - m0 : 64;
      for (uint32_t i = 0; i < mb; ++i) acc[i] = 0.0;
      uint32_t cur_g = UINT32_MAX;
      uint16_t s = 0, zs = 0;
      for (uint32_t r = 0; r < k; ++r) {
        uint32_t g = r / (uint32_t)groupsize;
        if (g != cur_g) {
          cur_g = g;
          s = scales[(size_t)g * n + col];
          int z = (int)stream_get(qzeros + (size_t)g * zw, 1, (uint64_t)col, bits, zw) + add_zero_bias;
          zs = hmul(int_to_half(z, mode), s, mode);
        }
        uint32_t q = stream_get(qweight + col, n, r, bits, kw);
        double w = (double)h2f(hfma_negc(int_to_half((int)q, mode), s, zs, mode));
        for (uint32_t i = 0; i < mb; ++i) acc[i] += (double)h2f(a[(size_t)(m0 + i) * k + r]) * w;
      }
      for (uint32_t i = 0; i < mb; ++i) out[(size_t)(m0 + i) * n + col] = d2h_rne(acc[i]);
    }
  }
}

/* ---- act-order ("desc_act") checkpoints: the group of row k is g_idx[k], not k / groupsize.
 * Restates /root/reference/src/cu/quant_cuda_kernel.cu:257-336 (VecQuant4MatMulKernel_G, compiled out
 * there by `#if 0` at :220): per k it reads  g = g_idx[k]  (:293-294), the scale  scales[g*width + w]
 * (:295-296) and the zero-point nibble of (g, w) plus one (:299-300), and forms
 *     W = hfma(half(q), scale, hmul(-scale, half(z + 1)))                     (:299-300, :306-309)
 * i.e. the same single-rounded product-sum as the contiguous-group kernels with the group looked up
 * per row.  The "+ 1" is that kernel's hard-wired GPTQ convention; here it is add_zero_bias as in the
 * two registered ops.  The reference accumulates in fp16 (:306-312); the checker keeps the op
 * semantics "dequantise, then multiply" with a wide accumulator, as xo_gemv does.  Any bit-width. */
void xo_dequant_gidx(uint16_t* out, const uint32_t* qweight, const uint16_t* scales, const uint32_t* qzeros, const int32_t* g_idx,
                     int bits, uint32_t k, uint32_t n, int add_zero_bias, int mode) {
  const uint32_t kw = kwords(k, bits), zw = kwords(n, bits);
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col)
    for (uint32_t r = 0; r < k; ++r) {
      const uint32_t g = (uint32_t)g_idx[r];
      const uint16_t s = scales[(size_t)g * n + col];
      const int z = (int)stream_get(qzeros + (size_t)g * zw, 1, (uint64_t)col, bits, zw) + add_zero_bias;
      const uint16_t zs = hmul(int_to_half(z, mode), s, mode);
      const uint32_t q = stream_get(qweight + col, n, r, bits, kw);
      out[(size_t)r * n + col] = hfma_negc(int_to_half((int)q, mode), s, zs, mode);
    }
}

void xo_gemv_gidx(uint16_t* out, const uint16_t* a, const uint32_t* qweight, const uint16_t* scales, const uint32_t* qzeros,
                  const int32_t* g_idx, int bits, uint32_t m, uint32_t k, uint32_t n, int add_zero_bias, int mode) {
  const uint32_t kw = kwords(k, bits), zw = kwords(n, bits);
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col)
    for (uint32_t i = 0; i < m; ++i) {
      double acc = 0.0;
      for (uint32_t r = 0; r < k; ++r) {
        const uint32_t g = (uint32_t)g_idx[r];
        const uint16_t s = scales[(size_t)g * n + col];
        const int z = (int)stream_get(qzeros + (size_t)g * zw, 1, (uint64_t)col, bits, zw) + add_zero_bias;
        const uint16_t zs = hmul(int_to_half(z, mode), s, mode);
        const uint32_t q = stream_get(qweight + col, n, r, bits, kw);
        acc += (double)h2f(a[(size_t)i * k + r]) * (double)h2f(hfma_negc(int_to_half((int)q, mode), s, zs, mode));
      }
      out[(size_t)i * n + col] = d2h_rne(acc);
    }
}

/* ---- transposed 4-bit layout of the reference's experimental gemv_NT
 * (/root/reference/src/cu/gemv_w4a16_nt.cu:64-159): qweight_t [N, ceil(K/8)] (a column's k-stream is
 * contiguous: `inB + n*weight_rows + k/8`, :104), scales_t [N, G] (`scales + g + n*group_nums`, :86,
 * :125), qzeros_t [N/8, G] with the nibble of column n at bits 4*(n%8) (`qzeros + n/8*group_nums + g`,
 * `>> 8*(tidy%4)` and `+4` for the column pair 2*tidy, 2*tidy+1, :87-89).  W = hfma(s, q, -(z*s))
 * (:145-148; no zero-point bias in that kernel: add_zero_bias = 0 reproduces it). */
void xo_gemv_w4_transposed(uint16_t* out, const uint16_t* a, const uint32_t* qweight_t, const uint16_t* scales_t,
                           const uint32_t* qzeros_t, int groupsize, uint32_t m, uint32_t k, uint32_t n, int add_zero_bias, int mode) {
  const uint32_t weight_rows = (k + 7) / 8, groups = (k + (uint32_t)groupsize - 1) / (uint32_t)groupsize;
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col)
    for (uint32_t i = 0; i < m; ++i) {
      double acc = 0.0;
      for (uint32_t r = 0; r < k; ++r) {
        const uint32_t g = r / (uint32_t)groupsize;
        const uint16_t s = scales_t[(size_t)col * groups + g];
        const int z = (int)((qzeros_t[(size_t)(col / 8) * groups + g] >> (4 * (col % 8))) & 0xFu) + add_zero_bias;
        const uint16_t zs = hmul(int_to_half(z, mode), s, mode);
        const uint32_t q = (qweight_t[(size_t)col * weight_rows + r / 8] >> (4 * (r % 8))) & 0xFu;
        acc += (double)h2f(a[(size_t)i * k + r]) * (double)h2f(hfma_negc(int_to_half((int)q, mode), s, zs, mode));
      }
      out[(size_t)i * n + col] = d2h_rne(acc);
    }
}

/* y = a . W for an already dequantised fp16 W[k,n] (used to turn a reference-simulator W into
 * the reference-side GEMV result; double accumulation, RNE store). */
void xo_matvec_f16(uint16_t* out, const uint16_t* a, const uint16_t* w, uint32_t m, uint32_t k, uint32_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col)
    for (uint32_t i = 0; i < m; ++i) {
      double acc = 0.0;
      for (uint32_t r = 0; r < k; ++r) acc += (double)h2f(a[(size_t)i * k + r]) * (double)h2f(w[(size_t)r * n + col]);
      out[(size_t)i * n + col] = d2h_rne(acc);
    }
}

/* Model of the rounding ORDER of the reference's CUDA kernel (gemv_w4a16_pt.cu:67-143), 4-bit,
 * with intended (uniform) add_zero_bias handling: per 8-k chunk the products
 * are accumulated in fp16 with hfma2 — lanes (k even, k odd) kept separately — the two fp16
 * partials are widened and added into an fp32 sum, chunks are walked per K-slice of `block_k`
 * rows and the 32 slices are then added pairwise as the shuffle tree does.  Brackets where the
 * CUDA output lands relative to xo_gemv; not used as a pass/fail oracle. */
void xo_gemv_w4_cuda_order(uint16_t* out, const uint16_t* a, const uint32_t* qweight, const uint16_t* scales,
                           const uint32_t* qzeros, int groupsize, uint32_t m, uint32_t k, uint32_t n, int add_zero_bias) {
  const int bits = 4, mode = XO_RNE;
  const uint32_t kw = kwords(k, bits), zw = kwords(n, bits);
  const uint32_t block_k = ((k + 31) / 32 + 7) / 8 * 8; /* gemv_w4a16_pt.cu:40 */
  const uint32_t slices = (k + block_k - 1) / block_k;
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < (int64_t)n; ++col)
    for (uint32_t i = 0; i < m; ++i) {
      float part[64];
      for (uint32_t sl = 0; sl < 64; ++sl) part[sl] = 0.f;
      for (uint32_t sl = 0; sl < slices && sl < 64; ++sl) {
        float sum = 0.f;
        for (uint32_t c0 = sl * block_k; c0 < (sl + 1) * block_k && c0 < k; c0 += 8) {
          uint32_t g = c0 / (uint32_t)groupsize;
          uint16_t s = scales[(size_t)g * n + col];
          int z = (int)stream_get(qzeros + (size_t)g * zw, 1, (uint64_t)col, bits, zw) + add_zero_bias;
          uint16_t zs = hmul(int_to_half(z, mode), s, mode);
          uint16_t r_even = 0, r_odd = 0; /* half2 accumulator .x / .y */
          for (uint32_t j = 0; j < 8 && c0 + j < k; ++j) {
            uint32_t q = stream_get(qweight + col, n, c0 + j, bits, kw);
            uint16_t w = hfma_negc(int_to_half((int)q, mode), s, zs, mode);
            uint16_t av = a[(size_t)i * k + c0 + j];
            uint16_t* r = (j & 1) ? &r_odd : &r_even;
            *r = d2h_rne((double)h2f(av) * (double)h2f(w) + (double)h2f(*r));
          }
          volatile float t = h2f(r_even) + h2f(r_odd);
          volatile float u = sum + t;
          sum = u;
        }
        part[sl] = sum;
      }
      for (uint32_t d = 16; d >= 1; d >>= 1) /* warpReduceSum<32>, gemv_w4a16_pt.cu:20-33 */
        for (uint32_t sl = 0; sl < d; ++sl) {
          volatile float t = part[sl] + part[sl + d];
          part[sl] = t;
        }
      out[(size_t)i * n + col] = d2h_rne((double)part[0]);
    }
}

/* scalar helpers exported for unit tests of the soft-float itself */
uint16_t xo_f2h_sim(float x) { return f2h_sim(x); }
uint16_t xo_d2h_rne(double x) { return d2h_rne(x); }
float xo_h2f(uint16_t h) { return h2f(h); }
uint16_t xo_hmul(uint16_t a, uint16_t b, int mode) { return hmul(a, b, mode); }
uint16_t xo_hfma_negc(uint16_t a, uint16_t b, uint16_t c, int mode) { return hfma_negc(a, b, c, mode); }

int xo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void xo_set_num_threads(int t) {
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}
