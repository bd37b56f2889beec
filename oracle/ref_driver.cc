// ref_driver.cc — C-ABI shim around the REFERENCE's own CPU simulator.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no reference code: it
// `#include`s /root/reference/src/cpp_simulate.cc where it lies (read-only mount, never copied
// into this repo) and exposes the reference's templated dequant simulator
//   cpu::DequantizeAndUnpackWeight3567_v2<ushort, WBITS>   (cpp_simulate.cc:568-691)
//   cpu::DequantizeAndUnpackWeight248<ushort, WBITS>       (cpp_simulate.cc:373-430)
// for every bit-width through one extern "C" entry point.  oracle/Makefile builds it into
// oracle/_ref/libxbit_ref.so (git-ignored; travels to the GPU box as a binary only).
//
// Driver obligations (SURVEY.md §8c "Driver requirements"):
//   * MATRIX_M/K/N are the globals the simulator's grid loops read (cpp_simulate.cc:8-10,570);
//   * the simulator walks whole 256-thread blocks and reads scale/zero rows `from..to`
//     inclusive (cpp_simulate.cc:597-608), i.e. one group row past the end and, for threads
//     past the last column pair, further still => inputs are copied into padded buffers;
//   * outputs of padding threads are never written (remains <= 0), so `out` needs no slack.
#include <cstdint>
#include <cstring>
#include <vector>

int MATRIX_M = 1;
int MATRIX_K = 0;
int MATRIX_N = 0;

#include "/root/reference/src/cpp_simulate.cc"

namespace {

template <int B>
void run_v2(uint16_t* out, const uint32_t* qw, const uint16_t* sc, const uint32_t* qz, int g, int k, int n) {
  cpu::DequantizeAndUnpackWeight3567_v2<cpu::ushort, B>(out, qw, sc, qz, g, k, n);
}
template <int B>
void run_248(uint16_t* out, uint32_t* qw, uint16_t* sc, uint32_t* qz, int g, int k, int n) {
  cpu::DequantizeAndUnpackWeight248<cpu::ushort, B>(out, qw, sc, qz, g, k, n);
}

}  // namespace

extern "C" {

// Reference dequant (simulator, no add_zero_bias parameter exists there).
//   variant 0: ...3567_v2<ushort,bits> (valid for bits 2..8)
//   variant 1: ...248<ushort,bits>     (bits 2 and 4 only; the 8-bit branch is broken upstream)
// Returns 0, or -1 for an unsupported (variant, bits) pair.
int ref_dequant(uint16_t* out, const uint32_t* qweight, const uint16_t* scales, const uint32_t* qzeros, int bits,
                int groupsize, int k, int n, int variant) {
  if (bits < 2 || bits > 8 || k <= 0 || n <= 0 || (n & 1)) return -1;
  MATRIX_K = k;
  MATRIX_N = n;
  const size_t kw = ((size_t)k * bits + 31) / 32;
  const size_t groups = ((size_t)k + groupsize - 1) / groupsize;
  const size_t zw = ((size_t)n * bits + 31) / 32;
  // thread count the simulator will walk, rounded up to whole blocks, plus slack
  const size_t threads = ((size_t)(n / 2) * (((size_t)k + 31) / 32) + 255) / 256 * 256 + 256;
  const size_t vrows = (threads + (size_t)(n / 2) - 1) / (size_t)(n / 2) + 1;  // virtual 32-k rows touched
  const size_t grows = vrows * 32 / (size_t)groupsize + 4;                       // group rows touched
  std::vector<uint32_t> qw((kw + vrows * 8 + 8) * (size_t)n, 0u);
  std::vector<uint16_t> sc((groups + grows) * (size_t)n, 0);
  std::vector<uint32_t> qz((groups + grows) * zw + 8, 0u);
  std::memcpy(qw.data(), qweight, kw * n * sizeof(uint32_t));
  std::memcpy(sc.data(), scales, groups * n * sizeof(uint16_t));
  std::memcpy(qz.data(), qzeros, groups * zw * sizeof(uint32_t));
  if (variant == 1) {
    // the 248 simulator hard-codes its grid for 8 values/word (cpp_simulate.cc:375): 4-bit only
    if (bits != 4 || (k % 8) != 0) return -1;
    run_248<4>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n);
    return 0;
  }
  switch (bits) {
    case 2: run_v2<2>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    case 3: run_v2<3>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    case 4: run_v2<4>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    case 5: run_v2<5>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    case 6: run_v2<6>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    case 7: run_v2<7>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    case 8: run_v2<8>(out, qw.data(), sc.data(), qz.data(), groupsize, k, n); break;
    default: return -1;
  }
  return 0;
}

// The simulator's scalar soft-float, for pinning the oracle's XO_SIM_HALF_UP arithmetic.
uint16_t ref_float_to_half(float x) { return cpu::float_to_half(x); }
float ref_half_to_float(uint16_t h) { return cpu::half_to_float(h); }

}  // extern "C"
