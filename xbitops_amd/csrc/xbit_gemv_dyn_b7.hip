// xbit_gemv_dyn_b7.hip — 7-bit instantiation of the staged-operand one-row GEMV (see xbit_gemv_dyn.h).
#include "xbit_gemv_dyn.h"

namespace xbit {

int launch_gemv_dyn_b7(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols) {
  return dyn::launch_dyn_bits<7>(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
}

int launch_gemv_dyn_multi_b7(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl) {
  return dyn::launch_dyn_bits<7>(p, stream, cpl, 16, true, false, nullptr, 1, nullptr, nullptr, 0, &sets);
}

}  // namespace xbit
