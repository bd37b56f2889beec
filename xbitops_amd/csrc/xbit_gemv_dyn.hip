// xbit_gemv_dyn.hip — bit-width dispatch for the staged-operand one-row GEMV.  The kernel template lives in
// xbit_gemv_dyn.h and is instantiated once per bit-width in xbit_gemv_dyn_b{2..8}.hip so that the widths compile in parallel.
#include "xbit_host.h"

namespace xbit {

int launch_gemv_dyn_b2(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);
int launch_gemv_dyn_b3(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);
int launch_gemv_dyn_b4(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);
int launch_gemv_dyn_b5(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);
int launch_gemv_dyn_b6(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);
int launch_gemv_dyn_b7(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);
int launch_gemv_dyn_b8(const GemvParams& p, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                       unsigned long long* acc, const int32_t* g_idx, uint32_t tcols);

int launch_gemv_dyn_multi_b2(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);
int launch_gemv_dyn_multi_b3(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);
int launch_gemv_dyn_multi_b4(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);
int launch_gemv_dyn_multi_b5(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);
int launch_gemv_dyn_multi_b6(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);
int launch_gemv_dyn_multi_b7(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);
int launch_gemv_dyn_multi_b8(const GemvParams& p, const GemvSets& sets, hipStream_t stream, int cpl);

int launch_gemv_dyn(const GemvParams& p, int bits, hipStream_t stream, int cpl, int nl, bool nt, bool stamped, uint64_t* trace, uint32_t splits,
                    unsigned long long* acc, const int32_t* g_idx, uint32_t tcols) {
  switch (bits) {
    case 2: return launch_gemv_dyn_b2(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    case 3: return launch_gemv_dyn_b3(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    case 4: return launch_gemv_dyn_b4(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    case 5: return launch_gemv_dyn_b5(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    case 6: return launch_gemv_dyn_b6(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    case 7: return launch_gemv_dyn_b7(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    case 8: return launch_gemv_dyn_b8(p, stream, cpl, nl, nt, stamped, trace, splits, acc, g_idx, tcols);
    default: return XBIT_ERR_BITS;
  }
}

int launch_gemv_dyn_multi(const GemvParams& p, const GemvSets& sets, int bits, hipStream_t stream, int cpl) {
  switch (bits) {
    case 2: return launch_gemv_dyn_multi_b2(p, sets, stream, cpl);
    case 3: return launch_gemv_dyn_multi_b3(p, sets, stream, cpl);
    case 4: return launch_gemv_dyn_multi_b4(p, sets, stream, cpl);
    case 5: return launch_gemv_dyn_multi_b5(p, sets, stream, cpl);
    case 6: return launch_gemv_dyn_multi_b6(p, sets, stream, cpl);
    case 7: return launch_gemv_dyn_multi_b7(p, sets, stream, cpl);
    case 8: return launch_gemv_dyn_multi_b8(p, sets, stream, cpl);
    default: return XBIT_ERR_BITS;
  }
}

}  // namespace xbit
