// xbit_gemv_mfma.hip — bit-width dispatch for the MFMA GEMV.  The kernel template lives in
// xbit_gemv_mfma.h and is instantiated once per bit-width in xbit_gemv_mfma_b{2..8}.hip so that the
// seven widths compile in parallel.
#include "xbit_host.h"

namespace xbit {

int launch_gemv_mfma_b2(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);
int launch_gemv_mfma_b3(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);
int launch_gemv_mfma_b4(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);
int launch_gemv_mfma_b5(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);
int launch_gemv_mfma_b6(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);
int launch_gemv_mfma_b7(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);
int launch_gemv_mfma_b8(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts);


int launch_gemv_mfma(const GemvParams& p, int bits, hipStream_t stream, const GemvLaunchOpts& opts) {
  switch (bits) {
    case 2: return launch_gemv_mfma_b2(p, stream, opts);
    case 3: return launch_gemv_mfma_b3(p, stream, opts);
    case 4: return launch_gemv_mfma_b4(p, stream, opts);
    case 5: return launch_gemv_mfma_b5(p, stream, opts);
    case 6: return launch_gemv_mfma_b6(p, stream, opts);
    case 7: return launch_gemv_mfma_b7(p, stream, opts);
    case 8: return launch_gemv_mfma_b8(p, stream, opts);
    default: return XBIT_ERR_BITS;
  }
}

}  // namespace xbit
