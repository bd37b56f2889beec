// xbit_gemv_mfma_b4.hip — 4-bit instantiation of the MFMA GEMV (see xbit_gemv_mfma.h).
#include "xbit_gemv_mfma.h"

namespace xbit {

int launch_gemv_mfma_b4(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) { return launch_bits<4>(p, stream, opts); }

}  // namespace xbit
