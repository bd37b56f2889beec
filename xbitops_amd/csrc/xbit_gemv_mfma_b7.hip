// xbit_gemv_mfma_b7.hip — 7-bit instantiation of the MFMA GEMV (see xbit_gemv_mfma.h).
#include "xbit_gemv_mfma.h"

namespace xbit {

int launch_gemv_mfma_b7(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) { return launch_bits<7>(p, stream, opts); }

}  // namespace xbit
