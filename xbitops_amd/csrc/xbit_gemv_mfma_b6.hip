// xbit_gemv_mfma_b6.hip — 6-bit instantiation of the MFMA GEMV (see xbit_gemv_mfma.h).
#include "xbit_gemv_mfma.h"

namespace xbit {

int launch_gemv_mfma_b6(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) { return launch_bits<6>(p, stream, opts); }

}  // namespace xbit
