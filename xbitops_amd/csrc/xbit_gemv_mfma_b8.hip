// xbit_gemv_mfma_b8.hip — 8-bit instantiation of the MFMA GEMV (see xbit_gemv_mfma.h).
#include "xbit_gemv_mfma.h"

namespace xbit {

int launch_gemv_mfma_b8(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) { return launch_bits<8>(p, stream, opts); }

}  // namespace xbit
