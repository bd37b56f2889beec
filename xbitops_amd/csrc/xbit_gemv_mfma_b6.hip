// xbit_gemv_mfma_b6.hip — 6-bit instantiation of the MFMA GEMV (see xbit_gemv_mfma.h).
#include "xbit_gemv_mfma.h"

namespace xbit {

int launch_gemv_mfma_b6(const GemvParams& p, hipStream_t stream, const GemvLaunchOpts& opts) { return launch_bits<6>(p, stream, opts); }
int launch_gemv_multi_b6(const GemvMultiParams& p, hipStream_t stream) { return launch_multi_bits<6>(p, stream); }

}  // namespace xbit
