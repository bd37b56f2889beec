// xbit_gemv_rows_b2.hip — 2-bit instantiation of the shared-activation-tile GEMV for 33..512 rows (see xbit_gemv_rows.h).
#include "xbit_gemv_rows.h"

namespace xbit {

int launch_gemv_rows_b2(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt) { return rows::launch_rows_bits<2>(p, stream, cpl, ncg, mt); }

}  // namespace xbit
