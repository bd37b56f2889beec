// xbit_gemv_rows.hip — bit-width dispatch for the shared-activation-tile GEMV (33..256 rows).  The kernel template lives in xbit_gemv_rows.h and is
// instantiated once per bit-width in xbit_gemv_rows_b{3..8}.hip so that the widths compile in parallel.
#include "xbit_host.h"

namespace xbit {

int launch_gemv_rows_b3(const GemvParams& p, hipStream_t stream, int cpl, int ncg);
int launch_gemv_rows_b4(const GemvParams& p, hipStream_t stream, int cpl, int ncg);
int launch_gemv_rows_b5(const GemvParams& p, hipStream_t stream, int cpl, int ncg);
int launch_gemv_rows_b6(const GemvParams& p, hipStream_t stream, int cpl, int ncg);
int launch_gemv_rows_b7(const GemvParams& p, hipStream_t stream, int cpl, int ncg);
int launch_gemv_rows_b8(const GemvParams& p, hipStream_t stream, int cpl, int ncg);

int launch_gemv_rows(const GemvParams& p, int bits, hipStream_t stream, int cpl, int ncg) {
  switch (bits) {
    case 2: return kGemvDynNotEligible;  // two k-blocks per lane and unit: no 32-row build fits into 256 registers
    case 3: return launch_gemv_rows_b3(p, stream, cpl, ncg);
    case 4: return launch_gemv_rows_b4(p, stream, cpl, ncg);
    case 5: return launch_gemv_rows_b5(p, stream, cpl, ncg);
    case 6: return launch_gemv_rows_b6(p, stream, cpl, ncg);
    case 7: return launch_gemv_rows_b7(p, stream, cpl, ncg);
    case 8: return launch_gemv_rows_b8(p, stream, cpl, ncg);
    default: return XBIT_ERR_BITS;
  }
}

// 33..256 rows in ONE launch, blocks of 32 rows in grid.y.  The kernel above first, then the staged-operand kernel's 32-row build (tiles of 16 * dyn_cpl
// columns) where the layer is not for it - except on few tiles (fewer than 192 of 32 columns) at up to 128 rows, where it is the other way round: the 32-row
// build's workgroups own 32-64 columns, and a layer that narrow does not fill the chip with workgroups of 96-256
// (profiles/r04/sweep_rows_shared_tile_v1.txt: 4096x4096 at 64 rows 9.4 us against 16.8, 11008x4096 21.9 against 37.4; at 96 / 128 rows the two tie; from
// 192 rows the shared tile wins there too: 17.9 us against 24.1, sweep_rows_192_256.txt).
int launch_gemv_rows_or_dyn(const GemvParams& p, int bits, hipStream_t stream, int dyn_cpl) {
  const bool dyn_first = (p.n + 31u) / 32u < 192u && p.m <= 128u;
  int st = kGemvDynNotEligible;
  if (!dyn_first) st = launch_gemv_rows(p, bits, stream);
  if (st == kGemvDynNotEligible) st = launch_gemv_dyn(p, bits, stream, dyn_cpl, 16, true, false, nullptr);
  if (st == kGemvDynNotEligible && dyn_first) st = launch_gemv_rows(p, bits, stream);
  return st;
}

}  // namespace xbit
