// xbit_gemv_rows.hip — bit-width dispatch for the shared-activation-tile GEMV (33..512 rows).  The kernel template lives in xbit_gemv_rows.h and is
// instantiated once per bit-width in xbit_gemv_rows_b{2..8}.hip so that the widths compile in parallel.
#include "xbit_host.h"

namespace xbit {

int launch_gemv_rows_b2(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);
int launch_gemv_rows_b3(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);
int launch_gemv_rows_b4(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);
int launch_gemv_rows_b5(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);
int launch_gemv_rows_b6(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);
int launch_gemv_rows_b7(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);
int launch_gemv_rows_b8(const GemvParams& p, hipStream_t stream, int cpl, int ncg, int mt);

int launch_gemv_rows(const GemvParams& p, int bits, hipStream_t stream, int cpl, int ncg, int mt) {
  switch (bits) {
    case 2: return launch_gemv_rows_b2(p, stream, cpl, ncg, mt);
    case 3: return launch_gemv_rows_b3(p, stream, cpl, ncg, mt);
    case 4: return launch_gemv_rows_b4(p, stream, cpl, ncg, mt);
    case 5: return launch_gemv_rows_b5(p, stream, cpl, ncg, mt);
    case 6: return launch_gemv_rows_b6(p, stream, cpl, ncg, mt);
    case 7: return launch_gemv_rows_b7(p, stream, cpl, ncg, mt);
    case 8: return launch_gemv_rows_b8(p, stream, cpl, ncg, mt);
    default: return XBIT_ERR_BITS;
  }
}

// 33..512 rows in ONE launch over blocks of rows.  The kernel above first, then the staged-operand kernel's 32-row build (tiles of 16 * dyn_cpl columns, the
// blocks of 32 rows in grid.y; up to 256 rows) where the layer is not for it - except on few tiles (fewer than 192 of 32 columns) at up to 64 rows, where it is the other way
// round: the 32-row build's workgroups own 32-64 columns, and two row blocks of a layer that narrow do not fill the chip with workgroups of 96-256
// (profiles/r04/sweep_rows_shared_tile_v1.txt: 4096x4096 at 64 rows 9.4 us against 16.8, 11008x4096 21.9 against 37.4; from 96 rows the shared tile is level or
// ahead: 11008x4096 at 128 rows 35.9 against 40.1, 4096x4096 at 256 rows 18.9 against 31.1 - sweep_rows_shared_tile_v2.txt, sweep_rows_192_256.txt).
int launch_gemv_rows_or_dyn(const GemvParams& p, int bits, hipStream_t stream, int dyn_cpl) {
  const bool dyn_first = (p.n + 31u) / 32u < 192u && p.m <= 64u;
  int st = kGemvDynNotEligible;
  if (!dyn_first) st = launch_gemv_rows(p, bits, stream);
  if (st == kGemvDynNotEligible) st = launch_gemv_dyn(p, bits, stream, dyn_cpl, 16, true, false, nullptr);
  if (st == kGemvDynNotEligible && dyn_first) st = launch_gemv_rows(p, bits, stream);
  return st;
}

}  // namespace xbit
