/* xbitops_tuning.h — entry points that exist in the MEASUREMENT build of the library only (libxbitops_hip_tuning.so:
 * the production sources with -DXBIT_TUNING=1, plus xbit_floor.hip).  Nothing here is part of the drop-in boundary
 * (include/xbitops_hip.h); bench.py and tools/ use it to put a pure-read floor next to a GEMV figure.  */
#ifndef XBITOPS_TUNING_H_
#define XBITOPS_TUNING_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define XBIT_TUNING_API __attribute__((visibility("default")))
#else
#define XBIT_TUNING_API
#endif

/* Read the packed weight rows `qweight` [packed_rows, n] (int32, 16-byte aligned, < 2 GiB) once and do nothing with them.
 * pattern 0: flat 16-byte stream.  pattern 32 / 48 / 64 / 128: whole-K column tiles of that lane-map width in the GEMV
 * kernels' own lane order and workgroup order, each tile owning `tcols` columns (0 = the whole width).
 * Returns 0, a negative XBIT_ERR_* or a positive hipError_t; asynchronous on `hip_stream`. */
XBIT_TUNING_API int xbit_tuning_read_floor(const void* qweight, uint32_t packed_rows, uint32_t n, int pattern, uint32_t tcols, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
