// mmq.hip -- prefill tile kernel (int8 MFMA), placeholder entry until the kernel lands:
// returns MI355_E_UNSUPPORTED so mi355_mul_mat falls back to column groups of the mat-vec kernel.
#include "common.h"

extern "C" int mi355_mul_mat_q_tiles(int, const void *, int64_t, const int8_t *, const float *, float *, int64_t,
                                     int64_t, int64_t, int64_t, void *) {
    return MI355_E_UNSUPPORTED;
}
