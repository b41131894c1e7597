// stream_read.hip -- the box's streaming ceiling, measured: every CU reads its share of a buffer once with the weight stream's
// own instruction (global_load_lds_dwordx4 nt, 1 KiB per wave instruction, eight in flight per wave, sixteen waves per CU)
// and does nothing with it.  bench.py reports the mat-vec kernels against this number as well as against the 8 TB/s of the
// data sheet (MI355X_MICROARCH.md quotes 6.4-6.8 TB/s for this access pattern).  Not part of the hot path.
#include "common.h"
#include "mmvq_units.h"

namespace {

constexpr int SR_WAVES = 16, SR_DEPTH = 8;

__global__ void __launch_bounds__(SR_WAVES * 64)
stream_read_kernel(const uint8_t * p, int64_t pieces) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * SR_WAVES, gw = (int64_t)blockIdx.x * SR_WAVES + wave;
    const int64_t per = (pieces + nw - 1) / nw, first = gw * per;
    int64_t n = pieces - first;
    n = n < 0 ? 0 : (n > per ? per : n);
    const uint8_t * src = p + (first << 10) + lane * 16;
    const unsigned ring = lds_addr(lds) + (unsigned)wave * SR_DEPTH * 1024;
    int64_t issued = 0;
    for (; issued < n && issued < SR_DEPTH; issued++) dma_piece_nt(src + (issued << 10), __builtin_amdgcn_readfirstlane(ring + ((unsigned)issued & (SR_DEPTH - 1)) * 1024));
    for (int64_t i = 0; i + SR_DEPTH < n; i++) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");           // piece i has landed: its slot takes piece i + 8
        dma_piece_nt(src + ((i + SR_DEPTH) << 10), __builtin_amdgcn_readfirstlane(ring + ((unsigned)i & (SR_DEPTH - 1)) * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

} // namespace

extern "C" int mi355_stream_read(const void * p, int64_t bytes, int workgroups, void * stream) {
    if (!p) return MI355_E_NULL;
    if (bytes < 0 || workgroups < 1 || workgroups > 65535) return MI355_E_SHAPE;
    if ((uintptr_t)p & 15) return MI355_E_ALIGN;
    if (bytes < 1024) return MI355_OK;
    const size_t lds = (size_t)SR_WAVES * SR_DEPTH * 1024;
    hipError_t e = allow_lds(stream_read_kernel, lds);
    if (e != hipSuccess) return (int)e;
    MI355_LAUNCH("stream_read", stream_read_kernel, dim3((unsigned)workgroups), dim3(SR_WAVES * 64), lds, (hipStream_t)stream, (const uint8_t *)p, bytes >> 10);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}
