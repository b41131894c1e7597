// dequantize.hip -- block dequantization to f32, bit-exact with the reference's
// dequantize_row_{q4_0,q4_1,q5_0,q5_1,q8_0,q4_K,q5_K,q6_K} (ggml/src/ggml-quants.c:249-357,1274-1296,
// 1476-1501,1684-1713).  The order of the f32 operations is part of the contract, so every
// product / difference is an explicit single-rounding intrinsic (no FMA contraction).
//
// Elementwise and HBM-bound: one lane per output element, consecutive lanes write consecutive
// floats (coalesced 256 B per wave-store); block headers are re-read through L1.
#include "common.h"
#include "deq_elem.h"

template <int TYPE>
__global__ void __launch_bounds__(256)
dequantize_kernel(const uint8_t * __restrict__ blocks, float * __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = deq_elem<TYPE>(blocks, i);
}

template <int TYPE>
static int launch_deq(const void * blocks, float * y, int64_t n, hipStream_t stream) {
    const int64_t nb = (n + 255) / 256;
    const unsigned grid = (unsigned)(nb < 8192 ? nb : 8192);
    MI355_LAUNCH("dequantize", (dequantize_kernel<TYPE>), dim3(grid), dim3(256), 0, stream, (const uint8_t *)blocks, y, n);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_dequantize_row(int type, const void * blocks, float * y, int64_t n, void * stream) {
    if (!blocks || !y) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0) return MI355_E_TYPE;
    if (n < 0 || n % bs != 0) return MI355_E_SHAPE;
    if (((uintptr_t)blocks & 1) || ((uintptr_t)y & 3)) return MI355_E_ALIGN;
    if (n == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
        case MI355_TYPE_F32: {
            hipError_t e = hipMemcpyAsync(y, blocks, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
            return (int)e;
        }
        case MI355_TYPE_F16:  return launch_deq<MI355_TYPE_F16>(blocks, y, n, s);
        case MI355_TYPE_Q4_0: return launch_deq<MI355_TYPE_Q4_0>(blocks, y, n, s);
        case MI355_TYPE_Q4_1: return launch_deq<MI355_TYPE_Q4_1>(blocks, y, n, s);
        case MI355_TYPE_Q5_0: return launch_deq<MI355_TYPE_Q5_0>(blocks, y, n, s);
        case MI355_TYPE_Q5_1: return launch_deq<MI355_TYPE_Q5_1>(blocks, y, n, s);
        case MI355_TYPE_BF16: return launch_deq<MI355_TYPE_BF16>(blocks, y, n, s);
        case MI355_TYPE_Q8_0: return launch_deq<MI355_TYPE_Q8_0>(blocks, y, n, s);
        case MI355_TYPE_Q8_1: return launch_deq<MI355_TYPE_Q8_1>(blocks, y, n, s);
        case MI355_TYPE_Q4_K: return launch_deq<MI355_TYPE_Q4_K>(blocks, y, n, s);
        case MI355_TYPE_Q5_K: return launch_deq<MI355_TYPE_Q5_K>(blocks, y, n, s);
        case MI355_TYPE_Q6_K: return launch_deq<MI355_TYPE_Q6_K>(blocks, y, n, s);
        case MI355_TYPE_Q2_K: return launch_deq<MI355_TYPE_Q2_K>(blocks, y, n, s);
        case MI355_TYPE_Q3_K: return launch_deq<MI355_TYPE_Q3_K>(blocks, y, n, s);
        case MI355_TYPE_IQ4_NL: return launch_deq<MI355_TYPE_IQ4_NL>(blocks, y, n, s);
        case MI355_TYPE_IQ4_XS: return launch_deq<MI355_TYPE_IQ4_XS>(blocks, y, n, s);
    }
    return MI355_E_TYPE;
}
