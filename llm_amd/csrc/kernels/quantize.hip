// quantize.hip -- f32 activations -> q8_1, bit-exact with quantize_row_q8_1_ref
// (ggml/src/ggml-quants.c:214-247); the GPU analogue it replaces is quantize_q8_1
// (ggml/src/ggml-cuda/quantize.cu:4-48), which rounds x/d and stores fp16(sum x) instead.
//
// HBM-bound elementwise pass: each lane owns 4 consecutive floats (one 128-bit load), the 8 lanes
// of a 32-value block reduce amax / sum with three xor-shuffles inside the 64-wide wavefront.
// Built with -ffp-contract=off; every float op below is a single correctly-rounded IEEE op so the
// result is independent of the compiler's fusion choices.
#include "common.h"

template <bool SOA, bool VEC>
__global__ void __launch_bounds__(256)
quantize_q8_1_kernel(const float * __restrict__ x, uint8_t * __restrict__ y_aos,
                     int8_t * __restrict__ act_q, float * __restrict__ act_ds,
                     int64_t k, int64_t ne1, int64_t ne2, int64_t nb1, int64_t nb2, int64_t nb3) {
    const int64_t i4 = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; // index of this lane's float4
    if (i4 * 4 >= k) return;                                           // whole 8-lane groups leave together
    const int64_t row = blockIdx.x;                                    // (i3*ne2 + i2)*ne1 + i1
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * xr = (const float *)((const char *)x + i1 * nb1 + i2 * nb2 + i3 * nb3);

    float v[4];
    if (VEC) {
        const v4f t = *(const v4f *)(xr + i4 * 4);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = xr[i4 * 4 + j];
    }

    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));

    const float d  = __fdiv_rn(amax, 127.0f);
    const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;

    int q[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // roundf(): half away from zero, as in the reference.  The reference then narrows float -> int8
        // through x86's cvttss2si, which wraps modulo 256 inside int32 range and yields 0x80000000
        // (low byte 0) for inf / NaN / |r| >= 2^31 (reached when d is an f32 denormal and 1/d
        // overflows); v_cvt_i32_f32 would saturate instead, so that case is made explicit.
        const float r = roundf(__fmul_rn(v[j], id));
        q[j] = fabsf(r) < 2147483648.0f ? (int)(int8_t)(int)r : 0;
        sum += q[j];
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o, 64);

    const uint32_t packed = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) |
                            ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
    const int64_t ib  = i4 >> 3;      // 32-block inside the row
    const int     sub = i4 & 7;       // this lane's dword inside the block
    const __half  dh  = __float2half_rn(d);
    if (SOA) {
        *(uint32_t *)(act_q + row * k + i4 * 4) = packed;
        if (sub == 0) {
            v2f ds; ds.x = __half2float(dh); ds.y = (float)sum;
            *(v2f *)(act_ds + (row * (k / QK) + ib) * 2) = ds;
        }
    } else {
        uint8_t * blk = y_aos + (row * (k / QK) + ib) * 36; // block_q8_1, ggml-common.h:217-226
        *(uint32_t *)(blk + 4 + sub * 4) = packed;
        if (sub == 0) {
            const __half sh = __float2half_rn(__fmul_rn((float)sum, d));
            const uint32_t dsw = (uint32_t)__half_as_ushort(dh) | ((uint32_t)__half_as_ushort(sh) << 16);
            *(uint32_t *)blk = dsw;
        }
    }
}

static int launch_quantize(bool soa, const float * x, void * y, int8_t * aq, float * ads, int64_t k,
                           int64_t ne1, int64_t ne2, int64_t ne3, int64_t nb1, int64_t nb2, int64_t nb3,
                           hipStream_t stream) {
    if (!x || (soa ? (!aq || !ads) : !y)) return MI355_E_NULL;
    if (k <= 0 || k % QK != 0 || ne1 < 0 || ne2 < 0 || ne3 < 0) return MI355_E_SHAPE;
    if (((uintptr_t)x | (uintptr_t)nb1 | (uintptr_t)nb2 | (uintptr_t)nb3) & 3) return MI355_E_ALIGN;
    if (soa ? (((uintptr_t)aq & 3) || ((uintptr_t)ads & 7)) : ((uintptr_t)y & 3)) return MI355_E_ALIGN;
    const int64_t rows = ne1 * ne2 * ne3;
    if (rows == 0) return MI355_OK;
    if (rows > 0x7fffffff) return MI355_E_SHAPE;
    const bool vec = (((uintptr_t)x | (uintptr_t)nb1 | (uintptr_t)nb2 | (uintptr_t)nb3) & 15) == 0;
    const dim3 block(256);
    const dim3 grid((unsigned)rows, (unsigned)((k / 4 + 255) / 256));
    if (grid.y > 65535) return MI355_E_SHAPE;
    uint8_t * ya = (uint8_t *)y;
    if (soa) {
        if (vec) MI355_LAUNCH("quantize_q8_1", (quantize_q8_1_kernel<true, true>), grid, block, 0, stream, x, ya, aq, ads, k, ne1, ne2, nb1, nb2, nb3);
        else     MI355_LAUNCH("quantize_q8_1", (quantize_q8_1_kernel<true, false>), grid, block, 0, stream, x, ya, aq, ads, k, ne1, ne2, nb1, nb2, nb3);
    } else {
        if (vec) MI355_LAUNCH("quantize_q8_1", (quantize_q8_1_kernel<false, true>), grid, block, 0, stream, x, ya, aq, ads, k, ne1, ne2, nb1, nb2, nb3);
        else     MI355_LAUNCH("quantize_q8_1", (quantize_q8_1_kernel<false, false>), grid, block, 0, stream, x, ya, aq, ads, k, ne1, ne2, nb1, nb2, nb3);
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" {

int mi355_quantize_row_q8_1(const float * x, void * y, int64_t k, int64_t ne1, int64_t ne2, int64_t ne3,
                            int64_t nb1, int64_t nb2, int64_t nb3, void * stream) {
    return launch_quantize(false, x, y, nullptr, nullptr, k, ne1, ne2, ne3, nb1, nb2, nb3, (hipStream_t)stream);
}

int64_t mi355_act_q_bytes(int64_t k, int64_t rows)  { return (k % QK || k < 0 || rows < 0) ? -1 : k * rows; }
int64_t mi355_act_ds_bytes(int64_t k, int64_t rows) { return (k % QK || k < 0 || rows < 0) ? -1 : k / QK * 8 * rows; }

int mi355_quantize_act(const float * x, int8_t * act_q, float * act_ds, int64_t k, int64_t ne1, int64_t ne2,
                       int64_t ne3, int64_t nb1, int64_t nb2, int64_t nb3, void * stream) {
    return launch_quantize(true, x, nullptr, act_q, act_ds, k, ne1, ne2, ne3, nb1, nb2, nb3, (hipStream_t)stream);
}

} // extern "C"
