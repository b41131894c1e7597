// mmf.hip -- MUL_MAT on float weights (f32 / f16 / bf16) for small batches, any strides.
//
// Replaces mul_mat_vec (ggml/src/ggml-cuda/mmv.cu) and the small-batch end of ggml_cuda_op_mul_mat_cublas
// (ggml-cuda.cu:1205-1351); oracle: ggml_compute_forward_mul_mat with vec_dot_type f16 / bf16 / f32
// (ggml-cpu/ggml-cpu.c:1196-1388).  This is what a graph WITHOUT flash attention multiplies by (K and V views of the
// cache: permuted, broadcast over the GQA heads) and what an unquantized GGUF's decode step streams; prefill-sized
// batches of a plain 2-D matrix go to the bf16 MFMA GEMM (mmdq.hip) instead.
//
// Roofline: HBM (weights once per group of NC columns).  One wave per weight row, 8 elements per lane and step with
// 16-byte loads where the row addresses allow it, f32 accumulation, DPP wave reduction.
#include "common.h"

namespace {

struct MArgs {
    const char * a; int64_t a_nb1, a_nb2, a_nb3;
    const char * b; int64_t b_nb1, b_nb2, b_nb3;
    char * d; int64_t d_nb1, d_nb2, d_nb3;
    int m, n, k, ne12, r2, r3;
};

template <int T> static __device__ __forceinline__ float ld_elem(const char * p, int64_t i) {
    if (T == MI355_TYPE_F32)  return ((const float *)p)[i];
    if (T == MI355_TYPE_F16)  return h2f(((const uint16_t *)p)[i]);
    return __uint_as_float((uint32_t)((const uint16_t *)p)[i] << 16);
}

// 8 consecutive elements starting at element i (16-byte aligned address)
template <int T> static __device__ __forceinline__ void ld8(const char * p, int64_t i, float * f) {
    if (T == MI355_TYPE_F32) {
        const v4f x = *(const v4f *)(p + i * 4), y = *(const v4f *)(p + i * 4 + 16);
        f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w; f[4] = y.x; f[5] = y.y; f[6] = y.z; f[7] = y.w;
    } else {
        const v4i q = *(const v4i *)(p + i * 2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = (uint32_t)q[j];
            if (T == MI355_TYPE_F16) { f[2 * j] = h2f((uint16_t)(w & 0xffff)); f[2 * j + 1] = h2f((uint16_t)(w >> 16)); }
            else                     { f[2 * j] = __uint_as_float(w << 16);    f[2 * j + 1] = __uint_as_float(w & 0xffff0000u); }
        }
    }
}

template <int TA, int TB, int NC>
__global__ void __launch_bounds__(256)
mmf_kernel(const MArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= a.m) return;
    const int c0 = blockIdx.y * NC;
    const int ch = blockIdx.z, i12 = ch % a.ne12, i13 = ch / a.ne12;
    const char * ar = a.a + (int64_t)row * a.a_nb1 + (int64_t)(i12 / a.r2) * a.a_nb2 + (int64_t)(i13 / a.r3) * a.a_nb3;
    const char * bc[NC];
    bool aligned = ((uintptr_t)ar & 15) == 0 && a.k % 8 == 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int col = c0 + c < a.n ? c0 + c : a.n - 1;          // columns past the end recompute the last one, never stored
        bc[c] = a.b + (int64_t)col * a.b_nb1 + (int64_t)i12 * a.b_nb2 + (int64_t)i13 * a.b_nb3;
        aligned = aligned && ((uintptr_t)bc[c] & 15) == 0;
    }
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0f;
    if (aligned) {
        for (int64_t i = (int64_t)lane * 8; i < a.k; i += 512) {
            float w[8];
            ld8<TA>(ar, i, w);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                float x[8];
                ld8<TB>(bc[c], i, x);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[c] += w[j] * x[j];
            }
        }
    } else {
        for (int64_t i = lane; i < a.k; i += 64) {
            const float w = ld_elem<TA>(ar, i);
#pragma unroll
            for (int c = 0; c < NC; c++) acc[c] += w * ld_elem<TB>(bc[c], i);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float s = wave_sum(acc[c]);
        if (lane == 0 && c0 + c < a.n)
            *(float *)(a.d + (int64_t)row * 4 + (int64_t)(c0 + c) * a.d_nb1 + (int64_t)i12 * a.d_nb2 + (int64_t)i13 * a.d_nb3) = s;
    }
}

template <int TA, int TB>
static int launch_mmf(const MArgs & a, int64_t nch, hipStream_t s) {
    const unsigned gx = (unsigned)((a.m + 3) / 4);
#define MMF(NC) MI355_LAUNCH("mmf", (mmf_kernel<TA, TB, NC>), dim3(gx, (unsigned)((a.n + NC - 1) / NC), (unsigned)nch), dim3(256), 0, s, a)
    if (a.n == 1)      MMF(1);
    else if (a.n == 2) MMF(2);
    else if (a.n <= 4) MMF(4);
    else               MMF(8);
#undef MMF
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

} // namespace

extern "C" int mi355_mul_mat_f(const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, void * stream) {
    if (!a || !b || !dst || !a->data || !b->data || !dst->data) return MI355_E_NULL;
    const int ta = a->type, tb = b->type;
    if (ta != MI355_TYPE_F32 && ta != MI355_TYPE_F16 && ta != MI355_TYPE_BF16) return MI355_E_TYPE;
    if ((tb != MI355_TYPE_F32 && tb != MI355_TYPE_F16) || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (a->nb[0] != mi355_type_size(ta) || b->nb[0] != mi355_type_size(tb) || dst->nb[0] != 4) return MI355_E_SHAPE;   // rows contiguous
    if (a->ne[0] != b->ne[0] || dst->ne[0] != a->ne[1] || dst->ne[1] != b->ne[1] || dst->ne[2] != b->ne[2] || dst->ne[3] != b->ne[3] ||
        a->ne[2] < 1 || a->ne[3] < 1 || b->ne[2] % a->ne[2] != 0 || b->ne[3] % a->ne[3] != 0) return MI355_E_SHAPE;
    const int64_t nch = b->ne[2] * b->ne[3];
    if (a->ne[0] > 0x7fffffff || a->ne[1] > 0x7fffffff || b->ne[1] > 0x7fffffff || nch > 65535 || (b->ne[1] + 7) / 8 > 65535) return MI355_E_SHAPE;
    const int64_t esz = mi355_type_size(ta), bsz = mi355_type_size(tb);
    if (((uintptr_t)a->data | a->nb[1] | a->nb[2] | a->nb[3]) & (esz - 1)) return MI355_E_ALIGN;
    if (((uintptr_t)b->data | b->nb[1] | b->nb[2] | b->nb[3]) & (bsz - 1)) return MI355_E_ALIGN;
    if (((uintptr_t)dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 3) return MI355_E_ALIGN;
    if (a->ne[1] == 0 || b->ne[1] == 0 || nch == 0) return MI355_OK;
    MArgs k;
    k.a = (const char *)a->data; k.a_nb1 = a->nb[1]; k.a_nb2 = a->nb[2]; k.a_nb3 = a->nb[3];
    k.b = (const char *)b->data; k.b_nb1 = b->nb[1]; k.b_nb2 = b->nb[2]; k.b_nb3 = b->nb[3];
    k.d = (char *)dst->data; k.d_nb1 = dst->nb[1]; k.d_nb2 = dst->nb[2]; k.d_nb3 = dst->nb[3];
    k.m = (int)a->ne[1]; k.n = (int)b->ne[1]; k.k = (int)a->ne[0]; k.ne12 = (int)b->ne[2];
    k.r2 = (int)(b->ne[2] / a->ne[2]); k.r3 = (int)(b->ne[3] / a->ne[3]);
    hipStream_t s = (hipStream_t)stream;
    if (a->ne[0] == 0) {        // K = 0: the result is all zeros
        for (int64_t c = 0; c < nch; c++)
            for (int64_t col = 0; col < b->ne[1]; col++) {
                hipError_t e = hipMemsetAsync(k.d + col * k.d_nb1 + (c % k.ne12) * k.d_nb2 + (c / k.ne12) * k.d_nb3, 0, (size_t)k.m * 4, s);
                if (e != hipSuccess) return (int)e;
            }
        return MI355_OK;
    }
#define TB_SWITCH(TA) (tb == MI355_TYPE_F32 ? launch_mmf<TA, MI355_TYPE_F32>(k, nch, s) : launch_mmf<TA, MI355_TYPE_F16>(k, nch, s))
    switch (ta) {
        case MI355_TYPE_F32:  return TB_SWITCH(MI355_TYPE_F32);
        case MI355_TYPE_F16:  return TB_SWITCH(MI355_TYPE_F16);
        case MI355_TYPE_BF16: return TB_SWITCH(MI355_TYPE_BF16);
    }
#undef TB_SWITCH
    return MI355_E_TYPE;
}
