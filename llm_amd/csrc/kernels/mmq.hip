// mmq.hip -- prefill MUL_MAT on block-quantized weights: LDS-staged int8 tiles feeding the CDNA4
// int8 MFMA (v_mfma_i32_32x32x32_i8), per-32-block f32 rescale.
//
// Replaces ggml_cuda_mul_mat_q / mul_mat_q (ggml/src/ggml-cuda/mmq.cu:68-224, mmq.cuh:2606-2841), whose
// AMD path has no matrix-core implementation in the reference snapshot (dp4a only, or dequantize +
// hipBLAS); this kernel is therefore designed from the hardware up.
//
// Roofline: int8 MFMA.  flops = 2*M*N*K (tests/test-backend-ops.cpp:2262-2265).
//
// Tiling
//   * workgroup = 256 threads = 4 waves, output tile 128 (weight rows m) x 128 (activation columns n),
//     K advanced 128 at a time (4 scale blocks of 32); LDS per workgroup ~45 KB -> 3 workgroups / CU.
//   * staging: every thread loads raw GGUF quant bytes with 128-bit loads, unpacks them to int8 in
//     registers (nibble / 6-bit fields -> signed bytes, offsets folded in) and writes 16-byte rows into
//     an LDS image with a 144-byte row pitch (128 + 16: conflict-free ds_read_b128 of MFMA operands).
//     Per (row, 32-block) f32 factors go to small LDS tables: ws = d*sc (weights), wm = dmin*m (q4_K /
//     q5_K mins), d8 and s8 = d8*sum(q8) (activations, from the q8_1 quantization).
//   * MFMA operand roles are chosen so that the *weight row* lands on the lane and the activation
//     column in the 16 accumulator registers: the weight-side factors are one scalar per lane, the
//     activation-side factors are four broadcast ds_read_b128 per tile.
//   * each wave owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles; per 32-block:
//         C = mfma_i32_32x32x32_i8(act_frag, w_frag, 0);  acc += float(C) * (ws * d8) - wm * s8
//     q6_K has 16-wide scale groups: its two halves of a 32-block sit on the two lane halves of the
//     operand, so two MFMAs with one half zeroed give the per-16 sums.
// The f32 epilogue per 32-block is the VALU cost of GGUF's fine-grained scales.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 128;
constexpr int PITCH = BK + 16;   // bytes per LDS row
constexpr int NKB = BK / 32;     // 32-blocks per stage

typedef int v16i __attribute__((ext_vector_type(16)));

struct MArgs {
    const uint8_t * W; int64_t w_row_stride;
    const int8_t * aq; const float * ads;
    float * dst; int64_t dst_col_stride;
    int m, n, k;
};

struct Lds {
    int8_t w[BM * PITCH];
    int8_t a[BN * PITCH];
    float  ws[2][NKB][BM];  // [sub][kb][m]: weight factor per 32-block (q6_K: per 16, sub = 0/1)
    float  wm[NKB][BM];     // min factor (q4_K / q5_K)
    float  d8[NKB][BN];
    float  s8[NKB][BN];
};

template <int TYPE> struct Traits { static constexpr bool MINS = false; static constexpr int SUB = 1; };
template <> struct Traits<MI355_TYPE_Q4_K> { static constexpr bool MINS = true; static constexpr int SUB = 1; };
template <> struct Traits<MI355_TYPE_Q5_K> { static constexpr bool MINS = true; static constexpr int SUB = 1; };
template <> struct Traits<MI355_TYPE_Q6_K> { static constexpr bool MINS = false; static constexpr int SUB = 2; };

static __device__ __forceinline__ void st16(int8_t * p, v4i v) { *(v4i *)p = v; }

static __device__ __forceinline__ void k4_pair(const v4i hdr, int g, float & ws0, float & ws1, float & wm0, float & wm1) {
    // scales / mins of sub-blocks 2g, 2g+1 (get_scale_min_k4, ggml-quants.c:625-632)
    const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;
    const int sh = (g & 1) * 16;
    uint32_t sc2, m2;
    if (g & 2) {
        sc2 = ((u2 >> sh) & 0x0f0fu) | (((u0 >> (sh + 6)) & 0x0303u) << 4);
        m2  = ((u2 >> (sh + 4)) & 0x0f0fu) | (((u1 >> (sh + 6)) & 0x0303u) << 4);
    } else {
        sc2 = (u0 >> sh) & 0x3f3fu;
        m2  = (u1 >> sh) & 0x3f3fu;
    }
    const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)((uint32_t)hdr.x >> 16));
    ws0 = d * (float)(sc2 & 0xff); ws1 = d * (float)(sc2 >> 8);
    wm0 = dmin * (float)(m2 & 0xff); wm1 = dmin * (float)(m2 >> 8);
}

// stage the weight tile rows [m0, m0+128) x k [k0, k0+128): thread t handles row t>>1, half t&1 (64 values)
template <int TYPE>
static __device__ __forceinline__ void stage_w(const MArgs & a, Lds & L, int m0, int k0, int tid) {
    const int r = tid >> 1, hf = tid & 1;
    const int row = m0 + r < a.m ? m0 + r : a.m - 1;
    const uint8_t * wrow = a.W + (int64_t)row * a.w_row_stride;
    int8_t * out = L.w + r * PITCH + hf * 64;
    if (TYPE == MI355_TYPE_Q4_K || TYPE == MI355_TYPE_Q5_K) {
        constexpr int BS = TYPE == MI355_TYPE_Q4_K ? 144 : 176;
        constexpr int QO = TYPE == MI355_TYPE_Q4_K ? 16 : 48;
        const uint8_t * blk = wrow + (int64_t)(k0 >> 8) * BS;
        const int g = ((k0 >> 7) & 1) * 2 + hf;            // 64-value group inside the super-block
        const v4i hdr = *(const v4i *)blk;
        v4i q0 = *(const v4i *)(blk + QO + g * 32), q1 = *(const v4i *)(blk + QO + g * 32 + 16);
        v4i lo0 = q0 & 0x0f0f0f0f, lo1 = q1 & 0x0f0f0f0f, hi0 = (q0 >> 4) & 0x0f0f0f0f, hi1 = (q1 >> 4) & 0x0f0f0f0f;
        if (TYPE == MI355_TYPE_Q5_K) {
            const v4i h0 = *(const v4i *)(blk + 16), h1 = *(const v4i *)(blk + 32);
            lo0 |= ((h0 >> (2 * g)) & 0x01010101) << 4;     lo1 |= ((h1 >> (2 * g)) & 0x01010101) << 4;
            hi0 |= ((h0 >> (2 * g + 1)) & 0x01010101) << 4; hi1 |= ((h1 >> (2 * g + 1)) & 0x01010101) << 4;
        }
        st16(out, lo0); st16(out + 16, lo1); st16(out + 32, hi0); st16(out + 48, hi1);
        float ws0, ws1, wm0, wm1;
        k4_pair(hdr, g, ws0, ws1, wm0, wm1);
        L.ws[0][hf * 2][r] = ws0; L.ws[0][hf * 2 + 1][r] = ws1;
        L.wm[hf * 2][r] = wm0;    L.wm[hf * 2 + 1][r] = wm1;
    } else if (TYPE == MI355_TYPE_Q6_K) {
        // stage = half nn of the super-block; this thread: values l = 16*hf .. +15 of each of the 4 quarters
        const uint8_t * blk = wrow + (int64_t)(k0 >> 8) * 210;
        const int nn = (k0 >> 7) & 1;
        const v4i qla = ld16_a2(blk + 64 * nn + 16 * hf), qlb = ld16_a2(blk + 64 * nn + 32 + 16 * hf);
        const v4i qh = ld16_a2(blk + 128 + 32 * nn + 16 * hf);
        v4i q[4];
        q[0] = (qla & 0x0f0f0f0f) | ((qh & 0x03030303) << 4);
        q[1] = (qlb & 0x0f0f0f0f) | (((qh >> 2) & 0x03030303) << 4);
        q[2] = ((qla >> 4) & 0x0f0f0f0f) | (((qh >> 4) & 0x03030303) << 4);
        q[3] = ((qlb >> 4) & 0x0f0f0f0f) | (((qh >> 6) & 0x03030303) << 4);
        const float d = h2f(ld_u16(blk + 208));
        const int8_t * sc = (const int8_t *)(blk + 192 + 8 * nn);
        int8_t * o = L.w + r * PITCH + hf * 16;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) q[i][j] = sub_bytes(q[i][j], 0x20202020);
            st16(o + 32 * i, q[i]);
            L.ws[hf][i][r] = d * (float)sc[hf + 2 * i];
        }
    } else if (TYPE == MI355_TYPE_Q8_0) {
        const uint8_t * blk = wrow + (int64_t)((k0 >> 5) + hf * 2) * 34;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            st16(out + 32 * b, ld16_a2(blk + 34 * b + 2));
            st16(out + 32 * b + 16, ld16_a2(blk + 34 * b + 18));
            L.ws[0][hf * 2 + b][r] = h2f(ld_u16(blk + 34 * b));
        }
    } else { // Q4_0
        const uint8_t * blk = wrow + (int64_t)((k0 >> 5) + hf * 2) * 18;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const v4i qs = ld16_a2(blk + 18 * b + 2);
            v4i lo, hi;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                lo[j] = sub_bytes(qs[j] & 0x0f0f0f0f, 0x08080808);
                hi[j] = sub_bytes((qs[j] >> 4) & 0x0f0f0f0f, 0x08080808);
            }
            st16(out + 32 * b, lo); st16(out + 32 * b + 16, hi);
            L.ws[0][hf * 2 + b][r] = h2f(ld_u16(blk + 18 * b));
        }
    }
}

// stage the activation tile columns [n0, n0+128) x k [k0, k0+128)
static __device__ __forceinline__ void stage_a(const MArgs & a, Lds & L, int n0, int k0, int tid) {
    const int r = tid >> 1, hf = tid & 1;
    const int col = n0 + r;
    int8_t * out = L.a + r * PITCH + hf * 64;
    if (col < a.n) {
        const int8_t * src = a.aq + (int64_t)col * a.k + k0 + hf * 64;
#pragma unroll
        for (int i = 0; i < 4; i++) st16(out + 16 * i, *(const v4i *)(src + 16 * i));
        const v4f ds = *(const v4f *)(a.ads + ((int64_t)col * (a.k / QK) + (k0 >> 5) + hf * 2) * 2);
        L.d8[hf * 2][r] = ds.x;     L.s8[hf * 2][r] = ds.x * ds.y;
        L.d8[hf * 2 + 1][r] = ds.z; L.s8[hf * 2 + 1][r] = ds.z * ds.w;
    } else {
        const v4i z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) st16(out + 16 * i, z);
        L.d8[hf * 2][r] = 0.f; L.s8[hf * 2][r] = 0.f; L.d8[hf * 2 + 1][r] = 0.f; L.s8[hf * 2 + 1][r] = 0.f;
    }
}

template <int TYPE>
__global__ void __launch_bounds__(256)
mmq_kernel(const MArgs a) {
    typedef Traits<TYPE> TR;
    __shared__ __attribute__((aligned(16))) Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int wm_ = wave & 1, wn_ = wave >> 1;       // wave's 64x64 sub-tile
    const int lr = lane & 31, lh = lane >> 5;

    float acc[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    for (int k0 = 0; k0 < a.k; k0 += BK) {
        __syncthreads();
        stage_w<TYPE>(a, L, m0, k0, tid);
        stage_a(a, L, n0, k0, tid);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) {
            v4i af[2], wf[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                af[t] = *(const v4i *)(L.a + (wn_ * 64 + t * 32 + lr) * PITCH + kb * 32 + lh * 16);
                wf[t] = *(const v4i *)(L.w + (wm_ * 64 + t * 32 + lr) * PITCH + kb * 32 + lh * 16);
            }
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {
                // activation factors of the 16 columns this lane's registers hold
                float d8[16], s8[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int nn = wn_ * 64 + tn * 32 + 8 * q + 4 * lh;
                    const v4f dv = *(const v4f *)&L.d8[kb][nn];
                    d8[4 * q] = dv.x; d8[4 * q + 1] = dv.y; d8[4 * q + 2] = dv.z; d8[4 * q + 3] = dv.w;
                    if (TR::MINS) {
                        const v4f sv = *(const v4f *)&L.s8[kb][nn];
                        s8[4 * q] = sv.x; s8[4 * q + 1] = sv.y; s8[4 * q + 2] = sv.z; s8[4 * q + 3] = sv.w;
                    }
                }
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    const int ml = wm_ * 64 + tm * 32 + lr;
#pragma unroll
                    for (int sub = 0; sub < TR::SUB; sub++) {
                        v4i wv = wf[tm];
                        if (TR::SUB == 2) { // keep only the 16 k-values of this scale group (they sit on lane half `sub`)
                            const v4i z = {0, 0, 0, 0};
                            wv = lh == sub ? wv : z;
                        }
                        const v16i zero = {0};
                        const v16i c = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[tn], wv, zero, 0, 0, 0);
                        const float ws = L.ws[sub][kb][ml];
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[tm][tn][r] += (float)c[r] * (ws * d8[r]);
                    }
                    if (TR::MINS) {
                        const float wmin = L.wm[kb][ml];
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[tm][tn][r] -= wmin * s8[r];
                    }
                }
            }
        }
    }

    // C register r of lane (lr, lh): column n = (r&3) + 8*(r>>2) + 4*lh, row m = lr
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        const int m = m0 + wm_ * 64 + tm * 32 + lr;
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + wn_ * 64 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < a.m && n < a.n) a.dst[(int64_t)n * a.dst_col_stride + m] = acc[tm][tn][r];
            }
    }
}

template <int TYPE>
static int launch(const MArgs & a, hipStream_t s, const char * tag) {
    const dim3 grid((unsigned)((a.n + BN - 1) / BN), (unsigned)((a.m + BM - 1) / BM));
    MI355_LAUNCH(tag, (mmq_kernel<TYPE>), grid, dim3(256), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

} // namespace

extern "C" int mi355_mul_mat_q(int type, const void * W, int64_t w_row_stride, const int8_t * act_q,
                                     const float * act_ds, float * dst, int64_t dst_col_stride,
                                     int64_t m, int64_t k, int64_t n, void * stream) {
    if (!W || !act_q || !act_ds || !dst) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0) return MI355_E_TYPE;
    if (m < 0 || n < 0 || k <= 0 || k % bs != 0 || m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff) return MI355_E_SHAPE;
    if (k % BK != 0) return MI355_E_UNSUPPORTED;          // caller falls back to the mat-vec kernel
    if ((n + BN - 1) / BN > 65535 * 32 || (m + BM - 1) / BM > 65535) return MI355_E_UNSUPPORTED;
    if (((uintptr_t)act_q & 15) || ((uintptr_t)act_ds & 15) || ((uintptr_t)dst & 3)) return MI355_E_ALIGN;
    const bool a16 = type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K;
    if (((uintptr_t)W | (uintptr_t)w_row_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (m == 0 || n == 0) return MI355_OK;
    MArgs a;
    a.W = (const uint8_t *)W; a.w_row_stride = w_row_stride; a.aq = act_q; a.ads = act_ds;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.m = (int)m; a.n = (int)n; a.k = (int)k;
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
        case MI355_TYPE_Q4_0: return launch<MI355_TYPE_Q4_0>(a, s, "mmq_q4_0");
        case MI355_TYPE_Q8_0: return launch<MI355_TYPE_Q8_0>(a, s, "mmq_q8_0");
        case MI355_TYPE_Q4_K: return launch<MI355_TYPE_Q4_K>(a, s, "mmq_q4_K");
        case MI355_TYPE_Q5_K: return launch<MI355_TYPE_Q5_K>(a, s, "mmq_q5_K");
        case MI355_TYPE_Q6_K: return launch<MI355_TYPE_Q6_K>(a, s, "mmq_q6_K");
    }
    return MI355_E_TYPE;
}
