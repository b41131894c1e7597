// mmqk.hip -- prefill MUL_MAT of the K-quants (Q4_K, Q5_K) on the int8 matrix cores with NO per-block rescale.
//
// Same contract as mul_mat_q (ggml/src/ggml-cuda/mmq.cu:68-224, mmq.cuh:2606-2841): dst[n][m] = sum_k W[m][k] x[n][k].
// The reference's MMQ (and mmq.hip here) rescales every 32x32 tile once per 32-block in f32 -- ~50 VALU instructions
// against 32 cycles of MFMA -- and the bf16 GEMM (mmdq.hip) pays ~2.5 VALU instructions per weight to dequantize.
// The K-quants do not need either: their sub-block scales are 6-bit INTEGERS under one f16 scale per 256 weights, and
// ggml's own CPU contract for them quantizes the activations per 256 values too (q8_K, quantize_row_q8_K_ref
// ggml-quants.c:2471-2508; ggml_vec_dot_q4_K_q8_K ggml-cpu/quants.c:515-588).  So, as in mmsk.hip (mmsk2_kernel):
//   * sc_j is split into P digits of B bits (2 x 3 for q4_K: 7 * 15 = 105; 3 x 2 for q5_K: 3 * 31 = 93 -- a signed
//     byte either way); digit * q is one v_pk_mul_lo_u16 per 4 weights, done ONCE per weight while the tile is staged;
//   * the inner loop is ds_read_b128 + v_mfma_i32_32x32x32_i8 into one i32 accumulator per digit, over the whole
//     super-block; one fold per 256 weights: acc += d8[col] * (d[row] * (C_0 + (C_1 << B) ..) - dmin[row] * mins);
//   * mins = sum_j m_j[row] * bsum_j[col] is itself two int8 MFMAs per tile and super-block: K slots 0..7 hold the low
//     7 bits of the 8 block sums, slots 8..15 the (signed) high bits, the weight side holds the 8 mins twice:
//     S1 = sum m lo + sum m hi, S2 = sum m hi (low slots zeroed), mins = S1 + 127 S2.
//
// Roofline: int8 MFMA (5 POP/s dense; the digit planes make it P MFMAs per 32x32x32 block, i.e. the bf16 rate for
// q4_K).  flops = 2 M N K (tests/test-backend-ops.cpp:2262-2265).
//
// Tiling: workgroup = 512 threads = 8 waves (2 x 4), output tile 128 weight rows x 128 activation columns, K step 128
// (four 32-blocks, two stages per super-block); a wave owns 64 rows x 32 columns (two i32 accumulators per digit:
// with 64 x 64 per wave the digit planes would need 192 accumulator registers).  LDS rows have a 144-byte pitch
// (128 B + 16: conflict-free ds_read_b128).  Double buffered, one barrier per stage, the loop body is unrolled over
// the 2 stages of a super-block so that scale words and "first MFMA of the super-block" are compile-time.
//
// STATUS (round 1): correct (tests/test_kernels_gpu.py, NMSE <= 1e-10 against the q8_K contract in f64) but NOT the
// default: 452 TFLOP/s on 14336 x 4096 x 512 against 564 for mmdq.hip.  History on the GPU (same shape): 64 x 128 tile,
// 4 waves, K step 64: 146 us.  Ablations of that version: ds_read + MFMA alone 52 us (1.1 PFLOP/s; the MFMAs
// themselves are 31 us), + global fetches 114 us, + staging and fold instead 120 us, everything 146-153 us.  A
// two-stage fetch distance, line-friendlier activation loads and halving the activation traffic (128-row tile) each
// changed nothing; K step 128 (16 MFMAs per wave and barrier instead of 8) gave 151 -> 133 us.  Next: in-kernel
// stamps per phase (as lab/mmsk_stamps.hip), interleaving the staging VALU with the MFMAs (sched_group_barrier), and
// the same treatment for the fold.
#include "common.h"

namespace {

typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr int QBM = 128, QBN = 128, QBK = 128, QPITCH = QBK + 16, QTHREADS = 512;

struct QArgs {
    const uint8_t * W; int64_t w_row_stride;
    const int8_t * aq; const float * ad; const int16_t * bs;     // q8_K planes: [n][k], [n][k/256], [n][k/16]
    float * dst; int64_t dst_col_stride;
    int m, n, k;
};

static __device__ __forceinline__ int pk_bytes_mul(int x, int rep) {
    return __builtin_bit_cast(int, (us2)(__builtin_bit_cast(us2, x) * __builtin_bit_cast(us2, rep)));
}

template <int TYPE>
__global__ void __launch_bounds__(QTHREADS)
mmqk_kernel(const QArgs a) {
    constexpr int P = TYPE == MI355_TYPE_Q4_K ? 2 : 3, B = TYPE == MI355_TYPE_Q4_K ? 3 : 2;
    constexpr int BS = TYPE == MI355_TYPE_Q4_K ? 144 : 176, QO = TYPE == MI355_TYPE_Q4_K ? 16 : 48;
    __shared__ __attribute__((aligned(16))) uint8_t Lw[2][P][QBM * QPITCH];
    __shared__ __attribute__((aligned(16))) uint8_t La[2][QBN * QPITCH];
    __shared__ __attribute__((aligned(16))) float Td8[2][QBN];      // per super-block tables, by super-block parity
    __shared__ __attribute__((aligned(16))) v4i   Tbs[2][QBN];      // block-sum digits of a column: lo[8] | hi[8]
    __shared__ __attribute__((aligned(16))) v2f   Tdw[2][QBM];      // {d, dmin} of a row
    __shared__ __attribute__((aligned(16))) v4i   Tm[2][QBM];       // the 8 mins of a row, twice
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * QBN, m0 = blockIdx.y * QBM;
    const int nsb = a.k / 256;
    const int wm = wave >> 2, wn = wave & 3;      // wave tile: rows 64 wm .. + 63, columns 32 wn .. + 31

    // staging roles.  weights: row, 32-byte unit of the stage (two per stage), 16-byte half of the unit
    const int sr = tid >> 2, su = (tid >> 1) & 1, sh = tid & 1;
    const uint8_t * wrow = a.W + (int64_t)(m0 + sr < a.m ? m0 + sr : a.m - 1) * a.w_row_stride;
    // activations: 4 consecutive lanes read the 128 contiguous bytes (one cache line) of one column and stage
    const int ac = tid >> 2, ap = tid & 3;
    const int8_t * acol = a.aq + (int64_t)(n0 + ac < a.n ? n0 + ac : a.n - 1) * a.k + ap * 32;
    // tables: threads 0..127 a column and a row each
    const int tcol = n0 + (tid & 127) < a.n ? n0 + (tid & 127) : a.n - 1;
    const float * tad = a.ad + (int64_t)tcol * nsb;
    const v4i * tbs = (const v4i *)(a.bs + (int64_t)tcol * (a.k / 16));
    const uint8_t * trow = a.W + (int64_t)(m0 + (tid & 127) < a.m ? m0 + (tid & 127) : a.m - 1) * a.w_row_stride;

    v16f acc[2];
    v16i C[P][2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;

    // registers of the stage in flight (stage G = 128 weights = units 2 G, 2 G + 1 of super-block sb)
    v4i wq, wqh, whdr, xa[2];
    float t_d8; v4i t_b0, t_b1, t_hdr;
    auto fetch = [&](int sb, int G, bool tables) {
        const uint8_t * blk = wrow + (int64_t)sb * BS;
        wq = *(const v4i *)(blk + QO + (2 * G + su) * 32 + sh * 16);
        if (TYPE == MI355_TYPE_Q5_K) wqh = *(const v4i *)(blk + 16 + sh * 16);
        whdr = *(const v4i *)blk;
        xa[0] = *(const v4i *)(acol + (int64_t)sb * 256 + G * 128);
        xa[1] = *(const v4i *)(acol + (int64_t)sb * 256 + G * 128 + 16);
        if (tables) {
            // every thread loads (the upper ones a duplicate): a masked load would be a branch in the main loop, and at
            // a branch the compiler's wait-count bookkeeping drains every load
            t_d8 = tad[sb]; t_b0 = tbs[sb * 2]; t_b1 = tbs[sb * 2 + 1];
            t_hdr = *(const v4i *)(trow + (int64_t)sb * BS);
        }
    };
    auto stash = [&](int sb, int G, int buf, bool tables) {
        // scales of blocks 2 g and 2 g + 1, g = 2 G + su, of this thread's row (get_scale_min_k4, ggml-quants.c:625-632)
        const uint32_t u0 = whdr.y, u2 = whdr.w;
        const uint32_t scw = G == 0 ? (u0 & 0x3f3f3f3fu) : ((u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u));
        const uint32_t sc0 = (scw >> (16 * su)) & 0xffu, sc1 = (scw >> (16 * su + 8)) & 0xffu;
        v4i x0 = wq & 0x0f0f0f0f, x1 = (wq >> 4) & 0x0f0f0f0f;
        if (TYPE == MI355_TYPE_Q5_K) {
            const int g2 = 4 * G + 2 * su;
            x0 |= ((wqh >> g2) & 0x01010101) << 4;
            x1 |= ((wqh >> (g2 + 1)) & 0x01010101) << 4;
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
            const uint32_t d0 = (sc0 >> (B * p)) & ((1u << B) - 1), d1 = (sc1 >> (B * p)) & ((1u << B) - 1);
            const int r0 = (int)(d0 | (d0 << 16)), r1 = (int)(d1 | (d1 << 16));
            v4i y0, y1;
#pragma unroll
            for (int i = 0; i < 4; i++) { y0[i] = pk_bytes_mul(x0[i], r0); y1[i] = pk_bytes_mul(x1[i], r1); }
            *(v4i *)(Lw[buf][p] + sr * QPITCH + su * 64 + sh * 16)      = y0;
            *(v4i *)(Lw[buf][p] + sr * QPITCH + su * 64 + 32 + sh * 16) = y1;
        }
        // columns past n repeat column n - 1 (never stored)
        *(v4i *)(La[buf] + ac * QPITCH + ap * 32)      = xa[0];
        *(v4i *)(La[buf] + ac * QPITCH + ap * 32 + 16) = xa[1];
        if (tables) {
            const int par = sb & 1;
            if (tid < QBN) {
                // 8 block sums (two int16 halves each) -> low 7 bits and signed high part, as bytes
                const int w[8] = {t_b0.x, t_b0.y, t_b0.z, t_b0.w, t_b1.x, t_b1.y, t_b1.z, t_b1.w};
                uint32_t lo[2] = {0, 0}, hi[2] = {0, 0};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int s = (w[j] << 16 >> 16) + (w[j] >> 16);
                    lo[j >> 2] |= (uint32_t)(s & 127) << (8 * (j & 3));
                    hi[j >> 2] |= (uint32_t)((s >> 7) & 0xff) << (8 * (j & 3));
                }
                Td8[par][tid] = t_d8;
                const v4i dg = {(int)lo[0], (int)lo[1], (int)hi[0], (int)hi[1]};
                Tbs[par][tid] = dg;
                const uint32_t h1 = t_hdr.z, h2 = t_hdr.w;
                const int mw0 = (int)(h1 & 0x3f3f3f3fu), mw1 = (int)(((h2 >> 4) & 0x0f0f0f0fu) | ((h1 >> 2) & 0x30303030u));
                v2f dw; dw.x = h2f((uint16_t)(t_hdr.x & 0xffff)); dw.y = h2f((uint16_t)((uint32_t)t_hdr.x >> 16));
                Tdw[par][tid] = dw;
                const v4i mm = {mw0, mw1, mw0, mw1};
                Tm[par][tid] = mm;
            }
        }
    };

    // prologue: stage 0 into buffer 0, stage 1 in flight
    fetch(0, 0, true);
    stash(0, 0, 0, true);
    fetch(0, 1, false);
    __syncthreads();

    for (int sb = 0; sb < nsb; sb++) {
#pragma unroll
        for (int G = 0; G < 2; G++) {
            const int buf = G;      // stage s = 2 sb + G
            // stage s + 1 -> the other LDS buffer; its registers then receive stage s + 2.  Unconditional: past the end
            // the last super-block is staged again into the buffer nobody reads, so the loop body stays straight-line.
            const int sbn = sb + 1 < nsb ? sb + 1 : sb;
            stash(G == 0 ? sb : sbn, G ^ 1, buf ^ 1, G == 1);
            fetch(sbn, G, G == 0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v4i aop = *(const v4i *)(La[buf] + (wn * 32 + lr) * QPITCH + j * 32 + lh * 16);
#pragma unroll
                for (int p = 0; p < P; p++)
#pragma unroll
                    for (int tm = 0; tm < 2; tm++) {
                        const v4i wop = *(const v4i *)(Lw[buf][p] + (wm * 64 + tm * 32 + lr) * QPITCH + j * 32 + lh * 16);
                        const v16i zero = {0};
                        C[p][tm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop, wop, (G == 0 && j == 0) ? zero : C[p][tm], 0, 0, 0);
                    }
            }
            if (G == 1) {
                // fold the super-block: lane = weight row, register r = column (r&3) + 8 (r>>2) + 4 lh of the wave's 32
                const int par = sb & 1;
                const v4i z = {0, 0, 0, 0};
                v4i a1 = lh == 0 ? Tbs[par][wn * 32 + lr] : z;
                v4i a2 = a1; a2.x = 0; a2.y = 0;
                v4f d8v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) d8v[q] = *(const v4f *)&Td8[par][wn * 32 + 8 * q + 4 * lh];
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    const v4i bm = lh == 0 ? Tm[par][wm * 64 + tm * 32 + lr] : z;
                    const v2f dw = Tdw[par][wm * 64 + tm * 32 + lr];
                    const v16i zero = {0};
                    const v16i s1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bm, zero, 0, 0, 0);
                    const v16i s2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, bm, zero, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        int t = C[0][tm][r];
#pragma unroll
                        for (int p = 1; p < P; p++) t += C[p][tm][r] << (B * p);
                        const int mi = s1[r] + 127 * s2[r];
                        const float u = __builtin_fmaf(-dw.y, (float)mi, dw.x * (float)t);
                        acc[tm][r] = __builtin_fmaf(d8v[r >> 2][r & 3], u, acc[tm][r]);
                    }
                }
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        const int mrow = m0 + wm * 64 + tm * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ncol = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mrow < a.m && ncol < a.n) a.dst[(int64_t)ncol * a.dst_col_stride + mrow] = acc[tm][r];
        }
    }
}

} // namespace

extern "C" int mi355_mul_mat_qk(int type, const void * W, int64_t w_row_stride, const int8_t * act_q, const float * act_d,
                                const int16_t * act_bsums, float * dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n,
                                void * stream) {
    if (!W || !act_q || !act_d || !act_bsums || !dst) return MI355_E_NULL;
    if (type != MI355_TYPE_Q4_K && type != MI355_TYPE_Q5_K) return mi355_blck_size(type) ? MI355_E_UNSUPPORTED : MI355_E_TYPE;
    if (m < 0 || n < 0 || k <= 0 || k % 256 != 0 || m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff || dst_col_stride < m) return MI355_E_SHAPE;
    if ((n + QBN - 1) / QBN > 0x7fffffff / 2 || (m + QBM - 1) / QBM > 65535) return MI355_E_UNSUPPORTED;
    if (((uintptr_t)act_q & 15) || ((uintptr_t)act_d & 3) || ((uintptr_t)act_bsums & 15) || ((uintptr_t)dst & 3)) return MI355_E_ALIGN;
    if (((uintptr_t)W | (uintptr_t)w_row_stride) & 15) return MI355_E_ALIGN;
    if (w_row_stride < mi355_row_size(type, k)) return MI355_E_SHAPE;
    if (m == 0 || n == 0) return MI355_OK;
    QArgs a;
    a.W = (const uint8_t *)W; a.w_row_stride = w_row_stride; a.aq = act_q; a.ad = act_d; a.bs = act_bsums;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.m = (int)m; a.n = (int)n; a.k = (int)k;
    const dim3 grid((unsigned)((n + QBN - 1) / QBN), (unsigned)((m + QBM - 1) / QBM));
    hipStream_t s = (hipStream_t)stream;
    if (type == MI355_TYPE_Q4_K) MI355_LAUNCH("mmqk_q4_K", (mmqk_kernel<MI355_TYPE_Q4_K>), grid, dim3(QTHREADS), 0, s, a);
    else                         MI355_LAUNCH("mmqk_q5_K", (mmqk_kernel<MI355_TYPE_Q5_K>), grid, dim3(QTHREADS), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}
