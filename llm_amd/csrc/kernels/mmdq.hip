// mmdq.hip -- prefill MUL_MAT on block-quantized weights as a dequantize-in-LDS bf16 MFMA GEMM.
//
// Same contract as mul_mat_q (ggml/src/ggml-cuda/mmq.cu:68-224): dst[n][m] = sum_k W[m][k] * x[n][k], f32 out,
// N > 8.  The int8 tile kernel (mmq.hip) follows the reference's MMQ arithmetic -- int8 x int8 -> i32 per
// 32-block, f32 rescale per block -- and that rescale is its bound on CDNA4: ~50 VALU instructions per
// 32x32 tile per 32-block against 32 cycles of MFMA, i.e. the matrix cores idle ~90 % of the time.
// Here the per-block scales are applied ONCE per weight while the tile is staged (each weight of the
// 128 x 64 tile is dequantized exactly as dequantize_row_* does, then rounded to bf16), the activations
// are rounded f32 -> bf16 once per distinct src1 (mi355_act_bf16; no q8_1 pass at all), and the inner loop is nothing but
// ds_read_b128 + v_mfma_f32_32x32x16_bf16 with f32 accumulation.  Error: two bf16 roundings per
// product (2^-9 relative each) ~ 3e-6 NMSE, well inside the 5e-4 budget of test-backend-ops and
// smaller than the q8 activation quantization of either reference backend.
//
// Roofline: bf16 MFMA (2.5 PFLOP/s dense).  flops = 2*M*N*K (tests/test-backend-ops.cpp:2262-2265).
//
// Tiling: workgroup = 256 threads = 4 waves, output tile 128 (weight rows) x 128 (activation columns),
// K step 64.  Each wave owns 64 x 64 = 2 x 2 MFMA tiles.  LDS rows have a 144-byte pitch (128 B of
// bf16 + 16) so the 16-byte operand reads of 32 consecutive rows spread over all banks.  LDS is double
// buffered with one barrier per stage: a wave dequantizes stage s+1 into the idle buffer and requests the
// global loads of stage s+2 before it issues the MFMAs of stage s.  (Fetching two stages ahead in a second
// register set was measured slower: 276-332 VGPRs drop the occupancy to one wave per SIMD.)
#include "common.h"
#include "moe_bucket.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace {

constexpr int BN = 128, BK = 64;   // BM (weight rows per workgroup) is a template parameter: 128, or 64 when M is small; BN the default column tile
constexpr int PITCH = BK * 2 + 16;   // bytes per LDS row

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct DArgs {
    const uint8_t * W; int64_t w_row_stride;
    const uint16_t * x; int64_t x_row_stride;  // bf16 activations (mi355_act_bf16), elements between columns
    float * dst; int64_t dst_col_stride;
    int m, n, k;
    // MUL_MAT_ID mode (IDS): columns are the (token, slot) pairs routed to one expert, listed by moe_bucket_kernel
    const int * sorted;      // [n_pairs] pair ids grouped by expert
    const int * off;         // [n_expert + 1] first entry of every expert in `sorted`
    const int * tile_off;    // [n_expert + 1] first column tile of every expert
    int n_expert, n_used, act_ne1;
    int64_t w_expert_stride;
    int dbg;                 // lab switch of mmdq_ws_kernel (MI355_MMDQ_DBG): 1 producers idle, 2 consumers idle
    float * part;            // split-K of the 128 x 256 role-split kernel: K slice z stores its partial tile at part[z][n][m], summed in z order by kz_sum_kernel
    // gate / up / GLU form (mmdq_ws_kernel<..., GLU>): W is the matrix whose product is activated, W2 the one it is multiplied by; m counts
    // PAIRS of rows (row i of W with row i of W2); dst receives act(W x) * (W2 x) as f32, hbf the same rounded to bf16 ([n][m], for the next GEMM)
    const uint8_t * W2; int64_t w2_row_stride; uint16_t * hbf; int glu_op;
    // pair form (mmdq_ws_kernel<..., TYPE2>): workgroups with blockIdx.y >= y2 compute a SECOND product, W2 (type TYPE2, m2 rows) times the
    // same activations, into part2 -- two small matrices (wk, wv) share one launch and the chip instead of taking turns on half of it
    int y2, m2; float * part2;
};

// 32 consecutive weights of one row, raw, as loaded from global memory
template <int TYPE> struct Raw;
template <> struct Raw<MI355_TYPE_Q4_K> { v4i q0, q1, hdr; };
template <> struct Raw<MI355_TYPE_Q5_K> { v4i q0, q1, h0, h1, hdr; };
template <> struct Raw<MI355_TYPE_Q6_K> { v4i l0, l1, h0, h1; uint32_t sc; uint16_t d; };
template <> struct Raw<MI355_TYPE_Q8_0> { v4i q0, q1; uint16_t d; };
template <> struct Raw<MI355_TYPE_Q4_0> { v4i qs; uint16_t d; };
template <> struct Raw<MI355_TYPE_Q4_1> { v4i qs; uint16_t d, m; };
template <> struct Raw<MI355_TYPE_Q5_0> { v4i qs; uint32_t qh; uint16_t d; };
template <> struct Raw<MI355_TYPE_Q5_1> { v4i qs; uint32_t qh; uint16_t d, m; };
template <> struct Raw<MI355_TYPE_Q2_K> { v4i q0, q1; uint16_t sc, d, dmin; };
template <> struct Raw<MI355_TYPE_Q3_K> { v4i q0, q1, h0, h1; uint16_t sl, sh, d; };
template <> struct Raw<MI355_TYPE_IQ4_NL> { v4i qs; uint16_t d; };
template <> struct Raw<MI355_TYPE_IQ4_XS> { v4i qs; uint16_t d, sh, sl; };
// float weights (an unquantized GGUF; K / V of a graph without flash attention): "dequantization" is the conversion to bf16
template <> struct Raw<MI355_TYPE_F16>  { v4i q[4]; };
template <> struct Raw<MI355_TYPE_BF16> { v4i q[4]; };
template <> struct Raw<MI355_TYPE_F32>  { v4i q[8]; };

template <int TYPE>
static __device__ __forceinline__ Raw<TYPE> load_raw(const uint8_t * wrow, int k) {   // k % 32 == 0
    Raw<TYPE> r;
    if constexpr (TYPE == MI355_TYPE_Q4_K) {
        const uint8_t * blk = wrow + (int64_t)(k >> 8) * 144;
        const int j = (k >> 5) & 7;
        r.q0 = *(const v4i *)(blk + 16 + (j >> 1) * 32);
        r.q1 = *(const v4i *)(blk + 32 + (j >> 1) * 32);
        r.hdr = *(const v4i *)blk;
    } else if constexpr (TYPE == MI355_TYPE_Q5_K) {
        const uint8_t * blk = wrow + (int64_t)(k >> 8) * 176;
        const int j = (k >> 5) & 7;
        r.q0 = *(const v4i *)(blk + 48 + (j >> 1) * 32);
        r.q1 = *(const v4i *)(blk + 64 + (j >> 1) * 32);
        r.h0 = *(const v4i *)(blk + 16);
        r.h1 = *(const v4i *)(blk + 32);
        r.hdr = *(const v4i *)blk;
    } else if constexpr (TYPE == MI355_TYPE_Q6_K) {
        // chunk = quarter i of half n: ql[64n + 32(i&1) + l] nibble (i>>1), qh[32n + l] >> 2i, sc[8n + 2i + l/16]
        const uint8_t * blk = wrow + (int64_t)(k >> 8) * 210;
        const int n = (k >> 7) & 1, i = (k >> 5) & 3;
        r.l0 = ld16_a2(blk + 64 * n + 32 * (i & 1));
        r.l1 = ld16_a2(blk + 64 * n + 32 * (i & 1) + 16);
        r.h0 = ld16_a2(blk + 128 + 32 * n);
        r.h1 = ld16_a2(blk + 128 + 32 * n + 16);
        r.sc = ld_u16(blk + 192 + 8 * n + 2 * i);
        r.d  = ld_u16(blk + 208);
    } else if constexpr (TYPE == MI355_TYPE_Q8_0) {
        const uint8_t * blk = wrow + (int64_t)(k >> 5) * 34;
        r.q0 = ld16_a2(blk + 2);
        r.q1 = ld16_a2(blk + 18);
        r.d  = ld_u16(blk);
    } else if constexpr (TYPE == MI355_TYPE_Q4_0) {
        const uint8_t * blk = wrow + (int64_t)(k >> 5) * 18;
        r.qs = ld16_a2(blk + 2);
        r.d  = ld_u16(blk);
    } else if constexpr (TYPE == MI355_TYPE_Q4_1) {
        const uint8_t * blk = wrow + (int64_t)(k >> 5) * 20;
        r.qs = ld16_a2(blk + 4);
        r.d  = ld_u16(blk); r.m = ld_u16(blk + 2);
    } else if constexpr (TYPE == MI355_TYPE_Q5_0) {
        const uint8_t * blk = wrow + (int64_t)(k >> 5) * 22;
        r.qs = ld16_a2(blk + 6);
        r.d  = ld_u16(blk); r.qh = (uint32_t)ld_u16(blk + 2) | ((uint32_t)ld_u16(blk + 4) << 16);
    } else if constexpr (TYPE == MI355_TYPE_Q5_1) {
        const uint8_t * blk = wrow + (int64_t)(k >> 5) * 24;
        r.qs = ld16_a2(blk + 8);
        r.d  = ld_u16(blk); r.m = ld_u16(blk + 2); r.qh = (uint32_t)ld_u16(blk + 4) | ((uint32_t)ld_u16(blk + 6) << 16);
    } else if constexpr (TYPE == MI355_TYPE_Q2_K) {
        // the 32 values at 128 n + 32 j: (qs[32 n + l] >> 2 j) & 3, scale bytes 8 n + 2 j (l < 16) and + 1 (ggml-quants.c:706-735)
        const uint8_t * blk = wrow + (int64_t)(k >> 8) * 84;
        const int n = (k >> 7) & 1, j = (k >> 5) & 3;
        r.q0 = ld16_a2(blk + 16 + 32 * n);
        r.q1 = ld16_a2(blk + 32 + 32 * n);
        r.sc = ld_u16(blk + 8 * n + 2 * j);
        r.d  = ld_u16(blk + 80); r.dmin = ld_u16(blk + 82);
    } else if constexpr (TYPE == MI355_TYPE_Q3_K) {
        // as q2_K plus bit 4 n + j of hmask[l]; scales is = 8 n + 2 j and is + 1: low nibbles in bytes is & 7 (and + 1), shifted by
        // 4 (is >> 3); high bits in bytes 8 + (is & 3) (and + 1), shifted by 2 (is >> 2) (ggml-quants.c:1050-1099)
        const uint8_t * blk = wrow + (int64_t)(k >> 8) * 110;
        const int n = (k >> 7) & 1, j = (k >> 5) & 3, is = 8 * n + 2 * j;
        r.q0 = ld16_a2(blk + 32 + 32 * n);
        r.q1 = ld16_a2(blk + 48 + 32 * n);
        r.h0 = ld16_a2(blk);
        r.h1 = ld16_a2(blk + 16);
        r.sl = ld_u16(blk + 96 + (is & 7));
        r.sh = ld_u16(blk + 104 + (is & 3));
        r.d  = ld_u16(blk + 108);
    } else if constexpr (TYPE == MI355_TYPE_IQ4_NL) {
        const uint8_t * blk = wrow + (int64_t)(k >> 5) * 18;
        r.qs = ld16_a2(blk + 2);
        r.d  = ld_u16(blk);
    } else if constexpr (TYPE == MI355_TYPE_IQ4_XS) {
        const uint8_t * blk = wrow + (int64_t)(k >> 8) * 136;
        const int ib = (k >> 5) & 7;
        r.qs = ld16_a2(blk + 8 + 16 * ib);
        r.d  = ld_u16(blk); r.sh = ld_u16(blk + 2); r.sl = ld_u16(blk + 4 + (ib >> 2) * 2);
    } else if constexpr (TYPE == MI355_TYPE_F32) {
#pragma unroll
        for (int i = 0; i < 8; i++) r.q[i] = ld16_a2(wrow + (int64_t)k * 4 + 16 * i);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) r.q[i] = ld16_a2(wrow + (int64_t)k * 2 + 16 * i);
    }
    return r;
}

static __device__ __forceinline__ void k4_one(const v4i hdr, int j, float & dsc, float & dm) {
    // get_scale_min_k4 (ggml-quants.c:625-632) for sub-block j, times d / dmin
    const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;
    const int g = j >> 1, sh = (g & 1) * 16 + (j & 1) * 8;
    uint32_t sc, m;
    if (g & 2) {
        sc = ((u2 >> sh) & 0x0fu) | (((u0 >> (sh + 6)) & 0x03u) << 4);
        m  = ((u2 >> (sh + 4)) & 0x0fu) | (((u1 >> (sh + 6)) & 0x03u) << 4);
    } else {
        sc = (u0 >> sh) & 0x3fu;
        m  = (u1 >> sh) & 0x3fu;
    }
    dsc = h2f((uint16_t)(hdr.x & 0xffff)) * (float)sc;
    dm  = h2f((uint16_t)((uint32_t)hdr.x >> 16)) * (float)m;
}

static __device__ __forceinline__ float ubyte(int w, int i) { return (float)((w >> (8 * i)) & 0xff); }
// kvalues_iq4nl, ggml-common.h:1077-1079
static __device__ __forceinline__ float iq4nl_f(int code) {
    constexpr uint64_t LO = 0xF6EADDCFBFAD9881ull, HI = 0x7159453526190D01ull;
    return (float)(int8_t)(((code & 8) ? HI : LO) >> (8 * (code & 7)));
}

// dequantize the 32 weights to f32 (same formulas as dequantize_row_*, ggml-quants.c)
template <int TYPE>
static __device__ __forceinline__ void dequant32(const Raw<TYPE> & r, int k, float * f) {
#pragma clang fp contract(fast)   // the result is rounded to bf16 anyway: one v_fma per weight instead of mul + sub
    if constexpr (TYPE == MI355_TYPE_Q4_K || TYPE == MI355_TYPE_Q5_K) {
        const int j = (k >> 5) & 7, sh = (j & 1) * 4;
        float dsc, dm;
        k4_one(r.hdr, j, dsc, dm);
#pragma unroll
        for (int w = 0; w < 8; w++) {
            int q = ((w < 4 ? r.q0[w] : r.q1[w - 4]) >> sh) & 0x0f0f0f0f;
            if constexpr (TYPE == MI355_TYPE_Q5_K) q |= (((w < 4 ? r.h0[w] : r.h1[w - 4]) >> j) & 0x01010101) << 4;
#pragma unroll
            for (int i = 0; i < 4; i++) f[4 * w + i] = dsc * ubyte(q, i) - dm;
        }
    } else if constexpr (TYPE == MI355_TYPE_Q6_K) {
        const int i2 = (k >> 5) & 3;
        const float d = h2f(r.d);
        const float s0 = d * (float)(int8_t)(r.sc & 0xff), s1 = d * (float)(int8_t)(r.sc >> 8);
        const float o0 = -32.0f * s0, o1 = -32.0f * s1;       // s (q - 32) as one fma per weight: s q - 32 s (exact: 32 s is a power-of-two multiple)
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int ql = ((w < 4 ? r.l0[w] : r.l1[w - 4]) >> (4 * (i2 >> 1))) & 0x0f0f0f0f;
            const int qh = (((w < 4 ? r.h0[w] : r.h1[w - 4]) >> (2 * i2)) & 0x03030303) << 4;
            const int q = ql | qh;
#pragma unroll
            for (int i = 0; i < 4; i++) f[4 * w + i] = (w < 4 ? s0 : s1) * ubyte(q, i) + (w < 4 ? o0 : o1);
        }
    } else if constexpr (TYPE == MI355_TYPE_Q8_0) {
        // d q with q signed: the byte read as unsigned is q + 128 (after flipping the sign bits), and v_cvt_f32_ubyteN converts it in one
        // instruction where a signed byte takes a bit-field extract and a convert: d (q + 128) - 128 d, one fma per weight
        const float d = h2f(r.d), o = -128.0f * d;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int q = (w < 4 ? r.q0[w] : r.q1[w - 4]) ^ (int)0x80808080;
#pragma unroll
            for (int i = 0; i < 4; i++) f[4 * w + i] = d * ubyte(q, i) + o;
        }
    } else if constexpr (TYPE == MI355_TYPE_Q4_0) {
        const float d = h2f(r.d);
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f[4 * w + i]      = d * (ubyte(r.qs[w] & 0x0f0f0f0f, i) - 8.0f);
                f[16 + 4 * w + i] = d * (ubyte((r.qs[w] >> 4) & 0x0f0f0f0f, i) - 8.0f);
            }
        }
    } else if constexpr (TYPE == MI355_TYPE_Q4_1 || TYPE == MI355_TYPE_Q5_0 || TYPE == MI355_TYPE_Q5_1) {
        // value = (nibble | 5th bit) * d + m; q5_0: (... - 16) * d (ggml-quants.c:269-341); bit e of qh belongs to element e
        const float d = h2f(r.d);
        float m = 0.0f, off = 0.0f;
        uint32_t qh = 0;
        if constexpr (TYPE != MI355_TYPE_Q5_0) m = h2f(r.m); else off = 16.0f;
        if constexpr (TYPE != MI355_TYPE_Q4_1) qh = r.qh;
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = 4 * w + i;
                const float lo = ubyte(r.qs[w] & 0x0f0f0f0f, i) + (float)(((qh >> e) & 1u) << 4);
                const float hi = ubyte((r.qs[w] >> 4) & 0x0f0f0f0f, i) + (float)(((qh >> (e + 16)) & 1u) << 4);
                f[e]      = d * (lo - off) + m;
                f[16 + e] = d * (hi - off) + m;
            }
        }
    } else if constexpr (TYPE == MI355_TYPE_Q2_K) {
        const int j = (k >> 5) & 3;
        const float d = h2f(r.d), dmin = h2f(r.dmin);
        const float d0 = d * (float)(r.sc & 0xf), m0 = dmin * (float)((r.sc >> 4) & 0xf);
        const float d1 = d * (float)((r.sc >> 8) & 0xf), m1 = dmin * (float)(r.sc >> 12);
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int q = ((w < 4 ? r.q0[w] : r.q1[w - 4]) >> (2 * j)) & 0x03030303;
#pragma unroll
            for (int i = 0; i < 4; i++) f[4 * w + i] = (w < 4 ? d0 : d1) * ubyte(q, i) - (w < 4 ? m0 : m1);
        }
    } else if constexpr (TYPE == MI355_TYPE_Q3_K) {
        const int n = (k >> 7) & 1, j = (k >> 5) & 3, is = 8 * n + 2 * j;
        const float d = h2f(r.d);
        const int lsh = 4 * (is >> 3), hsh = 2 * (is >> 2);
        const int s0 = (int)(((r.sl >> lsh) & 0xf) | (((r.sh >> hsh) & 3) << 4)) - 32;
        const int s1 = (int)(((r.sl >> (8 + lsh)) & 0xf) | (((r.sh >> (8 + hsh)) & 3) << 4)) - 32;
        const float d0 = d * (float)s0, d1 = d * (float)s1;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const int q = (((w < 4 ? r.q0[w] : r.q1[w - 4]) >> (2 * j)) & 0x03030303) | ((((w < 4 ? r.h0[w] : r.h1[w - 4]) >> (4 * n + j)) & 0x01010101) << 2);
#pragma unroll
            for (int i = 0; i < 4; i++) f[4 * w + i] = (w < 4 ? d0 : d1) * (ubyte(q, i) - 4.0f);
        }
    } else if constexpr (TYPE == MI355_TYPE_IQ4_NL || TYPE == MI355_TYPE_IQ4_XS) {
        float d = h2f(r.d);
        if constexpr (TYPE == MI355_TYPE_IQ4_XS) {
            const int ib = (k >> 5) & 7;
            const int ls = (int)(((r.sl >> (4 * (ib & 3))) & 0xf) | (((r.sh >> (2 * ib)) & 3) << 4));
            d *= (float)(ls - 32);
        }
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f[4 * w + i]      = d * iq4nl_f((r.qs[w] >> (8 * i)) & 0xf);
                f[16 + 4 * w + i] = d * iq4nl_f((r.qs[w] >> (8 * i + 4)) & 0xf);
            }
        }
    } else if constexpr (TYPE == MI355_TYPE_F16) {
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f[8 * w + 2 * i]     = h2f((uint16_t)((uint32_t)r.q[w][i] & 0xffff));
                f[8 * w + 2 * i + 1] = h2f((uint16_t)((uint32_t)r.q[w][i] >> 16));
            }
    } else if constexpr (TYPE == MI355_TYPE_BF16) {
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f[8 * w + 2 * i]     = __uint_as_float((uint32_t)r.q[w][i] << 16);
                f[8 * w + 2 * i + 1] = __uint_as_float((uint32_t)r.q[w][i] & 0xffff0000u);
            }
    } else {
#pragma unroll
        for (int w = 0; w < 8; w++)
#pragma unroll
            for (int i = 0; i < 4; i++) f[4 * w + i] = __int_as_float(r.q[w][i]);
    }
}

// 32 f32 -> 32 bf16 (round to nearest even), 64 bytes into LDS
static __device__ __forceinline__ void store_bf16x32(uint8_t * dst, const float * f) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (__bf16)f[8 * c + i];
        *(bf16x8 *)(dst + 16 * c) = v;
    }
}

// BNT: activation columns per workgroup, 128 or -- with 128 weight rows and 8 waves -- 256: every weight of the tile is
// dequantized once per BNT columns, so the wide tile halves the dequantization work per flop (the loop's limiter).
template <int TYPE, int BM, bool IDS, int BNT = 128>
__global__ void __launch_bounds__(BM == 256 || BNT == 256 ? 512 : 256)
mmdq_kernel(const DArgs a) {
    constexpr int BN = BNT;
    constexpr int NT = BM == 256 || BNT == 256 ? 512 : 256;    // threads: 8 waves for the 256-row and the 256-column tile, else 4
    constexpr int WAVES_M = BM / 64, WAVES_N = NT / 64 / WAVES_M; // every wave owns 64 weight rows x (BN / WAVES_N) columns
    constexpr int TN = BN / WAVES_N / 32;                      // 32-column MFMA tiles per wave: 1 (BM 64) or 2 (BM 128, 256)
    __shared__ __attribute__((aligned(16))) uint8_t Lw[2][BM * PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t La[2][BN * PITCH];
    __shared__ int s_pair[IDS ? BN : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (MUL_MAT_ID: grid (row tiles, column tiles), so that the column tiles no expert fills come last in dispatch order -- see mmdq_ws_kernel)
    const int n0 = (IDS ? blockIdx.y : blockIdx.x) * BN, m0 = (IDS ? blockIdx.x : blockIdx.y) * BM;
    const int wm_ = wave % WAVES_M, wn_ = wave / WAVES_M;
    const int lr = lane & 31, lh = lane >> 5;

    // staging role: row / column (tid >> 1), 32-value half (tid & 1)
    const int sr = tid >> 1, hf = tid & 1;
    const int wr = m0 + sr < a.m ? m0 + sr : a.m - 1;
    const uint8_t * wrow = a.W + (int64_t)wr * a.w_row_stride;
    bool col_ok;
    const uint16_t * xcol;
    if (IDS) {
        // column tile blockIdx.y -> (expert, tile inside the expert); the grid is sized for the worst case
        // (the expert = the number of experts whose tiles end at or before ti: table entries requested together and counted with a
        // ballot, not walked one dependent load at a time)
        const int ti = blockIdx.y;
        const int total_tiles = a.tile_off[a.n_expert];
        int e = 0;
        for (int b0 = 0; b0 < a.n_expert; b0 += 64) {
            const int end = b0 + lane < a.n_expert ? a.tile_off[b0 + lane + 1] : 0x7fffffff;
            e += __builtin_popcountll(__ballot(end <= ti));
        }
        if (ti >= total_tiles) return;
        const int base = a.off[e] + (ti - a.tile_off[e]) * BN, cnt = a.off[e + 1] - base;
        if (tid < BN) s_pair[tid] = tid < cnt ? a.sorted[base + tid] : -1;
        __syncthreads();
        const int pair = sr < BN ? s_pair[sr] : -1;
        col_ok = pair >= 0;
        const int tok = col_ok ? pair / a.n_used : 0, slot = col_ok ? pair % a.n_used : 0;
        xcol = a.x + (int64_t)(tok * a.act_ne1 + slot % a.act_ne1) * a.x_row_stride;
        wrow += (int64_t)e * a.w_expert_stride;
    } else {
        const int col = n0 + sr;
        col_ok = col < a.n;
        xcol = a.x + (int64_t)(col_ok ? col : 0) * a.x_row_stride;
    }

    const bool w_stager = sr < BM;              // BM = 64: the upper half of the workgroup stages activations only
    const bool a_stager = sr < BN;              // BM = 256 (512 threads): the upper half stages weights only
    v16f acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    Raw<TYPE> wraw;
    v4i xraw[4];
    auto fetch = [&](int k) {          // global -> registers: this thread's 32 weights and 32 activations of stage k
        if (w_stager) wraw = load_raw<TYPE>(wrow, k + hf * 32);
        if (a_stager) {
#pragma unroll
            for (int i = 0; i < 4; i++) xraw[i] = ((const v4i *)(xcol + k + hf * 32))[i];
        }
    };
    auto stash = [&](int k, int buf) { // registers -> LDS buffer `buf` (weights dequantized and rounded to bf16)
        if (w_stager) {
            float f[32];
            dequant32<TYPE>(wraw, k + hf * 32, f);
            store_bf16x32(Lw[buf] + sr * PITCH + hf * 64, f);
        }
        if (a_stager) {
            const v4i z = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; i++) *(v4i *)(La[buf] + sr * PITCH + hf * 64 + 16 * i) = col_ok ? xraw[i] : z;
        }
    };

    // Double-buffered LDS, one barrier per stage: while the MFMAs of stage s read buffer s&1, the same wave
    // dequantizes stage s+1 (fetched one iteration earlier) into the other buffer and requests stage s+2.
    // split-K (gridDim.z == 2): this workgroup covers half of K and adds its partial tile into a zeroed dst
    const int kbeg = (int)blockIdx.z * (a.k / (int)gridDim.z), kend = kbeg + a.k / (int)gridDim.z;
    fetch(kbeg);
    stash(kbeg, 0);
    if (kbeg + BK < kend) fetch(kbeg + BK);
    __syncthreads();
    for (int k0 = kbeg, buf = 0; k0 < kend; k0 += BK, buf ^= 1) {
        if (k0 + BK < kend) {
            stash(k0 + BK, buf ^ 1);
            if (k0 + 2 * BK < kend) fetch(k0 + 2 * BK);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            bf16x8 af[TN], wf[2];
#pragma unroll
            for (int t = 0; t < TN; t++) af[t] = *(const bf16x8 *)(La[buf] + (wn_ * 32 * TN + t * 32 + lr) * PITCH + ks * 32 + lh * 16);
#pragma unroll
            for (int t = 0; t < 2; t++)  wf[t] = *(const bf16x8 *)(Lw[buf] + (wm_ * 64 + t * 32 + lr) * PITCH + ks * 32 + lh * 16);
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tn], wf[tm], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

    // C register r of lane (lr, lh): activation column n = (r&3) + 8*(r>>2) + 4*lh, weight row m = lr
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        const int m = m0 + wm_ * 64 + tm * 32 + lr;
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int nl = wn_ * 32 * TN + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int n = IDS ? s_pair[nl] : n0 + nl;
                if (m < a.m && (IDS ? n >= 0 : n < a.n)) {
                    float * o = a.dst + (int64_t)n * a.dst_col_stride + m;
                    if (gridDim.z > 1) atomicAdd(o, acc[tm][tn][r]);   // two addends onto 0: the sum does not depend on the order
                    else *o = acc[tm][tn][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Role-split form of the 128 x 256 tile (768 threads).  In mmdq_kernel every wave first dequantizes its share of the next
// stage (VALU) and then issues its MFMAs, all in step between two barriers -- so the matrix pipe idles while the vector
// pipe works and the other way round (SQ counters on 14336 x 4096 x 512: MFMA busy 31 %, waves parked 39 %).  Here
//   * waves 0..7 multiply: a wave owns 64 x 64 of the tile (2 x 2 MFMA tiles), two such waves per SIMD cover each other's
//     LDS latency;
//   * waves 8..11 dequantize: 256 threads x 32 weights per stage (load, dequantize, round to bf16, write LDS), fetching two
//     stages ahead in a register ring -- they hold no accumulators, so the registers are there;
//   * the activation tile (bf16, a plain copy: 36 KB per stage, two thirds of the LDS writes and of the load instructions of
//     the uniform kernel) goes global -> LDS by DMA (global_load_lds_dwordx4: no registers, no VALU, no ds_write), issued
//     by the multiplying waves two stages ahead into a ring of three buffers.  A row of the LDS image is 8 chunks of data and
//     one of padding (144-byte pitch), so of the 64 lanes of a 1 KiB piece every ninth is masked off.
// Same arithmetic and per-element summation order as mmdq_kernel (K ascending in steps of 16): bit-identical results.
// Columns past n are not zero-filled (the DMA cannot): they read a valid column and their products are never stored.
// ------------------------------------------------------------------------------------------------
// the gated activation, the arithmetic of elementwise.hip's act_glu (ggml_silu_f32 vec.h:637-639, ggml_gelu_f32, relu)
static __device__ __forceinline__ float glu_act_dq(int op, float x) {
    switch (op) {
        case 0: return x > 0.0f ? x : 0.0f;
        case 1: { const float c = 0.044715f, s = 0.79788456080286535587989211986876f;
                  return 0.5f * x * (1.0f + tanhf(s * x * (1.0f + c * x * x))); }
        default: return x / (1.0f + expf(-x));
    }
}

static __device__ __forceinline__ void gemm_dma_piece(const void * gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// BMW x BNW: 128 x 256 (8 multiplying waves of 64 x 64, 4 dequantizing; wide launches) or 64 x 128 (4 multiplying waves of
// 64 x 32, 2 dequantizing; two workgroups per CU, with split-K for the launches of few tiles: gridDim.z == 2 adds the two
// halves of K into a zeroed dst, exactly as mmdq_kernel does).
// IDS (MUL_MAT_ID, prefill-sized batches): the tile's 256 columns are (token, slot) pairs routed to one expert, listed by
// moe_bucket_kernel with tiles of BNW pairs -- the activation DMA gathers their rows (every lane brings its own address), the weight rows
// are the expert's, the results scatter to the pairs' rows.
template <int TYPE, int BMW, int BNW, bool GLU = false, int TYPE2 = -1, bool IDS = false>
__global__ void __launch_bounds__(BMW * BNW == 128 * 256 ? 768 : 384)
mmdq_ws_kernel(const DArgs a_in) {
    static_assert(!GLU || (BMW == 128 && BNW == 256), "the gate / up / GLU form exists for the 128 x 256 tile");
    constexpr bool PAIR = TYPE2 >= 0;
    static_assert(!PAIR || !GLU, "pair and GLU forms are separate");
    static_assert(!IDS || (!PAIR && !GLU && BMW == 128 && BNW == 256), "the MUL_MAT_ID form exists for the plain 128 x 256 tile");
    const bool second = PAIR && (int)blockIdx.y >= a_in.y2;
    DArgs a = a_in;
    if (PAIR && second) { a.W = a_in.W2; a.w_row_stride = a_in.w2_row_stride; a.m = a_in.m2; a.part = a_in.part2; }
    // MUL_MAT_ID: the grid is (row tiles, column tiles) -- row tiles fastest, so that the column tiles no expert fills (the grid is sized for the
    // worst routing) are the END of the dispatch order.  With them interleaved (column tiles fastest) the 32 x 8 live workgroups of an 8x7B
    // ffn_down at 512 tokens did not land one per CU: 2.4 us per stage against 1.4 with 896 live workgroups (tools/moe_gemm_routing_lab.py shapes)
    const int by = IDS ? (int)blockIdx.x : PAIR && second ? (int)blockIdx.y - a_in.y2 : (int)blockIdx.y;
    const int bx = IDS ? (int)blockIdx.y : (int)blockIdx.x;
    constexpr int TN = BMW == 128 ? 2 : 1;                                    // 32-column MFMA tiles per multiplying wave
    constexpr int WAVES_M = BMW / 64, NMW = WAVES_M * (BNW / (32 * TN));      // multiplying waves
    constexpr int NPIECE = BNW * PITCH / 1024;                                // 1 KiB pieces per activation stage: 36 / 18
    constexpr int MAXP = (NPIECE + NMW - 1) / NMW;
    static_assert(BNW * PITCH % 1024 == 0 && MAXP <= BK / 16 + 1, "whole pieces; at most one per k-step plus one");
    static_assert((BMW == 128 && BNW == 256) || (BMW == 64 && BNW == 128), "tile");
    __shared__ __attribute__((aligned(1024))) uint8_t La[3][BNW * PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t Lw[2][BMW * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // GLU: the tile's 128 weight rows are 64 rows of W (tile rows 0..63) and the SAME 64 rows of W2 (tile rows 64..127)
    const int n0 = bx * BNW, m0 = by * (GLU ? BMW / 2 : BMW);
    const int kspan = a.k / (int)gridDim.z, kbeg = (int)blockIdx.z * kspan;
    const int nstage = kspan / BK;
    __shared__ int s_pair[IDS ? BNW : 1];
    int expert = 0;
    if constexpr (IDS) {
        // column tile bx -> (expert, tile inside the expert), as in mmdq_kernel; the grid is sized for the worst case
        const int ti = bx;
        const int total_tiles = a.tile_off[a.n_expert];
        for (int b0 = 0; b0 < a.n_expert; b0 += 64) {
            const int end = b0 + lane < a.n_expert ? a.tile_off[b0 + lane + 1] : 0x7fffffff;
            expert += __builtin_popcountll(__ballot(end <= ti));
        }
        if (ti >= total_tiles) return;                                // (the whole workgroup, before any barrier)
        const int base = a.off[expert] + (ti - a.tile_off[expert]) * BNW, cnt = a.off[expert + 1] - base;
        if (tid < BNW) s_pair[tid] = tid < cnt ? a.sorted[base + tid] : -1;
        __syncthreads();
    }

    if (wave >= NMW) {                                                // ---- dequantizing waves
        if (a.dbg & 8) return;                                        // lab: the multiplying waves run free (no stage barriers)
        const int pt = tid - 64 * NMW, sr = pt >> 1, hf = pt & 1;     // weight row sr of the tile, 32-value half hf of the stage
        const int srm = GLU ? (sr & 63) : sr;
        const int wr = m0 + srm < a.m ? m0 + srm : a.m - 1;
        const uint8_t * wrow = GLU && sr >= 64 ? a.W2 + (int64_t)wr * a.w2_row_stride
                                                : a.W + (IDS ? (int64_t)expert * a.w_expert_stride : 0) + (int64_t)wr * a.w_row_stride;
#ifndef MMDQ_PFD
#define MMDQ_PFD 2        // stages of raw weight bytes a dequantizing thread holds in flight
#endif
        constexpr int PFD = MMDQ_PFD;
        auto produce = [&](auto type_tag) {
            constexpr int T = decltype(type_tag)::value;
            Raw<T> wraw[PFD];
#pragma unroll
            for (int j = 0; j < PFD; j++) if (j < nstage) wraw[j] = load_raw<T>(wrow, kbeg + j * BK + hf * 32);
            auto stash = [&](int slot, int st) {
                float f[32];
                dequant32<T>(wraw[slot], kbeg + st * BK + hf * 32, f);
                store_bf16x32(Lw[st & 1] + sr * PITCH + hf * 64, f);
            };
            stash(0, 0);
            if (PFD < nstage) wraw[0] = load_raw<T>(wrow, kbeg + PFD * BK + hf * 32);
            __syncthreads();
            for (int st0 = 0; st0 < nstage; st0 += PFD) {
#pragma unroll
                for (int j = 0; j < PFD; j++) {
                    const int st = st0 + j;                   // the others multiply stage st; this writes stage st + 1 (slot (j + 1) % PFD)
                    if (st < nstage) {
                        if (st + 1 < nstage && !(a.dbg & 1)) {
                            stash((j + 1) % PFD, st + 1);
                            if (st + 1 + PFD < nstage) wraw[(j + 1) % PFD] = load_raw<T>(wrow, kbeg + (st + 1 + PFD) * BK + hf * 32);
                        }
                        __syncthreads();
                    }
                }
            }
        };
        if constexpr (PAIR && TYPE2 != TYPE) {
            if (second) produce(std::integral_constant<int, TYPE2>{}); else produce(std::integral_constant<int, TYPE>{});
        } else {
            produce(std::integral_constant<int, TYPE>{});
        }
        return;
    }

    // ---- multiplying waves; they also move the activation tile
    const int wm_ = wave % WAVES_M, wn_ = wave / WAVES_M;   // 64 weight rows x 32 TN columns of the tile
    const int lr = lane & 31, lh = lane >> 5;
    // this wave's pieces of a stage's activation image: piece p = wave + NMW i; lane l copies chunk c = 64 p + l: row c / 9,
    // 16-byte chunk c % 9 of its 144-byte pitch (chunk 8 is padding: lane off)
    const uint16_t * src[MAXP];
    bool on[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; i++) {
        const int p = wave + NMW * i, c = 64 * p + lane, r = (c * 7282) >> 16, j = c - 9 * r;
        int col;
        if constexpr (IDS) {       // the pair's activation row; slots past the expert's count re-read the tile's first pair, their products are never stored
            int pair = r < BNW ? s_pair[r] : -1;
            if (pair < 0) pair = s_pair[0];
            col = (pair / a.n_used) * a.act_ne1 + (pair % a.n_used) % a.act_ne1;
        } else {
            col = n0 + r < a.n ? n0 + r : a.n - 1;
        }
        on[i]  = p < NPIECE && j < 8;
        src[i] = a.x + (int64_t)col * a.x_row_stride + kbeg + j * 8;
    }
    const unsigned la0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)&La[0][0];
    auto dma_piece_i = [&](int st, int i) {                 // piece wave + NMW i of stage st -> La[st % 3]
        if (i < MAXP && wave + NMW * i < NPIECE) {          // wave-uniform
            const unsigned base = la0 + (unsigned)(st % 3) * (BNW * PITCH);
            if (on[i]) gemm_dma_piece(src[i] + (int64_t)st * BK, __builtin_amdgcn_readfirstlane(base + (unsigned)(wave + NMW * i) * 1024));
        }
    };
    auto dma_stage = [&](int st) {
#pragma unroll
        for (int i = 0; i < MAXP; i++) dma_piece_i(st, i);
    };
    v16f acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    dma_stage(0);
    if (nstage > 1) dma_stage(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int KS = BK / 16;
    for (int st = 0; st < nstage; st++) {
        const uint8_t * lw = Lw[st & 1], * la = La[st % 3];
        const bool more = st + 2 < nstage && !(a.dbg & 4);  // stage st + 2 goes out now; its buffer was last read in stage st - 1
        bf16x8 af[2][TN], wf[2][2];                         // fragments of k-step ks in set ks & 1: the next step's are read before this step's MFMAs
        auto frags = [&](int ks, int set) {
#pragma unroll
            for (int t = 0; t < 2; t++)  wf[set][t] = *(const bf16x8 *)(lw + (wm_ * 64 + t * 32 + lr) * PITCH + ks * 32 + lh * 16);
#pragma unroll
            for (int t = 0; t < TN; t++) af[set][t] = *(const bf16x8 *)(la + (wn_ * 32 * TN + t * 32 + lr) * PITCH + ks * 32 + lh * 16);
        };
        if (!(a.dbg & 2)) frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (a.dbg & 2) { if (more) { dma_piece_i(st + 2, ks); if (ks == 0) dma_piece_i(st + 2, KS); } continue; }
            if (ks + 1 < KS) frags(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][tn], wf[ks & 1][tm], acc[tm][tn], 0, 0, 0);
            // the DMA requests ride behind the MFMAs just issued (one piece per k-step, the fifth with the first)
            if (more) { dma_piece_i(st + 2, ks); if (ks == 0) dma_piece_i(st + 2, KS); }
        }
        // stage st + 1 was requested a whole stage ago; what this iteration requested (stage st + 2) may stay in flight
        if (st + 2 < nstage) {
            if (wave + NMW * (MAXP - 1) < NPIECE) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MAXP) : "memory");
            else                                  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MAXP - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(a.dbg & 8)) __syncthreads();
    }
    if constexpr (GLU) {
        // waves wm_ = 0 hold (W x) for the tile's 64 row pairs, waves wm_ = 1 hold (W2 x) for the same rows.  They swap halves through
        // LDS (the stage buffers are dead; the dequantizing waves have left, the barrier counts the remaining ones; same (register,
        // lane) slot on both sides): wm_ = 0 keeps rows 0..31 and receives their (W2 x), wm_ = 1 keeps rows 32..63 and receives their
        // (W x) -- so both wave sets do half of the activation arithmetic and of the stores.
        float * X = (float *)&La[0][0] + (size_t)(wn_ * 2 + wm_) * TN * 16 * 64;
        const float * Xo = (const float *)&La[0][0] + (size_t)(wn_ * 2 + (wm_ ^ 1)) * TN * 16 * 64;
        const int give = wm_ ^ 1, keep = wm_;             // tile-row block (tm) handed over / kept
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) X[(tn * 16 + r) * 64 + lane] = give ? acc[1][tn][r] : acc[0][tn][r];
        __syncthreads();
        const int m = m0 + keep * 32 + lr;
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + wn_ * 32 * TN + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float other = Xo[(tn * 16 + r) * 64 + lane];
                const float mine = keep ? acc[1][tn][r] : acc[0][tn][r];
                const float g = wm_ == 0 ? mine : other, u = wm_ == 0 ? other : mine;
                const float h = glu_act_dq(a.glu_op, g) * u;
                // two neighbouring rows share one 4-byte store of the bf16 copy (m even: a.m is even here)
                const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)h);
                const uint32_t nb_ = (uint32_t)__shfl_xor((int)hb, 1, 64);
                if (m < a.m && n < a.n) {
                    a.dst[(int64_t)n * a.dst_col_stride + m] = h;
                    if (!(lr & 1)) *(uint32_t *)(a.hbf + (int64_t)n * a.m + m) = hb | (nb_ << 16);
                }
            }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        const int m = m0 + wm_ * 64 + tm * 32 + lr;
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int nl = wn_ * 32 * TN + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int n = IDS ? s_pair[nl] : n0 + nl;
                if (m < a.m && (IDS ? n >= 0 : n < a.n)) {
                    float * o = a.dst + (int64_t)n * a.dst_col_stride + m;
                    if (a.part)            a.part[((int64_t)blockIdx.z * a.n + n) * a.m + m] = acc[tm][tn][r];
                    else if (gridDim.z > 1) atomicAdd(o, acc[tm][tn][r]);   // two addends onto 0: the sum does not depend on the order
                    else *o = acc[tm][tn][r];
                }
            }
    }
}

// dst[n][m] = ((part[0] + part[1]) + part[2]) + ...: the K slices of a split-K launch, added in slice order.  VEC: dst is as dense as
// the partial planes ([n][m], m % 4 == 0): four elements per lane, no index arithmetic.
template <bool VEC>
__global__ void __launch_bounds__(256)
kz_sum_kernel(const float * part, float * dst, int64_t dst_col_stride, int m, int n, int kz) {
    const int64_t total = (int64_t)m * n, plane = total;
    if (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
            v4f v = *(const v4f *)(part + i);
            for (int z = 1; z < kz; z++) v = v + *(const v4f *)(part + z * plane + i);
            *(v4f *)(dst + i) = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            float v = part[i];
            for (int z = 1; z < kz; z++) v += part[z * plane + i];
            dst[(i / m) * dst_col_stride + i % m] = v;
        }
    }
}

// f32 (4-D, strided rows) -> bf16 [rows][k], round to nearest even; one lane per 8 values
__global__ void __launch_bounds__(256)
act_bf16_kernel(const char * x, uint16_t * y, int k, int ne1, int ne2, int64_t nb1, int64_t nb2, int64_t nb3) {
    const int row = blockIdx.x;
    const int i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * xr = (const float *)(x + i1 * nb1 + i2 * nb2 + i3 * nb3);
    for (int c = blockIdx.y * 256 + threadIdx.x; c * 8 < k; c += gridDim.y * 256) {
        const v4f a = ((const v4f *)xr)[2 * c], b = ((const v4f *)xr)[2 * c + 1];
        bf16x8 v;
        v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
        v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
        *(bf16x8 *)(y + (int64_t)row * k + 8 * c) = v;
    }
}

// K slices of the 128 x 256 role-split kernel for a matrix of m_total rows: 1 when its tiles fill the chip, 2 / 4 when a half / a
// quarter / an eighth of that, 0 when the launch does not take that kernel's split-K form at all
static int kz_slices(int64_t m_total, int64_t n, int64_t k) {
    static const int kz_ws = [] { const char * e = getenv("MI355_MMDQ_KZ"); return e ? atoi(e) : 1; }();
    static const int wide = [] { const char * e = getenv("MI355_MMDQ_WIDE"); return e ? atoi(e) : 2; }();
    const int64_t tiles = ((n + 255) / 256) * ((m_total + 127) / 128);
    int kzw = tiles >= 200 ? 1 : 2 * tiles >= 200 ? 2 : 4 * tiles >= 200 ? 4 : 8 * tiles >= 200 ? 8 : 0;
    // more than 128 columns: a partly empty 256-column tile still beats two 128-column ones (14336 x 4096 x 192: 69 -> 48 us);
    // 8 slices where K is long enough (4096 x 14336 x 256: 32 tiles; the 64 x 128 tile took 73 us there)
    if (kzw > 1 && (!kz_ws || wide != 2 || n <= 128 || k % (BK * kzw) != 0 || k / kzw < 1024)) kzw = 0;
    return kzw;
}

// K slices of the 64 x 128 role-split kernel, the tile of the launches too small for the one above (M = 1024 at N = 512: wk, wv): 2 or 4
// while every workgroup still finds a CU of its own; 0: not that form (one K range, or the two-addend atomic form of the uniform kernel)
static int kz_slices_small(int64_t m_total, int64_t n, int64_t k) {
    static const int kz_ws = [] { const char * e = getenv("MI355_MMDQ_KZ"); return e ? atoi(e) : 1; }();
    static const int wide = [] { const char * e = getenv("MI355_MMDQ_WIDE"); return e ? atoi(e) : 2; }();
    static const int small_ws = [] { const char * e = getenv("MI355_MMDQ_SMALL"); return e ? atoi(e) : 1; }();
    if (!kz_ws || wide != 2 || small_ws != 1 || kz_slices(m_total, n, k) != 0) return 0;
    const int64_t tiles = ((n + BN - 1) / BN) * ((m_total + 63) / 64);
    for (int kz = 4; kz >= 2; kz /= 2)
        if (tiles * kz <= 256 && k % (BK * kz) == 0 && k / kz >= 1024) return kz;
    return 0;
}

// m_total: rows of the WHOLE weight matrix when `a` is a row slice of it (row split).  Whether K is cut in two -- the one
// choice here that changes the f32 summation order -- is decided from the whole matrix, so a slice produces the bits the
// unsplit matrix would; the tile height only changes which workgroup computes an element, not how.
template <int TYPE>
static int launch(const DArgs & a_in, int64_t m_total, hipStream_t s, const char * tag, int * parts_kz = nullptr) {
    // parts_kz: the caller wants the K-slice planes, not their sum -- a.part receives [kz][n][m] (kz == 1: a.dst is that plane)
    DArgs a = a_in;
    // 128-row tiles reuse each staged activation tile twice as often; 64-row tiles when those would leave CUs idle
    const int64_t nb = (a.n + BN - 1) / BN;
    const bool full_big = nb * ((m_total + 255) / 256) >= 200 || nb * ((m_total + 127) / 128) >= 384;
    // few tiles (M <= ~6K at N = 512): also split K in two so that every CU holds two workgroups.  The halves are
    // added into a zeroed dst with float atomics; with exactly two addends the result is order-independent.
    const unsigned kz = !full_big && nb * ((m_total + 63) / 64) < 448 && a.k % (2 * BK) == 0 && a.k >= 1024 ? 2 : 1;
    // The loop is bound by the CU-side load path (~10 B/cycle/CU): every workgroup re-reads its activation tile, so
    // the bytes moved per flop fall with the tile height.  256-row tiles (8 waves) when they still cover the chip.
    static const int wide = [] { const char * e = getenv("MI355_MMDQ_WIDE"); return e ? atoi(e) : 2; }();   // 0: 256 x 128 tile, 1: 128 x 256 uniform, 2: 128 x 256 role-split
    static const int small_ws = [] { const char * e = getenv("MI355_MMDQ_SMALL"); return e ? atoi(e) : 1; }();
    const int64_t nb2 = (a.n + 255) / 256;
    const bool dma_ok = ((uintptr_t)a.x & 15) == 0 && a.x_row_stride % 8 == 0;      // 16-byte chunks of the activation rows
    // 128 x 256 tiles of the WHOLE matrix: enough of them for the chip -> one K range; a half / a quarter of that -> K in 2 / 4
    // slices (M = 4096..8192 at N = 512), each slice's partial tile to the workspace, summed in slice order.  The small tiles such
    // launches used to take move 2-3x the activation bytes per flop (4096 x 14336 x 512: 143 -> 8x us).
    const int kzb = kz_slices(m_total, a.n, a.k), kzs = kz_slices_small(m_total, a.n, a.k);
    const int kzw = kzb > 1 ? kzb : kzs;
    if (kzw > 1 && dma_ok) {
        DArgs b = a;
        // the split-K forms need the caller's workspace (mi355_mul_mat_dq_workspace_bytes): this library allocates nothing
        if (!b.part) return MI355_E_WORKSPACE;
        if (kzb > 1) MI355_LAUNCH(tag, (mmdq_ws_kernel<TYPE, 128, 256>), dim3((unsigned)nb2, (unsigned)((a.m + 127) / 128), (unsigned)kzw), dim3(768), 0, s, b);
        else         MI355_LAUNCH(tag, (mmdq_ws_kernel<TYPE, 64, 128>), dim3((unsigned)nb, (unsigned)((a.m + 63) / 64), (unsigned)kzw), dim3(384), 0, s, b);
        MI355_CHECK_LAUNCH();
        if (parts_kz) { *parts_kz = kzw; return MI355_OK; }
        const int64_t total = (int64_t)a.m * a.n;
        if (a.dst_col_stride == a.m && a.m % 4 == 0 && ((uintptr_t)a.dst & 15) == 0 && ((uintptr_t)b.part & 15) == 0)
            MI355_LAUNCH("mmdq_kz_sum", kz_sum_kernel<true>, dim3((unsigned)std::min<int64_t>((total / 4 + 255) / 256, 2048)), dim3(256), 0, s, b.part, a.dst, a.dst_col_stride, a.m, a.n, kzw);
        else
            MI355_LAUNCH("mmdq_kz_sum", kz_sum_kernel<false>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, b.part, a.dst, a.dst_col_stride, a.m, a.n, kzw);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    a.part = nullptr;             // a caller's workspace is only for the split-K forms above
    if (parts_kz) *parts_kz = 1;
    if (wide == 2 && kz == 1 && a.n > 128 && nb2 * ((a.m + 127) / 128) >= 200 && dma_ok) {
        MI355_LAUNCH(tag, (mmdq_ws_kernel<TYPE, 128, 256>), dim3((unsigned)nb2, (unsigned)((a.m + 127) / 128)), dim3(768), 0, s, a);
    } else if (wide && kz == 1 && a.n >= 256 && nb2 * ((a.m + 127) / 128) >= 200) {
        MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 128, false, 256>), dim3((unsigned)nb2, (unsigned)((a.m + 127) / 128)), dim3(512), 0, s, a);
    } else if (kz == 1 && nb * ((a.m + 255) / 256) >= 200) {
        MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 256, false>), dim3((unsigned)nb, (unsigned)((a.m + 255) / 256)), dim3(512), 0, s, a);
    } else if (kz == 1 && nb * ((a.m + 127) / 128) >= 384) {
        MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 128, false>), dim3((unsigned)nb, (unsigned)((a.m + 127) / 128)), dim3(256), 0, s, a);
    } else {
        if (kz > 1) {
            hipError_t e = a.dst_col_stride == a.m
                ? hipMemsetAsync(a.dst, 0, (size_t)a.n * a.m * sizeof(float), s)
                : hipMemset2DAsync(a.dst, (size_t)a.dst_col_stride * sizeof(float), 0, (size_t)a.m * sizeof(float), (size_t)a.n, s);
            if (e != hipSuccess) return (int)e;
        }
        // role-split 64 x 128 tiles where the launch cannot fill the chip anyway (<= one workgroup per CU: M = 1024 at N = 512 runs
        // 25 us instead of 32); with two resident workgroups per CU the uniform kernel's waves cover each other better (M = 4096: 51 vs 56 us)
        static const int mid = [] { const char * e = getenv("MI355_MMDQ_MID"); return e ? atoi(e) : 0; }();
        if (mid && kz == 2 && a.m >= 2048)
            MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 128, false>), dim3((unsigned)nb, (unsigned)((a.m + 127) / 128), kz), dim3(256), 0, s, a);
        else if (wide == 2 && dma_ok && (small_ws == 2 || (small_ws == 1 && nb * ((a.m + 63) / 64) * kz <= 256)))
            MI355_LAUNCH(tag, (mmdq_ws_kernel<TYPE, 64, 128>), dim3((unsigned)nb, (unsigned)((a.m + 63) / 64), kz), dim3(384), 0, s, a);
        else
            MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 64, false>), dim3((unsigned)nb, (unsigned)((a.m + 63) / 64), kz), dim3(256), 0, s, a);
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// ---- MUL_MAT_ID for prefill-sized batches: route on the device (moe_bucket.h), then one GEMM launch over all experts.
// the role-split 128 x 256 tile for MUL_MAT_ID (tiles of 256 pairs; the five weight types a MoE GGUF is made of)
static bool ids_wide(int type, int64_t m) {
    static const int on = [] { const char * e = getenv("MI355_MMDQ_ID_WIDE"); return e ? atoi(e) : 1; }();
    return on && m >= 1024 && (type == MI355_TYPE_Q4_0 || type == MI355_TYPE_Q8_0 || type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K || type == MI355_TYPE_Q6_K);
}

template <int TYPE>
static int launch_ids(const DArgs & a_in, int max_tiles, hipStream_t s, const char * tag) {
    const DArgs & a = a_in;
    if constexpr (TYPE == MI355_TYPE_Q4_0 || TYPE == MI355_TYPE_Q8_0 || TYPE == MI355_TYPE_Q4_K || TYPE == MI355_TYPE_Q5_K || TYPE == MI355_TYPE_Q6_K) {
        if (ids_wide(TYPE, a.m)) {
            DArgs a = a_in;       // (the lab switch of the role-split kernel, as in launch())
            { static const int dbg = [] { const char * e = getenv("MI355_MMDQ_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
            MI355_LAUNCH(tag, (mmdq_ws_kernel<TYPE, 128, 256, false, -1, true>), dim3((unsigned)((a.m + 127) / 128), (unsigned)max_tiles), dim3(768), 0, s, a);
            MI355_CHECK_LAUNCH();
            return MI355_OK;
        }
    }
    if (a.m >= 1024) MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 128, true>), dim3((unsigned)((a.m + 127) / 128), (unsigned)max_tiles), dim3(256), 0, s, a);
    else             MI355_LAUNCH(tag, (mmdq_kernel<TYPE, 64, true>), dim3((unsigned)((a.m + 63) / 64), (unsigned)max_tiles), dim3(256), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

} // namespace

extern "C" int mi355_act_bf16(const float * x, uint16_t * y, int64_t k, int64_t ne1, int64_t ne2, int64_t ne3,
                              int64_t nb1, int64_t nb2, int64_t nb3, void * stream) {
    if (!x || !y) return MI355_E_NULL;
    if (k <= 0 || k % 8 != 0 || ne1 < 0 || ne2 < 0 || ne3 < 0 || k > 0x7fffffff) return MI355_E_SHAPE;
    const int64_t rows = ne1 * ne2 * ne3;
    if (rows > 0x7fffffff) return MI355_E_SHAPE;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)nb1 | (uintptr_t)nb2 | (uintptr_t)nb3) & 15) return MI355_E_ALIGN;
    if (rows == 0) return MI355_OK;
    const unsigned gy = (unsigned)((k / 8 + 255) / 256);
    MI355_LAUNCH("act_bf16", act_bf16_kernel, dim3((unsigned)rows, gy < 64 ? gy : 64), dim3(256), 0, (hipStream_t)stream,
                 (const char *)x, y, (int)k, (int)ne1, (int)ne2, nb1, nb2, nb3);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_mul_mat_dq(int type, const void * W, int64_t w_row_stride, const uint16_t * x, int64_t x_row_stride,
                                float * dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n, void * stream) {
    return mi355_mul_mat_dq_slice(type, W, w_row_stride, x, x_row_stride, dst, dst_col_stride, m, m, k, n, stream);
}

extern "C" int mi355_mul_mat_dq_slice(int type, const void * W, int64_t w_row_stride, const uint16_t * x, int64_t x_row_stride,
                                      float * dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n, void * stream) {
    return mi355_mul_mat_dq_slice_ws(type, W, w_row_stride, x, x_row_stride, dst, dst_col_stride, m, m_total, k, n, nullptr, 0, stream);
}

extern "C" int64_t mi355_mul_mat_dq_workspace_bytes(int64_t m, int64_t m_total, int64_t k, int64_t n) {
    if (m < 0 || m_total < m || k <= 0 || n < 0) return -1;
    const int kzb = k % BK == 0 ? kz_slices(m_total, n, k) : 0;
    const int kzw = kzb > 1 || k % BK != 0 ? kzb : kz_slices_small(m_total, n, k);
    return kzw > 1 ? (int64_t)kzw * n * m * (int64_t)sizeof(float) : 0;
}

static int dq_run(int type, const void * W, int64_t w_row_stride, const uint16_t * x, int64_t x_row_stride,
                  float * dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n,
                  void * workspace, int64_t workspace_bytes, void * stream, int * parts_kz);

extern "C" int mi355_mul_mat_dq_slice_ws(int type, const void * W, int64_t w_row_stride, const uint16_t * x, int64_t x_row_stride,
                                         float * dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n,
                                         void * workspace, int64_t workspace_bytes, void * stream) {
    return dq_run(type, W, w_row_stride, x, x_row_stride, dst, dst_col_stride, m, m_total, k, n, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int64_t mi355_mul_mat_dq_parts_bytes(int64_t m, int64_t m_total, int64_t k, int64_t n) {
    const int64_t ws = mi355_mul_mat_dq_workspace_bytes(m, m_total, k, n);
    if (ws < 0) return ws;
    return ws > 0 ? ws : n * m * (int64_t)sizeof(float);
}

extern "C" int mi355_mul_mat_dq_parts(int type, const void * W, int64_t w_row_stride, const uint16_t * x, int64_t x_row_stride,
                                      float * parts, int64_t parts_bytes, int64_t m, int64_t m_total, int64_t k, int64_t n,
                                      int32_t * kz, void * stream) {
    if (!parts || !kz) return MI355_E_NULL;
    if (((uintptr_t)parts & 15) || parts_bytes < mi355_mul_mat_dq_parts_bytes(m, m_total, k, n)) return MI355_E_WORKSPACE;
    int nz = 0;
    // the buffer is both the split-K workspace and, for the launches that do not split, the product itself
    const int rc = dq_run(type, W, w_row_stride, x, x_row_stride, parts, m, m, m_total, k, n, parts, parts_bytes, stream, &nz);
    if (rc == MI355_OK) *kz = m == 0 || n == 0 ? 1 : nz;
    return rc;
}

static int dq_run(int type, const void * W, int64_t w_row_stride, const uint16_t * x, int64_t x_row_stride,
                  float * dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n,
                  void * workspace, int64_t workspace_bytes, void * stream, int * parts_kz) {
    if (!W || !x || !dst) return MI355_E_NULL;
    if (!parts_kz && workspace && (((uintptr_t)workspace & 15) || workspace_bytes < mi355_mul_mat_dq_workspace_bytes(m, m_total, k, n))) return MI355_E_WORKSPACE;
    if (m_total < m) return MI355_E_SHAPE;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0 || type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (m < 0 || n < 0 || k <= 0 || k % bs != 0 || m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff) return MI355_E_SHAPE;
    if (k % BK != 0) return MI355_E_UNSUPPORTED;          // caller falls back to the mat-vec kernel
    if ((n + BN - 1) / BN > 0x7fffffff / 2 || (m + 63) / 64 > 65535) return MI355_E_UNSUPPORTED;
    if (((uintptr_t)x & 15) || (x_row_stride & 7) || ((uintptr_t)dst & 3) || x_row_stride < k) return MI355_E_ALIGN;
    const bool a16 = type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K;
    if (((uintptr_t)W | (uintptr_t)w_row_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (w_row_stride < mi355_row_size(type, k)) return MI355_E_SHAPE;
    if (m == 0 || n == 0) return MI355_OK;
    DArgs a;
    a.W = (const uint8_t *)W; a.w_row_stride = w_row_stride; a.x = x; a.x_row_stride = x_row_stride;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.m = (int)m; a.n = (int)n; a.k = (int)k;
    a.sorted = a.off = a.tile_off = nullptr; a.n_expert = a.n_used = a.act_ne1 = 0; a.w_expert_stride = 0; a.part = (float *)workspace;
    a.W2 = nullptr; a.w2_row_stride = 0; a.hbf = nullptr; a.glu_op = 0; a.y2 = 0; a.m2 = 0; a.part2 = nullptr;
    { static const int dbg = [] { const char * e = getenv("MI355_MMDQ_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
        case MI355_TYPE_Q4_0: return launch<MI355_TYPE_Q4_0>(a, m_total, s, "mmdq_q4_0", parts_kz);
        case MI355_TYPE_Q8_0: return launch<MI355_TYPE_Q8_0>(a, m_total, s, "mmdq_q8_0", parts_kz);
        case MI355_TYPE_Q4_K: return launch<MI355_TYPE_Q4_K>(a, m_total, s, "mmdq_q4_K", parts_kz);
        case MI355_TYPE_Q5_K: return launch<MI355_TYPE_Q5_K>(a, m_total, s, "mmdq_q5_K", parts_kz);
        case MI355_TYPE_Q6_K: return launch<MI355_TYPE_Q6_K>(a, m_total, s, "mmdq_q6_K", parts_kz);
        case MI355_TYPE_Q4_1: return launch<MI355_TYPE_Q4_1>(a, m_total, s, "mmdq_q4_1", parts_kz);
        case MI355_TYPE_Q5_0: return launch<MI355_TYPE_Q5_0>(a, m_total, s, "mmdq_q5_0", parts_kz);
        case MI355_TYPE_Q5_1: return launch<MI355_TYPE_Q5_1>(a, m_total, s, "mmdq_q5_1", parts_kz);
        case MI355_TYPE_Q2_K: return launch<MI355_TYPE_Q2_K>(a, m_total, s, "mmdq_q2_K", parts_kz);
        case MI355_TYPE_Q3_K: return launch<MI355_TYPE_Q3_K>(a, m_total, s, "mmdq_q3_K", parts_kz);
        case MI355_TYPE_IQ4_NL: return launch<MI355_TYPE_IQ4_NL>(a, m_total, s, "mmdq_iq4_nl", parts_kz);
        case MI355_TYPE_IQ4_XS: return launch<MI355_TYPE_IQ4_XS>(a, m_total, s, "mmdq_iq4_xs", parts_kz);
        case MI355_TYPE_F16:  return launch<MI355_TYPE_F16>(a, m_total, s, "mmdq_f16", parts_kz);
        case MI355_TYPE_BF16: return launch<MI355_TYPE_BF16>(a, m_total, s, "mmdq_bf16", parts_kz);
        case MI355_TYPE_F32:  return launch<MI355_TYPE_F32>(a, m_total, s, "mmdq_f32", parts_kz);
    }
    return MI355_E_TYPE;
}

template <int TYPE>
static int launch_glu(const DArgs & a, hipStream_t s, const char * tag) {
    MI355_LAUNCH(tag, (mmdq_ws_kernel<TYPE, 128, 256, true>), dim3((unsigned)((a.n + 255) / 256), (unsigned)((a.m + 63) / 64)), dim3(768), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// dst[n][m] = act(Wa x)[n][m] * (Wb x)[n][m] and the same rounded to bf16: ffn_gate, ffn_up and the GLU between them in one launch.
// Offered only where mi355_mul_mat_dq would run each of the two products as ONE K range of 128 x 256 role-split tiles: every
// element is then accumulated by the same MFMA sequence as there, so the result has the bits of the three separate launches.
extern "C" int mi355_glu_mul_mat_dq(int type, int glu_op, const void * Wa, int64_t wa_row_stride, const void * Wb, int64_t wb_row_stride,
                                    const uint16_t * x, int64_t x_row_stride, float * dst, int64_t dst_col_stride, uint16_t * dst_bf16,
                                    int64_t m, int64_t k, int64_t n, void * stream) {
    if (!Wa || !Wb || !x || !dst || !dst_bf16) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0 || type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (glu_op < 0 || glu_op > 2) return MI355_E_TYPE;
    if (m < 0 || n < 0 || k <= 0 || k % bs != 0 || m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff) return MI355_E_SHAPE;
    if (((uintptr_t)x & 15) || (x_row_stride & 7) || ((uintptr_t)dst & 3) || ((uintptr_t)dst_bf16 & 3) || x_row_stride < k) return MI355_E_ALIGN;
    const bool a16 = type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K;
    if (((uintptr_t)Wa | (uintptr_t)wa_row_stride | (uintptr_t)Wb | (uintptr_t)wb_row_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (wa_row_stride < mi355_row_size(type, k) || wb_row_stride < mi355_row_size(type, k) || dst_col_stride < m) return MI355_E_SHAPE;
    static const int wide = [] { const char * e = getenv("MI355_MMDQ_WIDE"); return e ? atoi(e) : 2; }();
    // the launch mi355_mul_mat_dq would choose for ONE of the two products (launch<TYPE>() above) must be the single-K-range role-split one
    if (k % BK != 0 || m % 2 != 0 || wide != 2 || n < 256 || ((n + 255) / 256) * ((m + 127) / 128) < 200 || (m + 63) / 64 > 65535)
        return MI355_E_UNSUPPORTED;
    if (m == 0 || n == 0) return MI355_OK;
    DArgs a;
    a.W = (const uint8_t *)Wa; a.w_row_stride = wa_row_stride; a.W2 = (const uint8_t *)Wb; a.w2_row_stride = wb_row_stride;
    a.x = x; a.x_row_stride = x_row_stride; a.dst = dst; a.dst_col_stride = dst_col_stride; a.hbf = dst_bf16; a.glu_op = glu_op;
    a.m = (int)m; a.n = (int)n; a.k = (int)k;
    a.sorted = a.off = a.tile_off = nullptr; a.n_expert = a.n_used = a.act_ne1 = 0; a.w_expert_stride = 0; a.part = nullptr; a.dbg = 0;
    a.y2 = 0; a.m2 = 0; a.part2 = nullptr;
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
        case MI355_TYPE_Q4_0: return launch_glu<MI355_TYPE_Q4_0>(a, s, "mmdq_glu_q4_0");
        case MI355_TYPE_Q8_0: return launch_glu<MI355_TYPE_Q8_0>(a, s, "mmdq_glu_q8_0");
        case MI355_TYPE_Q4_K: return launch_glu<MI355_TYPE_Q4_K>(a, s, "mmdq_glu_q4_K");
        case MI355_TYPE_Q5_K: return launch_glu<MI355_TYPE_Q5_K>(a, s, "mmdq_glu_q5_K");
        case MI355_TYPE_Q6_K: return launch_glu<MI355_TYPE_Q6_K>(a, s, "mmdq_glu_q6_K");
        case MI355_TYPE_Q4_1: return launch_glu<MI355_TYPE_Q4_1>(a, s, "mmdq_glu_q4_1");
        case MI355_TYPE_Q5_0: return launch_glu<MI355_TYPE_Q5_0>(a, s, "mmdq_glu_q5_0");
        case MI355_TYPE_Q5_1: return launch_glu<MI355_TYPE_Q5_1>(a, s, "mmdq_glu_q5_1");
        case MI355_TYPE_F16:  return launch_glu<MI355_TYPE_F16>(a, s, "mmdq_glu_f16");
        case MI355_TYPE_BF16: return launch_glu<MI355_TYPE_BF16>(a, s, "mmdq_glu_bf16");
        case MI355_TYPE_F32:  return launch_glu<MI355_TYPE_F32>(a, s, "mmdq_glu_f32");
    }
    return MI355_E_TYPE;
}

template <int TA, int TB>
static int launch_pair(const DArgs & a, int kz, hipStream_t s) {
    const unsigned ty = (unsigned)((a.m + 63) / 64);
    MI355_LAUNCH("mmdq_pair", (mmdq_ws_kernel<TA, 64, 128, false, TB>), dim3((unsigned)((a.n + BN - 1) / BN), 2 * ty, (unsigned)kz), dim3(384), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// Two products of the same activations in ONE launch: mi355_mul_mat_dq_parts for Wa and for Wb (both m rows), where each alone would be
// a split-K launch of 64 x 128 tiles that leaves half the chip idle (wk and wv of a GQA model at prefill: M = 1024).  Every workgroup
// computes what it would have computed in the separate launch, so the planes hold the same bits.  Same type, or a q4_K / q5_K matrix
// with a q6_K one (llama.cpp's Q4_K_M / Q5_K_M recipes give attn_v more bits); anything else: MI355_E_UNSUPPORTED, two launches.
extern "C" int mi355_mul_mat_dq_parts_pair(int type_a, const void * Wa, int64_t wa_row_stride, float * parts_a,
                                           int type_b, const void * Wb, int64_t wb_row_stride, float * parts_b, int64_t parts_bytes_each,
                                           const uint16_t * x, int64_t x_row_stride, int64_t m, int64_t k, int64_t n, int32_t * kz, void * stream) {
    if (!Wa || !Wb || !parts_a || !parts_b || !x || !kz) return MI355_E_NULL;
    const int64_t bsa = mi355_blck_size(type_a), bsb = mi355_blck_size(type_b);
    if (bsa == 0 || bsb == 0 || type_a == MI355_TYPE_Q8_1 || type_b == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (m <= 0 || n <= 0 || k <= 0 || k % bsa != 0 || k % bsb != 0 || m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff) return MI355_E_SHAPE;
    if (k % BK != 0 || 2 * ((m + 63) / 64) > 65535) return MI355_E_UNSUPPORTED;
    const int nz = kz_slices(m, n, k) == 0 ? kz_slices_small(m, n, k) : 0;
    if (nz < 2) return MI355_E_UNSUPPORTED;               // not the launch form this entry pairs up
    if (((uintptr_t)x & 15) || (x_row_stride & 7) || x_row_stride < k || (((uintptr_t)parts_a | (uintptr_t)parts_b) & 15)) return MI355_E_ALIGN;
    if (parts_bytes_each < (int64_t)nz * n * m * (int64_t)sizeof(float)) return MI355_E_WORKSPACE;
    const bool a16 = type_a == MI355_TYPE_Q4_K || type_a == MI355_TYPE_Q5_K, b16 = type_b == MI355_TYPE_Q4_K || type_b == MI355_TYPE_Q5_K;
    if ((((uintptr_t)Wa | (uintptr_t)wa_row_stride) & (a16 ? 15 : 1)) || (((uintptr_t)Wb | (uintptr_t)wb_row_stride) & (b16 ? 15 : 1))) return MI355_E_ALIGN;
    if (wa_row_stride < mi355_row_size(type_a, k) || wb_row_stride < mi355_row_size(type_b, k)) return MI355_E_SHAPE;
    DArgs a;
    a.W = (const uint8_t *)Wa; a.w_row_stride = wa_row_stride; a.W2 = (const uint8_t *)Wb; a.w2_row_stride = wb_row_stride;
    a.x = x; a.x_row_stride = x_row_stride; a.dst = parts_a; a.dst_col_stride = m; a.hbf = nullptr; a.glu_op = 0;
    a.m = (int)m; a.n = (int)n; a.k = (int)k; a.part = parts_a; a.part2 = parts_b; a.m2 = (int)m; a.y2 = (int)((m + 63) / 64);
    a.sorted = a.off = a.tile_off = nullptr; a.n_expert = a.n_used = a.act_ne1 = 0; a.w_expert_stride = 0; a.dbg = 0;
    hipStream_t s = (hipStream_t)stream;
    *kz = nz;
#define PAIR_CASE(TA, TB) if (type_a == TA && type_b == TB) return launch_pair<TA, TB>(a, nz, s)
    PAIR_CASE(MI355_TYPE_Q4_K, MI355_TYPE_Q4_K); PAIR_CASE(MI355_TYPE_Q4_K, MI355_TYPE_Q6_K);
    PAIR_CASE(MI355_TYPE_Q5_K, MI355_TYPE_Q5_K); PAIR_CASE(MI355_TYPE_Q5_K, MI355_TYPE_Q6_K);
    PAIR_CASE(MI355_TYPE_Q6_K, MI355_TYPE_Q6_K); PAIR_CASE(MI355_TYPE_Q8_0, MI355_TYPE_Q8_0);
    PAIR_CASE(MI355_TYPE_Q4_0, MI355_TYPE_Q4_0); PAIR_CASE(MI355_TYPE_Q4_1, MI355_TYPE_Q4_1);
    PAIR_CASE(MI355_TYPE_Q5_0, MI355_TYPE_Q5_0); PAIR_CASE(MI355_TYPE_Q5_1, MI355_TYPE_Q5_1);
    PAIR_CASE(MI355_TYPE_F16, MI355_TYPE_F16);   PAIR_CASE(MI355_TYPE_BF16, MI355_TYPE_BF16);
#undef PAIR_CASE
    return MI355_E_UNSUPPORTED;
}

// workspace of mi355_mul_mat_id_dq: bf16 activations [n_tokens * b_ne1][k], then the routing tables
static int64_t id_ws_act_bytes(int64_t k, int64_t n_tokens, int64_t b_ne1) { return (n_tokens * b_ne1 * k * 2 + 255) / 256 * 256; }

extern "C" int64_t mi355_mul_mat_id_dq_workspace_bytes(int64_t k, int64_t n_tokens, int64_t b_ne1, int64_t n_used, int64_t n_expert) {
    if (k <= 0 || n_tokens < 0 || b_ne1 < 1 || n_used < 1 || n_expert < 1) return -1;
    return id_ws_act_bytes(k, n_tokens, b_ne1) + (n_tokens * n_used + 2 * (n_expert + 1)) * 4 + 256;
}

extern "C" int mi355_mul_mat_id_dq(int type, const void * as, int64_t w_row_stride, int64_t w_expert_stride, const float * b,
                                   int64_t b_nb1, int64_t b_nb2, const int32_t * ids, int64_t ids_stride, float * dst, int64_t m,
                                   int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1, void * workspace,
                                   int64_t workspace_bytes, void * stream) {
    if (!as || !b || !ids || !dst || !workspace) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0 || type == MI355_TYPE_F32 || type == MI355_TYPE_F16 || type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (m < 0 || k <= 0 || k % bs != 0 || n_expert < 1 || n_expert > 256 || n_used < 1 || n_tokens < 0 || (b_ne1 != 1 && b_ne1 != n_used) ||
        m > 0x7fffffff || k > 0x7fffffff || n_tokens * n_used > 0x3fffffff) return MI355_E_SHAPE;
    if (k % BK != 0) return MI355_E_UNSUPPORTED;
    if (workspace_bytes < mi355_mul_mat_id_dq_workspace_bytes(k, n_tokens, b_ne1, n_used, n_expert)) return MI355_E_WORKSPACE;
    if (((uintptr_t)workspace & 15) || ((uintptr_t)dst & 3) || ((uintptr_t)ids & 3)) return MI355_E_ALIGN;
    const bool a16 = type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K;
    if (((uintptr_t)as | (uintptr_t)w_row_stride | (uintptr_t)w_expert_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (w_row_stride < mi355_row_size(type, k)) return MI355_E_SHAPE;
    if (m == 0 || n_tokens == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    uint16_t * act = (uint16_t *)workspace;
    int * sorted = (int *)((char *)workspace + id_ws_act_bytes(k, n_tokens, b_ne1));
    int * off = sorted + n_tokens * n_used, * tile_off = off + n_expert + 1;
    int rc = mi355_act_bf16(b, act, k, b_ne1, n_tokens, 1, b_nb1, b_nb2, 0, stream);
    if (rc != MI355_OK) return rc;
    const int bn = ids_wide(type, m) ? 256 : BN;            // pairs per column tile of the kernel that will run
    MI355_LAUNCH("moe_bucket", moe_bucket_kernel, dim3(1), dim3(1024), 0, s, ids, ids_stride, (int)n_tokens, (int)n_used, (int)n_expert, bn,
                 sorted, off, tile_off);
    MI355_CHECK_LAUNCH();
    DArgs a;
    a.W = (const uint8_t *)as; a.w_row_stride = w_row_stride; a.x = act; a.x_row_stride = k;
    a.dst = dst; a.dst_col_stride = m; a.m = (int)m; a.n = (int)(n_tokens * n_used); a.k = (int)k;
    a.sorted = sorted; a.off = off; a.tile_off = tile_off; a.n_expert = (int)n_expert; a.n_used = (int)n_used; a.act_ne1 = (int)b_ne1;
    a.w_expert_stride = w_expert_stride; a.dbg = 0; a.part = nullptr;
    a.W2 = nullptr; a.w2_row_stride = 0; a.hbf = nullptr; a.glu_op = 0; a.y2 = 0; a.m2 = 0; a.part2 = nullptr;
    const int max_tiles = (int)((n_tokens * n_used + bn - 1) / bn + n_expert);   // sum over experts of ceil(cnt_e / bn) never exceeds this
    switch (type) {
        case MI355_TYPE_Q4_0: return launch_ids<MI355_TYPE_Q4_0>(a, max_tiles, s, "mmdq_id_q4_0");
        case MI355_TYPE_Q8_0: return launch_ids<MI355_TYPE_Q8_0>(a, max_tiles, s, "mmdq_id_q8_0");
        case MI355_TYPE_Q4_K: return launch_ids<MI355_TYPE_Q4_K>(a, max_tiles, s, "mmdq_id_q4_K");
        case MI355_TYPE_Q5_K: return launch_ids<MI355_TYPE_Q5_K>(a, max_tiles, s, "mmdq_id_q5_K");
        case MI355_TYPE_Q6_K: return launch_ids<MI355_TYPE_Q6_K>(a, max_tiles, s, "mmdq_id_q6_K");
        case MI355_TYPE_Q4_1: return launch_ids<MI355_TYPE_Q4_1>(a, max_tiles, s, "mmdq_id_q4_1");
        case MI355_TYPE_Q5_0: return launch_ids<MI355_TYPE_Q5_0>(a, max_tiles, s, "mmdq_id_q5_0");
        case MI355_TYPE_Q5_1: return launch_ids<MI355_TYPE_Q5_1>(a, max_tiles, s, "mmdq_id_q5_1");
        case MI355_TYPE_Q2_K: return launch_ids<MI355_TYPE_Q2_K>(a, max_tiles, s, "mmdq_id_q2_k");
        case MI355_TYPE_Q3_K: return launch_ids<MI355_TYPE_Q3_K>(a, max_tiles, s, "mmdq_id_q3_k");
        case MI355_TYPE_IQ4_NL: return launch_ids<MI355_TYPE_IQ4_NL>(a, max_tiles, s, "mmdq_id_iq4_nl");
        case MI355_TYPE_IQ4_XS: return launch_ids<MI355_TYPE_IQ4_XS>(a, max_tiles, s, "mmdq_id_iq4_xs");
    }
    return MI355_E_TYPE;
}
