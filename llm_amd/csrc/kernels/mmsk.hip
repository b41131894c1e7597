// mmsk.hip -- "skinny" MUL_MAT for 2 <= N <= 32 activation columns per pass (batched decode, short prompts,
// speculative decoding, per-expert columns of MUL_MAT_ID) on the int8 matrix cores.
//
// Same contract and arithmetic as the mat-vec (mmvq.hip: q8_1 activations, i32 dot per 32-block, f32 scales);
// replaces mul_mat_vec_q with ncols_dst 2..8 (ggml/src/ggml-cuda/mmvq.cu:137-224) and the small-batch end of
// mul_mat_q (mmq.cuh:3095-3175, mmq_x 8..32).  The mat-vec kernel spends one v_dot4 per 4 weights PER COLUMN, so
// it falls from the HBM roofline at 1 column to 9 % of it at 8; here one v_mfma_i32_32x32x32_i8 covers 32 weight
// rows x 32 columns x one 32-block, and the cost per weight byte does not depend on N (<= 32).
//
// Roofline: HBM (weights streamed once per pass of 32 columns); the per-block f32 rescale (about 64 VALU
// instructions per MFMA) is sized to run at about the CU-side load limit (~9 B/cycle/CU).
//
// Shape of the kernel
//   * workgroup = 512 threads = 8 waves, all on the same 32 weight rows; wave w takes super-steps (256 weights of
//     K) w, w+8, ...; the 8 partial 32x32 tiles are added through LDS at the end.  When M/32 workgroups cannot
//     fill the chip, K is also split over 2 workgroups whose tiles are added into a zeroed dst with float atomics
//     (two addends: the sum does not depend on the order).
//   * weights go global -> registers directly in MFMA operand layout: lane (row = lane & 31, half = lane >> 5)
//     loads the 16 weights [16 half, 16 half + 16) of a 32-block of ITS row (one dwordx4 for two q4_K blocks),
//     unpacks them to int8 with a few bit ops; next super-step's loads are issued before the current MFMAs.
//   * activations (q8_1 SoA, mi355_quantize_act) are read per block straight from global / L2: lane
//     (col = lane & 31, half) loads 16 int8 + {d, sum} of its column; columns >= N repeat column N-1.
//   * C[r] of lane (col, half) is row (r&3) + 8 (r>>2) + 4 half: the activation scale is a per-lane scalar, the
//     weight scales of the 16 rows come from a small per-wave LDS table written by the lanes that own the rows.
#include "common.h"
#include "moe_bucket.h"

namespace {

typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int SK_WAVES = 8, SK_THREADS = SK_WAVES * 64;

struct SArgs {
    const uint8_t * W; int64_t w_row_stride;
    const int8_t * aq; const float * ads;     // [n][k], [n][k/32][2]
    float * dst; int64_t dst_col_stride;
    int m, n, k;
    IdsArgs ids;      // MUL_MAT_ID mode when ids.sorted != nullptr: blockIdx.y = pass of 32 gathered columns of one expert
    int m_cut;        // rows of the whole matrix when the launch covers a row slice (0: m): decides whether K is cut over two workgroups
};

// per-wave LDS scale table: [8 blocks of the super-step][2 sub-groups (q6_K)][32 rows]
struct WaveTab { float ws[8][2][32]; float wm[8][32]; };

template <int TYPE> struct SK;

// ---- Q4_K / Q5_K: super-step = one 256-weight super-block
template <int TYPE> struct SKK45 {
    static constexpr int SUB = 1;
    static constexpr bool MINS = true;
    static constexpr int BS = TYPE == MI355_TYPE_Q4_K ? 144 : 176;
    static constexpr int QO = TYPE == MI355_TYPE_Q4_K ? 16 : 48;
    struct Raw { v4i q[4]; v4i qh; v4i hdr; };
    static __device__ __forceinline__ Raw load(const uint8_t * row, int sb, int lh) {
        const uint8_t * blk = row + (int64_t)sb * BS;
        Raw r;
#pragma unroll
        for (int g = 0; g < 4; g++) r.q[g] = *(const v4i *)(blk + QO + g * 32 + lh * 16);
        if (TYPE == MI355_TYPE_Q5_K) r.qh = *(const v4i *)(blk + 16 + lh * 16);
        r.hdr = *(const v4i *)blk;
        return r;
    }
    // int8 operand of block j for this lane's 16 weights; sub unused
    static __device__ __forceinline__ v4i frag(const Raw & r, int j, int, int) {
        v4i x = (r.q[j >> 1] >> (4 * (j & 1))) & 0x0f0f0f0f;
        if (TYPE == MI355_TYPE_Q5_K) x |= ((r.qh >> j) & 0x01010101) << 4;
        return x;
    }
    static __device__ __forceinline__ void scales(const Raw & r, WaveTab & t, int lr) {
        const uint32_t u0 = r.hdr.y, u1 = r.hdr.z, u2 = r.hdr.w;
        const float d = h2f((uint16_t)(r.hdr.x & 0xffff)), dmin = h2f((uint16_t)((uint32_t)r.hdr.x >> 16));
#pragma unroll
        for (int j = 0; j < 8; j++) {   // get_scale_min_k4, ggml-quants.c:625-632
            uint32_t sc, m;
            if (j < 4) { sc = (u0 >> (8 * j)) & 0x3f; m = (u1 >> (8 * j)) & 0x3f; }
            else {
                sc = ((u2 >> (8 * (j - 4))) & 0x0f) | (((u0 >> (8 * (j - 4) + 6)) & 0x03) << 4);
                m  = ((u2 >> (8 * (j - 4) + 4)) & 0x0f) | (((u1 >> (8 * (j - 4) + 6)) & 0x03) << 4);
            }
            t.ws[j][0][lr] = d * (float)sc;
            t.wm[j][lr]    = dmin * (float)m;
        }
    }
};
template <> struct SK<MI355_TYPE_Q4_K> : SKK45<MI355_TYPE_Q4_K> {};
template <> struct SK<MI355_TYPE_Q5_K> : SKK45<MI355_TYPE_Q5_K> {};

// ---- Q6_K: 16-weight scale groups -> the two halves of a 32-block (lane halves) carry different scales:
// two MFMAs per block, each with the other half's operand zeroed (SUB = 2)
template <> struct SK<MI355_TYPE_Q6_K> {
    static constexpr int SUB = 2;
    static constexpr bool MINS = false;
    struct Raw { v4i ql[4]; v4i qh[2]; v4i sc; uint16_t d; };
    static __device__ __forceinline__ Raw load(const uint8_t * row, int sb, int lh) {
        const uint8_t * blk = row + (int64_t)sb * 210;
        Raw r;
#pragma unroll
        for (int c = 0; c < 4; c++) r.ql[c] = ld16_a2(blk + 32 * c + lh * 16);     // ql[64 n + 32 (q&1) + l], c = 2n + (q&1)
#pragma unroll
        for (int n = 0; n < 2; n++) r.qh[n] = ld16_a2(blk + 128 + 32 * n + lh * 16);
        r.sc = ld16_a2(blk + 192);
        r.d  = ld_u16(blk + 208);
        return r;
    }
    static __device__ __forceinline__ v4i frag(const Raw & r, int j, int lh, int sub) {
        // block j = 4 n + q: ql chunk 2n + (q&1), nibble q>>1; qh[n] >> 2q (dequantize_row_q6_K, ggml-quants.c:1695-1711)
        const int n = j >> 2, q = j & 3;
        v4i x = ((r.ql[2 * n + (q & 1)] >> (4 * (q >> 1))) & 0x0f0f0f0f) | (((r.qh[n] >> (2 * q)) & 0x03030303) << 4);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = sub_bytes(x[i], 0x20202020);
        const v4i z = {0, 0, 0, 0};
        return lh == sub ? x : z;
    }
    static __device__ __forceinline__ void scales(const Raw & r, WaveTab & t, int lr) {
        const float d = h2f(r.d);
        const int8_t * sc = (const int8_t *)&r.sc;      // sc[8 n + 2 q + half]
#pragma unroll
        for (int j = 0; j < 8; j++) {
            t.ws[j][0][lr] = d * (float)sc[2 * j];
            t.ws[j][1][lr] = d * (float)sc[2 * j + 1];
        }
    }
};

// ---- Q8_0 / Q4_0: super-step = 8 consecutive 32-blocks
template <> struct SK<MI355_TYPE_Q8_0> {
    static constexpr int SUB = 1;
    static constexpr bool MINS = false;
    struct Raw { v4i q[8]; uint16_t d[8]; };
    static __device__ __forceinline__ Raw load(const uint8_t * row, int sb, int lh) {
        Raw r;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint8_t * blk = row + (int64_t)(sb * 8 + j) * 34;
            r.q[j] = ld16_a2(blk + 2 + lh * 16);
            r.d[j] = ld_u16(blk);
        }
        return r;
    }
    static __device__ __forceinline__ v4i frag(const Raw & r, int j, int, int) { return r.q[j]; }
    static __device__ __forceinline__ void scales(const Raw & r, WaveTab & t, int lr) {
#pragma unroll
        for (int j = 0; j < 8; j++) t.ws[j][0][lr] = h2f(r.d[j]);
    }
};
template <> struct SK<MI355_TYPE_Q4_0> {
    static constexpr int SUB = 1;
    static constexpr bool MINS = false;
    struct Raw { v4i q[8]; uint16_t d[8]; };
    static __device__ __forceinline__ Raw load(const uint8_t * row, int sb, int) {
        Raw r;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint8_t * blk = row + (int64_t)(sb * 8 + j) * 18;
            r.q[j] = ld16_a2(blk + 2);      // both halves need the same 16 bytes: low nibbles = weights 0..15, high = 16..31
            r.d[j] = ld_u16(blk);
        }
        return r;
    }
    static __device__ __forceinline__ v4i frag(const Raw & r, int j, int lh, int) {
        v4i x = (r.q[j] >> (4 * lh)) & 0x0f0f0f0f;
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = sub_bytes(x[i], 0x08080808);
        return x;
    }
    static __device__ __forceinline__ void scales(const Raw & r, WaveTab & t, int lr) {
#pragma unroll
        for (int j = 0; j < 8; j++) t.ws[j][0][lr] = h2f(r.d[j]);
    }
};

// The 32 columns of a workgroup.  Plain MUL_MAT: columns 0 .. n-1 of the pass (columns >= n repeat column n-1 and are
// never stored), K possibly split over gridDim.y workgroups.  MUL_MAT_ID: blockIdx.y is a pass of up to 32 (token,
// slot) pairs routed to one expert (moe_bucket.h): activation row = token * act_ne1 + slot % act_ne1, dst column = the
// pair id, weights = that expert's matrix.  Returns false (uniformly) for a pass beyond the last one.
struct ColMap { int ncol, ky_idx, ky_cnt; int64_t w_off; };
static __device__ __forceinline__ bool col_map(const IdsArgs & I, int n, int tid, int * s_row, int * s_pair, ColMap & cm) {
    if (I.sorted) {
        // the expert of column pass ti = the number of experts whose passes end at or before it: the table entries are requested
        // together (64 per step) and counted with a ballot, instead of walked one dependent load at a time
        const int ti = blockIdx.y, ln = tid & 63;
        const int total = I.tile_off[I.n_expert];
        int e = 0;
        for (int b0 = 0; b0 < I.n_expert; b0 += 64) {
            const int end = b0 + ln < I.n_expert ? I.tile_off[b0 + ln + 1] : 0x7fffffff;
            e += __builtin_popcountll(__ballot(end <= ti));
        }
        if (ti >= total) return false;
        const int base = I.off[e] + (ti - I.tile_off[e]) * 32, left = I.off[e + 1] - base;
        cm.ncol = left < 32 ? left : 32;
        if (tid < 32) {
            const int pair = I.sorted[base + (tid < cm.ncol ? tid : cm.ncol - 1)];
            s_pair[tid] = pair;
            s_row[tid]  = (pair / I.n_used) * I.act_ne1 + (pair % I.n_used) % I.act_ne1;
        }
        cm.w_off = (int64_t)e * I.w_expert_stride; cm.ky_idx = 0; cm.ky_cnt = 1;
    } else {
        cm.ncol = n;
        if (tid < 32) { const int c = tid < n ? tid : n - 1; s_row[tid] = c; s_pair[tid] = c; }
        cm.w_off = 0; cm.ky_idx = (int)blockIdx.y; cm.ky_cnt = (int)gridDim.y;
    }
    __syncthreads();
    return true;
}

template <int TYPE>
__global__ void __launch_bounds__(SK_THREADS)
mmsk_kernel(const SArgs a) {
    typedef SK<TYPE> T;
    __shared__ WaveTab tab[SK_WAVES];
    __shared__ float red[SK_WAVES][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 32;
    __shared__ int s_row[32], s_pair[32];
    ColMap cm;
    if (!col_map(a.ids, a.n, tid, s_row, s_pair, cm)) return;
    const int row = m0 + lr < a.m ? m0 + lr : a.m - 1;
    const uint8_t * wrow = a.W + cm.w_off + (int64_t)row * a.w_row_stride;
    const int col = s_row[lr];
    const int8_t * aq = a.aq + (int64_t)col * a.k + lh * 16;
    const float * ads = a.ads + (int64_t)col * (a.k / QK) * 2;

    // super-steps of this workgroup (K split over ky_cnt workgroups) and of this wave
    const int nsb = a.k / 256, per = (nsb + cm.ky_cnt - 1) / cm.ky_cnt;
    const int sb0 = cm.ky_idx * per, sb1 = sb0 + per < nsb ? sb0 + per : nsb;

    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    WaveTab & t = tab[wave];
    int sb = sb0 + wave;
    typename T::Raw cur;
    if (sb < sb1) cur = T::load(wrow, sb, lh);
    for (; sb < sb1; sb += SK_WAVES) {
        typename T::Raw nxt;
        if (sb + SK_WAVES < sb1) nxt = T::load(wrow, sb + SK_WAVES, lh);
        // activations of the 8 blocks: 16 int8 of this lane's column and half, {d, sum}
        v4i bq[8];
        v2f bd[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bq[j] = *(const v4i *)(aq + (int64_t)(sb * 8 + j) * 32);
            bd[j] = *(const v2f *)(ads + (int64_t)(sb * 8 + j) * 2);
        }
        if (lh == 0) T::scales(cur, t, lr);      // lanes 0..31 own rows 0..31 (the upper half holds the same rows)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float d8 = bd[j].x, s8 = bd[j].x * bd[j].y;
#pragma unroll
            for (int sub = 0; sub < T::SUB; sub++) {
                const v16i zero = {0};
                const v16i c = __builtin_amdgcn_mfma_i32_32x32x32_i8(T::frag(cur, j, lh, sub), bq[j], zero, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const v4f ws = *(const v4f *)&t.ws[j][sub][8 * g + 4 * lh];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[4 * g + e] = __builtin_fmaf(ws[e], d8 * (float)c[4 * g + e], acc[4 * g + e]);
                }
            }
            if (T::MINS) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const v4f wm = *(const v4f *)&t.wm[j][8 * g + 4 * lh];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[4 * g + e] = __builtin_fmaf(-wm[e], s8, acc[4 * g + e]);
                }
            }
        }
        if (sb + SK_WAVES < sb1) cur = nxt;
    }

    // add the 8 waves' tiles
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][r][lane] = acc[r];
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += SK_THREADS) {
        const int r = e >> 6, l = e & 63;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; w++) s += red[w][r][l];
        const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ocol = l & 31;
        if (orow < a.m && ocol < cm.ncol) {
            float * o = a.dst + (int64_t)s_pair[ocol] * a.dst_col_stride + orow;
            if (cm.ky_cnt > 1) atomicAdd(o, s); else *o = s;
        }
    }
}

template <int TYPE>
static int launch(const SArgs & a, hipStream_t s, const char * tag) {
    const unsigned nb = (unsigned)((a.m + 31) / 32);
    const unsigned ky = ((a.m_cut ? a.m_cut : a.m) + 31) / 32 < 256 && a.k >= 2048 ? 2 : 1;      // few row tiles (of the WHOLE matrix): split K over two workgroups
    if (ky > 1) {
        hipError_t e = a.dst_col_stride == a.m
            ? hipMemsetAsync(a.dst, 0, (size_t)a.n * a.m * sizeof(float), s)
            : hipMemset2DAsync(a.dst, (size_t)a.dst_col_stride * sizeof(float), 0, (size_t)a.m * sizeof(float), (size_t)a.n, s);
        if (e != hipSuccess) return (int)e;
    }
    MI355_LAUNCH(tag, (mmsk_kernel<TYPE>), dim3(nb, ky), dim3(SK_THREADS), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}


// =====================================================================================================================
// K-quants on q8_K activations: no per-block rescale at all.
//
// The f32 rescale above costs ~64 VALU instructions per MFMA, and that, not HBM, bounds the kernel (measured 1.8 TB/s
// on q4_K).  The K-quants carry INTEGER sub-block scales under one f16 super-block scale, and ggml's own CPU contract for
// them quantizes the activations per 256 values (q8_K: quantize_row_q8_K_ref, ggml-quants.c:2471-2508;
// ggml_vec_dot_q4_K_q8_K, ggml-cpu/quants.c).  With one activation scale per super-block the whole super-block sum
//     sum_j sc_j * (q_j . a_j)
// is an integer and can stay on the matrix core: the 6-bit sc_j is split into P digits of B bits, digit * q fits a
// signed byte (7 * 15 = 105 for q4_K with 2 x 3 bits; 3 * 31 = 93 for q5_K with 3 x 2 bits) and is one
// v_pk_mul_lo_u16 per 4 weights (no carry crosses a byte), and P MFMAs per 32-block accumulate in i32 over all 8
// blocks.  One conversion per super-block: acc += d_row * d8_col * float(C_0 + (C_1 << B) + ...).
// The min term sum_j (dmin m_j)_row * (d8 bsum_j)_col is a K = 8 f32 product: 4 x v_mfma_f32_32x32x2_f32 per super-block
// into one accumulator that lives through the whole K loop.
// Several weight matrices of one type and one K against the SAME activations in one launch (wq / wk / wv of a small batch: three
// launches of 128 + 32 + 32 tiles, each cut in two along K with a memset in front, become one of 192 tiles with neither).
struct SkGroup {
    int n;
    int tile_end[4];         // running count of 32-row tiles; entries past n hold the total
    const uint8_t * W[4]; int64_t w_row_stride[4];
    float * dst[4]; int64_t dst_col_stride[4];
    int m[4];
};
struct SkMat { const uint8_t * W; int64_t w_row_stride; float * dst; int64_t dst_col_stride; int m, tile, col0; };

struct S2Args {
    const uint8_t * W; int64_t w_row_stride;
    const int8_t * aq; const float * ad; const int16_t * bs;     // [n][k], [n][k/256], [n][k/16]
    float * dst; int64_t dst_col_stride;
    int m, n, k;
    IdsArgs ids;
    SkGroup grp;             // grp.n > 0: blockIdx.x runs over the 32-row tiles of several matrices (W, dst, m above are unused)
    int m_cut;               // rows of the WHOLE matrix when this launch covers a row slice of it (0: a.m): what sk_tpw is asked with
    int ct;                  // column tiles of 32 per row tile (plain MUL_MAT, 33..64 columns: 2; else 1): tile index t -> (t / ct, column half t % ct)
#ifdef SK_STAMPS
    long long * stamps;      // diagnostic build only (lab/mmsk_stamps.hip): [workgroup][wave][8] s_memtime / s_memrealtime
#endif
};

#ifdef SK_STAMPS
#define SK_STAMP(i) do { if (lane == 0) st_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

// the matrix a 32-row tile belongs to (t is wave-uniform: scalar selects on kernel arguments)
static __device__ __forceinline__ SkMat sk_mat(const S2Args & a, int t) {
    int col0 = 0;
    if (a.ct > 1) { col0 = (t % a.ct) * 32; t /= a.ct; }      // consecutive tiles of a workgroup: the column halves of one row tile
    SkMat r = { a.W, a.w_row_stride, a.dst, a.dst_col_stride, a.m, t, col0 };
    if (a.grp.n > 0) {
        const int gi = (t >= a.grp.tile_end[0]) + (t >= a.grp.tile_end[1]) + (t >= a.grp.tile_end[2]);
        r.W   = gi == 0 ? a.grp.W[0]   : gi == 1 ? a.grp.W[1]   : gi == 2 ? a.grp.W[2]   : a.grp.W[3];
        r.dst = gi == 0 ? a.grp.dst[0] : gi == 1 ? a.grp.dst[1] : gi == 2 ? a.grp.dst[2] : a.grp.dst[3];
        r.w_row_stride   = gi == 0 ? a.grp.w_row_stride[0]   : gi == 1 ? a.grp.w_row_stride[1]   : gi == 2 ? a.grp.w_row_stride[2]   : a.grp.w_row_stride[3];
        r.dst_col_stride = gi == 0 ? a.grp.dst_col_stride[0] : gi == 1 ? a.grp.dst_col_stride[1] : gi == 2 ? a.grp.dst_col_stride[2] : a.grp.dst_col_stride[3];
        r.m    = gi == 0 ? a.grp.m[0] : gi == 1 ? a.grp.m[1] : gi == 2 ? a.grp.m[2] : a.grp.m[3];
        r.tile = t - (gi == 0 ? 0 : gi == 1 ? a.grp.tile_end[0] : gi == 2 ? a.grp.tile_end[1] : a.grp.tile_end[2]);
        if (t >= a.grp.tile_end[3]) r.m = 0;      // past the last matrix: an empty tile
    }
    return r;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
// per-byte x * digit for bytes x, digit with x * digit < 256: v_pk_mul_lo_u16 on the two 16-bit halves
static __device__ __forceinline__ int pk_bytes_mul(int x, int rep) {
    return __builtin_bit_cast(int, (us2)(__builtin_bit_cast(us2, x) * __builtin_bit_cast(us2, rep)));
}

// Operand path: a lane of the MFMA owns one weight ROW (and one activation COLUMN), so loading operands straight
// from global memory makes every load instruction touch 32 different cache lines for 16-32 bytes each; the lines
// have to survive in the 32 KB L1 until the other 4 loads of the super-step come for the rest of them, and with 8
// waves x 2 super-steps in flight they do not (measured: 2.4 TB/s, no change from prefetching or fewer waves).
// So each wave stages its super-step through a private LDS region with COALESCED loads -- consecutive lanes read
// consecutive 16-byte chunks of a row's 144/176-byte super-block (7 / 5 rows per instruction), and 4 columns x
// 256 bytes of activations per instruction -- and reads the operands back row-per-lane with ds_read_b128 (pitches
// 144 / 176 / 272 bytes: conflict-free).  The region is private to the wave: LDS operations of one wave execute in
// order, so no workgroup barrier is involved.  Each wave takes a CONTIGUOUS range of super-blocks.
// TPW: row tiles per workgroup.  With K = 4096 a wave of an 8-wave tile has TWO super-steps between a ~1900-cycle setup and a
// ~2700-cycle reduction (stamps, lab/mmsk_stamps.hip), and 448 tiles of 14336 rows are 1.75 rounds on 256 CUs.  TPW = 2 gives each tile
// WAVES / 2 waves with twice the steps: the fixed costs are paid once per 64 rows and the launch is one round.
template <int TYPE, int WAVES, int TPW = 1>
__global__ void __launch_bounds__(WAVES * 64)
mmsk2_kernel(const S2Args a) {
    typedef SK<TYPE> T;
    constexpr int WK = WAVES / TPW;            // waves that split a tile's K
    constexpr int P = TYPE == MI355_TYPE_Q4_K ? 2 : 3, B = TYPE == MI355_TYPE_Q4_K ? 3 : 2;
    constexpr int BS = T::BS, QO = T::QO;
    constexpr int CH = BS / 16, RPI = 64 / CH, NI = (32 + RPI - 1) / RPI;      // 16-byte chunks per row; rows per load; loads
    constexpr int APITCH = 272, WBYTES = 32 * BS, ABYTES = 32 * APITCH, REGION = WBYTES + ABYTES + 16;   // + a dump slot
    static_assert(REGION >= 16 * 64 * 4, "the staging region doubles as the reduction buffer");
    __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES * REGION];
    __shared__ float drow[WAVES][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: everything derived from it stays in SGPRs
    const int lr = lane & 31, lh = lane >> 5;
    const int tile = wave / WK, kw = wave % WK;
    const SkMat mt = sk_mat(a, (int)blockIdx.x * TPW + tile);
    const int m0 = mt.tile * 32;
    const bool tile_ok = m0 < mt.m;             // (an odd number of 32-row tiles: the last workgroup's second tile is empty)
    uint8_t * Lw = lds + wave * REGION, * La = Lw + WBYTES;
    __shared__ int s_row[32], s_pair[32];
    ColMap cm;
    if (!col_map(a.ids, a.n, tid, s_row, s_pair, cm)) return;
    // this wave's columns: a pass of an expert's pairs (IDS), or columns col0 .. col0 + 31 of the batch
    const bool ids_mode = a.ids.sorted != nullptr;
    const int ncol = ids_mode ? cm.ncol : (a.n - mt.col0 < 32 ? (a.n - mt.col0 > 0 ? a.n - mt.col0 : 0) : 32);
    auto srow = [&](int j) { return ids_mode ? s_row[j] : (mt.col0 + j < a.n ? mt.col0 + j : a.n - 1); };
#ifdef SK_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane == 0) st_[7] = __builtin_amdgcn_s_memrealtime();
#endif
    SK_STAMP(0);

    // Staging roles.  Addresses are a wave-uniform base (advanced per super-block in SGPRs) plus a 32-bit lane offset
    // that never changes, so the loop spends no vector instructions on addressing.  Lanes without a chunk (lane 63
    // of 7 x 9, rows past 31) load chunk 0 and write it to the dump slot instead of being masked off.
    const uint8_t * wbase = mt.W + cm.w_off + (int64_t)m0 * mt.w_row_stride;
    const int wr_ = lane / CH, wc_ = lane % CH;
    uint32_t woff[NI], wlds[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int r = i * RPI + wr_;
        const bool v = lane < RPI * CH && r < 32;
        const int rr = !tile_ok ? 0 : (m0 + r < mt.m ? r : mt.m - 1 - m0);      // rows past the end repeat the last row (never stored)
        woff[i] = v ? (uint32_t)rr * (uint32_t)mt.w_row_stride + wc_ * 16 : 0u;
        wlds[i] = v ? r * BS + wc_ * 16 : WBYTES + ABYTES;
    }
    // Only the columns that exist are staged, four per load instruction: the activations of a super-step are 8 KB for 32 columns
    // against 4.6 KB of weights, and both go through the CU's load path -- at 8 columns they are 2 KB.  Lanes of the MFMA's other
    // columns read column 0 from LDS (finite values, results never stored).
    const int na = __builtin_amdgcn_readfirstlane((ncol + 3) >> 2);
    const int lrc = lr < 4 * na ? lr : 0;
    uint32_t aoff[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        aoff[i] = (uint32_t)srow(i * 4 + (lane >> 4)) * (uint32_t)a.k + (lane & 15) * 16;
    }
    const int col = srow(lr);
    const uint32_t doff = (uint32_t)col * (uint32_t)(a.k / 256) * 4;
    const uint32_t soff = (uint32_t)col * (uint32_t)(a.k / 8) + lh * 16;      // bsums of blocks 4 lh .. 4 lh + 3 (two int16 each)

    // super-blocks of this workgroup (K split over gridDim.y workgroups) and, contiguously, of this wave
    const int nsb = a.k / 256, per = (nsb + cm.ky_cnt - 1) / cm.ky_cnt;
    const int sb0 = cm.ky_idx * per, sb1 = sb0 + per < nsb ? sb0 + per : nsb;
    const int perw = (sb1 - sb0 + WK - 1) / WK;
    const int my0 = sb0 + kw * perw, my1 = !tile_ok ? my0 : (my0 + perw < sb1 ? my0 + perw : sb1);

    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    // Operands in flight: both are requested one super-step ahead.  (Requesting the weights two steps ahead -- a second
    // register set, activations issued first so that their wait does not retire the younger weights -- measured
    // 14.4 -> 14.0 us on 14336 x 4096: in-kernel stamps show a step costs ~3000 cycles of issue for two waves per SIMD,
    // which already covers the ~2 us memory latency.)
    struct StageW { v4i w[NI]; };
    struct StageA { v4i q[8]; v4i s; float d; };
    auto fetch_w = [&](int sb_) {
        StageW r;
        const uint8_t * wb = wbase + (int64_t)sb_ * BS;
#pragma unroll
        for (int i = 0; i < NI; i++) r.w[i] = *(const v4i *)(wb + woff[i]);
        return r;
    };
    auto fetch_a = [&](int sb_) {
        StageA r;
        const int8_t * ab = a.aq + (int64_t)sb_ * 256;
#pragma unroll
        for (int i = 0; i < 8; i++) if (i < na) r.q[i] = *(const v4i *)(ab + aoff[i]);
        r.s = *(const v4i *)((const uint8_t *)a.bs + (int64_t)sb_ * 32 + soff);
        r.d = *(const float *)((const uint8_t *)a.ad + (int64_t)sb_ * 4 + doff);
        return r;
    };
    const int last = my1 - 1;
    auto clampsb = [&](int sb_) { return sb_ < last ? sb_ : last; };      // past the end: re-read the last one (harmless)

    StageA sa;
    auto step = [&](int sb, StageW & sw) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; i++) *(v4i *)(Lw + wlds[i]) = sw.w[i];
#pragma unroll
        for (int i = 0; i < 8; i++) if (i < na) *(v4i *)(La + (i * 4 + (lane >> 4)) * APITCH + (lane & 15) * 16) = sa.q[i];
        const float d8 = sa.d;
        const v4i sv = sa.s;
        __builtin_amdgcn_wave_barrier();
        sa = fetch_a(clampsb(sb + 1));
        sw = fetch_w(clampsb(sb + 1));

        const uint8_t * wl = Lw + lr * BS;
        const v4i hdr = *(const v4i *)wl;
        typename T::Raw cur;
#pragma unroll
        for (int g = 0; g < 4; g++) cur.q[g] = *(const v4i *)(wl + QO + g * 32 + lh * 16);
        if (TYPE == MI355_TYPE_Q5_K) cur.qh = *(const v4i *)(wl + 16 + lh * 16);

        // this lane's row: 8 scales and 8 mins as packed bytes (get_scale_min_k4, ggml-quants.c:625-632)
        const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;
        const uint32_t scw[2] = { u0 & 0x3f3f3f3fu, (u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u) };
        const uint32_t mw[2]  = { u1 & 0x3f3f3f3fu, ((u2 >> 4) & 0x0f0f0f0fu) | ((u1 >> 2) & 0x30303030u) };
        const float d = h2f((uint16_t)(hdr.x & 0xffff)), ndmin = -h2f((uint16_t)((uint32_t)hdr.x >> 16));
        if (lh == 0) drow[wave][lr] = d;

        // min term, subtracted on the matrix core: the k index of the f32 MFMA is the lane half; step i covers
        // blocks i (lower half) and 4 + i
        const uint32_t mh = lh ? mw[1] : mw[0];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sum = (sv[i] << 16 >> 16) + (sv[i] >> 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ndmin * (float)((mh >> (8 * i)) & 0xffu), d8 * (float)sum, acc, 0, 0, 0);
        }

        // digit p of the 8 scales, still packed; v_perm_b32 copies digit j into both 16-bit halves of a word
        uint32_t digw[P][2];
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int h = 0; h < 2; h++) digw[p][h] = (scw[h] >> (B * p)) & (((1u << B) - 1) * 0x01010101u);

        v16i C[P];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const v4i x = T::frag(cur, j, lh, 0);
            const v4i bq = *(const v4i *)(La + lrc * APITCH + j * 32 + lh * 16);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const int rep = (int)__builtin_amdgcn_perm(0u, digw[p][j >> 2], 0x0c000c00u | (uint32_t)(j & 3) | ((uint32_t)(j & 3) << 16));
                v4i xp;
#pragma unroll
                for (int i = 0; i < 4; i++) xp[i] = pk_bytes_mul(x[i], rep);
                const v16i zero = {0};
                C[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xp, bq, j == 0 ? zero : C[p], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4f dr = *(const v4f *)&drow[wave][8 * g + 4 * lh];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int t = C[0][4 * g + e];
#pragma unroll
                for (int p = 1; p < P; p++) t += C[p][4 * g + e] << (B * p);
                acc[4 * g + e] = __builtin_fmaf(dr[e], d8 * (float)t, acc[4 * g + e]);
            }
        }
    };

    if (my0 < my1) {
        StageW w0 = fetch_w(my0);
        sa = fetch_a(my0);
        SK_STAMP(1);
        for (int sb = my0; sb < my1; sb++) {
            step(sb, w0);
            if (sb == my0) SK_STAMP(2);
        }
    }

    // add the waves' tiles (the staging regions are dead: reuse them)
    SK_STAMP(3);
    __syncthreads();
    SK_STAMP(4);
    float * red = (float *)lds;      // [WAVES][16][64]
    if (lr < ncol) {                 // a lane's 16 results are one column's: columns that do not exist are neither written nor added
#pragma unroll
        for (int r = 0; r < 16; r++) red[(wave * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < TPW * 16 * 64; e += WAVES * 64) {
        const int tl = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const SkMat mo = sk_mat(a, (int)blockIdx.x * TPW + tl);      // (tl is wave-uniform: 1024 elements per tile, 64 per wave)
        const int orow = mo.tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ocol = l & 31;
        const int ncol_o = ids_mode ? cm.ncol : (a.n - mo.col0 < 32 ? a.n - mo.col0 : 32);
        if (orow < mo.m && ocol < ncol_o) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < WK; w++) s += red[((tl * WK + w) * 16 + r) * 64 + l];
            float * o = mo.dst + (int64_t)(ids_mode ? s_pair[ocol] : mo.col0 + ocol) * mo.dst_col_stride + orow;
            if (cm.ky_cnt > 1) atomicAdd(o, s); else *o = s;
        }
    }
#ifdef SK_STAMPS
    SK_STAMP(5);
    if (lane == 0 && a.stamps) {
        st_[6] = __builtin_amdgcn_s_memrealtime();
        long long * o = a.stamps + ((int64_t)(blockIdx.x * gridDim.y + blockIdx.y) * WAVES + wave) * 8;
        for (int i = 0; i < 8; i++) o[i] = st_[i];
    }
#endif
}

// =====================================================================================================================
// q6_K on q8_K activations (ggml_vec_dot_q6_K_q8_K, ggml-cpu/quants.c: isum = sum_j sc_j * (q_j . a_j) in integers per 256-value
// super-block, then one f32 scale d_w * d8).  q6_K's int8 scales sit on 16-weight sub-blocks, half of the 32-deep MFMA used above,
// which forced two half-empty MFMAs and an f32 rescale per sub-block (mmsk_kernel<q6_K>: ~1 TB/s).  Here one
// v_mfma_i32_32x32x16_i8 IS one sub-block, and the operands are swapped -- activations as A, weights as B -- so that a lane's 16
// results belong to ONE weight row (the lane's) and 16 activation columns: the sub-block scale is then a per-lane scalar and is
// applied in integers, acc_i[r] += sc_j * C[r] (v_mad_i32_i24: |C| <= 16 * 32 * 127, 16 sub-blocks stay below 2^31), with one
// conversion per super-block: acc[r] += (d_row * d8_col) * float(acc_i[r]).  No digit split, no bsums, no per-sub-block float work.
// Staging as in mmsk2_kernel (coalesced loads into a wave-private LDS region); the 210-byte block is laid out as a 224-byte image:
// chunks 0..12 = bytes 0..207, chunk 13 = bytes 194..209 (so d sits at image offset 222).
template <int WAVES, int TPW = 1>
__global__ void __launch_bounds__(WAVES * 64)
mmsk6_kernel(const S2Args a) {
    constexpr int WK = WAVES / TPW;            // waves that split a tile's K (see mmsk2_kernel)
    constexpr int BLK = 210, IMG = 224, CH = 14, RPI = 64 / CH, NI = (32 + RPI - 1) / RPI;
    constexpr int APITCH = 272, WBYTES = 32 * IMG, ABYTES = 32 * APITCH, REGION = WBYTES + ABYTES + 16;   // + a dump slot
    static_assert(REGION >= 16 * 64 * 4, "the staging region doubles as the reduction buffer");
    __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES * REGION];
    __shared__ __attribute__((aligned(16))) float d8s[WAVES][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int tile = wave / WK, kw = wave % WK;
    const SkMat mt = sk_mat(a, (int)blockIdx.x * TPW + tile);
    const int m0 = mt.tile * 32;
    const bool tile_ok = m0 < mt.m;
    uint8_t * Lw = lds + wave * REGION, * La = Lw + WBYTES;
    __shared__ int s_row[32], s_pair[32];
    ColMap cm;
    if (!col_map(a.ids, a.n, tid, s_row, s_pair, cm)) return;
    // this wave's columns: a pass of an expert's pairs (IDS), or columns col0 .. col0 + 31 of the batch
    const bool ids_mode = a.ids.sorted != nullptr;
    const int ncol = ids_mode ? cm.ncol : (a.n - mt.col0 < 32 ? (a.n - mt.col0 > 0 ? a.n - mt.col0 : 0) : 32);
    auto srow = [&](int j) { return ids_mode ? s_row[j] : (mt.col0 + j < a.n ? mt.col0 + j : a.n - 1); };

    const uint8_t * wbase = mt.W + cm.w_off + (int64_t)m0 * mt.w_row_stride;
    const int wr_ = lane / CH, wc_ = lane % CH;
    uint32_t woff[NI], wlds[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int r = i * RPI + wr_;
        const bool v = lane < RPI * CH && r < 32;
        const int rr = !tile_ok ? 0 : (m0 + r < mt.m ? r : mt.m - 1 - m0);      // rows past the end repeat the last row (never stored)
        woff[i] = v ? (uint32_t)rr * (uint32_t)mt.w_row_stride + (wc_ < 13 ? wc_ * 16 : 194) : 0u;
        wlds[i] = v ? r * IMG + wc_ * 16 : WBYTES + ABYTES;
    }
    const int na = __builtin_amdgcn_readfirstlane((ncol + 3) >> 2);      // only the columns that exist are staged (see mmsk2_kernel)
    const int lrc = lr < 4 * na ? lr : 0;
    uint32_t aoff[8];
#pragma unroll
    for (int i = 0; i < 8; i++) aoff[i] = (uint32_t)srow(i * 4 + (lane >> 4)) * (uint32_t)a.k + (lane & 15) * 16;
    const uint32_t doff = (uint32_t)srow(lr) * (uint32_t)(a.k / 256) * 4;

    const int nsb = a.k / 256, per = (nsb + cm.ky_cnt - 1) / cm.ky_cnt;
    const int sb0 = cm.ky_idx * per, sb1 = sb0 + per < nsb ? sb0 + per : nsb;
    const int perw = (sb1 - sb0 + WK - 1) / WK;
    const int my0 = sb0 + kw * perw, my1 = !tile_ok ? my0 : (my0 + perw < sb1 ? my0 + perw : sb1);

    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    struct StageW { v4i w[NI]; };
    struct StageA { v4i q[8]; float d; };
    auto fetch_w = [&](int sb_) {
        StageW r;
        const uint8_t * wb = wbase + (int64_t)sb_ * BLK;
#pragma unroll
        for (int i = 0; i < NI; i++) r.w[i] = ld16_a2(wb + woff[i]);
        return r;
    };
    auto fetch_a = [&](int sb_) {
        StageA r;
        const int8_t * ab = a.aq + (int64_t)sb_ * 256;
#pragma unroll
        for (int i = 0; i < 8; i++) if (i < na) r.q[i] = *(const v4i *)(ab + aoff[i]);
        r.d = *(const float *)((const uint8_t *)a.ad + (int64_t)sb_ * 4 + doff);
        return r;
    };
    const int last = my1 - 1;
    auto clampsb = [&](int sb_) { return sb_ < last ? sb_ : last; };

    StageA sa;
    auto step = [&](int sb, StageW & sw) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; i++) *(v4i *)(Lw + wlds[i]) = sw.w[i];
#pragma unroll
        for (int i = 0; i < 8; i++) if (i < na) *(v4i *)(La + (i * 4 + (lane >> 4)) * APITCH + (lane & 15) * 16) = sa.q[i];
        if (lh == 0) d8s[wave][lr] = sa.d;
        __builtin_amdgcn_wave_barrier();
        sa = fetch_a(clampsb(sb + 1));
        sw = fetch_w(clampsb(sb + 1));

        const uint8_t * wl = Lw + lr * IMG;          // this lane's weight row (both lane halves: its two 8-weight halves of a sub-block)
        const uint8_t * al = La + lrc * APITCH + 8 * lh;  // ... and, as the A operand, activation column lr
        const v4i scv = *(const v4i *)(wl + 192);
        const float d = h2f(*(const uint16_t *)(wl + 222));
        int acc_i[16];
#pragma unroll
        for (int r = 0; r < 16; r++) acc_i[r] = 0;
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int is = 0; is < 2; is++) {
                // dequantize_row_q6_K (ggml-quants.c:1695-1711): sub-block j = 8 n + 2 qd + is takes ql[64 n + 32 (qd & 1) + 16 is ..], nibble qd >> 1,
                // and bits 2 qd of qh[32 n + 16 is ..]; this lane: 8 of its 16 weights
                const v2i ql0 = *(const v2i *)(wl + 64 * n + 16 * is + 8 * lh);
                const v2i ql1 = *(const v2i *)(wl + 64 * n + 32 + 16 * is + 8 * lh);
                const v2i qh  = *(const v2i *)(wl + 128 + 32 * n + 16 * is + 8 * lh);
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const int j = 8 * n + 2 * qd + is;
                    const v2i ql = (qd & 1) ? ql1 : ql0;
                    v2i x = ((ql >> (4 * (qd >> 1))) & 0x0f0f0f0f) | (((qh >> (2 * qd)) & 0x03030303) << 4);
                    x[0] = sub_bytes(x[0], 0x20202020); x[1] = sub_bytes(x[1], 0x20202020);
                    const v2i av = *(const v2i *)(al + 16 * j);
                    const v16i zero = {0};
                    const v16i c = __builtin_amdgcn_mfma_i32_32x32x16_i8(__builtin_bit_cast(long, av), __builtin_bit_cast(long, x), zero, 0, 0, 0);
                    const int sc = (int)(int8_t)((uint32_t)scv[j >> 2] >> (8 * (j & 3)));
#pragma unroll
                    for (int r = 0; r < 16; r++) acc_i[r] += __mul24(c[r], sc);
                }
            }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4f d8 = *(const v4f *)&d8s[wave][8 * g + 4 * lh];
#pragma unroll
            for (int e = 0; e < 4; e++) acc[4 * g + e] = __builtin_fmaf(d * d8[e], (float)acc_i[4 * g + e], acc[4 * g + e]);
        }
    };

    if (my0 < my1) {
        StageW w0 = fetch_w(my0);
        sa = fetch_a(my0);
        for (int sb = my0; sb < my1; sb++) step(sb, w0);
    }

    // add the waves' tiles (the staging regions are dead: reuse them).  C[r] of lane l: weight row l & 31, column (r&3) + 8 (r>>2) + 4 (l>>5)
    __syncthreads();
    float * red = (float *)lds;      // [WAVES][16][64]
#pragma unroll
    for (int r = 0; r < 16; r++)     // (columns that do not exist are neither written nor added)
        if ((r & 3) + 8 * (r >> 2) + 4 * lh < ncol) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    for (int e = tid; e < TPW * 16 * 64; e += WAVES * 64) {
        const int tl = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const SkMat mo = sk_mat(a, (int)blockIdx.x * TPW + tl);
        const int orow = mo.tile * 32 + (l & 31), ocol = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        const int ncol_o = ids_mode ? cm.ncol : (a.n - mo.col0 < 32 ? a.n - mo.col0 : 32);
        if (orow < mo.m && ocol < ncol_o) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < WK; w++) s += red[((tl * WK + w) * 16 + r) * 64 + l];
            float * o = mo.dst + (int64_t)(ids_mode ? s_pair[ocol] : mo.col0 + ocol) * mo.dst_col_stride + orow;
            if (cm.ky_cnt > 1) atomicAdd(o, s); else *o = s;
        }
    }
}

// Dense: where one 8-wave workgroup per tile is more than a round and a half of the chip and K is short (a wave would have two steps
// between a ~2000-cycle setup and a ~2000-cycle reduction), a tile gets TWO waves with four times the steps, in workgroups of four
// waves and two tiles (two of them share a CU, with independent barriers).  A function of (m, k) alone: see the group launch.
// MUL_MAT_ID (n_expert > 0): the grid is that many workgroups per column pass and there is about a pass per expert in use, so four
// tiles per workgroup (two waves each) still fill the chip and the fixed costs are paid once per 128 rows: 8x7B-shaped Q4_K_M pp8 /
// 16 / 32 / 64 604 / 1089 / 1995 / 3754 t/s with one tile, 638 / 1160 / 2130 / 4108 with two, 659 / 1193 / 2192 / 4183 with four
// (profiles/r03_skinny_tiles_per_workgroup.txt).  MI355_SK_TPW / MI355_SK_TPW_ID override.
static int sk_tpw(int64_t m, int64_t k, int64_t n_expert = 0) {
    static const int v = [] { const char * e = getenv("MI355_SK_TPW"); return e ? atoi(e) : 0; }();
    static const int vi = [] { const char * e = getenv("MI355_SK_TPW_ID"); return e ? atoi(e) : 0; }();
    const int64_t nb = (m + 31) / 32;
    if (n_expert > 0) {
        if (vi) return vi == 42 ? 42 : vi == 4 ? 4 : vi == 2 ? 2 : 1;
        return (nb + 3) / 4 * n_expert >= 256 ? 4 : (nb + 1) / 2 * n_expert >= 256 ? 2 : 1;
    }
    if (v) return v == 2 ? 2 : 1;
    return nb >= 384 && k <= 8192 ? 2 : 1;
}

// Fewer than 256 row tiles: K is cut over two workgroups (atomic adds onto a zeroed dst) only where K is long.  At K = 4096 a wave of
// an uncut tile has two steps instead of one, which costs less than the memset in front of the launch: 8B Q4_K_M steps of 8 / 16 / 32
// tokens 2117 / 3931 / 6643 t/s cut from 2048, 2116 / 3962 / 6818 from 8192, 2037 / 3741 / 6441 never cut (ffn_down, K = 14336, wants the cut).
// MI355_SK_KY_MIN_K overrides.
static int sk_ky_min_k() {
    static const int v = [] { const char * e = getenv("MI355_SK_KY_MIN_K"); return e ? atoi(e) : 8193; }();
    return v;
}

// 33..64 columns (a.ct == 2): the two column halves of a row tile are two "tiles" of one workgroup -- the weights cross HBM once and
// the launch's fixed costs are paid once, where two passes of 32 pay both twice; a tile's K is cut over half as many waves.
static int launch6(const S2Args & a, hipStream_t s, const char * tag) {
    const unsigned nb = (unsigned)((a.m + 31) / 32);
    const int64_t m_cut = a.m_cut ? a.m_cut : a.m;
    if (a.ct == 2) {
        if (sk_tpw(m_cut, a.k) == 2) MI355_LAUNCH(tag, (mmsk6_kernel<8, 4>), dim3((nb + 1) / 2, 1), dim3(512), 0, s, a);
        else                       MI355_LAUNCH(tag, (mmsk6_kernel<8, 2>), dim3(nb, 1), dim3(512), 0, s, a);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    if (sk_tpw(m_cut, a.k) == 2) {
        MI355_LAUNCH(tag, (mmsk6_kernel<4, 2>), dim3((nb + 1) / 2, 1), dim3(256), 0, s, a);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    const unsigned ky = (m_cut + 31) / 32 < 256 && a.k >= sk_ky_min_k() ? 2 : 1;
    if (ky > 1) {
        hipError_t e = a.dst_col_stride == a.m
            ? hipMemsetAsync(a.dst, 0, (size_t)a.n * a.m * sizeof(float), s)
            : hipMemset2DAsync(a.dst, (size_t)a.dst_col_stride * sizeof(float), 0, (size_t)a.m * sizeof(float), (size_t)a.n, s);
        if (e != hipSuccess) return (int)e;
    }
    MI355_LAUNCH(tag, (mmsk6_kernel<8>), dim3(nb, ky), dim3(512), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

template <int TYPE>
static int launch2(const S2Args & a, hipStream_t s, const char * tag) {
    const unsigned nb = (unsigned)((a.m + 31) / 32);
    const int64_t m_cut = a.m_cut ? a.m_cut : a.m;
    if (a.ct == 2) {
        if (sk_tpw(m_cut, a.k) == 2) MI355_LAUNCH(tag, (mmsk2_kernel<TYPE, 8, 4>), dim3((nb + 1) / 2, 1), dim3(512), 0, s, a);
        else                       MI355_LAUNCH(tag, (mmsk2_kernel<TYPE, 8, 2>), dim3(nb, 1), dim3(512), 0, s, a);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    if (sk_tpw(m_cut, a.k) == 2) {
        MI355_LAUNCH(tag, (mmsk2_kernel<TYPE, 4, 2>), dim3((nb + 1) / 2, 1), dim3(256), 0, s, a);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    const unsigned ky = (m_cut + 31) / 32 < 256 && a.k >= sk_ky_min_k() ? 2 : 1;
    if (ky > 1) {
        hipError_t e = a.dst_col_stride == a.m
            ? hipMemsetAsync(a.dst, 0, (size_t)a.n * a.m * sizeof(float), s)
            : hipMemset2DAsync(a.dst, (size_t)a.dst_col_stride * sizeof(float), 0, (size_t)a.m * sizeof(float), (size_t)a.n, s);
        if (e != hipSuccess) return (int)e;
    }
    MI355_LAUNCH(tag, (mmsk2_kernel<TYPE, 8>), dim3(nb, ky), dim3(512), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// q8_K activation quantization (quantize_row_q8_K_ref, ggml-quants.c:2471-2508), one wave per 256 values, 4 per lane.
// SoA output: q int8 [rows][k], d f32 [rows][k/256], bsums int16 [rows][k/16].
__global__ void __launch_bounds__(256)
quant_q8k_kernel(const char * x, int8_t * q, float * dd, int16_t * bs, int k, int ne1, int ne2, int64_t nb1, int64_t nb2, int64_t nb3, int64_t total) {
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= total) return;
    const int lane = threadIdx.x & 63, bpr = k / 256;
    const int64_t row = blk / bpr;
    const int ib = (int)(blk % bpr);
    const int i1 = (int)(row % ne1), i2 = (int)((row / ne1) % ne2);
    const int64_t i3 = row / ((int64_t)ne1 * ne2);
    const v4f v = *(const v4f *)((const float *)(x + i1 * nb1 + i2 * nb2 + i3 * nb3) + ib * 256 + lane * 4);
    // the value of largest magnitude, first occurrence on ties (the reference scans with a strict >)
    float ax = 0.0f, mine = 0.0f;
    int idx = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const float t = fabsf(v[j]); if (t > ax) { ax = t; idx = j; mine = v[j]; } }
    float amax = ax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    int first = ax == amax ? lane * 4 + idx : 1024;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    const float maxv = __shfl(mine, (first >> 2) & 63, 64);
    int packed = 0, sum = 0;
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = __fdiv_rn(-127.0f, maxv);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // nearest_int (ggml-quants.c:559-565): round to nearest even through the 2^23 + 2^22 magic number
            const float val = __fadd_rn(__fmul_rn(iscale, v[j]), 12582912.0f);
            int qi = (__float_as_int(val) & 0x007fffff) - 0x00400000;
            qi = qi < 127 ? qi : 127;
            sum += qi;
            packed |= (qi & 0xff) << (8 * j);
        }
        d = __fdiv_rn(1.0f, iscale);
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    *(int *)(q + row * k + ib * 256 + lane * 4) = packed;
    if ((lane & 3) == 0) bs[row * (k / 16) + ib * 16 + (lane >> 2)] = (int16_t)sum;
    if (lane == 0) dd[row * bpr + ib] = d;
}

} // namespace

extern "C" int mi355_mul_mat_skinny(int type, const void * W, int64_t w_row_stride, const int8_t * act_q, const float * act_ds,
                                    float * dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n, void * stream) {
    return mi355_mul_mat_skinny_slice(type, W, w_row_stride, act_q, act_ds, dst, dst_col_stride, m, m, k, n, stream);
}

extern "C" int mi355_mul_mat_skinny_slice(int type, const void * W, int64_t w_row_stride, const int8_t * act_q, const float * act_ds,
                                          float * dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n, void * stream) {
    if (!W || !act_q || !act_ds || !dst) return MI355_E_NULL;
    if (m_total < m || m_total > 0x7fffffff) return MI355_E_SHAPE;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0 || type == MI355_TYPE_F32 || type == MI355_TYPE_F16 || type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (m < 0 || n < 0 || k <= 0 || k % bs != 0 || m > 0x7fffffff || k > 0x7fffffff || dst_col_stride < m) return MI355_E_SHAPE;
    if (k % 256 != 0 || n > 32) return MI355_E_UNSUPPORTED;      // callers split wider batches into passes of 32 columns
    if (((uintptr_t)act_q & 15) || ((uintptr_t)act_ds & 7) || ((uintptr_t)dst & 3)) return MI355_E_ALIGN;
    const bool a16 = type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K;
    if (((uintptr_t)W | (uintptr_t)w_row_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (w_row_stride < mi355_row_size(type, k)) return MI355_E_SHAPE;
    if (m == 0 || n == 0) return MI355_OK;
    SArgs a = {};
    a.W = (const uint8_t *)W; a.w_row_stride = w_row_stride; a.aq = act_q; a.ads = act_ds;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.m = (int)m; a.n = (int)n; a.k = (int)k; a.m_cut = (int)m_total;
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
        case MI355_TYPE_Q4_0: return launch<MI355_TYPE_Q4_0>(a, s, "mmsk_q4_0");
        case MI355_TYPE_Q8_0: return launch<MI355_TYPE_Q8_0>(a, s, "mmsk_q8_0");
        case MI355_TYPE_Q4_K: return launch<MI355_TYPE_Q4_K>(a, s, "mmsk_q4_K");
        case MI355_TYPE_Q5_K: return launch<MI355_TYPE_Q5_K>(a, s, "mmsk_q5_K");
        case MI355_TYPE_Q6_K: return launch<MI355_TYPE_Q6_K>(a, s, "mmsk_q6_K");
    }
    return MI355_E_TYPE;
}

extern "C" int mi355_quantize_act_q8k(const float * x, int8_t * q, float * d, int16_t * bsums, int64_t k, int64_t ne1, int64_t ne2, int64_t ne3,
                                      int64_t nb1, int64_t nb2, int64_t nb3, void * stream) {
    if (!x || !q || !d || !bsums) return MI355_E_NULL;
    if (k <= 0 || k % 256 != 0 || ne1 < 0 || ne2 < 0 || ne3 < 0 || k > 0x7fffffff || ne1 > 0x7fffffff || ne2 > 0x7fffffff) return MI355_E_SHAPE;
    const int64_t rows = ne1 * ne2 * ne3;
    if (((uintptr_t)x | (uintptr_t)nb1 | (uintptr_t)nb2 | (uintptr_t)nb3) & 15) return MI355_E_ALIGN;
    if (((uintptr_t)q & 15) || ((uintptr_t)d & 3) || ((uintptr_t)bsums & 15)) return MI355_E_ALIGN;
    if (rows == 0) return MI355_OK;
    const int64_t total = rows * (k / 256);
    if ((total + 3) / 4 > 0x7fffffff) return MI355_E_SHAPE;
    MI355_LAUNCH("quantize_q8_K", quant_q8k_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                 (const char *)x, q, d, bsums, (int)k, (int)ne1, (int)ne2, nb1, nb2, nb3, total);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_mul_mat_skinny_k(int type, const void * W, int64_t w_row_stride, const int8_t * act_q, const float * act_d,
                                      const int16_t * act_bsums, float * dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n,
                                      void * stream) {
    return mi355_mul_mat_skinny_k_slice(type, W, w_row_stride, act_q, act_d, act_bsums, dst, dst_col_stride, m, m, k, n, stream);
}

// m rows of a matrix of m_total rows (a row-split slice): how a tile's K is cut over waves and workgroups follows the WHOLE matrix, so
// the slice's results are the bits the unsplit launch gives those rows
extern "C" int mi355_mul_mat_skinny_k_slice(int type, const void * W, int64_t w_row_stride, const int8_t * act_q, const float * act_d,
                                            const int16_t * act_bsums, float * dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k,
                                            int64_t n, void * stream) {
    if (!W || !act_q || !act_d || !act_bsums || !dst) return MI355_E_NULL;
    if (m_total < m || m_total > 0x7fffffff) return MI355_E_SHAPE;
    if (type != MI355_TYPE_Q4_K && type != MI355_TYPE_Q5_K && type != MI355_TYPE_Q6_K) return mi355_blck_size(type) ? MI355_E_UNSUPPORTED : MI355_E_TYPE;
    if (m < 0 || n < 0 || k <= 0 || k % 256 != 0 || m > 0x7fffffff || k > 0x7fffffff || dst_col_stride < m) return MI355_E_SHAPE;
    if (n > 64) return MI355_E_UNSUPPORTED;
    if (w_row_stride > 0x3ffffff || k > 0x1ffffff) return MI355_E_UNSUPPORTED;      // 32-bit lane offsets: 32 rows / 64 columns
    if (((uintptr_t)act_q & 15) || ((uintptr_t)act_d & 3) || ((uintptr_t)act_bsums & 15) || ((uintptr_t)dst & 3)) return MI355_E_ALIGN;
    if (((uintptr_t)W | (uintptr_t)w_row_stride) & (type == MI355_TYPE_Q6_K ? 1 : 15)) return MI355_E_ALIGN;
    if (w_row_stride < mi355_row_size(type, k)) return MI355_E_SHAPE;
    if (m == 0 || n == 0) return MI355_OK;
    S2Args a = {};
    a.W = (const uint8_t *)W; a.w_row_stride = w_row_stride; a.aq = act_q; a.ad = act_d; a.bs = act_bsums;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.m = (int)m; a.n = (int)n; a.k = (int)k; a.ct = n > 32 ? 2 : 1; a.m_cut = (int)m_total;
    hipStream_t s = (hipStream_t)stream;
    if (type == MI355_TYPE_Q6_K) return launch6(a, s, "mmsk6_q6_K");
    if (type == MI355_TYPE_Q4_K) return launch2<MI355_TYPE_Q4_K>(a, s, "mmsk2_q4_K");
    return launch2<MI355_TYPE_Q5_K>(a, s, "mmsk2_q5_K");
}

extern "C" int mi355_mul_mat_skinny_k_group(int type, int n_mats, const void * const * W, const int64_t * w_row_stride, float * const * dst,
                                            const int64_t * dst_col_stride, const int64_t * m, const int8_t * act_q, const float * act_d,
                                            const int16_t * act_bsums, int64_t k, int64_t n, void * stream) {
    if (!W || !w_row_stride || !dst || !dst_col_stride || !m || !act_q || !act_d || !act_bsums) return MI355_E_NULL;
    if (type != MI355_TYPE_Q4_K && type != MI355_TYPE_Q5_K && type != MI355_TYPE_Q6_K) return mi355_blck_size(type) ? MI355_E_UNSUPPORTED : MI355_E_TYPE;
    if (n_mats < 1 || n_mats > 4) return MI355_E_SHAPE;
    if (n < 0 || k <= 0 || k % 256 != 0 || k > 0x1ffffff) return MI355_E_SHAPE;
    if (n > 64) return MI355_E_UNSUPPORTED;
    if (((uintptr_t)act_q & 15) || ((uintptr_t)act_d & 3) || ((uintptr_t)act_bsums & 15)) return MI355_E_ALIGN;
    S2Args a = {};
    a.aq = act_q; a.ad = act_d; a.bs = act_bsums; a.n = (int)n; a.k = (int)k; a.ct = n > 32 ? 2 : 1;
    int64_t tiles = 0;
    for (int i = 0; i < n_mats; i++) {
        if (!W[i] || !dst[i]) return MI355_E_NULL;
        if (m[i] <= 0 || m[i] > 0x7fffffff || dst_col_stride[i] < m[i] || w_row_stride[i] < mi355_row_size(type, k)) return MI355_E_SHAPE;
        if (w_row_stride[i] > 0x3ffffff) return MI355_E_UNSUPPORTED;
        if ((((uintptr_t)W[i] | (uintptr_t)w_row_stride[i]) & (type == MI355_TYPE_Q6_K ? 1 : 15)) || ((uintptr_t)dst[i] & 3)) return MI355_E_ALIGN;
        tiles += (m[i] + 31) / 32;
        a.grp.W[i] = (const uint8_t *)W[i]; a.grp.w_row_stride[i] = w_row_stride[i]; a.grp.dst[i] = dst[i];
        a.grp.dst_col_stride[i] = dst_col_stride[i]; a.grp.m[i] = (int)m[i]; a.grp.tile_end[i] = (int)tiles;
    }
    if (tiles > 0x3fffffff) return MI355_E_SHAPE;
    for (int i = n_mats; i < 4; i++) a.grp.tile_end[i] = (int)tiles;
    a.grp.n = n_mats;
    if (n == 0) return MI355_OK;
    // K is never cut over two workgroups here: no memset, no atomics
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)tiles;
    // Every matrix is cut over the waves exactly as its own launch would cut it (sk_tpw and the K cut depend on (m, k) alone), so a
    // matrix gives the same bits alone and in a group; matrices that disagree, or whose own launch cuts K over two workgroups, are
    // left to the caller to launch one by one.
    const int tpw = sk_tpw(m[0], k);
    for (int i = 0; i < n_mats; i++) {
        if (sk_tpw(m[i], k) != tpw) return MI355_E_UNSUPPORTED;
        if ((m[i] + 31) / 32 < 256 && k >= sk_ky_min_k()) return MI355_E_UNSUPPORTED;
    }
    if (a.ct == 2) {     // 33..64 columns: column halves as tiles (see launch2)
        if (tpw == 2) {
            const unsigned g = (nb + 1) / 2;
            if (type == MI355_TYPE_Q6_K)      MI355_LAUNCH("mmsk6_group_q6_K", (mmsk6_kernel<8, 4>), dim3(g, 1), dim3(512), 0, s, a);
            else if (type == MI355_TYPE_Q4_K) MI355_LAUNCH("mmsk2_group_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 8, 4>), dim3(g, 1), dim3(512), 0, s, a);
            else                              MI355_LAUNCH("mmsk2_group_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 8, 4>), dim3(g, 1), dim3(512), 0, s, a);
        } else {
            if (type == MI355_TYPE_Q6_K)      MI355_LAUNCH("mmsk6_group_q6_K", (mmsk6_kernel<8, 2>), dim3(nb, 1), dim3(512), 0, s, a);
            else if (type == MI355_TYPE_Q4_K) MI355_LAUNCH("mmsk2_group_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 8, 2>), dim3(nb, 1), dim3(512), 0, s, a);
            else                              MI355_LAUNCH("mmsk2_group_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 8, 2>), dim3(nb, 1), dim3(512), 0, s, a);
        }
    }
    else if (tpw == 2) {      // four waves, two tiles (a workgroup's tiles may belong to two matrices: the lookup is per wave)
        const unsigned nb2 = (nb + 1) / 2;
        if (type == MI355_TYPE_Q6_K)      MI355_LAUNCH("mmsk6_group_q6_K", (mmsk6_kernel<4, 2>), dim3(nb2, 1), dim3(256), 0, s, a);
        else if (type == MI355_TYPE_Q4_K) MI355_LAUNCH("mmsk2_group_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 4, 2>), dim3(nb2, 1), dim3(256), 0, s, a);
        else                              MI355_LAUNCH("mmsk2_group_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 4, 2>), dim3(nb2, 1), dim3(256), 0, s, a);
    }
    else if (type == MI355_TYPE_Q6_K) MI355_LAUNCH("mmsk6_group_q6_K", (mmsk6_kernel<8>), dim3(nb, 1), dim3(512), 0, s, a);
    else if (type == MI355_TYPE_Q4_K) MI355_LAUNCH("mmsk2_group_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 8>), dim3(nb, 1), dim3(512), 0, s, a);
    else                              MI355_LAUNCH("mmsk2_group_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 8>), dim3(nb, 1), dim3(512), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// ---- MUL_MAT_ID for 2 .. a few hundred (token, slot) pairs: every expert's matrix is streamed ONCE per 32 of its
// columns (the mat-vec path reads it once per pair; the grouped GEMM fills a 128-column tile with a handful of columns)
static int64_t ids_ws_layout(int type, int64_t k, int64_t rows, int64_t * b0, int64_t * b1, int64_t * b2) {
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    if (type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K || type == MI355_TYPE_Q6_K) { *b0 = up(rows * k); *b1 = up(rows * (k / 256) * 4); *b2 = up(rows * (k / 16) * 2); }
    else                                                    { *b0 = up(rows * k); *b1 = up(rows * (k / 32) * 8);  *b2 = 0; }
    return *b0 + *b1 + *b2;
}

extern "C" int64_t mi355_mul_mat_id_skinny_workspace_bytes(int type, int64_t k, int64_t n_tokens, int64_t b_ne1, int64_t n_used, int64_t n_expert) {
    if (k <= 0 || n_tokens < 0 || b_ne1 < 1 || n_used < 1 || n_expert < 1) return -1;
    int64_t b0, b1, b2;
    return ids_ws_layout(type, k, n_tokens * b_ne1, &b0, &b1, &b2) + (n_tokens * n_used + 2 * (n_expert + 1)) * 4 + 256;
}

static int mul_mat_id_skinny_impl(int type, const void * as, const void * as2, int64_t w_row_stride, int64_t w_expert_stride, const float * b,
                                  int64_t b_nb1, int64_t b_nb2, const int32_t * ids, int64_t ids_stride, float * dst, float * dst2, int64_t m,
                                  int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1, void * workspace,
                                  int64_t workspace_bytes, void * stream, int32_t * tables = nullptr, int flags = 0);

extern "C" int64_t mi355_mul_mat_id_tables_bytes(int64_t n_tokens, int64_t n_used, int64_t n_expert) {
    if (n_tokens < 0 || n_used < 1 || n_expert < 1) return -1;
    return ((n_tokens * n_used + 2 * (n_expert + 1)) * 4 + 255) / 256 * 256;
}

// as2 / dst2 may be NULL (one tensor).  tables: the routing tables' place when they should outlive the call (else inside the workspace);
// flags: MI355_IDS_ACT_READY the workspace already starts with b's activation planes, MI355_IDS_TABLES_READY `tables` already holds the
// tables of these ids
extern "C" int mi355_mul_mat_id_skinny_ex(int type, const void * as0, const void * as1, int64_t w_row_stride, int64_t w_expert_stride,
                                          const float * b, int64_t b_nb1, int64_t b_nb2, const int32_t * ids, int64_t ids_stride, float * dst0,
                                          float * dst1, int64_t m, int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1,
                                          void * workspace, int64_t workspace_bytes, int32_t * tables, int flags, void * stream) {
    if ((as1 == nullptr) != (dst1 == nullptr)) return MI355_E_NULL;
    if (flags & ~3) return MI355_E_UNSUPPORTED;
    if ((flags & MI355_IDS_TABLES_READY) && !tables) return MI355_E_NULL;
    if ((uintptr_t)tables & 3) return MI355_E_ALIGN;
    if (as1 && type != MI355_TYPE_Q4_K && type != MI355_TYPE_Q5_K && type != MI355_TYPE_Q6_K) return mi355_blck_size(type) ? MI355_E_UNSUPPORTED : MI355_E_TYPE;
    return mul_mat_id_skinny_impl(type, as0, as1, w_row_stride, w_expert_stride, b, b_nb1, b_nb2, ids, ids_stride, dst0, dst1, m, k, n_expert, n_used,
                                  n_tokens, b_ne1, workspace, workspace_bytes, stream, tables, flags);
}

extern "C" int mi355_mul_mat_id_skinny(int type, const void * as, int64_t w_row_stride, int64_t w_expert_stride, const float * b,
                                       int64_t b_nb1, int64_t b_nb2, const int32_t * ids, int64_t ids_stride, float * dst, int64_t m,
                                       int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1, void * workspace,
                                       int64_t workspace_bytes, void * stream) {
    return mul_mat_id_skinny_impl(type, as, nullptr, w_row_stride, w_expert_stride, b, b_nb1, b_nb2, ids, ids_stride, dst, nullptr, m, k, n_expert, n_used,
                                  n_tokens, b_ne1, workspace, workspace_bytes, stream);
}

extern "C" int mi355_mul_mat_id_skinny_pair(int type, const void * as0, const void * as1, int64_t w_row_stride, int64_t w_expert_stride,
                                            const float * b, int64_t b_nb1, int64_t b_nb2, const int32_t * ids, int64_t ids_stride, float * dst0,
                                            float * dst1, int64_t m, int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1,
                                            void * workspace, int64_t workspace_bytes, void * stream) {
    if (!as1 || !dst1) return MI355_E_NULL;
    if (type != MI355_TYPE_Q4_K && type != MI355_TYPE_Q5_K && type != MI355_TYPE_Q6_K) return mi355_blck_size(type) ? MI355_E_UNSUPPORTED : MI355_E_TYPE;
    return mul_mat_id_skinny_impl(type, as0, as1, w_row_stride, w_expert_stride, b, b_nb1, b_nb2, ids, ids_stride, dst0, dst1, m, k, n_expert, n_used,
                                  n_tokens, b_ne1, workspace, workspace_bytes, stream);
}

// as2 / dst2 (K-quants only): a second expert tensor of the same type, shape and strides on the same b and ids -- ffn_up_exps and
// ffn_gate_exps of a MoE layer -- computed by the same launch (blockIdx.x runs over the row tiles of both, S2Args.grp): one activation
// quantization, one bucketing and one grid instead of two of each.  Each result has the bits of its own call (the cut of a tile over
// the waves is the same function of (m, k, n_expert) in both).
static int mul_mat_id_skinny_impl(int type, const void * as, const void * as2, int64_t w_row_stride, int64_t w_expert_stride, const float * b,
                                  int64_t b_nb1, int64_t b_nb2, const int32_t * ids, int64_t ids_stride, float * dst, float * dst2, int64_t m,
                                  int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1, void * workspace,
                                  int64_t workspace_bytes, void * stream, int32_t * tables, int flags) {
    if (!as || !b || !ids || !dst || !workspace) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0 || type == MI355_TYPE_F32 || type == MI355_TYPE_F16 || type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (m < 0 || k <= 0 || k % bs != 0 || n_expert < 1 || n_expert > 256 || n_used < 1 || n_tokens < 0 || (b_ne1 != 1 && b_ne1 != n_used) ||
        m > 0x7fffffff || k > 0x3ffffff || n_tokens * n_used > 0x3fffffff || w_row_stride > 0x3ffffff) return MI355_E_SHAPE;
    if (k % 256 != 0) return MI355_E_UNSUPPORTED;
    if (workspace_bytes < mi355_mul_mat_id_skinny_workspace_bytes(type, k, n_tokens, b_ne1, n_used, n_expert)) return MI355_E_WORKSPACE;
    if (((uintptr_t)workspace & 15) || ((uintptr_t)dst & 3) || ((uintptr_t)ids & 3)) return MI355_E_ALIGN;
    const bool kq = type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K || type == MI355_TYPE_Q6_K;      // q8_K activations
    if (((uintptr_t)as | (uintptr_t)as2 | (uintptr_t)w_row_stride | (uintptr_t)w_expert_stride) & (kq && type != MI355_TYPE_Q6_K ? 15 : 1)) return MI355_E_ALIGN;
    if ((uintptr_t)dst2 & 3) return MI355_E_ALIGN;
    if (w_row_stride < mi355_row_size(type, k)) return MI355_E_SHAPE;
    if (m == 0 || n_tokens == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = n_tokens * b_ne1;
    int64_t b0, b1, b2;
    const int64_t act_bytes = ids_ws_layout(type, k, rows, &b0, &b1, &b2);
    char * ws = (char *)workspace;
    int * sorted = tables ? tables : (int *)(ws + act_bytes);
    int * off = sorted + n_tokens * n_used, * tile_off = off + n_expert + 1;
    // activations of all tokens, then the routing tables
    if (!(flags & MI355_IDS_ACT_READY)) {
        const int rc = kq ? mi355_quantize_act_q8k(b, (int8_t *)ws, (float *)(ws + b0), (int16_t *)(ws + b0 + b1), k, b_ne1, n_tokens, 1, b_nb1, b_nb2, 0, stream)
                          : mi355_quantize_act(b, (int8_t *)ws, (float *)(ws + b0), k, b_ne1, n_tokens, 1, b_nb1, b_nb2, 0, stream);
        if (rc != MI355_OK) return rc;
    }
    if (!(flags & MI355_IDS_TABLES_READY)) {
        MI355_LAUNCH("moe_bucket", moe_bucket_kernel, dim3(1), dim3(1024), 0, s, ids, ids_stride, (int)n_tokens, (int)n_used, (int)n_expert, 32,
                     sorted, off, tile_off);
        MI355_CHECK_LAUNCH();
    }
    IdsArgs I;
    I.sorted = sorted; I.off = off; I.tile_off = tile_off; I.n_expert = (int)n_expert; I.n_used = (int)n_used; I.act_ne1 = (int)b_ne1;
    I.w_expert_stride = w_expert_stride;
    const unsigned nb1 = (unsigned)((m + 31) / 32);
    const unsigned max_tiles = (unsigned)((n_tokens * n_used + 31) / 32 + n_expert);   // sum over experts of ceil(cnt_e / 32) never exceeds this
    if (max_tiles > 65535) return MI355_E_UNSUPPORTED;
    if (kq) {
        S2Args a = {};
        a.W = (const uint8_t *)as; a.w_row_stride = w_row_stride; a.aq = (const int8_t *)ws; a.ad = (const float *)(ws + b0);
        a.bs = (const int16_t *)(ws + b0 + b1); a.dst = dst; a.dst_col_stride = m; a.m = (int)m; a.n = 32; a.k = (int)k; a.ids = I;
        unsigned nb = nb1;
        if (as2) {
            a.grp.n = 2;
            a.grp.W[0] = (const uint8_t *)as; a.grp.W[1] = (const uint8_t *)as2; a.grp.dst[0] = dst; a.grp.dst[1] = dst2;
            for (int i = 0; i < 2; i++) { a.grp.w_row_stride[i] = w_row_stride; a.grp.dst_col_stride[i] = m; a.grp.m[i] = (int)m; a.grp.tile_end[i] = (int)nb1 * (i + 1); }
            a.grp.tile_end[2] = a.grp.tile_end[3] = 2 * (int)nb1;
            nb = 2 * nb1;
        }
        const int tpw = sk_tpw(m, k, n_expert);
        if (tpw == 42) {      // lab: two tiles x two waves in 4-wave workgroups (two per CU)
            const unsigned nb2 = (nb + 1) / 2;
            if (type == MI355_TYPE_Q4_K)      MI355_LAUNCH("mmsk2_id_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 4, 2>), dim3(nb2, max_tiles), dim3(256), 0, s, a);
            else if (type == MI355_TYPE_Q5_K) MI355_LAUNCH("mmsk2_id_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 4, 2>), dim3(nb2, max_tiles), dim3(256), 0, s, a);
            else                              MI355_LAUNCH("mmsk6_id_q6_K", (mmsk6_kernel<4, 2>), dim3(nb2, max_tiles), dim3(256), 0, s, a);
        } else if (tpw == 4) {
            const unsigned nb4 = (nb + 3) / 4;
            if (type == MI355_TYPE_Q4_K)      MI355_LAUNCH("mmsk2_id_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 8, 4>), dim3(nb4, max_tiles), dim3(512), 0, s, a);
            else if (type == MI355_TYPE_Q5_K) MI355_LAUNCH("mmsk2_id_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 8, 4>), dim3(nb4, max_tiles), dim3(512), 0, s, a);
            else                              MI355_LAUNCH("mmsk6_id_q6_K", (mmsk6_kernel<8, 4>), dim3(nb4, max_tiles), dim3(512), 0, s, a);
        } else if (tpw == 2) {
            const unsigned nb2 = (nb + 1) / 2;
            if (type == MI355_TYPE_Q4_K)      MI355_LAUNCH("mmsk2_id_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 8, 2>), dim3(nb2, max_tiles), dim3(512), 0, s, a);
            else if (type == MI355_TYPE_Q5_K) MI355_LAUNCH("mmsk2_id_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 8, 2>), dim3(nb2, max_tiles), dim3(512), 0, s, a);
            else                              MI355_LAUNCH("mmsk6_id_q6_K", (mmsk6_kernel<8, 2>), dim3(nb2, max_tiles), dim3(512), 0, s, a);
        }
        else if (type == MI355_TYPE_Q4_K) MI355_LAUNCH("mmsk2_id_q4_K", (mmsk2_kernel<MI355_TYPE_Q4_K, 8>), dim3(nb, max_tiles), dim3(512), 0, s, a);
        else if (type == MI355_TYPE_Q5_K) MI355_LAUNCH("mmsk2_id_q5_K", (mmsk2_kernel<MI355_TYPE_Q5_K, 8>), dim3(nb, max_tiles), dim3(512), 0, s, a);
        else                              MI355_LAUNCH("mmsk6_id_q6_K", (mmsk6_kernel<8>), dim3(nb, max_tiles), dim3(512), 0, s, a);
    } else {
        if (as2) return MI355_E_UNSUPPORTED;
        const unsigned nb = nb1;
        SArgs a = {};
        a.W = (const uint8_t *)as; a.w_row_stride = w_row_stride; a.aq = (const int8_t *)ws; a.ads = (const float *)(ws + b0);
        a.dst = dst; a.dst_col_stride = m; a.m = (int)m; a.n = 32; a.k = (int)k; a.ids = I;
        switch (type) {
            case MI355_TYPE_Q8_0: MI355_LAUNCH("mmsk_id_q8_0", (mmsk_kernel<MI355_TYPE_Q8_0>), dim3(nb, max_tiles), dim3(SK_THREADS), 0, s, a); break;
            case MI355_TYPE_Q4_0: MI355_LAUNCH("mmsk_id_q4_0", (mmsk_kernel<MI355_TYPE_Q4_0>), dim3(nb, max_tiles), dim3(SK_THREADS), 0, s, a); break;
            default: return MI355_E_TYPE;
        }
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}
