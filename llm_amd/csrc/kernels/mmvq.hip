// mmvq.hip -- decode mat-vec on GGUF block-quantized weights for gfx950.
//
// Replaces mul_mat_vec_q / ggml_cuda_mul_mat_vec_q (ggml/src/ggml-cuda/mmvq.cu:137-224,491-561) and
// the vec_dot_*_q8_1 family (ggml/src/ggml-cuda/vecdotq.cuh); not derived from them: the work
// split, load shapes and arithmetic form are chosen for CDNA4.
//
// Roofline: HBM.  Algorithmic bytes per launch = M*(K/bs)*B + 4K + 4M.
//
// Shape of the kernel
//   * a wavefront (64 lanes) owns RPW consecutive weight rows; a 256-thread workgroup = 4 waves.
//   * the row is cut into "units" of 16..48 bytes of quant data; lane l takes units l, l+64, ...
//     so one wave-load covers a contiguous 1 KiB (q4_K: 8 lanes x 16 B per 144 B block) of the
//     row with 128-bit loads: weights go straight to VGPRs (streamed once: nontemporal), no LDS.
//   * the nibbles / 6-bit fields are unpacked with a handful of bit ops per dword and fed to
//     v_dot4c_i32_i8 against the q8_1 activations (internal SoA layout, see mi355_quantize_act);
//     integer sums are exact, the per-32 scales are applied in f32.
//   * wavefront-level xor-shuffle reduction, one f32 store per (row, column).
//
// Arithmetic contract (tests/test_mmvq.py): dst = sum over 32-blocks of
//     d8 * ( dW_j * sum(qW*q8) - mW_j * sum(q8) )        (q4_K / q5_K)
//     d8 * dW_j * sum((qW - off)*q8)                     (q4_0: off 8, q6_K: off 32, q8_0: off 0)
// i.e. exactly dequant(W) . dequant_q8_1(x) up to f32 summation order.
#include "common.h"

namespace {

struct ActPtr {
    const int8_t * q;   // [k]
    const float  * ds;  // [k/32][2]
};

// ------------------------------------------------------------------------------------------------
// Per-type unit traits.  A unit is what one lane processes per step:
//   ELEMS      weights covered
//   Raw        registers as loaded;  load(row_ptr, u)
//   Dec        decoded int8x4 dwords + f32 scales;  decode(raw, u)
//   Act        activation registers for the unit;  load_act(act, u)
//   dot(dec, act) -> f32 partial
// ------------------------------------------------------------------------------------------------
template <int TYPE> struct Unit;

// ---- Q4_K: 144 B super-block = {half d, half dmin, u8 scales[12], u8 qs[128]} (ggml-common.h:285-295)
// unit = 16 B of qs: chunk p (0..7) of block b.  Low nibbles are sub-block 2g, high nibbles 2g+1
// (g = p/2), elements 16*(p&1) .. +15 of each (dequantize_row_q4_K, ggml-quants.c:1286-1293).
struct K4Scales { float dsc0, dsc1, dm0, dm1; };

static __device__ __forceinline__ K4Scales k4_scales(const v4i hdr, int g) {
    // hdr.x = {d, dmin}; hdr.y/z/w = the 12 packed 6-bit scales/mins (get_scale_min_k4, ggml-quants.c:625-632)
    const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;
    const uint32_t sc_lo = u0 & 0x3f3f3f3fu;
    const uint32_t m_lo  = u1 & 0x3f3f3f3fu;
    const uint32_t sc_hi = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
    const uint32_t m_hi  = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
    const int sh = (g & 1) * 16;
    const uint32_t sc2 = ((g & 2) ? sc_hi : sc_lo) >> sh;
    const uint32_t m2  = ((g & 2) ? m_hi : m_lo) >> sh;
    const float d    = h2f((uint16_t)(hdr.x & 0xffff));
    const float dmin = h2f((uint16_t)((uint32_t)hdr.x >> 16));
    K4Scales s;
    s.dsc0 = d * (float)(sc2 & 0xff);
    s.dsc1 = d * (float)((sc2 >> 8) & 0xff);
    s.dm0  = dmin * (float)(m2 & 0xff);
    s.dm1  = dmin * (float)((m2 >> 8) & 0xff);
    return s;
}

struct Act2x16 { v4i lo, hi; v4f ds; int slo, shi; }; // 2 x 16 int8, {d0, s0, d1, s1}, in-lane sums

static __device__ __forceinline__ Act2x16 load_act_k45(const ActPtr a, int u) {
    const int b = u >> 3, p = u & 7, g = p >> 1;
    const int e0 = b * QK_K + g * 64 + (p & 1) * 16;
    Act2x16 r;
    r.lo = *(const v4i *)(a.q + e0);
    r.hi = *(const v4i *)(a.q + e0 + 32);
    r.ds = *(const v4f *)(a.ds + (b * 8 + g * 2) * 2);
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { s0 = dot4(0x01010101, r.lo[i], s0); s1 = dot4(0x01010101, r.hi[i], s1); }
    r.slo = s0; r.shi = s1;
    return r;
}

struct DecK45 { v4i lo, hi; K4Scales s; };

static __device__ __forceinline__ float dot_k45(const DecK45 & w, const Act2x16 & a) {
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { s0 = dot4(w.lo[i], a.lo[i], s0); s1 = dot4(w.hi[i], a.hi[i], s1); }
    return a.ds.x * (w.s.dsc0 * (float)s0 - w.s.dm0 * (float)a.slo) +
           a.ds.z * (w.s.dsc1 * (float)s1 - w.s.dm1 * (float)a.shi);
}

template <> struct Unit<MI355_TYPE_Q4_K> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i qs, hdr; };
    typedef DecK45 Dec;
    typedef Act2x16 Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 3) * 144;
        Raw r;
        r.qs  = ld_nt16(blk + 16 + (u & 7) * 16);
        r.hdr = *(const v4i *)blk;
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        Dec d;
        d.lo = r.qs & 0x0f0f0f0f;
        d.hi = (r.qs >> 4) & 0x0f0f0f0f;
        d.s  = k4_scales(r.hdr, (u & 7) >> 1);
        return d;
    }
    static __device__ __forceinline__ Act load_act(const ActPtr a, int u) { return load_act_k45(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) { return dot_k45(w, a); }
};

// ---- Q5_K: 176 B = {half d, half dmin, scales[12], qh[32], qs[128]} (ggml-common.h:302-313);
// as q4_K plus bit (2g) / (2g+1) of qh[16*(p&1) + j] as the 5th bit (ggml-quants.c:1495-1496).
template <> struct Unit<MI355_TYPE_Q5_K> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i qs, qh, hdr; };
    typedef DecK45 Dec;
    typedef Act2x16 Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 3) * 176;
        Raw r;
        r.qs  = ld_nt16(blk + 48 + (u & 7) * 16);
        r.qh  = *(const v4i *)(blk + 16 + (u & 1) * 16);
        r.hdr = *(const v4i *)blk;
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        const int g = (u & 7) >> 1;
        Dec d;
        d.lo = (r.qs & 0x0f0f0f0f) | (((r.qh >> (2 * g)) & 0x01010101) << 4);
        d.hi = ((r.qs >> 4) & 0x0f0f0f0f) | (((r.qh >> (2 * g + 1)) & 0x01010101) << 4);
        d.s  = k4_scales(r.hdr, g);
        return d;
    }
    static __device__ __forceinline__ Act load_act(const ActPtr a, int u) { return load_act_k45(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) { return dot_k45(w, a); }
};

// ---- Q6_K: 210 B = {ql[128], qh[64], int8 scales[16], half d} (ggml-common.h:320-325), 2-byte aligned.
// unit (b, n, h): 16 B of qh (qh[32n + 16h ..]) with ql[64n + 16h ..] and ql[64n + 32 + 16h ..]
// -> 4 groups of 16 weights at elements 128n + 32i + 16h (i = 0..3), scale sc[8n + h + 2i]
// (dequantize_row_q6_K, ggml-quants.c:1695-1711).
template <> struct Unit<MI355_TYPE_Q6_K> {
    static constexpr int ELEMS = 64;
    struct Raw { v4i qla, qlb, qh; v2i sc; uint16_t d; };
    struct Dec { v4i q[4]; float dsc[4]; };
    struct Act { v4i q[4]; float d[4]; };
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 2) * 210;
        const int n = (u >> 1) & 1, h = u & 1;
        Raw r;
        r.qla = ld_nt16_a2(blk + 64 * n + 16 * h);
        r.qlb = ld_nt16_a2(blk + 64 * n + 32 + 16 * h);
        r.qh  = ld_nt16_a2(blk + 128 + 32 * n + 16 * h);
        r.sc  = ((const v2i_a2 *)(blk + 192 + 8 * n))->v;
        r.d   = ld_u16(blk + 208);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        const int h = u & 1;
        Dec d;
        d.q[0] = (r.qla & 0x0f0f0f0f) | ((r.qh & 0x03030303) << 4);
        d.q[1] = (r.qlb & 0x0f0f0f0f) | (((r.qh >> 2) & 0x03030303) << 4);
        d.q[2] = ((r.qla >> 4) & 0x0f0f0f0f) | (((r.qh >> 4) & 0x03030303) << 4);
        d.q[3] = ((r.qlb >> 4) & 0x0f0f0f0f) | (((r.qh >> 6) & 0x03030303) << 4);
        const float dd = h2f(r.d);
        // scales h, h+2 live in sc.x (bytes h, h+2); h+4, h+6 in sc.y
        const int s0 = (int)(int8_t)((uint32_t)r.sc.x >> (8 * h));
        const int s1 = (int)(int8_t)((uint32_t)r.sc.x >> (8 * h + 16));
        const int s2 = (int)(int8_t)((uint32_t)r.sc.y >> (8 * h));
        const int s3 = (int)(int8_t)((uint32_t)r.sc.y >> (8 * h + 16));
        d.dsc[0] = dd * (float)s0; d.dsc[1] = dd * (float)s1; d.dsc[2] = dd * (float)s2; d.dsc[3] = dd * (float)s3;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) d.q[i][j] = sub_bytes(d.q[i][j], 0x20202020);
        return d;
    }
    static __device__ __forceinline__ Act load_act(const ActPtr a, int u) {
        const int b = u >> 2, n = (u >> 1) & 1, h = u & 1;
        const int e0 = b * QK_K + n * 128 + h * 16;
        Act r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.q[i] = *(const v4i *)(a.q + e0 + 32 * i);
        const v4f d01 = *(const v4f *)(a.ds + (b * 8 + n * 4) * 2);
        const v4f d23 = *(const v4f *)(a.ds + (b * 8 + n * 4 + 2) * 2);
        r.d[0] = d01.x; r.d[1] = d01.z; r.d[2] = d23.x; r.d[3] = d23.z;
        return r;
    }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) s = dot4(w.q[i][j], a.q[i][j], s);
            acc += w.dsc[i] * a.d[i] * (float)s;
        }
        return acc;
    }
};

// ---- Q8_0: 34 B = {half d, int8 qs[32]} (ggml-common.h:210-213); unit = one block
template <> struct Unit<MI355_TYPE_Q8_0> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i q0, q1; uint16_t d; };
    struct Dec { v4i q0, q1; float d; };
    struct Act { v4i q0, q1; float d; };
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)u * 34;
        Raw r;
        r.q0 = ld_nt16_a2(blk + 2);
        r.q1 = ld_nt16_a2(blk + 18);
        r.d  = ld_u16(blk);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int) { Dec d; d.q0 = r.q0; d.q1 = r.q1; d.d = h2f(r.d); return d; }
    static __device__ __forceinline__ Act load_act(const ActPtr a, int u) {
        Act r;
        r.q0 = *(const v4i *)(a.q + u * 32);
        r.q1 = *(const v4i *)(a.q + u * 32 + 16);
        r.d  = a.ds[u * 2];
        return r;
    }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { s = dot4(w.q0[j], a.q0[j], s); s = dot4(w.q1[j], a.q1[j], s); }
        return w.d * a.d * (float)s;
    }
};

// ---- Q4_0: 18 B = {half d, u8 qs[16]} (ggml-common.h:168-171); unit = one block;
// low nibbles are elements 0..15, high nibbles 16..31, value = nibble - 8 (ggml-quants.c:259-265)
template <> struct Unit<MI355_TYPE_Q4_0> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i qs; uint16_t d; };
    struct Dec { v4i lo, hi; float d; };
    typedef Unit<MI355_TYPE_Q8_0>::Act Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)u * 18;
        Raw r;
        r.qs = ld_nt16_a2(blk + 2);
        r.d  = ld_u16(blk);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int) {
        Dec d;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d.lo[j] = sub_bytes(r.qs[j] & 0x0f0f0f0f, 0x08080808);
            d.hi[j] = sub_bytes((r.qs[j] >> 4) & 0x0f0f0f0f, 0x08080808);
        }
        d.d = h2f(r.d);
        return d;
    }
    static __device__ __forceinline__ Act load_act(const ActPtr a, int u) { return Unit<MI355_TYPE_Q8_0>::load_act(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { s = dot4(w.lo[j], a.q0[j], s); s = dot4(w.hi[j], a.q1[j], s); }
        return w.d * a.d * (float)s;
    }
};

// ------------------------------------------------------------------------------------------------
struct KArgs {
    const uint8_t * W; int64_t w_row_stride, w_ch_stride;
    const int8_t * act_q; const float * act_ds; int64_t act_ch_stride;
    float * dst; int64_t dst_col_stride, dst_ch_stride;
    int m, k, ch_ratio;
    const int32_t * ids; int ids_stride, ids_nused, act_ne1, n_expert;
};

template <int TYPE, int NCOLS, int RPW>
__global__ void __launch_bounds__(256)
mmvq_kernel(const KArgs a) {
    typedef Unit<TYPE> U;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ch   = blockIdx.y;

    int64_t ch_w, act_row;
    if (a.ids) {
        const int tok = ch / a.ids_nused, slot = ch % a.ids_nused;
        const int ex = a.ids[(int64_t)tok * a.ids_stride + slot];
        if (ex < 0 || ex >= a.n_expert) return; // out-of-range expert id: leave dst untouched
        ch_w    = ex;
        act_row = (int64_t)tok * a.act_ch_stride + slot % a.act_ne1;
    } else {
        ch_w    = ch / a.ch_ratio;
        act_row = (int64_t)ch * a.act_ch_stride;
    }

    const int row0 = (blockIdx.x * 4 + wave) * RPW;
    if (row0 >= a.m) return;

    const uint8_t * wbase = a.W + ch_w * a.w_ch_stride;
    const uint8_t * wrow[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const int row = row0 + r < a.m ? row0 + r : a.m - 1; // tail rows re-read the last row, never stored
        wrow[r] = wbase + (int64_t)row * a.w_row_stride;
    }
    ActPtr act[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; c++) {
        act[c].q  = a.act_q + (act_row + c) * a.k;
        act[c].ds = a.act_ds + (act_row + c) * (a.k / QK) * 2;
    }

    float acc[RPW][NCOLS];
#pragma unroll
    for (int r = 0; r < RPW; r++)
#pragma unroll
        for (int c = 0; c < NCOLS; c++) acc[r][c] = 0.0f;

    const int nunits = a.k / U::ELEMS;
#pragma unroll 2
    for (int u = lane; u < nunits; u += 64) {
        typename U::Raw raw[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) raw[r] = U::load(wrow[r], u);
        typename U::Act av[NCOLS];
#pragma unroll
        for (int c = 0; c < NCOLS; c++) av[c] = U::load_act(act[c], u);
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const typename U::Dec w = U::decode(raw[r], u);
#pragma unroll
            for (int c = 0; c < NCOLS; c++) acc[r][c] += U::dot(w, av[c]);
        }
    }

    float * dch = a.dst + (int64_t)ch * a.dst_ch_stride;
#pragma unroll
    for (int r = 0; r < RPW; r++)
#pragma unroll
        for (int c = 0; c < NCOLS; c++) {
            const float v = wave_sum(acc[r][c]);
            if (lane == 0 && row0 + r < a.m) dch[(int64_t)c * a.dst_col_stride + row0 + r] = v;
        }
}

template <int TYPE, int NCOLS, int RPW>
static int launch(const KArgs & a, int nchannels, hipStream_t stream) {
    const dim3 grid((unsigned)((a.m + 4 * RPW - 1) / (4 * RPW)), (unsigned)nchannels);
    mmvq_kernel<TYPE, NCOLS, RPW><<<grid, 256, 0, stream>>>(a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

template <int TYPE>
static int dispatch_cols(const KArgs & a, int ncols, int nchannels, hipStream_t s) {
    // rows per wave: enough waves to fill 256 CUs x 8 waves first, then deepen per-wave work
    const int64_t rows_total = (int64_t)a.m * nchannels;
    constexpr int R2 = TYPE == MI355_TYPE_Q6_K ? 1 : 2; // q6_K units hold 3x the registers
    switch (ncols) {
        case 1:
            if (rows_total >= 16384) return launch<TYPE, 1, 4>(a, nchannels, s);
            if (rows_total >= 4096)  return launch<TYPE, 1, 2>(a, nchannels, s);
            return launch<TYPE, 1, 1>(a, nchannels, s);
        case 2: return launch<TYPE, 2, R2>(a, nchannels, s);
        case 3: return launch<TYPE, 3, R2>(a, nchannels, s);
        case 4: return launch<TYPE, 4, R2>(a, nchannels, s);
        case 5: return launch<TYPE, 5, 1>(a, nchannels, s);
        case 6: return launch<TYPE, 6, 1>(a, nchannels, s);
        case 7: return launch<TYPE, 7, 1>(a, nchannels, s);
        case 8: return launch<TYPE, 8, 1>(a, nchannels, s);
    }
    return MI355_E_SHAPE;
}

} // namespace

extern "C" int mi355_mul_mat_vec_q(const mi355_mmvq_args * p, void * stream) {
    if (!p || !p->W || !p->act_q || !p->act_ds || !p->dst) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(p->type);
    if (bs == 0 || p->type == MI355_TYPE_F32 || p->type == MI355_TYPE_F16 || p->type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (p->m < 0 || p->k <= 0 || p->k % bs != 0 || p->ncols < 1 || p->ncols > 8 || p->nchannels < 1 ||
        p->nchannels > 65535 || p->ch_ratio < 1 || p->m > 0x7fffffff || p->k > 0x7fffffff) return MI355_E_SHAPE;
    if (p->ids && (p->ids_nused < 1 || p->act_ne1 < 1 || p->n_expert < 1)) return MI355_E_SHAPE;
    // alignment the loads rely on: activations 16 B; q4_K/q5_K blocks 16 B; the 2-byte-aligned
    // types (q4_0, q8_0, q6_K) only need their natural 2 B
    if (((uintptr_t)p->act_q & 15) || ((uintptr_t)p->act_ds & 15) || ((uintptr_t)p->dst & 3)) return MI355_E_ALIGN;
    const bool a16 = p->type == MI355_TYPE_Q4_K || p->type == MI355_TYPE_Q5_K;
    const uintptr_t wal = (uintptr_t)p->W | (uintptr_t)p->w_row_stride | (uintptr_t)p->w_ch_stride;
    if (wal & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (p->w_row_stride < mi355_row_size(p->type, p->k)) return MI355_E_SHAPE;
    if (p->m == 0) return MI355_OK;

    KArgs a;
    a.W = (const uint8_t *)p->W; a.w_row_stride = p->w_row_stride; a.w_ch_stride = p->w_ch_stride;
    a.act_q = p->act_q; a.act_ds = p->act_ds; a.act_ch_stride = p->act_ch_stride;
    a.dst = p->dst; a.dst_col_stride = p->dst_col_stride; a.dst_ch_stride = p->dst_ch_stride;
    a.m = (int)p->m; a.k = (int)p->k; a.ch_ratio = p->ch_ratio;
    a.ids = p->ids; a.ids_stride = p->ids_stride; a.ids_nused = p->ids_nused; a.act_ne1 = p->act_ne1; a.n_expert = p->n_expert;
    hipStream_t s = (hipStream_t)stream;
    switch (p->type) {
        case MI355_TYPE_Q4_0: return dispatch_cols<MI355_TYPE_Q4_0>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q8_0: return dispatch_cols<MI355_TYPE_Q8_0>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q4_K: return dispatch_cols<MI355_TYPE_Q4_K>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q5_K: return dispatch_cols<MI355_TYPE_Q5_K>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q6_K: return dispatch_cols<MI355_TYPE_Q6_K>(a, p->ncols, p->nchannels, s);
    }
    return MI355_E_TYPE;
}
