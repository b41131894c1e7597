// mmvq.hip -- decode mat-vec on GGUF block-quantized weights for gfx950.
//
// Replaces mul_mat_vec_q / ggml_cuda_mul_mat_vec_q (ggml/src/ggml-cuda/mmvq.cu:137-224,491-561) and
// the vec_dot_*_q8_1 family (ggml/src/ggml-cuda/vecdotq.cuh); not derived from them: the work
// split, load shapes and arithmetic form are chosen for CDNA4.
//
// Roofline: HBM.  Algorithmic bytes per launch = M*(K/bs)*B + 4K + 4M.
//
// Shape of the kernel (measured ablations: llm_amd/csrc/lab/q4k_lab.hip)
//   * a wavefront (64 lanes) owns one weight row at a time; a 256-thread workgroup = 4 waves.
//   * the q8_1 activations (internal SoA layout, see mi355_quantize_act) are staged once per
//     workgroup in LDS; in the loop they come back with ds_read_b128, so the vector-memory pipe
//     carries weights only (the TA/L1 path, not HBM, was the first bound of the naive version).
//   * the row is cut into "units"; lane l takes units l, l+64, ...  A q4_K unit is one 64-value
//     group = 32 B of nibbles (two 128-bit loads) + the block's 16 B header, so the 6-bit
//     scale/min decode and the f32 epilogue are paid once per 64 weights.  Weights go straight
//     to VGPRs (no LDS round trip), double-buffered in registers across loop iterations.
//   * nibbles / 6-bit fields are unpacked with a few bit ops per dword and fed to
//     v_dot4c_i32_i8; integer sums are exact, the per-32 scales are applied in f32.
//   * wavefront-level xor-shuffle reduction, one f32 store per (row, column).
//
// Arithmetic contract (tests/test_kernels_gpu.py): dst = sum over 32-blocks of
//     d8 * ( dW_j * sum(qW*q8) - mW_j * sum(q8) )        (q4_K / q5_K)
//     d8 * dW_j * sum((qW - off)*q8)                     (q4_0: off 8, q6_K: off 32, q8_0: off 0)
// i.e. exactly dequant(W) . dequant_q8_1(x) up to f32 summation order.
#include "common.h"
#include "rope.h"
#include <stdlib.h>
#include <string>
#include <string.h>

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
// unit = group g (0..3) of block b: qs[32g .. 32g+31]; low nibbles are sub-block 2g, high nibbles
// sub-block 2g+1, 32 values each (dequantize_row_q4_K, ggml-quants.c:1286-1293).
struct K4Scales { int sc0, sc1; float fm0, fm1, d, dmin; };

static __device__ __forceinline__ K4Scales k4_scales(const v4i hdr, int g) {
    // hdr.x = {d, dmin}; hdr.y/z/w = the 12 packed 6-bit scales/mins (get_scale_min_k4, ggml-quants.c:625-632)
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
    K4Scales s;
    s.sc0 = (int)(sc2 & 0xff); s.sc1 = (int)(sc2 >> 8);
    s.fm0 = (float)(m2 & 0xff); s.fm1 = (float)(m2 >> 8);
    s.d    = h2f((uint16_t)(hdr.x & 0xffff));
    s.dmin = h2f((uint16_t)((uint32_t)hdr.x >> 16));
    return s;
}

// activations of one 64-value group: 64 int8 + {d_lo, sum_lo, d_hi, sum_hi}
struct Act64 { v4i q[4]; v4f ds; };

template <class AP>
static __device__ __forceinline__ Act64 load_act64(const AP a, int u) {
    Act64 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.q[i] = *(const v4i *)(a.q + u * 64 + i * 16);
    r.ds = *(const v4f *)(a.ds + u * 4);
    return r;
}

struct DecK45 { v4i lo[2], hi[2]; K4Scales s; };

static __device__ __forceinline__ float dot_k45(const DecK45 & w, const Act64 & a) {
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s0 = dot4(w.lo[0][i], a.q[0][i], s0);
        s0 = dot4(w.lo[1][i], a.q[1][i], s0);
        s1 = dot4(w.hi[0][i], a.q[2][i], s1);
        s1 = dot4(w.hi[1][i], a.q[3][i], s1);
    }
    // integer sub-block scale folded in before the single int->float conversion (|s*sc| < 2^24)
    const float x = (float)(s0 * w.s.sc0) * a.ds.x + (float)(s1 * w.s.sc1) * a.ds.z;
    const float y = w.s.fm0 * (a.ds.x * a.ds.y) + w.s.fm1 * (a.ds.z * a.ds.w);
    return w.s.d * x - w.s.dmin * y;
}

template <> struct Unit<MI355_TYPE_Q4_K> {
    static constexpr int ELEMS = 64;
    struct Raw { v4i q0, q1, hdr; };
    typedef DecK45 Dec;
    typedef Act64 Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 2) * 144;
        Raw r;
        r.q0  = *(const v4i *)(blk + 16 + (u & 3) * 32);
        r.q1  = *(const v4i *)(blk + 32 + (u & 3) * 32);
        r.hdr = *(const v4i *)blk;
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        Dec d;
        d.lo[0] = r.q0 & 0x0f0f0f0f;        d.lo[1] = r.q1 & 0x0f0f0f0f;
        d.hi[0] = (r.q0 >> 4) & 0x0f0f0f0f; d.hi[1] = (r.q1 >> 4) & 0x0f0f0f0f;
        d.s = k4_scales(r.hdr, u & 3);
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return load_act64(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) { return dot_k45(w, a); }
};

// ---- Q5_K: 176 B = {half d, half dmin, scales[12], qh[32], qs[128]} (ggml-common.h:302-313);
// as q4_K plus bit (2g) / (2g+1) of qh[l] as the 5th bit of value l of the low / high sub-block
// (dequantize_row_q5_K, ggml-quants.c:1495-1496).
template <> struct Unit<MI355_TYPE_Q5_K> {
    static constexpr int ELEMS = 64;
    struct Raw { v4i q0, q1, h0, h1, hdr; };
    typedef DecK45 Dec;
    typedef Act64 Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 2) * 176;
        Raw r;
        r.q0  = *(const v4i *)(blk + 48 + (u & 3) * 32);
        r.q1  = *(const v4i *)(blk + 64 + (u & 3) * 32);
        r.h0  = *(const v4i *)(blk + 16);
        r.h1  = *(const v4i *)(blk + 32);
        r.hdr = *(const v4i *)blk;
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        const int g = u & 3;
        Dec d;
        d.lo[0] = (r.q0 & 0x0f0f0f0f) | (((r.h0 >> (2 * g)) & 0x01010101) << 4);
        d.lo[1] = (r.q1 & 0x0f0f0f0f) | (((r.h1 >> (2 * g)) & 0x01010101) << 4);
        d.hi[0] = ((r.q0 >> 4) & 0x0f0f0f0f) | (((r.h0 >> (2 * g + 1)) & 0x01010101) << 4);
        d.hi[1] = ((r.q1 >> 4) & 0x0f0f0f0f) | (((r.h1 >> (2 * g + 1)) & 0x01010101) << 4);
        d.s = k4_scales(r.hdr, g);
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return load_act64(a, u); }
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
        r.qla = ld16_a2(blk + 64 * n + 16 * h);
        r.qlb = ld16_a2(blk + 64 * n + 32 + 16 * h);
        r.qh  = ld16_a2(blk + 128 + 32 * n + 16 * h);
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
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) {
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
        r.q0 = ld16_a2(blk + 2);
        r.q1 = ld16_a2(blk + 18);
        r.d  = ld_u16(blk);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int) { Dec d; d.q0 = r.q0; d.q1 = r.q1; d.d = h2f(r.d); return d; }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) {
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
        r.qs = ld16_a2(blk + 2);
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
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return Unit<MI355_TYPE_Q8_0>::load_act(a, u); }
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

// LDS / global flavours of the activation pointers (address spaces stay static so the LDS build
// reads with ds_read_b128, not flat loads)
struct ActLds { const __attribute__((address_space(3))) int8_t * q; const __attribute__((address_space(3))) float * ds; };

// One wave, one weight row: the body shared by the single-matrix and the grouped kernel.
//   wmat: first row of the matrix; row: this wave's row (rows >= m compute on the last row and store nothing)
template <int TYPE, int NCOLS, bool LDSACT>
static __device__ __forceinline__ void mmvq_row(const uint8_t * wmat, int64_t w_row_stride, int m, int row, const int8_t * gq,
                                                const float * gds, int k, float * dch, int64_t dst_col_stride, uint8_t * lds,
                                                const float * res = nullptr, float * sum = nullptr) {
    typedef Unit<TYPE> U;
    const int lane = threadIdx.x & 63;
    const int rowc = row < m ? row : m - 1;           // tail waves re-read the last row, never store
    const uint8_t * wrow = wmat + (int64_t)rowc * w_row_stride;

    // Uniform trip count (lanes past the end re-read the last unit and contribute 0) so the loads
    // are unconditional; the first weight loads are issued before the activation staging so their
    // HBM latency overlaps it.
    const int nunits = k / U::ELEMS;
    const int niter  = (nunits + 63) >> 6;
    typename U::Raw cur = U::load(wrow, lane < nunits ? lane : nunits - 1);
    // residual epilogue (sum = result + res, the ADD node that follows wo / ffn_down): requested now, used at the end
    float rv[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; c++) rv[c] = (res && lane == 0 && row < m) ? res[(int64_t)c * dst_col_stride + row] : 0.0f;

    if (LDSACT) {
        // activations -> LDS: [NCOLS][k] int8, then [NCOLS][k/32][2] f32 (columns are contiguous rows of act)
        const int nq = NCOLS * k / 16, nds = NCOLS * (k / QK); // 16-byte chunks of q, {d, sum} pairs
        for (int i = threadIdx.x; i < nq; i += 256) ((v4i *)lds)[i] = ((const v4i *)gq)[i];
        v2f * lds_ds = (v2f *)(lds + (size_t)NCOLS * k);
        for (int i = threadIdx.x; i < nds; i += 256) lds_ds[i] = ((const v2f *)gds)[i];
        __syncthreads();
    }

    float acc[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; c++) acc[c] = 0.0f;

    // Register double buffer: the weights of iteration it+1 are requested before iteration it is
    // consumed.  sched_barrier pins "all loads first": left alone the scheduler sinks each load
    // next to its use and the loop degenerates into load / s_waitcnt vmcnt(0) pairs.
    for (int it = 0; it < niter; it++) {
        const int  u_raw = it * 64 + lane;
        const bool valid = u_raw < nunits;
        const int  u     = valid ? u_raw : nunits - 1;
        typename U::Raw nxt;
        if (it + 1 < niter) nxt = U::load(wrow, u_raw + 64 < nunits ? u_raw + 64 : nunits - 1);
        typename U::Act av[NCOLS];
        if (LDSACT) {
            const auto lq  = (const __attribute__((address_space(3))) int8_t *)lds;
            const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)NCOLS * k);
#pragma unroll
            for (int c = 0; c < NCOLS; c++) av[c] = U::load_act(ActLds{lq + c * k, lds_ + c * (k / QK) * 2}, u);
        } else {
#pragma unroll
            for (int c = 0; c < NCOLS; c++) av[c] = U::load_act(ActPtr{gq + (int64_t)c * k, gds + (int64_t)c * (k / QK) * 2}, u);
        }
        __builtin_amdgcn_sched_barrier(0);
        const typename U::Dec w = U::decode(cur, u);
#pragma unroll
        for (int c = 0; c < NCOLS; c++) {
            const float v = U::dot(w, av[c]);
            acc[c] += valid ? v : 0.0f;
        }
        if (it + 1 < niter) cur = nxt;
    }

#pragma unroll
    for (int c = 0; c < NCOLS; c++) {
        const float v = wave_sum(acc[c]);
        if (lane == 0 && row < m) {
            const int64_t o = (int64_t)c * dst_col_stride + row;
            if (sum != dch) dch[o] = v;             // sum == dch: the ADD is in place on the mat-vec result
            if (res) sum[o] = v + rv[c];
        }
    }
}

template <int TYPE, int NCOLS, bool LDSACT>
__global__ void __launch_bounds__(256)
mmvq_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6;
    const int ch   = blockIdx.y;

    int64_t ch_w, act_row;
    if (a.ids) {
        const int tok = ch / a.ids_nused, slot = ch % a.ids_nused;
        const int ex = a.ids[(int64_t)tok * a.ids_stride + slot];
        if (ex < 0 || ex >= a.n_expert) return; // out-of-range expert id (uniform per workgroup): dst untouched
        ch_w    = ex;
        act_row = (int64_t)tok * a.act_ch_stride + slot % a.act_ne1;
    } else {
        ch_w    = ch / a.ch_ratio;
        act_row = (int64_t)ch * a.act_ch_stride;
    }
    mmvq_row<TYPE, NCOLS, LDSACT>(a.W + ch_w * a.w_ch_stride, a.w_row_stride, a.m, blockIdx.x * 4 + wave,
                                  a.act_q + act_row * a.k, a.act_ds + act_row * (a.k / QK) * 2, a.k,
                                  a.dst + (int64_t)ch * a.dst_ch_stride, a.dst_col_stride, lds);
}

// Grouped launch: up to MI355_MMVQ_MAX_GROUP matrices of the same type and K that read the SAME activations
// (wq/wk/wv, ffn gate/up) share one grid, so the per-launch floor (dispatch + cold-start latency + tail) is paid
// once.  Workgroups [blk_end[i-1], blk_end[i]) belong to matrix i.
struct KGroup {
    const uint8_t * W[MI355_MMVQ_MAX_GROUP]; int64_t w_row_stride[MI355_MMVQ_MAX_GROUP];
    float * dst[MI355_MMVQ_MAX_GROUP]; int64_t dst_col_stride[MI355_MMVQ_MAX_GROUP];
    int m[MI355_MMVQ_MAX_GROUP], blk_end[MI355_MMVQ_MAX_GROUP];
    const int8_t * act_q; const float * act_ds;
    int k, n;
    const float * res[MI355_MMVQ_MAX_GROUP]; float * sum[MI355_MMVQ_MAX_GROUP];   // optional residual epilogue per matrix
};

template <int TYPE, int NCOLS, bool LDSACT>
__global__ void __launch_bounds__(256)
mmvq_group_kernel(const KGroup a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6;
    int mi = 0, blk0 = 0;
#pragma unroll
    for (int i = 0; i + 1 < MI355_MMVQ_MAX_GROUP; i++) {
        if (i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i]) { mi = i + 1; blk0 = a.blk_end[i]; }
    }
    mmvq_row<TYPE, NCOLS, LDSACT>(a.W[mi], a.w_row_stride[mi], a.m[mi], ((int)blockIdx.x - blk0) * 4 + wave, a.act_q, a.act_ds,
                                  a.k, a.dst[mi], a.dst_col_stride[mi], lds, a.res[mi], a.sum[mi]);
}

// ------------------------------------------------------------------------------------------------
// Fused attention-input stage of a decode step: Q, K and V projections of one token, the rotary embedding of Q
// and K, and the store of K and V into the KV cache -- ONE launch for what the graph spells as 3 MUL_MAT +
// 2 ROPE + 2 CPY nodes (7 launches).  A wave owns the two weight rows of one rotated pair (both rows read the
// same activations, so the LDS reads are shared), lane 0 rotates the pair and stores it: Q as f32, K and V as
// f16 (or f32) straight into the cache rows.  Row sums, rotation and conversion use the same code as the
// separate kernels, so the results are bit-identical to the unfused graph.
// ------------------------------------------------------------------------------------------------
// RMS_NORM -> MUL(weight) -> q8_1 of one token's row, written straight into the activation area of LDS ([k] int8, then
// [k/32] {d, sum}): the prologue of the fused decode kernels.  Every workgroup normalises the row for itself (k floats
// out of L2, hidden behind the first weight loads) with the arithmetic of rms_norm_kernel -- one 16-value chunk per
// thread, wave butterflies, wave sums added in wave order -- so the values are bit-identical to the separate norm launch
// (NT = 256 covers rows up to 4096 like its 256-thread form, NT = 512 rows up to 8192 like its 512-thread form; waves
// without a chunk add 0).  `store`: this workgroup also writes the f32 result (the graph's MUL node).
template <int NT>
static __device__ __forceinline__ void norm_to_lds(const float * nx, const float * nw, float * nout, float eps, int k, uint8_t * lds, bool store) {
    __shared__ float sh[NT / 64];
    const int nc = k >> 4, c = threadIdx.x;
    v4f v[4], wv[4];
    float s = 0.0f;
    if (c < nc) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v[q]  = ((const v4f *)nx)[4 * c + q];
            wv[q] = ((const v4f *)nw)[4 * c + q];
            s += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
        }
    }
    s = block_sum_n<NT / 64>(s, sh);
    const float scale = 1.0f / sqrtf(s / (float)k + eps);
    if (c < nc) {
        float o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v4f t = v[q] * scale;
            t = t * wv[q];
            if (store) ((v4f *)nout)[4 * c + q] = t;
            o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
        }
        quant16_soa(o, (int8_t *)lds + 16 * c, (float *)(lds + (size_t)k) + (c >> 1) * 2, (c & 1) == 0);
    }
    __syncthreads();
}

template <int TYPE, bool LDSACT>
static __device__ __forceinline__ void mmvq_row2(const uint8_t * w0, const uint8_t * w1, const int8_t * gq, const float * gds,
                                                 int k, uint8_t * lds, float & s0, float & s1, const float * nx = nullptr,
                                                 const float * nw = nullptr, float * nout = nullptr, float eps = 0.0f) {
    typedef Unit<TYPE> U;
    const int lane = threadIdx.x & 63;
    const int nunits = k / U::ELEMS;
    const int niter  = (nunits + 63) >> 6;
    const int u0 = lane < nunits ? lane : nunits - 1;
    typename U::Raw c0 = U::load(w0, u0), c1 = U::load(w1, u0);
    if (LDSACT && nx) {
        norm_to_lds<256>(nx, nw, nout, eps, k, lds, blockIdx.x == 0);
    } else if (LDSACT) {
        const int nq = k / 16, nds = k / QK;
        for (int i = threadIdx.x; i < nq; i += 256) ((v4i *)lds)[i] = ((const v4i *)gq)[i];
        v2f * lds_ds = (v2f *)(lds + (size_t)k);
        for (int i = threadIdx.x; i < nds; i += 256) lds_ds[i] = ((const v2f *)gds)[i];
        __syncthreads();
    }
    float acc0 = 0.0f, acc1 = 0.0f;
    for (int it = 0; it < niter; it++) {
        const int  u_raw = it * 64 + lane;
        const bool valid = u_raw < nunits;
        const int  u     = valid ? u_raw : nunits - 1;
        typename U::Raw n0, n1;
        if (it + 1 < niter) {
            const int un = u_raw + 64 < nunits ? u_raw + 64 : nunits - 1;
            n0 = U::load(w0, un); n1 = U::load(w1, un);
        }
        typename U::Act av;
        if (LDSACT) {
            const auto lq  = (const __attribute__((address_space(3))) int8_t *)lds;
            const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)k);
            av = U::load_act(ActLds{lq, lds_}, u);
        } else {
            av = U::load_act(ActPtr{gq, gds}, u);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float v0 = U::dot(U::decode(c0, u), av);
        const float v1 = U::dot(U::decode(c1, u), av);
        acc0 += valid ? v0 : 0.0f;
        acc1 += valid ? v1 : 0.0f;
        if (it + 1 < niter) { c0 = n0; c1 = n1; }
    }
    s0 = wave_sum(acc0);
    s1 = wave_sum(acc1);
}

struct KQkv {
    const uint8_t * W[3]; int64_t w_row_stride[3];
    int blk_end[3];                      // workgroups [blk_end[i-1], blk_end[i]) belong to matrix i (0 Q, 1 K, 2 V)
    const int8_t * act_q; const float * act_ds; int k;
    float * q_out; void * k_out; void * v_out; int k_f16, v_f16;
    const int32_t * pos; const float * ff; RopeP rp;
    const float * nx, * nw; float * nout; float eps;    // optional rms-norm prologue (see norm_to_lds)
};

template <int TQ, int TK, int TV>
__global__ void __launch_bounds__(256)
mmvq_qkv_rope_kernel(const KQkv a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int mat = 0, blk0 = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) if ((int)blockIdx.x >= a.blk_end[i]) { mat = i + 1; blk0 = a.blk_end[i]; }
    const int pair = ((int)blockIdx.x - blk0) * 4 + wave;
    int r0, r1, ic = 0;
    bool rot = false;
    if (mat < 2) {
        const int half = a.rp.ne0 / 2, h = pair / half;
        ic = pair % half;
        if (2 * ic >= a.rp.n_dims) { r0 = h * a.rp.ne0 + 2 * ic; r1 = r0 + 1; }                      // passed through
        else if (a.rp.mode & 2)    { r0 = h * a.rp.ne0 + ic; r1 = r0 + a.rp.n_dims / 2; rot = true; } // NEOX
        else                       { r0 = h * a.rp.ne0 + 2 * ic; r1 = r0 + 1; rot = true; }           // NORMAL
    } else {
        r0 = 2 * pair; r1 = r0 + 1;
    }
    float c = 1.0f, sn = 0.0f;
    if (rot) rope_cos_sin(a.rp, a.pos[0], a.ff, ic, c, sn);
    const uint8_t * w = a.W[mat];
    const int64_t rs = a.w_row_stride[mat];
    float s0, s1;
    if (mat == 0)      mmvq_row2<TQ, true>(w + r0 * rs, w + r1 * rs, a.act_q, a.act_ds, a.k, lds, s0, s1, a.nx, a.nw, a.nout, a.eps);
    else if (mat == 1) mmvq_row2<TK, true>(w + r0 * rs, w + r1 * rs, a.act_q, a.act_ds, a.k, lds, s0, s1, a.nx, a.nw, a.nout, a.eps);
    else               mmvq_row2<TV, true>(w + r0 * rs, w + r1 * rs, a.act_q, a.act_ds, a.k, lds, s0, s1, a.nx, a.nw, a.nout, a.eps);
    if (lane == 0) {
        float y0 = s0, y1 = s1;
        if (rot) { y0 = s0 * c - s1 * sn; y1 = s0 * sn + s1 * c; }
        if (mat == 0) { a.q_out[r0] = y0; a.q_out[r1] = y1; }
        else {
            void * o = mat == 1 ? a.k_out : a.v_out;
            if (mat == 1 ? a.k_f16 : a.v_f16) { ((__half *)o)[r0] = __float2half_rn(y0); ((__half *)o)[r1] = __float2half_rn(y1); }
            else                              { ((float *)o)[r0] = y0; ((float *)o)[r1] = y1; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused feed-forward input of a decode step: gate and up projections of one token, the gated activation and the
// q8_1 form of the result for ffn_down -- ONE launch for what the graph spells as 2 MUL_MAT + GLU (2 launches as
// grouped mat-vec + glu_quant).  512 threads = 8 waves; a workgroup owns 32 consecutive outputs (= one q8_1 block, so
// the quantization needs no cross-workgroup step), a wave 4 of them, one after the other: for each it streams the
// gate row and the up row (same activations, shared LDS reads) with the next pair's loads already in flight.
// Row sums, activation and quantization use the same code as the separate kernels: bit-identical results.
// ------------------------------------------------------------------------------------------------
struct KGlu {
    const uint8_t * wg, * wu; int64_t rsg, rsu;
    const int8_t * act_q; const float * act_ds; int k, m, op;
    float * gate_out, * up_out, * dst; int8_t * out_q; float * out_ds;
    const float * nx, * nw; float * nout; float eps;    // optional rms-norm prologue: activations = q8_1((nx * scale) * nw)
    // optional expert routing (MUL_MAT_ID pair of a decode token): blockIdx.y = slot, matrices of expert ids[slot], all slots read
    // the same activations, outputs of slot s at [s * m]
    const int32_t * ids; int n_expert, n_slots; int64_t wg_ex_stride, wu_ex_stride;
};

static __device__ __forceinline__ float glu_act(int op, float x) {
    switch (op) {
        case 0: return x > 0.0f ? x : 0.0f;                                  // REGLU
        case 1: { const float c = 0.044715f, s = 0.79788456080286535587989211986876f; // GEGLU (tanh gelu, ggml_gelu_f32)
                  return 0.5f * x * (1.0f + tanhf(s * x * (1.0f + c * x * x))); }
        default: return x / (1.0f + expf(-x));                               // SWIGLU (ggml_silu_f32, vec.h:637-639)
    }
}

constexpr int GLU_THREADS = 512;      // 8 waves x GLU_PAIRS (4 or 8) outputs = 32 or 64 outputs per workgroup

template <int TYPE, int GLU_PAIRS>
__global__ void __launch_bounds__(GLU_THREADS, (TYPE == MI355_TYPE_Q5_K || TYPE == MI355_TYPE_Q6_K) ? 2 : 4)
mmvq_glu_kernel(const KGlu a) {
    typedef Unit<TYPE> U;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = a.k;
    const int row0 = (blockIdx.x * 8 + wave) * GLU_PAIRS;                     // first of this wave's outputs
    int64_t ex = 0;
    if (a.ids) {
        ex = a.ids[blockIdx.y];
        if (ex < 0 || ex >= a.n_expert) return;      // out-of-range expert id (uniform per workgroup): outputs untouched
    }
    const int64_t so = (int64_t)blockIdx.y * a.m;     // this slot's outputs
    const uint8_t * wg = a.wg + ex * a.wg_ex_stride + (int64_t)row0 * a.rsg, * wu = a.wu + ex * a.wu_ex_stride + (int64_t)row0 * a.rsu;
    const int nunits = k / U::ELEMS;
    const int niter  = (nunits + 63) >> 6;
    const int u0 = lane < nunits ? lane : nunits - 1;
    typename U::Raw c0 = U::load(wg, u0), c1 = U::load(wu, u0);
    if (a.nx) {
        norm_to_lds<GLU_THREADS>(a.nx, a.nw, a.nout, a.eps, k, lds, blockIdx.x == 0);
    } else {
        const int nq = k / 16, nds = k / QK;
        for (int i = threadIdx.x; i < nq; i += GLU_THREADS) ((v4i *)lds)[i] = ((const v4i *)a.act_q)[i];
        v2f * lds_ds = (v2f *)(lds + (size_t)k);
        for (int i = threadIdx.x; i < nds; i += GLU_THREADS) lds_ds[i] = ((const v2f *)a.act_ds)[i];
        __syncthreads();
    }
    const auto lq   = (const __attribute__((address_space(3))) int8_t *)lds;
    const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)k);
    float sg[GLU_PAIRS], su[GLU_PAIRS];
#pragma unroll
    for (int p = 0; p < GLU_PAIRS; p++) {
        float acc0 = 0.0f, acc1 = 0.0f;
        for (int it = 0; it < niter; it++) {
            const int  u_raw = it * 64 + lane;
            const bool valid = u_raw < nunits;
            const int  u     = valid ? u_raw : nunits - 1;
            typename U::Raw n0, n1;
            const bool more_here = it + 1 < niter;
            if (more_here || p + 1 < GLU_PAIRS) {
                const int unr = more_here ? u_raw + 64 : lane;
                const int un  = unr < nunits ? unr : nunits - 1;
                const int64_t adv = more_here ? 0 : 1;                       // next unit of this pair, or first of the next
                n0 = U::load(wg + (p + adv) * a.rsg, un); n1 = U::load(wu + (p + adv) * a.rsu, un);
            }
            const typename U::Act av = U::load_act(ActLds{lq, lds_}, u);
            __builtin_amdgcn_sched_barrier(0);
            const float v0 = U::dot(U::decode(c0, u), av);
            const float v1 = U::dot(U::decode(c1, u), av);
            acc0 += valid ? v0 : 0.0f;
            acc1 += valid ? v1 : 0.0f;
            if (more_here || p + 1 < GLU_PAIRS) { c0 = n0; c1 = n1; }
        }
        sg[p] = wave_sum(acc0);
        su[p] = wave_sum(acc1);
    }
    // lanes 0..3 finish one output each; the 32 gated values meet in LDS and 8 lanes quantize the block
    float * hs = (float *)(lds + (((size_t)k + (size_t)k / QK * 8 + 15) & ~(size_t)15));
    if (lane < GLU_PAIRS) {
        float g = sg[0], u = su[0];
#pragma unroll
        for (int p = 1; p < GLU_PAIRS; p++) if (lane == p) { g = sg[p]; u = su[p]; }
        if (a.gate_out) a.gate_out[so + row0 + lane] = g;
        if (a.up_out)   a.up_out[so + row0 + lane]   = u;
        hs[wave * GLU_PAIRS + lane] = glu_act(a.op, g) * u;
    }
    __syncthreads();
    if (threadIdx.x < 2 * GLU_PAIRS) {
        const v4f h = ((const v4f *)hs)[threadIdx.x];
        const int64_t o0 = so + (int64_t)blockIdx.x * (8 * GLU_PAIRS) + 4 * threadIdx.x;
        *(v4f *)(a.dst + o0) = h;
        if (a.out_q) {
            const float o[4] = {h.x, h.y, h.z, h.w};
            quant4_soa(o, a.out_q + o0, a.out_ds + (o0 >> 5) * 2, (threadIdx.x & 7) == 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Expert feed-forward output of one decode token in one launch (build_moe_ffn, src/llama-graph.cpp:760-795):
//     y = MUL_MAT_ID(ffn_down_exps, h, ids) [m, n_slots]   y = MUL(y, weights [1, n_slots])   out = ADD(ADD(y_0, y_1), ...)
//     [l_out = ADD(out, residual)]
// A wave owns one output row and walks the slots: row `r` of expert ids[s] against slot s's activations (all slots'
// activations staged in LDS), weighted, summed in slot order, plus the residual.  Every graph tensor is written.
// ------------------------------------------------------------------------------------------------
constexpr int MOE_DOWN_MAX_SLOTS = 8;
struct KMoeDown {
    const uint8_t * W; int64_t w_row_stride, w_ex_stride;
    const int8_t * act_q; const float * act_ds;         // [n_slots][k], [n_slots][k/32][2]
    const int32_t * ids; const float * wts;             // [n_slots] each
    float * y_out, * yw_out, * sum_out;                 // [n_slots][m] unweighted (optional), [n_slots][m] weighted, [m]
    const float * res; float * res_out;                 // optional residual and its sum
    int m, k, n_slots, n_expert;
};

template <int TYPE, bool LDSACT>
__global__ void __launch_bounds__(256)
mmvq_moe_down_kernel(const KMoeDown a) {
    typedef Unit<TYPE> U;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = a.k, row = blockIdx.x * 4 + wave;
    const int rowc = row < a.m ? row : a.m - 1;
    const int nunits = k / U::ELEMS, niter = (nunits + 63) >> 6;
    const size_t slot_q = (size_t)k, slot_ds = (size_t)k / QK * 8;
    int ex[MOE_DOWN_MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < MOE_DOWN_MAX_SLOTS; s++) {
        const int e = s < a.n_slots ? a.ids[s] : 0;
        ex[s] = (e < 0 || e >= a.n_expert) ? -1 : e;     // out-of-range id: the slot contributes nothing and is not stored
    }
    const uint8_t * wrow0 = a.W + (int64_t)(ex[0] < 0 ? 0 : ex[0]) * a.w_ex_stride + (int64_t)rowc * a.w_row_stride;
    typename U::Raw cur = U::load(wrow0, lane < nunits ? lane : nunits - 1);
    const float rv = (a.res && lane == 0 && row < a.m) ? a.res[row] : 0.0f;
    if (LDSACT) {
        const int nq = a.n_slots * k / 16, nds = a.n_slots * (k / QK);
        for (int i = threadIdx.x; i < nq; i += 256) ((v4i *)lds)[i] = ((const v4i *)a.act_q)[i];
        v2f * lds_ds = (v2f *)(lds + (size_t)a.n_slots * slot_q);
        for (int i = threadIdx.x; i < nds; i += 256) lds_ds[i] = ((const v2f *)a.act_ds)[i];
        __syncthreads();
    }
    float total = 0.0f;
    for (int s = 0; s < a.n_slots; s++) {
        int e = 0, en = 0;
#pragma unroll
        for (int t = 0; t < MOE_DOWN_MAX_SLOTS; t++) { if (t == s) e = ex[t]; if (t == s + 1) en = ex[t]; }
        const uint8_t * wrow  = a.W + (int64_t)(e < 0 ? 0 : e) * a.w_ex_stride + (int64_t)rowc * a.w_row_stride;
        const uint8_t * wnext = a.W + (int64_t)(en < 0 ? 0 : en) * a.w_ex_stride + (int64_t)rowc * a.w_row_stride;
        float acc = 0.0f;
        for (int it = 0; it < niter; it++) {
            const int  u_raw = it * 64 + lane;
            const bool valid = u_raw < nunits;
            const int  u     = valid ? u_raw : nunits - 1;
            typename U::Raw nxt;
            const bool more_here = it + 1 < niter;
            if (more_here || s + 1 < a.n_slots) {
                const int unr = more_here ? u_raw + 64 : lane;
                nxt = U::load(more_here ? wrow : wnext, unr < nunits ? unr : nunits - 1);
            }
            typename U::Act av;
            if (LDSACT) {
                const auto lq  = (const __attribute__((address_space(3))) int8_t *)lds + (size_t)s * slot_q;
                const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)a.n_slots * slot_q + (size_t)s * slot_ds);
                av = U::load_act(ActLds{lq, lds_}, u);
            } else {
                av = U::load_act(ActPtr{a.act_q + (int64_t)s * k, a.act_ds + (int64_t)s * (k / QK) * 2}, u);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float v = U::dot(U::decode(cur, u), av);
            acc += valid ? v : 0.0f;
            if (more_here || s + 1 < a.n_slots) cur = nxt;
        }
        const float y = wave_sum(acc);
        if (e >= 0) {
            const float yw = y * a.wts[s];
            if (lane == 0 && row < a.m) {
                if (a.y_out) a.y_out[(int64_t)s * a.m + row] = y;
                a.yw_out[(int64_t)s * a.m + row] = yw;
            }
            total = s == 0 ? yw : total + yw;
        }
    }
    if (lane == 0 && row < a.m) {
        if (a.sum_out != a.res_out) a.sum_out[row] = total;      // in place: only the residual sum survives
        if (a.res) a.res_out[row] = total + rv;
    }
}


static const char * type_name(int t) {
    switch (t) {
        case MI355_TYPE_Q4_0: return "q4_0"; case MI355_TYPE_Q8_0: return "q8_0"; case MI355_TYPE_Q4_K: return "q4_K";
        case MI355_TYPE_Q5_K: return "q5_K"; case MI355_TYPE_Q6_K: return "q6_K";
    }
    return "?";
}

static constexpr size_t LDS_ACT_LIMIT = 64 * 1024; // beyond this, activations are read through L1 instead

template <int TYPE, int NCOLS>
static int launch(const KArgs & a, int nchannels, hipStream_t stream) {
    const dim3 grid((unsigned)((a.m + 3) / 4), (unsigned)nchannels);
    const size_t need = (size_t)NCOLS * ((size_t)a.k + (size_t)a.k / QK * 8);
    static const std::string tag = std::string("mmvq_") + type_name(TYPE) + "_n" + std::to_string(NCOLS);
    if (need <= LDS_ACT_LIMIT) MI355_LAUNCH(tag.c_str(), (mmvq_kernel<TYPE, NCOLS, true>), grid, dim3(256), need, stream, a);
    else                       MI355_LAUNCH(tag.c_str(), (mmvq_kernel<TYPE, NCOLS, false>), grid, dim3(256), 0, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

template <int TYPE, int NCOLS>
static int launch_group(const KGroup & a, hipStream_t stream) {
    const dim3 grid((unsigned)a.blk_end[a.n - 1]);
    const size_t need = (size_t)NCOLS * ((size_t)a.k + (size_t)a.k / QK * 8);
    static const std::string tag = std::string("mmvq_") + type_name(TYPE) + "_n" + std::to_string(NCOLS);
    if (need <= LDS_ACT_LIMIT) MI355_LAUNCH(tag.c_str(), (mmvq_group_kernel<TYPE, NCOLS, true>), grid, dim3(256), need, stream, a);
    else                       MI355_LAUNCH(tag.c_str(), (mmvq_group_kernel<TYPE, NCOLS, false>), grid, dim3(256), 0, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

template <int TYPE>
static int dispatch_group_cols(const KGroup & a, int ncols, hipStream_t s) {
    switch (ncols) {
        case 1: return launch_group<TYPE, 1>(a, s);
        case 2: return launch_group<TYPE, 2>(a, s);
        case 3: return launch_group<TYPE, 3>(a, s);
        case 4: return launch_group<TYPE, 4>(a, s);
        case 5: return launch_group<TYPE, 5>(a, s);
        case 6: return launch_group<TYPE, 6>(a, s);
        case 7: return launch_group<TYPE, 7>(a, s);
        case 8: return launch_group<TYPE, 8>(a, s);
    }
    return MI355_E_SHAPE;
}

template <int TYPE>
static int dispatch_cols(const KArgs & a, int ncols, int nchannels, hipStream_t s) {
    switch (ncols) {
        case 1: return launch<TYPE, 1>(a, nchannels, s);
        case 2: return launch<TYPE, 2>(a, nchannels, s);
        case 3: return launch<TYPE, 3>(a, nchannels, s);
        case 4: return launch<TYPE, 4>(a, nchannels, s);
        case 5: return launch<TYPE, 5>(a, nchannels, s);
        case 6: return launch<TYPE, 6>(a, nchannels, s);
        case 7: return launch<TYPE, 7>(a, nchannels, s);
        case 8: return launch<TYPE, 8>(a, nchannels, s);
    }
    return MI355_E_SHAPE;
}

} // namespace

static int validate(const mi355_mmvq_args * p) {
    if (!p || !p->W || !p->act_q || !p->act_ds || !p->dst) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(p->type);
    if (bs == 0 || p->type == MI355_TYPE_F32 || p->type == MI355_TYPE_F16 || p->type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (p->m < 0 || p->k <= 0 || p->k % bs != 0 || p->ncols < 1 || p->ncols > 8 || p->nchannels < 1 ||
        p->nchannels > 65535 || p->ch_ratio < 1 || p->m > 0x7fffffff || p->k > 0x7fffffff) return MI355_E_SHAPE;
    if (p->ids && (p->ids_nused < 1 || p->act_ne1 < 1 || p->n_expert < 1)) return MI355_E_SHAPE;
    // alignment the loads rely on: activations 16 B; q4_K/q5_K blocks 16 B; the 2-byte-aligned
    // types (q4_0, q8_0, q6_K) only need their natural 2 B
    if (((uintptr_t)p->act_q & 15) || ((uintptr_t)p->act_ds & 7) || ((uintptr_t)p->dst & 3)) return MI355_E_ALIGN;
    if (p->k % QK_K == 0 && ((uintptr_t)p->act_ds & 15)) return MI355_E_ALIGN;
    const bool a16 = p->type == MI355_TYPE_Q4_K || p->type == MI355_TYPE_Q5_K;
    const uintptr_t wal = (uintptr_t)p->W | (uintptr_t)p->w_row_stride | (uintptr_t)p->w_ch_stride;
    if (wal & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (p->w_row_stride < mi355_row_size(p->type, p->k)) return MI355_E_SHAPE;
    return MI355_OK;
}

extern "C" int mi355_mul_mat_vec_q(const mi355_mmvq_args * p, void * stream) {
    const int rc = validate(p);
    if (rc != MI355_OK) return rc;
    if (p->m == 0) return MI355_OK;
    if (p->nchannels == 1 && !p->ids) return mi355_mul_mat_vec_q_group(p, 1, stream);   // one kernel for the plain case
    if (p->residual || p->sum) return MI355_E_UNSUPPORTED;     // the residual epilogue exists for the plain case only

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

extern "C" int mi355_mul_mat_vec_q_group(const mi355_mmvq_args * p, int n_mats, void * stream) {
    if (!p) return MI355_E_NULL;
    if (n_mats < 1 || n_mats > MI355_MMVQ_MAX_GROUP) return MI355_E_SHAPE;
    KGroup a;
    memset(&a, 0, sizeof(a));
    int64_t blocks = 0;
    int n = 0;
    for (int i = 0; i < n_mats; i++) {
        const int rc = validate(&p[i]);
        if (rc != MI355_OK) return rc;
        if (p[i].nchannels != 1 || p[i].ids) return MI355_E_UNSUPPORTED;
        if (p[i].type != p[0].type || p[i].k != p[0].k || p[i].ncols != p[0].ncols || p[i].act_q != p[0].act_q ||
            p[i].act_ds != p[0].act_ds) return MI355_E_SHAPE;
        if (p[i].m == 0) continue;
        a.W[n] = (const uint8_t *)p[i].W; a.w_row_stride[n] = p[i].w_row_stride;
        a.dst[n] = p[i].dst; a.dst_col_stride[n] = p[i].dst_col_stride; a.m[n] = (int)p[i].m;
        if ((p[i].residual == nullptr) != (p[i].sum == nullptr)) return MI355_E_NULL;
        if (((uintptr_t)p[i].residual | (uintptr_t)p[i].sum) & 3) return MI355_E_ALIGN;
        a.res[n] = p[i].residual; a.sum[n] = p[i].sum;
        blocks += (p[i].m + 3) / 4;
        if (blocks > 0x7fffffff) return MI355_E_SHAPE;
        a.blk_end[n] = (int)blocks;
        n++;
    }
    if (n == 0) return MI355_OK;
    a.n = n; a.k = (int)p[0].k; a.act_q = p[0].act_q; a.act_ds = p[0].act_ds;
    hipStream_t s = (hipStream_t)stream;
    switch (p[0].type) {
        case MI355_TYPE_Q4_0: return dispatch_group_cols<MI355_TYPE_Q4_0>(a, p[0].ncols, s);
        case MI355_TYPE_Q8_0: return dispatch_group_cols<MI355_TYPE_Q8_0>(a, p[0].ncols, s);
        case MI355_TYPE_Q4_K: return dispatch_group_cols<MI355_TYPE_Q4_K>(a, p[0].ncols, s);
        case MI355_TYPE_Q5_K: return dispatch_group_cols<MI355_TYPE_Q5_K>(a, p[0].ncols, s);
        case MI355_TYPE_Q6_K: return dispatch_group_cols<MI355_TYPE_Q6_K>(a, p[0].ncols, s);
    }
    return MI355_E_TYPE;
}

template <int TQ, int TK, int TV>
static int launch_qkv(const KQkv & a, hipStream_t stream) {
    const size_t need = (size_t)a.k + (size_t)a.k / QK * 8;
    if (need > LDS_ACT_LIMIT) return MI355_E_UNSUPPORTED;
    MI355_LAUNCH("mmvq_qkv_rope", (mmvq_qkv_rope_kernel<TQ, TK, TV>), dim3((unsigned)a.blk_end[2]), dim3(256), need, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_qkv_rope_matvec(const mi355_qkv_rope_args * p, void * stream) {
    if (!p || !p->wq || !p->wk || !p->wv || !p->q_out || !p->k_out || !p->v_out || !p->pos) return MI355_E_NULL;
    if (p->norm_x ? (!p->norm_w || !p->norm_out) : (!p->act_q || !p->act_ds)) return MI355_E_NULL;
    if (p->norm_x && (p->k > 4096 || p->k % 32 != 0)) return MI355_E_UNSUPPORTED;   // 256-thread workgroups: one 16-value chunk per thread
    if (p->norm_x && (((uintptr_t)p->norm_x | (uintptr_t)p->norm_w | (uintptr_t)p->norm_out) & 15)) return MI355_E_ALIGN;
    const int type_k = p->type_k ? p->type_k : p->type_qk;      // 0: wk has wq's type
    const int64_t bsa = mi355_blck_size(p->type_qk), bsb = mi355_blck_size(p->type_v), bsk = mi355_blck_size(type_k);
    auto weight = [](int t) { return t == MI355_TYPE_Q4_0 || t == MI355_TYPE_Q8_0 || t == MI355_TYPE_Q4_K || t == MI355_TYPE_Q5_K || t == MI355_TYPE_Q6_K; };
    if (!weight(p->type_qk) || !weight(p->type_v) || !weight(type_k)) return MI355_E_TYPE;
    if ((p->k_out_type != MI355_TYPE_F16 && p->k_out_type != MI355_TYPE_F32) || (p->v_out_type != MI355_TYPE_F16 && p->v_out_type != MI355_TYPE_F32)) return MI355_E_TYPE;
    if (p->k <= 0 || p->k % bsa != 0 || p->k % bsb != 0 || p->k % bsk != 0 || p->k > 0x7fffffff) return MI355_E_SHAPE;
    if (p->m_q <= 0 || p->m_k <= 0 || p->m_v <= 0 || p->m_q % 8 || p->m_k % 8 || p->m_v % 8 || p->m_q + p->m_k + p->m_v > 0x3fffffff) return MI355_E_SHAPE;
    if (p->head_dim < 2 || (p->head_dim & 1) || p->m_q % p->head_dim || p->m_k % p->head_dim || p->n_dims < 2 || (p->n_dims & 1) || p->n_dims > p->head_dim) return MI355_E_SHAPE;
    if (p->mode != 0 && p->mode != 2) return MI355_E_UNSUPPORTED;
    if (((uintptr_t)p->q_out & 3) || ((uintptr_t)p->k_out & 3) || ((uintptr_t)p->v_out & 3)) return MI355_E_ALIGN;
    if (!p->norm_x && (((uintptr_t)p->act_q & 15) || ((uintptr_t)p->act_ds & 7))) return MI355_E_ALIGN;
    if (!p->norm_x && p->k % QK_K == 0 && ((uintptr_t)p->act_ds & 15)) return MI355_E_ALIGN;
    const bool a16 = p->type_qk == MI355_TYPE_Q4_K || p->type_qk == MI355_TYPE_Q5_K, b16 = p->type_v == MI355_TYPE_Q4_K || p->type_v == MI355_TYPE_Q5_K;
    const bool k16 = type_k == MI355_TYPE_Q4_K || type_k == MI355_TYPE_Q5_K;
    if (((uintptr_t)p->wq | (uintptr_t)p->wq_row_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (((uintptr_t)p->wk | (uintptr_t)p->wk_row_stride) & (k16 ? 15 : 1)) return MI355_E_ALIGN;
    if (((uintptr_t)p->wv | (uintptr_t)p->wv_row_stride) & (b16 ? 15 : 1)) return MI355_E_ALIGN;
    if (p->wq_row_stride < mi355_row_size(p->type_qk, p->k) || p->wk_row_stride < mi355_row_size(type_k, p->k) ||
        p->wv_row_stride < mi355_row_size(p->type_v, p->k)) return MI355_E_SHAPE;
    KQkv a;
    a.W[0] = (const uint8_t *)p->wq; a.W[1] = (const uint8_t *)p->wk; a.W[2] = (const uint8_t *)p->wv;
    a.w_row_stride[0] = p->wq_row_stride; a.w_row_stride[1] = p->wk_row_stride; a.w_row_stride[2] = p->wv_row_stride;
    a.blk_end[0] = (int)(p->m_q / 8); a.blk_end[1] = a.blk_end[0] + (int)(p->m_k / 8); a.blk_end[2] = a.blk_end[1] + (int)(p->m_v / 8);
    a.act_q = p->act_q; a.act_ds = p->act_ds; a.k = (int)p->k;
    a.q_out = p->q_out; a.k_out = p->k_out; a.v_out = p->v_out;
    a.k_f16 = p->k_out_type == MI355_TYPE_F16; a.v_f16 = p->v_out_type == MI355_TYPE_F16;
    a.pos = p->pos; a.ff = p->freq_factors;
    a.nx = p->norm_x; a.nw = p->norm_w; a.nout = p->norm_out; a.eps = p->norm_eps;
    a.rp = rope_params(p->n_dims, p->mode, p->head_dim, p->n_ctx_orig, p->freq_base, p->freq_scale, p->ext_factor, p->attn_factor,
                       p->beta_fast, p->beta_slow);
    hipStream_t s = (hipStream_t)stream;
    const int tq = p->type_qk, tv = p->type_v;
    if (type_k != tq) {      // 8-expert Q4_K_M: wq Q4_K, wk and wv bumped to Q8_0 (llama-quant.cpp:300-311)
        if (tq == MI355_TYPE_Q4_K && type_k == MI355_TYPE_Q8_0 && tv == MI355_TYPE_Q8_0) return launch_qkv<MI355_TYPE_Q4_K, MI355_TYPE_Q8_0, MI355_TYPE_Q8_0>(a, s);
        return MI355_E_UNSUPPORTED;
    }
    if (tq == MI355_TYPE_Q4_K && tv == MI355_TYPE_Q4_K) return launch_qkv<MI355_TYPE_Q4_K, MI355_TYPE_Q4_K, MI355_TYPE_Q4_K>(a, s);
    if (tq == MI355_TYPE_Q4_K && tv == MI355_TYPE_Q6_K) return launch_qkv<MI355_TYPE_Q4_K, MI355_TYPE_Q4_K, MI355_TYPE_Q6_K>(a, s);
    if (tq == MI355_TYPE_Q4_K && tv == MI355_TYPE_Q5_K) return launch_qkv<MI355_TYPE_Q4_K, MI355_TYPE_Q4_K, MI355_TYPE_Q5_K>(a, s);
    if (tq == MI355_TYPE_Q5_K && tv == MI355_TYPE_Q5_K) return launch_qkv<MI355_TYPE_Q5_K, MI355_TYPE_Q5_K, MI355_TYPE_Q5_K>(a, s);
    if (tq == MI355_TYPE_Q6_K && tv == MI355_TYPE_Q6_K) return launch_qkv<MI355_TYPE_Q6_K, MI355_TYPE_Q6_K, MI355_TYPE_Q6_K>(a, s);
    if (tq == MI355_TYPE_Q8_0 && tv == MI355_TYPE_Q8_0) return launch_qkv<MI355_TYPE_Q8_0, MI355_TYPE_Q8_0, MI355_TYPE_Q8_0>(a, s);
    if (tq == MI355_TYPE_Q4_0 && tv == MI355_TYPE_Q4_0) return launch_qkv<MI355_TYPE_Q4_0, MI355_TYPE_Q4_0, MI355_TYPE_Q4_0>(a, s);
    return MI355_E_UNSUPPORTED;
}

template <int TYPE>
static int launch_glu_mv(const KGlu & a, hipStream_t stream) {
    const size_t need = (((size_t)a.k + (size_t)a.k / QK * 8 + 15) & ~(size_t)15) + 64 * sizeof(float);
    if (need > LDS_ACT_LIMIT) return MI355_E_UNSUPPORTED;
    // 4 outputs per wave.  (8 per wave -- half the workgroups, for grids beyond the 512 workgroups a full chip holds at
    // once -- measured slower: 2x28672x8192 q4_K 61 us against 56; the plugin keeps such shapes on the grouped mat-vec.)
    MI355_LAUNCH("mmvq_glu", (mmvq_glu_kernel<TYPE, 4>), dim3((unsigned)(a.m / 32), (unsigned)(a.ids ? a.n_slots : 1)), dim3(GLU_THREADS), need, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_glu_matvec(const mi355_glu_matvec_args * p, void * stream) {
    if (!p || !p->w_gate || !p->w_up || !p->dst) return MI355_E_NULL;
    if (p->norm_x ? (!p->norm_w || !p->norm_out) : (!p->act_q || !p->act_ds)) return MI355_E_NULL;
    if ((p->out_q == nullptr) != (p->out_ds == nullptr)) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(p->type);
    if (bs == 0 || p->type == MI355_TYPE_F32 || p->type == MI355_TYPE_F16 || p->type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (p->glu_op < 0 || p->glu_op > 2) return MI355_E_UNSUPPORTED;
    if (p->ids && (p->n_slots < 1 || p->n_slots > 64 || p->n_expert < 1 || p->norm_x)) return MI355_E_SHAPE;
    if (p->m <= 0 || p->m % 32 != 0 || p->m > 0x3fffffff || p->k <= 0 || p->k % bs != 0 || p->k > 0x7fffffff) return MI355_E_SHAPE;
    if (((uintptr_t)p->dst & 15) || ((uintptr_t)p->gate_out & 3) || ((uintptr_t)p->up_out & 3) || ((uintptr_t)p->out_q & 3) ||
        ((uintptr_t)p->out_ds & 7)) return MI355_E_ALIGN;
    if (p->norm_x) {
        if (p->k > 8192 || p->k % 32 != 0) return MI355_E_UNSUPPORTED;        // one 16-value chunk per thread
        if (((uintptr_t)p->norm_x | (uintptr_t)p->norm_w | (uintptr_t)p->norm_out) & 15) return MI355_E_ALIGN;
    } else {
        if (((uintptr_t)p->act_q & 15) || ((uintptr_t)p->act_ds & 7)) return MI355_E_ALIGN;
        if (p->k % QK_K == 0 && ((uintptr_t)p->act_ds & 15)) return MI355_E_ALIGN;
    }
    const bool a16 = p->type == MI355_TYPE_Q4_K || p->type == MI355_TYPE_Q5_K;
    const uintptr_t wal = (uintptr_t)p->w_gate | (uintptr_t)p->gate_row_stride | (uintptr_t)p->w_up | (uintptr_t)p->up_row_stride;
    if (wal & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (p->gate_row_stride < mi355_row_size(p->type, p->k) || p->up_row_stride < mi355_row_size(p->type, p->k)) return MI355_E_SHAPE;
    // the q8_1 output must not be the buffer the activations are read from: workgroups finish at different times
    {
        const char * i0 = (const char *)p->act_q, * i1 = i0 + p->k, * d0 = (const char *)p->act_ds, * d1 = d0 + p->k / QK * 8;
        const char * o0 = (const char *)p->out_q, * o1 = o0 + p->m, * e0 = (const char *)p->out_ds, * e1 = e0 + p->m / QK * 8;
        if (p->out_q && !p->norm_x && ((o0 < i1 && i0 < o1) || (o0 < d1 && d0 < o1) || (e0 < i1 && i0 < e1) || (e0 < d1 && d0 < e1))) return MI355_E_SHAPE;
    }
    KGlu a;
    a.wg = (const uint8_t *)p->w_gate; a.wu = (const uint8_t *)p->w_up; a.rsg = p->gate_row_stride; a.rsu = p->up_row_stride;
    a.act_q = p->act_q; a.act_ds = p->act_ds; a.k = (int)p->k; a.m = (int)p->m; a.op = p->glu_op;
    a.gate_out = p->gate_out; a.up_out = p->up_out; a.dst = p->dst; a.out_q = p->out_q; a.out_ds = p->out_ds;
    a.nx = p->norm_x; a.nw = p->norm_w; a.nout = p->norm_out; a.eps = p->norm_eps;
    a.ids = p->ids; a.n_expert = p->n_expert; a.n_slots = p->n_slots; a.wg_ex_stride = p->gate_expert_stride; a.wu_ex_stride = p->up_expert_stride;
    if (p->ids && ((p->gate_expert_stride | p->up_expert_stride) & (a16 ? 15 : 1))) return MI355_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (p->type) {
        case MI355_TYPE_Q4_0: return launch_glu_mv<MI355_TYPE_Q4_0>(a, s);
        case MI355_TYPE_Q8_0: return launch_glu_mv<MI355_TYPE_Q8_0>(a, s);
        case MI355_TYPE_Q4_K: return launch_glu_mv<MI355_TYPE_Q4_K>(a, s);
        case MI355_TYPE_Q5_K: return launch_glu_mv<MI355_TYPE_Q5_K>(a, s);
        case MI355_TYPE_Q6_K: return launch_glu_mv<MI355_TYPE_Q6_K>(a, s);
    }
    return MI355_E_TYPE;
}

template <int TYPE>
static int launch_moe_down(const KMoeDown & a, hipStream_t stream) {
    const size_t need = (size_t)a.n_slots * ((size_t)a.k + (size_t)a.k / QK * 8);
    const dim3 grid((unsigned)((a.m + 3) / 4));
    if (need <= LDS_ACT_LIMIT) MI355_LAUNCH("mmvq_moe_down", (mmvq_moe_down_kernel<TYPE, true>), grid, dim3(256), need, stream, a);
    else                       MI355_LAUNCH("mmvq_moe_down", (mmvq_moe_down_kernel<TYPE, false>), grid, dim3(256), 0, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_moe_down_matvec(const mi355_moe_down_args * p, void * stream) {
    if (!p || !p->W || !p->act_q || !p->act_ds || !p->ids || !p->weights || !p->weighted_out || !p->sum_out) return MI355_E_NULL;
    if ((p->residual == nullptr) != (p->residual_out == nullptr)) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(p->type);
    if (bs == 0 || p->type == MI355_TYPE_F32 || p->type == MI355_TYPE_F16 || p->type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (p->m <= 0 || p->m > 0x7fffffff || p->k <= 0 || p->k % bs != 0 || p->k > 0x7fffffff || p->n_expert < 1) return MI355_E_SHAPE;
    if (p->n_slots < 1 || p->n_slots > MOE_DOWN_MAX_SLOTS) return MI355_E_UNSUPPORTED;
    if (((uintptr_t)p->act_q & 15) || ((uintptr_t)p->act_ds & 7) || (p->k & 15)) return MI355_E_ALIGN;
    if (p->k % QK_K == 0 && ((uintptr_t)p->act_ds & 15)) return MI355_E_ALIGN;
    const bool a16 = p->type == MI355_TYPE_Q4_K || p->type == MI355_TYPE_Q5_K;
    if (((uintptr_t)p->W | (uintptr_t)p->w_row_stride | (uintptr_t)p->w_expert_stride) & (a16 ? 15 : 1)) return MI355_E_ALIGN;
    if (p->w_row_stride < mi355_row_size(p->type, p->k)) return MI355_E_SHAPE;
    if (((uintptr_t)p->out | (uintptr_t)p->weighted_out | (uintptr_t)p->sum_out | (uintptr_t)p->residual | (uintptr_t)p->residual_out |
         (uintptr_t)p->weights | (uintptr_t)p->ids) & 3) return MI355_E_ALIGN;
    KMoeDown a;
    a.W = (const uint8_t *)p->W; a.w_row_stride = p->w_row_stride; a.w_ex_stride = p->w_expert_stride;
    a.act_q = p->act_q; a.act_ds = p->act_ds; a.ids = p->ids; a.wts = p->weights;
    a.y_out = p->out; a.yw_out = p->weighted_out; a.sum_out = p->sum_out; a.res = p->residual;
    a.res_out = p->residual ? p->residual_out : nullptr;
    a.m = (int)p->m; a.k = (int)p->k; a.n_slots = p->n_slots; a.n_expert = p->n_expert;
    hipStream_t s = (hipStream_t)stream;
    switch (p->type) {
        case MI355_TYPE_Q4_0: return launch_moe_down<MI355_TYPE_Q4_0>(a, s);
        case MI355_TYPE_Q8_0: return launch_moe_down<MI355_TYPE_Q8_0>(a, s);
        case MI355_TYPE_Q4_K: return launch_moe_down<MI355_TYPE_Q4_K>(a, s);
        case MI355_TYPE_Q5_K: return launch_moe_down<MI355_TYPE_Q5_K>(a, s);
        case MI355_TYPE_Q6_K: return launch_moe_down<MI355_TYPE_Q6_K>(a, s);
    }
    return MI355_E_TYPE;
}
