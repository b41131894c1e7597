// mmvq_units.h -- device code shared by the decode mat-vec kernels (mmvq.hip) and the persistent decode-layer kernel
// (mmvq_layer.hip): per-type unit traits (load / decode / dot of what one lane processes per step), the LDS-DMA piece copy,
// counted vmcnt waits and the LDS images of the K-quant blocks.  Everything here has internal linkage.
#pragma once
#include "common.h"
#include <mutex>
#include <set>
#include <utility>
#include <stdlib.h>

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

// ---- Q4_1: 20 B = {half d, half m, u8 qs[16]} (ggml-common.h:175-187); value = nibble * d + m (ggml-quants.c:269-288)
// The block sum of the activations comes with them ({d, sum} of the q8_1 SoA), as for the K-quants' mins.
struct Act32s { v4i q0, q1; float d, s; };
template <class AP>
static __device__ __forceinline__ Act32s load_act32s(const AP a, int u) {
    Act32s r;
    r.q0 = *(const v4i *)(a.q + u * 32);
    r.q1 = *(const v4i *)(a.q + u * 32 + 16);
    const v2f ds = *(const v2f *)(a.ds + u * 2);
    r.d = ds.x; r.s = ds.y;
    return r;
}
// bit i of the low 4 bits of b -> bit 0 of byte i
static __device__ __forceinline__ int spread4(uint32_t b) { return (int)(((b & 0xfu) * 0x00204081u) & 0x01010101u); }

template <> struct Unit<MI355_TYPE_Q4_1> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i qs; uint16_t d, m; };
    struct Dec { v4i lo, hi; float d, m; };
    typedef Act32s Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)u * 20;
        Raw r;
        r.qs = ld16_a2(blk + 4);
        r.d  = ld_u16(blk); r.m = ld_u16(blk + 2);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int) {
        Dec d;
        d.lo = r.qs & 0x0f0f0f0f; d.hi = (r.qs >> 4) & 0x0f0f0f0f;
        d.d = h2f(r.d); d.m = h2f(r.m);
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return load_act32s(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { s = dot4(w.lo[j], a.q0[j], s); s = dot4(w.hi[j], a.q1[j], s); }
        return w.d * a.d * (float)s + w.m * (a.d * a.s);
    }
};

// ---- Q5_0: 22 B = {half d, u8 qh[4], u8 qs[16]} (ggml-common.h:191-196); value = ((nibble | bit << 4) - 16) * d, bit e of
// qh belongs to element e (ggml-quants.c:290-314)
template <> struct Unit<MI355_TYPE_Q5_0> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i qs; uint16_t qh0, qh1, d; };
    struct Dec { v4i lo, hi; float d; };
    typedef Unit<MI355_TYPE_Q8_0>::Act Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)u * 22;
        Raw r;
        r.qs = ld16_a2(blk + 6);
        r.d = ld_u16(blk); r.qh0 = ld_u16(blk + 2); r.qh1 = ld_u16(blk + 4);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int) {
        Dec d;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d.lo[j] = sub_bytes((r.qs[j] & 0x0f0f0f0f) | (spread4((uint32_t)r.qh0 >> (4 * j)) << 4), 0x10101010);
            d.hi[j] = sub_bytes(((r.qs[j] >> 4) & 0x0f0f0f0f) | (spread4((uint32_t)r.qh1 >> (4 * j)) << 4), 0x10101010);
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

// ---- Q5_1: 24 B = {half d, half m, u8 qh[4], u8 qs[16]} (ggml-common.h:198-209); value = (nibble | bit << 4) * d + m
// (ggml-quants.c:316-341)
template <> struct Unit<MI355_TYPE_Q5_1> {
    static constexpr int ELEMS = 32;
    struct Raw { v4i qs; uint16_t qh0, qh1, d, m; };
    struct Dec { v4i lo, hi; float d, m; };
    typedef Act32s Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)u * 24;
        Raw r;
        r.qs = ld16_a2(blk + 8);
        r.d = ld_u16(blk); r.m = ld_u16(blk + 2); r.qh0 = ld_u16(blk + 4); r.qh1 = ld_u16(blk + 6);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int) {
        Dec d;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d.lo[j] = (r.qs[j] & 0x0f0f0f0f) | (spread4((uint32_t)r.qh0 >> (4 * j)) << 4);
            d.hi[j] = ((r.qs[j] >> 4) & 0x0f0f0f0f) | (spread4((uint32_t)r.qh1 >> (4 * j)) << 4);
        }
        d.d = h2f(r.d); d.m = h2f(r.m);
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return load_act32s(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { s = dot4(w.lo[j], a.q0[j], s); s = dot4(w.hi[j], a.q1[j], s); }
        return w.d * a.d * (float)s + w.m * (a.d * a.s);
    }
};

// ---- the third batch (the remainder): q2_K, q3_K, iq4_nl, iq4_xs -------------------------------------------------
// kvalues_iq4nl (ggml-common.h:1077-1079) packed four to a dword; iq4nl_lut4 maps the four 4-bit codes in the bytes of `idx`
// (each 0..15) to their int8 values with three v_perm_b32: entries 0..7, entries 8..15, then a per-byte pick by bit 3.
static __device__ __forceinline__ int iq4nl_lut4(uint32_t idx) {
    constexpr uint32_t T0 = 0xBFAD9881u, T1 = 0xF6EADDCFu, T2 = 0x26190D01u, T3 = 0x71594535u;
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(T1, T0, sel), hi = __builtin_amdgcn_perm(T3, T2, sel);
    return (int)__builtin_amdgcn_perm(hi, lo, 0x03020100u | ((idx >> 1) & 0x04040404u));
}

// ---- IQ4_NL: 18 B = {half d, u8 qs[16]} (ggml-common.h:405-410); as q4_0 with the nibble looked up in the code book
// (dequantize_row_iq4_nl, ggml-quants.c:2428-2444)
template <> struct Unit<MI355_TYPE_IQ4_NL> {
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
            d.lo[j] = iq4nl_lut4((uint32_t)r.qs[j] & 0x0f0f0f0fu);
            d.hi[j] = iq4nl_lut4(((uint32_t)r.qs[j] >> 4) & 0x0f0f0f0fu);
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

// ---- IQ4_XS: 136 B = {half d, u16 scales_h, u8 scales_l[4], u8 qs[128]} (ggml-common.h:412-418).  Sub-block ib (32 values): qs[16 ib ..],
// low nibbles first; its 6-bit scale is nibble ib of scales_l | bits 2 ib.. of scales_h << 4, minus 32 (dequantize_row_iq4_xs,
// ggml-quants.c:2446-2468).  unit = sub-blocks 2g, 2g + 1 of block b.
template <> struct Unit<MI355_TYPE_IQ4_XS> {
    static constexpr int ELEMS = 64;
    struct Raw { v4i q0, q1; v2i hdr; };
    struct Dec { v4i lo0, hi0, lo1, hi1; float d0, d1; };
    typedef Act64 Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 2) * 136;
        Raw r;
        r.q0  = ld16_a2(blk + 8 + (u & 3) * 32);
        r.q1  = ld16_a2(blk + 24 + (u & 3) * 32);
        r.hdr = ((const v2i_a2 *)blk)->v;
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        const int g = u & 3;
        Dec d;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d.lo0[j] = iq4nl_lut4((uint32_t)r.q0[j] & 0x0f0f0f0fu); d.hi0[j] = iq4nl_lut4(((uint32_t)r.q0[j] >> 4) & 0x0f0f0f0fu);
            d.lo1[j] = iq4nl_lut4((uint32_t)r.q1[j] & 0x0f0f0f0fu); d.hi1[j] = iq4nl_lut4(((uint32_t)r.q1[j] >> 4) & 0x0f0f0f0fu);
        }
        const uint32_t sl = ((uint32_t)r.hdr.y >> (8 * g)) & 0xffu, sh = ((uint32_t)r.hdr.x >> (16 + 4 * g)) & 0xfu;
        const int ls0 = (int)((sl & 0xfu) | ((sh & 3u) << 4)), ls1 = (int)((sl >> 4) | ((sh >> 2) << 4));
        const float dd = h2f((uint16_t)((uint32_t)r.hdr.x & 0xffffu));
        d.d0 = dd * (float)(ls0 - 32); d.d1 = dd * (float)(ls1 - 32);
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return load_act64(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s0 = dot4(w.lo0[j], a.q[0][j], s0); s0 = dot4(w.hi0[j], a.q[1][j], s0);
            s1 = dot4(w.lo1[j], a.q[2][j], s1); s1 = dot4(w.hi1[j], a.q[3][j], s1);
        }
        return w.d0 * a.ds.x * (float)s0 + w.d1 * a.ds.z * (float)s1;
    }
};

// ---- Q2_K: 84 B = {u8 scales[16], u8 qs[64], half d, half dmin} (ggml-common.h:256-267).  Element 128 n + 32 j + 16 h + l is
// (qs[32 n + 16 h + l] >> 2 j) & 3 with scale byte 8 n + 2 j + h: low nibble * d the scale, high nibble * dmin the minimum
// (dequantize_row_q2_K, ggml-quants.c:706-735).  unit (b, n, h) as for q6_K: 16 bytes of qs -> 4 groups (j) of 16 weights.
// A minimum applies to 16 values, half a q8_1 block, so the activations' sum is taken here (4 dot4 with ones per group).
typedef Unit<MI355_TYPE_Q6_K>::Act ActK16;
template <> struct Unit<MI355_TYPE_Q2_K> {
    static constexpr int ELEMS = 64;
    struct Raw { v4i qs; v2i sc; uint16_t d, dmin; };
    struct Dec { v4i q[4]; float dl[4], ml[4]; };
    typedef ActK16 Act;
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 2) * 84;
        const int n = (u >> 1) & 1, h = u & 1;
        Raw r;
        r.qs = ld16_a2(blk + 16 + 32 * n + 16 * h);
        r.sc = ((const v2i_a2 *)(blk + 8 * n))->v;
        r.d = ld_u16(blk + 80); r.dmin = ld_u16(blk + 82);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        const int h = u & 1;
        Dec d;
        const float dd = h2f(r.d), dm = h2f(r.dmin);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d.q[j] = (r.qs >> (2 * j)) & 0x03030303;
            const uint32_t s = ((uint32_t)(j < 2 ? r.sc.x : r.sc.y) >> (8 * h + 16 * (j & 1))) & 0xffu;
            d.dl[j] = dd * (float)(s & 0xfu); d.ml[j] = dm * (float)(s >> 4);
        }
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return Unit<MI355_TYPE_Q6_K>::load_act(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int s = 0, t = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { s = dot4(w.q[i][j], a.q[i][j], s); t = dot4(0x01010101, a.q[i][j], t); }
            acc += a.d[i] * (w.dl[i] * (float)s - w.ml[i] * (float)t);
        }
        return acc;
    }
};

// ---- Q3_K: 110 B = {u8 hmask[32], u8 qs[64], u8 scales[12], half d} (ggml-common.h:273-279).  Element 128 n + 32 j + 16 h + l is
// ((qs[32 n + 16 h + l] >> 2 j) & 3) - (bit 4 n + j of hmask[16 h + l] ? 0 : 4), scale 8 n + 2 j + h of sixteen 6-bit values
// (low 4 bits: nibble (is >> 3) of byte is & 7; high 2 bits: bits 2 (is >> 2).. of byte 8 + (is & 3)), minus 32
// (dequantize_row_q3_K, ggml-quants.c:1050-1099).  unit (b, n, h): 16 bytes of qs + 16 of hmask -> 4 groups of 16 weights.
template <> struct Unit<MI355_TYPE_Q3_K> {
    static constexpr int ELEMS = 64;
    struct Raw { v4i qs, hm; uint32_t s0, s1, s2; uint16_t d; };
    struct Dec { v4i q[4]; float dl[4]; };
    typedef ActK16 Act;
    static __device__ __forceinline__ uint32_t ld_u32_a2(const uint8_t * p) { return (uint32_t)ld_u16(p) | ((uint32_t)ld_u16(p + 2) << 16); }
    static __device__ __forceinline__ Raw load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (int64_t)(u >> 2) * 110;
        const int n = (u >> 1) & 1, h = u & 1;
        Raw r;
        r.qs = ld16_a2(blk + 32 + 32 * n + 16 * h);
        r.hm = ld16_a2(blk + 16 * h);
        r.s0 = ld_u32_a2(blk + 96); r.s1 = ld_u32_a2(blk + 100); r.s2 = ld_u32_a2(blk + 104);
        r.d  = ld_u16(blk + 108);
        return r;
    }
    static __device__ __forceinline__ Dec decode(const Raw & r, int u) {
        const int n = (u >> 1) & 1, h = u & 1;
        Dec d;
        const float dd = h2f(r.d);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const v4i v = ((r.qs >> (2 * j)) & 0x03030303) | (((r.hm >> (4 * n + j)) & 0x01010101) << 2);
#pragma unroll
            for (int c = 0; c < 4; c++) d.q[j][c] = sub_bytes(v[c], 0x04040404);
            const int is = 8 * n + 2 * j + h;
            const uint32_t lo = ((is & 4 ? r.s1 : r.s0) >> (8 * (is & 3) + 4 * (is >> 3))) & 0xfu;
            const uint32_t hi = (r.s2 >> (8 * (is & 3) + 2 * (is >> 2))) & 3u;
            d.dl[j] = dd * (float)((int)(lo | (hi << 4)) - 32);
        }
        return d;
    }
    template <class AP> static __device__ __forceinline__ Act load_act(const AP a, int u) { return Unit<MI355_TYPE_Q6_K>::load_act(a, u); }
    static __device__ __forceinline__ float dot(const Dec & w, const Act & a) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) s = dot4(w.q[i][j], a.q[i][j], s);
            acc += w.dl[i] * a.d[i] * (float)s;
        }
        return acc;
    }
};

static __device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

typedef __attribute__((address_space(3))) uint8_t lds_u8;

static __device__ __forceinline__ unsigned lds_addr(const void * p) { return (unsigned)(uintptr_t)(const lds_u8 *)p; }

// 1 KiB: lane l copies 16 B from `gsrc` (its own address) to LDS byte address `lds_dst` (wave-uniform) + 16 l.
// The lgkmcnt(0) makes this wave's earlier LDS reads of the slot complete before it can be overwritten.
static __device__ __forceinline__ void dma_piece_nt(const void * gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

static __device__ __forceinline__ void dma_piece_plain(const void * gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// wait until at most q vector-memory operations of this wave are outstanding (q wave-uniform)
static __device__ __forceinline__ void wait_vm_le(int q) {
    switch (q) {
        case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1:  asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2:  asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3:  asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4:  asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5:  asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6:  asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7:  asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8:  asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9:  asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: if (q < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // q >= 16: nothing this deep is ever in flight
    }
}

constexpr int DMA_MAX_SLOTS = 16;

template <int TYPE> struct DmaImg { static constexpr bool OK = false; };

static __device__ __forceinline__ v4i lds16(const lds_u8 * ring, int off) { return *(const __attribute__((address_space(3))) v4i *)(ring + off); }

template <> struct DmaImg<MI355_TYPE_Q4_K> {
    static constexpr bool OK = true;
    static constexpr int IMG = 144, CPB = 9;         // bytes of a block's image in LDS, 16-byte chunks per block
    static __device__ __forceinline__ int src(int c) { return 16 * c; }      // byte offset in the row of image chunk c
    static __device__ __forceinline__ Unit<MI355_TYPE_Q4_K>::Raw read(const lds_u8 * ring, int mask, int io, int u) {
        const int g = u & 3;
        Unit<MI355_TYPE_Q4_K>::Raw r;
        r.hdr = lds16(ring, io & mask);
        r.q0  = lds16(ring, (io + 16 + g * 32) & mask);
        r.q1  = lds16(ring, (io + 32 + g * 32) & mask);
        return r;
    }
};
template <> struct DmaImg<MI355_TYPE_Q5_K> {
    static constexpr bool OK = true;
    static constexpr int IMG = 176, CPB = 11;
    static __device__ __forceinline__ int src(int c) { return 16 * c; }
    static __device__ __forceinline__ Unit<MI355_TYPE_Q5_K>::Raw read(const lds_u8 * ring, int mask, int io, int u) {
        const int g = u & 3;
        Unit<MI355_TYPE_Q5_K>::Raw r;
        r.hdr = lds16(ring, io & mask);
        r.h0  = lds16(ring, (io + 16) & mask);
        r.h1  = lds16(ring, (io + 32) & mask);
        r.q0  = lds16(ring, (io + 48 + g * 32) & mask);
        r.q1  = lds16(ring, (io + 64 + g * 32) & mask);
        return r;
    }
};
template <> struct DmaImg<MI355_TYPE_Q6_K> {
    static constexpr bool OK = true;
    static constexpr int IMG = 224, CPB = 14;        // 210-byte blocks, re-laid out: see src()
    static __device__ __forceinline__ int src(int c) {
        const int b = (int)(((unsigned)c * 18725u) >> 18), j = c - 14 * b;         // c / 14 for c < 2^16
        return b * 210 + (j < 13 ? 16 * j : 194);
    }
    static __device__ __forceinline__ Unit<MI355_TYPE_Q6_K>::Raw read(const lds_u8 * ring, int mask, int io, int u) {
        const int n = (u >> 1) & 1, h = u & 1;
        Unit<MI355_TYPE_Q6_K>::Raw r;
        r.qla = lds16(ring, (io + 64 * n + 16 * h) & mask);
        r.qlb = lds16(ring, (io + 64 * n + 32 + 16 * h) & mask);
        r.qh  = lds16(ring, (io + 128 + 32 * n + 16 * h) & mask);
        r.sc  = *(const __attribute__((address_space(3))) v2i *)(ring + ((io + 192 + 8 * n) & mask));
        r.d   = *(const __attribute__((address_space(3))) uint16_t *)(ring + ((io + 222) & mask));
        return r;
    }
};

// Slots a wave's ring must have for type TYPE at row length k.  One step of DStream::run reads min(nb, 16) block images starting
// at r * (row image) + it * 16 * IMG -- wherever that falls inside a 1 KiB piece -- and all pieces it touches must be resident
// together: q4_K / q5_K steps (2304 / 2816 B) fit 4 slots at any offset; a q6_K step is 3584 B and fits 4 slots only when
// every start is a multiple of 512 (row images of whole 16-block steps: K = 4096, 8192, 28672), else 8 (K = 14336: 12.25 KiB per
// row, the second row of a wave starts 256 B into a piece).
// multi_row: a wave streams more than one row (the next row's image starts where this one ends, wherever that is inside a piece)
template <int TYPE>
static int dma_need_slots(int k, bool multi_row = true) {
    if constexpr (!DmaImg<TYPE>::OK) return 0;
    else {
        const int nb = k / QK_K, img = DmaImg<TYPE>::IMG;
        const int span = (nb < 16 ? nb : 16) * img;
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        const int g = gcd(multi_row ? gcd((nb * img) % 1024, 1024) : 1024, gcd((16 * img) % 1024, 1024));     // every step starts at a multiple of g
        const int bytes = span + 1024 - g;
        int sl = 2;
        while (sl * 1024 < bytes) sl *= 2;
        return sl;
    }
}

// dynamic LDS beyond 64 KB has to be asked for, per kernel and device
static hipError_t allow_lds_any(const void * fn, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);   // the kernels' static arrays take < 1 KB
    if (e != hipSuccess) return e;
    done.insert({fn, dev});
    return hipSuccess;
}
template <class K>
static hipError_t allow_lds(K kernel, size_t bytes) { return allow_lds_any((const void *)kernel, bytes); }

// LDS / global flavours of the activation pointers (address spaces stay static so the LDS build
// reads with ds_read_b128, not flat loads)
struct ActLds { const __attribute__((address_space(3))) int8_t * q; const __attribute__((address_space(3))) float * ds; };

} // namespace
