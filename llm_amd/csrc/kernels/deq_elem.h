// deq_elem.h -- one dequantized element of a GGUF block row, bit-exact with the reference's
// dequantize_row_* (ggml/src/ggml-quants.c:249-267,343-357,706-735,1050-1099,1274-1296,1476-1501,1684-1713,2428-2468).
// Every product / difference is an explicit single-rounding intrinsic (no FMA contraction).
#pragma once
#include "common.h"

static __device__ __forceinline__ void scale_min_k4(int j, const uint8_t * q, int & sc, int & m) {
    // get_scale_min_k4, ggml-quants.c:625-632
    if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
    else       { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// kvalues_iq4nl, ggml-common.h:1077-1079
static __device__ __forceinline__ float iq4nl_value(int code) {
    constexpr uint64_t LO = 0xF6EADDCFBFAD9881ull, HI = 0x7159453526190D01ull;
    return (float)(int8_t)(((code & 8) ? HI : LO) >> (8 * (code & 7)));
}

template <int TYPE>
static __device__ __forceinline__ float deq_elem(const uint8_t * __restrict__ base, int64_t i) {
    if (TYPE == MI355_TYPE_Q4_0) {
        const uint8_t * b = base + (i / QK) * 18;
        const int j = i % QK;
        const float d = h2f(*(const uint16_t *)b);
        const int q = j < 16 ? (b[2 + j] & 0x0F) : (b[2 + j - 16] >> 4);
        return __fmul_rn((float)(q - 8), d);
    } else if (TYPE == MI355_TYPE_Q4_1) {
        // dequantize_row_q4_1, ggml-quants.c:269-288: y = q * d + m
        const uint8_t * b = base + (i / QK) * 20;
        const int j = i % QK;
        const float d = h2f(*(const uint16_t *)b), m = h2f(*(const uint16_t *)(b + 2));
        const int q = j < 16 ? (b[4 + j] & 0x0F) : (b[4 + j - 16] >> 4);
        return __fadd_rn(__fmul_rn((float)q, d), m);
    } else if (TYPE == MI355_TYPE_Q5_0 || TYPE == MI355_TYPE_Q5_1) {
        // dequantize_row_q5_0 / _q5_1, ggml-quants.c:290-341: bit j (low half) / j + 16 (high half) of qh is the 5th bit
        const int bsz = TYPE == MI355_TYPE_Q5_0 ? 22 : 24, qo = TYPE == MI355_TYPE_Q5_0 ? 6 : 8;
        const uint8_t * b = base + (i / QK) * bsz;
        const int j = i % QK;
        const float d = h2f(*(const uint16_t *)b);
        const uint8_t * qhp = b + (TYPE == MI355_TYPE_Q5_0 ? 2 : 4);
        const uint32_t qh = (uint32_t)qhp[0] | ((uint32_t)qhp[1] << 8) | ((uint32_t)qhp[2] << 16) | ((uint32_t)qhp[3] << 24);
        const int nib = j < 16 ? (b[qo + j] & 0x0F) : (b[qo + j - 16] >> 4);
        const int q = nib | (int)(((qh >> j) & 1u) << 4);
        if (TYPE == MI355_TYPE_Q5_0) return __fmul_rn((float)(q - 16), d);
        return __fadd_rn(__fmul_rn((float)q, d), h2f(*(const uint16_t *)(b + 2)));
    } else if (TYPE == MI355_TYPE_Q8_0) {
        const uint8_t * b = base + (i / QK) * 34;
        const float d = h2f(*(const uint16_t *)b);
        return __fmul_rn((float)((const int8_t *)b)[2 + i % QK], d);
    } else if (TYPE == MI355_TYPE_Q8_1) {
        const uint8_t * b = base + (i / QK) * 36;
        const float d = h2f(*(const uint16_t *)b);
        return __fmul_rn((float)((const int8_t *)b)[4 + i % QK], d);
    } else if (TYPE == MI355_TYPE_Q4_K || TYPE == MI355_TYPE_Q5_K) {
        const int bsz = TYPE == MI355_TYPE_Q4_K ? 144 : 176;
        const uint8_t * b = base + (i / QK_K) * bsz;
        const int e  = i % QK_K;
        const int j  = e / 32;           // sub-block 0..7
        const int l  = e % 32;
        const float d   = h2f(*(const uint16_t *)b);
        const float min = h2f(*(const uint16_t *)(b + 2));
        int sc, m;
        scale_min_k4(j, b + 4, sc, m);
        const float d1 = __fmul_rn(d, (float)sc);
        const float m1 = __fmul_rn(min, (float)m);
        const uint8_t * qs = b + (TYPE == MI355_TYPE_Q4_K ? 16 : 48) + (j / 2) * 32;
        int q = (j & 1) ? (qs[l] >> 4) : (qs[l] & 0xF);
        if (TYPE == MI355_TYPE_Q5_K) q += ((b[16 + l] >> j) & 1) ? 16 : 0;
        return __fsub_rn(__fmul_rn(d1, (float)q), m1);
    } else if (TYPE == MI355_TYPE_Q6_K) {
        const uint8_t * b = base + (i / QK_K) * 210;
        const int e  = i % QK_K;
        const int n  = e / 128;          // half of the super-block
        const int r  = e % 128;
        const int g  = r / 32;           // which of q1..q4
        const int l  = r % 32;
        const uint8_t * ql = b + 64 * n;
        const uint8_t * qh = b + 128 + 32 * n;
        const int8_t  * sc = (const int8_t *)(b + 192 + 8 * n);
        const float d = h2f(*(const uint16_t *)(b + 208));
        const int lo = (g & 1) ? ql[l + 32] : ql[l];
        const int nib = (g & 2) ? (lo >> 4) : (lo & 0xF);
        const int q = (int)(int8_t)(nib | (((qh[l] >> (2 * g)) & 3) << 4)) - 32;
        return __fmul_rn(__fmul_rn(d, (float)sc[l / 16 + 2 * g]), (float)q);
    } else if (TYPE == MI355_TYPE_Q2_K) {
        // dequantize_row_q2_K, ggml-quants.c:706-735: element 128 n + 32 j + 16 h + l
        const uint8_t * b = base + (i / QK_K) * 84;
        const int e = i % QK_K, n = e / 128, j = (e % 128) / 32, h = (e % 32) / 16, l = e % 16;
        const int sc = b[8 * n + 2 * j + h];
        const float dl = __fmul_rn(h2f(*(const uint16_t *)(b + 80)), (float)(sc & 0xF));
        const float ml = __fmul_rn(h2f(*(const uint16_t *)(b + 82)), (float)(sc >> 4));
        const int q = (b[16 + 32 * n + 16 * h + l] >> (2 * j)) & 3;
        return __fsub_rn(__fmul_rn(dl, (float)q), ml);
    } else if (TYPE == MI355_TYPE_Q3_K) {
        // dequantize_row_q3_K, ggml-quants.c:1050-1099
        const uint8_t * b = base + (i / QK_K) * 110;
        const int e = i % QK_K, n = e / 128, j = (e % 128) / 32, h = (e % 32) / 16, l = e % 16;
        const int is = 8 * n + 2 * j + h;
        const int sc = ((b[96 + (is & 7)] >> (4 * (is >> 3))) & 0xF) | (((b[104 + (is & 3)] >> (2 * (is >> 2))) & 3) << 4);
        const float dl = __fmul_rn(h2f(*(const uint16_t *)(b + 108)), (float)(sc - 32));
        const int q = ((b[32 + 32 * n + 16 * h + l] >> (2 * j)) & 3) - ((b[16 * h + l] >> (4 * n + j)) & 1 ? 0 : 4);
        return __fmul_rn(dl, (float)q);
    } else if (TYPE == MI355_TYPE_IQ4_NL) {
        // dequantize_row_iq4_nl, ggml-quants.c:2428-2444
        const uint8_t * b = base + (i / QK) * 18;
        const int j = i % QK;
        const int code = j < 16 ? (b[2 + j] & 0x0F) : (b[2 + j - 16] >> 4);
        return __fmul_rn(h2f(*(const uint16_t *)b), iq4nl_value(code));
    } else if (TYPE == MI355_TYPE_IQ4_XS) {
        // dequantize_row_iq4_xs, ggml-quants.c:2446-2468
        const uint8_t * b = base + (i / QK_K) * 136;
        const int e = i % QK_K, ib = e / 32, j = e % 32;
        const int sh = *(const uint16_t *)(b + 2);
        const int ls = ((b[4 + ib / 2] >> (4 * (ib % 2))) & 0xF) | (((sh >> (2 * ib)) & 3) << 4);
        const float dl = __fmul_rn(h2f(*(const uint16_t *)b), (float)(ls - 32));
        const int code = j < 16 ? (b[8 + 16 * ib + j] & 0x0F) : (b[8 + 16 * ib + j - 16] >> 4);
        return __fmul_rn(dl, iq4nl_value(code));
    } else if (TYPE == MI355_TYPE_F16) {
        return h2f(((const uint16_t *)base)[i]);
    } else if (TYPE == MI355_TYPE_BF16) {
        return __uint_as_float((uint32_t)((const uint16_t *)base)[i] << 16);
    } else if (TYPE == MI355_TYPE_F32) {
        return ((const float *)base)[i];
    }
    return 0.0f;
}
