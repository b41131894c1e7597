// q4k_lab.hip -- ablation lab for the q4_K mat-vec (NOT product code; tuning evidence only).
// Each variant removes one ingredient so its cost shows up as a delta (guide section 7, "Ablate").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../kernels/common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum { F_HDR = 1, F_ACT = 2, F_MATH = 4, F_NT = 8, F_LDSACT = 16 };

// variant: bit flags above; RPW rows per wave; BLK threads per block
template <int FLAGS, int RPW, int BLK>
__global__ void __launch_bounds__(BLK)
lab_kernel(const uint8_t * __restrict__ W, int64_t row_stride, const int8_t * __restrict__ aq, const float * __restrict__ ads,
           float * __restrict__ dst, int m, int k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = BLK / 64;
    if (FLAGS & F_LDSACT) {
        // stage activations: q [k] then ds [k/32*2 floats]
        const int nq = k / 16, nds = k / 32 * 2 / 4;
        for (int i = threadIdx.x; i < nq; i += BLK) ((v4i *)lds)[i] = ((const v4i *)aq)[i];
        for (int i = threadIdx.x; i < nds; i += BLK) ((v4f *)(lds + k))[i] = ((const v4f *)ads)[i];
        __syncthreads();
    }
    const int row0 = (blockIdx.x * WPB + wave) * RPW;
    if (row0 >= m) return;
    const uint8_t * wrow[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) wrow[r] = W + (int64_t)(row0 + r < m ? row0 + r : m - 1) * row_stride;
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) acc[r] = 0.f;
    const int nunits = k / 32, niter = (nunits + 63) >> 6;
    v4i cq[RPW], ch[RPW];
    auto LOAD = [&](v4i * q, v4i * h, int u) {
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const uint8_t * blk = wrow[r] + (int64_t)(u >> 3) * 144;
            q[r] = (FLAGS & F_NT) ? ld_nt16(blk + 16 + (u & 7) * 16) : *(const v4i *)(blk + 16 + (u & 7) * 16);
            if (FLAGS & F_HDR) h[r] = *(const v4i *)blk; else h[r] = v4i{0x3c003c00, 0x01020304, 0x05060708, 0x090a0b0c};
        }
    };
    LOAD(cq, ch, lane < nunits ? lane : nunits - 1);
    for (int it = 0; it < niter; it++) {
        const int u_raw = it * 64 + lane;
        const bool valid = u_raw < nunits;
        const int u = valid ? u_raw : nunits - 1;
        v4i nq_[RPW], nh_[RPW];
        if (it + 1 < niter) LOAD(nq_, nh_, u_raw + 64 < nunits ? u_raw + 64 : nunits - 1);
        const int b = u >> 3, p = u & 7, g = p >> 1;
        const int e0 = b * 256 + g * 64 + (p & 1) * 16;
        v4i alo, ahi; v4f ds;
        if (FLAGS & F_ACT) {
            if (FLAGS & F_LDSACT) {
                alo = *(const v4i *)(lds + e0); ahi = *(const v4i *)(lds + e0 + 32);
                ds = *(const v4f *)(lds + k + (b * 8 + g * 2) * 8);
            } else {
                alo = *(const v4i *)(aq + e0); ahi = *(const v4i *)(aq + e0 + 32);
                ds = *(const v4f *)(ads + (b * 8 + g * 2) * 2);
            }
        } else { alo = v4i{u, 2, 3, 4}; ahi = v4i{5, 6, u, 8}; ds = v4f{1.f, 0.f, 2.f, 0.f}; }
        __builtin_amdgcn_sched_barrier(0);
        int slo = 0, shi = 0;
        if (FLAGS & F_MATH) {
#pragma unroll
            for (int i = 0; i < 4; i++) { slo = dot4(0x01010101, alo[i], slo); shi = dot4(0x01010101, ahi[i], shi); }
        }
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            if (FLAGS & F_MATH) {
                const v4i lo = cq[r] & 0x0f0f0f0f, hi = (cq[r] >> 4) & 0x0f0f0f0f;
                const uint32_t u0 = ch[r].y, u1 = ch[r].z, u2 = ch[r].w;
                const uint32_t sc_lo = u0 & 0x3f3f3f3fu, m_lo = u1 & 0x3f3f3f3fu;
                const uint32_t sc_hi = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
                const uint32_t m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
                const int sh = (g & 1) * 16;
                const uint32_t sc2 = ((g & 2) ? sc_hi : sc_lo) >> sh, m2 = ((g & 2) ? m_hi : m_lo) >> sh;
                const float d = h2f((uint16_t)(ch[r].x & 0xffff)), dmin = h2f((uint16_t)((uint32_t)ch[r].x >> 16));
                int s0 = 0, s1 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { s0 = dot4(lo[i], alo[i], s0); s1 = dot4(hi[i], ahi[i], s1); }
                const float v = ds.x * (d * (float)(sc2 & 0xff) * (float)s0 - dmin * (float)(m2 & 0xff) * (float)slo) +
                                ds.z * (d * (float)((sc2 >> 8) & 0xff) * (float)s1 - dmin * (float)((m2 >> 8) & 0xff) * (float)shi);
                acc[r] += valid ? v : 0.f;
            } else {
                const v4i x = cq[r] ^ ch[r] ^ alo ^ ahi;
                acc[r] += (float)(x.x ^ x.y ^ x.z ^ x.w) * ds.x;
            }
        }
        if (it + 1 < niter) {
#pragma unroll
            for (int r = 0; r < RPW; r++) { cq[r] = nq_[r]; ch[r] = nh_[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const float v = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < m) dst[row0 + r] = v;
    }
}


// v3: activations staged in LDS, header loaded by 8 lanes only and broadcast with ds_bpermute
template <int RPW, int BLK, bool PAD>
__global__ void __launch_bounds__(BLK)
lab3_kernel(const uint8_t * __restrict__ W, int64_t row_stride, const int8_t * __restrict__ aq, const float * __restrict__ ads,
            float * __restrict__ dst, int m, int k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = BLK / 64;
    constexpr int BSTR = PAD ? 256 + 32 : 256;  // LDS stride of one 256-value activation block
    const int nblk = k / 256;
    uint8_t * lds_ds = lds + nblk * BSTR;
    {
        const int nq = k / 16, nds = k / 32 * 2 / 4;
        for (int i = threadIdx.x; i < nq; i += BLK) *(v4i *)(lds + (i >> 4) * BSTR + (i & 15) * 16) = ((const v4i *)aq)[i];
        for (int i = threadIdx.x; i < nds; i += BLK) ((v4f *)lds_ds)[i] = ((const v4f *)ads)[i];
    }
    const int row0 = (blockIdx.x * WPB + wave) * RPW;
    const uint8_t * wrow[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) wrow[r] = W + (int64_t)(row0 + r < m ? row0 + r : m - 1) * row_stride;
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) acc[r] = 0.f;
    const int nunits = k / 32, niter = (nunits + 63) >> 6;
    v4i cq[RPW], ch[RPW];
    auto LOAD = [&](v4i * q, v4i * h, int it) {
        const int u_raw = it * 64 + lane;
        const int u = u_raw < nunits ? u_raw : nunits - 1;
        const int hb_raw = it * 8 + lane;                 // lanes 0..7: header of block it*8 + lane
        const int hb = hb_raw < nblk ? hb_raw : nblk - 1;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            q[r] = *(const v4i *)(wrow[r] + (int64_t)(u >> 3) * 144 + 16 + (u & 7) * 16);
            h[r] = v4i{0, 0, 0, 0};
            if (lane < 8) h[r] = *(const v4i *)(wrow[r] + (int64_t)hb * 144);
        }
    };
    LOAD(cq, ch, 0);
    __syncthreads();
    if (row0 >= m) return;
    const int bsel = (lane >> 3) << 2;
    for (int it = 0; it < niter; it++) {
        const int u_raw = it * 64 + lane;
        const bool valid = u_raw < nunits;
        const int u = valid ? u_raw : nunits - 1;
        v4i nq_[RPW], nh_[RPW];
        if (it + 1 < niter) LOAD(nq_, nh_, it + 1);
        const int b = u >> 3, p = u & 7, g = p >> 1;
        const int eo = b * BSTR + g * 64 + (p & 1) * 16;
        const v4i alo = *(const v4i *)(lds + eo), ahi = *(const v4i *)(lds + eo + 32);
        const v4f ds = *(const v4f *)(lds_ds + (b * 8 + g * 2) * 8);
        __builtin_amdgcn_sched_barrier(0);
        int slo = 0, shi = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { slo = dot4(0x01010101, alo[i], slo); shi = dot4(0x01010101, ahi[i], shi); }
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            v4i hdr;
#pragma unroll
            for (int i = 0; i < 4; i++) hdr[i] = __builtin_amdgcn_ds_bpermute(bsel, ch[r][i]);
            const v4i lo = cq[r] & 0x0f0f0f0f, hi = (cq[r] >> 4) & 0x0f0f0f0f;
            const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;
            const uint32_t sc_lo = u0 & 0x3f3f3f3fu, m_lo = u1 & 0x3f3f3f3fu;
            const uint32_t sc_hi = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
            const uint32_t m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
            const int sh = (g & 1) * 16;
            const uint32_t sc2 = ((g & 2) ? sc_hi : sc_lo) >> sh, m2 = ((g & 2) ? m_hi : m_lo) >> sh;
            const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)((uint32_t)hdr.x >> 16));
            int s0 = 0, s1 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { s0 = dot4(lo[i], alo[i], s0); s1 = dot4(hi[i], ahi[i], s1); }
            const float v = ds.x * (d * (float)(sc2 & 0xff) * (float)s0 - dmin * (float)(m2 & 0xff) * (float)slo) +
                            ds.z * (d * (float)((sc2 >> 8) & 0xff) * (float)s1 - dmin * (float)((m2 >> 8) & 0xff) * (float)shi);
            acc[r] += valid ? v : 0.f;
        }
        if (it + 1 < niter) {
#pragma unroll
            for (int r = 0; r < RPW; r++) { cq[r] = nq_[r]; ch[r] = nh_[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const float v = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < m) dst[row0 + r] = v;
    }
}

template <int RPW, int BLK, bool PAD>
static void launch3(const uint8_t * W, int64_t rs, const int8_t * aq, const float * ads, float * dst, int m, int k, hipStream_t st) {
    constexpr int WPB = BLK / 64;
    const int grid = (m + WPB * RPW - 1) / (WPB * RPW);
    const size_t sh = (size_t)k / 256 * (PAD ? 288 : 256) + k / 32 * 8;
    lab3_kernel<RPW, BLK, PAD><<<grid, BLK, sh, st>>>(W, rs, aq, ads, dst, m, k);
}


// v4: 32 B of qs per lane (one 64-value group = sub-blocks 2g, 2g+1 complete): 4 lanes per block,
// scales decoded once per 64 values, activation sums come from act_ds (no in-lane ones-dot).
template <int RPW, int BLK>
__global__ void __launch_bounds__(BLK)
lab4_kernel(const uint8_t * __restrict__ W, int64_t row_stride, const int8_t * __restrict__ aq, const float * __restrict__ ads,
            float * __restrict__ dst, int m, int k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = BLK / 64;
    uint8_t * lds_ds = lds + k;
    {
        const int nq = k / 16, nds = k / 32 * 2 / 4;
        for (int i = threadIdx.x; i < nq; i += BLK) ((v4i *)lds)[i] = ((const v4i *)aq)[i];
        for (int i = threadIdx.x; i < nds; i += BLK) ((v4f *)lds_ds)[i] = ((const v4f *)ads)[i];
    }
    const int row0 = (blockIdx.x * WPB + wave) * RPW;
    const uint8_t * wrow[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) wrow[r] = W + (int64_t)(row0 + r < m ? row0 + r : m - 1) * row_stride;
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) acc[r] = 0.f;
    const int nunits = k / 64, niter = (nunits + 63) >> 6;
    v4i cq0[RPW], cq1[RPW], ch[RPW];
    auto LOAD = [&](v4i * q0, v4i * q1, v4i * h, int it) {
        const int u_raw = it * 64 + lane;
        const int u = u_raw < nunits ? u_raw : nunits - 1;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const uint8_t * blk = wrow[r] + (int64_t)(u >> 2) * 144;
            q0[r] = *(const v4i *)(blk + 16 + (u & 3) * 32);
            q1[r] = *(const v4i *)(blk + 32 + (u & 3) * 32);
            h[r]  = *(const v4i *)blk;
        }
    };
    LOAD(cq0, cq1, ch, 0);
    __syncthreads();
    if (row0 >= m) return;
    for (int it = 0; it < niter; it++) {
        const int u_raw = it * 64 + lane;
        const bool valid = u_raw < nunits;
        const int u = valid ? u_raw : nunits - 1;
        v4i nq0[RPW], nq1[RPW], nh[RPW];
        if (it + 1 < niter) LOAD(nq0, nq1, nh, it + 1);
        const int g = u & 3;
        const v4i a0 = *(const v4i *)(lds + u * 64), a1 = *(const v4i *)(lds + u * 64 + 16);
        const v4i a2 = *(const v4i *)(lds + u * 64 + 32), a3 = *(const v4i *)(lds + u * 64 + 48);
        const v4f ds = *(const v4f *)(lds_ds + u * 16); // {d_lo, sum_lo, d_hi, sum_hi}
        __builtin_amdgcn_sched_barrier(0);
        const float ms_lo = ds.x * ds.y, ms_hi = ds.z * ds.w;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const v4i hdr = ch[r];
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
            int s0 = 0, s1 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s0 = dot4(cq0[r][i] & 0x0f0f0f0f, a0[i], s0);
                s0 = dot4(cq1[r][i] & 0x0f0f0f0f, a1[i], s0);
                s1 = dot4((cq0[r][i] >> 4) & 0x0f0f0f0f, a2[i], s1);
                s1 = dot4((cq1[r][i] >> 4) & 0x0f0f0f0f, a3[i], s1);
            }
            const float x = (float)(s0 * (int)(sc2 & 0xff)) * ds.x + (float)(s1 * (int)(sc2 >> 8)) * ds.z;
            const float y = (float)(m2 & 0xff) * ms_lo + (float)(m2 >> 8) * ms_hi;
            const float v = d * x - dmin * y;
            acc[r] += valid ? v : 0.f;
        }
        if (it + 1 < niter) {
#pragma unroll
            for (int r = 0; r < RPW; r++) { cq0[r] = nq0[r]; cq1[r] = nq1[r]; ch[r] = nh[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const float v = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < m) dst[row0 + r] = v;
    }
}

template <int RPW, int BLK>
static void launch4(const uint8_t * W, int64_t rs, const int8_t * aq, const float * ads, float * dst, int m, int k, hipStream_t st) {
    constexpr int WPB = BLK / 64;
    const int grid = (m + WPB * RPW - 1) / (WPB * RPW);
    lab4_kernel<RPW, BLK><<<grid, BLK, (size_t)k + k / 32 * 8, st>>>(W, rs, aq, ads, dst, m, k);
}


// v5: persistent waves.  Each wave walks rows w, w+NW, ... and prefetches the next (row, it) step
// into registers while computing the current one; activations staged once per workgroup.
template <int BLK>
__global__ void __launch_bounds__(BLK)
lab5_kernel(const uint8_t * __restrict__ W, int64_t row_stride, const int8_t * __restrict__ aq, const float * __restrict__ ads,
            float * __restrict__ dst, int m, int k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = BLK / 64;
    uint8_t * lds_ds = lds + k;
    const int NW = gridDim.x * WPB;
    const int w  = blockIdx.x * WPB + wave;
    const int nunits = k / 64, niter = (nunits + 63) >> 6;
    const int nrows_w = w < m ? (m - w + NW - 1) / NW : 0;
    const int total = nrows_w * niter;
    v4i cq0, cq1, ch;
    auto LOAD = [&](v4i & q0, v4i & q1, v4i & h, int row, int it) {
        const int u_raw = it * 64 + lane;
        const int u = u_raw < nunits ? u_raw : nunits - 1;
        const uint8_t * blk = W + (int64_t)row * row_stride + (int64_t)(u >> 2) * 144;
        q0 = *(const v4i *)(blk + 16 + (u & 3) * 32);
        q1 = *(const v4i *)(blk + 32 + (u & 3) * 32);
        h  = *(const v4i *)blk;
    };
    if (total > 0) LOAD(cq0, cq1, ch, w, 0);
    {
        const int nq = k / 16, nds = k / 32 * 2 / 4;
        for (int i = threadIdx.x; i < nq; i += BLK) ((v4i *)lds)[i] = ((const v4i *)aq)[i];
        for (int i = threadIdx.x; i < nds; i += BLK) ((v4f *)lds_ds)[i] = ((const v4f *)ads)[i];
    }
    __syncthreads();
    int row = w, it = 0;
    float acc = 0.f;
    for (int j = 0; j < total; j++) {
        int nrow = row, nit = it + 1;
        if (nit == niter) { nit = 0; nrow = row + NW; }
        v4i nq0, nq1, nh;
        if (j + 1 < total) LOAD(nq0, nq1, nh, nrow, nit);
        const int u_raw = it * 64 + lane;
        const bool valid = u_raw < nunits;
        const int u = valid ? u_raw : nunits - 1;
        const int g = u & 3;
        const v4i a0 = *(const v4i *)(lds + u * 64), a1 = *(const v4i *)(lds + u * 64 + 16);
        const v4i a2 = *(const v4i *)(lds + u * 64 + 32), a3 = *(const v4i *)(lds + u * 64 + 48);
        const v4f ds = *(const v4f *)(lds_ds + u * 16);
        __builtin_amdgcn_sched_barrier(0);
        {
            const float ms_lo = ds.x * ds.y, ms_hi = ds.z * ds.w;
            const uint32_t u0 = ch.y, u1 = ch.z, u2 = ch.w;
            const int sh = (g & 1) * 16;
            uint32_t sc2, m2;
            if (g & 2) {
                sc2 = ((u2 >> sh) & 0x0f0fu) | (((u0 >> (sh + 6)) & 0x0303u) << 4);
                m2  = ((u2 >> (sh + 4)) & 0x0f0fu) | (((u1 >> (sh + 6)) & 0x0303u) << 4);
            } else {
                sc2 = (u0 >> sh) & 0x3f3fu;
                m2  = (u1 >> sh) & 0x3f3fu;
            }
            const float d = h2f((uint16_t)(ch.x & 0xffff)), dmin = h2f((uint16_t)((uint32_t)ch.x >> 16));
            int s0 = 0, s1 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s0 = dot4(cq0[i] & 0x0f0f0f0f, a0[i], s0);
                s0 = dot4(cq1[i] & 0x0f0f0f0f, a1[i], s0);
                s1 = dot4((cq0[i] >> 4) & 0x0f0f0f0f, a2[i], s1);
                s1 = dot4((cq1[i] >> 4) & 0x0f0f0f0f, a3[i], s1);
            }
            const float x = (float)(s0 * (int)(sc2 & 0xff)) * ds.x + (float)(s1 * (int)(sc2 >> 8)) * ds.z;
            const float y = (float)(m2 & 0xff) * ms_lo + (float)(m2 >> 8) * ms_hi;
            const float v = d * x - dmin * y;
            acc += valid ? v : 0.f;
        }
        if (it == niter - 1) {
            const float v = wave_sum(acc);
            if (lane == 0) dst[row] = v;
            acc = 0.f;
        }
        cq0 = nq0; cq1 = nq1; ch = nh;
        row = nrow; it = nit;
    }
}

template <int BLK, int WGPC>
static void launch5(const uint8_t * W, int64_t rs, const int8_t * aq, const float * ads, float * dst, int m, int k, hipStream_t st) {
    constexpr int WPB = BLK / 64;
    int grid = 256 * WGPC;
    const int need = (m + WPB - 1) / WPB;
    if (grid > need) grid = need;
    lab5_kernel<BLK><<<grid, BLK, (size_t)k + k / 32 * 8, st>>>(W, rs, aq, ads, dst, m, k);
}

// ceiling: contiguous streaming read of the same bytes, 16 B per lane, grid-stride
__global__ void __launch_bounds__(256) stream_kernel(const v4i * __restrict__ p, int64_t n16, float * dst) {
    v4i acc = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const v4i a = ld_nt16(p + i), b = ld_nt16(p + i + stride), c = ld_nt16(p + i + 2 * stride), d = ld_nt16(p + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= ld_nt16(p + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) dst[0] = 1.f;
}

struct Variant { const char * name; void (*launch)(const uint8_t *, int64_t, const int8_t *, const float *, float *, int, int, hipStream_t); };

template <int FLAGS, int RPW, int BLK>
static void launch(const uint8_t * W, int64_t rs, const int8_t * aq, const float * ads, float * dst, int m, int k, hipStream_t st) {
    constexpr int WPB = BLK / 64;
    const int grid = (m + WPB * RPW - 1) / (WPB * RPW);
    const size_t sh = (FLAGS & F_LDSACT) ? (size_t)k + k / 32 * 8 : 0;
    lab_kernel<FLAGS, RPW, BLK><<<grid, BLK, sh, st>>>(W, rs, aq, ads, dst, m, k);
}

int main(int argc, char ** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 28672, k = argc > 2 ? atoi(argv[2]) : 8192;
    const int64_t rs = (int64_t)k / 256 * 144;
    const size_t wbytes = (size_t)rs * m, wstride = (wbytes + 4095) / 4096 * 4096;
    const size_t pool_bytes = (size_t)3 << 30;
    const int nmat = (int)(pool_bytes / wstride);
    uint8_t * pool; CK(hipMalloc(&pool, pool_bytes));
    { std::vector<uint32_t> h(pool_bytes / 4); uint32_t s = 1; for (auto & x : h) { s = s * 1664525u + 1013904223u; x = (s >> 3) & 0x3bff3bffu; }
      CK(hipMemcpy(pool, h.data(), pool_bytes, hipMemcpyHostToDevice)); }
    int8_t * aq; float * ads; float * dst;
    CK(hipMalloc(&aq, k)); CK(hipMalloc(&ads, k / 32 * 8)); CK(hipMalloc(&dst, (size_t)m * 4));
    CK(hipMemset(aq, 1, k)); CK(hipMemset(ads, 0, k / 32 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int ALL = F_HDR | F_ACT | F_MATH | F_NT;
    const Variant vs[] = {
        {"full rpw2 b256", launch<ALL, 2, 256>},
        {"no-math no-act no-hdr rpw2", launch<F_NT, 2, 256>},
        {"v5 b256 x4", launch5<256, 4>},
        {"v5 b256 x6", launch5<256, 6>},
        {"v5 b256 x8", launch5<256, 8>},
        {"v5 b512 x2", launch5<512, 2>},
        {"v5 b512 x3", launch5<512, 3>},
        {"v5 b512 x4", launch5<512, 4>},
        {"v5 b1024 x1", launch5<1024, 1>},
        {"v5 b1024 x2", launch5<1024, 2>},
        {"v4 rpw1 b256", launch4<1, 256>},
        {"v4 rpw2 b256", launch4<2, 256>},
        {"ldsact rpw2 b256", launch<ALL | F_LDSACT, 2, 256>},
    };
    printf("q4_K lab m=%d k=%d  (%.1f MB weights, %d matrices in pool)\n", m, k, wbytes / 1e6, nmat);
    const int iters = 100;
    for (int rep = 0; rep < 2; rep++) {
        for (const Variant & v : vs) {
            for (int w = 0; w < 2; w++) v.launch(pool + (size_t)(w % nmat) * wstride, rs, aq, ads, dst, m, k, st);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) v.launch(pool + (size_t)((i + 2) % nmat) * wstride, rs, aq, ads, dst, m, k, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("  %-32s %8.2f us  %7.1f GB/s\n", v.name, us, wbytes / us * 1e-3);
        }
        {
            for (int g : {4096}) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; i++) stream_kernel<<<g, 256, 0, st>>>((const v4i *)(pool + (size_t)((i + 2) % nmat) * wstride), wbytes / 16, dst);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                printf("  stream grid %-20d %8.2f us  %7.1f GB/s\n", g, us, wbytes / us * 1e-3);
            }
        }
    }
    return 0;
}
