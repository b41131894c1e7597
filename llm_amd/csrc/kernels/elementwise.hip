// elementwise.hip -- the non-matmul ops of a Llama / Mixtral decode graph.
// All of them are HBM/launch-bound streaming passes: one lane per element (or per row for the
// reductions), coalesced along ne[0], 128-bit accesses where the layout allows.
// Behavioural spec = the reference's ggml-cpu implementations cited per entry point in
// include/mi355_kernels.h; nothing here is derived from ggml-cuda.
#include "common.h"
#include "deq_elem.h"
#include "rope.h"

namespace {

struct T4 {
    char *  data;
    int64_t ne[4];
    int64_t nb[4];
};

static T4 t4(const mi355_tensor * t) {
    T4 r;
    r.data = (char *)t->data;
    for (int i = 0; i < 4; i++) { r.ne[i] = t->ne[i]; r.nb[i] = t->nb[i]; }
    return r;
}

static int64_t nelem(const mi355_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
static int64_t nrows(const mi355_tensor * t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
static bool same_shape(const mi355_tensor * a, const mi355_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
static unsigned grid_for(int64_t n, int block = 256) {
    const int64_t g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > 262144 ? 262144 : g));
}

static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sum over the 256 threads of a block (4 waves)
static __device__ __forceinline__ float block_sum(float v, float * sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
static __device__ __forceinline__ float block_max(float v, float * sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ---------------------------------------------------------------- binary with broadcast
template <int OP>
__global__ void __launch_bounds__(256)
binary_kernel(const T4 a, const T4 b, const T4 d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0];
        const int64_t i1 = r % d.ne[1], r2 = r / d.ne[1];
        const int64_t i2 = r2 % d.ne[2], i3 = r2 / d.ne[2];
        const float x = *(const float *)(a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
        const float y = *(const float *)(b.data + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] +
                                         (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3]);
        float v;
        if (OP == MI355_ADD) v = x + y; else if (OP == MI355_MUL) v = x * y; else if (OP == MI355_SUB) v = x - y; else v = x / y;
        *(float *)(d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = v;
    }
}

// q8_K planes (quantize_row_q8_K_ref, ggml-quants.c:2471-2508; the arithmetic of quant_q8k_kernel in mmsk.hip) of 16 consecutive
// values per lane: the 16 lanes of a DPP row own one 256-value block (amax and the first element that attains it are reduced
// over the row), a lane's 16 values are exactly one bsums entry.  All 16 lanes of a row must be active.
static __device__ __forceinline__ void quant16_q8k(const float (&o)[16], int8_t * q16, float * d_blk, int16_t * bs1, int lane16) {
    float ax = 0.0f, mine = 0.0f;
    int idx = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const float t = fabsf(o[j]); if (t > ax) { ax = t; idx = j; mine = o[j]; } }
    float amax = ax;
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 16));
    int first = ax == amax ? lane16 * 16 + idx : 1024;          // first occurrence on ties (the reference scans with a strict >)
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) first = min(first, __shfl_xor(first, m, 16));
    const float maxv = __shfl(mine, (first >> 4) & 15, 16);
    int packed[4] = {0, 0, 0, 0}, sum = 0;
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = __fdiv_rn(-127.0f, maxv);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float val = __fadd_rn(__fmul_rn(iscale, o[j]), 12582912.0f);      // nearest_int, ggml-quants.c:559-565
            int qi = (__float_as_int(val) & 0x007fffff) - 0x00400000;
            qi = qi < 127 ? qi : 127;
            sum += qi;
            packed[j >> 2] |= (qi & 0xff) << (8 * (j & 3));
        }
        d = __fdiv_rn(1.0f, iscale);
    }
    *(v4i *)q16 = v4i{packed[0], packed[1], packed[2], packed[3]};
    *bs1 = (int16_t)sum;
    if (lane16 == 0) *d_blk = d;
}

// ---------------------------------------------------------------- rms norm (one block per row)
// y = x * rsqrt(mean(x^2) + eps) [* w]: the row is read once with 128-bit loads and kept in registers
// (rows up to 8192 floats); rows that are longer or not 16-byte aligned take the two-pass path.
// HAS_ADD: the row is first formed as x + y and that sum is also stored (residual add feeding the next norm).
// NT threads: 256, or 512 for rows of more than 4096 values (one 16-value chunk per thread up to 8192: with 256 threads
// an 8192-wide row -- Llama-3-70B -- took 8.4 us against 5.0 for 4096)
template <bool HAS_W, bool HAS_ADD, int NT>
__global__ void __launch_bounds__(NT)
rms_norm_kernel(const T4 x, const T4 y, const T4 sum, const float * __restrict__ w, const T4 d, float eps, bool vec,
                int8_t * __restrict__ aq, float * __restrict__ ads, uint16_t * __restrict__ abf,
                const float * __restrict__ xparts, int kz, int16_t * __restrict__ kbs) {
    // kbs: the quantized form is q8_K planes (aq = int8 [rows][n], ads = d [rows][n / 256], kbs = bsums [rows][n / 16]) for the K-quant
    // skinny kernels of 4..32-column batches, instead of q8_1 (n % 256 == 0)
    // xparts: x is not in memory yet -- it is the sum of kz planes [kz][rows][n] (the K slices of the GEMM that produces it,
    // mi355_mul_mat_dq_parts), added here in slice order as kz_sum_kernel would have (16-value-chunk path only; x.data is not read)
    __shared__ float sh[NT / 64];
    const int row = blockIdx.x;
    const int ne1 = (int)x.ne[1], ne2 = (int)x.ne[2];
    const int i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * xr = (const float *)(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float * dr = (float *)(d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int n = (int)x.ne[0];
    if (vec && (n & 15) == 0) {
        // a thread owns chunks of 16 consecutive floats (chunk c = tid + 256 j): half a q8_1 block, so the fused
        // quantization needs one lane-pair exchange per block instead of an 8-lane tree per 4 values
        constexpr int MAXC = 8192 / 16 / NT > 0 ? 8192 / 16 / NT : 1;      // n <= 8192 (NT 256: 2 chunks, 512: 1)
        const int nc = n >> 4;
        v4f v[MAXC][4], wv[MAXC][4];
        float s = 0.0f;
        const float * yr = HAS_ADD ? (const float *)(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]) : nullptr;
        float * sr = HAS_ADD ? (float *)(sum.data + i1 * sum.nb[1] + i2 * sum.nb[2] + i3 * sum.nb[3]) : nullptr;
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            const int c = threadIdx.x + j * NT;
            if (c < nc) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (xparts) {
                        const v4f * pr = (const v4f *)(xparts + (int64_t)row * n) + 4 * c + q;
                        const int64_t plane4 = (int64_t)gridDim.x * (n >> 2);
                        v[j][q] = pr[0];
                        for (int z = 1; z < kz; z++) v[j][q] = v[j][q] + pr[z * plane4];
                    } else {
                        v[j][q] = ((const v4f *)xr)[4 * c + q];
                    }
                    if (HAS_W) wv[j][q] = ((const v4f *)w)[4 * c + q];   // requested now: its latency hides behind the reduction
                    if (HAS_ADD) {
                        v[j][q] = v[j][q] + ((const v4f *)yr)[4 * c + q];
                        ((v4f *)sr)[4 * c + q] = v[j][q];
                    }
                    s += v[j][q].x * v[j][q].x + v[j][q].y * v[j][q].y + v[j][q].z * v[j][q].z + v[j][q].w * v[j][q].w;
                }
            }
        }
        s = block_sum_n<NT / 64>(s, sh);
        const float scale = 1.0f / sqrtf(s / (float)n + eps);
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            const int c = threadIdx.x + j * NT;
            if (c < nc) {
                float o[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v4f t = v[j][q] * scale;            // (x * scale) then * w: the op order of the unfused graph
                    if (HAS_W) t = t * wv[j][q];
                    ((v4f *)dr)[4 * c + q] = t;
                    o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
                }
                if (aq && kbs) quant16_q8k(o, aq + (int64_t)row * n + 16 * c, ads + (int64_t)row * (n >> 8) + (c >> 4), kbs + (int64_t)row * (n >> 4) + c, c & 15);
                else if (aq) quant16_soa(o, aq + (int64_t)row * n + 16 * c, ads + ((int64_t)row * (n / QK) + (c >> 1)) * 2, (c & 1) == 0);
                if (abf) {      // bf16 form of the row for the prefill GEMM (the rounding of mi355_act_bf16)
                    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
                    bf8 b0, b1;
#pragma unroll
                    for (int i = 0; i < 8; i++) { b0[i] = (__bf16)o[i]; b1[i] = (__bf16)o[8 + i]; }
                    bf8 * bo = (bf8 *)(abf + (int64_t)row * n + 16 * c);
                    bo[0] = b0; bo[1] = b1;
                }
            }
        }
        return;
    }
    if (vec) {
        constexpr int MAXV = 8;
        const int n4 = n >> 2;
        v4f v[MAXV], wv[MAXV];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXV; j++) {
            const int i = threadIdx.x + j * NT;
            if (i < n4) {
                v[j] = ((const v4f *)xr)[i];
                if (HAS_W) wv[j] = ((const v4f *)w)[i];
                if (HAS_ADD) {
                    const float * yr = (const float *)(y.data + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
                    v[j] = v[j] + ((const v4f *)yr)[i];
                    float * sr = (float *)(sum.data + i1 * sum.nb[1] + i2 * sum.nb[2] + i3 * sum.nb[3]);
                    ((v4f *)sr)[i] = v[j];
                }
                s += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
            }
        }
        s = block_sum_n<NT / 64>(s, sh);
        const float scale = 1.0f / sqrtf(s / (float)n + eps);
#pragma unroll
        for (int j = 0; j < MAXV; j++) {
            const int i = threadIdx.x + j * NT;
            if (i < n4) {
                v4f o = v[j] * scale;
                if (HAS_W) o = o * wv[j];
                ((v4f *)dr)[i] = o;
            }
        }
        return;
    }
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += NT) { const float t = xr[i]; s += t * t; }
    s = block_sum_n<NT / 64>(s, sh);
    const float scale = 1.0f / sqrtf(s / (float)n + eps);
    for (int i = threadIdx.x; i < n; i += NT) { float o = xr[i] * scale; if (HAS_W) o *= w[i]; dr[i] = o; }
}

// Prefill form of "K-slice planes + residual -> sum, RMS norm * weight, bf16 rows" for rows of at most 4096 values: a workgroup takes
// TWO rows and requests both before it touches the first, so the second row's loads are in flight while the first is reduced and
// stored (with one row per workgroup the whole launch loads, then reduces, then stores in step: 60 MB in 20 us).  Thread c owns
// chunk c of 16 values, the per-thread partial sums and the wave / block reduction are those of rms_norm_kernel: the same bits.
template <int KZ, int NT>
__global__ void __launch_bounds__(NT)
add_rms_norm_planes_kernel(const float * __restrict__ parts, int64_t plane, const float * y, int64_t ys, float * sum, int64_t ss,
                           const float * __restrict__ w, float * d, int64_t ds, uint16_t * __restrict__ abf, int n, int rows, float eps) {
    // (y, sum and d may be the same memory -- an in-place residual add, a norm result on the dead residual -- hence no __restrict__:
    // every load of a row precedes every store to it in program order, per thread on the same indices)
    __shared__ float sh[NT / 64];
    const int c = threadIdx.x, nc = n >> 4;       // NT = 256 for rows of up to 4096 values, 512 up to 8192: rms_norm_kernel's own choice
    const bool on = c < nc;
    const int r0 = blockIdx.x, r1 = blockIdx.x + gridDim.x;
    v4f P[2][KZ][4], Y[2][4], W[4];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int r = u ? r1 : r0;
        if (on && r < rows) {
#pragma unroll
            for (int z = 0; z < KZ; z++)
#pragma unroll
                for (int q = 0; q < 4; q++) P[u][z][q] = ((const v4f *)(parts + z * plane + (int64_t)r * n))[4 * c + q];
#pragma unroll
            for (int q = 0; q < 4; q++) Y[u][q] = ((const v4f *)(y + (int64_t)r * ys))[4 * c + q];
        }
        if (u == 0 && on) {
#pragma unroll
            for (int q = 0; q < 4; q++) W[q] = ((const v4f *)w)[4 * c + q];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int r = u ? r1 : r0;
        if (r >= rows) break;                      // uniform over the workgroup
        v4f v[4];
        float s = 0.0f;
        if (on) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v[q] = P[u][0][q];
#pragma unroll
                for (int z = 1; z < KZ; z++) v[q] = v[q] + P[u][z][q];
                v[q] = v[q] + Y[u][q];
                ((v4f *)(sum + (int64_t)r * ss))[4 * c + q] = v[q];
                s += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
            }
        }
        s = block_sum_n<NT / 64>(s, sh);
        const float scale = 1.0f / sqrtf(s / (float)n + eps);
        if (on) {
            float o[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v4f t = v[q] * scale;
                t = t * W[q];
                ((v4f *)(d + (int64_t)r * ds))[4 * c + q] = t;
                o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
            }
            typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
            bf8 b0, b1;
#pragma unroll
            for (int i = 0; i < 8; i++) { b0[i] = (__bf16)o[i]; b1[i] = (__bf16)o[8 + i]; }
            bf8 * bo = (bf8 *)(abf + (int64_t)r * n + 16 * c);
            bo[0] = b0; bo[1] = b1;
        }
    }
}

// ---------------------------------------------------------------- rope
// one lane per rotated pair; rows = (head i1, token i2, batch i3)
__global__ void __launch_bounds__(256)
rope_kernel(const T4 x, const int32_t * __restrict__ pos, const float * __restrict__ ff, const T4 d, const RopeP p, int64_t npairs) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int half = p.ne0 / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) {
        const int ic = (int)(i % half);          // pair index inside the row
        const int64_t r = i / half;
        const int64_t i1 = r % x.ne[1], r2 = r / x.ne[1];
        const int64_t i2 = r2 % x.ne[2], i3 = r2 / x.ne[2];
        const char * xs = x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
        char * ds = d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
        const int i0 = 2 * ic;
        if (i0 >= p.n_dims) { // dims beyond n_dims are copied through (ops.cpp: tail loops)
            ((float *)ds)[i0]     = ((const float *)xs)[i0];
            ((float *)ds)[i0 + 1] = ((const float *)xs)[i0 + 1];
            continue;
        }
        // theta = pos * theta_scale^(i0/2) (the CPU builds it by repeated multiplication; powf agrees to ~1 ulp
        // of the angle, inside the op's 1e-7 NMSE budget for the positions llama.cpp uses)
        float c, s;
        rope_cos_sin(p, pos[i2], ff, ic, c, s);
        if (p.mode & 2) { // NEOX: pairs (ic, ic + n_dims/2)
            const float x0 = ((const float *)xs)[ic], x1 = ((const float *)xs)[ic + p.n_dims / 2];
            ((float *)ds)[ic]                = x0 * c - x1 * s;
            ((float *)ds)[ic + p.n_dims / 2] = x0 * s + x1 * c;
        } else {          // NORMAL: adjacent pairs
            const float x0 = ((const float *)xs)[i0], x1 = ((const float *)xs)[i0 + 1];
            ((float *)ds)[i0]     = x0 * c - x1 * s;
            ((float *)ds)[i0 + 1] = x0 * s + x1 * c;
        }
    }
}

// prefill form: one lane per (rotated pair, token) that walks the heads, so cos / sin (a powf and a sincosf) are computed once per
// (pair, token) instead of once per head -- the arithmetic per element is unchanged (bit-identical to rope_kernel)
__global__ void __launch_bounds__(256)
rope_heads_kernel(const T4 x, const int32_t * __restrict__ pos, const float * __restrict__ ff, const T4 d, const RopeP p, int64_t npt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int half = p.ne0 / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npt; i += stride) {
        const int ic = (int)(i % half);          // pair index inside the row
        const int64_t r = i / half;
        const int64_t i2 = r % x.ne[2], i3 = r / x.ne[2];
        const int i0 = 2 * ic;
        const bool pass = i0 >= p.n_dims;        // dims beyond n_dims are copied through (ops.cpp: tail loops)
        float c = 1.0f, s = 0.0f;
        if (!pass) rope_cos_sin(p, pos[i2], ff, ic, c, s);
        const int e0 = pass ? i0 : (p.mode & 2) ? ic : i0, e1 = pass ? i0 + 1 : (p.mode & 2) ? ic + p.n_dims / 2 : i0 + 1;
        // 8 heads at a time, all loads before the first store (x and d may be the same memory, which keeps the compiler from
        // overlapping the iterations on its own; a lane only ever touches its own two elements of each head)
        for (int64_t h0 = 0; h0 < x.ne[1]; h0 += 8) {
            float x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t i1 = h0 + u < x.ne[1] ? h0 + u : x.ne[1] - 1;
                const float * xs = (const float *)(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
                x0[u] = xs[e0]; x1[u] = xs[e1];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (h0 + u >= x.ne[1]) break;
                float * ds = (float *)(d.data + (h0 + u) * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
                if (pass) { ds[e0] = x0[u]; ds[e1] = x1[u]; }
                else      { ds[e0] = x0[u] * c - x1[u] * s; ds[e1] = x0[u] * s + x1[u] * c; }
            }
        }
    }
}

// prefill Q / K / V epilogue: the K-slice planes of the three GEMMs (mi355_mul_mat_dq_parts) -> q rotated (f32), k rotated and v into
// the KV-cache rows (f16 / f32).  One lane per (rotated pair, token, group of 8 heads) for q and k, one per 4 values for v.  Per
// element: the slice sum of kz_sum_kernel (mmdq.hip), the rotation of rope_kernel, the conversion of cpy_kernel -- the same bits.
struct QkvFin {
    const float * qp, * kp, * vp;      // planes [kz][n][m]
    int kzq, kzk, kzv, mq, mk, mv, n;
    float * q_out; char * k_out, * v_out; int k_f16, v_f16;
    const int32_t * pos; const float * ff; RopeP p;
};

static __device__ __forceinline__ float plane_sum(const float * p, int64_t i, int64_t plane, int kz) {
    float v = p[i];
    for (int z = 1; z < kz; z++) v += p[z * plane + i];
    return v;
}

__global__ void __launch_bounds__(256)
qkv_finish_kernel(const QkvFin a) {
    const int half = a.p.ne0 / 2, hd = a.p.ne0;
    const int nhq = a.mq / hd, nhk = a.mk / hd, gq = (nhq + 7) / 8, gk = (nhk + 7) / 8;
    const int64_t nrope = (int64_t)half * a.n * (gq + gk), nv = (int64_t)a.n * (a.mv / 4);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrope + nv; i += stride) {
        if (i >= nrope) {                              // v: 4 values
            const int64_t j = (i - nrope) * 4;
            v4f v = *(const v4f *)(a.vp + j);
            for (int z = 1; z < a.kzv; z++) v = v + *(const v4f *)(a.vp + (int64_t)z * a.n * a.mv + j);
            if (a.v_f16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};      // round to nearest even, as __float2half_rn
                *(h4 *)(a.v_out + j * 2) = hv;
            } else {
                *(v4f *)(a.v_out + j * 4) = v;
            }
            continue;
        }
        const int ic = (int)(i % half);
        const int64_t r = i / half;
        const int tok = (int)(r % a.n), g = (int)(r / a.n);
        const bool is_k = g >= gq;
        const int h0 = (is_k ? g - gq : g) * 8, nh = is_k ? nhk : nhq, m = is_k ? a.mk : a.mq, kz = is_k ? a.kzk : a.kzq;
        const float * part = is_k ? a.kp : a.qp;
        const int i0 = 2 * ic;
        const bool pass = i0 >= a.p.n_dims;
        float c = 1.0f, s = 0.0f;
        if (!pass) rope_cos_sin(a.p, a.pos[tok], a.ff, ic, c, s);
        const int e0 = pass ? i0 : (a.p.mode & 2) ? ic : i0, e1 = pass ? i0 + 1 : (a.p.mode & 2) ? ic + a.p.n_dims / 2 : i0 + 1;
        const int64_t plane = (int64_t)a.n * m, row = (int64_t)tok * m;
        float x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int h = h0 + u < nh ? h0 + u : nh - 1;
            x0[u] = plane_sum(part, row + h * hd + e0, plane, kz);
            x1[u] = plane_sum(part, row + h * hd + e1, plane, kz);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (h0 + u >= nh) break;
            const float y0 = pass ? x0[u] : x0[u] * c - x1[u] * s, y1 = pass ? x1[u] : x0[u] * s + x1[u] * c;
            const int64_t o = row + (h0 + u) * hd;
            if (!is_k)        { a.q_out[o + e0] = y0; a.q_out[o + e1] = y1; }
            else if (a.k_f16) { ((__half *)a.k_out)[o + e0] = __float2half_rn(y0); ((__half *)a.k_out)[o + e1] = __float2half_rn(y1); }
            else              { ((float *)a.k_out)[o + e0] = y0; ((float *)a.k_out)[o + e1] = y1; }
        }
    }
}

// the same for fully rotated heads (n_dims == head_dim, head_dim % 8 == 0; every llama-family model): a lane owns FOUR rotated pairs
// of one token in 4 heads, so every load is a 16-byte one (q / k planes) and k leaves as 8-byte f16 stores.  Same arithmetic per element.
typedef _Float16 qf_h4 __attribute__((ext_vector_type(4)));

// v: 4 values at flat index j
static __device__ __forceinline__ void qkv_finish_v4(const QkvFin & a, int64_t j) {
    v4f v = *(const v4f *)(a.vp + j);
    for (int z = 1; z < a.kzv; z++) v = v + *(const v4f *)(a.vp + (int64_t)z * a.n * a.mv + j);
    if (a.v_f16) { const qf_h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; *(qf_h4 *)(a.v_out + j * 2) = hv; }
    else         *(v4f *)(a.v_out + j * 4) = v;
}

// pairs 4c .. 4c + 3 of token tok in the 4 heads of head group g (q groups first, then k), rotated by (cs, sn)
static __device__ __forceinline__ void qkv_finish_quad(const QkvFin & a, int tok, int g, int c, const float * cs, const float * sn) {
    const int hd = a.p.ne0;
    const int nhq = a.mq / hd, nhk = a.mk / hd, gq = (nhq + 3) / 4;
    const bool neox = (a.p.mode & 2) != 0;
    const bool is_k = g >= gq;
    const int h0 = (is_k ? g - gq : g) * 4, nh = is_k ? nhk : nhq, m = is_k ? a.mk : a.mq, kz = is_k ? a.kzk : a.kzq;
    const float * part = is_k ? a.kp : a.qp;
    // NEOX: A = x0 of the 4 pairs, B = their x1 (half a head further); NORMAL: A = (x0, x1) of pairs 0, 1, B = of pairs 2, 3
    const int ea = neox ? 4 * c : 8 * c, eb = neox ? hd / 2 + 4 * c : 8 * c + 4;
    const int64_t plane = (int64_t)a.n * m, row = (int64_t)tok * m;
    v4f A[4], B[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int h = h0 + u < nh ? h0 + u : nh - 1;
        const float * pa = part + row + h * hd + ea, * pb = part + row + h * hd + eb;
        A[u] = *(const v4f *)pa; B[u] = *(const v4f *)pb;
        for (int z = 1; z < kz; z++) { A[u] = A[u] + *(const v4f *)(pa + z * plane); B[u] = B[u] + *(const v4f *)(pb + z * plane); }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (h0 + u >= nh) break;
        v4f YA, YB;
        if (neox) {
            YA = v4f{A[u].x * cs[0] - B[u].x * sn[0], A[u].y * cs[1] - B[u].y * sn[1], A[u].z * cs[2] - B[u].z * sn[2], A[u].w * cs[3] - B[u].w * sn[3]};
            YB = v4f{A[u].x * sn[0] + B[u].x * cs[0], A[u].y * sn[1] + B[u].y * cs[1], A[u].z * sn[2] + B[u].z * cs[2], A[u].w * sn[3] + B[u].w * cs[3]};
        } else {
            YA = v4f{A[u].x * cs[0] - A[u].y * sn[0], A[u].x * sn[0] + A[u].y * cs[0], A[u].z * cs[1] - A[u].w * sn[1], A[u].z * sn[1] + A[u].w * cs[1]};
            YB = v4f{B[u].x * cs[2] - B[u].y * sn[2], B[u].x * sn[2] + B[u].y * cs[2], B[u].z * cs[3] - B[u].w * sn[3], B[u].z * sn[3] + B[u].w * cs[3]};
        }
        const int64_t o = row + (h0 + u) * hd;
        if (!is_k) { *(v4f *)(a.q_out + o + ea) = YA; *(v4f *)(a.q_out + o + eb) = YB; }
        else if (a.k_f16) {
            const qf_h4 ha = {(_Float16)YA.x, (_Float16)YA.y, (_Float16)YA.z, (_Float16)YA.w}, hb = {(_Float16)YB.x, (_Float16)YB.y, (_Float16)YB.z, (_Float16)YB.w};
            *(qf_h4 *)((__half *)a.k_out + o + ea) = ha; *(qf_h4 *)((__half *)a.k_out + o + eb) = hb;
        } else { *(v4f *)((float *)a.k_out + o + ea) = YA; *(v4f *)((float *)a.k_out + o + eb) = YB; }
    }
}

__global__ void __launch_bounds__(256)
qkv_finish_vec_kernel(const QkvFin a) {
    const int hd = a.p.ne0, nquad = hd / 8;
    const int nhq = a.mq / hd, nhk = a.mk / hd, gq = (nhq + 3) / 4, gk = (nhk + 3) / 4;
    const int64_t nrope = (int64_t)nquad * a.n * (gq + gk), nv = (int64_t)a.n * (a.mv / 4);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrope + nv; i += stride) {
        if (i >= nrope) { qkv_finish_v4(a, (i - nrope) * 4); continue; }
        const int c = (int)(i % nquad);                // pairs 4c .. 4c + 3
        const int64_t r = i / nquad;
        const int tok = (int)(r % a.n), g = (int)(r / a.n);
        float cs[4], sn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rope_cos_sin(a.p, a.pos[tok], a.ff, 4 * c + u, cs[u], sn[u]);
        qkv_finish_quad(a, tok, g, c, cs, sn);
    }
}

// Few tokens (a step of 5..64): one workgroup per token.  In the kernel above every lane computes the cos / sin of its four pairs -- a
// chain of four powf + sincosf that ten lanes (one per head group) repeat -- and at 16 tokens the launch is 26 workgroups of little
// else: 7.2 us.  Here the token's hd / 2 rotations are computed once, by hd / 2 lanes, into LDS.  Same function, same arguments, same
// arithmetic after it: the same bits.
__global__ void __launch_bounds__(256)
qkv_finish_tok_kernel(const QkvFin a) {
    __shared__ float s_cs[256], s_sn[256];
    const int hd = a.p.ne0, nquad = hd / 8, tok = blockIdx.x;
    const int nhq = a.mq / hd, nhk = a.mk / hd, gq = (nhq + 3) / 4, gk = (nhk + 3) / 4;
    for (int p = threadIdx.x; p < hd / 2; p += 256) rope_cos_sin(a.p, a.pos[tok], a.ff, p, s_cs[p], s_sn[p]);
    __syncthreads();
    const int nrope = nquad * (gq + gk), nv = a.mv / 4;
    for (int i = threadIdx.x; i < nrope + nv; i += 256) {
        if (i >= nrope) { qkv_finish_v4(a, ((int64_t)tok * nv + (i - nrope)) * 4); continue; }
        const int c = i % nquad, g = i / nquad;
        qkv_finish_quad(a, tok, g, c, s_cs + 4 * c, s_sn + 4 * c);
    }
}

// ---------------------------------------------------------------- glu
static __device__ __forceinline__ float act_glu(int op, float x) {
    switch (op) {
        case 0: return x > 0.0f ? x : 0.0f;                                  // REGLU
        case 1: { const float c = 0.044715f, s = 0.79788456080286535587989211986876f; // GEGLU (tanh gelu, ggml_gelu_f32)
                  return 0.5f * x * (1.0f + tanhf(s * x * (1.0f + c * x * x))); }
        default: return x / (1.0f + expf(-x));                               // SWIGLU (ggml_silu_f32, vec.h:637-639)
    }
}

__global__ void __launch_bounds__(256)
glu_kernel(int op, const char * a, const char * b, int64_t a_row, int64_t b_row, char * d, int64_t d_row, int nc, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / nc; const int c = (int)(i % nc);
        const float x = ((const float *)(a + r * a_row))[c];
        const float g = ((const float *)(b + r * b_row))[c];
        ((float *)(d + r * d_row))[c] = act_glu(op, x) * g;
    }
}

// vectorized split-GLU that also stores the bf16 form of the result (prefill: the result feeds the ffn_down GEMM)
__global__ void __launch_bounds__(256)
glu_bf16_kernel(int op, const char * a, const char * b, int64_t a_row, int64_t b_row, char * d, int64_t d_row, int nc, uint16_t * __restrict__ abf) {
    const int i = blockIdx.y * 256 + threadIdx.x;    // float4 index inside the row
    if (i * 4 >= nc) return;
    const int64_t r = blockIdx.x;
    const v4f x = ((const v4f *)(a + r * a_row))[i], g = ((const v4f *)(b + r * b_row))[i];
    const v4f ov = {act_glu(op, x.x) * g.x, act_glu(op, x.y) * g.y, act_glu(op, x.z) * g.z, act_glu(op, x.w) * g.w};
    ((v4f *)(d + r * d_row))[i] = ov;
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    const bf4 bv = {(__bf16)ov.x, (__bf16)ov.y, (__bf16)ov.z, (__bf16)ov.w};
    *(bf4 *)(abf + r * nc + 4 * i) = bv;
}

// vectorized split-GLU with fused q8_1 quantization of the result (decode: the result feeds ffn_down)
__global__ void __launch_bounds__(256)
glu_quant_kernel(int op, const char * a, const char * b, int64_t a_row, int64_t b_row, char * d, int64_t d_row, int nc,
                 int8_t * __restrict__ aq, float * __restrict__ ads) {
    const int i = blockIdx.y * 256 + threadIdx.x;    // float4 index inside the row
    if (i * 4 >= nc) return;                         // nc % 32 == 0: whole 8-lane groups leave together
    const int64_t r = blockIdx.x;
    const v4f x = ((const v4f *)(a + r * a_row))[i], g = ((const v4f *)(b + r * b_row))[i];
    float o[4] = {act_glu(op, x.x) * g.x, act_glu(op, x.y) * g.y, act_glu(op, x.z) * g.z, act_glu(op, x.w) * g.w};
    v4f ov = {o[0], o[1], o[2], o[3]};
    ((v4f *)(d + r * d_row))[i] = ov;
    quant4_soa(o, aq + r * nc + 4 * i, ads + (r * (nc / QK) + (i >> 3)) * 2, (i & 7) == 0);
}

// vectorized split-GLU that also writes the q8_K planes of the result (batches of 5..32 rows: the result feeds ffn_down's skinny kernel).
// 16 values per lane, nc % 256 == 0: whole 16-lane rows (one q8_K block each) leave together
__global__ void __launch_bounds__(256)
glu_q8k_kernel(int op, const char * a, const char * b, int64_t a_row, int64_t b_row, char * d, int64_t d_row, int nc,
               int8_t * __restrict__ aq, float * __restrict__ ad, int16_t * __restrict__ kbs) {
    const int c = blockIdx.y * 256 + threadIdx.x;    // 16-value chunk inside the row
    if (c * 16 >= nc) return;
    const int64_t r = blockIdx.x;
    v4f x[4], g[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { x[j] = ((const v4f *)(a + r * a_row))[4 * c + j]; g[j] = ((const v4f *)(b + r * b_row))[4 * c + j]; }
    float o[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        o[4 * j + 0] = act_glu(op, x[j].x) * g[j].x; o[4 * j + 1] = act_glu(op, x[j].y) * g[j].y;
        o[4 * j + 2] = act_glu(op, x[j].z) * g[j].z; o[4 * j + 3] = act_glu(op, x[j].w) * g[j].w;
        ((v4f *)(d + r * d_row))[4 * c + j] = v4f{o[4 * j + 0], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
    }
    quant16_q8k(o, aq + r * nc + 16 * c, ad + r * (nc >> 8) + (c >> 4), kbs + r * (nc >> 4) + c, c & 15);
}

// ---------------------------------------------------------------- cpy
template <class TS, class TD> static __device__ __forceinline__ TD cvt(TS v);
template <> __device__ __forceinline__ float  cvt<float, float>(float v) { return v; }
template <> __device__ __forceinline__ __half cvt<float, __half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __half cvt<__half, __half>(__half v) { return v; }
template <> __device__ __forceinline__ float  cvt<__half, float>(__half v) { return __half2float(v); }
// bf16 as the reference rounds it (ggml_compute_fp32_to_bf16, ggml-impl.h:453-466): nearest even on the bits, NaNs made quiet
struct bf16_bits { uint16_t bits; };
template <> __device__ __forceinline__ bf16_bits cvt<float, bf16_bits>(float v) {
    const uint32_t i = __float_as_uint(v);
    bf16_bits h;
    if ((i & 0x7fffffffu) > 0x7f800000u) h.bits = (uint16_t)((i >> 16) | 64);
    else                                 h.bits = (uint16_t)((i + (0x7fffu + ((i >> 16) & 1u))) >> 16);
    return h;
}

static __device__ __forceinline__ int64_t off4(const T4 & t, int64_t i, int64_t blk, int64_t step0 = 0) {
    // linear index in units of blk elements along ne[0] -> byte offset; step0 = bytes per unit along ne[0]
    // (defaults to nb[0]: right for blk == 1 and for block-quantized tensors, whose nb[0] is the block size)
    const int64_t n0 = t.ne[0] / blk;
    const int64_t i0 = i % n0, r = i / n0;
    const int64_t i1 = r % t.ne[1], r2 = r / t.ne[1];
    const int64_t i2 = r2 % t.ne[2], i3 = r2 / t.ne[2];
    return i0 * (step0 ? step0 : t.nb[0]) + i1 * t.nb[1] + i2 * t.nb[2] + i3 * t.nb[3];
}

template <class TS, class TD>
__global__ void __launch_bounds__(256)
cpy_kernel(const T4 s, const T4 d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        *(TD *)(d.data + off4(d, i, 1)) = cvt<TS, TD>(*(const TS *)(s.data + off4(s, i, 1)));
}

// 32 f32 -> one q8_0 block (quantize_row_q8_0_ref, ggml-quants.c:188-211): the one arithmetic behind CPY, SET_ROWS and the KV store
static __device__ __forceinline__ void quant_block_q8_0(const float * xs, uint8_t * blk) {
    float v[32], amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; j++) { v[j] = xs[j]; amax = fmaxf(amax, fabsf(v[j])); }
    const float dd = __fdiv_rn(amax, 127.0f);
    const float id = dd != 0.0f ? __fdiv_rn(1.0f, dd) : 0.0f;
    *(uint16_t *)blk = __half_as_ushort(__float2half_rn(dd));
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const float r = roundf(__fmul_rn(v[j], id));
        ((int8_t *)blk)[2 + j] = fabsf(r) < 2147483648.0f ? (int8_t)(int)r : 0;
    }
}

// 32 f32 -> one q4_0 block (quantize_row_q4_0_ref, ggml-quants.c:25-60): d = (the value of largest magnitude) / -8, q = min(15, trunc(x / d
// + 8.5)), value j in the low nibble of byte j, value j + 16 in its high nibble
static __device__ __forceinline__ void quant_block_q4_0(const float * xs, uint8_t * blk) {
    float v[32], amax = 0.0f, mx = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; j++) { v[j] = xs[j]; if (amax < fabsf(v[j])) { amax = fabsf(v[j]); mx = v[j]; } }
    const float dd = __fmul_rn(mx, -0.125f);      // max / -8, exact; an all-zero block stores d = -0 as the reference does (0x8000)
    const float id = dd != 0.0f ? __fdiv_rn(1.0f, dd) : 0.0f;
    *(uint16_t *)blk = mx == 0.0f ? (uint16_t)0x8000 : __half_as_ushort(__float2half_rn(dd));   // (+0 / -8 = -0: spelt out, the product's sign did not survive the compiler)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float x0 = __fmul_rn(v[j], id), x1 = __fmul_rn(v[16 + j], id);
        // (int8_t)(x + 8.5f): the sum lies in [0.5, 16.5] for finite input; NaN / inf input converts to 0 here (undefined in C)
        const float s0 = __fadd_rn(x0, 8.5f), s1 = __fadd_rn(x1, 8.5f);
        const int q0 = fabsf(s0) < 2147483648.0f ? (int)(int8_t)(int)s0 : 0, q1 = fabsf(s1) < 2147483648.0f ? (int)(int8_t)(int)s1 : 0;
        const uint8_t u0 = (uint8_t)q0 < 15 ? (uint8_t)q0 : 15, u1 = (uint8_t)q1 < 15 ? (uint8_t)q1 : 15;
        blk[2 + j] = (uint8_t)(u0 | (u1 << 4));
    }
}
template <int TYPE> static __device__ __forceinline__ void quant_block(const float * xs, uint8_t * blk) {
    if (TYPE == MI355_TYPE_Q8_0) quant_block_q8_0(xs, blk); else quant_block_q4_0(xs, blk);
}
template <int TYPE> struct QBlk { static constexpr int bytes = TYPE == MI355_TYPE_Q8_0 ? 34 : 18; };

// K and V rows of one decode token, f32 -> q8_0 cache rows, in one launch; the destinations come from the arguments or, under graph
// replay, from the device table the fused Q/K/V launch uses (slot pair 2 s, 2 s + 1)
template <int TYPE>
__global__ void __launch_bounds__(64)
kv_store_q_kernel(const float * k_src, int nbk, uint8_t * k_dst, const float * v_src, int nbv, uint8_t * v_dst, void * const * tab, int slot) {
    const int ib = blockIdx.x * 64 + threadIdx.x;
    if (ib >= nbk + nbv) return;
    if (tab) { k_dst = (uint8_t *)tab[2 * slot]; v_dst = (uint8_t *)tab[2 * slot + 1]; }
    if (ib < nbk) quant_block<TYPE>(k_src + 32 * ib, k_dst + QBlk<TYPE>::bytes * ib);
    else          quant_block<TYPE>(v_src + 32 * (ib - nbk), v_dst + QBlk<TYPE>::bytes * (ib - nbk));
}

// f32 -> q8_0 / q4_0: one lane per 32-value block
template <int TYPE>
__global__ void __launch_bounds__(256)
cpy_f32_q_kernel(const T4 s, const T4 d, int64_t nblocks) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t ib = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ib < nblocks; ib += stride) {
        // the 32 source values are consecutive along the source's ne[0] (ne[0] % 32 == 0, nb[0] == 4)
        const float * xs = (const float *)(s.data + off4(s, ib, 32, 32 * sizeof(float)));
        uint8_t * blk = (uint8_t *)(d.data + off4(d, ib, 32));
        quant_block<TYPE>(xs, blk);
    }
}

// ---------------------------------------------------------------- set_rows
// dst[:, ids[i, i2 % ids.ne1, i3 % ids.ne2], i2, i3] = convert(src[:, i, i2, i3])   (ggml_compute_forward_set_rows_f32,
// ggml-cpu/ops.cpp:5179-5227).  Rows whose index is out of range are skipped (the CPU asserts).
static __device__ __forceinline__ int64_t set_rows_dst(const T4 & ids, const T4 & d, int64_t i1, int64_t i2, int64_t i3) {
    const int64_t r = *(const int64_t *)(ids.data + i1 * ids.nb[0] + (i2 % ids.ne[1]) * ids.nb[1] + (i3 % ids.ne[2]) * ids.nb[2]);
    return r >= 0 && r < d.ne[1] ? r * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3] : -1;
}

template <class TD>
__global__ void __launch_bounds__(256)
set_rows_kernel(const T4 s, const T4 ids, const T4 d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t i0 = i % s.ne[0], r = i / s.ne[0];
        const int64_t i1 = r % s.ne[1], r2 = r / s.ne[1];
        const int64_t i2 = r2 % s.ne[2], i3 = r2 / s.ne[2];
        const int64_t o = set_rows_dst(ids, d, i1, i2, i3);
        if (o >= 0) *(TD *)(d.data + o + i0 * sizeof(TD)) = cvt<float, TD>(*(const float *)(s.data + i0 * 4 + i1 * s.nb[1] + i2 * s.nb[2] + i3 * s.nb[3]));
    }
}

// q8_0 / q4_0 destination rows: one lane per 32-value block (quantize_row_q8_0_ref / quantize_row_q4_0_ref)
template <int TYPE>
__global__ void __launch_bounds__(256)
set_rows_q_kernel(const T4 s, const T4 ids, const T4 d, int64_t nblocks) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nb0 = s.ne[0] / 32;
    for (int64_t ib = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ib < nblocks; ib += stride) {
        const int64_t b0 = ib % nb0, r = ib / nb0;
        const int64_t i1 = r % s.ne[1], r2 = r / s.ne[1];
        const int64_t i2 = r2 % s.ne[2], i3 = r2 / s.ne[2];
        const int64_t o = set_rows_dst(ids, d, i1, i2, i3);
        if (o < 0) continue;
        const float * xs = (const float *)(s.data + b0 * 128 + i1 * s.nb[1] + i2 * s.nb[2] + i3 * s.nb[3]);
        uint8_t * blk = (uint8_t *)(d.data + o + b0 * QBlk<TYPE>::bytes);
        quant_block<TYPE>(xs, blk);
    }
}

// ---------------------------------------------------------------- get_rows
template <int TYPE>
__global__ void __launch_bounds__(256)
get_rows_kernel(const T4 s, const T4 ids, const T4 d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0];
        const int64_t i10 = r % d.ne[1], r2 = r / d.ne[1];
        const int64_t i11 = r2 % d.ne[2], i12 = r2 / d.ne[2];
        const int32_t row = *(const int32_t *)(ids.data + i10 * ids.nb[0] + i11 * ids.nb[1] + i12 * ids.nb[2]);
        const uint8_t * src_row = (const uint8_t *)(s.data + (int64_t)row * s.nb[1] + i11 * s.nb[2] + i12 * s.nb[3]);
        *(float *)(d.data + i0 * d.nb[0] + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3]) = deq_elem<TYPE>(src_row, i0);
    }
}

// ---------------------------------------------------------------- scale
__global__ void __launch_bounds__(256)
scale_kernel(const float * x, float * d, float scale, float bias, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = x[i] * scale + bias;
}

// ---------------------------------------------------------------- element-wise maps (UNARY ops, SQR / SQRT / SIN / COS / LOG, CLAMP)
// the arithmetic of ggml-cpu/unary-ops.cpp:3-65 and vec.h:547-640 (f32 in, f32 out, any strides)
static __device__ __forceinline__ float map_elem(int op, float x, float p0, float p1) {
    switch (op) {
        case MI355_MAP_ABS:         return fabsf(x);
        case MI355_MAP_SGN:         return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
        case MI355_MAP_NEG:         return -x;
        case MI355_MAP_STEP:        return x > 0.0f ? 1.0f : 0.0f;
        case MI355_MAP_TANH:        return tanhf(x);
        case MI355_MAP_ELU:         return x > 0.0f ? x : expm1f(x);
        case MI355_MAP_RELU:        return x > 0.0f ? x : 0.0f;
        case MI355_MAP_SIGMOID:     return 1.0f / (1.0f + expf(-x));
        case MI355_MAP_GELU:        return 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x)));
        case MI355_MAP_GELU_QUICK:  return x * (1.0f / (1.0f + expf(-1.702f * x)));
        case MI355_MAP_SILU:        return x / (1.0f + expf(-x));
        case MI355_MAP_HARDSWISH:   return x * fminf(1.0f, fmaxf(0.0f, (x + 3.0f) / 6.0f));
        case MI355_MAP_HARDSIGMOID: return fminf(1.0f, fmaxf(0.0f, (x + 3.0f) / 6.0f));
        case MI355_MAP_EXP:         return expf(x);
        case MI355_MAP_GELU_ERF:    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440084436210484f));
        case MI355_MAP_SQR:         return x * x;
        case MI355_MAP_SQRT:        return sqrtf(x);
        case MI355_MAP_SIN:         return sinf(x);
        case MI355_MAP_COS:         return cosf(x);
        case MI355_MAP_LOG:         return logf(x);
        default:                    return x < p0 ? p0 : (x > p1 ? p1 : x);      // MI355_MAP_CLAMP
    }
}
__global__ void __launch_bounds__(256)
map_kernel(int op, const T4 x, const T4 d, float p0, float p1, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0];
        const int64_t i1 = r % d.ne[1], r2 = r / d.ne[1];
        const int64_t i2 = r2 % d.ne[2], i3 = r2 / d.ne[2];
        const float v = *(const float *)(x.data + i0 * x.nb[0] + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
        *(float *)(d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = map_elem(op, v, p0, p1);
    }
}

// ---------------------------------------------------------------- NORM (LayerNorm without the affine part) and L2_NORM: one wave per row
// NORM: mean, then the variance of (x - mean), as ggml_compute_forward_norm_f32 (ops.cpp:3905-3960) does in two passes;
// L2_NORM: x / max(sqrt(sum x^2), eps) (ops.cpp:4314-4358)
template <bool L2>
__global__ void __launch_bounds__(256)
norm_kernel(const T4 x, const T4 d, float eps, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const float * xr = (const float *)(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float * dr = (float *)(d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int64_t n = x.ne[0];
    float mean = 0.0f, scale;
    if (L2) {
        float s = 0.0f;
        for (int64_t i = lane; i < n; i += 64) s += xr[i] * xr[i];
        s = wave_sum(s);
        scale = 1.0f / fmaxf(sqrtf(s), eps);
    } else {
        float s = 0.0f;
        for (int64_t i = lane; i < n; i += 64) s += xr[i];
        mean = wave_sum(s) / (float)n;
        float s2 = 0.0f;
        for (int64_t i = lane; i < n; i += 64) { const float v = xr[i] - mean; s2 += v * v; }
        s2 = wave_sum(s2);
        scale = 1.0f / sqrtf(s2 / (float)n + eps);
    }
    for (int64_t i = lane; i < n; i += 64) dr[i] = (xr[i] - mean) * scale;
}

// ---------------------------------------------------------------- soft_max (one block per row)
template <bool MASK_F16>
__global__ void __launch_bounds__(256)
soft_max_kernel(const T4 x, const T4 m, bool has_mask, const T4 d, float scale, float max_bias, float m0, float m1, uint32_t n_head_log2) {
    __shared__ float sh[4];
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const float * xr = (const float *)(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float * dr = (float *)(d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const char * mr = has_mask ? m.data + i1 * m.nb[1] + (i2 % m.ne[2]) * m.nb[2] + (i3 % m.ne[3]) * m.nb[3] : nullptr;
    const uint32_t h = (uint32_t)i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, h + 1) : powf(m1, 2 * (h - n_head_log2) + 1)) : 1.0f;
    const int n = (int)x.ne[0];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = xr[i] * scale;
        if (mr) v += slope * (MASK_F16 ? __half2float(((const __half *)mr)[i]) : ((const float *)mr)[i]);
        dr[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_max(mx, sh);
    float sum = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) { const float e = expf(dr[i] - mx); dr[i] = e; sum += e; }
    sum = block_sum(sum, sh);
    const float inv = 1.0f / sum;
    for (int i = threadIdx.x; i < n; i += 256) dr[i] *= inv;
}

} // namespace

#define CHECK_T(t) do { if (!(t) || !(t)->data) return MI355_E_NULL; } while (0)

extern "C" {

int mi355_binary_f32(int op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, void * stream) {
    CHECK_T(a); CHECK_T(b); CHECK_T(dst);
    if (a->type != MI355_TYPE_F32 || b->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(a, dst)) return MI355_E_SHAPE;
    for (int i = 0; i < 4; i++) if (b->ne[i] <= 0 || dst->ne[i] % b->ne[i] != 0) return nelem(dst) == 0 ? MI355_OK : MI355_E_SHAPE;
    const int64_t n = nelem(dst);
    if (n == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for(n)), block(256);
    switch (op) {
        case MI355_ADD: MI355_LAUNCH("binary", (binary_kernel<MI355_ADD>), grid, block, 0, s, t4(a), t4(b), t4(dst), n); break;
        case MI355_MUL: MI355_LAUNCH("binary", (binary_kernel<MI355_MUL>), grid, block, 0, s, t4(a), t4(b), t4(dst), n); break;
        case MI355_SUB: MI355_LAUNCH("binary", (binary_kernel<MI355_SUB>), grid, block, 0, s, t4(a), t4(b), t4(dst), n); break;
        case MI355_DIV: MI355_LAUNCH("binary", (binary_kernel<MI355_DIV>), grid, block, 0, s, t4(a), t4(b), t4(dst), n); break;
        default: return MI355_E_UNSUPPORTED;
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

static int launch_rms_norm(const mi355_tensor * x, const mi355_tensor * y, const mi355_tensor * sum, const float * w,
                           const mi355_tensor * dst, float eps, int8_t * aq, float * ads, void * stream, uint16_t * abf = nullptr,
                           const float * xparts = nullptr, int kz = 0, int16_t * kbs = nullptr) {
    CHECK_T(x); CHECK_T(dst);
    if (kbs && (!aq || !ads || !w || x->ne[0] % 256 != 0 || (((uintptr_t)aq | (uintptr_t)kbs) & 15) || ((uintptr_t)ads & 3))) return MI355_E_UNSUPPORTED;
    if (xparts && (!y || !abf || kz < 1 || kz > 8 || ((uintptr_t)xparts & 15) || x->ne[0] % 16 != 0)) return MI355_E_UNSUPPORTED;
    if (x->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(x, dst) || x->nb[0] != 4 || dst->nb[0] != 4) return MI355_E_SHAPE;
    const int64_t rows = nrows(x);
    if (rows == 0 || x->ne[0] == 0) return MI355_OK;
    if (rows > 0x7fffffff) return MI355_E_SHAPE;
    uintptr_t al = (uintptr_t)x->data | (uintptr_t)dst->data | (uintptr_t)w;
    for (int i = 1; i < 4; i++) al |= (uintptr_t)x->nb[i] | (uintptr_t)dst->nb[i];
    if (y) {
        CHECK_T(y); CHECK_T(sum);
        if (y->type != MI355_TYPE_F32 || sum->type != MI355_TYPE_F32) return MI355_E_TYPE;
        if (!same_shape(x, y) || !same_shape(x, sum) || y->nb[0] != 4 || sum->nb[0] != 4) return MI355_E_SHAPE;
        al |= (uintptr_t)y->data | (uintptr_t)sum->data;
        for (int i = 1; i < 4; i++) al |= (uintptr_t)y->nb[i] | (uintptr_t)sum->nb[i];
    }
    const bool vec = (al & 15) == 0 && x->ne[0] % 4 == 0 && x->ne[0] <= 8192;
    if (aq && (!vec || x->ne[0] % QK != 0 || !ads)) return MI355_E_UNSUPPORTED;
    if (abf && (!vec || x->ne[0] % 16 != 0 || ((uintptr_t)abf & 15))) return MI355_E_UNSUPPORTED;
    if (y && (!vec || !w)) return MI355_E_UNSUPPORTED;
    const dim3 grid((unsigned)rows);
    hipStream_t s = (hipStream_t)stream;
    if (x->ne[0] > 4096) {
        const dim3 block(512);
        if (y)      MI355_LAUNCH("add_rms_norm_mul", (rms_norm_kernel<true, true, 512>), grid, block, 0, s, t4(x), t4(y), t4(sum), w, t4(dst), eps, vec, aq, ads, abf, xparts, kz, kbs);
        else if (w) MI355_LAUNCH("rms_norm_mul", (rms_norm_kernel<true, false, 512>), grid, block, 0, s, t4(x), t4(x), t4(x), w, t4(dst), eps, vec, aq, ads, abf, nullptr, 0, kbs);
        else        MI355_LAUNCH("rms_norm", (rms_norm_kernel<false, false, 512>), grid, block, 0, s, t4(x), t4(x), t4(x), w, t4(dst), eps, vec, aq, ads, abf, nullptr, 0, (int16_t *)nullptr);
    } else {
        const dim3 block(256);
        if (y)      MI355_LAUNCH("add_rms_norm_mul", (rms_norm_kernel<true, true, 256>), grid, block, 0, s, t4(x), t4(y), t4(sum), w, t4(dst), eps, vec, aq, ads, abf, xparts, kz, kbs);
        else if (w) MI355_LAUNCH("rms_norm_mul", (rms_norm_kernel<true, false, 256>), grid, block, 0, s, t4(x), t4(x), t4(x), w, t4(dst), eps, vec, aq, ads, abf, nullptr, 0, kbs);
        else        MI355_LAUNCH("rms_norm", (rms_norm_kernel<false, false, 256>), grid, block, 0, s, t4(x), t4(x), t4(x), w, t4(dst), eps, vec, aq, ads, abf, nullptr, 0, (int16_t *)nullptr);
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_rms_norm_f32(const mi355_tensor * x, const mi355_tensor * dst, float eps, void * stream) {
    return launch_rms_norm(x, nullptr, nullptr, nullptr, dst, eps, nullptr, nullptr, stream);
}

int mi355_rms_norm_mul_f32(const mi355_tensor * x, const float * w, const mi355_tensor * dst, float eps, void * stream) {
    if (!w) return MI355_E_NULL;
    return launch_rms_norm(x, nullptr, nullptr, w, dst, eps, nullptr, nullptr, stream);
}

int mi355_rms_norm_mul_quant_f32(const mi355_tensor * x, const float * w, const mi355_tensor * dst, float eps,
                                 int8_t * act_q, float * act_ds, void * stream) {
    if (!w || !act_q || !act_ds) return MI355_E_NULL;
    return launch_rms_norm(x, nullptr, nullptr, w, dst, eps, act_q, act_ds, stream);
}

int mi355_rms_norm_mul_q8k_f32(const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * sum, const float * w,
                               const mi355_tensor * dst, float eps, int8_t * act_q, float * act_d, int16_t * act_bsums, void * stream) {
    if (!w || !act_q || !act_d || !act_bsums || ((b == nullptr) != (sum == nullptr))) return MI355_E_NULL;
    return launch_rms_norm(a, b, sum, w, dst, eps, act_q, act_d, stream, nullptr, nullptr, 0, act_bsums);
}

int mi355_add_rms_norm_mul_f32(const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * sum, const float * w,
                               const mi355_tensor * dst, float eps, int8_t * act_q, float * act_ds, void * stream) {
    if (!w || !b || !sum) return MI355_E_NULL;
    if ((act_q == nullptr) != (act_ds == nullptr)) return MI355_E_NULL;
    return launch_rms_norm(a, b, sum, w, dst, eps, act_q, act_ds, stream);
}

int mi355_rms_norm_mul_bf16_f32(const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * sum, const float * w,
                                const mi355_tensor * dst, float eps, uint16_t * act_bf16, void * stream) {
    if (!w || !act_bf16 || ((b == nullptr) != (sum == nullptr))) return MI355_E_NULL;
    return launch_rms_norm(a, b, sum, w, dst, eps, nullptr, nullptr, stream, act_bf16);
}

int mi355_parts_add_rms_norm_mul_bf16_f32(const float * a_parts, int kz, const mi355_tensor * b, const mi355_tensor * sum, const float * w,
                                          const mi355_tensor * dst, float eps, uint16_t * act_bf16, void * stream) {
    if (!a_parts || !b || !sum || !w || !act_bf16) return MI355_E_NULL;
    CHECK_T(sum);
    // a has sum's shape and is dense: [rows][ne0] per plane
    mi355_tensor a = *sum;
    a.data = (void *)a_parts;
    a.nb[0] = 4; a.nb[1] = a.ne[0] * 4; a.nb[2] = a.nb[1] * a.ne[1]; a.nb[3] = a.nb[2] * a.ne[2];
    // two rows per workgroup (see add_rms_norm_planes_kernel) where the shapes are the plain ones of a prefill layer
    CHECK_T(b); CHECK_T(dst);
    static const bool two_rows = getenv("MI355_NORM_TWO_ROWS") == nullptr || atoi(getenv("MI355_NORM_TWO_ROWS")) != 0;
    const int64_t n = sum->ne[0], rows = sum->ne[1];
    uintptr_t al = (uintptr_t)a_parts | (uintptr_t)b->data | (uintptr_t)sum->data | (uintptr_t)dst->data | (uintptr_t)w | (uintptr_t)act_bf16 |
                   (uintptr_t)b->nb[1] | (uintptr_t)sum->nb[1] | (uintptr_t)dst->nb[1];
    if (two_rows && kz >= 1 && kz <= 4 && n % 16 == 0 && n <= 8192 && rows >= 2 && rows <= 0x7fffffff && sum->ne[2] == 1 && sum->ne[3] == 1 &&
        same_shape(b, sum) && same_shape(dst, sum) && b->type == MI355_TYPE_F32 && sum->type == MI355_TYPE_F32 && dst->type == MI355_TYPE_F32 &&
        b->nb[0] == 4 && sum->nb[0] == 4 && dst->nb[0] == 4 && (al & 15) == 0) {
        const dim3 grid((unsigned)((rows + 1) / 2));
        const int64_t plane = rows * n;
        hipStream_t s = (hipStream_t)stream;
#define NORM_PLANES_NT(KZ, NT) MI355_LAUNCH("add_rms_norm_mul", (add_rms_norm_planes_kernel<KZ, NT>), grid, dim3(NT), 0, s, a_parts, plane, (const float *)b->data, \
        (int64_t)(b->nb[1] / 4), (float *)sum->data, (int64_t)(sum->nb[1] / 4), w, (float *)dst->data, (int64_t)(dst->nb[1] / 4), act_bf16, (int)n, (int)rows, eps)
#define NORM_PLANES(KZ) do { if (n > 4096) NORM_PLANES_NT(KZ, 512); else NORM_PLANES_NT(KZ, 256); } while (0)
        switch (kz) { case 1: NORM_PLANES(1); break; case 2: NORM_PLANES(2); break; case 3: NORM_PLANES(3); break; default: NORM_PLANES(4); break; }
#undef NORM_PLANES
#undef NORM_PLANES_NT
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    return launch_rms_norm(&a, b, sum, w, dst, eps, nullptr, nullptr, stream, act_bf16, a_parts, kz);
}

int mi355_rope_f32(const mi355_tensor * x, const int32_t * pos, const float * freq_factors, const mi355_tensor * dst,
                   int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                   float attn_factor, float beta_fast, float beta_slow, void * stream) {
    CHECK_T(x); CHECK_T(dst);
    if (!pos) return MI355_E_NULL;
    if (x->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(x, dst) || x->nb[0] != 4 || dst->nb[0] != 4 || n_dims > x->ne[0] || (n_dims & 1) || (x->ne[0] & 1)) return MI355_E_SHAPE;
    if (mode != 0 && mode != 2) return MI355_E_UNSUPPORTED; // mrope / vision are not on this path
    const RopeP p = rope_params(n_dims, mode, (int)x->ne[0], n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
    const int64_t npairs = nelem(x) / 2;
    if (npairs == 0) return MI355_OK;
    // many heads and enough (pair, token) work to fill the chip: compute cos / sin once per (pair, token), walk the heads
    const int64_t npt = npairs / (x->ne[1] > 0 ? x->ne[1] : 1);
    if (x->ne[1] >= 4 && npt >= 16384)
        MI355_LAUNCH("rope", rope_heads_kernel, dim3(grid_for(npt)), dim3(256), 0, (hipStream_t)stream, t4(x), pos, freq_factors, t4(dst), p, npt);
    else
        MI355_LAUNCH("rope", rope_kernel, dim3(grid_for(npairs)), dim3(256), 0, (hipStream_t)stream, t4(x), pos, freq_factors, t4(dst), p, npairs);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_qkv_finish(const mi355_qkv_finish_args * f, void * stream) {
    if (!f || !f->q_parts || !f->k_parts || !f->v_parts || !f->q_out || !f->k_out || !f->v_out || !f->pos) return MI355_E_NULL;
    if (f->kz_q < 1 || f->kz_k < 1 || f->kz_v < 1 || f->kz_q > 8 || f->kz_k > 8 || f->kz_v > 8) return MI355_E_SHAPE;
    const int64_t hd = f->head_dim;
    if (hd <= 0 || (hd & 1) || f->n_dims > hd || (f->n_dims & 1) || f->n < 0 || f->m_q <= 0 || f->m_k <= 0 || f->m_v <= 0 ||
        f->m_q % hd || f->m_k % hd || f->m_v % 4 || f->n > 0x7fffffff || f->m_q > 0x7fffffff || f->m_k > 0x7fffffff || f->m_v > 0x7fffffff) return MI355_E_SHAPE;
    if (f->mode != 0 && f->mode != 2) return MI355_E_UNSUPPORTED;
    if ((f->k_out_type != MI355_TYPE_F16 && f->k_out_type != MI355_TYPE_F32) || (f->v_out_type != MI355_TYPE_F16 && f->v_out_type != MI355_TYPE_F32)) return MI355_E_TYPE;
    if (((uintptr_t)f->v_parts & 15) || ((uintptr_t)f->v_out & 15) || ((uintptr_t)f->q_parts & 3) || ((uintptr_t)f->k_parts & 3) ||
        ((uintptr_t)f->q_out & 3) || ((uintptr_t)f->k_out & 3) || ((f->n * f->m_v) & 3)) return MI355_E_ALIGN;
    if (f->n == 0) return MI355_OK;
    QkvFin a;
    a.qp = f->q_parts; a.kp = f->k_parts; a.vp = f->v_parts; a.kzq = f->kz_q; a.kzk = f->kz_k; a.kzv = f->kz_v;
    a.mq = (int)f->m_q; a.mk = (int)f->m_k; a.mv = (int)f->m_v; a.n = (int)f->n;
    a.q_out = f->q_out; a.k_out = (char *)f->k_out; a.v_out = (char *)f->v_out;
    a.k_f16 = f->k_out_type == MI355_TYPE_F16; a.v_f16 = f->v_out_type == MI355_TYPE_F16;
    a.pos = f->pos; a.ff = f->freq_factors;
    a.p = rope_params(f->n_dims, f->mode, (int)hd, f->n_ctx_orig, f->freq_base, f->freq_scale, f->ext_factor, f->attn_factor, f->beta_fast, f->beta_slow);
    const uintptr_t al = (uintptr_t)f->q_parts | (uintptr_t)f->k_parts | (uintptr_t)f->q_out | (uintptr_t)f->k_out;
    if (f->n_dims == hd && hd % 8 == 0 && (al & 15) == 0) {
        const int64_t gq = (f->m_q / hd + 3) / 4, gk = (f->m_k / hd + 3) / 4;
        const int64_t lanes = hd / 8 * f->n * (gq + gk) + f->n * (f->m_v / 4);
        if (f->n <= 64 && hd <= 512) MI355_LAUNCH("qkv_finish", qkv_finish_tok_kernel, dim3((unsigned)f->n), dim3(256), 0, (hipStream_t)stream, a);
        else                         MI355_LAUNCH("qkv_finish", qkv_finish_vec_kernel, dim3(grid_for(lanes)), dim3(256), 0, (hipStream_t)stream, a);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    const int64_t gq = (f->m_q / hd + 7) / 8, gk = (f->m_k / hd + 7) / 8;
    const int64_t lanes = hd / 2 * f->n * (gq + gk) + f->n * (f->m_v / 4);
    MI355_LAUNCH("qkv_finish", qkv_finish_kernel, dim3(grid_for(lanes)), dim3(256), 0, (hipStream_t)stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

static int launch_glu(int glu_op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, int swapped,
                      int8_t * aq, float * ads, void * stream, uint16_t * abf = nullptr, int16_t * kbs = nullptr);

int mi355_glu_q8k_f32(int glu_op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, int swapped,
                      int8_t * act_q, float * act_d, int16_t * act_bsums, void * stream) {
    if (!act_q || !act_d || !act_bsums) return MI355_E_NULL;
    return launch_glu(glu_op, a, b, dst, swapped, act_q, act_d, stream, nullptr, act_bsums);
}

int mi355_glu_bf16_f32(int glu_op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, int swapped,
                       uint16_t * act_bf16, void * stream) {
    if (!act_bf16) return MI355_E_NULL;
    return launch_glu(glu_op, a, b, dst, swapped, nullptr, nullptr, stream, act_bf16);
}

int mi355_glu_f32(int glu_op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, int swapped, void * stream) {
    return launch_glu(glu_op, a, b, dst, swapped, nullptr, nullptr, stream);
}

int mi355_glu_quant_f32(int glu_op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, int swapped,
                        int8_t * act_q, float * act_ds, void * stream) {
    if (!act_q || !act_ds) return MI355_E_NULL;
    return launch_glu(glu_op, a, b, dst, swapped, act_q, act_ds, stream);
}

static int launch_glu(int glu_op, const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, int swapped,
                      int8_t * aq, float * ads, void * stream, uint16_t * abf, int16_t * kbs) {
    CHECK_T(a); CHECK_T(dst);
    if (b && !b->data) return MI355_E_NULL;
    if (a->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32 || (b && b->type != MI355_TYPE_F32)) return MI355_E_TYPE;
    if (glu_op < 0 || glu_op > 2) return MI355_E_UNSUPPORTED;
    const int64_t nc = b ? a->ne[0] : a->ne[0] / 2;
    if (dst->ne[0] != nc || nrows(dst) != nrows(a) || (b && !same_shape(a, b))) return MI355_E_SHAPE;
    // rows contiguous ("contiguous_1"): row r lives at data + r*nb[1]
    auto rows_dense = [](const mi355_tensor * t) { return t->nb[0] == 4 && t->nb[2] == t->nb[1] * t->ne[1] && t->nb[3] == t->nb[2] * t->ne[2]; };
    if (!rows_dense(a) || !rows_dense(dst) || (b && !rows_dense(b))) return MI355_E_UNSUPPORTED;
    const int64_t n = nc * nrows(dst);
    if (n == 0) return MI355_OK;
    const char * ap = (const char *)a->data, * bp = b ? (const char *)b->data : (const char *)a->data;
    if (!b) { ap += swapped ? nc * 4 : 0; bp += swapped ? 0 : nc * 4; }
    if (abf) {
        const uintptr_t al = (uintptr_t)ap | (uintptr_t)bp | (uintptr_t)dst->data | (uintptr_t)a->nb[1] | (uintptr_t)(b ? b->nb[1] : a->nb[1]) | (uintptr_t)dst->nb[1];
        if ((al & 15) || ((uintptr_t)abf & 7) || nc % 4 != 0 || nrows(dst) > 0x7fffffff) return MI355_E_UNSUPPORTED;
        const dim3 grid((unsigned)nrows(dst), (unsigned)((nc / 4 + 255) / 256));
        MI355_LAUNCH("glu_bf16", glu_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, glu_op, ap, bp, a->nb[1],
                     b ? b->nb[1] : a->nb[1], (char *)dst->data, dst->nb[1], (int)nc, abf);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    if (aq && kbs) {
        const uintptr_t al = (uintptr_t)ap | (uintptr_t)bp | (uintptr_t)dst->data | (uintptr_t)a->nb[1] | (uintptr_t)(b ? b->nb[1] : a->nb[1]) | (uintptr_t)dst->nb[1] | (uintptr_t)aq;
        if ((al & 15) || nc % 256 != 0 || nrows(dst) > 0x7fffffff) return MI355_E_UNSUPPORTED;
        const dim3 grid((unsigned)nrows(dst), (unsigned)((nc / 16 + 255) / 256));
        MI355_LAUNCH("glu_q8k", glu_q8k_kernel, grid, dim3(256), 0, (hipStream_t)stream, glu_op, ap, bp, a->nb[1],
                     b ? b->nb[1] : a->nb[1], (char *)dst->data, dst->nb[1], (int)nc, aq, ads, kbs);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    if (aq) {
        const uintptr_t al = (uintptr_t)ap | (uintptr_t)bp | (uintptr_t)dst->data | (uintptr_t)a->nb[1] | (uintptr_t)(b ? b->nb[1] : a->nb[1]) | (uintptr_t)dst->nb[1];
        if ((al & 15) || nc % QK != 0 || nrows(dst) > 0x7fffffff) return MI355_E_UNSUPPORTED;
        const dim3 grid((unsigned)nrows(dst), (unsigned)((nc / 4 + 255) / 256));
        MI355_LAUNCH("glu_quant", glu_quant_kernel, grid, dim3(256), 0, (hipStream_t)stream, glu_op, ap, bp, a->nb[1],
                     b ? b->nb[1] : a->nb[1], (char *)dst->data, dst->nb[1], (int)nc, aq, ads);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    MI355_LAUNCH("glu", glu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, glu_op, ap, bp, a->nb[1],
                 b ? b->nb[1] : a->nb[1], (char *)dst->data, dst->nb[1], (int)nc, n);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_cpy(const mi355_tensor * src, const mi355_tensor * dst, void * stream) {
    CHECK_T(src); CHECK_T(dst);
    const int64_t n = nelem(src);
    if (n != nelem(dst)) return MI355_E_SHAPE;
    if (n == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256);
    const int st = src->type, dt = dst->type;
    if (st == MI355_TYPE_F32 && (dt == MI355_TYPE_Q8_0 || dt == MI355_TYPE_Q4_0)) {
        if (src->ne[0] % 32 || dst->ne[0] % 32 || src->nb[0] != 4 || dst->nb[0] != (dt == MI355_TYPE_Q8_0 ? 34 : 18)) return MI355_E_SHAPE;
        const int64_t nblk = n / 32;
        if (dt == MI355_TYPE_Q8_0) MI355_LAUNCH("cpy", cpy_f32_q_kernel<MI355_TYPE_Q8_0>, dim3(grid_for(nblk)), block, 0, s, t4(src), t4(dst), nblk);
        else                       MI355_LAUNCH("cpy", cpy_f32_q_kernel<MI355_TYPE_Q4_0>, dim3(grid_for(nblk)), block, 0, s, t4(src), t4(dst), nblk);
    } else if (st == MI355_TYPE_F32 && dt == MI355_TYPE_BF16) {
        MI355_LAUNCH("cpy", (cpy_kernel<float, bf16_bits>), dim3(grid_for(n)), block, 0, s, t4(src), t4(dst), n);
    } else if (st == MI355_TYPE_F32 && dt == MI355_TYPE_F32) {
        MI355_LAUNCH("cpy", (cpy_kernel<float, float>), dim3(grid_for(n)), block, 0, s, t4(src), t4(dst), n);
    } else if (st == MI355_TYPE_F32 && dt == MI355_TYPE_F16) {
        MI355_LAUNCH("cpy", (cpy_kernel<float, __half>), dim3(grid_for(n)), block, 0, s, t4(src), t4(dst), n);
    } else if (st == MI355_TYPE_F16 && dt == MI355_TYPE_F16) {
        MI355_LAUNCH("cpy", (cpy_kernel<__half, __half>), dim3(grid_for(n)), block, 0, s, t4(src), t4(dst), n);
    } else if (st == MI355_TYPE_F16 && dt == MI355_TYPE_F32) {
        MI355_LAUNCH("cpy", (cpy_kernel<__half, float>), dim3(grid_for(n)), block, 0, s, t4(src), t4(dst), n);
    } else {
        return MI355_E_TYPE;
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_kv_store_q8_0(const float * k_src, int64_t n_k, void * k_dst, const float * v_src, int64_t n_v, void * v_dst,
                        void * const * kv_table, int kv_slot, void * stream) {
    return mi355_kv_store_q(MI355_TYPE_Q8_0, k_src, n_k, k_dst, v_src, n_v, v_dst, kv_table, kv_slot, stream);
}

int mi355_kv_store_q(int type, const float * k_src, int64_t n_k, void * k_dst, const float * v_src, int64_t n_v, void * v_dst,
                     void * const * kv_table, int kv_slot, void * stream) {
    if (type != MI355_TYPE_Q8_0 && type != MI355_TYPE_Q4_0) return MI355_E_TYPE;
    if (!k_src || !v_src || (!kv_table && (!k_dst || !v_dst))) return MI355_E_NULL;
    if (n_k < 0 || n_v < 0 || n_k % 32 || n_v % 32 || n_k + n_v > 0x7fffffff) return MI355_E_SHAPE;
    if ((((uintptr_t)k_src | (uintptr_t)v_src) & 3) || (((uintptr_t)k_dst | (uintptr_t)v_dst) & 1) || ((uintptr_t)kv_table & 7)) return MI355_E_ALIGN;
    if (kv_table && kv_slot < 0) return MI355_E_SHAPE;
    const int nb = (int)((n_k + n_v) / 32);
    if (nb == 0) return MI355_OK;
    if (type == MI355_TYPE_Q8_0)
        MI355_LAUNCH("kv_store", kv_store_q_kernel<MI355_TYPE_Q8_0>, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     k_src, (int)(n_k / 32), (uint8_t *)k_dst, v_src, (int)(n_v / 32), (uint8_t *)v_dst, kv_table, kv_slot);
    else
        MI355_LAUNCH("kv_store", kv_store_q_kernel<MI355_TYPE_Q4_0>, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     k_src, (int)(n_k / 32), (uint8_t *)k_dst, v_src, (int)(n_v / 32), (uint8_t *)v_dst, kv_table, kv_slot);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_set_rows(const mi355_tensor * src, const mi355_tensor * ids, const mi355_tensor * dst, void * stream) {
    CHECK_T(src); CHECK_T(ids); CHECK_T(dst);
    if (src->type != MI355_TYPE_F32 || ids->type != MI355_TYPE_I64) return MI355_E_TYPE;
    if (src->nb[0] != 4 || dst->nb[0] != mi355_type_size(dst->type)) return MI355_E_SHAPE;        // rows contiguous
    if (src->ne[0] != dst->ne[0] || src->ne[2] != dst->ne[2] || src->ne[3] != dst->ne[3] || ids->ne[0] != src->ne[1] || ids->ne[1] < 1 ||
        ids->ne[2] < 1 || src->ne[2] % ids->ne[1] != 0 || src->ne[3] % ids->ne[2] != 0 || ids->ne[3] != 1) return MI355_E_SHAPE;
    if (((uintptr_t)ids->data | ids->nb[0] | ids->nb[1] | ids->nb[2]) & 7) return MI355_E_ALIGN;
    if (((uintptr_t)src->data | src->nb[1] | src->nb[2] | src->nb[3]) & 3) return MI355_E_ALIGN;
    const int64_t n = nelem(src);
    if (n == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256);
    switch (dst->type) {
        case MI355_TYPE_F32:
            if (((uintptr_t)dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 3) return MI355_E_ALIGN;
            MI355_LAUNCH("set_rows", (set_rows_kernel<float>), dim3(grid_for(n)), block, 0, s, t4(src), t4(ids), t4(dst), n); break;
        case MI355_TYPE_F16:
            if (((uintptr_t)dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 1) return MI355_E_ALIGN;
            MI355_LAUNCH("set_rows", (set_rows_kernel<__half>), dim3(grid_for(n)), block, 0, s, t4(src), t4(ids), t4(dst), n); break;
        case MI355_TYPE_BF16:
            if (((uintptr_t)dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 1) return MI355_E_ALIGN;
            MI355_LAUNCH("set_rows", (set_rows_kernel<bf16_bits>), dim3(grid_for(n)), block, 0, s, t4(src), t4(ids), t4(dst), n); break;
        case MI355_TYPE_Q8_0:
        case MI355_TYPE_Q4_0:
            if (src->ne[0] % 32) return MI355_E_SHAPE;
            if (((uintptr_t)dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 1) return MI355_E_ALIGN;
            if (dst->type == MI355_TYPE_Q8_0) MI355_LAUNCH("set_rows", set_rows_q_kernel<MI355_TYPE_Q8_0>, dim3(grid_for(n / 32)), block, 0, s, t4(src), t4(ids), t4(dst), n / 32);
            else                              MI355_LAUNCH("set_rows", set_rows_q_kernel<MI355_TYPE_Q4_0>, dim3(grid_for(n / 32)), block, 0, s, t4(src), t4(ids), t4(dst), n / 32);
            break;
        default: return MI355_E_TYPE;
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_get_rows(const mi355_tensor * src, const mi355_tensor * ids, const mi355_tensor * dst, void * stream) {
    CHECK_T(src); CHECK_T(ids); CHECK_T(dst);
    if (dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (dst->ne[0] != src->ne[0] || dst->ne[1] != ids->ne[0] || dst->ne[2] != ids->ne[1] || dst->ne[3] != ids->ne[2]) return MI355_E_SHAPE;
    const int64_t n = nelem(dst);
    if (n == 0) return MI355_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for(n)), block(256);
#define GR(T) MI355_LAUNCH("get_rows", (get_rows_kernel<T>), grid, block, 0, s, t4(src), t4(ids), t4(dst), n); break
    switch (src->type) {
        case MI355_TYPE_F32:  GR(MI355_TYPE_F32);
        case MI355_TYPE_F16:  GR(MI355_TYPE_F16);
        case MI355_TYPE_Q4_0: GR(MI355_TYPE_Q4_0);
        case MI355_TYPE_Q4_1: GR(MI355_TYPE_Q4_1);
        case MI355_TYPE_Q5_0: GR(MI355_TYPE_Q5_0);
        case MI355_TYPE_Q5_1: GR(MI355_TYPE_Q5_1);
        case MI355_TYPE_BF16: GR(MI355_TYPE_BF16);
        case MI355_TYPE_Q8_0: GR(MI355_TYPE_Q8_0);
        case MI355_TYPE_Q4_K: GR(MI355_TYPE_Q4_K);
        case MI355_TYPE_Q5_K: GR(MI355_TYPE_Q5_K);
        case MI355_TYPE_Q6_K: GR(MI355_TYPE_Q6_K);
        case MI355_TYPE_Q2_K: GR(MI355_TYPE_Q2_K);
        case MI355_TYPE_Q3_K: GR(MI355_TYPE_Q3_K);
        case MI355_TYPE_IQ4_NL: GR(MI355_TYPE_IQ4_NL);
        case MI355_TYPE_IQ4_XS: GR(MI355_TYPE_IQ4_XS);
        default: return MI355_E_TYPE;
    }
#undef GR
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_scale_f32(const mi355_tensor * x, const mi355_tensor * dst, float scale, float bias, void * stream) {
    CHECK_T(x); CHECK_T(dst);
    if (x->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(x, dst)) return MI355_E_SHAPE;
    const int64_t n = nelem(x);
    if (n == 0) return MI355_OK;
    MI355_LAUNCH("scale", scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float *)x->data, (float *)dst->data, scale, bias, n);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_map_f32(int op, const mi355_tensor * x, const mi355_tensor * dst, float p0, float p1, void * stream) {
    CHECK_T(x); CHECK_T(dst);
    if (op < 0 || op > MI355_MAP_CLAMP) return MI355_E_UNSUPPORTED;
    if (x->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(x, dst)) return MI355_E_SHAPE;
    if (((uintptr_t)x->data | (uintptr_t)dst->data | (uintptr_t)(x->nb[0] | x->nb[1] | x->nb[2] | x->nb[3] | dst->nb[0] | dst->nb[1] | dst->nb[2] | dst->nb[3])) & 3) return MI355_E_ALIGN;
    const int64_t n = nelem(x);
    if (n == 0) return MI355_OK;
    MI355_LAUNCH("map", map_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, t4(x), t4(dst), p0, p1, n);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_norm_f32(const mi355_tensor * x, const mi355_tensor * dst, float eps, int l2, void * stream) {
    CHECK_T(x); CHECK_T(dst);
    if (x->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(x, dst) || x->nb[0] != 4 || dst->nb[0] != 4 || !(eps >= 0.0f)) return MI355_E_SHAPE;
    if (((uintptr_t)x->data | (uintptr_t)dst->data | (uintptr_t)(x->nb[1] | x->nb[2] | x->nb[3] | dst->nb[1] | dst->nb[2] | dst->nb[3])) & 3) return MI355_E_ALIGN;
    const int64_t rows = nrows(x);
    if (rows == 0 || x->ne[0] == 0) return MI355_OK;
    if ((rows + 3) / 4 > 0x7fffffff) return MI355_E_SHAPE;
    if (l2) MI355_LAUNCH("l2_norm", norm_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, t4(x), t4(dst), eps, rows);
    else    MI355_LAUNCH("norm", norm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, t4(x), t4(dst), eps, rows);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

int mi355_soft_max_f32(const mi355_tensor * x, const mi355_tensor * mask, const mi355_tensor * dst, float scale, float max_bias, void * stream) {
    CHECK_T(x); CHECK_T(dst);
    if (mask && !mask->data) return MI355_E_NULL;
    if (x->type != MI355_TYPE_F32 || dst->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (mask && mask->type != MI355_TYPE_F16 && mask->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (!same_shape(x, dst) || x->nb[0] != 4 || dst->nb[0] != 4) return MI355_E_SHAPE;
    const int64_t rows = nrows(x);
    if (rows == 0 || x->ne[0] == 0) return MI355_OK;
    if (rows > 0x7fffffff) return MI355_E_SHAPE;
    const uint32_t n_head = (uint32_t)x->ne[2];
    const uint32_t n_head_log2 = 1u << (uint32_t)floorf(log2f((float)n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    T4 m = mask ? t4(mask) : t4(x);
    hipStream_t s = (hipStream_t)stream;
    if (mask && mask->type == MI355_TYPE_F16)
        MI355_LAUNCH("soft_max", (soft_max_kernel<true>), dim3((unsigned)rows), dim3(256), 0, s, t4(x), m, true, t4(dst), scale, max_bias, m0, m1, n_head_log2);
    else
        MI355_LAUNCH("soft_max", (soft_max_kernel<false>), dim3((unsigned)rows), dim3(256), 0, s, t4(x), m, mask != nullptr, t4(dst), scale, max_bias, m0, m1, n_head_log2);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

} // extern "C"


// ------------------------------------------------------------------------------------------------
// The small ops of a Mixture-of-Experts router (llama-graph.cpp:650-795): logits = MUL_MAT(f32 gate_inp, x),
// SOFT_MAX, top-k = ARGSORT(desc) + view, GET_ROWS of the probabilities, SUM_ROWS + DIV to renormalise.
// ------------------------------------------------------------------------------------------------
namespace {

// bitonic argsort of one row per workgroup in LDS (rows of at most 1024 values, padded to a power of two; padding
// sorts to the end).  Ties are ordered by index, so the result is deterministic.
__global__ void __launch_bounds__(1024)
argsort_kernel(const T4 x, const T4 dst, int npad, int desc) {
    __shared__ int idx[1024];
    const int col = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const float * xr = (const float *)(x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    const int n = (int)x.ne[0];
    idx[col] = col;
    __syncthreads();
    // "a sorts after b" in the requested order; indices >= n are padding and sort after everything
    auto after = [&](int a, int b) {
        if (a >= n) return b < n || a > b;
        if (b >= n) return false;
        const float va = xr[a], vb = xr[b];
        if (va == vb) return a > b;
        return desc ? va < vb : va > vb;
    };
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = col ^ j;
            if (ixj > col) {
                const int a = idx[col], b = idx[ixj];
                const bool up = (col & k) == 0;      // ascending run: the smaller goes first
                if (up ? after(a, b) : after(b, a)) { idx[col] = b; idx[ixj] = a; }
            }
            __syncthreads();
        }
    }
    if (col < n) ((int32_t *)(dst.data + i1 * dst.nb[1] + i2 * dst.nb[2] + i3 * dst.nb[3]))[col] = idx[col];
}

// one wave per row
__global__ void __launch_bounds__(256)
sum_rows_kernel(const T4 x, const T4 dst, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const char * xr = x.data + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
    float s = 0.0f;
    for (int64_t i = lane; i < x.ne[0]; i += 64) s += *(const float *)(xr + i * x.nb[0]);
    s = wave_sum(s);
    if (lane == 0) *(float *)(dst.data + i1 * dst.nb[1] + i2 * dst.nb[2] + i3 * dst.nb[3]) = s;
}

// f32 x f32 MUL_MAT for small weight matrices (the MoE router: 8 x 4096): one 256-thread workgroup per output element,
// 16-byte loads when both rows allow it (one wave per output read its 16 KB row in 64 dependent steps: 28 us)
__global__ void __launch_bounds__(256)
mul_mat_f32_kernel(const T4 a, const T4 b, const T4 d) {
    __shared__ float part[4];
    const int64_t o = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t m = o % d.ne[0], n = (o / d.ne[0]) % d.ne[1], i2 = (o / (d.ne[0] * d.ne[1])) % d.ne[2], i3 = o / (d.ne[0] * d.ne[1] * d.ne[2]);
    const float * ar = (const float *)(a.data + m * a.nb[1] + (i2 / (b.ne[2] / a.ne[2])) * a.nb[2] + (i3 / (b.ne[3] / a.ne[3])) * a.nb[3]);
    const float * br = (const float *)(b.data + n * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3]);
    const int64_t K = a.ne[0];
    float s = 0.0f;
    if ((((uintptr_t)ar | (uintptr_t)br) & 15) == 0) {
        const int64_t k4 = K / 4;
        for (int64_t i = tid; i < k4; i += 256) {
            const v4f x = ((const v4f *)ar)[i], y = ((const v4f *)br)[i];
            s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        for (int64_t k = k4 * 4 + tid; k < K; k += 256) s += ar[k] * br[k];
    } else {
        for (int64_t k = tid; k < K; k += 256) s += ar[k] * br[k];
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) *(float *)(d.data + m * d.nb[0] + n * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = part[0] + part[1] + part[2] + part[3];
}

// The whole router of a MoE layer for decode-sized batches in ONE launch (one workgroup per token):
//   logits = gate_inp . x, probs = softmax(logits), order = argsort(probs, desc), w = probs[order[0 .. n_used)],
//   wsum = sum(w), wn = w / wsum
// i.e. the node chain MUL_MAT SOFT_MAX ARGSORT GET_ROWS SUM_ROWS DIV of build_moe_ffn (llama-graph.cpp:660-720), six
// launches of ~4 us otherwise.  Every intermediate tensor of the chain is written, and each value is computed with
// the same operation order as the separate kernels above, so the results are bit-identical to the unfused graph.
// NORM: x is the layer's un-normalised feed-forward input; the workgroup first forms the row RMS_NORM(x) * norm_w exactly as
// rms_norm_kernel<true, false, 256> does (one 16-value chunk per thread, k <= 4096: the same per-thread sums, the same block
// reduction, (x * scale) * w), stores it (xn) with its q8_1 form (aq / ads, what the experts' mat-vecs read) and routes on it out of
// LDS -- the ffn norm launch of a MoE decode step disappears into the router's.
template <bool NORM>
__global__ void __launch_bounds__(256)
moe_router_kernel(const char * x, int64_t x_stride, const char * gate, int64_t gate_stride, int k, int n_expert, int n_used,
                  float * logits, float * probs, int32_t * order, float * w, float * wsum, float * wn,
                  const float * __restrict__ norm_w, float eps, float * __restrict__ xn, int8_t * __restrict__ aq, float * __restrict__ ads) {
    __shared__ float sh[4], p_s[256];
    __shared__ int ord_s[256];
    __shared__ __attribute__((aligned(16))) float xs[NORM ? 4096 : 4];
    const int tid = threadIdx.x, tok = blockIdx.x;
    const float * br = (const float *)(x + (int64_t)tok * x_stride);
    if (NORM) {
        const int nc = k >> 4, c = tid;
        v4f v[4], wv[4];
        float s = 0.0f;
        if (c < nc) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v[q] = ((const v4f *)br)[4 * c + q];
                wv[q] = ((const v4f *)norm_w)[4 * c + q];
                s += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
            }
        }
        s = block_sum_n<4>(s, sh);
        const float scale = 1.0f / sqrtf(s / (float)k + eps);
        if (c < nc) {
            float o[16];
            float * dr = xn + (int64_t)tok * k;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v4f t = v[q] * scale;
                t = t * wv[q];
                ((v4f *)dr)[4 * c + q] = t;
                ((v4f *)xs)[4 * c + q] = t;
                o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
            }
            if (aq) quant16_soa(o, aq + (int64_t)tok * k + 16 * c, ads + ((int64_t)tok * (k / QK) + (c >> 1)) * 2, (c & 1) == 0);
        }
        __syncthreads();
        br = xs;
    }
    // logits: as mul_mat_f32_kernel (256 threads per output, same per-thread and cross-wave summation order), eight
    // experts at a time so that their loads are in flight together (one expert after the other took 20 us)
    __shared__ float part8[8][4];
    const bool al16 = (((uintptr_t)gate | (uintptr_t)gate_stride | (uintptr_t)br) & 15) == 0;
    for (int e0 = 0; e0 < n_expert; e0 += 8) {
        float s[8];
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] = 0.0f;
        if (al16) {
            const int k4 = k / 4;
            for (int i = tid; i < k4; i += 256) {
                const v4f b = ((const v4f *)br)[i];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u < n_expert ? e0 + u : n_expert - 1;
                    const v4f a = ((const v4f *)(gate + (int64_t)e * gate_stride))[i];
                    s[u] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
                }
            }
            for (int i = k4 * 4 + tid; i < k; i += 256)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u < n_expert ? e0 + u : n_expert - 1;
                    s[u] += ((const float *)(gate + (int64_t)e * gate_stride))[i] * br[i];
                }
        } else {
            for (int i = tid; i < k; i += 256)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u < n_expert ? e0 + u : n_expert - 1;
                    s[u] += ((const float *)(gate + (int64_t)e * gate_stride))[i] * br[i];
                }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float r = wave_sum(s[u]);
            if ((tid & 63) == 0) part8[u][tid >> 6] = r;
        }
        __syncthreads();
        if (tid < 8 && e0 + tid < n_expert) {
            const float l = part8[tid][0] + part8[tid][1] + part8[tid][2] + part8[tid][3];
            logits[(int64_t)tok * n_expert + e0 + tid] = l; p_s[e0 + tid] = l;
        }
        __syncthreads();
    }
    // softmax: as soft_max_kernel (scale 1, no mask)
    const float v = tid < n_expert ? p_s[tid] * 1.0f : -INFINITY;
    const float mx = block_max(v, sh);
    const float ex = tid < n_expert ? expf(v - mx) : 0.0f;
    const float sum = block_sum(ex, sh);
    const float pr = ex * (1.0f / sum);
    __syncthreads();
    if (tid < n_expert) { p_s[tid] = pr; probs[(int64_t)tok * n_expert + tid] = pr; }
    __syncthreads();
    // argsort, descending, ties by index: position = number of entries that sort before this one
    if (tid < n_expert) {
        int rank = 0;
        for (int j = 0; j < n_expert; j++) rank += (p_s[j] > pr || (p_s[j] == pr && j < tid)) ? 1 : 0;
        ord_s[rank] = tid;
    }
    __syncthreads();
    if (tid < n_expert) order[(int64_t)tok * n_expert + tid] = ord_s[tid];
    // gather, sum (as sum_rows_kernel: one wave, lane-strided, butterfly), normalise
    if (tid < 64) {
        float s = 0.0f;
        for (int i = tid; i < n_used; i += 64) { const float wi = p_s[ord_s[i]]; w[(int64_t)tok * n_used + i] = wi; s += wi; }
        s = wave_sum(s);
        if (tid == 0) wsum[tok] = s;
        for (int i = tid; i < n_used; i += 64) wn[(int64_t)tok * n_used + i] = p_s[ord_s[i]] / s;
    }
}

} // namespace

// ---------------------------------------------------------------- prefetch (lab)
// reads `bytes` at p with 16-byte loads and drops them: pulls a weight range into the memory-side cache ahead of the launch that
// streams it (run on a second stream beside the previous launch).  The accumulated value is stored only if it equals a magic
// number, which keeps the loads alive.
__global__ void __launch_bounds__(256)
prefetch_kernel(const v4i * __restrict__ p, int64_t n16, int * sink) {
    v4i acc = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const v4i a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc = acc ^ a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc = acc ^ p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x5a17c3d9 && sink) *sink = acc.x;
}

extern "C" int mi355_prefetch(const void * p, int64_t bytes, int workgroups, void * stream) {
    if (!p) return MI355_E_NULL;
    if (bytes < 0 || workgroups < 1 || workgroups > 65535) return MI355_E_SHAPE;
    if ((uintptr_t)p & 15) return MI355_E_ALIGN;
    if (bytes < 16) return MI355_OK;
    MI355_LAUNCH("prefetch", prefetch_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (const v4i *)p, bytes / 16, (int *)nullptr);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_argsort_f32(const mi355_tensor * x, const mi355_tensor * dst, int descending, void * stream) {
    if (!x || !dst || !x->data || !dst->data) return MI355_E_NULL;
    if (x->type != MI355_TYPE_F32 || x->nb[0] != 4 || dst->nb[0] != 4 || !same_shape(x, dst)) return MI355_E_SHAPE;
    if (x->ne[0] > 1024) return MI355_E_UNSUPPORTED;
    const int64_t rows = nrows(x);
    if (rows == 0 || x->ne[0] == 0) return MI355_OK;
    if (rows > 0x7fffffff) return MI355_E_SHAPE;
    int npad = 1;
    while (npad < x->ne[0]) npad <<= 1;
    MI355_LAUNCH("argsort", argsort_kernel, dim3((unsigned)rows), dim3(npad < 64 ? 64 : npad), 0, (hipStream_t)stream, t4(x), t4(dst), npad, descending);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_sum_rows_f32(const mi355_tensor * x, const mi355_tensor * dst, void * stream) {
    if (!x || !dst || !x->data || !dst->data) return MI355_E_NULL;
    if (x->type != MI355_TYPE_F32 || dst->ne[0] != 1 || dst->ne[1] != x->ne[1] || dst->ne[2] != x->ne[2] || dst->ne[3] != x->ne[3]) return MI355_E_SHAPE;
    const int64_t rows = nrows(x);
    if (rows == 0) return MI355_OK;
    if ((rows + 3) / 4 > 0x7fffffff) return MI355_E_SHAPE;
    MI355_LAUNCH("sum_rows", sum_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, t4(x), t4(dst), rows);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_mul_mat_f32(const mi355_tensor * a, const mi355_tensor * b, const mi355_tensor * dst, void * stream) {
    if (!a || !b || !dst || !a->data || !b->data || !dst->data) return MI355_E_NULL;
    if (a->type != MI355_TYPE_F32 || b->type != MI355_TYPE_F32) return MI355_E_TYPE;
    if (a->nb[0] != 4 || b->nb[0] != 4 || a->ne[0] != b->ne[0] || dst->ne[0] != a->ne[1] || dst->ne[1] != b->ne[1] || dst->ne[2] != b->ne[2] ||
        dst->ne[3] != b->ne[3] || a->ne[2] < 1 || a->ne[3] < 1 || b->ne[2] % a->ne[2] != 0 || b->ne[3] % a->ne[3] != 0) return MI355_E_SHAPE;
    const int64_t total = nelem(dst);
    if (total == 0) return MI355_OK;
    if (total > 0x7fffffff) return MI355_E_SHAPE;
    MI355_LAUNCH("mul_mat_f32", mul_mat_f32_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, t4(a), t4(b), t4(dst));
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_moe_router_f32(const float * x, int64_t x_row_stride, const float * gate_inp, int64_t gate_row_stride, int64_t k,
                                    int64_t n_expert, int64_t n_used, int64_t n_tokens, float * logits, float * probs, int32_t * order,
                                    float * w, float * wsum, float * wn, void * stream) {
    if (!x || !gate_inp || !logits || !probs || !order || !w || !wsum || !wn) return MI355_E_NULL;
    if (k <= 0 || k > 0x7fffffff || n_expert < 1 || n_expert > 256 || n_used < 1 || n_used > n_expert || n_tokens < 0 || n_tokens > 0x7fffffff) return MI355_E_SHAPE;
    if ((x_row_stride | gate_row_stride) & 3) return MI355_E_ALIGN;
    if (n_tokens == 0) return MI355_OK;
    MI355_LAUNCH("moe_router", moe_router_kernel<false>, dim3((unsigned)n_tokens), dim3(256), 0, (hipStream_t)stream, (const char *)x, x_row_stride,
                 (const char *)gate_inp, gate_row_stride, (int)k, (int)n_expert, (int)n_used, logits, probs, order, w, wsum, wn,
                 (const float *)nullptr, 0.0f, (float *)nullptr, (int8_t *)nullptr, (float *)nullptr);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_moe_router_norm_f32(const float * x, int64_t x_row_stride, const float * norm_w, float eps, float * xn, int8_t * act_q,
                                         float * act_ds, const float * gate_inp, int64_t gate_row_stride, int64_t k, int64_t n_expert,
                                         int64_t n_used, int64_t n_tokens, float * logits, float * probs, int32_t * order, float * w,
                                         float * wsum, float * wn, void * stream) {
    if (!x || !norm_w || !xn || !gate_inp || !logits || !probs || !order || !w || !wsum || !wn) return MI355_E_NULL;
    if ((act_q == nullptr) != (act_ds == nullptr)) return MI355_E_NULL;
    if (k <= 0 || n_expert < 1 || n_expert > 256 || n_used < 1 || n_used > n_expert || n_tokens < 0 || n_tokens > 0x7fffffff) return MI355_E_SHAPE;
    if (k > 4096 || k % 32 != 0) return MI355_E_UNSUPPORTED;        // one 16-value chunk per thread, whole q8_1 blocks per lane pair
    if (((uintptr_t)x | (uintptr_t)x_row_stride | (uintptr_t)norm_w | (uintptr_t)xn | (uintptr_t)act_q) & 15) return MI355_E_ALIGN;
    if (((uintptr_t)act_ds & 7) || (gate_row_stride & 3)) return MI355_E_ALIGN;
    if (n_tokens == 0) return MI355_OK;
    MI355_LAUNCH("moe_router_norm", moe_router_kernel<true>, dim3((unsigned)n_tokens), dim3(256), 0, (hipStream_t)stream, (const char *)x, x_row_stride,
                 (const char *)gate_inp, gate_row_stride, (int)k, (int)n_expert, (int)n_used, logits, probs, order, w, wsum, wn,
                 norm_w, eps, xn, act_q, act_ds);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// The experts' results of a batch combined: weighted[t][s][:] = e[t][s][:] * w[t][s], sum[t][:] = weighted[t][0][:] + weighted[t][1][:]
// -- the MUL and the ADD behind ffn_moe_down (build_moe_ffn, src/llama-graph.cpp:773-790) as one pass over the rows, each value
// with the one f32 multiply / add the two launches would have done.
__global__ __launch_bounds__(256) void moe_combine2_kernel(const float4 * __restrict__ e, const char * __restrict__ w, int64_t w_token_stride,
                                                           int64_t w_slot_stride, float4 * weighted, float4 * __restrict__ sum, int m4) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= m4) return;
    const int64_t t = blockIdx.y;
    const float w0 = *(const float *)(w + t * w_token_stride), w1 = *(const float *)(w + t * w_token_stride + w_slot_stride);
    const float4 a = e[(2 * t) * m4 + c], b = e[(2 * t + 1) * m4 + c];
    const float4 p0 = make_float4(a.x * w0, a.y * w0, a.z * w0, a.w * w0), p1 = make_float4(b.x * w1, b.y * w1, b.z * w1, b.w * w1);
    weighted[(2 * t) * m4 + c] = p0;
    weighted[(2 * t + 1) * m4 + c] = p1;
    sum[t * m4 + c] = make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
}

extern "C" int mi355_moe_combine2_f32(const float * e, const float * w, int64_t w_token_stride, int64_t w_slot_stride, float * weighted,
                                      float * sum, int64_t m, int64_t n_tokens, void * stream) {
    if (!e || !w || !weighted || !sum) return MI355_E_NULL;
    if (m <= 0 || m > 0x7fffffff || n_tokens < 0 || n_tokens > 65535) return MI355_E_SHAPE;
    if (m % 4 != 0) return MI355_E_UNSUPPORTED;
    if ((((uintptr_t)e | (uintptr_t)weighted | (uintptr_t)sum) & 15) || ((w_token_stride | w_slot_stride | (int64_t)(uintptr_t)w) & 3)) return MI355_E_ALIGN;
    if (n_tokens == 0) return MI355_OK;
    const int m4 = (int)(m / 4);
    MI355_LAUNCH("moe_combine2", moe_combine2_kernel, dim3((unsigned)((m4 + 255) / 256), (unsigned)n_tokens), dim3(256), 0, (hipStream_t)stream,
                 (const float4 *)e, (const char *)w, w_token_stride, w_slot_stride, (float4 *)weighted, (float4 *)sum, m4);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}
