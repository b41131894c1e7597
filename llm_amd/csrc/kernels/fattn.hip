// fattn.hip -- FLASH_ATTN_EXT with f16 / q8_0 / q4_0 KV cache for gfx950.
//
// Contract: ggml.h:2012-2033, ggml.c:4733-4786; oracle ggml_compute_forward_flash_attn_ext_f16
// (ggml/src/ggml-cpu/ops.cpp:7975-8192).  Replaces ggml_cuda_flash_attn_ext (ggml-cuda/fattn.cu:271-346,
// vec kernels fattn-vec-f32.cuh) -- not derived from them.
//
// Decode-shaped kernel (HBM-bound: K and V rows are streamed once):
//   * one 512-thread workgroup per (kv head, query token, batch) and a slice of the KV range;
//     all `R` query heads that share the kv head (GQA) are processed together, so K/V are read once.
//   * LPR = D/8 lanes cooperate on one KV row (each lane owns 8 consecutive features: one 128-bit
//     load of f16, or 8 bytes of q8_0), so a wave-load covers 64/LPR complete, contiguous rows.
//   * every lane group runs its own online softmax (running max / sum / 8-wide f32 accumulator per
//     query head) over rows g, g+G, ...; the G groups are merged through LDS at the end.
//   * the KV range is split over `nsplit` workgroups that write (m, l, acc) partials to a workspace and meet
//     at an arrival counter; the workgroup that arrives last merges them (same rescaling rule) -- one launch.
// All accumulation is f32 (the CPU oracle accumulates f16 V in fp16; NMSE budget 5e-4).
#include "common.h"
#include <mutex>

namespace {

struct FArgs {
    const char * q; int64_t q_nb1, q_nb2, q_nb3;
    const char * k; int64_t k_nb1, k_nb2, k_nb3;
    const char * v; int64_t v_nb1, v_nb2, v_nb3;
    const char * mask; int64_t m_nb1, m_nb2, m_nb3, m_ne2, m_ne3;
    char * dst; int64_t d_nb1, d_nb2, d_nb3;
    int n_q, n_kv, n_head, n_head_kv, ne3;
    int dv;  // V's head size (<= the kernels' D like d; DeepSeek's 192 / 128)
    int d;   // head size.  The kernels' D is the next size they are built for (vec: 64 / 128 / 256, matrix cores: a multiple of 32);
             // lanes and tile columns past d hold zeros and store nothing
    float scale, max_bias, logit_softcap, m0, m1;
    uint32_t n_head_log2;
    int nsplit;
    float * part; // [ne3][n_q][n_head][nsplit][D + 2] when nsplit > 1
    int * counters; // [ne3][n_q][gridDim.x] arrival counters of the KV slices (all zero between launches)
    int8_t * act_q; float * act_ds; // optional: q8_1 form of dst (token row = all heads), written with the result
    uint16_t * act_bf;              // optional (matrix-core kernel): bf16 form of dst, rows [ne3][n_q][n_head * d]
    int kvs;                        // matrix-core kernel: KV blocks dealt to kvs workgroups per unit group (blockIdx.z % kvs), partials merged after
#ifdef FA_STAMPS
    long long * stamps;             // diagnostic build only (lab/fattn_stamps.hip): [workgroup][8] s_memtime / s_memrealtime
#endif
};

#ifdef FA_STAMPS
#define FA_STAMP(i) do { if (threadIdx.x == 0) st_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define FA_STAMP_FLUSH() do { if (threadIdx.x == 0 && a.stamps) { st_[6] = __builtin_amdgcn_s_memrealtime(); \
        long long * o_ = a.stamps + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8; for (int i_ = 0; i_ < 8; i_++) o_[i_] = st_[i_]; } } while (0)
#else
#define FA_STAMP(i) do { } while (0)
#define FA_STAMP_FLUSH() do { } while (0)
#endif

// 8 consecutive features of a K/V row as f32
template <int TYPE>
static __device__ __forceinline__ void load8(const char * row, int j, float * out) {
    if (TYPE == MI355_TYPE_F16) {
        const v4i raw = *(const v4i *)(row + j * 16);
        const __half2 * h = (const __half2 *)&raw;
#pragma unroll
        for (int i = 0; i < 4; i++) { const float2 f = __half22float2(h[i]); out[2 * i] = f.x; out[2 * i + 1] = f.y; }
    } else if (TYPE == MI355_TYPE_BF16) {              // bf16 = the upper half of the f32 (ggml_compute_bf16_to_fp32, ggml-impl.h)
        const v4i raw = *(const v4i *)(row + j * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) { out[2 * i] = __int_as_float(raw[i] << 16); out[2 * i + 1] = __int_as_float(raw[i] & 0xffff0000); }
    } else if (TYPE == MI355_TYPE_Q8_0) {
        const char * blk = row + (j >> 2) * 34;
        const float d = h2f(*(const uint16_t *)blk);
        const int8_t * qs = (const int8_t *)(blk + 2 + (j & 3) * 8);
        int w0, w1;
        __builtin_memcpy(&w0, qs, 4); __builtin_memcpy(&w1, qs + 4, 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[i]     = __fmul_rn((float)(int8_t)(w0 >> (8 * i)), d);
            out[4 + i] = __fmul_rn((float)(int8_t)(w1 >> (8 * i)), d);
        }
    } else { // Q4_0: 18 B blocks; element e of the block: e < 16 -> low nibble of qs[e], else high nibble of qs[e-16]
        const char * blk = row + (j >> 2) * 18;
        const float d = h2f(*(const uint16_t *)blk);
        const int e0 = (j & 3) * 8;             // 0, 8, 16, 24
        const uint8_t * qs = (const uint8_t *)(blk + 2 + (e0 & 15));
        const int sh = e0 >= 16 ? 4 : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = __fmul_rn((float)(((qs[i] >> sh) & 0xF) - 8), d);
    }
}

// sum over the 16 lanes of a DPP row, result in every lane: four rotate-and-add steps on the VALU (v_add_f32 row_ror)
// instead of four ds_bpermute round trips through the LDS crossbar
template <int CTRL> static __device__ __forceinline__ float dpp_rot(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ float row16_sum(float v) {
    v += dpp_rot<0x128>(v);      // row_ror:8
    v += dpp_rot<0x124>(v);      // row_ror:4
    v += dpp_rot<0x122>(v);      // row_ror:2
    v += dpp_rot<0x121>(v);      // row_ror:1
    return v;
}

// One lane's 8 consecutive features of a K or V row.  f16 rows stay packed (4 VGPRs): the K dot product then runs on
// v_dot2_f32_f16 against the query rounded to f16 -- which is what the CPU oracle does too (it converts Q to the K
// type's vec_dot type, ops.cpp:8061-8066) -- and V feeds the mixed-precision FMA (f16 operand, f32 accumulate) with no
// conversion instruction.  Quantized rows are dequantized to f32 when loaded.
typedef _Float16 fa_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 fa_half2 __attribute__((ext_vector_type(2)));
template <int TYPE> struct FaRow {
    float f[8];
    __device__ __forceinline__ void load(const char * row, int j) { load8<TYPE>(row, j, f); }
    __device__ __forceinline__ float get(int i) const { return f[i]; }
};
template <> struct FaRow<MI355_TYPE_F16> {
    fa_half8 h;
    __device__ __forceinline__ void load(const char * row, int j) { h = *(const fa_half8 *)(row + j * 16); }
    __device__ __forceinline__ float get(int i) const { return (float)h[i]; }
};
// the query features of one head in the form the K type's dot product wants
template <int TK> struct FaQ {
    float f[8];
    __device__ __forceinline__ void set(const v4f q0, const v4f q1) { f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w; }
    __device__ __forceinline__ float dot(const FaRow<TK> & k) const {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) t += f[i] * k.f[i];
        return t;
    }
};
// bf16 K: the query is rounded to bf16 first, as the CPU path does (Q is converted to the K type's vec_dot type, ops.cpp:8061-8066)
template <> struct FaQ<MI355_TYPE_BF16> {
    float f[8];
    static __device__ __forceinline__ float r(float x) { return (float)(__bf16)x; }
    __device__ __forceinline__ void set(const v4f q0, const v4f q1) { f[0] = r(q0.x); f[1] = r(q0.y); f[2] = r(q0.z); f[3] = r(q0.w); f[4] = r(q1.x); f[5] = r(q1.y); f[6] = r(q1.z); f[7] = r(q1.w); }
    __device__ __forceinline__ float dot(const FaRow<MI355_TYPE_BF16> & k) const {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) t += f[i] * k.f[i];
        return t;
    }
};
template <> struct FaQ<MI355_TYPE_F16> {
    fa_half8 h;
    __device__ __forceinline__ void set(const v4f q0, const v4f q1) {
        h[0] = (_Float16)q0.x; h[1] = (_Float16)q0.y; h[2] = (_Float16)q0.z; h[3] = (_Float16)q0.w;
        h[4] = (_Float16)q1.x; h[5] = (_Float16)q1.y; h[6] = (_Float16)q1.z; h[7] = (_Float16)q1.w;
    }
    __device__ __forceinline__ float dot(const FaRow<MI355_TYPE_F16> & k) const {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const fa_half2 a = {h[2 * i], h[2 * i + 1]}, b = {k.h[2 * i], k.h[2 * i + 1]};
            t = __builtin_amdgcn_fdot2(a, b, t, false);
        }
        return t;
    }
};

static __device__ __forceinline__ float alibi_slope(const FArgs & a, uint32_t h) {
    if (!(a.max_bias > 0.0f)) return 1.0f;
    return h < a.n_head_log2 ? powf(a.m0, h + 1) : powf(a.m1, 2 * (h - a.n_head_log2) + 1);
}

// one output element (head h, token iq, feature d) and, if asked for, its q8_1 form: the 32 lanes that hold a
// 32-block (d / 32 of one head; lanes are consecutive in d) quantize it together exactly as quantize_row_q8_1_ref
template <int D>
static __device__ __forceinline__ void fattn_store(const FArgs & a, int h, int iq, int i3, int d, float v) {
    float * o = (float *)(a.dst + (int64_t)h * a.d_nb1 + (int64_t)iq * a.d_nb2 + (int64_t)i3 * a.d_nb3);
    o[d] = v;
    if (a.act_q) {
        float amax = fabsf(v);
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
        const float dd = __fdiv_rn(amax, 127.0f);
        const float id = dd != 0.0f ? __fdiv_rn(1.0f, dd) : 0.0f;
        const float r = roundf(__fmul_rn(v, id));
        const int q = fabsf(r) < 2147483648.0f ? (int)(int8_t)(int)r : 0;
        int sum = q;
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) sum += __shfl_xor(sum, s, 64);
        const int64_t row = (int64_t)i3 * a.n_q + iq, kk = (int64_t)a.n_head * a.dv, e = (int64_t)h * a.dv + d;
        a.act_q[row * kk + e] = (int8_t)q;
        if ((d & 31) == 0) { v2f ds; ds.x = __half2float(__float2half_rn(dd)); ds.y = (float)sum; *(v2f *)(a.act_ds + (row * (kk / 32) + e / 32) * 2) = ds; }
    }
}

constexpr int FA_THREADS = 512; // 8 waves: the KV loop is a chain of dependent loads, so rows are spread wide
                                 // (1024 threads would cap the kernel at 128 VGPRs and spill for R >= 4)

// U: KV rows per lane group and step.  2: the pipelined loop for long ranges (the rows of step i + 1 are requested before step i is
// consumed).  8: short contexts in ONE workgroup per head group -- a step requests 8 rows per lane group (256 rows at D = 128) at
// once, so n_kv <= 512 is one or two memory round trips with no KV split, no partials, no arrival counter and no slice merge.
template <int D, int R, int TK, int TV, int U = 2>
__global__ void __launch_bounds__(FA_THREADS)
fattn_vec_kernel(const FArgs a) {
    constexpr int LPR = D / 8;               // lanes per KV row
    constexpr int G   = FA_THREADS / LPR;    // row groups per workgroup
    constexpr int NW  = FA_THREADS / 64;     // waves
    extern __shared__ __attribute__((aligned(16))) float sm[]; // [R][G][D + 2], [R][G], [R][2]
    const int tid = threadIdx.x;
#ifdef FA_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st_[7] = __builtin_amdgcn_s_memrealtime();
#endif
    FA_STAMP(0);
    const int j = tid % LPR, g = tid / LPR;
    const bool live = j * 8 < a.d;                // (head sizes 40 / 72 / 80 / 96 / 112: the lanes past the row work on zeros)
    const int jl = live ? j : 0, jv = j * 8 < a.dv ? j : 0;
    const int ratio = a.n_head / a.n_head_kv;
    const int hgroups = (ratio + R - 1) / R;      // workgroups per kv head
    const int hk = blockIdx.x / hgroups;          // kv head
    const int h0 = hk * ratio + (blockIdx.x % hgroups) * R;
    const int iq = blockIdx.y;
    const int i3 = blockIdx.z / a.nsplit, split = blockIdx.z % a.nsplit;

    const char * kbase = a.k + hk * a.k_nb2 + i3 * a.k_nb3;
    const char * vbase = a.v + hk * a.v_nb2 + i3 * a.v_nb3;
    // KV slice of this workgroup
    const int per = (a.n_kv + a.nsplit - 1) / a.nsplit;
    const int kv0 = split * per, kv1 = kv0 + per < a.n_kv ? kv0 + per : a.n_kv;
    const float scale = a.logit_softcap != 0.0f ? a.scale / a.logit_softcap : a.scale;
    const char * mrow[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int h = h0 + r < hk * ratio + ratio ? h0 + r : hk * ratio + ratio - 1; // clamp (ratio % R != 0)
        mrow[r] = a.mask ? a.mask + iq * a.m_nb1 + (h % a.m_ne2) * a.m_nb2 + (i3 % a.m_ne3) * a.m_nb3 : nullptr;
    }

    // A step is U rows per lane group (rows ic, ic + G, ...): K, V and the mask values of a step are requested together, and
    // the first step's before anything else -- the query loads, which were written by the launch just before this one, then
    // overlap the first K/V round trip instead of preceding it.
    // PIPE (U = 2): software pipeline, the rows of step i + 1 are requested before step i is consumed: twice the loads in flight
    // per round trip, and one rescale of the running sums per two rows.
    constexpr bool PIPE = U == 2;
    FaRow<TK> kf[U], kn[PIPE ? U : 1];
    FaRow<TV> vf[U], vn[PIPE ? U : 1];
    __half mk[R][U], mkn[PIPE ? R : 1][PIPE ? U : 1];
    auto request = [&](int at, FaRow<TK> * kd, FaRow<TV> * vd, __half (*md)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int iu = at + u * G < kv1 ? at + u * G : (at < kv1 ? at : kv1 - 1);     // past the end: re-read a valid row, ignored below
            kd[u].load(kbase + (int64_t)iu * a.k_nb1, jl);
            vd[u].load(vbase + (int64_t)iu * a.v_nb1, jv);
#pragma unroll
            for (int r = 0; r < R; r++) md[r][u] = mrow[r] ? ((const __half *)mrow[r])[iu] : __half(0.0f);
        }
    };
    int ic = kv0 + g;
    if (kv0 < kv1) request(ic, kf, vf, mk);
    FA_STAMP(1);

    // this lane's 8 query features for each of the R heads
    FaQ<TK> qf[R];
    float slope[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int h = h0 + r < hk * ratio + ratio ? h0 + r : hk * ratio + ratio - 1; // clamp (ratio % R != 0)
        const float * qp = (const float *)(a.q + iq * a.q_nb1 + h * a.q_nb2 + i3 * a.q_nb3) + jl * 8;
        v4f q0 = *(const v4f *)qp, q1 = *(const v4f *)(qp + 4);
        if (!live) { q0 = v4f{0.0f, 0.0f, 0.0f, 0.0f}; q1 = q0; }     // a zero query: whatever K row bytes the lane re-read add nothing
        qf[r].set(q0, q1);
        slope[r] = alibi_slope(a, (uint32_t)h);
    }
    float m[R], l[R], acc[R][8];
#pragma unroll
    for (int r = 0; r < R; r++) {
        m[r] = -INFINITY; l[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) acc[r][i] = 0.0f;
    }

    for (; ic < kv1; ic += U * G) {
        if constexpr (PIPE) request(ic + U * G, kn, vn, mkn);
        float mv[R][U];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int iu = ic + u * G;
                mv[r][u] = iu >= kv1 ? -INFINITY : (mrow[r] ? slope[r] * __half2float(mk[r][u]) : 0.0f);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            float s[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                float t = qf[r].dot(kf[u]);
                if constexpr (LPR == 16) t = row16_sum(t);
                else {
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) t += __shfl_xor(t, o, 64);
                }
                t *= scale;
                if (a.logit_softcap != 0.0f) t = a.logit_softcap * tanhf(t);
                s[u] = mv[r][u] == -INFINITY ? -INFINITY : t + mv[r][u];   // masked out (ops.cpp:8105-8108) or past the end
            }
            float mn = m[r];
#pragma unroll
            for (int u = 0; u < U; u++) mn = fmaxf(mn, s[u]);
            if (mn == -INFINITY) continue;          // nothing seen yet and both rows masked; uniform inside the lane group
            const float corr = __expf(m[r] - mn);   // m == -inf: 0.  (v_exp_f32: relative error ~1e-6, far inside the op's budget)
            float p[U];
#pragma unroll
            for (int u = 0; u < U; u++) p[u] = __expf(s[u] - mn);
            m[r] = mn;
            float ls = l[r] * corr;
#pragma unroll
            for (int u = 0; u < U; u++) ls += p[u];
            l[r] = ls;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float t = acc[r][i] * corr;
#pragma unroll
                for (int u = 0; u < U; u++) t = __builtin_fmaf(p[u], vf[u].get(i), t);
                acc[r][i] = t;
            }
        }
        if constexpr (PIPE) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                kf[u] = kn[u]; vf[u] = vn[u];
#pragma unroll
                for (int r = 0; r < R; r++) mk[r][u] = mkn[r][u];
            }
        } else {
            if (ic + U * G < kv1) request(ic + U * G, kf, vf, mk);
        }
    }

    FA_STAMP(2);
    // merge of the G lane groups through LDS: every group stores its (acc, m, l); one wave per head turns the G maxima
    // into weights exp(m_g - M) and the total L; then one thread per (head, feature) adds up the weighted accumulators.
    // (A butterfly of xor-shuffles over the groups of a wave followed by a merge of the waves cost 2.2 us here -- 40
    // ds_bpermute and 16 expf in dependent chains; this is two barriers and 2 x G LDS reads per output.)
    float * wsm = sm + (size_t)R * G * (D + 2);          // [R][G] weights, then [R][2] {M, L}
    float * mlsm = wsm + R * G;
#pragma unroll
    for (int r = 0; r < R; r++) {
        float * slot = sm + ((size_t)r * G + g) * (D + 2);
        *(v4f *)(slot + j * 8)     = v4f{acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
        *(v4f *)(slot + j * 8 + 4) = v4f{acc[r][4], acc[r][5], acc[r][6], acc[r][7]};
        if (j == 0) { slot[D] = m[r]; slot[D + 1] = l[r]; }
    }
    __syncthreads();
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int r = wave; r < R; r += NW) {
            const float * slot = sm + ((size_t)r * G + (lane < G ? lane : 0)) * (D + 2);
            const float mg = lane < G ? slot[D] : -INFINITY, lg = lane < G ? slot[D + 1] : 0.0f;
            float M = mg;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
            const float w = mg == -INFINITY ? 0.0f : __expf(mg - M);
            const float L = wave_sum(lg * w);
            if (lane < G) wsm[r * G + lane] = w;
            if (lane == 0) { mlsm[2 * r] = M; mlsm[2 * r + 1] = L; }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < R * D; idx += FA_THREADS) {
        const int r = idx / D, d = idx % D;
        const int h = h0 + r;
        if (h >= hk * ratio + ratio || d >= a.dv) continue;
        const float * base = sm + (size_t)r * G * (D + 2) + d;
        const float * wr = wsm + r * G;
        float O = 0.0f;
#pragma unroll 8
        for (int gg = 0; gg < G; gg++) O = __builtin_fmaf(base[(size_t)gg * (D + 2)], wr[gg], O);
        const float M = mlsm[2 * r], L = mlsm[2 * r + 1];
        if (a.nsplit == 1) {
            fattn_store<D>(a, h, iq, i3, d, O / L);
        } else {
            float * p = a.part + ((((size_t)i3 * a.n_q + iq) * a.n_head + h) * a.nsplit + split) * (D + 2);
            p[d] = O;
            if (d == 0) { p[D] = M; p[D + 1] = L; }
        }
    }
    FA_STAMP(3);
    if (a.nsplit == 1) { FA_STAMP_FLUSH(); return; }

    // The slices of one (kv-head group, token) meet at an arrival counter; whoever arrives last merges all partials
    // and writes the result, so there is no second launch.  Hand-off (one release and one acquire per workgroup, not
    // per wave -- a device-scope fence writes back / invalidates caches and costs microseconds): the barrier waits
    // for every wave's partial stores, then ONE lane counts the arrival with an acq_rel device-scope atomic, which
    // publishes the workgroup's stores and, in the merging workgroup, drops this CU's stale L1 lines.
    // Ordering: every wave first waits for its OWN partial stores to have left the CU (s_waitcnt vmcnt(0): the barrier's
    // workgroup-scope fence emits no vmcnt wait on gfx9 without tgsplit, and s_barrier waits on no counter), then the
    // barrier, then thread 0's agent-scope release (L2 write-back) and the counter add -- so all waves' partials are
    // ordered before the increment that the merging workgroup, possibly on another XCD, acquires.
    __shared__ int last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int * cnt = a.counters + ((size_t)i3 * a.n_q + iq) * gridDim.x + blockIdx.x;
        // release only: the acquire half (cache invalidation) is paid by the one workgroup that goes on to merge
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == a.nsplit - 1;
        if (last_flag) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
        }
    }
    __syncthreads();
    FA_STAMP(4);
    if (!last_flag) { FA_STAMP_FLUSH(); return; }
    for (int idx = tid; idx < R * D; idx += FA_THREADS) {
        const int r = idx / D, d = idx % D;
        const int h = h0 + r;
        if (h >= hk * ratio + ratio || d >= a.dv) continue;
        const float * p = a.part + (((size_t)i3 * a.n_q + iq) * a.n_head + h) * a.nsplit * (D + 2);
        // eight slices at a time with all their loads issued before the first use: the partials sit in L2 / HBM and
        // a load-use-load chain over the slices would cost one memory round trip per slice
        float M = -INFINITY, L = 0.0f, O = 0.0f;
        for (int s0 = 0; s0 < a.nsplit; s0 += 8) {
            float mv[8], lv[8], ov[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int sc = s0 + u < a.nsplit ? s0 + u : a.nsplit - 1;
                mv[u] = p[sc * (D + 2) + D]; lv[u] = p[sc * (D + 2) + D + 1]; ov[u] = p[sc * (D + 2) + d];
            }
            float mb = M;
#pragma unroll
            for (int u = 0; u < 8; u++) if (s0 + u < a.nsplit) mb = fmaxf(mb, mv[u]);
            const float c0 = M == -INFINITY ? 0.0f : expf(M - mb);
            L *= c0; O *= c0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float w = (s0 + u < a.nsplit && mv[u] != -INFINITY) ? expf(mv[u] - mb) : 0.0f;
                L += lv[u] * w; O += ov[u] * w;
            }
            M = mb;
        }
        fattn_store<D>(a, h, iq, i3, d, O / L);
    }
    FA_STAMP(5);
    FA_STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// Prefill: tile kernel on the f16 matrix cores (v_mfma_f32_32x32x16_f16); f16 K/V, or q8_0 / q4_0 K/V dequantized
// to f16 while a block is staged; ALiBi slopes and the logit softcap as the vec kernel has them.
//   workgroup = 256 threads = 4 waves; a wave owns 32 query rows of one head and the 4 waves take up to 4 heads of one KV
//   head (or further 32-query blocks), so the KV blocks of 64 rows the workgroup walks are staged in LDS once for all of
//   them; K and V both row-major (coalesced 16-byte loads, 16-byte LDS stores).
//   S^T[kv][q] = K Q^T is computed transposed on purpose: the accumulator layout then puts one query per
//   lane (column) and 16 kv rows in its registers, so the online-softmax row reductions are in-register
//   plus one cross-half shuffle, and exp(S^T) packed to f16 is already the B operand of O^T = V^T P^T.
//   The kv order inside a 16-wide MFMA step is whatever the accumulator layout delivers,
//       kv(t, s, half, j) = 32 t + (j & 3) + 8 (2 s + (j >> 2)) + 4 half,
//   and the V^T operand is gathered from the row-major tile in exactly that order with ds_read_b64_tr_b16.  A block whose
//   mask tile is entirely -inf for the workgroup (the upper triangle of causal prefill) is not even staged: masks are read
//   and voted on one block ahead, and the K/V loads of the next block fly during the MFMAs of the current one.
// ------------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int MMA_BKV = 64, MMA_WAVES = 4, MMA_THREADS = 64 * MMA_WAVES;

// 8 consecutive features of a K/V row on their way into the LDS tile: `load` only requests the bytes (so the loads of the
// next block can fly during the current one), `f16` turns them into 8 halfs when the tile is written -- f16 rows are
// copied, q8_0 / q4_0 rows are dequantized (d * q in f32, as the vec kernel does) and rounded to f16.
template <int TYPE> struct MmaChunk {
    v4i raw;
    __device__ __forceinline__ void load(const char * row, int j) { raw = *(const v4i *)(row + j * 16); }
    __device__ __forceinline__ v4i f16() const { return raw; }
};
template <> struct MmaChunk<MI355_TYPE_Q8_0> {
    int w0, w1; uint16_t d;
    __device__ __forceinline__ void load(const char * row, int j) {
        const char * blk = row + (j >> 2) * 34;
        d = *(const uint16_t *)blk;
        const int8_t * qs = (const int8_t *)(blk + 2 + (j & 3) * 8);
        __builtin_memcpy(&w0, qs, 4); __builtin_memcpy(&w1, qs + 4, 4);
    }
    __device__ __forceinline__ v4i f16() const {
        const float dd = h2f(d);
        half8 h;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[i]     = (_Float16)__fmul_rn((float)(int8_t)(w0 >> (8 * i)), dd);
            h[4 + i] = (_Float16)__fmul_rn((float)(int8_t)(w1 >> (8 * i)), dd);
        }
        return *(const v4i *)&h;
    }
};
template <> struct MmaChunk<MI355_TYPE_Q4_0> {
    int w0, w1; uint16_t d; int sh;
    __device__ __forceinline__ void load(const char * row, int j) {
        const char * blk = row + (j >> 2) * 18;
        d = *(const uint16_t *)blk;
        const int e0 = (j & 3) * 8;
        const uint8_t * qs = (const uint8_t *)(blk + 2 + (e0 & 15));
        sh = e0 >= 16 ? 4 : 0;
        __builtin_memcpy(&w0, qs, 4); __builtin_memcpy(&w1, qs + 4, 4);
    }
    __device__ __forceinline__ v4i f16() const {
        const float dd = h2f(d);
        half8 h;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[i]     = (_Float16)__fmul_rn((float)((((uint32_t)w0 >> (8 * i + sh)) & 0xF) - 8.0f), dd);
            h[4 + i] = (_Float16)__fmul_rn((float)((((uint32_t)w1 >> (8 * i + sh)) & 0xF) - 8.0f), dd);
        }
        return *(const v4i *)&h;
    }
};

// SPLIT: for launches of fewer (head, 32-query) units than the chip has SIMDs -- every prefill ubatch of 512 tokens: 32 heads x 16
// blocks = 512 -- a unit is shared by TWO waves that take alternate 64-row KV blocks (a workgroup stages 128 rows per step for 2 units x
// 2 halves) and merge their (m, l, O) through LDS at the end: twice the workgroups, half the chain of KV steps per wave.
template <int D, int TKV, bool SPLIT = false>
__global__ void __launch_bounds__(MMA_THREADS, SPLIT || D > 128 ? 1 : 2)
fattn_mma_kernel(const FArgs a) {
    constexpr int BKV = SPLIT ? 2 * MMA_BKV : MMA_BKV;      // KV rows staged per step
    constexpr int UNITS = SPLIT ? MMA_WAVES / 2 : MMA_WAVES;
    constexpr int KP = D * 2 + 16;          // K tile row pitch (bytes)
    constexpr int VP = D * 2 + ((D / 32) % 2 == 1 ? 0 : 64);   // V tile row pitch (bytes), 16 or 48 dwords mod 64: 4 consecutive rows land on 4 disjoint groups of 16 banks, so
                                            // the transposed reads below (4 rows x 64 B per 32-lane half) are conflict-free
    constexpr int DT = D / 32, KS = D / 16;
    __shared__ __attribute__((aligned(16))) uint8_t Ks[BKV * KP];
    __shared__ __attribute__((aligned(16))) uint8_t Vs[BKV * VP];   // row-major like K; consumed column-wise with ds_read_b64_tr_b16
    const int tid = threadIdx.x, lane = tid & 63, wave_ = tid >> 6;
    const int wave = SPLIT ? wave_ % UNITS : wave_;           // the (head, query block) unit of this wave inside the workgroup
    const int koff = SPLIT ? (wave_ / UNITS) * MMA_BKV : 0;   // its 64 rows of every staged step
    const int lr = lane & 31, lh = lane >> 5;
    // The 4 waves share every K/V tile: they take up to 4 heads of one KV head (GQA) and, if that leaves waves over, further
    // 32-query blocks -- so a tile is staged once for 4 x 32 (head, query) rows and the causal skip works on 32-query blocks.
    const int ratio = a.n_head / a.n_head_kv;
    const int hpw = UNITS == 4 && ratio % 4 == 0 ? 4 : ratio % 2 == 0 ? 2 : 1;  // heads per workgroup (divides the GQA ratio)
    const int qpw = UNITS / hpw;                                              // 32-query blocks per workgroup
    const int kvs = SPLIT ? 1 : a.kvs, zs = SPLIT ? 0 : (int)blockIdx.z % kvs;    // this workgroup's share of the KV blocks: zs, zs + kvs, ...
    const int h = (int)blockIdx.y * hpw + wave % hpw, i3 = (int)blockIdx.z / kvs;
    const int hk = h / ratio;
    const int q = ((int)blockIdx.x * qpw + wave / hpw) * 32 + lr;
    const int KV_BEG = zs * BKV, KV_STEP = kvs * BKV;
    const int qc = q < a.n_q ? q : a.n_q - 1;

    const __half * mrow = a.mask ? (const __half *)(a.mask + (int64_t)qc * a.m_nb1 + (int64_t)(h % a.m_ne2) * a.m_nb2 +
                                                    (int64_t)(i3 % a.m_ne3) * a.m_nb3) : nullptr;
    const char * kbase = a.k + (int64_t)hk * a.k_nb2 + (int64_t)i3 * a.k_nb3;
    const char * vbase = a.v + (int64_t)hk * a.v_nb2 + (int64_t)i3 * a.v_nb3;

    v16f o[DT];
#pragma unroll
    for (int i = 0; i < DT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    constexpr int CH = BKV * D * 2 / 16 / MMA_THREADS;   // 16-byte chunks per thread per tile
    const bool mvec = mrow && (((uintptr_t)a.mask | (uintptr_t)a.m_nb1 | (uintptr_t)a.m_nb2 | (uintptr_t)a.m_nb3) & 7) == 0;   // 4 mask values per load
    typedef _Float16 mh4 __attribute__((ext_vector_type(4)));
    // mask tile of this wave for the block at kv0: 2 S^T tiles x 16 kv rows per lane, kept packed (f16)
    auto issue_mask = [&](int kv0, mh4 (&mp)[2][4]) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kvl = kv0 + 32 * t + 8 * g + 4 * lh;     // 4 consecutive kv rows
                mh4 v;
                if (mvec && kvl + 4 <= a.n_kv) {
                    v = *(const mh4 *)(mrow + kvl);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int kv = kvl + e;
                        v[e] = kv < a.n_kv ? (mrow ? *(const _Float16 *)(mrow + kv) : (_Float16)0.0f) : (_Float16)(-INFINITY);
                    }
                }
                mp[t][g] = v;
            }
    };
    auto mask_any = [&](const mh4 (&mp)[2][4]) -> bool {
        bool any = false;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) any |= (float)mp[t][g][e] != -INFINITY;
        return any;
    };
    // K/V rows of one block: global -> registers (issue) -> LDS (commit).  The loads of block j+1 are issued before block j is
    // computed, so their latency hides behind its MFMAs.
    MmaChunk<TKV> kreg[CH], vreg[CH];
    auto issue = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int c = tid + MMA_THREADS * i;
            const int row = c / (D / 8), cc = c % (D / 8);
            const int kv = kv0 + row < a.n_kv ? kv0 + row : a.n_kv - 1;
            if (cc * 8 < a.d)  kreg[i].load(kbase + (int64_t)kv * a.k_nb1, cc);
            if (cc * 8 < a.dv) vreg[i].load(vbase + (int64_t)kv * a.v_nb1, cc);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int c = tid + MMA_THREADS * i;
            const int row = c / (D / 8), cc = c % (D / 8);
            // tile columns past the head size: zeros
            *(v4i *)(Ks + row * KP + cc * 16) = cc * 8 < a.d  ? kreg[i].f16() : v4i{0, 0, 0, 0};
            *(v4i *)(Vs + row * VP + cc * 16) = cc * 8 < a.dv ? vreg[i].f16() : v4i{0, 0, 0, 0};
        }
    };
    // A block that is masked out for every query of the workgroup (the upper triangle of causal prefill: almost half of all
    // blocks) costs its mask loads and one barrier, no K/V traffic: the mask of block j+1 is read, and voted on, during block j.
    // The mask loads run two blocks ahead (requested during block j-1, voted on at the top of block j for block j+1), so the
    // vote does not wait on memory either.
    mh4 mcur[2][4], mnxt[2][4], mnx2[2][4];
    // prologue: the first K/V rows, the first two mask tiles and the queries are all requested before anything waits -- one memory
    // round trip instead of three (queries -> mask vote -> K/V).  The first block is fetched without asking the mask: it is almost
    // never masked out for a whole workgroup, and a wasted fetch costs less than a round trip.
    if (KV_BEG < a.n_kv) issue(KV_BEG);
    issue_mask(KV_BEG + koff, mnxt);
    issue_mask(KV_BEG + KV_STEP + koff, mnx2);              // (rows past n_kv read as -inf without touching memory)
    // Q^T operand: this lane's query row, features 16 ks + 8 lh .. + 8, pre-scaled, f16
    // (the scores live in the log2 domain: Q carries scale * log2(e), the mask is multiplied by log2(e) when it is added, and
    // the exponentials are bare v_exp_f32)
    constexpr float LOG2E = 1.4426950408889634f;
    // logit softcap (Gemma-2): Q carries scale / softcap instead, and the scores go through softcap * tanh before the mask is added
    const bool capped = a.logit_softcap != 0.0f;
    const float qs = capped ? a.scale / a.logit_softcap : a.scale * LOG2E;
    const float mslope = alibi_slope(a, (uint32_t)h) * LOG2E;       // ALiBi: the mask is the position bias, scaled per head (1 without max_bias)
    half8 qf[KS];
    {
        const float * qp = (const float *)(a.q + (int64_t)qc * a.q_nb1 + (int64_t)h * a.q_nb2 + (int64_t)i3 * a.q_nb3);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            v4f x0 = {0.0f, 0.0f, 0.0f, 0.0f}, x1 = x0;
            if (16 * ks + 8 * lh < a.d) { x0 = *(const v4f *)(qp + 16 * ks + 8 * lh); x1 = *(const v4f *)(qp + 16 * ks + 8 * lh + 4); }
            qf[ks][0] = (_Float16)(x0.x * qs); qf[ks][1] = (_Float16)(x0.y * qs);
            qf[ks][2] = (_Float16)(x0.z * qs); qf[ks][3] = (_Float16)(x0.w * qs);
            qf[ks][4] = (_Float16)(x1.x * qs); qf[ks][5] = (_Float16)(x1.y * qs);
            qf[ks][6] = (_Float16)(x1.z * qs); qf[ks][7] = (_Float16)(x1.w * qs);
        }
    }
    bool any_n = KV_BEG < a.n_kv ? mask_any(mnxt) : false;
    bool wg_n = __syncthreads_or(any_n);
    for (int kv0 = KV_BEG; kv0 < a.n_kv; kv0 += KV_STEP) {
        const bool wg = wg_n, any = any_n;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < 4; g++) { mcur[t][g] = mnxt[t][g]; mnxt[t][g] = mnx2[t][g]; }
        any_n = kv0 + KV_STEP < a.n_kv ? mask_any(mnxt) : false;
        wg_n = __syncthreads_or(any_n);                     // (also: the previous block is fully consumed, the tiles are free)
        if (wg) commit();
        if (wg_n) issue(kv0 + KV_STEP);
        issue_mask(kv0 + 2 * KV_STEP + koff, mnx2);
        if (!wg) continue;
        __syncthreads();
        if (!__any(any)) continue;     // whole tile masked out for every query of this wave (uniform per wave; every wave still
                                       // reaches the two barriers of the next iteration)
        float mv[2][16];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) mv[t][4 * g + e] = (float)mcur[t][g][e];

        // ---- S^T = K Q^T
        v16f sacc[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[t][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const half8 kf = *(const half8 *)(Ks + (koff + 32 * t + lr) * KP + ks * 32 + lh * 16);
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[t], 0, 0, 0);
            }
        }
        // ---- online softmax for query column lr
        if (capped) {      // softcap * tanh(x), tanh(x) = 1 - 2 / (1 + e^2x) on v_exp_f32 / v_rcp_f32 (x -> +-inf: e = inf / 0, exact limits)
            const float cap2 = a.logit_softcap * LOG2E;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float e = __builtin_amdgcn_exp2f(sacc[t][r] * (2.0f * LOG2E));
                    sacc[t][r] = cap2 - 2.0f * cap2 * __builtin_amdgcn_rcpf(1.0f + e);
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) { sacc[t][r] = __builtin_fmaf(mv[t][r], mslope, sacc[t][r]); mx = fmaxf(mx, sacc[t][r]); }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float msafe = mn == -INFINITY ? 0.0f : mn;          // fully masked so far: every p below is exp(-inf) = 0
        const float corr = __builtin_amdgcn_exp2f(m - msafe);      // m = -inf -> 0
        float ps = 0.0f;
        half8 pf[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float p = __builtin_amdgcn_exp2f(sacc[t][r] - msafe);
                ps += p;
                pf[t][r >> 3][r & 7] = (_Float16)p;
            }
        ps += __shfl_xor(ps, 32, 64);
        l = l * corr + ps;
        m = mn;
        if (__any(corr != 1.0f)) {      // the running maxima settle after a few blocks: most blocks need no rescale at all
#pragma unroll
            for (int i = 0; i < DT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[i][r] *= corr;
        }
        // ---- O^T += V^T P^T.  A operand: lane (lr, lh) needs V[kv][d = 32 i + lr] for the 8 kv of its k slots, in the order
        // the S^T accumulators gave P: kv = 32 t + 16 s2 + 8 (j >> 2) + 4 lh + (j & 3).  ds_read_b64_tr_b16 hands every lane of
        // a 16-lane group one COLUMN of a 4-row x 16-column block (lane 4q + p supplies the address of row q, columns 4p..4p+3),
        // so two reads -- rows base + 4 lh + 0..3 and base + 8 + 4 lh + 0..3, columns 16 (lr >> 4) .. + 15 of d tile i -- are the
        // operand, straight from the row-major tile.
        {
            typedef short tr4 __attribute__((ext_vector_type(4)));
            const int q4 = (lane & 15) >> 2, p4 = lane & 3;
            const uint8_t * vb = Vs + (koff + 4 * lh + q4) * VP + (16 * ((lane >> 4) & 1) + 4 * p4) * 2;
#pragma unroll
            for (int i = 0; i < DT; i++)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        const uint8_t * pb = vb + (32 * t + 16 * s2) * VP + 64 * i;
                        const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4 *)pb);
                        const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4 *)(pb + 8 * VP));
                        half8 vf;
                        __builtin_memcpy(&vf, &lo, 8);
                        __builtin_memcpy((char *)&vf + 8, &hi, 8);
                        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[t][s2], o[i], 0, 0, 0);
                    }
        }
    }

    if constexpr (SPLIT) {
        // the two waves of a unit hold the online-softmax state of alternate KV blocks: the second hands (O, m, l) over through the
        // tiles (free after the barrier), same register and lane on both sides, and the first merges with the usual rescale
        static_assert(UNITS * 64 * (DT * 16 + 2) * 4 <= BKV * KP, "the hand-over fits the K tile");
        float * X = (float *)Ks + (size_t)wave * (DT * 16 + 2) * 64;
        __syncthreads();
        if (koff) {
#pragma unroll
            for (int i = 0; i < DT; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) X[(i * 16 + r) * 64 + lane] = o[i][r];
            X[(DT * 16) * 64 + lane] = m; X[(DT * 16 + 1) * 64 + lane] = l;
        }
        __syncthreads();
        if (koff) return;
        const float m1 = X[(DT * 16) * 64 + lane], l1 = X[(DT * 16 + 1) * 64 + lane];
        const float mn = fmaxf(m, m1), msafe = mn == -INFINITY ? 0.0f : mn;
        const float c0 = __builtin_amdgcn_exp2f(m - msafe), c1 = __builtin_amdgcn_exp2f(m1 - msafe);     // -inf -> 0
        l = l * c0 + l1 * c1;
#pragma unroll
        for (int i = 0; i < DT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] = o[i][r] * c0 + X[(i * 16 + r) * 64 + lane] * c1;
    }
    if (!SPLIT && kvs > 1) {
        // partial state of this share of the KV blocks: (O unnormalised, m in the log2 domain, l) per (token, head); fattn_mma_merge_kernel adds them
        if (q < a.n_q) {
            float * pp = a.part + ((((int64_t)i3 * a.n_q + q) * a.n_head + h) * kvs + zs) * (D + 4);      // 16-byte aligned rows
#pragma unroll
            for (int i = 0; i < DT; i++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *(v4f *)(pp + 32 * i + 8 * g + 4 * lh) = v4f{o[i][4 * g], o[i][4 * g + 1], o[i][4 * g + 2], o[i][4 * g + 3]};
            if (lh == 0) { pp[D] = m; pp[D + 1] = l; }
        }
        return;
    }
    if (q < a.n_q) {
        const float inv = 1.0f / l;
        float * op = (float *)(a.dst + (int64_t)h * a.d_nb1 + (int64_t)q * a.d_nb2 + (int64_t)i3 * a.d_nb3);
#pragma unroll
        for (int i = 0; i < DT; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (32 * i + 8 * g + 4 * lh >= a.dv) continue;
                v4f v = {o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
                *(v4f *)(op + 32 * i + 8 * g + 4 * lh) = v;
                if (a.act_bf) {     // the rounding of mi355_act_bf16, for the attention-output GEMM that reads this row next
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    const bf4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                    *(bf4 *)(a.act_bf + (((int64_t)i3 * a.n_q + q) * a.n_head + h) * a.dv + 32 * i + 8 * g + 4 * lh) = b;
                }
            }
    }
}

// merge of the kvs partial states of every (token, head) row: O = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M); one lane per 4 features
template <int D>
__global__ void __launch_bounds__(256)
fattn_mma_merge_kernel(const FArgs a, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = i / (D / 4);
    const int d4 = (int)(i % (D / 4));
    if (row >= rows || 4 * d4 >= a.dv) return;
    const float * pp = a.part + row * a.kvs * (D + 4);
    float M = -INFINITY;
    for (int s = 0; s < a.kvs; s++) M = fmaxf(M, pp[s * (D + 4) + D]);
    const float msafe = M == -INFINITY ? 0.0f : M;
    v4f O = {0.0f, 0.0f, 0.0f, 0.0f};
    float L = 0.0f;
    for (int s = 0; s < a.kvs; s++) {
        const float w = __builtin_amdgcn_exp2f(pp[s * (D + 4) + D] - msafe);      // -inf -> 0
        L += pp[s * (D + 4) + D + 1] * w;
        O = O + *(const v4f *)(pp + s * (D + 4) + 4 * d4) * w;
    }
    const float inv = 1.0f / L;
    const v4f v = O * inv;
    const int h = (int)(row % a.n_head);
    const int64_t t = row / a.n_head;
    const int q = (int)(t % a.n_q), i3 = (int)(t / a.n_q);
    *(v4f *)((float *)(a.dst + (int64_t)h * a.d_nb1 + (int64_t)q * a.d_nb2 + (int64_t)i3 * a.d_nb3) + 4 * d4) = v;
    if (a.act_bf) {
        typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
        const bf4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *(bf4 *)(a.act_bf + row * a.dv + 4 * d4) = b;
    }
}

template <int D>
static int launch_mma(const FArgs & a, int tkv, hipStream_t s) {
    const int ratio = a.n_head / a.n_head_kv;
    if (a.kvs > 1) {
        // KV blocks dealt to kvs workgroups per group of 4 units (the loop is bound by the CU's global-load rate, ~11 B/clk: the bytes
        // a CU stages per query row are what counts -- half the KV rows for twice the queries of the two-wave form), merged afterwards
        const int hpw = ratio % 4 == 0 ? 4 : ratio % 2 == 0 ? 2 : 1, qpw = MMA_WAVES / hpw, bq = 32 * qpw;
        const dim3 grid((unsigned)((a.n_q + bq - 1) / bq), (unsigned)(a.n_head / hpw), (unsigned)(a.ne3 * a.kvs));
        if (tkv == MI355_TYPE_F16)       MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_F16>), grid, dim3(MMA_THREADS), 0, s, a);
        else if (tkv == MI355_TYPE_Q8_0) MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_Q8_0>), grid, dim3(MMA_THREADS), 0, s, a);
        else                             MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_Q4_0>), grid, dim3(MMA_THREADS), 0, s, a);
        MI355_CHECK_LAUNCH();
        const int64_t rows = (int64_t)a.ne3 * a.n_q * a.n_head;
        MI355_LAUNCH("fattn_merge", fattn_mma_merge_kernel<D>, dim3((unsigned)((rows * (D / 4) + 255) / 256)), dim3(256), 0, s, a, rows);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    // fewer (head, 32-query) units than SIMDs, and more than one KV step each: two waves per unit (MI355_FA_MMA_SPLIT=0: never)
    static const int split_on = [] { const char * e = getenv("MI355_FA_MMA_SPLIT"); return e ? atoi(e) : 1; }();
    const int64_t units = (int64_t)((a.n_q + 31) / 32) * a.n_head * a.ne3;
    if (split_on > (D > 128 ? 1 : 0) && units < 1024 && a.n_kv > MMA_BKV) {     // (D = 256: the two-wave form's 128-row staging registers spill; MI355_FA_MMA_SPLIT=2 to try it)
        const int hpw = ratio % 2 == 0 ? 2 : 1, qpw = (MMA_WAVES / 2) / hpw, bq = 32 * qpw;
        const dim3 grid((unsigned)((a.n_q + bq - 1) / bq), (unsigned)(a.n_head / hpw), (unsigned)a.ne3);
        if (tkv == MI355_TYPE_F16)       MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_F16, true>), grid, dim3(MMA_THREADS), 0, s, a);
        else if (tkv == MI355_TYPE_Q8_0) MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_Q8_0, true>), grid, dim3(MMA_THREADS), 0, s, a);
        else                             MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_Q4_0, true>), grid, dim3(MMA_THREADS), 0, s, a);
        MI355_CHECK_LAUNCH();
        return MI355_OK;
    }
    const int hpw = ratio % 4 == 0 ? 4 : ratio % 2 == 0 ? 2 : 1, qpw = MMA_WAVES / hpw, bq = 32 * qpw;
    const dim3 grid((unsigned)((a.n_q + bq - 1) / bq), (unsigned)(a.n_head / hpw), (unsigned)a.ne3);
    if (tkv == MI355_TYPE_F16)       MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_F16>), grid, dim3(MMA_THREADS), 0, s, a);
    else if (tkv == MI355_TYPE_Q8_0) MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_Q8_0>), grid, dim3(MMA_THREADS), 0, s, a);
    else                             MI355_LAUNCH("fattn_mma", (fattn_mma_kernel<D, MI355_TYPE_Q4_0>), grid, dim3(MMA_THREADS), 0, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

template <int D, int R, int TK, int TV, int U = 2>
static int launch_vec(const FArgs & a, hipStream_t s) {
    const int ratio = a.n_head / a.n_head_kv;
    const int hgroups = (ratio + R - 1) / R;
    const dim3 grid((unsigned)(a.n_head_kv * hgroups), (unsigned)a.n_q, (unsigned)(a.ne3 * a.nsplit));
    constexpr int G = FA_THREADS / (D / 8);
    const size_t sh = ((size_t)R * G * (D + 2) + (size_t)R * G + 2 * R) * sizeof(float);
    if (sh > 48 * 1024) { // raise the dynamic-LDS cap once per instantiation
        static const hipError_t attr = hipFuncSetAttribute((const void *)fattn_vec_kernel<D, R, TK, TV, U>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (attr != hipSuccess) return (int)attr;
    }
    MI355_LAUNCH("fattn_vec", (fattn_vec_kernel<D, R, TK, TV, U>), grid, dim3(FA_THREADS), sh, s, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

// heads of one KV head that share a workgroup (and each K/V row it loads).  More heads = fewer K/V reads, which only pays
// when those reads bound the op; on short contexts the per-row arithmetic of the heads in one workgroup is the longer
// chain: Llama-3-8B decode at n_kv 256 (GQA 4) 390.9 t/s with 4, 405.7 with 2, 394.6 with 1; Llama-3-70B (GQA 8) 78.9
// with 8, 82.7 with 4.
static int fa_heads_per_wg(int64_t ratio, int64_t d, int64_t n_kv) {
    // GQA 4, f16, 8 KV heads (tools/bench_fattn.py), 2 / 4 heads per workgroup: n_kv 4096 17.5 / 20.1 us, 16384 29.7 / 28.5,
    // 65536 83.4 / 60.7 (4.4 TB/s): with 2 every K/V row is fetched by two workgroups, which long contexts cannot afford
    if (n_kv < 2048) return ratio >= 2 ? 2 : 1;
    if (ratio >= 8) return d <= 128 ? 8 : 4;
    if (ratio >= 4) return n_kv >= 16384 ? 4 : 2;
    return ratio >= 2 ? 2 : 1;
}

// short contexts: no KV split, 8 rows per lane group in flight (see fattn_vec_kernel); MI355_FA_SHORT=0 restores the split
static int fa_short_rows() {
    static const int v = [] { const char * e = getenv("MI355_FA_SHORT"); return e ? atoi(e) : 512; }();
    return v;
}
// heads per workgroup there: 1 -- the K/V bytes of a short context are nothing, the per-row arithmetic of a second head is the longer
// chain (Llama-3-8B tg128, n_kv 256: 497.5 t/s with the KV split, 507.1 with 2 heads per workgroup, 514.3 with 1)
static int fa_short_heads() {
    static const int v = [] { const char * e = getenv("MI355_FA_SHORT_R"); return e ? atoi(e) : 1; }();
    return v;
}

template <int D, int TK, int TV>
static int dispatch_r(const FArgs & a, hipStream_t s) {
    if constexpr (D <= 128 && TK == TV) {
        if (a.nsplit == 1 && a.n_kv <= fa_short_rows() && a.n_kv > 0) {
            const int ratio = a.n_head / a.n_head_kv;
            if (ratio >= 2 && fa_short_heads() >= 2) return launch_vec<D, 2, TK, TV, 8>(a, s);
            return launch_vec<D, 1, TK, TV, 8>(a, s);
        }
    }
    switch (fa_heads_per_wg(a.n_head / a.n_head_kv, D, a.n_kv)) {
        case 8: if constexpr (D <= 128) return launch_vec<D, 8, TK, TV>(a, s);
        case 4: return launch_vec<D, 4, TK, TV>(a, s);
        case 2: return launch_vec<D, 2, TK, TV>(a, s);
    }
    return launch_vec<D, 1, TK, TV>(a, s);
}

template <int D>
static int dispatch_types(const FArgs & a, int tk, int tv, hipStream_t s) {
    if (tk == MI355_TYPE_F16 && tv == MI355_TYPE_F16)   return dispatch_r<D, MI355_TYPE_F16, MI355_TYPE_F16>(a, s);
    if (tk == MI355_TYPE_Q8_0 && tv == MI355_TYPE_Q8_0) return dispatch_r<D, MI355_TYPE_Q8_0, MI355_TYPE_Q8_0>(a, s);
    if (tk == MI355_TYPE_Q4_0 && tv == MI355_TYPE_Q4_0) return dispatch_r<D, MI355_TYPE_Q4_0, MI355_TYPE_Q4_0>(a, s);
    if (tk == MI355_TYPE_BF16 && tv == MI355_TYPE_BF16) return dispatch_r<D, MI355_TYPE_BF16, MI355_TYPE_BF16>(a, s);
    return MI355_E_TYPE;
}

// Head sizes: any multiple of 8 up to 256 (the reference's kernels cover 40, 64, 72, 80, 96, 112, 128, 256 of these: fattn.cu:271-346,
// test-backend-ops.cpp:4866-4906).  The vec kernel is built for 64 / 128 / 256 features per row and the matrix-core kernel for 64 / 96 / 128 /
// 256; another size runs on the next one up, with zeros past the row.
static int fa_vec_d(int64_t d) { return d <= 64 ? 64 : d <= 128 ? 128 : 256; }
static int fa_mma_d(int64_t d) { return d <= 64 ? 64 : d <= 96 ? 96 : d <= 128 ? 128 : 256; }
// lab switch: head sizes above this take the vec kernel for prefill too (MI355_FA_MMA_MAX_D=128: the behaviour before the 256 tile kernel)
static int fa_mma_max_d() {
    static const int v = [] { const char * e = getenv("MI355_FA_MMA_MAX_D"); return e ? atoi(e) : 256; }();
    return v;
}

} // namespace

extern "C" {

// workspace = [MI355_FATTN_WS_HEADER bytes of arrival counters][partials]
constexpr int64_t WS_HEADER = 65536;

int64_t mi355_flash_attn_ext_workspace_bytes(const mi355_fattn_args * p) {
    if (!p) return -1;
    // header + partials for the largest split the launcher may choose: 64 slices for the vec kernel, 8 shares for the matrix-core one
    const int64_t slices = p->n_q >= 32 ? 8 : 64;
    if (p->d < 8 || p->d > 256 || p->dv < 0 || p->dv > p->d) return -1;
    return WS_HEADER + (int64_t)p->ne3 * p->n_q * p->n_head * slices * (fa_vec_d(p->d) + (p->n_q >= 32 ? 4 : 2)) * (int64_t)sizeof(float);
}

int mi355_flash_attn_ext(const mi355_fattn_args * p, void * workspace, int64_t workspace_bytes, void * stream) {
    return mi355_flash_attn_ext_quant(p, workspace, workspace_bytes, nullptr, nullptr, stream);
}

static int fattn_run(const mi355_fattn_args * p, void * workspace, int64_t workspace_bytes, int8_t * act_q, float * act_ds, uint16_t * act_bf, void * stream);

int mi355_flash_attn_ext_quant(const mi355_fattn_args * p, void * workspace, int64_t workspace_bytes, int8_t * act_q, float * act_ds,
                               void * stream) {
    return fattn_run(p, workspace, workspace_bytes, act_q, act_ds, nullptr, stream);
}

int mi355_flash_attn_ext_bf16(const mi355_fattn_args * p, void * workspace, int64_t workspace_bytes, uint16_t * act_bf16, void * stream) {
    if (!act_bf16) return MI355_E_NULL;
    if ((uintptr_t)act_bf16 & 7) return MI355_E_ALIGN;
    return fattn_run(p, workspace, workspace_bytes, nullptr, nullptr, act_bf16, stream);
}

static int fattn_run(const mi355_fattn_args * p, void * workspace, int64_t workspace_bytes, int8_t * act_q, float * act_ds, uint16_t * act_bf, void * stream) {
    if (!p || !p->q || !p->k || !p->v || !p->dst) return MI355_E_NULL;
    const int64_t dv_ = p->dv ? p->dv : p->d;                      // V's head size: 0 = the same as K's
    if (p->d < 8 || p->d > 256 || p->d % 8 != 0 || dv_ < 8 || dv_ > p->d || dv_ % 8 != 0) return MI355_E_UNSUPPORTED;
    if ((p->type_k == MI355_TYPE_Q8_0 || p->type_k == MI355_TYPE_Q4_0) && p->d % 32 != 0) return MI355_E_SHAPE;
    if ((p->type_v == MI355_TYPE_Q8_0 || p->type_v == MI355_TYPE_Q4_0) && dv_ % 32 != 0) return MI355_E_SHAPE;
    if (act_q && dv_ % 32 != 0) return MI355_E_UNSUPPORTED;      // the q8_1 form's 32-blocks would straddle heads
    if (p->n_head_kv < 1 || p->n_head % p->n_head_kv != 0 || p->n_q < 0 || p->n_kv < 0 || p->ne3 < 1) return MI355_E_SHAPE;
    if (p->n_q > 65535 || p->ne3 * 64 > 65535) return MI355_E_SHAPE;
    if (p->mask && (p->m_ne2 < 1 || p->m_ne3 < 1)) return MI355_E_SHAPE;
    if (((uintptr_t)p->q | (uintptr_t)p->q_nb1 | (uintptr_t)p->q_nb2 | (uintptr_t)p->q_nb3) & 15) return MI355_E_ALIGN;
    if ((p->type_k == MI355_TYPE_F16 || p->type_k == MI355_TYPE_BF16) && (((uintptr_t)p->k | (uintptr_t)p->k_nb1 | (uintptr_t)p->k_nb2 | (uintptr_t)p->k_nb3) & 15)) return MI355_E_ALIGN;
    if ((p->type_v == MI355_TYPE_F16 || p->type_v == MI355_TYPE_BF16) && (((uintptr_t)p->v | (uintptr_t)p->v_nb1 | (uintptr_t)p->v_nb2 | (uintptr_t)p->v_nb3) & 15)) return MI355_E_ALIGN;
    if ((act_q == nullptr) != (act_ds == nullptr)) return MI355_E_NULL;
    if (act_q && (((uintptr_t)act_q & 15) || ((uintptr_t)act_ds & 7))) return MI355_E_ALIGN;
    if (p->n_q == 0 || p->n_head == 0) return MI355_OK;

    FArgs a;
    a.d = (int)p->d; a.dv = (int)dv_;
    const int64_t dv = fa_vec_d(p->d), dm = fa_mma_d(p->d);
    a.act_q = act_q; a.act_ds = act_ds; a.act_bf = nullptr; a.counters = nullptr; a.kvs = 1;
    a.q = (const char *)p->q; a.q_nb1 = p->q_nb1; a.q_nb2 = p->q_nb2; a.q_nb3 = p->q_nb3;
    a.k = (const char *)p->k; a.k_nb1 = p->k_nb1; a.k_nb2 = p->k_nb2; a.k_nb3 = p->k_nb3;
    a.v = (const char *)p->v; a.v_nb1 = p->v_nb1; a.v_nb2 = p->v_nb2; a.v_nb3 = p->v_nb3;
    a.mask = (const char *)p->mask; a.m_nb1 = p->m_nb1; a.m_nb2 = p->m_nb2; a.m_nb3 = p->m_nb3; a.m_ne2 = p->m_ne2; a.m_ne3 = p->m_ne3;
    a.dst = (char *)p->dst; a.d_nb1 = p->d_nb1; a.d_nb2 = p->d_nb2; a.d_nb3 = p->d_nb3;
    a.n_q = (int)p->n_q; a.n_kv = (int)p->n_kv; a.n_head = (int)p->n_head; a.n_head_kv = (int)p->n_head_kv; a.ne3 = (int)p->ne3;
    a.scale = p->scale; a.max_bias = p->max_bias; a.logit_softcap = p->logit_softcap;
    a.n_head_log2 = 1u << (uint32_t)floorf(log2f((float)p->n_head));
    a.m0 = powf(2.0f, -(p->max_bias) / a.n_head_log2);
    a.m1 = powf(2.0f, -(p->max_bias / 2.0f) / a.n_head_log2);

    // prefill-sized batches of f16 / q8_0 / q4_0 KV: matrix-core tile kernel
    if (p->type_k == p->type_v && (p->type_k == MI355_TYPE_F16 || p->type_k == MI355_TYPE_Q8_0 || p->type_k == MI355_TYPE_Q4_0) &&
        p->n_q >= 32 && dm <= fa_mma_max_d() &&
        p->n_head <= 65535 && p->ne3 <= 65535 &&
        ((p->d_nb1 | p->d_nb2 | p->d_nb3 | (int64_t)(uintptr_t)p->dst) & 15) == 0 && (!p->mask || (p->m_nb1 & 1) == 0)) {
        if (act_q) return MI355_E_UNSUPPORTED;   // the fused q8_1 output exists for decode-sized batches only
        a.nsplit = 1; a.part = nullptr; a.act_bf = act_bf; a.kvs = 1;
        // fewer units than SIMDs (a 512-token ubatch) against a long context: deal the KV blocks to 4 workgroups per unit group (two per CU)
        // (MI355_FA_KVS: shares, 0 = never; MI355_FA_KVS_MIN: shortest context, default 1024)
        static const int kvs_env = [] { const char * e = getenv("MI355_FA_KVS"); return e ? atoi(e) : 4; }();
        static const int kvs_min = [] { const char * e = getenv("MI355_FA_KVS_MIN"); return e ? atoi(e) : 1024; }();
        const int64_t units = (int64_t)((p->n_q + 31) / 32) * p->n_head * p->ne3;
        const int64_t part_bytes = (int64_t)p->ne3 * p->n_q * p->n_head * kvs_env * (dm + 4) * (int64_t)sizeof(float);
        if (kvs_env > 1 && kvs_env <= 8 && units < 1024 && p->n_kv >= kvs_min && workspace && ((uintptr_t)workspace & 15) == 0 &&
            workspace_bytes >= WS_HEADER + part_bytes && p->ne3 * kvs_env <= 65535) {
            a.kvs = kvs_env;
            a.part = (float *)((char *)workspace + WS_HEADER);
        }
        switch (dm) {
            case 64:  return launch_mma<64>(a, p->type_k, (hipStream_t)stream);
            case 96:  return launch_mma<96>(a, p->type_k, (hipStream_t)stream);
            case 128: return launch_mma<128>(a, p->type_k, (hipStream_t)stream);
        }
        return launch_mma<256>(a, p->type_k, (hipStream_t)stream);
    }
    if (act_bf) return MI355_E_UNSUPPORTED;      // the bf16 copy exists for the matrix-core kernel only
    // KV split.  The KV loop is a chain of memory round trips (one per G = 512/(d/8) rows), so for decode-sized
    // grids the range is cut until every workgroup has about one round trip of rows -- all loads of the op are
    // then in flight at once -- up to 64 slices, bounded by the workspace; the partials are merged by
    int nsplit = 1;
    const int64_t base_wgs = (int64_t)p->n_head_kv * p->n_q * p->ne3;
    const int64_t G = FA_THREADS / (dv / 8);
    const int64_t ratio = p->n_head / p->n_head_kv;
    const int64_t rr = fa_heads_per_wg(ratio, dv, p->n_kv);      // R chosen by dispatch_r
    const int64_t n_counters = (int64_t)p->ne3 * p->n_q * p->n_head_kv * ((ratio + rr - 1) / rr);
    const bool short_ctx = dv <= 128 && p->type_k == p->type_v && p->n_kv <= fa_short_rows();
    if (!short_ctx && workspace && workspace_bytes > WS_HEADER && ((uintptr_t)workspace & 15) == 0 && n_counters * 4 <= WS_HEADER &&
        base_wgs < 512 && p->n_kv >= 4 * G) {
        // Every slice costs a partial store, a device-scope arrival and a share of the merge, so more slices only pay
        // while they shorten the chain of round trips: one round trip (G rows) per slice up to 8 slices, then slices of
        // >= 4 round trips, and never more than one workgroup per CU in total.  Measured, 8 KV heads x GQA 4, f16
        // (tools/bench_fattn.py; us with the former "one round trip per slice, up to 64 slices" rule -> now):
        // n_kv 1024 21.0 -> 12.6, 4096 31.2 -> 19.3, 16384 51.2 -> 36.3, 65536 104.8 -> 103; twice the workgroups (two per
        // CU) is slower at every length, also with the cheaper inner loop (65536, 4 heads per workgroup: 60.7 -> 75.8).
        int64_t want = (p->n_kv + G - 1) / G;
        const int64_t hgroups = (ratio + rr - 1) / rr;
        const int64_t fill = 256 / (base_wgs * hgroups > 0 ? base_wgs * hgroups : 1);
        const int64_t chain4 = (p->n_kv + 4 * G - 1) / (4 * G);
        if (want > 8) want = chain4 > 8 ? chain4 : 8;
        if (want > fill) want = fill > 8 ? fill : 8;
        if (want > 64) want = 64;
        const int64_t per = (int64_t)p->ne3 * p->n_q * p->n_head * (dv + 2) * (int64_t)sizeof(float);
        if (per > 0 && want * per > workspace_bytes - WS_HEADER) want = (workspace_bytes - WS_HEADER) / per;
        if (want > 1) nsplit = (int)want;
    }
    a.nsplit = nsplit;
    a.counters = (int *)workspace;
    a.part = (float *)((char *)workspace + WS_HEADER);
    hipStream_t s = (hipStream_t)stream;
    switch (dv) {
        case 64:  return dispatch_types<64>(a, p->type_k, p->type_v, s);
        case 128: return dispatch_types<128>(a, p->type_k, p->type_v, s);
        case 256: return dispatch_types<256>(a, p->type_k, p->type_v, s);
    }
    return MI355_E_UNSUPPORTED;
}

// ---- the entry point that allocates nothing --------------------------------------------------------------------------
// Callers under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc cannot bring a workspace.  The
// split-KV scratch (arrival counters + (m, l, acc) partials) then lives in a file-scope device array of fixed size; every
// GPU the library runs on holds its own copy until the process ends.  In host code the array's NAME is only a stand-in
// for the symbol (passing it to a kernel compiles and faults), so the device address is fetched once per device with
// hipGetSymbolAddress, with that device current, and only that pointer is used afterwards.
#define MI355_FATTN_SCRATCH_BYTES (65536 + (4 << 20))
static __device__ __align__(256) unsigned char scratch[MI355_FATTN_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void * g_scratch_ptr[MAX_DEVICES] = {};

// stream: the stream the call will launch on (nullptr: the current device's null stream).  Its device must be the current one -- the
// array's copy is picked by the current device, and a launch on another GPU's stream would be handed a pointer into the wrong GPU.
int scratch_for_current_device(void ** out, hipStream_t stream = nullptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return MI355_E_UNSUPPORTED;
    if (stream) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) == hipSuccess && (int)sdev != dev) return MI355_E_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void * ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));      // this device's copy of the array
        if (e != hipSuccess) return (int)e;
        if (!ptr || ((uintptr_t)ptr & 255)) return MI355_E_ALIGN;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MI355_OK;
}
} // namespace

// also resolves the array's address on the current device, so that a caller who will first use the entry point under stream capture
// can have the symbol lookup happen outside it (a failure here is reported by the entry point itself)
int64_t mi355_flash_attn_ext_noalloc_max_bytes(void) {
    void * ws = nullptr;
    (void)scratch_for_current_device(&ws);
    return MI355_FATTN_SCRATCH_BYTES;
}

int mi355_flash_attn_ext_noalloc(const mi355_fattn_args * p, void * stream) {
    if (!p) return MI355_E_NULL;
    // the scratch must hold what the launcher would ask a caller's workspace for; larger problems (long prefill batches,
    // many heads x many slices) are refused, never silently run with fewer slices
    const int64_t need = mi355_flash_attn_ext_workspace_bytes(p);
    if (need < 0 || need > MI355_FATTN_SCRATCH_BYTES) return MI355_E_WORKSPACE;
    void * ws = nullptr;
    const int rc = scratch_for_current_device(&ws, (hipStream_t)stream);
    if (rc != MI355_OK) return rc;
    // The array keeps the last call's contents.  The partials are written before they are read; the arrival counters
    // must start at zero: every call leaves them at zero again, and they are cleared here as well (a memset node under
    // capture) so that a call cut short by an error cannot poison the next one.
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)WS_HEADER, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return mi355_flash_attn_ext_quant(p, ws, MI355_FATTN_SCRATCH_BYTES, nullptr, nullptr, stream);
}

} // extern "C"
