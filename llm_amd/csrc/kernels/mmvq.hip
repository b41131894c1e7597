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
#include "mmvq_units.h"
#include <algorithm>
#include <mutex>
#include <set>
#include <utility>
#include "rope.h"
#include <stdlib.h>
#include <string>
#include <string.h>
#include <type_traits>

namespace {



// ================================================================================================
// Single-column streaming core (decode: ncols == 1).
//
// Where a decode launch's time went (llm_amd/csrc/lab/timeline.cpp, stamps inside the kernels): the marginal rate of the
// weight stream is ~6.7 TB/s, but every launch paid 4-5 us that had nothing to do with bytes -- the activation loads
// were issued behind the first weight loads and vmcnt retires in order, so staging (or the norm prologue) waited for a
// whole round of weights (2.5-4 us) while only one unit per lane was in flight; then the loop had one round trip per
// iteration.  Here:
//   * a wave owns `nrows` rows (any function of the row index gives the row pointer: consecutive rows, rotated pairs,
//     gate/up alternation) and walks them as one flat sequence of (row, unit-iteration) steps;
//   * PF steps of raw weights are requested before anything else is waited for and stay PF steps ahead in a register
//     ring of PF + 1 slots (the slot refilled at a step is the one the previous step finished with);
//   * the activation / norm-input loads are issued BEFORE the first weight loads (they are L2 hits and come back while
//     the weights are still in flight), the staging arithmetic runs under the weight latency;
//   * row sums end in lane (row index within the wave), so a wave stores / post-processes all its rows at once.
// The per-lane accumulation order (units lane, lane + 64, ...) and wave_sum are those of mmvq_row, so results are
// bit-identical to the multi-column kernel's.
// ================================================================================================

template <int TYPE, int PF, class RowPtr>
struct WStream {
    typedef Unit<TYPE> U;
    static constexpr int NB = PF + 1;
    typename U::Raw ring[NB];
    RowPtr rowp;
    int nrows, nunits, niter, lane;
    int lr, lit;                 // next (row, iteration) to request

    __device__ __forceinline__ WStream(RowPtr rp, int nrows_, int k) : rowp(rp), nrows(nrows_) {
#ifdef MI355_NO_WEIGHTS   // lab: the staging alone
        nrows = 0;
#endif
        nunits = k / U::ELEMS; niter = (nunits + 63) >> 6; lane = threadIdx.x & 63; lr = 0; lit = 0;
    }
    __device__ __forceinline__ void request(typename U::Raw & dst) {
        if (lr < nrows) {                                   // wave-uniform
            const int u_raw = lit * 64 + lane;
            dst = U::load(rowp(lr), u_raw < nunits ? u_raw : nunits - 1);   // lanes past the end re-read the last unit, contribute 0
            if (++lit == niter) { lit = 0; ++lr; }
        }
    }
    __device__ __forceinline__ void begin() {
#pragma unroll
        for (int p = 0; p < PF; p++) request(ring[p]);
    }
    // returns, in lane r, the sum of the wave's r-th row (nrows <= 64)
    template <class AP>
    __device__ __forceinline__ float run(const AP act) {
        float out = 0.0f, acc = 0.0f;
        int r = 0, it = 0;
        while (r < nrows) {
#pragma unroll
            for (int p = 0; p < NB; p++) {
                if (r < nrows) {
                    request(ring[(p + PF) % NB]);
                    const int  u_raw = it * 64 + lane;
                    const bool valid = u_raw < nunits;
                    const int  u     = valid ? u_raw : nunits - 1;
                    const typename U::Act av = U::load_act(act, u);
                    __builtin_amdgcn_sched_barrier(0);
                    const float v = U::dot(U::decode(ring[p], u), av);
                    acc += valid ? v : 0.0f;
                    if (++it == niter) {
                        const float s = wave_sum(acc);
                        out = lane == r ? s : out;
                        acc = 0.0f; it = 0; ++r;
                    }
                }
            }
        }
        return out;
    }
};

// ================================================================================================
// LDS-DMA weight stream (q4_K, q5_K, q6_K; decode, one column).
//
// The register ring above keeps PF steps (~3 KB per wave) of weights in flight, and every 16-byte lane load of a
// block-quantized row touches half a cache line twice.  Here a wave streams its rows with global_load_lds_dwordx4: one
// instruction moves 64 x 16 B = 1 KiB from per-lane source addresses straight into LDS (no VGPR destination), so
//   * D KiB per wave are in flight at any time, however few registers the kernel has (D = 4..8: 64-128 KB per CU) --
//     what covers the norm / quantization prologue of the fused launches and the HBM latency of a cold start;
//   * every byte is requested exactly once (the block header is no longer re-read by the four lanes that share it), so
//     the loads can be non-temporal;
//   * the LDS image is ours to lay out: the "image" of a row is its blocks back to back, a q6_K block (210 B, 2-byte
//     aligned in memory) padded to 224 B so that every field sits at a 16-byte aligned offset (chunk 13 re-reads bytes
//     194..209 so that d lands at image offset 222).
// The images of a wave's rows form one byte stream, cut into 1 KiB pieces that go round a private ring of D slots;
// a piece may span two rows (the source address is per lane).  vmcnt retires in order, so "pieces [0, n) have landed" is
// `s_waitcnt vmcnt(issued - n)`; the DMA is inline asm (the compiler would drain every DMA before any LDS read), which
// also means the compiler's own vmcnt bookkeeping never under-waits: extra outstanding operations only make its waits
// longer.  Units are read back with ds_read_b128 and go through the same decode / dot as the register stream: results
// are bit-identical.
// ================================================================================================

template <int TYPE, class RowPtr>
struct DStream {
    typedef Unit<TYPE> U;
    typedef DmaImg<TYPE> I;
    static_assert(U::ELEMS == 64, "the K-quant units");
    RowPtr rowp;
    int nrows, nunits, niter, lane, nb;
    int cpr, ri;                 // chunks (16 B) and bytes of a row image
    int npieces, issued;
    int rr, cc;                  // this lane's next chunk: row index, chunk index within the row image
    unsigned ring; int slots;    // LDS byte address of the wave's ring, number of 1 KiB slots (a power of two)

    __device__ __forceinline__ DStream(RowPtr rp, int nrows_, int k, unsigned ring_, int slots_) : rowp(rp), nrows(nrows_), ring(ring_), slots(slots_) {
#ifdef MI355_NO_WEIGHTS
        nrows = 0;
#endif
        nb = k / QK_K; nunits = nb * 4; niter = (nunits + 63) >> 6; lane = threadIdx.x & 63;
        cpr = nb * I::CPB; ri = cpr * 16;
        npieces = (nrows * cpr + 63) >> 6; issued = 0;
        rr = 0; cc = lane;                          // cpr >= 64 (launchers): at most one row boundary per piece
        if (cc >= cpr) { cc -= cpr; ++rr; }
    }
    __device__ __forceinline__ void issue_upto(int target) {
        target = target < npieces ? target : npieces;
        while (issued < target) {                       // wave-uniform
            if (rr < nrows)                             // the last piece may be partial
                dma_piece_nt(rowp(rr) + I::src(cc), __builtin_amdgcn_readfirstlane(ring + ((unsigned)(issued & (slots - 1)) << 10)));
            cc += 64;
            const bool wrap = cc >= cpr;
            cc -= wrap ? cpr : 0; rr += wrap ? 1 : 0;
            ++issued;
        }
    }
    // Before the activations are ready only `first` pieces go out: the memory path of a CU returns data in request order,
    // so a deep burst of weights in front of the other waves' activation / norm-input loads delays the whole launch.
    __device__ __forceinline__ void begin(int first) { issue_upto(slots < first ? slots : first); }
    // returns, in lane r, the sum of the wave's r-th row (nrows <= 64)
    template <class AP>
    __device__ __forceinline__ float run(const AP act) {
        const lds_u8 * rg = (const lds_u8 *)(uintptr_t)ring;
        const int mask = (slots << 10) - 1;
        float out = 0.0f;
        issue_upto(slots);                                          // the activations are ready: the ring's full depth
        for (int r = 0; r < nrows; r++) {
            float acc = 0.0f;
            for (int it = 0; it < niter; it++) {
                const int  u_raw = it * 64 + lane;
                const bool valid = u_raw < nunits;
                const int  u     = valid ? u_raw : nunits - 1;
                const int  bend  = (it + 1) * 16 < nb ? (it + 1) * 16 : nb;
                const int  C     = r * ri + bend * I::IMG;          // the stream is consumed up to here by this step
                wait_vm_le(issued - ((C + 1023) >> 10));
                const typename U::Raw raw = I::read(rg, mask, r * ri + (u >> 2) * I::IMG, u);
                const typename U::Act av = U::load_act(act, u);
                issue_upto((C >> 10) + slots);                      // (waits for the LDS reads above: the slots they free are refilled)
                const float v = U::dot(U::decode(raw, u), av);
                acc += valid ? v : 0.0f;
            }
            const float s = wave_sum(acc);
            out = lane == r ? s : out;
        }
        return out;
    }
};

// Slots per wave ring for a workgroup of `nw` waves that may use `lds_budget` bytes of LDS beyond `used`: the largest
// power of two <= 8 that fits (0: no room, use the register stream).
// Which launches use the LDS-DMA stream (MI355_DMA_MASK overrides; 0: none).  Measured on the 8B decode sweep, cold weights,
// hipGraph replay (bench.py): 1.417 ms per step with none, 1.387 with the q6_K mat-vec alone (the robust gain: its 210-byte
// blocks stop straddling cache lines), 1.378 with all four; llama-bench tg128 486.6 / 492.1 / 493.9 t/s.  The kernels'
// time is mostly the prologue chain and the per-CU limit on loads in flight (~48 KB), which the DMA ring does not lift.
enum { DMA_MV_K45 = 1, DMA_MV_K6 = 2, DMA_QKV = 4, DMA_GLU = 8 };
static bool dma_enabled(int which) {
    static const int mask = [] { const char * e = getenv("MI355_DMA_MASK"); return e ? atoi(e) : (DMA_MV_K45 | DMA_MV_K6 | DMA_QKV | DMA_GLU); }();
    return (mask & which) != 0;
}
static int dma_slots(size_t used, int nw, size_t lds_budget, int need) {
    static const int cap = [] { const char * e = getenv("MI355_DMA_SLOTS"); return e ? atoi(e) : 4; }();
    int s = cap > DMA_MAX_SLOTS ? DMA_MAX_SLOTS : cap;
    while (s & (s - 1)) s &= s - 1;
    if (s < need) s = need;
    if (s < 2 || s > DMA_MAX_SLOTS || used + (size_t)nw * s * 1024 > lds_budget) {
        s = need;                                   // the budget does not hold the preferred depth: the least that is correct, or none
        if (s < 2 || s > DMA_MAX_SLOTS || used + (size_t)nw * s * 1024 > lds_budget) return 0;
    }
    return s;
}

static int dma_first() {
    static const int v = [] { const char * e = getenv("MI355_DMA_FIRST"); return e ? atoi(e) : 4; }();
    return v < 1 ? 1 : v;
}

template <int TYPE, int PF, bool DMA, class RowPtr>
static __device__ __forceinline__ auto make_stream(RowPtr rp, int nrows, int k, unsigned ring, int slots) {
    if constexpr (DMA) return DStream<TYPE, RowPtr>(rp, nrows, k, ring, slots);
    else               return WStream<TYPE, PF, RowPtr>(rp, nrows, k);
}


// Activations of one token into LDS ([k] int8, then [k/32] {d, sum}), in two phases so that the global loads go out before
// the weight stream's and the arithmetic runs under its latency.  Three sources:
//   quantized (act_q / act_ds from mi355_quantize_act): a copy;
//   f32 row x: q8_1(x), the arithmetic of quantize_kernel / quant16_soa;
//   f32 row x with a norm weight: q8_1((x * rsqrt(mean(x^2) + eps)) * w), the arithmetic of rms_norm_kernel -- one 16-value
//   chunk per thread, wave sums added in wave order -- so the values are bit-identical to the separate launches (the
//   stand-alone kernel runs 256 threads up to k = 4096 and 512 up to 8192: chunks sit in the same waves here, the
//   remaining waves add 0).  `store`: this workgroup also writes the f32 result (the graph's MUL node).
struct ActSrc {
    const int8_t * q; const float * ds;                 // quantized source, or
    const float * x, * nw; float * nout; float eps;     // f32 source (nw == nullptr: no norm)
};

// How the staging is ordered against the weight stream (llm_amd/csrc/lab/timeline.cpp, stamps inside the kernels):
//   * a CU keeps about 48 KB of loads in flight; a wave that issues a load beyond that stalls AT THE ISSUE until older
//     loads return.  Sixteen waves requesting their first weights fill that at once, so a wave that has started its
//     weight stream reaches a workgroup barrier microseconds late -- the staging must not end in a barrier that the
//     streaming waves take part in after their first request;
//   * vmcnt retires in order: anything a wave loads after its weight requests returns behind them.
// So: the waves that hold a piece of the activations ("staging waves": one 16-value chunk per lane for an f32 source, a
// slice of the copy for a quantized one) fetch and finish it BEFORE their first weight request and meet each other only
// through counters in LDS; the other waves request weights at once; everybody polls the "image complete" counter just
// before the first LDS read, long after it has been reached.
template <int NT_>                                      // threads per workgroup; 0: blockDim.x (a multiple of 64, <= 1024)
struct ActStage {
    int  k, NT, n_stage;            // n_stage: waves that signal `done`

    typedef __attribute__((address_space(3))) unsigned lds_u32;     // explicit LDS pointers: a generic one would poll with flat
    typedef __attribute__((address_space(3))) float    lds_f32;     // loads, which wait on vmcnt -- behind the weight stream
    static __device__ __forceinline__ lds_u32 * ctr()  { __shared__ unsigned c[4]; return (lds_u32 *)c; }
    static __device__ __forceinline__ lds_f32 * sums() { __shared__ float sh[16]; return (lds_f32 *)sh; }
    static __device__ __forceinline__ void lds_wait(volatile lds_u32 * c, unsigned need) {
        while (*c < need) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    }
    static __device__ __forceinline__ void lds_signal(lds_u32 * c) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS writes are done (LDS ops of a wave run in order)
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // Stage the activations of one token into LDS ([k] int8, then [k/32] {d, sum}).  Call before the wave's first weight
    // request.  f32 source: the arithmetic of rms_norm_kernel / quantize_kernel (see ActSrc), wave sums added in wave order.
    // The only workgroup barrier (counters zeroed) sits behind the activation requests, before anything else is in flight.
    // dma: a quantized source is copied by LDS-DMA (the image [k] int8 + [k/32] {d, sum} is contiguous in LDS); the caller
    // then starts its weight stream and calls dma_done(weight pieces issued) -- nothing waits before the weights are asked for
    bool act_dma = false;
    __device__ __forceinline__ void stage(const ActSrc & a, int k_, uint8_t * lds, bool store, void * stamps = nullptr, bool dma = false) {
        k = k_; NT = NT_ ? NT_ : (int)blockDim.x;
        const int t = threadIdx.x, wave = wave_id();
        if (t < 4) ctr()[t] = 0;
        if (!a.x && dma) {
            const int nq = k >> 4, nch = nq + (k >> 6), np = (nch + 63) >> 6, nw = NT >> 6;     // 16-byte chunks of q, of q + ds; pieces
            n_stage = nw; act_dma = true;
            for (int p = wave; p < np; p += nw) {
                const int c = p * 64 + (t & 63);
                if (c < nch)
                    dma_piece_plain(c < nq ? (const uint8_t *)a.q + 16 * c : (const uint8_t *)a.ds + 16 * (c - nq),
                                    __builtin_amdgcn_readfirstlane(lds_addr(lds) + (unsigned)p * 1024));
            }
            __syncthreads();         // counters zeroed (the compiler does not see the DMA: no vmcnt wait here)
            return;
        }
        if (a.x) {
            const int nc = k >> 4, nsw = (nc + 63) >> 6;
            n_stage = nsw;
#ifndef MI355_NO_STAGE_PRIO
            if (wave < nsw) __builtin_amdgcn_s_setprio(3);       // the launch waits for these waves: they issue ahead of the streaming ones
#endif
            const bool has = t < nc;
            v4f v[4], wv[4];
            if (has) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v[q] = ((const v4f *)a.x)[4 * t + q];
                    if (a.nw) wv[q] = ((const v4f *)a.nw)[4 * t + q];
                }
            }
            __syncthreads();
            if (wave >= nsw) return;
            float s = 0.0f;
            if (has) {
#pragma unroll
                for (int q = 0; q < 4; q++) s += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
            }
#ifdef MI355_STAMPS
            asm volatile("" : "+v"(s));
            STAMP_AT(stamps, 5);
#endif
            float scale = 1.0f;
            if (a.nw) {
                lds_f32 * sh = sums();
                const float wsum = wave_sum(s);
                if ((t & 63) == 0) sh[wave] = wsum;
                lds_signal(&ctr()[0]);
                lds_wait(&ctr()[0], (unsigned)nsw);
                float tot = ((volatile lds_f32 *)sh)[0];
                for (int i = 1; i < nsw; i++) tot += ((volatile lds_f32 *)sh)[i];
                scale = 1.0f / sqrtf(tot / (float)k + a.eps);
            }
#ifdef MI355_STAMPS
            asm volatile("" : "+v"(scale));
            STAMP_AT(stamps, 6);
#endif
            if (has) {
                float o[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v4f x = v[q];
                    if (a.nw) { x = x * scale; x = x * wv[q]; if (store) ((v4f *)a.nout)[4 * t + q] = x; }
                    o[4 * q] = x.x; o[4 * q + 1] = x.y; o[4 * q + 2] = x.z; o[4 * q + 3] = x.w;
                }
                quant16_soa(o, (int8_t *)lds + 16 * t, (float *)(lds + (size_t)k) + (t >> 1) * 2, (t & 1) == 0);
            }
        } else {
            n_stage = NT >> 6;
            const int nq = k >> 4, nds = k / QK;
            v2f * lds_ds = (v2f *)(lds + (size_t)k);
            for (int i = t; i < nq; i += NT) ((v4i *)lds)[i] = ((const v4i *)a.q)[i];
            for (int i = t; i < nds; i += NT) lds_ds[i] = ((const v2f *)a.ds)[i];
            __syncthreads();
        }
        lds_signal(&ctr()[1]);
#ifndef MI355_NO_STAGE_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    // after the wave's first weight pieces have been requested (`after` of them): its activation pieces have landed
    __device__ __forceinline__ void dma_done(int after) {
        if (act_dma) { wait_vm_le(after); lds_signal(&ctr()[1]); }
    }
    // before the first LDS read of the activations
    __device__ __forceinline__ void wait() { lds_wait(&ctr()[1], (unsigned)n_stage); }
};

// ------------------------------------------------------------------------------------------------
struct KArgs {
    const uint8_t * W; int64_t w_row_stride, w_ch_stride;
    const int8_t * act_q; const float * act_ds; int64_t act_ch_stride;
    float * dst; int64_t dst_col_stride, dst_ch_stride;
    int m, k, ch_ratio;
    const int32_t * ids; int ids_stride, ids_nused, act_ne1, n_expert;
};


// One wave, one weight row: the body shared by the single-matrix and the grouped kernel.
//   wmat: first row of the matrix; row: this wave's row (rows >= m compute on the last row and store nothing)
template <int TYPE, int NCOLS, bool LDSACT, int NT = 256>
static __device__ __forceinline__ void mmvq_row(const uint8_t * wmat, int64_t w_row_stride, int m, int row, const int8_t * gq,
                                                const float * gds, int k, float * dch, int64_t dst_col_stride, uint8_t * lds,
                                                const float * res = nullptr, float * sum = nullptr, void * stamps = nullptr) {
    typedef Unit<TYPE> U;
    STAMP_AT(stamps, 0);
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
        for (int i = threadIdx.x; i < nq; i += NT) ((v4i *)lds)[i] = ((const v4i *)gq)[i];
        v2f * lds_ds = (v2f *)(lds + (size_t)NCOLS * k);
        for (int i = threadIdx.x; i < nds; i += NT) lds_ds[i] = ((const v2f *)gds)[i];
        __syncthreads();
    }
    STAMP_AT(stamps, 1);

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
        if constexpr (NCOLS <= 2) {
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
        } else {
            // 3..8 columns: the unit is decoded once and the columns' activations are read one column at a time (all of them at once
            // would take 16+ registers per column: 252 VGPRs at 8 columns of q6_K)
            __builtin_amdgcn_sched_barrier(0);
            const typename U::Dec w = U::decode(cur, u);
#pragma unroll
            for (int c = 0; c < NCOLS; c++) {
                typename U::Act av;
                if (LDSACT) {
                    const auto lq  = (const __attribute__((address_space(3))) int8_t *)lds;
                    const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)NCOLS * k);
                    av = U::load_act(ActLds{lq + c * k, lds_ + c * (k / QK) * 2}, u);
                } else {
                    av = U::load_act(ActPtr{gq + (int64_t)c * k, gds + (int64_t)c * (k / QK) * 2}, u);
                }
                const float v = U::dot(w, av);
                acc[c] += valid ? v : 0.0f;
                // two columns in flight, not all of them: the empty asm pins the dot products before the barrier (being pure,
                // they would otherwise sink below every column's loads)
                if ((c & 1) && c + 1 < NCOLS) { asm volatile("" : "+v"(acc[c - 1]), "+v"(acc[c])); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        if (it + 1 < niter) cur = nxt;
    }
    STAMP_AT(stamps, 2);

#pragma unroll
    for (int c = 0; c < NCOLS; c++) {
        const float v = wave_sum(acc[c]);
        if (lane == 0 && row < m) {
            const int64_t o = (int64_t)c * dst_col_stride + row;
            if (sum != dch) dch[o] = v;             // sum == dch: the ADD is in place on the mat-vec result
            if (res) sum[o] = v + rv[c];
        }
    }
    STAMP_AT(stamps, 3);
}

// NT: 256 threads (4 rows per workgroup) or, for 3..8 columns of a tall matrix, 1024 (16 rows): every workgroup stages all
// NCOLS activation columns in LDS, so with 4 rows it moves more activation bytes (from L2) than weight bytes
template <int TYPE, int NCOLS, bool LDSACT, int NT = 256>
__global__ void __launch_bounds__(NT)
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
    mmvq_row<TYPE, NCOLS, LDSACT, NT>(a.W + ch_w * a.w_ch_stride, a.w_row_stride, a.m, blockIdx.x * (NT / 64) + wave,
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
    int k, n, rpb;                                                                // rpb: rows per workgroup the blocks were dealt for (4 or 16)
    const float * res[MI355_MMVQ_MAX_GROUP]; float * sum[MI355_MMVQ_MAX_GROUP];   // optional residual epilogue per matrix
    STAMP_FIELD
};

template <int TYPE, int NCOLS, bool LDSACT, int NT = 256>
__global__ void __launch_bounds__(NT)
mmvq_group_kernel(const KGroup a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6;
    int mi = 0, blk0 = 0;
#pragma unroll
    for (int i = 0; i + 1 < MI355_MMVQ_MAX_GROUP; i++) {
        if (i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i]) { mi = i + 1; blk0 = a.blk_end[i]; }
    }
    mmvq_row<TYPE, NCOLS, LDSACT, NT>(a.W[mi], a.w_row_stride[mi], a.m[mi], ((int)blockIdx.x - blk0) * (NT / 64) + wave, a.act_q, a.act_ds,
                                  a.k, a.dst[mi], a.dst_col_stride[mi], lds, a.res[mi], a.sum[mi], STAMP_PTR(a));
}

// Single-column form of the grouped launch (decode): 1024 threads = 16 waves per workgroup, one workgroup per CU's
// worth of rows, `rpw` consecutive rows per wave (1 for a 4096-row matrix: 256 workgroups).  Optional residual epilogue
// per matrix; the activations come quantized, as an f32 row to quantize, or as an f32 row to normalise and quantize.
constexpr int MV1_THREADS = 1024;
struct KStream {
    const uint8_t * W[MI355_MMVQ_MAX_GROUP]; int64_t w_row_stride[MI355_MMVQ_MAX_GROUP];
    float * dst[MI355_MMVQ_MAX_GROUP];
    const float * res[MI355_MMVQ_MAX_GROUP]; float * sum[MI355_MMVQ_MAX_GROUP];
    int m[MI355_MMVQ_MAX_GROUP], blk_end[MI355_MMVQ_MAX_GROUP];
    int n, k, rpw;
    int dma_slots, ring_off, dma_first;     // LDS-DMA stream: 1 KiB slots per wave ring, byte offset of the rings in dynamic LDS, pieces requested before the activations are ready
    ActSrc act;
    STAMP_FIELD
};

template <int TYPE, bool LDSACT, bool DMA>
__global__ void __launch_bounds__(MV1_THREADS)
mmvq1_kernel(const KStream a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    STAMP_AT(STAMP_PTR(a), 0);
#ifndef MV_PF
#define MV_PF 1        // steps requested ahead; 2 and 3 measured 3-4 % slower over a layer (timeline.cpp: the staging waves slow down)
#endif
    constexpr int PF = MV_PF;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    int mi = 0, blk0 = 0;
#pragma unroll
    for (int i = 0; i + 1 < MI355_MMVQ_MAX_GROUP; i++) {
        if (i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i]) { mi = i + 1; blk0 = a.blk_end[i]; }
    }
    const int row0 = (((int)blockIdx.x - blk0) * (MV1_THREADS / 64) + wave) * a.rpw;
    int nrows = a.m[mi] - row0;
    nrows = nrows < 0 ? 0 : (nrows > a.rpw ? a.rpw : nrows);
    ActStage<MV1_THREADS> st;
    if (LDSACT) st.stage(a.act, a.k, lds, blockIdx.x == 0, STAMP_PTR(a), DMA);
    STAMP_AT(STAMP_PTR(a), 4);
    const float * res = a.res[mi];
    const float rv = (res && lane < nrows) ? res[row0 + lane] : 0.0f;
#ifdef MI355_SAME_ROW      // lab: every wave reads row 0 (cache hits): the loop's compute rate
    const uint8_t * wmat = a.W[mi];
    const int64_t rs = 0;
#else
    const uint8_t * wmat = a.W[mi] + (int64_t)row0 * a.w_row_stride[mi];
    const int64_t rs = a.w_row_stride[mi];
#endif
    auto rowp = [=](int r) { return wmat + (int64_t)r * rs; };
    typename std::conditional<DMA, DStream<TYPE, decltype(rowp)>, WStream<TYPE, PF, decltype(rowp)>>::type
        ws = make_stream<TYPE, PF, DMA>(rowp, nrows, a.k, lds_addr(lds) + a.ring_off + wave * (a.dma_slots << 10), a.dma_slots);
    if constexpr (DMA) ws.begin(a.dma_first); else ws.begin();
    if constexpr (DMA) st.dma_done(ws.issued);
    float out;
    if (LDSACT) {
        st.wait();
        STAMP_AT(STAMP_PTR(a), 1);
        const auto lq  = (const __attribute__((address_space(3))) int8_t *)lds;
        const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)a.k);
        out = ws.run(ActLds{lq, lds_});
    } else {
        STAMP_AT(STAMP_PTR(a), 1);
        out = ws.run(ActPtr{a.act.q, a.act.ds});
    }
    STAMP_AT(STAMP_PTR(a), 2);
    if (lane < nrows) {
        const int64_t o = row0 + lane;
        float * dst = a.dst[mi], * sum = a.sum[mi];
        if (sum != dst) dst[o] = out;               // sum == dst: the ADD is in place on the mat-vec result
        if (res) sum[o] = out + rv;
    }
    STAMP_AT(STAMP_PTR(a), 3);
}

// ------------------------------------------------------------------------------------------------
// Fused attention-input stage of a decode step: Q, K and V projections of one token, the rotary embedding of Q
// and K, and the store of K and V into the KV cache -- ONE launch for what the graph spells as 3 MUL_MAT +
// 2 ROPE + 2 CPY nodes (7 launches).  A wave owns the two weight rows of one rotated pair (both rows read the
// same activations, so the LDS reads are shared), lane 0 rotates the pair and stores it: Q as f32, K and V as
// f16 (or f32) straight into the cache rows.  Row sums, rotation and conversion use the same code as the
// separate kernels, so the results are bit-identical to the unfused graph.
// ------------------------------------------------------------------------------------------------
struct KQkv {
    const uint8_t * W[3]; int64_t w_row_stride[3];
    int pair_end[3];                     // rotated pairs [pair_end[i-1], pair_end[i]) belong to matrix i (0 Q, 1 K, 2 V)
    int ppw;                             // consecutive pairs per wave (1, 2 or 4: divides every matrix's pair count)
    ActSrc act; int k;
    float * q_out; void * k_out; void * v_out; int k_f16, v_f16;
    const int32_t * pos; const float * ff; RopeP rp;
    void * const * kv_tab; int kv_slot;      // optional: K / V destinations read from a device table (graph replay)
    int dma_slots, ring_off, dma_first;      // LDS-DMA stream (see KStream)
    STAMP_FIELD
};

// Up to 1024 threads; the Q | K | V pair list is dealt to the waves of the grid in order, `ppw` consecutive pairs per wave
// (3072 pairs of an 8B token = 256 workgroups x 12 waves x 1 pair), so a wave's matrix is its own (wave-uniform) while a
// workgroup may straddle two matrices.  The two rows of pair j end in lanes 2j, 2j + 1; lane 2j rotates and stores.
constexpr int QKV_MAX_PPW = 4;

template <int TYPE, bool DMA_>
static __device__ __forceinline__ void qkv_body(const KQkv & a, ActStage<0> & st, uint8_t * lds, int mat, int pair0, int np) {
    const int lane = threadIdx.x & 63;
    // rows of pair j of this wave: r0 = base + j * step, r1 = r0 + delta (wave-uniform; a wave's pairs sit in one head)
    int base, step, delta, ic0 = -1;
    if (mat < 2) {
        const int half = a.rp.ne0 / 2, h = pair0 / half, ic = pair0 % half;
        if (2 * ic >= a.rp.n_dims) { base = h * a.rp.ne0 + 2 * ic; step = 2; delta = 1; }                           // passed through
        else if (a.rp.mode & 2)    { base = h * a.rp.ne0 + ic; step = 1; delta = a.rp.n_dims / 2; ic0 = ic; }      // NEOX
        else                       { base = h * a.rp.ne0 + 2 * ic; step = 2; delta = 1; ic0 = ic; }                // NORMAL
    } else {
        base = 2 * pair0; step = 2; delta = 1;
    }
    const uint8_t * w = a.W[mat] + (int64_t)base * a.w_row_stride[mat];
    const int64_t rs = a.w_row_stride[mat];
    auto rowp = [=](int r) { return w + (int64_t)((r >> 1) * step + (r & 1) * delta) * rs; };
    constexpr int PF = 2;        // a wave has 2 * ppw * (k / 4096) steps in all: both rows of an 8B pair are in flight at once
    constexpr bool DMA = DMA_ && DmaImg<TYPE>::OK;
    typename std::conditional<DMA, DStream<TYPE, decltype(rowp)>, WStream<TYPE, PF, decltype(rowp)>>::type
        ws = make_stream<TYPE, PF, DMA>(rowp, 2 * np, a.k, lds_addr(lds) + a.ring_off + wave_id() * (a.dma_slots << 10), a.dma_slots);
    if constexpr (DMA_ && !DMA) st.dma_done(0);       // a register-stream body inside a DMA launch: its activation pieces first
    if constexpr (DMA) ws.begin(a.dma_first); else ws.begin();
    if constexpr (DMA) st.dma_done(ws.issued);
    // the rotation of pair j (lane 2j) while the weights are in flight: a powf and a sincosf, ~1 us if left to the end
    float rc = 1.0f, rsn = 0.0f;
    if (ic0 >= 0 && lane < 2 * np && !(lane & 1)) rope_cos_sin(a.rp, a.pos[0], a.ff, ic0 + (lane >> 1), rc, rsn);
    st.wait();
    STAMP_AT(STAMP_PTR(a), 1);
    const auto lq  = (const __attribute__((address_space(3))) int8_t *)lds;
    const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)a.k);
    const float out = ws.run(ActLds{lq, lds_});
    STAMP_AT(STAMP_PTR(a), 2);
    const float other = dpp_f32<0xB1>(out);          // lane 2j: the sum of row 2j + 1
    if (lane < 2 * np && !(lane & 1)) {
        const int j = lane >> 1;
        const int r0 = base + j * step, r1 = r0 + delta;
        float y0 = out, y1 = other;
        if (ic0 >= 0) { y0 = out * rc - other * rsn; y1 = out * rsn + other * rc; }
        if (mat == 0) { a.q_out[r0] = y0; a.q_out[r1] = y1; }
        else {
            void * o = mat == 1 ? a.k_out : a.v_out;
            if (a.kv_tab) o = a.kv_tab[2 * a.kv_slot + (mat == 1 ? 0 : 1)];
            if (mat == 1 ? a.k_f16 : a.v_f16) { ((__half *)o)[r0] = __float2half_rn(y0); ((__half *)o)[r1] = __float2half_rn(y1); }
            else                              { ((float *)o)[r0] = y0; ((float *)o)[r1] = y1; }
        }
    }
    STAMP_AT(STAMP_PTR(a), 3);
}

constexpr int QKV_MAX_WAVES = 12;      // 768 threads: 3 waves per SIMD, 168 VGPRs (the three type bodies of a mixed launch need ~140)

template <int TQ, int TK, int TV, bool DMA>
__global__ void __launch_bounds__(QKV_MAX_WAVES * 64)
mmvq_qkv_rope_kernel(const KQkv a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    STAMP_AT(STAMP_PTR(a), 0);
    ActStage<0> st;
    st.stage(a.act, a.k, lds, blockIdx.x == 0, STAMP_PTR(a), DMA);
    STAMP_AT(STAMP_PTR(a), 4);
    const int gw = (int)blockIdx.x * (int)(blockDim.x >> 6) + wave_id();
    const int p_first = gw * a.ppw;
    int np = a.pair_end[2] - p_first;
    np = np < 0 ? 0 : (np > a.ppw ? a.ppw : np);
    const int pf  = np ? p_first : 0;                        // idle tail waves: matrix 0, no rows
    const int mat = pf < a.pair_end[0] ? 0 : (pf < a.pair_end[1] ? 1 : 2);
    const int pair0 = pf - (mat == 0 ? 0 : a.pair_end[mat - 1]);
    if (TQ == TK && TK == TV) {
        qkv_body<TQ, DMA>(a, st, lds, mat, pair0, np);
    } else {
        // one barrier sequence per branch: every wave of the workgroup executes exactly one of them
        if (mat == 0)      qkv_body<TQ, DMA>(a, st, lds, 0, pair0, np);
        else if (mat == 1) qkv_body<TK, DMA>(a, st, lds, 1, pair0, np);
        else               qkv_body<TV, DMA>(a, st, lds, 2, pair0, np);
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
    ActSrc act; int k, m, op;
    float * gate_out, * up_out, * dst; int8_t * out_q; float * out_ds;
    // optional expert routing (MUL_MAT_ID pair of a decode token): blockIdx.y = slot, matrices of expert ids[slot], all slots read
    // the same activations, outputs of slot s at [s * m]
    const int32_t * ids; int n_expert, n_slots; int64_t wg_ex_stride, wu_ex_stride;
    int dma_slots, ring_off, dma_first;      // LDS-DMA stream (see KStream)
    STAMP_FIELD
};

static __device__ __forceinline__ float glu_act(int op, float x) {
    switch (op) {
        case 0: return x > 0.0f ? x : 0.0f;                                  // REGLU
        case 1: { const float c = 0.044715f, s = 0.79788456080286535587989211986876f; // GEGLU (tanh gelu, ggml_gelu_f32)
                  return 0.5f * x * (1.0f + tanhf(s * x * (1.0f + c * x * x))); }
        default: return x / (1.0f + expf(-x));                               // SWIGLU (ggml_silu_f32, vec.h:637-639)
    }
}

constexpr int GLU_PAIRS   = 4;        // outputs per wave: NT / 64 waves x 4 = 32 (512 threads) or 64 (1024) outputs per workgroup,
                                      // whole q8_1 blocks of the result
#ifndef GLU_PF
#define GLU_PF 1
#endif

template <int TYPE, int GLU_THREADS, bool DMA>
__global__ void __launch_bounds__(GLU_THREADS)
mmvq_glu_kernel(const KGlu a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    STAMP_AT(STAMP_PTR(a), 0);
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int k = a.k;
    constexpr int NW = GLU_THREADS / 64;
    // outputs per wave: GLU_PAIRS, except that when half the waves first normalise and quantize the token (f32 source,
    // k / 1024 == NW / 2 staging waves) those start their weight stream ~3 us late and take 3 outputs, the others 5
    int np = GLU_PAIRS, lrow0 = wave * GLU_PAIRS;
    if (DMA && GLU_THREADS == 512 && a.act.x && (k >> 10) == NW / 2) {
        np    = wave < NW / 2 ? 3 : 5;
        lrow0 = wave < NW / 2 ? 3 * wave : 3 * (NW / 2) + 5 * (wave - NW / 2);
    }
    const int row0 = blockIdx.x * NW * GLU_PAIRS + lrow0;                     // first of this wave's outputs
    int64_t ex = 0;
    if (a.ids) {
        ex = a.ids[blockIdx.y];
        if (ex < 0 || ex >= a.n_expert) return;      // out-of-range expert id (uniform per workgroup): outputs untouched
    }
    ActStage<GLU_THREADS> st;
    st.stage(a.act, k, lds, blockIdx.x == 0 && blockIdx.y == 0, STAMP_PTR(a), DMA);
    STAMP_AT(STAMP_PTR(a), 4);
    const int64_t so = (int64_t)blockIdx.y * a.m;     // this slot's outputs
    const uint8_t * wg = a.wg + ex * a.wg_ex_stride + (int64_t)row0 * a.rsg, * wu = a.wu + ex * a.wu_ex_stride + (int64_t)row0 * a.rsu;
#ifdef MI355_SAME_ROW
    wg = a.wg; wu = a.wu;
    const int64_t rsg = 0, rsu = 0;
#else
    const int64_t rsg = a.rsg, rsu = a.rsu;
#endif
    // the wave's 8 rows in the order gate 0, up 0, gate 1, up 1, ...: output j ends in lanes 2j (gate) and 2j + 1 (up)
    auto rowp = [=](int r) { return (r & 1) ? wu + (int64_t)(r >> 1) * rsu : wg + (int64_t)(r >> 1) * rsg; };
    constexpr int PF = GLU_PF;
    typename std::conditional<DMA, DStream<TYPE, decltype(rowp)>, WStream<TYPE, PF, decltype(rowp)>>::type
        ws = make_stream<TYPE, PF, DMA>(rowp, 2 * np, k, lds_addr(lds) + a.ring_off + wave * (a.dma_slots << 10), a.dma_slots);
    if constexpr (DMA) ws.begin(a.dma_first); else ws.begin();
    if constexpr (DMA) st.dma_done(ws.issued);
    st.wait();
    STAMP_AT(STAMP_PTR(a), 1);
    const auto lq   = (const __attribute__((address_space(3))) int8_t *)lds;
    const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)k);
    const float out = ws.run(ActLds{lq, lds_});
    STAMP_AT(STAMP_PTR(a), 2);
    // even lanes 0..6 finish one output each; the 32 gated values meet in LDS and 8 lanes quantize the block
    const float up = dpp_f32<0xB1>(out);
    float * hs = (float *)(lds + (((size_t)k + (size_t)k / QK * 8 + 15) & ~(size_t)15));
    if (lane < 2 * np && !(lane & 1)) {
        const int j = lane >> 1;
        if (a.gate_out) a.gate_out[so + row0 + j] = out;
        if (a.up_out)   a.up_out[so + row0 + j]   = up;
        hs[lrow0 + j] = glu_act(a.op, out) * up;
    }
    __syncthreads();
    if (threadIdx.x < NW * GLU_PAIRS / 4) {
        const v4f h = ((const v4f *)hs)[threadIdx.x];
        const int64_t o0 = so + (int64_t)blockIdx.x * (NW * GLU_PAIRS) + 4 * threadIdx.x;
        *(v4f *)(a.dst + o0) = h;
        if (a.out_q) {
            const float o[4] = {h.x, h.y, h.z, h.w};
            quant4_soa(o, a.out_q + o0, a.out_ds + (o0 >> 5) * 2, (threadIdx.x & 7) == 0);
        }
    }
    STAMP_AT(STAMP_PTR(a), 3);
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
        case MI355_TYPE_Q4_1: return "q4_1"; case MI355_TYPE_Q5_0: return "q5_0"; case MI355_TYPE_Q5_1: return "q5_1";
        case MI355_TYPE_Q2_K: return "q2_K"; case MI355_TYPE_Q3_K: return "q3_K"; case MI355_TYPE_IQ4_NL: return "iq4_nl"; case MI355_TYPE_IQ4_XS: return "iq4_xs";
    }
    return "?";
}

static constexpr size_t LDS_ACT_LIMIT = 64 * 1024; // beyond this, activations are read through L1 instead

// 16 rows per workgroup for 3..8 columns when that still gives every CU a workgroup
static int tall_threads() { static const int t = [] { const char * e = getenv("MI355_MMVQ_TALL"); return e ? atoi(e) : 1024; }(); return t; }
constexpr size_t LDS_TOTAL = 159 * 1024;            // per CU, less the kernels' small static arrays

template <int TYPE, int NCOLS>
static int launch(const KArgs & a, int nchannels, hipStream_t stream) {
    const size_t need = (size_t)NCOLS * ((size_t)a.k + (size_t)a.k / QK * 8);
    static const std::string tag = std::string("mmvq_") + type_name(TYPE) + "_n" + std::to_string(NCOLS);
    if constexpr (NCOLS >= 3) {
        if (tall_threads() == 1024 && need <= LDS_TOTAL && (int64_t)a.m * nchannels >= 16 * 192 && (need > 48 * 1024 || (int64_t)a.m * nchannels <= 6144)) {
            hipError_t e = allow_lds_any((const void *)mmvq_kernel<TYPE, NCOLS, true, 1024>, need);
            if (e != hipSuccess) return (int)e;
            MI355_LAUNCH(tag.c_str(), (mmvq_kernel<TYPE, NCOLS, true, 1024>), dim3((unsigned)((a.m + 15) / 16), (unsigned)nchannels), dim3(1024), need, stream, a);
            MI355_CHECK_LAUNCH();
            return MI355_OK;
        }
    }
    const dim3 grid((unsigned)((a.m + 3) / 4), (unsigned)nchannels);
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
    if constexpr (NCOLS >= 3) {
        if (a.rpb == 16) {
            hipError_t e = allow_lds_any((const void *)mmvq_group_kernel<TYPE, NCOLS, true, 1024>, need);
            if (e != hipSuccess) return (int)e;
            MI355_LAUNCH(tag.c_str(), (mmvq_group_kernel<TYPE, NCOLS, true, 1024>), grid, dim3(1024), need, stream, a);
            MI355_CHECK_LAUNCH();
            return MI355_OK;
        }
    }
    if (a.rpb != 4) return MI355_E_SHAPE;
    if (need <= LDS_ACT_LIMIT) MI355_LAUNCH(tag.c_str(), (mmvq_group_kernel<TYPE, NCOLS, true>), grid, dim3(256), need, stream, a);
    else                       MI355_LAUNCH(tag.c_str(), (mmvq_group_kernel<TYPE, NCOLS, false>), grid, dim3(256), 0, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

template <int TYPE>
static int launch_stream(KStream & a, hipStream_t stream) {
    const dim3 grid((unsigned)a.blk_end[a.n - 1]);
    const size_t need = (size_t)a.k + (size_t)a.k / QK * 8;
    static const std::string tag = std::string("mmvq_") + type_name(TYPE) + "_n1";
    if (need <= LDS_ACT_LIMIT) {
        if constexpr (DmaImg<TYPE>::OK) {
            const size_t used = (need + 255) & ~(size_t)255;
            const int slots = a.k % QK_K == 0 && a.k >= 2048 && a.k < (1 << 19) && dma_enabled(TYPE == MI355_TYPE_Q6_K ? DMA_MV_K6 : DMA_MV_K45) ? dma_slots(used, MV1_THREADS / 64, LDS_TOTAL, dma_need_slots<TYPE>(a.k, a.rpw > 1)) : 0;
            if (slots) {
                a.dma_slots = slots; a.ring_off = (int)used; a.dma_first = dma_first();
                const size_t total = used + (size_t)(MV1_THREADS / 64) * slots * 1024;
                hipError_t e = allow_lds(mmvq1_kernel<TYPE, true, true>, total);
                if (e != hipSuccess) return (int)e;
                MI355_LAUNCH(tag.c_str(), (mmvq1_kernel<TYPE, true, true>), grid, dim3(MV1_THREADS), total, stream, a);
                MI355_CHECK_LAUNCH();
                return MI355_OK;
            }
        }
        MI355_LAUNCH(tag.c_str(), (mmvq1_kernel<TYPE, true, false>), grid, dim3(MV1_THREADS), need, stream, a);
    }
    else if (a.act.x)           return MI355_E_UNSUPPORTED;
    else                        MI355_LAUNCH(tag.c_str(), (mmvq1_kernel<TYPE, false, false>), grid, dim3(MV1_THREADS), 0, stream, a);
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
    if (!p || !p->W || !p->dst) return MI355_E_NULL;
    if (p->src_f32) {
        // activations given as an f32 row, quantized (and optionally normalised first) by every workgroup for itself
        if (p->ncols != 1 || p->nchannels != 1 || p->ids) return MI355_E_UNSUPPORTED;
        if (p->k > 16384 || p->k % 32 != 0) return MI355_E_UNSUPPORTED;          // one 16-value chunk per thread, 1024 threads
        if ((p->norm_w == nullptr) != (p->norm_out == nullptr)) return MI355_E_NULL;
        if (((uintptr_t)p->src_f32 | (uintptr_t)p->norm_w | (uintptr_t)p->norm_out) & 15) return MI355_E_ALIGN;
    } else if (!p->act_q || !p->act_ds) return MI355_E_NULL;
    const int64_t bs = mi355_blck_size(p->type);
    if (bs < QK || p->type == MI355_TYPE_Q8_1) return MI355_E_TYPE;
    if (p->m < 0 || p->k <= 0 || p->k % bs != 0 || p->ncols < 1 || p->ncols > 8 || p->nchannels < 1 ||
        p->nchannels > 65535 || p->ch_ratio < 1 || p->m > 0x7fffffff || p->k > 0x7fffffff) return MI355_E_SHAPE;
    if (p->ids && (p->ids_nused < 1 || p->act_ne1 < 1 || p->n_expert < 1)) return MI355_E_SHAPE;
    // alignment the loads rely on: activations 16 B; q4_K/q5_K blocks 16 B; the 2-byte-aligned
    // types (q4_0, q8_0, q6_K) only need their natural 2 B
    if ((uintptr_t)p->dst & 3) return MI355_E_ALIGN;
    if (!p->src_f32 && (((uintptr_t)p->act_q & 15) || ((uintptr_t)p->act_ds & 7))) return MI355_E_ALIGN;
    if (!p->src_f32 && p->k % QK_K == 0 && ((uintptr_t)p->act_ds & 15)) return MI355_E_ALIGN;
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
        case MI355_TYPE_Q4_1: return dispatch_cols<MI355_TYPE_Q4_1>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q5_0: return dispatch_cols<MI355_TYPE_Q5_0>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q5_1: return dispatch_cols<MI355_TYPE_Q5_1>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q2_K: return dispatch_cols<MI355_TYPE_Q2_K>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_Q3_K: return dispatch_cols<MI355_TYPE_Q3_K>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_IQ4_NL: return dispatch_cols<MI355_TYPE_IQ4_NL>(a, p->ncols, p->nchannels, s);
        case MI355_TYPE_IQ4_XS: return dispatch_cols<MI355_TYPE_IQ4_XS>(a, p->ncols, p->nchannels, s);
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
    // rows per workgroup of the multi-column kernel: 16 (1024 threads) for 3..8 columns when the group still gives ~every CU a
    // workgroup -- a workgroup stages all columns' activations, which with 4 rows is more bytes than the weights it streams
    int64_t rows_all = 0;
    for (int i = 0; i < n_mats; i++) rows_all += p[i].m > 0 ? p[i].m : 0;
    const size_t need_cols = (size_t)(p[0].ncols > 0 ? p[0].ncols : 1) * ((size_t)p[0].k + (size_t)p[0].k / QK * 8);
    // (measured, q6_K x 8 columns: 4096 x 14336 54 -> 26 us, 4096 x 4096 10.8 -> 9.8 us, but 14336 x 4096 29 -> 34 us: only where the
    // columns do not fit a 4-row workgroup's LDS budget, or the matrix is short)
    const int rpb = p[0].ncols >= 3 && p[0].ncols <= 8 && p[0].k > 0 && tall_threads() == 1024 && need_cols <= LDS_TOTAL && rows_all >= 16 * 192 &&
                    (need_cols > 48 * 1024 || rows_all <= 6144) ? 16 : 4;
    for (int i = 0; i < n_mats; i++) {
        const int rc = validate(&p[i]);
        if (rc != MI355_OK) return rc;
        if (p[i].nchannels != 1 || p[i].ids) return MI355_E_UNSUPPORTED;
        if (p[i].type != p[0].type || p[i].k != p[0].k || p[i].ncols != p[0].ncols || p[i].act_q != p[0].act_q ||
            p[i].act_ds != p[0].act_ds || p[i].src_f32 != p[0].src_f32 || p[i].norm_w != p[0].norm_w) return MI355_E_SHAPE;
        if (p[i].m == 0) continue;
        a.W[n] = (const uint8_t *)p[i].W; a.w_row_stride[n] = p[i].w_row_stride;
        a.dst[n] = p[i].dst; a.dst_col_stride[n] = p[i].dst_col_stride; a.m[n] = (int)p[i].m;
        if ((p[i].residual == nullptr) != (p[i].sum == nullptr)) return MI355_E_NULL;
        if (((uintptr_t)p[i].residual | (uintptr_t)p[i].sum) & 3) return MI355_E_ALIGN;
        a.res[n] = p[i].residual; a.sum[n] = p[i].sum;
        blocks += (p[i].m + rpb - 1) / rpb;
        if (blocks > 0x7fffffff) return MI355_E_SHAPE;
        a.blk_end[n] = (int)blocks;
        n++;
    }
    if (n == 0) return MI355_OK;
    a.n = n; a.k = (int)p[0].k; a.act_q = p[0].act_q; a.act_ds = p[0].act_ds; a.rpb = rpb;
    STAMP_SET(a);
    hipStream_t s = (hipStream_t)stream;
    if (p[0].ncols == 1) {
        KStream b;
        memset(&b, 0, sizeof(b));
        int64_t rows = 0;
        for (int i = 0; i < n; i++) rows += a.m[i];
        // one workgroup of 16 waves per CU: rows per wave so that the grid is about 256 workgroups (a wave's sums end
        // in its lanes, hence at most 64 rows)
        int64_t rpw = (rows + 4095) / 4096;
        if (rpw > 64) rpw = 64;
        int64_t blk = 0;
        for (int i = 0; i < n; i++) {
            b.W[i] = a.W[i]; b.w_row_stride[i] = a.w_row_stride[i]; b.dst[i] = a.dst[i]; b.res[i] = a.res[i]; b.sum[i] = a.sum[i];
            b.m[i] = a.m[i];
            blk += (a.m[i] + 16 * rpw - 1) / (16 * rpw);
            b.blk_end[i] = (int)blk;
        }
        b.n = n; b.k = a.k; b.rpw = (int)rpw;
        b.act = ActSrc{p[0].act_q, p[0].act_ds, p[0].src_f32, p[0].norm_w, p[0].norm_out, p[0].norm_eps};
        STAMP_SET(b);
        switch (p[0].type) {
            case MI355_TYPE_Q4_0: return launch_stream<MI355_TYPE_Q4_0>(b, s);
            case MI355_TYPE_Q8_0: return launch_stream<MI355_TYPE_Q8_0>(b, s);
            case MI355_TYPE_Q4_K: return launch_stream<MI355_TYPE_Q4_K>(b, s);
            case MI355_TYPE_Q5_K: return launch_stream<MI355_TYPE_Q5_K>(b, s);
            case MI355_TYPE_Q6_K: return launch_stream<MI355_TYPE_Q6_K>(b, s);
            case MI355_TYPE_Q4_1: return launch_stream<MI355_TYPE_Q4_1>(b, s);
            case MI355_TYPE_Q5_0: return launch_stream<MI355_TYPE_Q5_0>(b, s);
            case MI355_TYPE_Q5_1: return launch_stream<MI355_TYPE_Q5_1>(b, s);
            case MI355_TYPE_Q2_K: return launch_stream<MI355_TYPE_Q2_K>(b, s);
            case MI355_TYPE_Q3_K: return launch_stream<MI355_TYPE_Q3_K>(b, s);
            case MI355_TYPE_IQ4_NL: return launch_stream<MI355_TYPE_IQ4_NL>(b, s);
            case MI355_TYPE_IQ4_XS: return launch_stream<MI355_TYPE_IQ4_XS>(b, s);
        }
        return MI355_E_TYPE;
    }
    if (p[0].src_f32) return MI355_E_UNSUPPORTED;          // the f32 prologue exists for one column only
    switch (p[0].type) {
        case MI355_TYPE_Q4_0: return dispatch_group_cols<MI355_TYPE_Q4_0>(a, p[0].ncols, s);
        case MI355_TYPE_Q8_0: return dispatch_group_cols<MI355_TYPE_Q8_0>(a, p[0].ncols, s);
        case MI355_TYPE_Q4_K: return dispatch_group_cols<MI355_TYPE_Q4_K>(a, p[0].ncols, s);
        case MI355_TYPE_Q5_K: return dispatch_group_cols<MI355_TYPE_Q5_K>(a, p[0].ncols, s);
        case MI355_TYPE_Q6_K: return dispatch_group_cols<MI355_TYPE_Q6_K>(a, p[0].ncols, s);
        case MI355_TYPE_Q4_1: return dispatch_group_cols<MI355_TYPE_Q4_1>(a, p[0].ncols, s);
        case MI355_TYPE_Q5_0: return dispatch_group_cols<MI355_TYPE_Q5_0>(a, p[0].ncols, s);
        case MI355_TYPE_Q5_1: return dispatch_group_cols<MI355_TYPE_Q5_1>(a, p[0].ncols, s);
        case MI355_TYPE_Q2_K: return dispatch_group_cols<MI355_TYPE_Q2_K>(a, p[0].ncols, s);
        case MI355_TYPE_Q3_K: return dispatch_group_cols<MI355_TYPE_Q3_K>(a, p[0].ncols, s);
        case MI355_TYPE_IQ4_NL: return dispatch_group_cols<MI355_TYPE_IQ4_NL>(a, p[0].ncols, s);
        case MI355_TYPE_IQ4_XS: return dispatch_group_cols<MI355_TYPE_IQ4_XS>(a, p[0].ncols, s);
    }
    return MI355_E_TYPE;
}

template <int TQ, int TK, int TV>
static int launch_qkv(KQkv & a, hipStream_t stream) {
    const size_t need = (size_t)a.k + (size_t)a.k / QK * 8;
    if (need > LDS_ACT_LIMIT) return MI355_E_UNSUPPORTED;
    // deal the pairs to one workgroup per CU: waves per workgroup x pairs per wave >= pairs per CU, with as few idle
    // wave slots as possible (8B: 3072 pairs = 256 x 12 waves x 1 pair; 70B: 5120 = 256 x 10 x 2)
    const int total = a.pair_end[2], per_cu = (total + 255) / 256;
    int ppw = 1;
    while (ppw < QKV_MAX_PPW && per_cu > QKV_MAX_WAVES * ppw) ppw *= 2;
    // a wave's pairs must not straddle a matrix, a head or the rotated / passed-through boundary
    while (ppw > 1 && (a.pair_end[0] % ppw || a.pair_end[1] % ppw || (a.rp.ne0 / 2) % ppw || (a.rp.n_dims / 2) % ppw)) ppw /= 2;
    int nw = (per_cu + ppw - 1) / ppw;
    nw = nw < 4 ? 4 : (nw > QKV_MAX_WAVES ? QKV_MAX_WAVES : nw);
    if (a.act.x && (a.k >> 4) > nw * 64) nw = ((a.k >> 4) + 63) / 64;     // the f32 prologue: one 16-value chunk per thread
    if (nw > QKV_MAX_WAVES) return MI355_E_UNSUPPORTED;
    a.ppw = ppw;
    const int grid = (total + nw * ppw - 1) / (nw * ppw);
    a.dma_slots = 0; a.ring_off = 0;
    if constexpr (DmaImg<TQ>::OK || DmaImg<TK>::OK || DmaImg<TV>::OK) {
        const size_t used = (need + 255) & ~(size_t)255;
        const int slots = a.k % QK_K == 0 && a.k >= 2048 && a.k < (1 << 19) && dma_enabled(DMA_QKV) ? dma_slots(used, nw, LDS_TOTAL, std::max(dma_need_slots<TQ>(a.k), std::max(dma_need_slots<TK>(a.k), dma_need_slots<TV>(a.k)))) : 0;
        if (slots) {
            a.dma_slots = slots; a.ring_off = (int)used; a.dma_first = dma_first();
            const size_t lds_total = used + (size_t)nw * slots * 1024;
            hipError_t e = allow_lds(mmvq_qkv_rope_kernel<TQ, TK, TV, true>, lds_total);
            if (e != hipSuccess) return (int)e;
            MI355_LAUNCH("mmvq_qkv_rope", (mmvq_qkv_rope_kernel<TQ, TK, TV, true>), dim3((unsigned)grid), dim3((unsigned)(nw * 64)), lds_total, stream, a);
            MI355_CHECK_LAUNCH();
            return MI355_OK;
        }
    }
    MI355_LAUNCH("mmvq_qkv_rope", (mmvq_qkv_rope_kernel<TQ, TK, TV, false>), dim3((unsigned)grid), dim3((unsigned)(nw * 64)), need, stream, a);
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}

extern "C" int mi355_qkv_rope_matvec(const mi355_qkv_rope_args * p, void * stream) {
    if (!p || !p->wq || !p->wk || !p->wv || !p->q_out || !p->k_out || !p->v_out || !p->pos) return MI355_E_NULL;
    if (p->norm_x ? (!p->norm_w || !p->norm_out) : (!p->act_q || !p->act_ds)) return MI355_E_NULL;
    if (p->norm_x && (p->k > 16 * 64 * QKV_MAX_WAVES || p->k % 32 != 0)) return MI355_E_UNSUPPORTED;  // one 16-value chunk per thread
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
    a.pair_end[0] = (int)(p->m_q / 2); a.pair_end[1] = a.pair_end[0] + (int)(p->m_k / 2); a.pair_end[2] = a.pair_end[1] + (int)(p->m_v / 2);
    a.act = ActSrc{p->act_q, p->act_ds, p->norm_x, p->norm_w, p->norm_out, p->norm_eps}; a.k = (int)p->k;
    a.q_out = p->q_out; a.k_out = p->k_out; a.v_out = p->v_out;
    a.k_f16 = p->k_out_type == MI355_TYPE_F16; a.v_f16 = p->v_out_type == MI355_TYPE_F16;
    a.pos = p->pos; a.ff = p->freq_factors;
    if (p->kv_table && (p->kv_slot < 0 || ((uintptr_t)p->kv_table & 7))) return MI355_E_SHAPE;
    a.kv_tab = p->kv_table; a.kv_slot = p->kv_slot;
    STAMP_SET(a);
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
static int launch_glu_mv(KGlu & a, hipStream_t stream) {
    const size_t need = (((size_t)a.k + (size_t)a.k / QK * 8 + 15) & ~(size_t)15) + 64 * sizeof(float);
    if (need > LDS_ACT_LIMIT) return MI355_E_UNSUPPORTED;
    // 4 outputs per wave.  (8 per wave -- half the workgroups, for grids beyond the 512 workgroups a full chip holds at
    // once -- measured slower: 2x28672x8192 q4_K 61 us against 56; the plugin keeps such shapes on the grouped mat-vec.)
#ifndef GLU_WIDE
#define GLU_WIDE 0      // 1024-thread workgroups (64 outputs): no faster than two 512-thread workgroups per CU
#endif
    a.dma_slots = 0; a.ring_off = 0;
    if constexpr (DmaImg<TYPE>::OK) {
        // two workgroups of 8 waves per CU: half the LDS each
        const size_t used = (need + 255) & ~(size_t)255;
        const int slots = a.k % QK_K == 0 && a.k >= 2048 && a.k < (1 << 19) && dma_enabled(DMA_GLU) ? dma_slots(used, 8, LDS_TOTAL / 2, dma_need_slots<TYPE>(a.k)) : 0;
        if (slots) {
            a.dma_slots = slots; a.ring_off = (int)used; a.dma_first = dma_first();
            const size_t total = used + (size_t)8 * slots * 1024;
            hipError_t e = allow_lds(mmvq_glu_kernel<TYPE, 512, true>, total);
            if (e != hipSuccess) return (int)e;
            MI355_LAUNCH("mmvq_glu", (mmvq_glu_kernel<TYPE, 512, true>), dim3((unsigned)(a.m / 32), (unsigned)(a.ids ? a.n_slots : 1)), dim3(512), total, stream, a);
            MI355_CHECK_LAUNCH();
            return MI355_OK;
        }
    }
    if (GLU_WIDE && a.m % 64 == 0)
        MI355_LAUNCH("mmvq_glu", (mmvq_glu_kernel<TYPE, 1024, false>), dim3((unsigned)(a.m / 64), (unsigned)(a.ids ? a.n_slots : 1)), dim3(1024), need, stream, a);
    else
        MI355_LAUNCH("mmvq_glu", (mmvq_glu_kernel<TYPE, 512, false>), dim3((unsigned)(a.m / 32), (unsigned)(a.ids ? a.n_slots : 1)), dim3(512), need, stream, a);
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
    a.act = ActSrc{p->act_q, p->act_ds, p->norm_x, p->norm_w, p->norm_out, p->norm_eps};
    a.k = (int)p->k; a.m = (int)p->m; a.op = p->glu_op;
    a.gate_out = p->gate_out; a.up_out = p->up_out; a.dst = p->dst; a.out_q = p->out_q; a.out_ds = p->out_ds;
    STAMP_SET(a);
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
