// mmvq_layer.hip -- the body of one decode token's transformer layer as ONE persistent launch for gfx950:
//
//     phase 0   attn_out = Wo . attn            ffn_inp = attn_out + residual            (MUL_MAT, ADD)
//     phase 1   h = act(Wgate . n1) * (Wup . n1),  n1 = rms_norm(ffn_inp) * w1            (RMS_NORM, MUL, 2 MUL_MAT, GLU)
//     phase 2   down = Wdown . h                  l_out = down + ffn_inp                  (MUL_MAT, ADD)
//     phase 3   the NEXT layer's attention input: q, k, v = rope / store of Wq|Wk|Wv . n3, n3 = rms_norm(l_out) * w3
//               (RMS_NORM, MUL, 3 MUL_MAT, 2 ROPE, 2 CPY) -- or the model's output: logits = Wout . n3
//
// i.e. what mi355_mul_mat_vec_q_group (+ residual), mi355_glu_matvec, mi355_mul_mat_vec_q_group (+ residual) and
// mi355_qkv_rope_matvec do in four launches (reference: 10 x mul_mat_vec_q + quantize_q8_1 + norm + binbcast + rope + cpy
// launches, ggml-cuda/mmvq.cu:137-224,491-561).  Every value is computed with the arithmetic of those kernels (same unit
// decode / dot, same per-lane accumulation order, same wave_sum, same norm and q8_1 code), so the results are bit-identical.
//
// Why one launch: a decode launch is ~3 us of fixed cost (dispatch, activation staging, tail) around a few us of streaming,
// during which HBM idles.  Here the weight stream never stops: one workgroup of 16 waves per CU;
//   * 12 STREAM waves each own a private ring of step slots in LDS, filled by global_load_lds_dwordx4 (LDS-DMA, no VGPR
//     destination) behind a constant vmcnt (step_stream.h).  A wave's rows of ALL phases form one stream; the issuing cursor
//     runs D - 1 steps ahead of the consuming one whatever phase that is in -- while a wave waits for the next phase's
//     activations, the first steps of that phase's weights are already landing;
//   * 4 COMM waves own everything that crosses workgroups, and issue no DMA, so their vmcnt drains are short: they
//     publish the workgroup's results (write-through sc1 stores, drained, then ONE agent-scope add on a sharded arrival
//     counter), poll the counter (sc1 loads), gather the full vector with sc1 buffer loads (L1-bypassing: no acquire
//     fence, no L2 write-back anywhere -- MI355X_MICROARCH.md, inter-workgroup visibility, first table row), normalise
//     and quantize it into the workgroup's LDS activation area and raise an LDS word for the stream waves.
// Every spin is bounded (s_memrealtime, 100 MHz): on expiry the wave records an error word and goes on, so the grid always
// drains.  All workgroups must be resident at once: the launcher sizes the grid to one workgroup per CU (LDS > 80 KiB
// each, so a CU never takes two).
#include "common.h"
#include "mmvq_units.h"
#include "step_stream.h"
#include "rope.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int DL_NW = 16, DL_THREADS = DL_NW * 64, DL_COMM = 4, DL_STREAM = DL_NW - DL_COMM, DL_CT = DL_COMM * 64;
constexpr int DL_RES = 256;                     // floats per result array in LDS (h, gate, up): outputs per workgroup of a phase
// arrival counters of an edge: DL_REPL replicas of a counter cut in DL_SHARDS shards, every word on a 128-byte line of its
// own.  A workgroup adds to its shard (b % 8) in EVERY replica (one wave instruction, 8 lanes) and polls the 8 shards of ONE
// replica ((b / 8) % 8): a word receives 32 adds and is polled by 32 workgroups -- 256 pollers on one line cost 2.6 us a round.
constexpr int DL_SHARDS = 8, DL_REPL = 8;
constexpr int DL_EDGES = 3;
constexpr int DL_EDGE_WORDS = DL_SHARDS * DL_REPL * 32;
constexpr int DL_ERR_WORD = DL_EDGES * DL_EDGE_WORDS;       // u32 index of the error word in the sync block
constexpr unsigned long long DL_TIMEOUT_TICKS = 400000ull;  // 4 ms of s_memrealtime

typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) float    lds_f32;
typedef __attribute__((address_space(1))) unsigned gu32;

struct DLMat { const uint8_t * w; int64_t rs; int type, m; };

struct KLayer {
    int nwg, pitch, res_off, ring_off;           // pitch: bytes of a ring slot (a full step of the largest type in the launch)
    int k0, k1, k2, k3;                         // activation length of the four phases
    int pw0, pw1, pw2;                          // rows (outputs) per workgroup of the three phases that end in an edge
    // phase 0
    DLMat wo; const int8_t * in_q; const float * in_ds; const float * in_x;
    const float * res0; float * dst0, * sum0;
    // phase 1
    const float * nw1; float * nout1; float eps1;
    DLMat wg; const uint8_t * wu; int glu_op; float * gate_out, * up_out, * hidden;
    // phase 2
    DLMat wd; const float * res2; float * dst2, * sum2;
    // phase 3
    int tail;                                   // 0 none, 1 q / k / v + rope + store, 2 plain rows (output matrix)
    const float * nw3; float * nout3; float eps3;
    DLMat t30, t31, t32; int pe0, pe1, pe2; int ppw, pw3, rpw3;   // ppw: consecutive rotated pairs per stream wave; pw3 / rpw3: rows per workgroup / wave (tail 2)    // ppw: rotated pairs per workgroup, pw3: rows per workgroup (tail 2)
    float * q_out; void * k_out, * v_out; int k_f16, v_f16;
    const int32_t * pos; const float * ff; RopeP rp; void * const * kv_tab; int kv_slot;
    unsigned * sync; int prio_mode;
    unsigned long long * stamps;                // lab: 16 s_memrealtime stamps per wave (mi355_decode_layer_debug_stamps), or null
};

#define DL_STAMP(a, i) do { if ((a).stamps && (threadIdx.x & 63) == 0) (a).stamps[((size_t)blockIdx.x * DL_NW + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// ------------------------------------------------------------------------------------------------ bounded waits
struct Bail {
    lds_u32 * abort; unsigned * err;
    __device__ __forceinline__ void fail(unsigned code) {
        if ((threadIdx.x & 63) == 0) {
            *abort = 1u;
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

// wait until *c >= need (c in LDS, raised by other waves of the workgroup)
static __device__ __forceinline__ void lds_wait_ge(lds_u32 * c, unsigned need, Bail bl, unsigned code) {
    unsigned long long t0 = 0;
    unsigned n = 0;
    while (*(volatile lds_u32 *)c < need) {
        __builtin_amdgcn_s_sleep(1);
        if ((++n & 255u) == 0) {
            if (*(volatile lds_u32 *)bl.abort) break;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = now;
            else if (now - t0 > DL_TIMEOUT_TICKS) { bl.fail(code); break; }
        }
    }
    asm volatile("" ::: "memory");
}
static __device__ __forceinline__ void lds_signal(lds_u32 * c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS writes are done (LDS ops of a wave run in order)
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

static __device__ __forceinline__ float glu_act_dl(int op, float x) {
    switch (op) {
        case 0: return x > 0.0f ? x : 0.0f;                                  // REGLU
        case 1: { const float c = 0.044715f, s = 0.79788456080286535587989211986876f; // GEGLU (tanh gelu, ggml_gelu_f32)
                  return 0.5f * x * (1.0f + tanhf(s * x * (1.0f + c * x * x))); }
        default: return x / (1.0f + expf(-x));                               // SWIGLU (ggml_silu_f32, vec.h:637-639)
    }
}

// ------------------------------------------------------------------------------------------------ stream waves
// q8_1 of 16-value chunk t of a vector another workgroup may have written in this launch -> the LDS activation area
static __device__ __forceinline__ void gather16(__amdgpu_buffer_rsrc_t rsrc, int chunk, v4f v[4]);
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void * p, int bytes);

static __device__ __forceinline__ void stage_hidden(const KLayer & a, uint8_t * lds) {
    const int nc = a.k2 >> 4;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.hidden, a.k2 * 4);
    for (int t0 = 0; t0 < nc; t0 += DL_THREADS) {
        const int t = t0 + (int)threadIdx.x;
        if (t < nc) {
            v4f v[4];
            gather16(rs, t, v);
            float o[16];
#pragma unroll
            for (int q = 0; q < 4; q++) { o[4 * q] = v[q].x; o[4 * q + 1] = v[q].y; o[4 * q + 2] = v[q].z; o[4 * q + 3] = v[q].w; }
            quant16_soa(o, (int8_t *)lds + 16 * t, (float *)(lds + (size_t)a.k2) + (t >> 1) * 2, (t & 1) == 0);
        }
    }
}

// staged activations of phase p are complete when ctr[0] reaches this (comm waves signal every phase, all waves phase 2)
static __device__ __forceinline__ unsigned staged_need(int p) { return p == 0 ? DL_COMM : (p == 1 ? 2 * DL_COMM : (p == 2 ? 2 * DL_COMM + DL_NW : 3 * DL_COMM + DL_NW)); }

// n items dealt to the DL_STREAM stream waves of a workgroup in contiguous runs: wave sw gets [start, start + cnt)
static __device__ __forceinline__ void deal(int n, int sw, int & start, int & cnt) {
    const int base = n / DL_STREAM, rem = n - base * DL_STREAM;
    cnt = base + (sw < rem ? 1 : 0);
    start = sw * base + (sw < rem ? sw : rem);
}

template <int D>
static __device__ __forceinline__ void stream_waves(const KLayer & a, uint8_t * lds, lds_u32 * ctr, Bail bl) {
    const int sw = wave_id() - DL_COMM, lane = threadIdx.x & 63, b = blockIdx.x;
    lds_f32 * res = (lds_f32 *)(lds + a.res_off);
    SStream<D> ss;
    int st0, c0, st1, c1, st2, c2;
    deal(a.pw0, sw, st0, c0);
    deal(2 * a.pw1, sw, st1, c1);            // phase 1 deals ROWS of the sequence gate 0, up 0, gate 1, up 1, ...
    deal(a.pw2, sw, st2, c2);
    ss.s0 = SSeg{a.wo.w + (int64_t)(b * a.pw0 + st0) * a.wo.rs, 2 * a.wo.rs, a.wo.rs, c0, a.k0 / QK_K, a.wo.type};
    {
        // row R of the sequence: matrix R & 1 (0 gate, 1 up), output b * pw1 + (R >> 1); a run that starts on an up row alternates up, gate
        const int64_t o0 = (int64_t)b * a.pw1 + (st1 >> 1), ud = (int64_t)(a.wu - a.wg.w);
        if (st1 & 1) ss.s1 = SSeg{a.wg.w + ud + o0 * a.wg.rs, a.wg.rs, a.wg.rs - ud, c1, a.k1 / QK_K, a.wg.type};
        else         ss.s1 = SSeg{a.wg.w + o0 * a.wg.rs, a.wg.rs, ud, c1, a.k1 / QK_K, a.wg.type};
    }
    ss.s2 = SSeg{a.wd.w + (int64_t)(b * a.pw2 + st2) * a.wd.rs, 2 * a.wd.rs, a.wd.rs, c2, a.k2 / QK_K, a.wd.type};
    // phase 3
    int mat = 0, np = 0, rbase = 0, rstep = 2, rdelta = 1, ic0 = -1, row3 = 0, c3 = 0;
    if (a.tail == 1) {
        const int p_first = (b * DL_STREAM + sw) * a.ppw;
        np = a.pe2 - p_first;
        np = np < 0 ? 0 : (np > a.ppw ? a.ppw : np);
        const int pf = np ? p_first : 0;
        mat = pf < a.pe0 ? 0 : (pf < a.pe1 ? 1 : 2);
        const int pair0 = pf - (mat == 0 ? 0 : (mat == 1 ? a.pe0 : a.pe1));
        if (mat < 2) {
            const int half = a.rp.ne0 / 2, h = pair0 / half, ic = pair0 % half;
            if (2 * ic >= a.rp.n_dims) { rbase = h * a.rp.ne0 + 2 * ic; rstep = 2; rdelta = 1; }                          // passed through
            else if (a.rp.mode & 2)    { rbase = h * a.rp.ne0 + ic; rstep = 1; rdelta = a.rp.n_dims / 2; ic0 = ic; }     // NEOX
            else                       { rbase = h * a.rp.ne0 + 2 * ic; rstep = 2; rdelta = 1; ic0 = ic; }               // NORMAL
        } else {
            rbase = 2 * pair0; rstep = 2; rdelta = 1;
        }
        const DLMat m3 = mat == 0 ? a.t30 : (mat == 1 ? a.t31 : a.t32);
        ss.s3 = SSeg{m3.w + (int64_t)rbase * m3.rs, rstep * m3.rs, rdelta * m3.rs, 2 * np, a.k3 / QK_K, m3.type};
    } else if (a.tail == 2) {
        row3 = b * a.pw3 + sw * a.rpw3;
        int end = (b + 1) * a.pw3 < a.t30.m ? (b + 1) * a.pw3 : a.t30.m;
        c3 = end - row3;
        c3 = c3 < 0 ? 0 : (c3 > a.rpw3 ? a.rpw3 : c3);
        ss.s3 = SSeg{a.t30.w + (int64_t)row3 * a.t30.rs, 2 * a.t30.rs, a.t30.rs, c3, a.k3 / QK_K, a.t30.type};
    } else {
        ss.s3 = SSeg{a.wo.w, 0, 0, 0, 16, MI355_TYPE_Q4_K};
    }
    ss.init(lds_addr(lds) + a.ring_off + sw * D * a.pitch, (unsigned)a.pitch, a.tail ? 4 : 3);
    ss.prio_balance = a.prio_mode != 0;
    ss.prime();
    DL_STAMP(a, 0);

    const auto lq = (const __attribute__((address_space(3))) int8_t *)lds;
    // ---- phase 0: attention output projection
    lds_wait_ge(&ctr[0], staged_need(0), bl, 0x100);
    DL_STAMP(a, 1);
    {
        const float out = ss.run_any(ss.s0, 0, ActLds{lq, (const __attribute__((address_space(3))) float *)(lds + (size_t)a.k0)});
        if (lane < c0) res[st0 + lane] = out;
        lds_signal(&ctr[1]);
        DL_STAMP(a, 2);
    }
    // ---- phase 1: gate / up rows; the two products meet in LDS, the publishing wave applies the activation
    lds_wait_ge(&ctr[0], staged_need(1), bl, 0x101);
    DL_STAMP(a, 3);
    {
        const float out = ss.run_any(ss.s1, 1, ActLds{lq, (const __attribute__((address_space(3))) float *)(lds + (size_t)a.k1)});
        if (lane < c1) {
            const int R = st1 + lane;
            res[((R & 1) ? 2 * DL_RES : DL_RES) + (R >> 1)] = out;
        }
        lds_signal(&ctr[1]);
        DL_STAMP(a, 4);
    }
    // ---- phase 2: down projection (every wave quantizes its share of h first)
    lds_wait_ge(&ctr[2], 2u, bl, 0x302);
    stage_hidden(a, lds);
    lds_signal(&ctr[0]);
    lds_wait_ge(&ctr[0], staged_need(2), bl, 0x102);
    DL_STAMP(a, 5);
    {
        const float out = ss.run_any(ss.s2, 2, ActLds{lq, (const __attribute__((address_space(3))) float *)(lds + (size_t)a.k2)});
        if (lane < c2) res[st2 + lane] = out;
        lds_signal(&ctr[1]);
        DL_STAMP(a, 6);
    }
    if (a.tail == 0) return;
    // ---- phase 3
    float rc = 1.0f, rsn = 0.0f;
    if (a.tail == 1 && ic0 >= 0 && lane < 2 * np && !(lane & 1)) rope_cos_sin(a.rp, a.pos[0], a.ff, ic0 + (lane >> 1), rc, rsn);
    lds_wait_ge(&ctr[0], staged_need(3), bl, 0x103);
    DL_STAMP(a, 7);
    const float out = ss.run_any(ss.s3, 3, ActLds{lq, (const __attribute__((address_space(3))) float *)(lds + (size_t)a.k3)});
    if (a.tail == 1) {
        const float other = dpp_f32<0xB1>(out);          // lane 2j: the sum of row 2j + 1
        if (lane < 2 * np && !(lane & 1)) {
            const int j = lane >> 1;
            const int r0 = rbase + j * rstep, r1 = r0 + rdelta;
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
    } else {
        if (lane < c3) a.q_out[row3 + lane] = out;
    }
    DL_STAMP(a, 8);
}

// ------------------------------------------------------------------------------------------------ comm waves
// 16 consecutive f32 values of a vector another workgroup may have written in this launch: sc1 loads (past this CU's L1)
static __device__ __forceinline__ void gather16(__amdgpu_buffer_rsrc_t rsrc, int chunk, v4f v[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const v4i x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, chunk * 64 + q * 16, 0, 16);       // aux 16 = sc1
        v[q] = __builtin_bit_cast(v4f, x);
    }
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void * p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// arrival of this workgroup at edge e: after the publishing wave's own stores have drained, one agent-scope add to the
// workgroup's shard of every replica (ONE wave instruction)
static __device__ __forceinline__ void edge_arrive(const KLayer & a, int e) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gu32 * cnt = (gu32 *)a.sync + (size_t)e * DL_EDGE_WORDS;
    const int lane = threadIdx.x & 63;
    if (lane < DL_REPL) __hip_atomic_fetch_add(cnt + (lane * DL_SHARDS + ((int)blockIdx.x & (DL_SHARDS - 1))) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and the wait for everybody's: lanes 0..7 poll the shards of this workgroup's replica with sc1 loads
static __device__ __forceinline__ void edge_wait(const KLayer & a, int e, Bail bl) {
    const int lane = threadIdx.x & 63, shard = lane & (DL_SHARDS - 1), rep = ((int)blockIdx.x / DL_SHARDS) & (DL_REPL - 1);
    gu32 * cnt = (gu32 *)a.sync + (size_t)e * DL_EDGE_WORDS + (rep * DL_SHARDS + shard) * 32;
    const unsigned expect = (unsigned)((a.nwg + DL_SHARDS - 1 - shard) / DL_SHARDS);      // workgroups with (b % 8) == shard
    unsigned long long t0 = 0;
    unsigned n = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= expect)) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++n & 63u) == 0) {
            if (*(volatile lds_u32 *)bl.abort) break;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = now;
            else if (now - t0 > DL_TIMEOUT_TICKS) { bl.fail(0x200u + (unsigned)e); break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // no instruction: keeps the gathers below the poll
}

// publish rows (phase 0 / 2): dst = result (optional), sum = result + residual, 4 per lane (pw <= 256), the sum write-through
static __device__ __forceinline__ void publish_rows(const KLayer & a, int e, const lds_f32 * res, int pw, int row0, const float * resid, bool resid_sc1, float * dst, float * sum,
                                                    int m, lds_u32 * done, unsigned need, Bail bl, unsigned code) {
    const int lane = threadIdx.x & 63;
    const bool has = 4 * lane < pw;
    v4f r = v4f{0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(resid, m * 4), rs_sum = make_rsrc(sum, m * 4);
    // the residual first (known since the last edge), then the wait for the stream waves' results
    if (has) {
        if (resid_sc1) r = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (row0 + 4 * lane) * 4, 0, 16));
        else           r = *(const v4f *)(resid + row0 + 4 * lane);
    }
    lds_wait_ge(done, need, bl, code);
    DL_STAMP(a, 2 + 4 * e);
    if (has) {
        const v4f o = *(const __attribute__((address_space(3))) v4f *)(res + 4 * lane);
        // Every store of phases 0..2 is write-through, also those nothing in this launch reads: ggml-alloc places tensors of later
        // nodes on the bytes of dead earlier ones (in a llama layer: l_out on ffn_norm's, the next layer's Qcur on ffn_gate's), and a
        // line left DIRTY in one XCD's L2 by an early phase would be written back at the end of the launch over what another XCD
        // stored there in a later phase.
        if (dst && dst != sum) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, o), make_rsrc(dst, m * 4), (row0 + 4 * lane) * 4, 0, 16);
        const v4f s = o + r;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, s), rs_sum, (row0 + 4 * lane) * 4, 0, 16);
    }
}

// norm + q8_1 of a gathered vector (k <= 8192) into the LDS activation area: the arithmetic of rms_norm_kernel (one 16-value
// chunk per lane of "virtual wave" chunk / 64, wave butterflies, wave sums added in wave order) and quant16_soa
static __device__ __forceinline__ void stage_norm(const KLayer & a, const float * x, const float * nw, float * nout, float eps, int k,
                                                  uint8_t * lds, lds_u32 * ctr, lds_f32 * wsum, unsigned & nsum_need, int e, Bail bl) {
    const int ct = threadIdx.x, cw = wave_id(), lane = ct & 63;
    const int nc = k >> 4, nsw = (nc + 63) >> 6;
    v4f wv[2][4], v[2][4];
    // the norm weights do not depend on the edge: requested before the wait (the publishing wave: between its arrival and its poll,
    // so that its drain covers its stores only)
    if (cw == 0) { edge_arrive(a, e); DL_STAMP(a, 3 + 4 * e); }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int t = ct + DL_CT * p;
        if (t < nc) {
#pragma unroll
            for (int q = 0; q < 4; q++) wv[p][q] = ((const v4f *)nw)[4 * t + q];
        }
    }
    if (cw == 0) {
        edge_wait(a, e, bl);
        if (lane == 0) ctr[2] = (unsigned)(e + 1);
        DL_STAMP(a, 4 + 4 * e);
    } else {
        lds_wait_ge(&ctr[2], (unsigned)(e + 1), bl, 0x300u + (unsigned)e);
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(x, k * 4);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int t = ct + DL_CT * p;
        if (t < nc) gather16(rs, t, v[p]);
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int t = ct + DL_CT * p, vw = cw + DL_COMM * p;
        if (vw < nsw) {                                             // wave-uniform
            float s = 0.0f;
            if (t < nc) {
#pragma unroll
                for (int q = 0; q < 4; q++) s += v[p][q].x * v[p][q].x + v[p][q].y * v[p][q].y + v[p][q].z * v[p][q].z + v[p][q].w * v[p][q].w;
            }
            const float ws = wave_sum(s);
            if (lane == 0) wsum[vw] = ws;
            lds_signal(&ctr[3]);
        }
    }
    nsum_need += (unsigned)nsw;
    lds_wait_ge(&ctr[3], nsum_need, bl, 0x310u + (unsigned)e);
    float tot = ((volatile lds_f32 *)wsum)[0];
    for (int i = 1; i < nsw; i++) tot += ((volatile lds_f32 *)wsum)[i];
    const float scale = 1.0f / sqrtf(tot / (float)k + eps);
    const bool store = blockIdx.x == 0;
    const __amdgpu_buffer_rsrc_t rs_n = make_rsrc(nout, k * 4);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int t = ct + DL_CT * p;
        if (t < nc) {
            float o[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v4f xx = v[p][q];
                xx = xx * scale; xx = xx * wv[p][q];
                if (store) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, xx), rs_n, (4 * t + q) * 16, 0, 16);      // write-through (publish_rows)
                o[4 * q] = xx.x; o[4 * q + 1] = xx.y; o[4 * q + 2] = xx.z; o[4 * q + 3] = xx.w;
            }
            quant16_soa(o, (int8_t *)lds + 16 * t, (float *)(lds + (size_t)k) + (t >> 1) * 2, (t & 1) == 0);
        }
    }
    if (store) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... and at memory before this workgroup's next arrival can be counted
    lds_signal(&ctr[0]);
    DL_STAMP(a, 5 + 4 * e);
}

static __device__ __forceinline__ void comm_waves(const KLayer & a, uint8_t * lds, lds_u32 * ctr, lds_f32 * wsum, Bail bl) {
    const int ct = threadIdx.x, cw = wave_id(), lane = ct & 63, b = blockIdx.x;
    const lds_f32 * res = (const lds_f32 *)(lds + a.res_off);
    unsigned nsum_need = 0;
    // ---- phase 0 input: the attention output, quantized (a copy) or f32 (q8_1 of it)
    if (a.in_x) {
        const int nc = a.k0 >> 4;
        for (int t0 = 0; t0 < nc; t0 += DL_CT) {
            const int t = t0 + ct;
            if (t < nc) {
                float o[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const v4f x = ((const v4f *)a.in_x)[4 * t + q];
                    o[4 * q] = x.x; o[4 * q + 1] = x.y; o[4 * q + 2] = x.z; o[4 * q + 3] = x.w;
                }
                quant16_soa(o, (int8_t *)lds + 16 * t, (float *)(lds + (size_t)a.k0) + (t >> 1) * 2, (t & 1) == 0);
            }
        }
    } else {
        const int nq = a.k0 >> 4, nds = a.k0 / QK;
        v2f * lds_ds = (v2f *)(lds + (size_t)a.k0);
        for (int i = ct; i < nq; i += DL_CT) ((v4i *)lds)[i] = ((const v4i *)a.in_q)[i];
        for (int i = ct; i < nds; i += DL_CT) lds_ds[i] = ((const v2f *)a.in_ds)[i];
    }
    lds_signal(&ctr[0]);
    DL_STAMP(a, 1);

    // ---- edge 0: ffn_inp = attn_out + residual, then n1 = rms_norm(ffn_inp) * w1 -> q8_1
    if (cw == 0) publish_rows(a, 0, res, a.pw0, b * a.pw0, a.res0, false, a.dst0, a.sum0, a.wo.m, &ctr[1], DL_STREAM * 1, bl, 0x400);
    stage_norm(a, a.sum0, a.nw1, a.nout1, a.eps1, a.k1, lds, ctr, wsum, nsum_need, 0, bl);

    // ---- edge 1: h = act(gate) * up (and the two products), then q8_1(h) by ALL waves of the workgroup
    if (cw == 0) {
        lds_wait_ge(&ctr[1], DL_STREAM * 2, bl, 0x401);
        DL_STAMP(a, 6);
        const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.hidden, a.k2 * 4);
        const int o0 = b * a.pw1;
        if (4 * lane < a.pw1) {
            const v4f gv = *(const __attribute__((address_space(3))) v4f *)(res + DL_RES + 4 * lane);
            const v4f uv = *(const __attribute__((address_space(3))) v4f *)(res + 2 * DL_RES + 4 * lane);
            if (a.gate_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, gv), make_rsrc(a.gate_out, a.k2 * 4), (o0 + 4 * lane) * 4, 0, 16);
            if (a.up_out)   __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, uv), make_rsrc(a.up_out, a.k2 * 4), (o0 + 4 * lane) * 4, 0, 16);
            v4f hv;
            hv.x = glu_act_dl(a.glu_op, gv.x) * uv.x; hv.y = glu_act_dl(a.glu_op, gv.y) * uv.y;
            hv.z = glu_act_dl(a.glu_op, gv.z) * uv.z; hv.w = glu_act_dl(a.glu_op, gv.w) * uv.w;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, hv), rs_h, (o0 + 4 * lane) * 4, 0, 16);
        }
        edge_arrive(a, 1);
        DL_STAMP(a, 7);
        edge_wait(a, 1, bl);
        if (lane == 0) ctr[2] = 2u;
        DL_STAMP(a, 8);
    } else {
        lds_wait_ge(&ctr[2], 2u, bl, 0x301);
    }
    stage_hidden(a, lds);
    lds_signal(&ctr[0]);
    DL_STAMP(a, 9);

    // ---- edge 2: l_out = down + ffn_inp, then (tail) n3 = rms_norm(l_out) * w3 -> q8_1
    if (cw == 0) publish_rows(a, 2, res, a.pw2, b * a.pw2, a.res2, true, a.dst2, a.sum2, a.wd.m, &ctr[1], DL_STREAM * 3, bl, 0x402);
    if (a.tail == 0) return;
    stage_norm(a, a.sum2, a.nw3, a.nout3, a.eps3, a.k3, lds, ctr, wsum, nsum_need, 2, bl);
}

template <int D>
__global__ void __launch_bounds__(DL_THREADS)
decode_layer_kernel(const KLayer a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ unsigned ctr_[8];                // 0 staged, 1 done, 2 go, 3 norm sums posted, 4 abort
    __shared__ float wsum_[16];
    lds_u32 * ctr = (lds_u32 *)ctr_;
    if (threadIdx.x < 8) ctr_[threadIdx.x] = 0;
    __syncthreads();                // the only workgroup barrier, before anything is in flight
    Bail bl{&ctr[4], a.sync + DL_ERR_WORD};
    if (wave_id() < DL_COMM) DL_STAMP(a, 0);
    if (wave_id() < DL_COMM && a.prio_mode) __builtin_amdgcn_s_setprio(3);      // the hand-offs are on every phase's critical path
    if (wave_id() < DL_COMM) comm_waves(a, lds, ctr, (lds_f32 *)wsum_, bl);
    else                     stream_waves<D>(a, lds, ctr, bl);
}

static bool kq(int t) { return t == MI355_TYPE_Q4_K || t == MI355_TYPE_Q5_K || t == MI355_TYPE_Q6_K; }

static int n_cus() {
    static std::mutex mu;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = n;
    }
    return cus[dev];
}

} // namespace

// lab: a device buffer of (workgroups x 16 waves x 16) u64 that every later launch stamps its phases into (nullptr: off)
static unsigned long long * g_dl_stamps = nullptr;
extern "C" MI355_API void mi355_decode_layer_debug_stamps(void * p) { g_dl_stamps = (unsigned long long *)p; }

extern "C" int64_t mi355_decode_layer_sync_bytes(void) { return (int64_t)(DL_ERR_WORD + 32) * 4; }

extern "C" int mi355_decode_layer(const mi355_decode_layer_args * p, void * sync, void * stream) {
    if (!p || !sync) return MI355_E_NULL;
    if (!p->wo || !p->w_gate || !p->w_up || !p->w_down || !p->residual || !p->ffn_inp || !p->ffn_norm_w || !p->ffn_norm_out ||
        !p->hidden || !p->l_out) return MI355_E_NULL;
    if (p->attn_f32 ? false : (!p->attn_q || !p->attn_ds)) return MI355_E_NULL;
    if (!kq(p->type_o) || !kq(p->type_gu) || !kq(p->type_d)) return MI355_E_UNSUPPORTED;
    if (p->glu_op < 0 || p->glu_op > 2) return MI355_E_UNSUPPORTED;
    const int64_t n_embd = p->n_embd, k_o = p->k_o, n_ff = p->n_ff;
    if (n_embd <= 0 || k_o <= 0 || n_ff <= 0 || n_embd > (1 << 20) || k_o > (1 << 20) || n_ff > (1 << 20)) return MI355_E_SHAPE;
    if (n_embd % 1024 || k_o % 1024 || n_ff % 1024) return MI355_E_UNSUPPORTED;      // rows of whole 4-block groups (step_stream.h)
    if (n_embd > 8192) return MI355_E_UNSUPPORTED;                  // the norm stage: two 16-value chunks per comm lane
    if (((uintptr_t)sync & 127)) return MI355_E_ALIGN;
    const uintptr_t f16 = (uintptr_t)p->attn_f32 | (uintptr_t)p->residual | (uintptr_t)p->attn_out | (uintptr_t)p->ffn_inp | (uintptr_t)p->ffn_norm_w |
                          (uintptr_t)p->ffn_norm_out | (uintptr_t)p->gate_out | (uintptr_t)p->up_out | (uintptr_t)p->hidden | (uintptr_t)p->down_out |
                          (uintptr_t)p->l_out | (uintptr_t)p->attn_q | (uintptr_t)p->attn_ds;
    if (f16 & 15) return MI355_E_ALIGN;
    auto wal = [](int t, const void * w, int64_t rs) { return (((uintptr_t)w | (uintptr_t)rs) & ((t == MI355_TYPE_Q6_K) ? 1 : 15)) == 0; };
    if (!wal(p->type_o, p->wo, p->wo_row_stride) || !wal(p->type_gu, p->w_gate, p->gu_row_stride) || !wal(p->type_gu, p->w_up, p->gu_row_stride) ||
        !wal(p->type_d, p->w_down, p->down_row_stride)) return MI355_E_ALIGN;
    if (p->wo_row_stride < mi355_row_size(p->type_o, k_o) || p->gu_row_stride < mi355_row_size(p->type_gu, n_embd) ||
        p->down_row_stride < mi355_row_size(p->type_d, n_ff)) return MI355_E_SHAPE;

    KLayer a;
    memset(&a, 0, sizeof(a));
    // one workgroup per CU; every phase that ends in an edge deals whole groups of 4 rows to every workgroup
    int nwg = n_cus();
    if (nwg <= 0) return MI355_E_UNSUPPORTED;
    if (nwg > 256) nwg = 256;
    while (nwg >= 32 && (n_embd % (4 * nwg) || n_ff % (4 * nwg))) nwg >>= 1;
    if (nwg < 32) return MI355_E_UNSUPPORTED;
    a.nwg = nwg;
    a.pw0 = (int)(n_embd / nwg); a.pw1 = (int)(n_ff / nwg); a.pw2 = a.pw0;
    if (a.pw0 > DL_RES || a.pw1 > DL_RES || (a.pw0 + DL_STREAM - 1) / DL_STREAM > 64 || (2 * a.pw1 + DL_STREAM - 1) / DL_STREAM > 64) return MI355_E_UNSUPPORTED;
    a.k0 = (int)k_o; a.k1 = (int)n_embd; a.k2 = (int)n_ff; a.k3 = (int)n_embd;
    a.wo = DLMat{(const uint8_t *)p->wo, p->wo_row_stride, p->type_o, (int)n_embd};
    a.in_q = p->attn_q; a.in_ds = p->attn_ds; a.in_x = p->attn_f32;
    a.res0 = p->residual; a.dst0 = p->attn_out; a.sum0 = p->ffn_inp;
    a.nw1 = p->ffn_norm_w; a.nout1 = p->ffn_norm_out; a.eps1 = p->norm_eps;
    a.wg = DLMat{(const uint8_t *)p->w_gate, p->gu_row_stride, p->type_gu, (int)n_ff}; a.wu = (const uint8_t *)p->w_up;
    a.glu_op = p->glu_op; a.gate_out = p->gate_out; a.up_out = p->up_out; a.hidden = p->hidden;
    a.wd = DLMat{(const uint8_t *)p->w_down, p->down_row_stride, p->type_d, (int)n_embd};
    a.res2 = p->ffn_inp; a.dst2 = p->down_out; a.sum2 = p->l_out;
    a.tail = p->tail;
    int pitch = std::max(step_pitch(p->type_o), std::max(step_pitch(p->type_gu), step_pitch(p->type_d)));
    if (p->tail == MI355_TAIL_QKV) {
        const mi355_qkv_rope_args * q = &p->qkv;
        if (!q->wq || !q->wk || !q->wv || !q->q_out || !q->k_out || !q->v_out || !q->pos || !q->norm_w || !q->norm_out) return MI355_E_NULL;
        if (q->norm_x != p->l_out || q->k != n_embd) return MI355_E_SHAPE;
        const int tk = q->type_k ? q->type_k : q->type_qk;
        if (!kq(q->type_qk) || !kq(tk) || !kq(q->type_v)) return MI355_E_UNSUPPORTED;
        if ((q->k_out_type != MI355_TYPE_F16 && q->k_out_type != MI355_TYPE_F32) || (q->v_out_type != MI355_TYPE_F16 && q->v_out_type != MI355_TYPE_F32)) return MI355_E_TYPE;
        if (q->m_q <= 0 || q->m_k <= 0 || q->m_v <= 0 || q->m_q % 8 || q->m_k % 8 || q->m_v % 8 || q->m_q + q->m_k + q->m_v > 0x3fffffff) return MI355_E_SHAPE;
        if (q->head_dim < 2 || (q->head_dim & 1) || q->m_q % q->head_dim || q->m_k % q->head_dim || q->n_dims < 2 || (q->n_dims & 1) || q->n_dims > q->head_dim) return MI355_E_SHAPE;
        if (q->mode != 0 && q->mode != 2) return MI355_E_UNSUPPORTED;
        if (((uintptr_t)q->q_out & 3) || ((uintptr_t)q->k_out & 3) || ((uintptr_t)q->v_out & 3) || (((uintptr_t)q->norm_w | (uintptr_t)q->norm_out) & 15)) return MI355_E_ALIGN;
        if (!wal(q->type_qk, q->wq, q->wq_row_stride) || !wal(tk, q->wk, q->wk_row_stride) || !wal(q->type_v, q->wv, q->wv_row_stride)) return MI355_E_ALIGN;
        if (q->wq_row_stride < mi355_row_size(q->type_qk, n_embd) || q->wk_row_stride < mi355_row_size(tk, n_embd) ||
            q->wv_row_stride < mi355_row_size(q->type_v, n_embd)) return MI355_E_SHAPE;
        if (q->kv_table && (q->kv_slot < 0 || ((uintptr_t)q->kv_table & 7))) return MI355_E_SHAPE;
        a.t30 = DLMat{(const uint8_t *)q->wq, q->wq_row_stride, q->type_qk, (int)q->m_q};
        a.t31 = DLMat{(const uint8_t *)q->wk, q->wk_row_stride, tk, (int)q->m_k};
        a.t32 = DLMat{(const uint8_t *)q->wv, q->wv_row_stride, q->type_v, (int)q->m_v};
        a.pe0 = (int)(q->m_q / 2); a.pe1 = a.pe0 + (int)(q->m_k / 2); a.pe2 = a.pe1 + (int)(q->m_v / 2);
        a.rp = rope_params(q->n_dims, q->mode, q->head_dim, q->n_ctx_orig, q->freq_base, q->freq_scale, q->ext_factor, q->attn_factor, q->beta_fast, q->beta_slow);
        // pairs per wave: the list Q | K | V dealt in order to nwg x DL_STREAM waves; a wave's pairs must not straddle a matrix, a
        // head or the rotated / passed-through boundary
        const int total = a.pe2, per_wg = (total + nwg - 1) / nwg;
        int ppw = 1;
        while (ppw < 32 && per_wg > DL_STREAM * ppw) ppw *= 2;
        if (a.pe0 % ppw || a.pe1 % ppw || (a.rp.ne0 / 2) % ppw || (a.rp.n_dims / 2) % ppw || per_wg > DL_STREAM * ppw) return MI355_E_UNSUPPORTED;
        a.ppw = ppw;
        a.nw3 = q->norm_w; a.nout3 = q->norm_out; a.eps3 = q->norm_eps;
        a.q_out = q->q_out; a.k_out = q->k_out; a.v_out = q->v_out;
        a.k_f16 = q->k_out_type == MI355_TYPE_F16; a.v_f16 = q->v_out_type == MI355_TYPE_F16;
        a.pos = q->pos; a.ff = q->freq_factors; a.kv_tab = q->kv_table; a.kv_slot = q->kv_slot;
        pitch = std::max(pitch, std::max(step_pitch(q->type_qk), std::max(step_pitch(tk), step_pitch(q->type_v))));
    } else if (p->tail == MI355_TAIL_ROWS) {
        const mi355_mmvq_args * o = &p->out;
        if (!o->W || !o->dst || !o->norm_w || !o->norm_out) return MI355_E_NULL;
        if (o->src_f32 != p->l_out || o->k != n_embd || o->ncols != 1 || o->nchannels != 1 || o->ids || o->residual || o->sum) return MI355_E_SHAPE;
        if (!kq(o->type)) return MI355_E_UNSUPPORTED;
        if (o->m <= 0 || o->m > 0x3fffffff) return MI355_E_SHAPE;
        if (!wal(o->type, o->W, o->w_row_stride) || ((uintptr_t)o->dst & 3) || (((uintptr_t)o->norm_w | (uintptr_t)o->norm_out) & 15)) return MI355_E_ALIGN;
        if (o->w_row_stride < mi355_row_size(o->type, n_embd)) return MI355_E_SHAPE;
        a.t30 = DLMat{(const uint8_t *)o->W, o->w_row_stride, o->type, (int)o->m};
        a.pw3 = (int)((o->m + nwg - 1) / nwg); a.rpw3 = (a.pw3 + DL_STREAM - 1) / DL_STREAM;
        if (a.rpw3 > 64) return MI355_E_UNSUPPORTED;
        a.nw3 = o->norm_w; a.nout3 = o->norm_out; a.eps3 = o->norm_eps;
        a.q_out = o->dst;
        pitch = std::max(pitch, step_pitch(o->type));
    } else if (p->tail != MI355_TAIL_NONE) return MI355_E_SHAPE;

    const size_t kmax = (size_t)std::max(std::max(a.k0, a.k1), a.k2);
    const size_t act_bytes = (kmax + kmax / QK * 8 + 255) & ~(size_t)255;
    a.res_off = (int)act_bytes;
    a.ring_off = (int)(act_bytes + 3 * DL_RES * sizeof(float));
    // steps per wave ring: 3 when the LDS holds them (two steps in flight while one is consumed), else 2
    static const int want = [] { const char * e = getenv("MI355_DL_DEPTH"); return e ? atoi(e) : 3; }();
    const size_t lds_budget = 159 * 1024;
    int depth = want >= 3 ? 3 : 2;
    if (a.ring_off + (size_t)DL_STREAM * depth * pitch > lds_budget) depth = 2;
    if (a.ring_off + (size_t)DL_STREAM * depth * pitch > lds_budget) return MI355_E_UNSUPPORTED;
    a.pitch = pitch;
    size_t total = a.ring_off + (size_t)DL_STREAM * depth * pitch;
    if (total <= 82 * 1024) total = 82 * 1024;                      // never two workgroups on one CU: the grid must be resident at once
    a.sync = (unsigned *)sync;
    a.stamps = g_dl_stamps;
    { static const int pm = [] { const char * e = getenv("MI355_DL_PRIO"); return e ? atoi(e) : 0; }(); a.prio_mode = pm; }
    if (depth == 3) {
        hipError_t e = allow_lds(decode_layer_kernel<3>, total);
        if (e != hipSuccess) return (int)e;
        MI355_LAUNCH("decode_layer", decode_layer_kernel<3>, dim3((unsigned)nwg), dim3(DL_THREADS), total, (hipStream_t)stream, a);
    } else {
        hipError_t e = allow_lds(decode_layer_kernel<2>, total);
        if (e != hipSuccess) return (int)e;
        MI355_LAUNCH("decode_layer", decode_layer_kernel<2>, dim3((unsigned)nwg), dim3(DL_THREADS), total, (hipStream_t)stream, a);
    }
    MI355_CHECK_LAUNCH();
    return MI355_OK;
}
