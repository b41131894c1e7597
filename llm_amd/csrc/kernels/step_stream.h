// step_stream.h -- the weight stream of the decode mat-vecs as whole "steps" (gfx950, K-quants).
//
// A wave consumes its rows in steps of 16 blocks (64 lanes x one 64-value unit; fewer blocks in a row's last step).  Here a
// step is also the unit of the LDS-DMA: its blocks' images (DmaImg<TYPE>: q4_K 144 B, q5_K 176 B, q6_K 224 B per block) are
// fetched by P wave instructions of L lanes each (P x L x 16 B = the step: q4_K 3 x 48 lanes, q5_K 4 x 44, q6_K 4 x 56) into one
// slot of a ring of D slots, with
//   * no vector arithmetic per fetch: the source is (SGPR pair: row base + step offset) + a per-lane offset that is the same
//     for every full step of a type, the destination a wave-uniform M0;
//   * a CONSTANT vmcnt in steady state -- every step of a type is P operations, so "step i has landed" is
//     `s_waitcnt vmcnt(P x (D - 2))` with an immediate, no counting and no switch;
//   * LDS read addresses = slot base + per-lane constants (a block never straddles anything: no masking per read).
// The piece-based stream it replaces (DStream, mmvq.hip) spent ~30 vector instructions per step on fetch addresses and ring
// offsets and ~25 scalar branches on a run-time vmcnt; per lane and step the arithmetic (Unit<TYPE>::decode / dot) and the
// order in which a row's units are added are unchanged, so results are bit-identical.
//
// The stream is a list of up to four SEGMENTS (a phase's rows of one type); the issuing cursor runs D - 1 steps ahead of the
// consuming one through segment boundaries, whatever the consumer waits for in between.
//
// Measured and NOT kept: a byte ring with slots of the step's own size (5 steps of q4_K in flight instead of 2)
// and rows grabbed from a per-workgroup LDS queue instead of a fixed deal -- both correct, both slower: the SIMD arbiter
// serves its oldest wave first, memory requests included, so deeper per-wave rings let the old waves run further ahead of the
// young ones, and the queue's grab (an LDS atomic and the row's address arithmetic per row) sits on the issue path.
#pragma once
#include "mmvq_units.h"

namespace {

struct SGeo { int P, cpb, blkb, img; };       // fetches per step, 16-byte chunks per block image, bytes of a block in memory / in LDS
static __device__ __forceinline__ SGeo sgeo(int type) {
    if (type == MI355_TYPE_Q6_K) return SGeo{4, 14, 210, 224};
    if (type == MI355_TYPE_Q5_K) return SGeo{4, 11, 176, 176};
    return SGeo{3, 9, 144, 144};
}
// byte offset, from the step's first block in memory, of image chunk c
static __device__ __forceinline__ unsigned step_src(int type, int c) {
    if (type == MI355_TYPE_Q6_K) {
        const int b = (int)(((unsigned)c * 18725u) >> 18), j = c - 14 * b;          // c / 14 for c < 2^16
        return (unsigned)(b * 210 + (j < 13 ? 16 * j : 194));                       // chunk 13 re-reads bytes 194..209: d lands at image offset 222
    }
    return 16u * (unsigned)c;
}

// rows of a wave in one segment: row r at base + (r >> 1) * sp + (r & 1) * dl, nb blocks each (nb % 4 == 0)
struct SSeg { const uint8_t * base; int64_t sp, dl; int nrows, nb, type; };

// one fetch: lanes with exec set copy 16 B from sbase + voff to LDS m0base + 16 * lane
static __device__ __forceinline__ void dma_saddr(unsigned voff, const uint8_t * sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int N> static __device__ __forceinline__ void wait_vm_imm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <class T> static __device__ __forceinline__ T ssel4(int i, T a, T b, T c, T d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }

template <int D>
struct SStream {
    SSeg s0, s1, s2, s3;
    int nseg;
    unsigned ring, pitch; int lane;
    // issuing cursor (wave-uniform)
    int iseg, ir, iit, islot, issued, total;
    SSeg cur; SGeo g; int niter, nb_last, L_full, L_last;
    const uint8_t * rb;                     // row `ir` of the cursor's segment
    unsigned vo[4], vl[4];                  // per-lane source offsets of the P fetches of a full / a row's last step
    // consuming cursor
    int cslot, consumed;
    bool prio_balance = false;

    static __device__ __forceinline__ int steps_of(const SSeg & s) { return s.nrows * ((s.nb + 15) >> 4); }
    __device__ __forceinline__ void load_seg(int i) {
#define SS_PICK(f) ssel4(i, s0.f, s1.f, s2.f, s3.f)
        cur.base = SS_PICK(base); cur.sp = SS_PICK(sp); cur.dl = SS_PICK(dl); cur.nrows = SS_PICK(nrows); cur.nb = SS_PICK(nb); cur.type = SS_PICK(type);
#undef SS_PICK
        g = sgeo(cur.type);
        niter = (cur.nb + 15) >> 4; nb_last = cur.nb - 16 * (niter - 1);
        L_full = 16 * g.cpb / g.P; L_last = nb_last * g.cpb / g.P;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            vo[p] = step_src(cur.type, p * L_full + lane);
            vl[p] = step_src(cur.type, p * L_last + lane);
        }
        ir = 0; iit = 0; rb = cur.base;
    }
    __device__ __forceinline__ void init(unsigned ring_, unsigned pitch_, int nseg_) {
        ring = ring_; pitch = pitch_; nseg = nseg_; lane = threadIdx.x & 63;
        total = steps_of(s0) + (nseg > 1 ? steps_of(s1) : 0) + (nseg > 2 ? steps_of(s2) : 0) + (nseg > 3 ? steps_of(s3) : 0);
        issued = 0; islot = 0; cslot = 0; consumed = 0;
        iseg = 0; load_seg(0);
        while (cur.nrows == 0 && iseg + 1 < nseg) { ++iseg; load_seg(iseg); }
    }
    // request the next step of the stream (if any) into slot `islot`
    __device__ __forceinline__ void issue_step() {
        if (issued >= total) return;                                // wave-uniform
        const bool last = iit == niter - 1;
        const int  L = last ? L_last : L_full;
        const uint8_t * src0 = rb + (int64_t)iit * (16 * g.blkb);
        // (wave-uniform by construction; the compiler cannot always prove it across the callers' polling loops)
        const uint8_t * src = (const uint8_t *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)src0 >> 32)) << 32) |
                                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)src0));
        const unsigned dst = ring + (unsigned)islot * pitch;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's reads of the slot (a step ago) are complete
        if (lane < L) {
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (p < g.P) dma_saddr(last ? vl[p] : vo[p], src, __builtin_amdgcn_readfirstlane(dst + (unsigned)(p * L * 16)));
        }
        ++issued;
        islot = islot + 1 == D ? 0 : islot + 1;
        if (++iit == niter) {
            iit = 0;
            if (++ir == cur.nrows) {
                if (issued < total) { do { ++iseg; load_seg(iseg); } while (cur.nrows == 0); }
            } else {
                rb = cur.base + (int64_t)(ir >> 1) * cur.sp + ((ir & 1) ? cur.dl : (int64_t)0);
            }
        }
    }
    __device__ __forceinline__ void prime() {
#pragma unroll
        for (int i = 0; i < D - 1; i++) issue_step();
    }
    // wait until the step about to be consumed has landed: everything but the steps requested after it
    template <int P>
    __device__ __forceinline__ void wait_step(int seg) {
        const int ahead = issued - consumed - 1;                    // steps requested after this one
        if (ahead == D - 2 && (P == 3 || iseg == seg)) wait_vm_imm<P * (D - 2)>();
        else if (ahead == D - 2) wait_vm_imm<3 * (D - 2)>();        // the cursor is in a later segment: at least 3 fetches per step
        else {                                                      // the end of the stream
            if (D > 3 && ahead == 1) wait_vm_imm<3>(); else wait_vm_imm<0>();
        }
    }
    // the rows of segment `seg` (TYPE = its type): returns, in lane r, the sum of row r (nrows <= 64)
    template <int TYPE, class AP>
    __device__ __forceinline__ float run(const SSeg & sg, int seg, const AP act) {
        typedef Unit<TYPE> U;
        typedef DmaImg<TYPE> I;
        constexpr int P = TYPE == MI355_TYPE_Q4_K ? 3 : 4;
        const lds_u8 * rg = (const lds_u8 *)(uintptr_t)0;
        const int nunits = sg.nb * 4, nit = (sg.nb + 15) >> 4;
        const unsigned lblk = (unsigned)(lane >> 2) * I::IMG;
        float out = 0.0f;
        for (int r = 0; r < sg.nrows; r++) {
            float acc = 0.0f;
            if (prio_balance) {
                // The SIMD's arbiter serves its oldest wave first (memory requests included), so with equal shares the youngest
                // waves of a workgroup finish a phase ~30 % after the oldest.  A wave that is behind asks for more: priority
                // 3 in the first quarter of its rows, 0 in the last.
                const int lvl = (4 * r) / sg.nrows;
                if (lvl == 0) __builtin_amdgcn_s_setprio(3); else if (lvl == 1) __builtin_amdgcn_s_setprio(2);
                else if (lvl == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
            for (int it = 0; it < nit; it++) {
                wait_step<P>(seg);
                issue_step();                                       // step i + D - 1 into the slot step i - 1 was read from
                const int  u     = it * 64 + lane;
                const bool valid = u < nunits;
                const unsigned sb = ring + (unsigned)cslot * pitch + lblk;
                const typename U::Raw raw = I::read(rg, -1, (int)sb, u);
                const typename U::Act av = U::load_act(act, valid ? u : nunits - 1);
                const float v = U::dot(U::decode(raw, u), av);
                acc += valid ? v : 0.0f;
                cslot = cslot + 1 == D ? 0 : cslot + 1;
                ++consumed;
            }
            const float s = wave_sum(acc);
            out = lane == r ? s : out;
        }
        return out;
    }
    template <class AP>
    __device__ __forceinline__ float run_any(const SSeg & sg, int seg, const AP act) {
        switch (sg.type) {                                          // wave-uniform
            case MI355_TYPE_Q4_K: return run<MI355_TYPE_Q4_K>(sg, seg, act);
            case MI355_TYPE_Q5_K: return run<MI355_TYPE_Q5_K>(sg, seg, act);
            default:              return run<MI355_TYPE_Q6_K>(sg, seg, act);
        }
    }
};

// LDS bytes of a ring slot that holds a full step of `type`
static inline int step_pitch(int type) { return type == MI355_TYPE_Q6_K ? 16 * 224 : (type == MI355_TYPE_Q5_K ? 16 * 176 : 16 * 144); }

} // namespace
