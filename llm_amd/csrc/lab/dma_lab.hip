// dma_lab.hip -- can a per-wave LDS-DMA ring stream q4_K weights from cold HBM faster than the register ring of the
// production mat-vec?  (NOT product code; tuning evidence only.)
//
// Each wave owns `rpw` consecutive rows = one contiguous byte stream.  It keeps D pieces of 1 KiB in flight with
// global_load_lds_dwordx4 (64 lanes x 16 B, no VGPR destination) into a private LDS ring, waits with counted vmcnt, reads
// its units back with ds_read_b128 and runs the production arithmetic (Unit<Q4_K>::decode / dot), so the result must be
// bit-identical to mi355_mul_mat_vec_q.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I llm_amd/csrc/kernels \
//         llm_amd/csrc/lab/dma_lab.hip llm_amd/csrc/kernels/{quantize,api,timing}.hip -o llm_amd/lib/dma_lab
#include "../kernels/mmvq.hip"

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

typedef __attribute__((address_space(3))) uint8_t lds_u8;

template <int NT = 0>
static __device__ __forceinline__ void dma16(const void * gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

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
        default: if (q < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct KDma {
    const uint8_t * W; int64_t rs; int m, k, rpw;
    const int8_t * act_q; const float * act_ds; float * dst;
};

// NW waves per workgroup, D pieces per wave ring; MATH 0: ds_read + xor only (the stream's ceiling)
template <int NW, int D, int MATH>
__global__ void __launch_bounds__(NW * 64)
dma_q4k_kernel(const KDma a) {
    typedef Unit<MI355_TYPE_Q4_K> U;
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    constexpr int S = D * 1024;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int k = a.k;
    const int kq = (k + 1023) & ~1023, dsb = k / QK * 8, dsq = (dsb + 1023) & ~1023;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_u8 *)lds;
    const unsigned ring = __builtin_amdgcn_readfirstlane(lds0 + kq + dsq + wave * S);

    const int row0 = (blockIdx.x * NW + wave) * a.rpw;
    int nrows = a.m - row0;
    nrows = nrows < 0 ? 0 : (nrows > a.rpw ? a.rpw : nrows);
    const uint8_t * base = a.W + (int64_t)row0 * a.rs;
    const int L = nrows * (int)a.rs;
    const int npieces = (L + 1023) >> 10;

    // activations first (they retire first): the act_q pieces and the act_ds pieces are dealt over the waves
    {
        const int nqp = kq >> 10, ndp = dsq >> 10;
        for (int p = wave; p < nqp + ndp; p += NW) {
            const bool isq = p < nqp;
            const int  pp  = isq ? p : p - nqp;
            const int  lim = isq ? k : dsb;
            int off = pp * 1024 + lane * 16;
            off = off + 16 <= lim ? off : lim - 16;
            const uint8_t * src = (isq ? (const uint8_t *)a.act_q : (const uint8_t *)a.act_ds) + off;
            dma16(src, __builtin_amdgcn_readfirstlane(lds0 + (isq ? 0 : kq) + pp * 1024));
        }
    }
    int issued = 0;
    auto issue_upto = [&](int target) {
        target = target < npieces ? target : npieces;
        while (issued < target) {
            int off = issued * 1024 + lane * 16;
            off = off + 16 <= L ? off : L - 16;
            dma16<(MATH >> 1) & 1>(base + off, __builtin_amdgcn_readfirstlane(ring + (issued & (D - 1)) * 1024));
            issued++;
        }
    };
    issue_upto(D);
    wait_vm_le(issued);                      // the activation pieces of this wave have landed
    __builtin_amdgcn_s_barrier();

    const auto lq   = (const __attribute__((address_space(3))) int8_t *)(lds_u8 *)lds;
    const auto lds_ = (const __attribute__((address_space(3))) float *)((lds_u8 *)lds + kq);
    const lds_u8 * rg = (const lds_u8 *)(uintptr_t)ring;
    const int nunits = k / 64, niter = (nunits + 63) >> 6, rs = (int)a.rs;
    float out = 0.0f;
    for (int r = 0; r < nrows; r++) {
        float acc = 0.0f;
        for (int it = 0; it < niter; it++) {
            const int  u_raw = it * 64 + lane;
            const bool valid = u_raw < nunits;
            const int  u     = valid ? u_raw : nunits - 1;
            const int  lastu = it * 64 + 63 < nunits ? it * 64 + 63 : nunits - 1;
            const int  E     = r * rs + (lastu >> 2) * 144 + 144;
            wait_vm_le(issued - ((E + 1023) >> 10));
            const int o = r * rs + (u >> 2) * 144, g = u & 3;
            U::Raw raw;
            raw.hdr = *(const __attribute__((address_space(3))) v4i *)(rg + (o & (S - 1)));
            raw.q0  = *(const __attribute__((address_space(3))) v4i *)(rg + ((o + 16 + g * 32) & (S - 1)));
            raw.q1  = *(const __attribute__((address_space(3))) v4i *)(rg + ((o + 32 + g * 32) & (S - 1)));
            if (MATH & 1) {
                const U::Act av = U::load_act(ActLds{lq, lds_}, u);
                const float v = U::dot(U::decode(raw, u), av);
                acc += valid ? v : 0.0f;
            } else {
                const v4i x = raw.hdr ^ raw.q0 ^ raw.q1;
                acc += __int_as_float((x.x ^ x.y ^ x.z ^ x.w) & 0x007fffff);
            }
            const int C = it == niter - 1 ? (r + 1) * rs : r * rs + (it * 16 + 16) * 144;
            asm volatile("" : "+v"(acc));    // the reads above are consumed before their slot is refilled
            issue_upto((C >> 10) + D);
        }
        const float s = wave_sum(acc);
        out = lane == r ? s : out;
    }
    if (lane < nrows) a.dst[row0 + lane] = out;
}

} // namespace

template <int NW, int D, int MATH>
static void launch_dma(const KDma & a, hipStream_t st) {
    const int kq = (a.k + 1023) & ~1023, dsq = (a.k / QK * 8 + 1023) & ~1023;
    const size_t need = (size_t)kq + dsq + (size_t)NW * D * 1024;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void *)dma_q4k_kernel<NW, D, MATH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
    if (need > 160 * 1024) { fprintf(stderr, "lds %zu too large\n", need); exit(1); }
    const int grid = (a.m + NW * a.rpw - 1) / (NW * a.rpw);
    hipLaunchKernelGGL((dma_q4k_kernel<NW, D, MATH>), dim3(grid), dim3(NW * 64), need, st, a);
}

struct Case { int64_t m, k; };

int main(int argc, char ** argv) {
    std::vector<Case> cases = {{4096, 4096}, {6144, 4096}, {14336, 4096}, {28672, 4096}, {4096, 14336}, {128256, 4096}};
    const size_t pool_bytes = (size_t)3 << 30;
    uint8_t * pool; CK(hipMalloc(&pool, pool_bytes));
    {
        std::vector<uint32_t> h(pool_bytes / 4);
        uint32_t s = 12345;
        for (auto & x : h) { s = s * 1664525u + 1013904223u; x = (s >> 3) & 0x3bff3bffu; }
        CK(hipMemcpy(pool, h.data(), pool_bytes, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%7s %7s | %-22s %9s %9s %7s\n", "m", "k", "variant", "us_cold", "GB/s", "%8TB/s");
    for (const Case & c : cases) {
        const int64_t rs = mi355_row_size(MI355_TYPE_Q4_K, c.k);
        const size_t wbytes = (size_t)rs * c.m, wstride = (wbytes + 4095) / 4096 * 4096;
        const int nmat = (int)(pool_bytes / wstride);
        float * x; CK(hipMalloc(&x, (size_t)c.k * 4));
        {
            std::vector<float> h((size_t)c.k);
            for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
            CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
        int8_t * aq; float * ads, * dst, * dst2;
        CK(hipMalloc(&aq, mi355_act_q_bytes(c.k, 1))); CK(hipMalloc(&ads, mi355_act_ds_bytes(c.k, 1)));
        CK(hipMalloc(&dst, (size_t)c.m * 4)); CK(hipMalloc(&dst2, (size_t)c.m * 4));
        int rc = mi355_quantize_act(x, aq, ads, c.k, 1, 1, 1, c.k * 4, 0, 0, st);
        if (rc) { fprintf(stderr, "quantize: %s\n", mi355_error_string(rc)); return 1; }
        mi355_mmvq_args pa; memset(&pa, 0, sizeof(pa));
        pa.type = MI355_TYPE_Q4_K; pa.w_row_stride = rs; pa.act_q = aq; pa.act_ds = ads; pa.dst = dst; pa.dst_col_stride = c.m;
        pa.m = c.m; pa.k = c.k; pa.ncols = 1; pa.nchannels = 1; pa.ch_ratio = 1;
        KDma ka; ka.rs = rs; ka.m = (int)c.m; ka.k = (int)c.k; ka.act_q = aq; ka.act_ds = ads; ka.dst = dst2;
        const double bytes = (double)wbytes + 4.0 * c.k + 4.0 * c.m;
        for (int v = 0; v < 7; v++) {
            const char * name = "";
            auto run = [&](const uint8_t * W) {
                pa.W = W; ka.W = W;
                switch (v) {
                    case 0: name = "production"; rc = mi355_mul_mat_vec_q(&pa, st); if (rc) { fprintf(stderr, "mmvq %d\n", rc); exit(1); } break;
                    case 1: name = "dma 16w D8";        ka.rpw = (int)((c.m + 4095) / 4096); launch_dma<16, 8, 1>(ka, st); break;
                    case 2: name = "dma 16w D8 nomath"; ka.rpw = (int)((c.m + 4095) / 4096); launch_dma<16, 8, 0>(ka, st); break;
                    case 3: name = "dma 16w D4";        ka.rpw = (int)((c.m + 4095) / 4096); launch_dma<16, 4, 1>(ka, st); break;
                    case 4: name = "dma 16w D8 nt";     ka.rpw = (int)((c.m + 4095) / 4096); launch_dma<16, 8, 3>(ka, st); break;
                    case 5: name = "dma 16w D4 nt";     ka.rpw = (int)((c.m + 4095) / 4096); launch_dma<16, 4, 3>(ka, st); break;
                    case 6: name = "dma 16w D8 nt nomath"; ka.rpw = (int)((c.m + 4095) / 4096); launch_dma<16, 8, 2>(ka, st); break;
                }
            };
            // correctness against the production kernel on matrix 0
            if (v > 0 && v != 2 && v != 6) {
                CK(hipMemsetAsync(dst2, 0xff, (size_t)c.m * 4, st));
                pa.dst = dst; const int vv = v; (void)vv;
                { const int keep = v; v = 0; run(pool); v = keep; }
                run(pool);
                CK(hipStreamSynchronize(st));
                std::vector<uint32_t> h1(c.m), h2(c.m);
                CK(hipMemcpy(h1.data(), dst, c.m * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dst2, c.m * 4, hipMemcpyDeviceToHost));
                size_t bad = 0; for (int64_t i = 0; i < c.m; i++) bad += h1[i] != h2[i];
                if (bad) printf("   MISMATCH variant %d: %zu of %lld rows differ (first: %08x vs %08x)\n", v, bad, (long long)c.m, h1[0], h2[0]);
            }
            const int iters = nmat < 200 ? nmat * (200 / nmat + 1) : 200;
            for (int w = 0; w < 3; w++) run(pool + (size_t)(w % nmat) * wstride);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) run(pool + (size_t)((i + 3) % nmat) * wstride);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("%7lld %7lld | %-22s %9.2f %9.1f %6.1f%%\n", (long long)c.m, (long long)c.k, name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0);
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(aq)); CK(hipFree(ads)); CK(hipFree(dst)); CK(hipFree(dst2));
    }
    return 0;
}
