// step_lab.hip -- the step-based LDS-DMA weight stream (kernels/step_stream.h) against the production mat-vec, cold weights.
// (NOT product code; tuning evidence only.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I llm_amd/csrc/kernels \
//         llm_amd/csrc/lab/step_lab.hip llm_amd/csrc/kernels/{quantize,api,timing}.hip -o llm_amd/lib/step_lab
#include "../kernels/mmvq.hip"
#include "../kernels/step_stream.h"

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

struct KStep {
    const uint8_t * W; int64_t rs; int m, k, rpw, type;
    const int8_t * act_q; const float * act_ds; float * dst;
};

template <int NW, int D>
__global__ void __launch_bounds__(NW * 64)
step_kernel(const KStep a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int k = a.k;
    const int act_bytes = (k + k / QK * 8 + 255) & ~255;
    const int row0 = (blockIdx.x * NW + wave) * a.rpw;
    int nrows = a.m - row0;
    nrows = nrows < 0 ? 0 : (nrows > a.rpw ? a.rpw : nrows);
    SStream<D> ss;
    ss.s0 = SSeg{a.W + (int64_t)row0 * a.rs, 2 * a.rs, a.rs, nrows, k / QK_K, a.type};
    ss.s1 = ss.s2 = ss.s3 = SSeg{a.W, 0, 0, 0, 16, a.type};
    const unsigned pitch = (unsigned)(a.type == MI355_TYPE_Q6_K ? 16 * 224 : (a.type == MI355_TYPE_Q5_K ? 16 * 176 : 16 * 144));
    ss.init(lds_addr(lds) + act_bytes + wave * D * pitch, pitch, 1);
    ss.prime();
    {
        const int nq = k >> 4, nds = k / QK;
        v2f * lds_ds = (v2f *)(lds + (size_t)k);
        for (int i = threadIdx.x; i < nq; i += NW * 64) ((v4i *)lds)[i] = ((const v4i *)a.act_q)[i];
        for (int i = threadIdx.x; i < nds; i += NW * 64) lds_ds[i] = ((const v2f *)a.act_ds)[i];
    }
    __syncthreads();
    const auto lq   = (const __attribute__((address_space(3))) int8_t *)lds;
    const auto lds_ = (const __attribute__((address_space(3))) float *)(lds + (size_t)k);
    const float out = ss.run_any(ss.s0, 0, ActLds{lq, lds_});
    if (lane < nrows) a.dst[row0 + lane] = out;
}

} // namespace

template <int NW, int D>
static void launch_step(const KStep & a, hipStream_t st) {
    const int act_bytes = (a.k + a.k / QK * 8 + 255) & ~255;
    const size_t need = (size_t)act_bytes + (size_t)NW * D * step_pitch(a.type);
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void *)step_kernel<NW, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); once = true; }
    if (need > 159 * 1024) { fprintf(stderr, "lds %zu too large\n", need); exit(1); }
    const int grid = (a.m + NW * a.rpw - 1) / (NW * a.rpw);
    hipLaunchKernelGGL((step_kernel<NW, D>), dim3(grid), dim3(NW * 64), need, st, a);
}

struct Case { int64_t m, k; int type; };

int main(int argc, char ** argv) {
    std::vector<Case> cases = {{4096, 4096, MI355_TYPE_Q4_K}, {6144, 4096, MI355_TYPE_Q4_K}, {28672, 4096, MI355_TYPE_Q4_K}, {4096, 14336, MI355_TYPE_Q4_K},
                               {4096, 14336, MI355_TYPE_Q6_K}, {128256, 4096, MI355_TYPE_Q6_K}, {28672, 4096, MI355_TYPE_Q5_K}};
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
    printf("%7s %7s %5s | %-22s %9s %9s %7s\n", "m", "k", "type", "variant", "us_cold", "GB/s", "%8TB/s");
    for (const Case & c : cases) {
        const int64_t rs = mi355_row_size(c.type, c.k);
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
        pa.type = c.type; pa.w_row_stride = rs; pa.act_q = aq; pa.act_ds = ads; pa.dst = dst; pa.dst_col_stride = c.m;
        pa.m = c.m; pa.k = c.k; pa.ncols = 1; pa.nchannels = 1; pa.ch_ratio = 1;
        KStep ka; ka.rs = rs; ka.m = (int)c.m; ka.k = (int)c.k; ka.type = c.type; ka.act_q = aq; ka.act_ds = ads; ka.dst = dst2;
        const double bytes = (double)wbytes + 4.0 * c.k + 4.0 * c.m;
        for (int v = 0; v < 5; v++) {
            const char * name = "";
            auto run = [&](const uint8_t * W) {
                pa.W = W; ka.W = W;
                switch (v) {
                    case 0: name = "production"; rc = mi355_mul_mat_vec_q(&pa, st); if (rc) { fprintf(stderr, "mmvq %d\n", rc); exit(1); } break;
                    case 1: name = "step 16w D2"; ka.rpw = (int)((c.m + 4095) / 4096); launch_step<16, 2>(ka, st); break;
                    case 2: name = "step 12w D3"; ka.rpw = (int)((c.m + 3071) / 3072); launch_step<12, 3>(ka, st); break;
                    case 3: name = "step 8w D3"; ka.rpw = (int)((c.m + 4095) / 4096); launch_step<8, 3>(ka, st); break;
                    case 4: name = "step 16w D3"; ka.rpw = (int)((c.m + 4095) / 4096); if (c.type != MI355_TYPE_Q6_K) launch_step<16, 3>(ka, st); else launch_step<16, 2>(ka, st); break;
                }
            };
            if (v > 0) {
                CK(hipMemsetAsync(dst2, 0xff, (size_t)c.m * 4, st));
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
            printf("%7lld %7lld %5d | %-22s %9.2f %9.1f %6.1f%%\n", (long long)c.m, (long long)c.k, c.type, name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0);
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(aq)); CK(hipFree(ads)); CK(hipFree(dst)); CK(hipFree(dst2));
    }
    return 0;
}
