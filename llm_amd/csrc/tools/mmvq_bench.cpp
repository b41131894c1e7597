// mmvq_bench -- native micro-benchmark of the decode mat-vec kernels through the C ABI.
// Cold-cache numbers: the weights rotate over a pool larger than the 256 MiB Infinity Cache so
// every launch streams from HBM; "hot" re-runs one matrix.
//   mmvq_bench [type m k ncols]...   (no args: the Llama-3-8B / 70B decode shapes)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi355_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static int type_of(const std::string & s) {
    if (s == "q4_0") return MI355_TYPE_Q4_0;
    if (s == "q8_0") return MI355_TYPE_Q8_0;
    if (s == "q4_K") return MI355_TYPE_Q4_K;
    if (s == "q5_K") return MI355_TYPE_Q5_K;
    if (s == "q6_K") return MI355_TYPE_Q6_K;
    fprintf(stderr, "unknown type %s\n", s.c_str()); exit(2);
}

struct Case { std::string tn; int64_t m, k; int n; };

int main(int argc, char ** argv) {
    std::vector<Case> cases;
    for (int i = 1; i + 3 < argc; i += 4) cases.push_back({argv[i], atoll(argv[i + 1]), atoll(argv[i + 2]), atoi(argv[i + 3])});
    if (cases.empty()) {
        for (const char * t : {"q4_K", "q6_K", "q4_0", "q8_0", "q5_K"}) {
            cases.push_back({t, 4096, 4096, 1});
            cases.push_back({t, 14336, 4096, 1});
            cases.push_back({t, 4096, 14336, 1});
        }
        cases.push_back({"q4_K", 1024, 4096, 1});
        cases.push_back({"q6_K", 128256, 4096, 1});
        cases.push_back({"q4_K", 28672, 8192, 1});
        cases.push_back({"q4_K", 8192, 28672, 1});
        for (int n : {2, 4, 8}) cases.push_back({"q4_K", 14336, 4096, n});
    }
    const size_t pool_bytes = (size_t)3 << 30; // 3 GiB of random bytes
    uint8_t * pool; CK(hipMalloc(&pool, pool_bytes));
    {
        std::vector<uint32_t> h(pool_bytes / 4);
        uint32_t s = 12345;
        for (auto & x : h) { s = s * 1664525u + 1013904223u; x = (s >> 3) & 0x3bff3bffu; } // fp16 fields finite
        CK(hipMemcpy(pool, h.data(), pool_bytes, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-6s %7s %7s %2s %10s %10s %9s %9s  %s\n", "type", "m", "k", "n", "us_cold", "GB/s_cold", "us_hot", "GB/s_hot", "%8TB/s(cold)");
    for (const Case & c : cases) {
        const int type = type_of(c.tn);
        const int64_t rs = mi355_row_size(type, c.k);
        const size_t wbytes = (size_t)rs * c.m;
        const size_t wstride = (wbytes + 4095) / 4096 * 4096;
        const int nmat = (int)(pool_bytes / wstride);
        if (nmat < 1) { printf("%s too large\n", c.tn.c_str()); continue; }
        float * x; CK(hipMalloc(&x, (size_t)c.k * c.n * 4));
        {
            std::vector<float> h((size_t)c.k * c.n);
            for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
            CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
        int8_t * aq; float * ads; float * dst;
        CK(hipMalloc(&aq, mi355_act_q_bytes(c.k, c.n)));
        CK(hipMalloc(&ads, mi355_act_ds_bytes(c.k, c.n)));
        CK(hipMalloc(&dst, (size_t)c.m * c.n * 4));
        int rc = mi355_quantize_act(x, aq, ads, c.k, c.n, 1, 1, c.k * 4, 0, 0, st);
        if (rc) { fprintf(stderr, "quantize: %s\n", mi355_error_string(rc)); return 1; }
        mi355_mmvq_args a; memset(&a, 0, sizeof(a));
        a.type = type; a.w_row_stride = rs; a.act_q = aq; a.act_ds = ads; a.dst = dst; a.dst_col_stride = c.m;
        a.m = c.m; a.k = c.k; a.ncols = c.n; a.nchannels = 1; a.ch_ratio = 1;
        const double bytes = (double)wbytes + 4.0 * c.k * c.n + 4.0 * c.m * c.n;
        double us[2];
        for (int hot = 0; hot < 2; hot++) {
            const int iters = hot ? 200 : (nmat < 200 ? nmat * (200 / nmat + 1) : 200);
            for (int w = 0; w < 3; w++) { a.W = pool + (size_t)(w % nmat) * wstride; rc = mi355_mul_mat_vec_q(&a, st); }
            if (rc) { fprintf(stderr, "mmvq: %s\n", mi355_error_string(rc)); return 1; }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) {
                a.W = pool + (size_t)(hot ? 0 : (i + 3) % nmat) * wstride;
                mi355_mul_mat_vec_q(&a, st);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[hot] = ms * 1e3 / iters;
        }
        printf("%-6s %7lld %7lld %2d %10.2f %10.1f %9.2f %9.1f  %5.1f%%\n", c.tn.c_str(), (long long)c.m, (long long)c.k, c.n,
               us[0], bytes / us[0] * 1e-3, us[1], bytes / us[1] * 1e-3, bytes / us[0] * 1e-3 / 8000.0 * 100.0);
        fflush(stdout);
        CK(hipFree(x)); CK(hipFree(aq)); CK(hipFree(ads)); CK(hipFree(dst));
    }
    return 0;
}
