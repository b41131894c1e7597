// ref_bench.cpp -- times the REFERENCE's ggml-cpu MUL_MAT (oracle/_ref libs built from
// /root/reference by oracle/Makefile) on the host cores.  TEST/BENCH INFRASTRUCTURE ONLY: this is
// the "cpu_baseline" leg of bench.py (kind = "reference").
//
//   ref_bench <type: q4_0|q8_0|q4_K|q5_K|q6_K> <m> <k> <n> <threads> <min_seconds> [n_matrices]
//
// n_matrices > 1 rotates over that many distinct weight matrices so the working set exceeds the
// CPU's last-level cache (cold weights, as in decode).  Prints one JSON line.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

int main(int argc, char ** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s type m k n threads min_seconds [n_matrices]\n", argv[0]); return 2; }
    const std::string tn = argv[1];
    const int64_t m = atoll(argv[2]), k = atoll(argv[3]), n = atoll(argv[4]);
    const int threads = atoi(argv[5]);
    const double min_s = atof(argv[6]);
    const int nmat = argc > 7 ? atoi(argv[7]) : 1;
    ggml_type t = GGML_TYPE_COUNT;
    for (int i = 0; i < GGML_TYPE_COUNT; i++) if (tn == ggml_type_name((ggml_type)i)) t = (ggml_type)i;
    if (t == GGML_TYPE_COUNT) { fprintf(stderr, "unknown type %s\n", tn.c_str()); return 2; }

    ggml_backend_t cpu = ggml_backend_cpu_init();
    ggml_backend_cpu_set_n_threads(cpu, threads);

    const size_t wbytes = ggml_row_size(t, k) * m;
    ggml_init_params ip = {(size_t)nmat * (wbytes + 4096) + (size_t)(k * n + m * n) * 4 * (nmat + 1) + (64u << 20), nullptr, false};
    ggml_context * ctx = ggml_init(ip);
    std::mt19937 rng(1234);
    std::vector<float> row(k);
    std::vector<ggml_tensor *> outs;
    ggml_tensor * b = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, k, n);
    for (int64_t i = 0; i < k * n; i++) ((float *)b->data)[i] = (float)(rng() >> 8) * (2.0f / 16777216.0f) - 1.0f;
    ggml_tensor * first = nullptr;
    for (int im = 0; im < nmat; im++) {
        ggml_tensor * a = ggml_new_tensor_2d(ctx, t, k, m);
        if (!first) {
            // quantize 64 distinct random rows with the reference's quantizer and tile them over the
            // matrix; further matrices are copies in distinct memory (contents do not affect timing)
            const int64_t nsrc = m < 64 ? m : 64;
            for (int64_t r = 0; r < nsrc; r++) {
                for (auto & x : row) x = (float)(rng() >> 8) * (2.0f / 16777216.0f) - 1.0f;
                ggml_quantize_chunk(t, row.data(), (char *)a->data + r * ggml_row_size(t, k), 0, 1, k, nullptr);
            }
            for (int64_t r = nsrc; r < m; r++)
                memcpy((char *)a->data + r * ggml_row_size(t, k), (char *)a->data + (r % nsrc) * ggml_row_size(t, k), ggml_row_size(t, k));
            first = a;
        } else {
            memcpy(a->data, first->data, wbytes);
        }
        outs.push_back(ggml_mul_mat(ctx, a, b));
    }
    ggml_cgraph * gf = ggml_new_graph(ctx);
    for (auto * o : outs) ggml_build_forward_expand(gf, o);

    ggml_backend_graph_compute(cpu, gf); // warm-up
    int runs = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    do {
        ggml_backend_graph_compute(cpu, gf);
        runs += nmat;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < min_s);
    const double us = el * 1e6 / runs;
    const double bytes = (double)wbytes + 4.0 * k * n + 4.0 * m * n;
    printf("{\"op\": \"MUL_MAT\", \"type\": \"%s\", \"m\": %lld, \"k\": %lld, \"n\": %lld, \"threads\": %d, \"n_matrices\": %d, "
           "\"runs\": %d, \"us_per_run\": %.3f, \"gbps\": %.3f, \"gflops\": %.3f}\n",
           tn.c_str(), (long long)m, (long long)k, (long long)n, threads, nmat, runs, us,
           bytes / us * 1e-3, 2.0 * m * k * n / us * 1e-3);
    ggml_free(ctx);
    ggml_backend_free(cpu);
    return 0;
}
