// test_split_mul_mat -- row-split (`-sm row`) MUL_MAT through the public ggml API.
//
// For every weight type and a set of shapes: the same quantized matrix is stored once in the plugin's
// split buffer type (get_proc_address "ggml_backend_split_buffer_type", ggml-backend.h:188) and once in
// the ordinary device buffer; MUL_MAT(W_split, x) and MUL_MAT(W_plain, x) run on the main device's
// backend and must agree BIT FOR BIT (each output row is computed by the same kernel code either way),
// and get_tensor of the split tensor must return the bytes that were set.  Needs >= 2 MI355 devices:
// on a single-GPU box run with GGML_MI355_VIRTUAL_DEVICES=2, 4 or 8 (all ggml devices map to the one GPU, so
// helper streams, the activation copy, cross-stream events and the row offsets are all exercised).
//   GGML_BACKEND_PATH=.../libggml-mi355.so GGML_MI355_VIRTUAL_DEVICES=2 test-split-mul-mat
#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

typedef ggml_backend_buffer_type_t (*split_buft_fn)(int main_device, const float * tensor_split);

static int n_fail = 0, n_ok = 0;

static void check(bool ok, const std::string & what) {
    if (ok) { n_ok++; } else { n_fail++; printf("FAIL: %s\n", what.c_str()); }
}

static bool run_case(ggml_backend_reg_t reg, split_buft_fn split_buft, int main_dev, const float * ts, ggml_type type,
                     int64_t M, int64_t K, int64_t N, int64_t ne2) {
    char tag[256];
    int tl = snprintf(tag, sizeof(tag), "%s M=%lld K=%lld N=%lld ne2=%lld main=%d ts=", ggml_type_name(type), (long long)M, (long long)K,
                      (long long)N, (long long)ne2, main_dev);
    for (int d = 0; ts && d < (int)ggml_backend_reg_dev_count(reg) && tl < (int)sizeof(tag) - 8; d++) tl += snprintf(tag + tl, sizeof(tag) - tl, d ? ",%g" : "%g", ts[d]);
    ggml_backend_dev_t dev = ggml_backend_reg_dev_get(reg, main_dev);
    ggml_backend_t backend = ggml_backend_dev_init(dev, nullptr);
    ggml_backend_buffer_type_t sbuft = split_buft(main_dev, ts);
    if (!backend || !sbuft) { check(false, std::string(tag) + ": no backend / split buffer type"); return false; }

    // weights: one context per buffer type
    ggml_init_params ip = { ggml_tensor_overhead() * 8, nullptr, true };
    ggml_context * cs = ggml_init(ip);
    ggml_context * cp = ggml_init(ip);
    ggml_tensor * Ws = ggml_new_tensor_2d(cs, type, K, M);
    ggml_tensor * Wp = ggml_new_tensor_2d(cp, type, K, M);
    ggml_backend_buffer_t bs = ggml_backend_alloc_ctx_tensors_from_buft(cs, sbuft);
    ggml_backend_buffer_t bp = ggml_backend_alloc_ctx_tensors_from_buft(cp, ggml_backend_dev_buffer_type(dev));
    if (!bs || !bp) { check(false, std::string(tag) + ": weight allocation"); return false; }

    std::mt19937 rng(1234 + (unsigned)(M * 31 + K * 7 + N));
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    std::vector<float> wf(M * K);
    for (float & v : wf) v = U(rng);
    std::vector<uint8_t> wq(ggml_nbytes(Ws));
    ggml_quantize_chunk(type, wf.data(), wq.data(), 0, M, K, nullptr);
    ggml_backend_tensor_set(Ws, wq.data(), 0, wq.size());
    ggml_backend_tensor_set(Wp, wq.data(), 0, wq.size());
    std::vector<uint8_t> back(wq.size(), 0xee);
    ggml_backend_tensor_get(Ws, back.data(), 0, back.size());
    check(memcmp(back.data(), wq.data(), wq.size()) == 0, std::string(tag) + ": set/get round trip of the split tensor");

    // graph on the main device
    ggml_init_params ig = { ggml_tensor_overhead() * 16 + ggml_graph_overhead(), nullptr, true };
    ggml_context * cg = ggml_init(ig);
    ggml_tensor * x  = ggml_new_tensor_3d(cg, GGML_TYPE_F32, K, N, ne2);
    ggml_tensor * ys = ggml_mul_mat(cg, Ws, x);
    ggml_tensor * yp = ggml_mul_mat(cg, Wp, x);
    const bool sup = ggml_backend_dev_supports_op(dev, ys);
    if (!sup) {   // a main device that owns no rows must refuse the op (ggml-cuda.cu:3141-3149)
        printf("  %-60s not supported on MI355_%d (main device owns no rows)\n", tag, main_dev);
        ggml_free(cg); ggml_backend_buffer_free(bs); ggml_backend_buffer_free(bp); ggml_free(cs); ggml_free(cp); ggml_backend_free(backend);
        return false;
    }
    ggml_cgraph * gf = ggml_new_graph(cg);
    ggml_build_forward_expand(gf, ys);
    ggml_build_forward_expand(gf, yp);
    ggml_backend_buffer_t bg = ggml_backend_alloc_ctx_tensors(cg, backend);
    std::vector<float> xf(K * N * ne2);
    for (float & v : xf) v = U(rng);
    ggml_backend_tensor_set(x, xf.data(), 0, xf.size() * sizeof(float));
    std::vector<float> poison(M * N * ne2, -12345.0f);
    ggml_backend_tensor_set(ys, poison.data(), 0, poison.size() * sizeof(float));
    bool ok = true;
    for (int rep = 0; rep < 2 && ok; rep++) {     // twice: the second run finds the activation copy already on the helper
        ok = ggml_backend_graph_compute(backend, gf) == GGML_STATUS_SUCCESS;
    }
    check(ok, std::string(tag) + ": graph_compute");
    std::vector<float> a(M * N * ne2), b(M * N * ne2);
    ggml_backend_tensor_get(ys, a.data(), 0, a.size() * sizeof(float));
    ggml_backend_tensor_get(yp, b.data(), 0, b.size() * sizeof(float));
    const bool same = memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
    double ref2 = 0;
    for (float v : b) ref2 += (double)v * v;
    check(same && ref2 > 0, std::string(tag) + ": split result == unsplit result (bitwise)");
    printf("  %-60s %s\n", tag, same ? "OK" : "MISMATCH");

    ggml_backend_buffer_free(bg); ggml_free(cg);
    ggml_backend_buffer_free(bs); ggml_backend_buffer_free(bp);
    ggml_free(cs); ggml_free(cp);
    ggml_backend_free(backend);
    return same;
}

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355");
    if (!reg) { printf("MI355 backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    const int ndev = (int)ggml_backend_reg_dev_count(reg);
    split_buft_fn split_buft = (split_buft_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_split_buffer_type");
    if (!split_buft) { printf("no ggml_backend_split_buffer_type\n"); return 1; }
    if (ndev == 1) {
        // one physical device: the "split" is the whole matrix on it.  What this leg is for: GGML_MI355_SPLIT_GATHER=rccl with a
        // one-rank communicator -- ncclCommInitAll, the local slice, the all-gather and the placement into dst run on hardware
        // (RCCL refuses the virtual devices of the other legs: they share a GPU), against the unsplit result, bit for bit
        const float one[16] = {1};
        const ggml_type types1[] = { GGML_TYPE_Q4_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K };
        int ran1 = 0;
        for (ggml_type t : types1) {
            ran1 += run_case(reg, split_buft, 0, one, t, 1024, 1024, 1, 1);
            ran1 += run_case(reg, split_buft, 0, one, t, 4096, 512, 5, 2);      // broadcast over ne2: 10 columns
            ran1 += run_case(reg, split_buft, 0, one, t, 2048, 1024, 40, 1);
            ran1 += run_case(reg, split_buft, 0, one, t, 4096, 4096, 512, 1);
        }
        printf("%d checks ok, %d failed, %d cases ran (of %d)\n", n_ok, n_fail, ran1, 20);
        return n_fail == 0 && ran1 >= 20 ? 0 : 1;
    }
    if (ndev < 2) { printf("need MI355 devices (GGML_MI355_VIRTUAL_DEVICES=2), have %d\n", ndev); return 2; }
    // the same (main_device, tensor_split) must give the same object (buffer types are compared by pointer)
    const float even[16] = {1, 1}, skew[16] = {3, 1};
    check(split_buft(0, even) == split_buft(0, even), "split buffer type is a singleton per (main, split)");
    check(split_buft(0, even) != split_buft(1, even), "distinct main devices give distinct buffer types");
    check(split_buft(0, even) != split_buft(0, skew), "distinct splits give distinct buffer types");

    const ggml_type types[] = { GGML_TYPE_Q4_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K };
    int ran = 0, want = 0;
    if (ndev == 2) {
        for (ggml_type t : types) {
            ran += run_case(reg, split_buft, 0, even, t, 512, 256, 1, 1);      // equal slices, decode
            ran += run_case(reg, split_buft, 0, even, t, 384, 512, 3, 1);      // 128 / 256 rows
            ran += run_case(reg, split_buft, 1, even, t, 1024, 256, 8, 1);     // the other device is the main one
            ran += run_case(reg, split_buft, 0, skew, t, 1024, 512, 5, 2);     // 768 / 256, broadcast over ne2
            ran += run_case(reg, split_buft, 0, even, t, 512, 512, 33, 1);     // prefill tile kernel on both halves
            ran += run_case(reg, split_buft, 1, skew, t, 640, 256, 64, 1);     // 384 / 256, tile kernel, main = 1
            // the prefill GEMM cuts K in two for small matrices (mmdq.hip): the choice must follow the WHOLE matrix, not
            // the slice, or the f32 summation order of a split matrix differs (K >= 1024, N >= 64)
            ran += run_case(reg, split_buft, 0, even, t, 4096, 4096, 512, 1);
            ran += run_case(reg, split_buft, 1, skew, t, 2048, 1024, 64, 1);
            want += 8;
        }
        // fewer rows than the rounding: device 0 owns nothing and must refuse, device 1 owns all and must run
        check(!run_case(reg, split_buft, 0, even, GGML_TYPE_Q4_K, 100, 256, 1, 1), "main device without rows refuses the op");
        check(run_case(reg, split_buft, 1, even, GGML_TYPE_Q4_K, 100, 256, 1, 1), "device that owns all rows runs it");
    } else {
        // 4 or 8 devices: the row boundaries of Llama-3-70B's matrices (M = 1024: 128 rows per rank at 8; 8192; 28672), a
        // device in the middle that owns no rows, skewed splits, every kernel regime
        float all[16] = {0}, hole[16] = {0}, ramp[16] = {0};
        for (int d = 0; d < ndev; d++) { all[d] = 1.0f; hole[d] = d == 1 ? 0.0f : 1.0f; ramp[d] = (float)(d + 1); }
        const int last = ndev - 1;
        for (ggml_type t : types) {
            ran += run_case(reg, split_buft, 0, all, t, 1024, 1024, 1, 1);        // wk / wv: 128 (or 256) rows per rank
            ran += run_case(reg, split_buft, 0, all, t, 8192, 512, 1, 1);         // wq / wo / down rows
            ran += run_case(reg, split_buft, last, all, t, 28672, 256, 2, 1);     // gate / up rows, the last device is main
            ran += run_case(reg, split_buft, 0, hole, t, 1024, 512, 1, 1);        // device 1 owns nothing
            ran += run_case(reg, split_buft, 2, hole, t, 2048, 1024, 40, 1);      // ... skinny regime, main = 2
            ran += run_case(reg, split_buft, ndev / 2, ramp, t, 4096, 1024, 96, 1); // unequal slices (device 0 may own none), GEMM, K cut decided on M = 4096
            ran += run_case(reg, split_buft, last, ramp, t, 1536, 512, 7, 2);     // broadcast over ne2
            want += 7;
        }
        // the skinny kernels choose their launch geometry (waves per tile, K over two workgroups) by the row count: it must be the WHOLE
        // matrix's, or a slice's f32 sums over K are added in another order than the unsplit matrix's
        ran += run_case(reg, split_buft, 0, all, GGML_TYPE_Q4_K, 28672, 4096, 16, 1);   // >= 384 tiles whole, < 384 per slice (two waves per tile)
        ran += run_case(reg, split_buft, 0, all, GGML_TYPE_Q6_K, 28672, 4096, 48, 1);   // ... with two column tiles per workgroup
        ran += run_case(reg, split_buft, last, all, GGML_TYPE_Q4_K, 16384, 8448, 16, 1); // >= 256 tiles whole (K uncut), < 256 per slice
        ran += run_case(reg, split_buft, 0, all, GGML_TYPE_Q8_0, 16384, 2048, 16, 1);   // the same for the q8_1 kernel
        want += 4;
        ran += run_case(reg, split_buft, 0, all, GGML_TYPE_Q4_K, 8192, 8192, 1, 1);    // 70B wq at full size
        ran += run_case(reg, split_buft, 0, all, GGML_TYPE_Q6_K, 1024, 8192, 1, 1);    // 70B wv (use_more_bits layers)
        ran += run_case(reg, split_buft, 0, all, GGML_TYPE_Q4_K, 8192, 8192, 512, 1);  // ... prefill
        want += 3;
        // the main device owns nothing: refuse
        check(!run_case(reg, split_buft, 1, hole, GGML_TYPE_Q4_K, 1024, 256, 1, 1), "main device without rows refuses the op");
    }
    printf("%d checks ok, %d failed, %d cases ran (of %d)\n", n_ok, n_fail, ran, want);
    return n_fail == 0 && ran >= want ? 0 : 1;
}
