// llama_bench_driver -- tg / pp timing through the reference's llama.h, the way its llama-bench tool
// measures them (tools/llama-bench/llama-bench.cpp:1747-1795: test_prompt = one llama_decode per
// n_batch chunk of random tokens, test_gen = n_gen single-token decodes, each followed by
// llama_synchronize; warm-up run first; reps timed with a monotonic clock).  llama-bench itself needs
// the generated build-info of the reference's cmake build, so this small driver stands in for it; the
// backend under test is picked up the same way (ggml_backend_load_all + GGML_BACKEND_PATH).
//   mi355-llama-bench -m model.gguf [-p 512] [-n 128] [-r 3] [-ngl 99] [-fa 1] [-ctk f16] [-ctv f16] [-t threads] [-b 2048] [-ub 512] [-sm none|layer|row] [-mg 0] [-ts 1,1]
#include "ggml.h"
#include "ggml-backend.h"
#include "llama.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static ggml_type kv_type(const std::string & s) {
    if (s == "f16") return GGML_TYPE_F16;
    if (s == "q8_0") return GGML_TYPE_Q8_0;
    if (s == "q4_0") return GGML_TYPE_Q4_0;
    if (s == "bf16") return GGML_TYPE_BF16;
    if (s == "f32") return GGML_TYPE_F32;
    fprintf(stderr, "unknown kv type %s\n", s.c_str()); exit(2);
}

// --dump-nodes: every graph node whose name starts with one of the prefixes is written (f32 / i32, rows made contiguous) to
// `path`, with a line "name type ne0 ne1 ne2 ne3 byte_offset" in `path`.idx -- the scheduler's eval callback
// (ggml_backend_sched_set_eval_callback, ggml-backend.cpp) computes the graph up to each such node, so whatever lies between
// two dumped nodes still reaches the backend as one (fusable) sub-graph.
struct node_dump {
    FILE * bin = nullptr, * idx = nullptr;
    std::vector<std::string> prefixes;
    std::vector<char> buf;
    long long off = 0;
};
static bool node_dump_cb(struct ggml_tensor * t, bool ask, void * user) {
    node_dump * d = (node_dump *)user;
    bool match = false;
    for (const std::string & p : d->prefixes) if (strncmp(t->name, p.c_str(), p.size()) == 0) { match = true; break; }
    if (ask) return match;
    if (!match || (t->type != GGML_TYPE_F32 && t->type != GGML_TYPE_I32)) return true;
    const size_t row = (size_t)t->ne[0] * 4;
    d->buf.resize(row);
    fprintf(d->idx, "%s %s %lld %lld %lld %lld %lld\n", t->name, ggml_type_name(t->type), (long long)t->ne[0], (long long)t->ne[1],
            (long long)t->ne[2], (long long)t->ne[3], d->off);
    for (int64_t i3 = 0; i3 < t->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < t->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < t->ne[1]; i1++) {
                ggml_backend_tensor_get(t, d->buf.data(), (size_t)(i1 * t->nb[1] + i2 * t->nb[2] + i3 * t->nb[3]), row);
                fwrite(d->buf.data(), 1, row, d->bin);
                d->off += (long long)row;
            }
    return true;
}

static bool test_prompt(llama_context * ctx, int n_prompt, int n_batch, int n_vocab) {
    std::vector<llama_token> tokens(n_batch);
    int done = 0;
    while (done < n_prompt) {
        const int n = std::min(n_prompt - done, n_batch);
        for (int i = 0; i < n; i++) tokens[i] = rand() % n_vocab;
        if (llama_decode(ctx, llama_batch_get_one(tokens.data(), n)) != 0) return false;
        done += n;
    }
    llama_synchronize(ctx);
    return true;
}

static bool test_gen(llama_context * ctx, int n_gen, int n_vocab) {
    llama_token tok = rand() % n_vocab;
    for (int i = 0; i < n_gen; i++) {
        if (llama_decode(ctx, llama_batch_get_one(&tok, 1)) != 0) return false;
        llama_synchronize(ctx);
        tok = rand() % n_vocab;
    }
    return true;
}

int main(int argc, char ** argv) {
    std::string model_path, ctk = "f16", ctv = "f16", dump_path, sm = "layer", ts_arg, nodes_path, node_prefixes = "l_out,result_norm";
    int main_gpu = 0;
    int n_prompt = 512, n_gen = 128, reps = 3, ngl = 99, fa = 1, threads = 8, n_batch = 2048, n_ubatch = 512;
    int dump_steps = 6;          // single-token decodes of the parity mode
    int big_batch = 160;         // its last, prefill-sized batch
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return std::string(argv[++i]); };
        if (a == "-m") model_path = next(); else if (a == "-p") n_prompt = atoi(next().c_str()); else if (a == "-n") n_gen = atoi(next().c_str());
        else if (a == "-r") reps = atoi(next().c_str()); else if (a == "-ngl") ngl = atoi(next().c_str()); else if (a == "-fa") fa = atoi(next().c_str());
        else if (a == "-ctk") ctk = next(); else if (a == "-ctv") ctv = next(); else if (a == "-t") threads = atoi(next().c_str());
        else if (a == "--dump-logits") dump_path = next();
        else if (a == "--decode-steps") dump_steps = atoi(next().c_str());
        else if (a == "--big-batch") big_batch = atoi(next().c_str());
        else if (a == "--dump-nodes") nodes_path = next(); else if (a == "--node-prefix") node_prefixes = next();
        else if (a == "-sm") sm = next(); else if (a == "-mg") main_gpu = atoi(next().c_str()); else if (a == "-ts") ts_arg = next();
        else if (a == "-b") n_batch = atoi(next().c_str()); else if (a == "-ub") n_ubatch = atoi(next().c_str());
        else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
    }
    if (model_path.empty()) { fprintf(stderr, "usage: %s -m model.gguf [...]\n", argv[0]); return 2; }

    ggml_backend_load_all();
    llama_backend_init();
    std::string devs;
    for (size_t i = 0; i < ggml_backend_dev_count(); i++) {
        devs += (i ? ", " : "") + std::string(ggml_backend_dev_name(ggml_backend_dev_get(i)));
    }

    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = ngl;
    mp.use_mmap = true;
    // -sm none|layer|row, -mg main gpu, -ts a,b,... (llama-bench.cpp:1018-1035 has the same switches)
    mp.split_mode = sm == "none" ? LLAMA_SPLIT_MODE_NONE : sm == "row" ? LLAMA_SPLIT_MODE_ROW : LLAMA_SPLIT_MODE_LAYER;
    mp.main_gpu = main_gpu;
    static float tensor_split[128] = {0};
    if (!ts_arg.empty()) {
        size_t pos = 0; int n = 0;
        while (pos <= ts_arg.size() && n < 128) {
            size_t e = ts_arg.find(',', pos);
            if (e == std::string::npos) e = ts_arg.size();
            tensor_split[n++] = (float)atof(ts_arg.substr(pos, e - pos).c_str());
            pos = e + 1;
        }
        mp.tensor_split = tensor_split;
    }
    llama_model * model = llama_model_load_from_file(model_path.c_str(), mp);
    if (!model) { fprintf(stderr, "failed to load %s\n", model_path.c_str()); return 1; }
    const int n_vocab = llama_vocab_n_tokens(llama_model_get_vocab(model));

    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = std::max(n_prompt + n_gen, 256) + (dump_path.empty() ? 0 : 112 + big_batch + dump_steps);
    cp.n_batch = n_batch; cp.n_ubatch = n_ubatch;
    cp.n_threads = threads; cp.n_threads_batch = threads;
    cp.flash_attn = fa != 0;
    cp.type_k = kv_type(ctk); cp.type_v = kv_type(ctv);
    cp.no_perf = true;
    node_dump nd;
    if (!nodes_path.empty()) {
        nd.bin = fopen(nodes_path.c_str(), "wb");
        nd.idx = fopen((nodes_path + ".idx").c_str(), "w");
        if (!nd.bin || !nd.idx) { perror(nodes_path.c_str()); return 1; }
        size_t pos = 0;
        while (pos <= node_prefixes.size()) {
            size_t e = node_prefixes.find(',', pos);
            if (e == std::string::npos) e = node_prefixes.size();
            if (e > pos) nd.prefixes.push_back(node_prefixes.substr(pos, e - pos));
            pos = e + 1;
        }
        cp.cb_eval = node_dump_cb;
        cp.cb_eval_user_data = &nd;
    }
    llama_context * ctx = llama_init_from_model(model, cp);
    if (!ctx) { fprintf(stderr, "failed to create context\n"); return 1; }

    srand(1234);
    if (!dump_path.empty()) {
        // parity mode: a fixed 48-token prompt in one batch, then 6 single-token decodes fed with fixed tokens, then two
        // small batches (3 and 2 tokens: the multi-column mat-vec path), one of 12 tokens (the skinny kernels and the q8_K planes
        // written by the norm kernel) and one of 160 tokens (--big-batch; the prefill GEMM path and its fused bf16 producers); the logits of the last prompt token and of the last token of every later step are written
        // as raw f32
        std::vector<llama_token> prompt(48);
        for (size_t i = 0; i < prompt.size(); i++) prompt[i] = (llama_token)((i * 7919 + 13) % n_vocab);
        FILE * f = fopen(dump_path.c_str(), "wb");
        if (!f) { perror(dump_path.c_str()); return 1; }
        if (llama_decode(ctx, llama_batch_get_one(prompt.data(), (int)prompt.size())) != 0) return 1;
        fwrite(llama_get_logits_ith(ctx, -1), sizeof(float), n_vocab, f);
        for (int s = 0; s < dump_steps; s++) {
            llama_token tok = (llama_token)((s * 104729 + 71) % n_vocab);
            if (llama_decode(ctx, llama_batch_get_one(&tok, 1)) != 0) return 1;
            fwrite(llama_get_logits_ith(ctx, -1), sizeof(float), n_vocab, f);
        }
        for (int nb : {3, 2, 12, big_batch}) {
            std::vector<llama_token> toks(nb);
            for (int t = 0; t < nb; t++) toks[t] = (llama_token)(((long long)t * 15485863 + nb * 31) % n_vocab);
            if (llama_decode(ctx, llama_batch_get_one(toks.data(), nb)) != 0) return 1;
            fwrite(llama_get_logits_ith(ctx, -1), sizeof(float), n_vocab, f);
        }
        fclose(f);
        if (nd.bin) { fclose(nd.bin); fclose(nd.idx); }
        printf("{\"dumped\": \"%s\", \"n_vocab\": %d, \"rows\": %d, \"devices\": \"%s\"}\n", dump_path.c_str(), n_vocab, 5 + dump_steps, devs.c_str());
        llama_free(ctx); llama_model_free(model); llama_backend_free();
        return 0;
    }
    std::vector<double> pp_ts, tg_ts;
    // warm-up (llama-bench.cpp:1925-1946)
    if (n_prompt > 0) test_prompt(ctx, n_prompt, n_batch, n_vocab);
    if (n_gen > 0) test_gen(ctx, 1, n_vocab);
    for (int r = 0; r < reps; r++) {
        if (n_prompt > 0) {
            llama_memory_clear(llama_get_memory(ctx), false);
            const double t0 = now_s();
            if (!test_prompt(ctx, n_prompt, n_batch, n_vocab)) { fprintf(stderr, "pp failed\n"); return 1; }
            pp_ts.push_back(n_prompt / (now_s() - t0));
        }
        if (n_gen > 0) {
            llama_memory_clear(llama_get_memory(ctx), false);
            const double t0 = now_s();
            if (!test_gen(ctx, n_gen, n_vocab)) { fprintf(stderr, "tg failed\n"); return 1; }
            tg_ts.push_back(n_gen / (now_s() - t0));
        }
    }
    auto stats = [](const std::vector<double> & v, double & avg, double & sd) {
        avg = 0; sd = 0;
        if (v.empty()) return;
        for (double x : v) avg += x;
        avg /= v.size();
        for (double x : v) sd += (x - avg) * (x - avg);
        sd = v.size() > 1 ? std::sqrt(sd / (v.size() - 1)) : 0;
    };
    double pa, ps, ta, ts;
    stats(pp_ts, pa, ps); stats(tg_ts, ta, ts);
    char desc[256];
    llama_model_desc(model, desc, sizeof(desc));
    printf("{\"model\": \"%s\", \"model_size\": %llu, \"n_params\": %llu, \"devices\": \"%s\", \"ngl\": %d, \"fa\": %d, \"type_k\": \"%s\", "
           "\"type_v\": \"%s\", \"n_threads\": %d, \"n_batch\": %d, \"n_ubatch\": %d, \"reps\": %d, "
           "\"pp\": %d, \"pp_avg_ts\": %.3f, \"pp_stddev_ts\": %.3f, \"tg\": %d, \"tg_avg_ts\": %.3f, \"tg_stddev_ts\": %.3f}\n",
           desc, (unsigned long long)llama_model_size(model), (unsigned long long)llama_model_n_params(model), devs.c_str(), ngl, fa,
           ctk.c_str(), ctv.c_str(), threads, n_batch, n_ubatch, reps, n_prompt, pa, ps, n_gen, ta, ts);
    llama_free(ctx);
    llama_model_free(model);
    llama_backend_free();
    return 0;
}
