// make_random_gguf -- writes a random-init llama-architecture GGUF with the tensor types of
// llama.cpp's Q4_K_M (or pure Q8_0 / Q4_0) recipe, for end-to-end benchmarks without any download.
// Uses the reference's public gguf.h / ggml.h API (writer: ggml/src/gguf.cpp); metadata keys as read by
// src/llama-model.cpp:447-596; type map per src/llama-quant.cpp:178-404; tokenizer "no_vocab"
// (src/llama-vocab.cpp:1383-1403).  Weights are random *valid* blocks (random quants, finite fp16
// scales), norms are 1.0.
//   make-random-gguf <out.gguf> <8b|70b|8x7b|tiny|tinymoe> [Q4_K_M|Q8_0|Q4_0|Q4_1|Q5_0|Q5_1|Q2_K|Q3_K|IQ4_NL|IQ4_XS|F16] [vocab.gguf]
// (Q2_K / Q3_K / IQ4_NL / IQ4_XS: every matrix in that one type -- "pure" -- not llama-quant's mixtures of the same names)
// 8x7b / tinymoe are Mixtral-style (llama architecture with 8 experts, 2 used): ffn_gate_inp f32, ffn_*_exps 3-D;
// Q4_K_M for 8 experts bumps attn_k / attn_v to Q8_0 and attn_output to Q5_K (llama-quant.cpp:300-311,370-378).
#include "ggml.h"
#include "gguf.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Dims { int n_embd, n_layer, n_head, n_head_kv, n_ff, n_vocab; bool is_70b; int n_expert = 0, n_expert_used = 0; };

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static bool use_more_bits(int i, int n) { return i < n / 8 || i >= 7 * n / 8 || (i - n / 8) % 3 == 2; } // llama-quant.cpp:185-187

static void fill_blocks(ggml_tensor * t) {
    const size_t nbytes = ggml_nbytes(t);
    uint64_t * p = (uint64_t *)t->data;
    for (size_t i = 0; i < nbytes / 8; i++) p[i] = rnd();
    for (size_t i = nbytes / 8 * 8; i < nbytes; i++) ((uint8_t *)t->data)[i] = (uint8_t)rnd();
    if (t->type == GGML_TYPE_F16) {           // an unquantized model: uniform in +-0.02
        ggml_fp16_t * h = (ggml_fp16_t *)t->data;
        for (size_t i = 0; i < nbytes / 2; i++) h[i] = ggml_fp32_to_fp16(((float)(rnd() >> 40) * (1.0f / 16777216.0f) - 0.5f) * 0.04f);
        return;
    }
    const size_t ts = ggml_type_size(t->type), nb = nbytes / ts;
    auto put_half = [&](uint8_t * at, float lo, float hi, bool sign) {
        float f = lo + (hi - lo) * (float)(rnd() >> 40) * (1.0f / 16777216.0f);
        if (sign && (rnd() & 1)) f = -f;
        const ggml_fp16_t h = ggml_fp32_to_fp16(f);
        memcpy(at, &h, 2);
    };
    for (size_t b = 0; b < nb; b++) {
        uint8_t * blk = (uint8_t *)t->data + b * ts;
        switch (t->type) {
            case GGML_TYPE_Q4_0: put_half(blk, 0.01f, 0.05f, true); break;
            case GGML_TYPE_Q5_0: put_half(blk, 0.005f, 0.025f, true); break;
            case GGML_TYPE_Q4_1: put_half(blk, 0.003f, 0.006f, false); put_half(blk + 2, -0.045f, -0.02f, false); break;   // values centred on 0
            case GGML_TYPE_Q5_1: put_half(blk, 0.0015f, 0.003f, false); put_half(blk + 2, -0.045f, -0.02f, false); break;
            case GGML_TYPE_Q8_0: put_half(blk, 0.001f, 0.005f, true); break;
            case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: {
                // w = d sc q - dmin m with q uniform in 0..15 (0..31): centred on 0 when every sub-block's min equals its scale
                // (6-bit fields of get_scale_min_k4, ggml-quants.c) and dmin = 7.5 d (15.5 d).  Uncentred weights (mean ~0.1) made
                // the random models' residual stream grow to 1e6, where a comparison with ggml-cpu sees nothing of a layer's work.
                const float c = t->type == GGML_TYPE_Q4_K ? 7.5f : 15.5f;
                const float dd = 0.0002f + 0.0006f * (float)(rnd() >> 40) * (1.0f / 16777216.0f);
                const ggml_fp16_t h = ggml_fp32_to_fp16(dd), hm = ggml_fp32_to_fp16(c * ggml_fp16_to_fp32(h));
                memcpy(blk, &h, 2); memcpy(blk + 2, &hm, 2);
                uint8_t * q = blk + 4;
                for (int j = 0; j < 4; j++) { q[4 + j] = q[j]; q[8 + j] = (uint8_t)((q[8 + j] & 0x0F) | (q[8 + j] << 4)); }
                break;
            }
            case GGML_TYPE_Q6_K: put_half(blk + 208, 0.00002f, 0.0001f, true); break;
            case GGML_TYPE_Q2_K:
                // w = d (sc & 15) q - dmin (sc >> 4), q uniform in 0..3: centred when every minimum nibble equals its scale nibble and dmin = 1.5 d
                for (int j = 0; j < 16; j++) blk[j] = (uint8_t)((blk[j] & 0x0F) * 0x11);
                {
                    const ggml_fp16_t h = ggml_fp32_to_fp16(0.0015f + 0.002f * (float)(rnd() >> 40) * (1.0f / 16777216.0f));
                    const ggml_fp16_t hm = ggml_fp32_to_fp16(1.5f * ggml_fp16_to_fp32(h));
                    memcpy(blk + 80, &h, 2); memcpy(blk + 82, &hm, 2);
                }
                break;
            case GGML_TYPE_Q3_K:   put_half(blk + 108, 0.0003f, 0.0007f, true); break;
            case GGML_TYPE_IQ4_NL: put_half(blk, 0.0002f, 0.0004f, true); break;
            case GGML_TYPE_IQ4_XS: put_half(blk, 0.00001f, 0.00003f, true); break;
            default: break;
        }
    }
}

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s out.gguf 8b|70b|70b2|8x7b|tiny|tinymoe|mid|wideffn|hd80|hd96|hd256 [Q4_K_M|Q8_0|Q4_0|Q4_1|Q5_0|Q5_1|Q2_K|Q3_K|IQ4_NL|IQ4_XS|F16] [vocab.gguf]\n", argv[0]); return 2; }
    const std::string out = argv[1], model = argv[2];
    std::string ftype = argc > 3 ? argv[3] : "Q4_K_M";
    // "<ftype>+F32": the same random blocks, stored dequantized as F32 -- the model's exact twin, against which both ggml-cpu's and
    // the plugin's activation rounding can be measured
    bool exact = false;
    if (ftype.size() > 4 && ftype.compare(ftype.size() - 4, 4, "+F32") == 0) { exact = true; ftype.resize(ftype.size() - 4); }
    // optional 4th argument: a vocabulary-only GGUF (the reference keeps some for its tokenizer tests, models/ggml-vocab-*.gguf)
    // whose tokenizer.* metadata is copied, so that tools which tokenize text (llama-cli) can run the random model
    const char * vocab_path = argc > 4 ? argv[4] : nullptr;
    Dims d;
    if (model == "8b")        d = {4096, 32, 32, 8, 14336, 128256, false};
    else if (model == "70b")  d = {8192, 80, 64, 8, 28672, 128256, true};
    else if (model == "70b2") d = {8192, 2, 64, 8, 28672, 32000, true};       // two layers of Llama-3-70B's shapes (K = 8192, 28672-row ffn, GQA 8, the 70B type rules): the matchers' other size class
    else if (model == "tiny") d = {1024, 4, 8, 2, 2816, 32000, false};
    else if (model == "mid")  d = {2048, 3, 16, 4, 5120, 32000, false};      // the smallest shapes the persistent decode-layer launch takes (multiples of 1024, >= 2048)
    else if (model == "wideffn") d = {1024, 2, 8, 2, 14336, 32000, false};   // Llama-8B's feed-forward width on a small trunk: a 512-token batch reaches the fused gate / up / GLU GEMM
    // head sizes other than 128: Phi-3-mini's 96 (3072 wide, GQA 4 here), StableLM / Phi-2's 80 (2560 wide, no GQA), Gemma's 256 (2048 wide, GQA 2)
    else if (model == "hd96")  d = {3072, 2, 32, 8, 8192, 8000, false};
    else if (model == "hd80")  d = {2560, 2, 32, 32, 6912, 8000, false};
    else if (model == "hd256") d = {2048, 2, 8, 4, 5632, 8000, false};
    else if (model == "8x7b")    { d = {4096, 32, 32, 8, 14336, 32000, false}; d.n_expert = 8; d.n_expert_used = 2; }
    else if (model == "tinymoe") { d = {1024, 4, 8, 2, 2816, 32000, false};    d.n_expert = 8; d.n_expert_used = 2; }
    else { fprintf(stderr, "unknown model %s\n", model.c_str()); return 2; }
    const ggml_type base = ftype == "Q8_0" ? GGML_TYPE_Q8_0 : ftype == "Q4_0" ? GGML_TYPE_Q4_0 : ftype == "Q4_1" ? GGML_TYPE_Q4_1 :
                           ftype == "Q5_0" ? GGML_TYPE_Q5_0 : ftype == "Q5_1" ? GGML_TYPE_Q5_1 : ftype == "F16" ? GGML_TYPE_F16 :
                           ftype == "Q2_K" ? GGML_TYPE_Q2_K : ftype == "Q3_K" ? GGML_TYPE_Q3_K : ftype == "IQ4_NL" ? GGML_TYPE_IQ4_NL :
                           ftype == "IQ4_XS" ? GGML_TYPE_IQ4_XS : GGML_TYPE_Q4_K;
    const bool km = ftype == "Q4_K_M";
    const int head_dim = d.n_embd / d.n_head, kv_dim = head_dim * d.n_head_kv;

    // size the context: every tensor lives in it until the file is written
    size_t total = 0;
    auto rs = [&](ggml_type, int64_t k, int64_t m) { return ggml_row_size(GGML_TYPE_F16, k) * m + 1024; };   // upper bound over the ftypes
    total += rs(GGML_TYPE_Q8_0, d.n_embd, d.n_vocab) * 2;   // upper bound over the ftypes (token_embd + output)
    for (int i = 0; i < d.n_layer; i++)
        total += rs(GGML_TYPE_Q8_0, d.n_embd, d.n_embd) * 2 + rs(GGML_TYPE_Q8_0, d.n_embd, kv_dim) * 2 + rs(GGML_TYPE_Q8_0, d.n_embd, d.n_ff) * 2 +
                 rs(GGML_TYPE_Q8_0, d.n_ff, d.n_embd) + 3 * (d.n_embd * 4 + 256);
    if (d.n_expert) total = total * d.n_expert + (size_t)d.n_layer * (d.n_embd * d.n_expert * 4 + 1024);   // generous upper bound
    if (exact) total *= 3;
    ggml_init_params ip = {total + (64u << 20), nullptr, false};
    ggml_context * ctx = ggml_init(ip);
    if (!ctx) { fprintf(stderr, "ggml_init failed (%zu bytes)\n", total); return 1; }
    gguf_context * g = gguf_init_empty();
    uint32_t vocab_n = 0;
    if (vocab_path) {
        gguf_init_params vp = {true, nullptr};
        gguf_context * gv = gguf_init_from_file(vocab_path, vp);
        if (!gv) { fprintf(stderr, "cannot read %s\n", vocab_path); return 1; }
        gguf_set_kv(g, gv);            // everything; the model keys below overwrite what they name
        const int64_t kt = gguf_find_key(gv, "tokenizer.ggml.tokens");
        if (kt >= 0) vocab_n = (uint32_t)gguf_get_arr_n(gv, kt);
        gguf_free(gv);
        if (vocab_n) d.n_vocab = (int)vocab_n;
    }

    gguf_set_val_str(g, "general.architecture", "llama");
    gguf_set_val_str(g, "general.name", ("random-llama-" + model + "-" + ftype).c_str());
    gguf_set_val_u32(g, "llama.context_length", 8192);
    gguf_set_val_u32(g, "llama.embedding_length", d.n_embd);
    gguf_set_val_u32(g, "llama.block_count", d.n_layer);
    gguf_set_val_u32(g, "llama.feed_forward_length", d.n_ff);
    gguf_set_val_u32(g, "llama.attention.head_count", d.n_head);
    gguf_set_val_u32(g, "llama.attention.head_count_kv", d.n_head_kv);
    gguf_set_val_f32(g, "llama.attention.layer_norm_rms_epsilon", 1e-5f);
    gguf_set_val_u32(g, "llama.rope.dimension_count", head_dim);
    gguf_set_val_f32(g, "llama.rope.freq_base", 500000.0f);
    gguf_set_val_u32(g, "llama.vocab_size", d.n_vocab);
    gguf_set_val_u32(g, "general.file_type", exact ? 0 : km ? 15 : base == GGML_TYPE_Q8_0 ? 7 : base == GGML_TYPE_F16 ? 1 : base == GGML_TYPE_Q4_1 ? 3 :
                                               base == GGML_TYPE_Q5_0 ? 8 : base == GGML_TYPE_Q5_1 ? 9 : base == GGML_TYPE_Q2_K ? 10 : base == GGML_TYPE_Q3_K ? 11 :
                                               base == GGML_TYPE_IQ4_NL ? 25 : base == GGML_TYPE_IQ4_XS ? 30 : 2);
    if (!vocab_path) gguf_set_val_str(g, "tokenizer.ggml.model", "no_vocab");
    if (d.n_expert) {
        gguf_set_val_u32(g, "llama.expert_count", d.n_expert);
        gguf_set_val_u32(g, "llama.expert_used_count", d.n_expert_used);
    }

    auto finish = [&](ggml_tensor * x, const std::string & name) {
        fill_blocks(x);
        if (exact) {
            ggml_tensor * y = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, x->ne[0], x->ne[1], x->ne[2]);
            ggml_get_type_traits(x->type)->to_float(x->data, (float *)y->data, ggml_nelements(x));
            x = y;
        }
        ggml_set_name(x, name.c_str());
        gguf_add_tensor(g, x);
    };
    auto add_w = [&](const std::string & name, ggml_type t, int64_t k, int64_t m) { finish(ggml_new_tensor_2d(ctx, t, k, m), name); };
    auto add_w3 = [&](const std::string & name, ggml_type t, int64_t k, int64_t m, int64_t e) { finish(ggml_new_tensor_3d(ctx, t, k, m, e), name); };
    auto add_norm = [&](const std::string & name) {
        ggml_tensor * x = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, d.n_embd);
        ggml_set_name(x, name.c_str());
        for (int i = 0; i < d.n_embd; i++) ((float *)x->data)[i] = 1.0f;
        gguf_add_tensor(g, x);
    };
    add_w("token_embd.weight", base, d.n_embd, d.n_vocab);
    for (int il = 0; il < d.n_layer; il++) {
        ggml_type t_v = base, t_down = base;
        if (km) {
            const bool more = use_more_bits(il, d.n_layer);
            t_v = more ? GGML_TYPE_Q6_K : (d.is_70b ? GGML_TYPE_Q5_K : GGML_TYPE_Q4_K); // llama-quant.cpp:291-299
            t_down = more ? GGML_TYPE_Q6_K : GGML_TYPE_Q4_K;                           // llama-quant.cpp:347-353
        }
        const std::string p = "blk." + std::to_string(il) + ".";
        add_norm(p + "attn_norm.weight");
        const bool e8 = km && d.n_expert == 8;
        add_w(p + "attn_q.weight", base, d.n_embd, d.n_embd);
        add_w(p + "attn_k.weight", e8 ? GGML_TYPE_Q8_0 : base, d.n_embd, kv_dim);
        add_w(p + "attn_v.weight", e8 ? GGML_TYPE_Q8_0 : t_v, d.n_embd, kv_dim);
        add_w(p + "attn_output.weight", e8 ? GGML_TYPE_Q5_K : base, d.n_embd, d.n_embd);
        add_norm(p + "ffn_norm.weight");
        if (d.n_expert) {
            ggml_tensor * r = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, d.n_embd, d.n_expert);      // router, never quantized
            ggml_set_name(r, (p + "ffn_gate_inp.weight").c_str());
            for (int64_t i = 0; i < (int64_t)d.n_embd * d.n_expert; i++)
                ((float *)r->data)[i] = ((float)(rnd() >> 40) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
            gguf_add_tensor(g, r);
            add_w3(p + "ffn_gate_exps.weight", base, d.n_embd, d.n_ff, d.n_expert);
            add_w3(p + "ffn_up_exps.weight", base, d.n_embd, d.n_ff, d.n_expert);
            add_w3(p + "ffn_down_exps.weight", t_down, d.n_ff, d.n_embd, d.n_expert);
        } else {
            add_w(p + "ffn_gate.weight", base, d.n_embd, d.n_ff);
            add_w(p + "ffn_up.weight", base, d.n_embd, d.n_ff);
            add_w(p + "ffn_down.weight", t_down, d.n_ff, d.n_embd);
        }
    }
    add_norm("output_norm.weight");
    add_w("output.weight", km ? GGML_TYPE_Q6_K : base, d.n_embd, d.n_vocab);

    if (!gguf_write_to_file(g, out.c_str(), false)) { fprintf(stderr, "write failed\n"); return 1; }
    printf("wrote %s (%s %s, %.2f GB)\n", out.c_str(), model.c_str(), ftype.c_str(), ggml_used_mem(ctx) / 1e9);
    gguf_free(g);
    ggml_free(ctx);
    return 0;
}
