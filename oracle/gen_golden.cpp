// gen_golden.cpp -- writes tests/golden/*.gold by running the REFERENCE's own functions
// (oracle/_ref/libggml-base.so + libggml-cpu.so, compiled from /root/reference by oracle/Makefile)
// through its public headers.  TEST INFRASTRUCTURE ONLY.  Run:  make -C oracle golden
//
// The fixtures are data (inputs + expected outputs); nothing of the reference's source travels.
//
// File format ("GOLD1"): u32 n_entries, then per entry
//   u32 name_len, name bytes, u32 dtype (0=f32 1=u8 2=i32 3=u16), u64 nbytes, payload.
#include "ggml.h"
#include "ggml-cpu.h"
#include "ggml-backend.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

struct Entry { std::string name; uint32_t dtype; std::vector<uint8_t> data; };
struct Gold {
    std::vector<Entry> e;
    void add(const std::string & n, uint32_t dt, const void * p, size_t nb) {
        Entry x; x.name = n; x.dtype = dt; x.data.assign((const uint8_t *)p, (const uint8_t *)p + nb); e.push_back(x);
    }
    void f32(const std::string & n, const std::vector<float> & v) { add(n, 0, v.data(), v.size()*4); }
    void u8 (const std::string & n, const std::vector<uint8_t> & v) { add(n, 1, v.data(), v.size()); }
    void i32(const std::string & n, const std::vector<int32_t> & v) { add(n, 2, v.data(), v.size()*4); }
    void u16(const std::string & n, const std::vector<uint16_t> & v) { add(n, 3, v.data(), v.size()*2); }
    void write(const std::string & path) {
        FILE * f = fopen(path.c_str(), "wb");
        if (!f) { perror(path.c_str()); exit(1); }
        fwrite("GOLD1", 1, 5, f);
        uint32_t n = e.size(); fwrite(&n, 4, 1, f);
        for (auto & x : e) {
            uint32_t l = x.name.size(); fwrite(&l, 4, 1, f); fwrite(x.name.data(), 1, l, f);
            fwrite(&x.dtype, 4, 1, f);
            uint64_t nb = x.data.size(); fwrite(&nb, 8, 1, f); fwrite(x.data.data(), 1, nb, f);
        }
        fclose(f);
        printf("wrote %s (%zu entries)\n", path.c_str(), e.size());
    }
};

static std::vector<float> uniform(size_t n, uint32_t seed, float lo = -1.0f, float hi = 1.0f) {
    std::mt19937 rng(seed);
    std::vector<float> v(n);
    for (auto & x : v) x = lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f);
    return v;
}

// deterministic data of tests/test-quantize-fns.cpp:31-35
static std::vector<float> cosdata(size_t n, float off) {
    std::vector<float> v(n);
    for (size_t i = 0; i < n; i++) v[i] = 0.1f + 2 * cosf(i + off);
    return v;
}

// q8_K has no from_float_ref entry in the base type-traits table (ggml.c:805-810); its reference
// quantizer is exported by ggml-base (ggml-quants.h) and is what ggml-cpu's generic path calls.
extern "C" void quantize_row_q8_K_ref(const float * x, void * y, int64_t k);

static std::vector<uint8_t> quant_ref(ggml_type t, const std::vector<float> & x) {
    std::vector<uint8_t> out(ggml_row_size(t, x.size()), 0);
    if (t == GGML_TYPE_Q8_K) quantize_row_q8_K_ref(x.data(), out.data(), x.size());
    else ggml_get_type_traits(t)->from_float_ref(x.data(), out.data(), x.size());
    return out;
}

static std::vector<float> dequant(ggml_type t, const std::vector<uint8_t> & b, size_t n) {
    std::vector<float> y(n);
    ggml_get_type_traits(t)->to_float(b.data(), y.data(), n);
    return y;
}

static const ggml_type WTYPES[] = {GGML_TYPE_Q4_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K};
// second batch: the remaining 32-value block types; their fixtures go to files of their own so that the
// first batch's files stay byte-identical
static const ggml_type WTYPES2[] = {GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1};
// third batch (the remainder): the 2- and 3-bit K-quants and the non-linear 4-bit types
static const ggml_type WTYPES3[] = {GGML_TYPE_Q2_K, GGML_TYPE_Q3_K, GGML_TYPE_IQ4_NL, GGML_TYPE_IQ4_XS};

// random *bytes* as blocks, with the fp16 scale fields forced to finite, moderate values so that
// every packed-scale bit pattern is covered without inf/nan
static std::vector<uint8_t> random_blocks(ggml_type t, size_t nblocks, uint32_t seed) {
    std::mt19937 rng(seed);
    const size_t ts = ggml_type_size(t);
    std::vector<uint8_t> b(nblocks * ts);
    for (auto & x : b) x = rng() & 0xff;
    auto half = [&](float lo, float hi) {
        float f = lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f);
        if (rng() & 1) f = -f;
        return ggml_fp32_to_fp16(f);
    };
    for (size_t i = 0; i < nblocks; i++) {
        uint8_t * p = b.data() + i * ts;
        ggml_fp16_t h;
        switch (t) {
            case GGML_TYPE_Q4_0: case GGML_TYPE_Q8_0: case GGML_TYPE_Q5_0:
                h = half(1e-3f, 0.1f); memcpy(p, &h, 2); break;
            case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_1:
                h = half(1e-3f, 0.1f); memcpy(p, &h, 2);
                h = half(1e-3f, 0.5f); memcpy(p + 2, &h, 2); break;
            case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K:
                h = half(1e-4f, 0.01f); memcpy(p, &h, 2);
                h = half(1e-4f, 0.01f); memcpy(p + 2, &h, 2); break;
            case GGML_TYPE_Q6_K:
                h = half(1e-5f, 1e-3f); memcpy(p + 208, &h, 2); break;
            case GGML_TYPE_Q2_K:
                h = half(1e-3f, 0.05f); memcpy(p + 80, &h, 2);
                h = half(1e-3f, 0.05f); memcpy(p + 82, &h, 2); break;
            case GGML_TYPE_Q3_K:
                h = half(1e-4f, 0.01f); memcpy(p + 108, &h, 2); break;
            case GGML_TYPE_IQ4_NL:
                h = half(1e-4f, 0.01f); memcpy(p, &h, 2); break;
            case GGML_TYPE_IQ4_XS:
                h = half(1e-5f, 1e-3f); memcpy(p, &h, 2); break;
            default: break;
        }
    }
    return b;
}

// ---- 1. activation quantizers ------------------------------------------------------------------
static void gen_quant(const std::string & dir) {
    Gold g;
    struct In { const char * name; std::vector<float> x; };
    std::vector<In> ins;
    ins.push_back({"cos0", cosdata(4096, 0.0f)});
    ins.push_back({"cos1", cosdata(4096, 1.0f)});
    ins.push_back({"u1234", uniform(2048, 1234)});
    ins.push_back({"wide", uniform(1024, 99, -300.0f, 300.0f)});
    {   // edge blocks: zeros, single +-max, exact ties at k+0.5, denormals, huge (fp16 d overflow), tiny
        std::vector<float> x(256 * 8, 0.0f);
        x[256 + 5] = 1.0f;                      // single +max
        x[512 + 7] = -3.5f;                     // single -max
        for (int j = 0; j < 256; j++) x[768 + j] = (j % 32 == 0) ? 127.0f : 0.5f + (j % 7); // ties: x*id = k+0.5
        for (int j = 0; j < 256; j++) x[1024 + j] = 1e-41f * (j - 128);   // f32 denormals
        for (int j = 0; j < 256; j++) x[1280 + j] = 3.0e4f * (j % 5 - 2) * 127.0f; // d above fp16 max
        for (int j = 0; j < 256; j++) x[1536 + j] = 6e-8f * ((j * 37) % 11 - 5);   // d in fp16 subnormal range
        for (int j = 0; j < 256; j++) x[1792 + j] = (j & 1) ? -0.0f : 0.0f;
        ins.push_back({"edge", x});
    }
    for (auto & in : ins) {
        g.f32(std::string("x.") + in.name, in.x);
        g.u8(std::string("q8_0.") + in.name, quant_ref(GGML_TYPE_Q8_0, in.x));
        g.u8(std::string("q8_1.") + in.name, quant_ref(GGML_TYPE_Q8_1, in.x));
        g.u8(std::string("q8_K.") + in.name, quant_ref(GGML_TYPE_Q8_K, in.x));
    }
    g.write(dir + "/quantize.gold");
}

// ---- 2. block dequantizers ---------------------------------------------------------------------
// the KV-cache store formats beyond q8_0: q4_0 blocks (quantize_row_q4_0_ref) and bf16 rows (ggml_fp32_to_bf16_row_ref) of the inputs
// of quantize.gold, in a file of their own so that the first fixture keeps its bytes
static void gen_quant2(const std::string & dir) {
    Gold g;
    struct In { const char * name; std::vector<float> x; };
    std::vector<In> ins;
    ins.push_back({"cos0", cosdata(4096, 0.0f)});
    ins.push_back({"u1234", uniform(2048, 1234)});
    ins.push_back({"wide", uniform(1024, 99, -300.0f, 300.0f)});
    {   // edge blocks: zeros, single +-max, a maximum attained with both signs, ties at k+0.5, denormals, huge, tiny, signed zeros
        std::vector<float> x(256 * 8, 0.0f);
        x[256 + 5] = 1.0f;
        x[512 + 7] = -3.5f;
        for (int j = 0; j < 256; j++) x[768 + j] = (j % 32 == 3) ? 8.0f : (j % 32 == 9) ? -8.0f : 0.5f * (j % 16) - 3.75f;
        for (int j = 0; j < 256; j++) x[1024 + j] = 1e-41f * (j - 128);
        for (int j = 0; j < 256; j++) x[1280 + j] = 3.0e4f * (j % 5 - 2) * 127.0f;
        for (int j = 0; j < 256; j++) x[1536 + j] = 6e-8f * ((j * 37) % 11 - 5);
        for (int j = 0; j < 256; j++) x[1792 + j] = (j & 1) ? -0.0f : 0.0f;
        ins.push_back({"edge", x});
    }
    for (auto & in : ins) {
        g.f32(std::string("x.") + in.name, in.x);
        g.u8(std::string("q4_0.") + in.name, quant_ref(GGML_TYPE_Q4_0, in.x));
        std::vector<uint16_t> b(in.x.size());
        ggml_fp32_to_bf16_row_ref(in.x.data(), (ggml_bf16_t *)b.data(), (int64_t)in.x.size());
        g.u16(std::string("bf16.") + in.name, b);
    }
    g.write(dir + "/quantize2.gold");
}

static void gen_dequant(const std::string & dir) {
    Gold g;
    const std::vector<float> src = uniform(2048, 4321);
    for (ggml_type t : WTYPES) {
        const std::string tn = ggml_type_name(t);
        const size_t n = 2048;
        const size_t nblocks = n / ggml_blck_size(t);
        auto rb = random_blocks(t, nblocks, 7 + (uint32_t)t);
        g.u8("rand.blocks." + tn, rb);
        g.f32("rand.deq." + tn, dequant(t, rb, n));
        // blocks produced by the reference's own weight quantizer (ggml_quantize_chunk, ggml.c)
        std::vector<uint8_t> qb(ggml_row_size(t, n));
        ggml_quantize_chunk(t, src.data(), qb.data(), 0, 1, n, nullptr);
        g.u8("quant.blocks." + tn, qb);
        g.f32("quant.deq." + tn, dequant(t, qb, n));
    }
    g.f32("quant.src", src);
    g.write(dir + "/dequantize.gold");
}

static void gen_dequant2(const std::string & dir) {
    Gold g;
    const std::vector<float> src = uniform(2048, 4321);
    for (ggml_type t : WTYPES2) {
        const std::string tn = ggml_type_name(t);
        const size_t n = 2048;
        auto rb = random_blocks(t, n / ggml_blck_size(t), 7 + (uint32_t)t);
        g.u8("rand.blocks." + tn, rb);
        g.f32("rand.deq." + tn, dequant(t, rb, n));
        std::vector<uint8_t> qb(ggml_row_size(t, n));
        ggml_quantize_chunk(t, src.data(), qb.data(), 0, 1, n, nullptr);
        g.u8("quant.blocks." + tn, qb);
        g.f32("quant.deq." + tn, dequant(t, qb, n));
    }
    g.f32("quant.src", src);
    g.write(dir + "/dequantize2.gold");
}

static void gen_dequant3(const std::string & dir) {
    Gold g;
    const std::vector<float> src = uniform(2048, 4321);
    for (ggml_type t : WTYPES3) {
        const std::string tn = ggml_type_name(t);
        const size_t n = 2048;
        auto rb = random_blocks(t, n / ggml_blck_size(t), 7 + (uint32_t)t);
        g.u8("rand.blocks." + tn, rb);
        g.f32("rand.deq." + tn, dequant(t, rb, n));
        std::vector<uint8_t> qb(ggml_row_size(t, n));
        ggml_quantize_chunk(t, src.data(), qb.data(), 0, 1, n, nullptr);
        g.u8("quant.blocks." + tn, qb);
        g.f32("quant.deq." + tn, dequant(t, qb, n));
    }
    g.f32("quant.src", src);
    g.write(dir + "/dequantize3.gold");
}

// ---- 3. ops on the ggml-cpu backend ------------------------------------------------------------
struct Ctx {
    ggml_context * c;
    Ctx(size_t mb = 256) { ggml_init_params p = {mb * 1024 * 1024, nullptr, false}; c = ggml_init(p); }
    ~Ctx() { ggml_free(c); }
};

static void compute(ggml_context * c, ggml_tensor * out, int nth = 4) {
    ggml_cgraph * gf = ggml_new_graph(c);
    ggml_build_forward_expand(gf, out);
    if (ggml_graph_compute_with_ctx(c, gf, nth) != GGML_STATUS_SUCCESS) { fprintf(stderr, "compute failed\n"); exit(1); }
}

static std::vector<uint8_t> quantize_w(ggml_type t, const std::vector<float> & w, int64_t nrows, int64_t k) {
    std::vector<uint8_t> q(ggml_row_size(t, k) * nrows);
    ggml_quantize_chunk(t, w.data(), q.data(), 0, nrows, k, nullptr);
    return q;
}

static void gen_mul_mat(const std::string & dir) {
    Gold g;
    struct Shape { int64_t m, k, n; };
    const Shape shapes[] = {{16, 256, 1}, {33, 512, 3}, {64, 1024, 8}, {24, 256, 17}};
    int si = 0;
    for (ggml_type t : WTYPES) {
        for (const Shape & s : shapes) {
            Ctx cx;
            const std::string tag = std::string(ggml_type_name(t)) + "." + std::to_string(s.m) + "x" + std::to_string(s.k) + "x" + std::to_string(s.n);
            auto wf = uniform(s.m * s.k, 100 + si), xf = uniform(s.k * s.n, 200 + si);
            si++;
            auto wq = quantize_w(t, wf, s.m, s.k);
            ggml_tensor * a = ggml_new_tensor_2d(cx.c, t, s.k, s.m);
            ggml_tensor * b = ggml_new_tensor_2d(cx.c, GGML_TYPE_F32, s.k, s.n);
            memcpy(a->data, wq.data(), wq.size());
            memcpy(b->data, xf.data(), xf.size() * 4);
            ggml_tensor * d = ggml_mul_mat(cx.c, a, b);
            compute(cx.c, d);
            std::vector<float> out((float *)d->data, (float *)d->data + s.m * s.n);
            g.i32("shape." + tag, {(int32_t)t, (int32_t)s.m, (int32_t)s.k, (int32_t)s.n});
            g.u8("w." + tag, wq);
            g.f32("x." + tag, xf);
            g.f32("dst." + tag, out);
        }
    }
    g.write(dir + "/mul_mat.gold");
}

// MUL_MAT on the second batch of quantized types and on float weights (f16 / bf16 / f32, what an unquantized GGUF or a
// graph without flash attention multiplies by)
static void gen_mul_mat2(const std::string & dir) {
    Gold g;
    struct Shape { int64_t m, k, n; };
    const Shape shapes[] = {{16, 256, 1}, {33, 512, 3}, {64, 1024, 8}, {24, 256, 17}, {40, 96, 2}};
    const ggml_type types[] = {GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1, GGML_TYPE_F16, GGML_TYPE_BF16, GGML_TYPE_F32};
    int si = 0;
    for (ggml_type t : types) {
        for (const Shape & s : shapes) {
            if (s.k % ggml_blck_size(t) != 0) continue;
            Ctx cx;
            const std::string tag = std::string(ggml_type_name(t)) + "." + std::to_string(s.m) + "x" + std::to_string(s.k) + "x" + std::to_string(s.n);
            auto wf = uniform(s.m * s.k, 300 + si), xf = uniform(s.k * s.n, 400 + si);
            si++;
            std::vector<uint8_t> wq;
            if (t == GGML_TYPE_F32) { wq.resize(wf.size() * 4); memcpy(wq.data(), wf.data(), wq.size()); }
            else if (t == GGML_TYPE_F16) { wq.resize(wf.size() * 2); ggml_fp32_to_fp16_row(wf.data(), (ggml_fp16_t *)wq.data(), wf.size()); }
            else if (t == GGML_TYPE_BF16) { wq.resize(wf.size() * 2); ggml_fp32_to_bf16_row_ref(wf.data(), (ggml_bf16_t *)wq.data(), wf.size()); }
            else wq = quantize_w(t, wf, s.m, s.k);
            ggml_tensor * a = ggml_new_tensor_2d(cx.c, t, s.k, s.m);
            ggml_tensor * b = ggml_new_tensor_2d(cx.c, GGML_TYPE_F32, s.k, s.n);
            memcpy(a->data, wq.data(), wq.size());
            memcpy(b->data, xf.data(), xf.size() * 4);
            ggml_tensor * d = ggml_mul_mat(cx.c, a, b);
            compute(cx.c, d);
            std::vector<float> out((float *)d->data, (float *)d->data + s.m * s.n);
            g.i32("shape." + tag, {(int32_t)t, (int32_t)s.m, (int32_t)s.k, (int32_t)s.n});
            g.u8("w." + tag, wq);
            g.f32("x." + tag, xf);
            g.f32("dst." + tag, out);
        }
    }
    g.write(dir + "/mul_mat2.gold");
}

static void gen_mul_mat3(const std::string & dir) {
    Gold g;
    struct Shape { int64_t m, k, n; };
    const Shape shapes[] = {{16, 256, 1}, {33, 512, 3}, {64, 1024, 8}, {24, 256, 17}, {40, 96, 2}};
    int si = 0;
    for (ggml_type t : WTYPES3) {
        for (const Shape & s : shapes) {
            if (s.k % ggml_blck_size(t) != 0) continue;
            Ctx cx;
            const std::string tag = std::string(ggml_type_name(t)) + "." + std::to_string(s.m) + "x" + std::to_string(s.k) + "x" + std::to_string(s.n);
            auto wf = uniform(s.m * s.k, 500 + si), xf = uniform(s.k * s.n, 600 + si);
            si++;
            auto wq = quantize_w(t, wf, s.m, s.k);
            ggml_tensor * a = ggml_new_tensor_2d(cx.c, t, s.k, s.m);
            ggml_tensor * b = ggml_new_tensor_2d(cx.c, GGML_TYPE_F32, s.k, s.n);
            memcpy(a->data, wq.data(), wq.size());
            memcpy(b->data, xf.data(), xf.size() * 4);
            ggml_tensor * d = ggml_mul_mat(cx.c, a, b);
            compute(cx.c, d);
            std::vector<float> out((float *)d->data, (float *)d->data + s.m * s.n);
            g.i32("shape." + tag, {(int32_t)t, (int32_t)s.m, (int32_t)s.k, (int32_t)s.n});
            g.u8("w." + tag, wq);
            g.f32("x." + tag, xf);
            g.f32("dst." + tag, out);
        }
    }
    g.write(dir + "/mul_mat3.gold");
}

static void gen_mul_mat_id(const std::string & dir) {
    Gold g;
    struct Shape { int64_t m, k, n_exp, n_used, n_tok; bool bcast; };
    const Shape shapes[] = {{32, 256, 4, 2, 1, true}, {48, 512, 8, 2, 5, true}, {32, 256, 8, 4, 3, false}};
    const ggml_type types[] = {GGML_TYPE_Q4_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_K, GGML_TYPE_Q6_K};
    int si = 0;
    for (ggml_type t : types) {
        for (const Shape & s : shapes) {
            Ctx cx;
            const std::string tag = std::string(ggml_type_name(t)) + "." + std::to_string(si);
            const int64_t b1 = s.bcast ? 1 : s.n_used;
            auto wf = uniform(s.m * s.k * s.n_exp, 300 + si), xf = uniform(s.k * b1 * s.n_tok, 400 + si);
            auto wq = quantize_w(t, wf, s.m * s.n_exp, s.k);
            std::mt19937 rng(500 + si);
            std::vector<int32_t> ids(s.n_used * s.n_tok);
            for (int64_t tk = 0; tk < s.n_tok; tk++) {
                std::vector<int32_t> perm(s.n_exp);
                for (int i = 0; i < s.n_exp; i++) perm[i] = i;
                for (int i = s.n_exp - 1; i > 0; i--) std::swap(perm[i], perm[rng() % (i + 1)]);
                for (int e = 0; e < s.n_used; e++) ids[tk * s.n_used + e] = perm[e];
            }
            si++;
            ggml_tensor * as = ggml_new_tensor_3d(cx.c, t, s.k, s.m, s.n_exp);
            ggml_tensor * b  = ggml_new_tensor_3d(cx.c, GGML_TYPE_F32, s.k, b1, s.n_tok);
            ggml_tensor * id = ggml_new_tensor_2d(cx.c, GGML_TYPE_I32, s.n_used, s.n_tok);
            memcpy(as->data, wq.data(), wq.size());
            memcpy(b->data, xf.data(), xf.size() * 4);
            memcpy(id->data, ids.data(), ids.size() * 4);
            ggml_tensor * d = ggml_mul_mat_id(cx.c, as, b, id);
            compute(cx.c, d);
            std::vector<float> out((float *)d->data, (float *)d->data + s.m * s.n_used * s.n_tok);
            g.i32("shape." + tag, {(int32_t)t, (int32_t)s.m, (int32_t)s.k, (int32_t)s.n_exp, (int32_t)s.n_used, (int32_t)s.n_tok, (int32_t)b1});
            g.u8("w." + tag, wq);
            g.f32("x." + tag, xf);
            g.i32("ids." + tag, ids);
            g.f32("dst." + tag, out);
        }
    }
    g.write(dir + "/mul_mat_id.gold");
}

static void gen_flash_attn(const std::string & dir) {
    Gold g;
    struct Case { int64_t d, nh, nhkv, nq, nkv; ggml_type tkv; bool mask; float max_bias, softcap; };
    const Case cases[] = {
        {128, 8, 2, 1, 256, GGML_TYPE_F16,  true,  0.0f, 0.0f},
        {128, 8, 2, 1, 256, GGML_TYPE_Q8_0, true,  0.0f, 0.0f},
        {128, 4, 4, 3, 128, GGML_TYPE_Q4_0, true,  0.0f, 0.0f},
        { 64, 4, 1, 5, 128, GGML_TYPE_F16,  true,  8.0f, 0.0f},
        { 64, 4, 2, 2,  96, GGML_TYPE_Q8_0, false, 0.0f, 10.0f},
        {128, 8, 8, 35, 192, GGML_TYPE_F16, true,  0.0f, 0.0f},
        {256, 2, 1, 1,  64, GGML_TYPE_Q8_0, true,  0.0f, 0.0f},
    };
    int ci = 0;
    for (const Case & cs : cases) {
        Ctx cx;
        const std::string tag = std::to_string(ci);
        auto qf = uniform(cs.d * cs.nq * cs.nh, 600 + ci);
        auto kf = uniform(cs.d * cs.nkv * cs.nhkv, 700 + ci);
        auto vf = uniform(cs.d * cs.nkv * cs.nhkv, 800 + ci);
        const int64_t nqp = GGML_PAD(cs.nq, GGML_KQ_MASK_PAD);
        std::vector<uint16_t> mk(cs.nkv * nqp);
        {
            std::mt19937 rng(900 + ci);
            for (int64_t i = 0; i < nqp; i++)
                for (int64_t j = 0; j < cs.nkv; j++) {
                    // causal-like: a random visible prefix per row, plus a few finite biases
                    const bool vis = j <= (cs.nkv - cs.nq + i) || i >= cs.nq;
                    float mval = vis ? ((rng() % 5 == 0) ? -0.5f * (float)(rng() % 4) : 0.0f) : -INFINITY;
                    mk[i * cs.nkv + j] = ggml_fp32_to_fp16(mval);
                }
        }
        ci++;
        ggml_tensor * q = ggml_new_tensor_4d(cx.c, GGML_TYPE_F32, cs.d, cs.nq, cs.nh, 1);
        ggml_tensor * k = ggml_new_tensor_4d(cx.c, cs.tkv, cs.d, cs.nkv, cs.nhkv, 1);
        ggml_tensor * v = ggml_new_tensor_4d(cx.c, cs.tkv, cs.d, cs.nkv, cs.nhkv, 1);
        ggml_tensor * m = cs.mask ? ggml_new_tensor_4d(cx.c, GGML_TYPE_F16, cs.nkv, nqp, 1, 1) : nullptr;
        memcpy(q->data, qf.data(), qf.size() * 4);
        std::vector<uint8_t> kq, vq;
        if (cs.tkv == GGML_TYPE_F16) {
            kq.resize(kf.size() * 2); vq.resize(vf.size() * 2);
            ggml_fp32_to_fp16_row(kf.data(), (ggml_fp16_t *)kq.data(), kf.size());
            ggml_fp32_to_fp16_row(vf.data(), (ggml_fp16_t *)vq.data(), vf.size());
        } else {
            kq = quantize_w(cs.tkv, kf, cs.nkv * cs.nhkv, cs.d);
            vq = quantize_w(cs.tkv, vf, cs.nkv * cs.nhkv, cs.d);
        }
        memcpy(k->data, kq.data(), kq.size());
        memcpy(v->data, vq.data(), vq.size());
        if (m) memcpy(m->data, mk.data(), mk.size() * 2);
        const float scale = 1.0f / sqrtf((float)cs.d);
        ggml_tensor * o = ggml_flash_attn_ext(cx.c, q, k, v, m, scale, cs.max_bias, cs.softcap);
        ggml_flash_attn_ext_set_prec(o, GGML_PREC_F32);
        compute(cx.c, o);
        std::vector<float> out((float *)o->data, (float *)o->data + cs.d * cs.nh * cs.nq);
        std::vector<float> prm = {scale, cs.max_bias, cs.softcap};
        g.i32("shape." + tag, {(int32_t)cs.d, (int32_t)cs.nh, (int32_t)cs.nhkv, (int32_t)cs.nq, (int32_t)cs.nkv, (int32_t)cs.tkv, (int32_t)cs.mask, (int32_t)nqp});
        g.f32("params." + tag, prm);
        g.f32("q." + tag, qf);
        g.u8("k." + tag, kq);
        g.u8("v." + tag, vq);
        if (m) g.u16("mask." + tag, mk);
        g.f32("dst." + tag, out);
    }
    g.write(dir + "/flash_attn_ext.gold");
}

int main(int argc, char ** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    ggml_cpu_init();
    gen_quant(dir);
    gen_dequant(dir);
    gen_mul_mat(dir);
    gen_mul_mat_id(dir);
    gen_flash_attn(dir);
    gen_dequant2(dir);
    gen_mul_mat2(dir);
    gen_dequant3(dir);
    gen_mul_mat3(dir);
    gen_quant2(dir);
    return 0;
}
