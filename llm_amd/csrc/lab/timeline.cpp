// timeline.cpp -- where does a decode layer's time go?  (NOT product code; tuning evidence only.)
// Builds the production mat-vec kernels with -DMI355_STAMPS (four s_memrealtime stamps per wave: entry, activations
// staged, weight loop done, results stored), replays a few Llama-3-8B decode layers' launches from a hipGraph and
// prints, per launch: the gap since the previous launch's last wave ended, how the waves' entries spread (ramp), when
// their loops end (tail) and the launch's span.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DMI355_STAMPS -I include -I llm_amd/csrc/kernels \
//         llm_amd/csrc/lab/timeline.cpp llm_amd/csrc/kernels/{mmvq,quantize,api,timing}.hip -o llm_amd/lib/timeline
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi355_kernels.h"

unsigned long long * mi355_stamp_host_ptr = nullptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define RC(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, mi355_error_string(r_)); exit(1); } } while (0)

struct Launch { std::string name; int waves; double bytes; unsigned long long * stamps; };

int main(int argc, char ** argv) {
    const int n_layers = argc > 1 ? atoi(argv[1]) : 6;
    const bool f32_wo = argc > 2 ? atoi(argv[2]) != 0 : true;     // wo quantizes its f32 input in its prologue (no quantize launch)
    const int64_t n_embd = 4096, n_ff = 14336, kv = 1024;
    const size_t stamp_slot = 1 << 20;   // bytes per launch
    hipStream_t st; CK(hipStreamCreate(&st));

    auto rnd_blocks = [&](size_t bytes) {
        uint8_t * d; CK(hipMalloc(&d, bytes));
        std::vector<uint32_t> h(bytes / 4 + 1);
        static uint32_t s = 12345;
        for (auto & x : h) { s = s * 1664525u + 1013904223u; x = (s >> 3) & 0x3bff3bffu; }
        CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
        return d;
    };
    auto rnd_f32 = [&](size_t n) {
        float * d; CK(hipMalloc(&d, n * 4));
        std::vector<float> h(n);
        for (size_t i = 0; i < n; i++) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
        CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        return d;
    };
    const int64_t rs_e = mi355_row_size(MI355_TYPE_Q4_K, n_embd), rs_f = mi355_row_size(MI355_TYPE_Q4_K, n_ff);
    const int64_t rs_f6 = mi355_row_size(MI355_TYPE_Q6_K, n_ff);
    float * x = rnd_f32(n_embd), * nw = rnd_f32(n_embd), * normed, * q_out, * attn = rnd_f32(n_embd), * wo_out, * resid = rnd_f32(n_embd), * resid_out;
    float * gate_out, * up_out, * hidden, * down_out;
    CK(hipMalloc(&normed, n_embd * 4)); CK(hipMalloc(&q_out, n_embd * 4)); CK(hipMalloc(&wo_out, n_embd * 4)); CK(hipMalloc(&resid_out, n_embd * 4));
    CK(hipMalloc(&gate_out, n_ff * 4)); CK(hipMalloc(&up_out, n_ff * 4)); CK(hipMalloc(&hidden, n_ff * 4)); CK(hipMalloc(&down_out, n_embd * 4));
    void * k_row, * v_row; CK(hipMalloc(&k_row, kv * 2)); CK(hipMalloc(&v_row, kv * 2));
    int32_t * pos; CK(hipMalloc(&pos, 4)); { int32_t p = 127; CK(hipMemcpy(pos, &p, 4, hipMemcpyHostToDevice)); }
    int8_t * aq_e, * aq_f; float * ads_e, * ads_f;
    CK(hipMalloc(&aq_e, n_embd)); CK(hipMalloc(&ads_e, n_embd / 32 * 8)); CK(hipMalloc(&aq_f, n_ff)); CK(hipMalloc(&ads_f, n_ff / 32 * 8));
    unsigned long long * stamps; CK(hipMalloc(&stamps, stamp_slot * 5 * n_layers)); CK(hipMemset(stamps, 0, stamp_slot * 5 * n_layers));

    std::vector<Launch> launches;
    auto next_stamps = [&](const char * name, int waves, double bytes) {
        unsigned long long * p = (unsigned long long *)((uint8_t *)stamps + stamp_slot * launches.size());
        if ((size_t)waves * 64 > stamp_slot) { fprintf(stderr, "stamp slot too small\n"); exit(1); }
        launches.push_back({name, waves, bytes, p});
        mi355_stamp_host_ptr = p;
    };

    struct LayerW { uint8_t * wq, * wk, * wv, * wo, * wg, * wu, * wd; int down_type; };
    std::vector<LayerW> L(n_layers);
    for (int l = 0; l < n_layers; l++) {
        L[l].wq = rnd_blocks(rs_e * n_embd); L[l].wk = rnd_blocks(rs_e * kv); L[l].wv = rnd_blocks(rs_e * kv); L[l].wo = rnd_blocks(rs_e * n_embd);
        L[l].wg = rnd_blocks(rs_e * n_ff); L[l].wu = rnd_blocks(rs_e * n_ff);
        L[l].down_type = (l & 1) ? MI355_TYPE_Q6_K : MI355_TYPE_Q4_K;
        L[l].wd = rnd_blocks(((l & 1) ? rs_f6 : rs_f) * n_embd);
    }

    auto record = [&]() {
        launches.clear();
        for (int l = 0; l < n_layers; l++) {
            {
                mi355_qkv_rope_args a; memset(&a, 0, sizeof(a));
                a.type_qk = a.type_v = MI355_TYPE_Q4_K; a.wq = L[l].wq; a.wk = L[l].wk; a.wv = L[l].wv;
                a.wq_row_stride = a.wk_row_stride = a.wv_row_stride = rs_e; a.m_q = n_embd; a.m_k = a.m_v = kv; a.k = n_embd;
                a.q_out = q_out; a.k_out = k_row; a.v_out = v_row; a.k_out_type = a.v_out_type = MI355_TYPE_F16; a.pos = pos;
                a.head_dim = a.n_dims = 128; a.mode = 0; a.n_ctx_orig = 8192; a.freq_base = 500000.0f; a.freq_scale = 1.0f; a.attn_factor = 1.0f;
                a.beta_fast = 32.0f; a.beta_slow = 1.0f; a.norm_x = x; a.norm_w = nw; a.norm_out = normed; a.norm_eps = 1e-5f;
                next_stamps("qkv_rope(norm)", (int)((n_embd + 2 * kv) / 8 * 4), (double)rs_e * (n_embd + 2 * kv));
                RC(mi355_qkv_rope_matvec(&a, st));
            }
            if (!f32_wo) RC(mi355_quantize_act(attn, aq_e, ads_e, n_embd, 1, 1, 1, n_embd * 4, 0, 0, st));
            {
                mi355_mmvq_args a; memset(&a, 0, sizeof(a));
                a.type = MI355_TYPE_Q4_K; a.W = L[l].wo; a.w_row_stride = rs_e; a.act_q = aq_e; a.act_ds = ads_e; a.dst = wo_out; a.dst_col_stride = n_embd;
                if (f32_wo) { a.act_q = nullptr; a.act_ds = nullptr; a.src_f32 = attn; }
                a.m = n_embd; a.k = n_embd; a.ncols = 1; a.nchannels = 1; a.ch_ratio = 1; a.residual = resid; a.sum = resid_out;
                next_stamps("wo+res", (int)n_embd, (double)rs_e * n_embd);
                RC(mi355_mul_mat_vec_q_group(&a, 1, st));
            }
            {
                mi355_glu_matvec_args a; memset(&a, 0, sizeof(a));
                a.type = MI355_TYPE_Q4_K; a.glu_op = 2; a.w_gate = L[l].wg; a.w_up = L[l].wu; a.gate_row_stride = a.up_row_stride = rs_e;
                a.m = n_ff; a.k = n_embd; a.gate_out = gate_out; a.up_out = up_out; a.dst = hidden; a.out_q = aq_f; a.out_ds = ads_f;
                a.norm_x = x; a.norm_w = nw; a.norm_out = normed; a.norm_eps = 1e-5f;
                next_stamps("glu(norm)", (int)(n_ff / 32 * 8), 2.0 * rs_e * n_ff);
                RC(mi355_glu_matvec(&a, st));
            }
            {
                mi355_mmvq_args a; memset(&a, 0, sizeof(a));
                a.type = L[l].down_type; a.W = L[l].wd; a.w_row_stride = (l & 1) ? rs_f6 : rs_f; a.act_q = aq_f; a.act_ds = ads_f; a.dst = down_out;
                a.dst_col_stride = n_embd; a.m = n_embd; a.k = n_ff; a.ncols = 1; a.nchannels = 1; a.ch_ratio = 1; a.residual = resid; a.sum = resid_out;
                next_stamps((l & 1) ? "down q6_K+res" : "down q4_K+res", (int)n_embd, (double)a.w_row_stride * n_embd);
                RC(mi355_mul_mat_vec_q_group(&a, 1, st));
            }
        }
    };

    record();                       // eager warm-up (loads the code objects)
    CK(hipStreamSynchronize(st));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    record();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double total_bytes = 0; for (auto & l : launches) total_bytes += l.bytes;
    printf("graph of %d layers: %.1f us per replay = %.2f us per layer, %.0f GB/s\n", n_layers, ms * 1e3 / reps, ms * 1e3 / reps / n_layers,
           total_bytes / (ms * 1e-3 / reps) * 1e-9);

    std::vector<unsigned long long> h(stamp_slot * launches.size() / 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    printf("times in us (100 MHz stamps). gap = first entry - previous launch's last end; others relative to the launch's first entry\n");
    printf("%-16s %6s %6s | %6s %6s %6s | %6s %6s | %6s %6s %6s %6s | %6s  %6s\n", "launch", "waves", "gap", "ent50", "ent90", "entmx", "act50", "actmx",
           "end10", "end50", "end90", "endmx", "span", "GB/s  lp50(act->loop done) st50(loop done->end)");
    unsigned long long prev_end = 0;
    for (size_t li = 0; li < launches.size(); li++) {
        const Launch & l = launches[li];
        const unsigned long long * p = h.data() + stamp_slot / 8 * li;
        std::vector<double> ent, act, lp, end, st3, fet, s5, s6, s4;
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int w = 0; w < l.waves; w++) { if (p[w * 8] == 0) continue; t0 = std::min(t0, p[w * 8]); t3 = std::max(t3, p[w * 8 + 3]); }
        for (int w = 0; w < l.waves; w++) {
            if (p[w * 8] == 0) continue;
            ent.push_back((p[w * 8] - t0) * 0.01); act.push_back((p[w * 8 + 1] - p[w * 8]) * 0.01);
            lp.push_back(((double)p[w * 8 + 2] - (double)p[w * 8 + 1]) * 0.01); end.push_back((p[w * 8 + 3] - t0) * 0.01);
            if (p[w * 8 + 5]) { s5.push_back(((double)p[w * 8 + 5] - (double)p[w * 8]) * 0.01); s6.push_back(((double)p[w * 8 + 6] - (double)p[w * 8]) * 0.01);
                                s4.push_back(((double)p[w * 8 + 4] - (double)p[w * 8]) * 0.01); }
            fet.push_back(((double)p[w * 8 + 4] - (double)p[w * 8]) * 0.01);
            st3.push_back(((double)p[w * 8 + 3] - (double)p[w * 8 + 2]) * 0.01);
        }
        if (ent.empty()) { printf("%-16s no stamps\n", l.name.c_str()); continue; }
        auto q = [](std::vector<double> & v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
        const double span = (t3 - t0) * 0.01;
        printf("%-16s %6zu %6.2f | %6.2f %6.2f %6.2f | %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f | %6.2f  %6.0f  %6.2f %6.2f  fetch50 %5.2f fetch90 %5.2f | staging waves: x arrived %5.2f reduced %5.2f staged %5.2f\n", l.name.c_str(), ent.size(),
               prev_end ? ((double)t0 - (double)prev_end) * 0.01 : 0.0, q(ent, 0.5), q(ent, 0.9), q(ent, 1.0), q(act, 0.5), q(act, 1.0), q(end, 0.1),
               q(end, 0.5), q(end, 0.9), q(end, 1.0), span, l.bytes / span * 1e-3, q(lp, 0.5), q(st3, 0.5), q(fet, 0.5), q(fet, 0.9), s5.empty() ? 0.0 : q(s5, 0.5), s6.empty() ? 0.0 : q(s6, 0.5), s4.empty() ? 0.0 : q(s4, 0.5));
        prev_end = t3;
        if (li >= 10 && li % 5 == 4 && li > 14) {}   // all layers printed; the first is colder than the rest
    }
    return 0;
}
