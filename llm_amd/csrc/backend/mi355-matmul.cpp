// mi355-matmul.cpp -- MUL_MAT / MUL_MAT_ID / FLASH_ATTN_EXT of the ggml-mi355 plugin: which kernel a product takes (a function of type, N and K
// only, so that a row-split matrix takes the unsplit one's path), the activation formats, the row-split execution, and the per-op dispatch.
#include "mi355-backend.h"

// ------------------------------------------------------------------------------------------------
// ops
// ------------------------------------------------------------------------------------------------
bool is_weight_type(enum ggml_type t) {
    return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K;
}

// MUL_MAT also takes the rest of the 32-value block types, the 2- / 3-bit K-quants and the non-linear 4-bit types (mat-vec + bf16
// GEMM, no fused / skinny kernels) and float weights
bool is_block_weight_type(enum ggml_type t) {
    return is_weight_type(t) || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 ||
           t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_IQ4_XS;
}
bool is_float_weight_type(enum ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16 || t == GGML_TYPE_BF16; }

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// MUL_MAT on block-quantized src0 (reference dispatch: ggml_cuda_mul_mat, ggml-cuda.cu:1988-2063;
// contract ggml.c:2971-2986; oracle ggml_compute_forward_mul_mat, ggml-cpu.c:1196-1388)
//
// Kernel per batch size N (columns of src1), from tools/bench_skinny.py on MI355X (weights 4096..14336 squared-ish):
//   N <= 3 (q4_K, q5_K) / 4 (q6_K) / 8 (others)   mat-vec (mmvq.hip), q8_1 activations, HBM-bound at N = 1
//   up to 32 (K-quants; GGML_MI355_SKINNY_MAX) skinny int8-MFMA kernels (mmsk.hip), one pass of 32 columns: the K-quants on
//                                         q8_K activations with the integer sub-block scales kept on the matrix core,
//                                         the other types on q8_1 with a per-block f32 rescale (up to 32 / 64)
//   beyond                                dequantize-in-LDS bf16 MFMA GEMM (mmdq.hip), bf16 activations
// GGML_MI355_MMQ_INT8=1 keeps prefill on the int8 MFMA tile kernel (mmq.hip), GGML_MI355_NO_SKINNY=1 removes the
// middle row.  The choice depends on (type, N, K) only, never on the row count, so a row-split matrix takes the same
// path (and produces the same bits) as the unsplit one.
// Float weights (f16 / bf16 / f32: an unquantized GGUF, or K / V of a graph without flash attention) take MM_FLOAT -- the
// strided, broadcasting mat-vec of mmf.hip straight on src1, one pass over the weights per 8 columns -- up to 16 columns
// and whenever the GEMM's preconditions fail (K % 64, alignment, f16 src1); beyond, the bf16 GEMM per channel.
// batches of more rows than this are "prefill-sized": the K-quant MUL_MATs that read them take the bf16 GEMM, so fused producers (norm,
// GLU, attention) store the bf16 rows alongside their f32 result.  A consumer that wants another format converts for itself, as always.

// Activations of the graph in flight whose MUL_MAT readers ALL have K-quant weights (a Llama layer's attn_norm: wq / wk / wv; not
// Mixtral's, whose wk / wv are q8_0): one q8_K quantization serves every reader, so the 4096 x 4096 class takes the skinny kernels
// up to 64 columns as well (8B Q4_K_M pp33 / 48 / 64 4910 / 6955 / 8783 -> 5117 / 7207 / 9032 t/s; with q8_0 siblings the GEMM stays
// ahead, 8x7B pp64 4317 against 4160).  Built by mi355_note_graph_activations before a graph's launches are issued -- the same for
// the fused and the unfused run, a captured and an eager one -- and asked by tensor identity.  GGML_MI355_NO_KQ_ONLY=1: off.
static thread_local std::vector<std::pair<const ggml_tensor *, bool>> g_act_readers;
void mi355_note_graph_activations(ggml_cgraph * cgraph) {
    g_act_readers.clear();
    if (MI355_ENV_SET("GGML_MI355_NO_KQ_ONLY")) return;
    const int n = ggml_graph_n_nodes(cgraph);
    for (int i = 0; i < n; i++) {
        const ggml_tensor * t = ggml_graph_node(cgraph, i);
        if (t->op != GGML_OP_MUL_MAT || is_float_weight_type(t->src[0]->type)) continue;
        const enum ggml_type wt = t->src[0]->type;
        const bool kq = wt == GGML_TYPE_Q4_K || wt == GGML_TYPE_Q5_K || wt == GGML_TYPE_Q6_K;
        bool found = false;
        for (auto & e : g_act_readers) if (e.first == t->src[1]) { e.second = e.second && kq; found = true; break; }
        if (!found) g_act_readers.emplace_back(t->src[1], kq);
    }
}
static bool act_read_by_kquants_only(const ggml_tensor * src1) {
    for (const auto & e : g_act_readers) if (e.first == src1) return e.second;
    return false;
}

// dst: the MUL_MAT node when the caller has it (its op_params carry the precision request)
int mi355_mul_mat_path(const ggml_tensor * src0, const ggml_tensor * src1, const ggml_tensor * dst) {
    static const bool int8_mmq = MI355_ENV_SET("GGML_MI355_MMQ_INT8");
    static const bool no_skinny = MI355_ENV_SET("GGML_MI355_NO_SKINNY");
    const int64_t K = src0->ne[0], N = src1->ne[1];
    const bool aligned = (((uintptr_t)src1->data | src1->nb[1] | src1->nb[2] | src1->nb[3]) & 15) == 0;
    if (is_float_weight_type(src0->type)) {
        // The bf16 GEMM rounds both operands to bf16 (8 significant bits).  f32 weights, and products the graph asks f32
        // precision for (GGML_PREC_F32: llama.cpp sets it on K*Q without flash attention, src/llama-graph.cpp build_attn_mha), stay
        // on the exact f32-accumulating mat-vec whatever the batch -- the reference keeps f16 inputs with f32 accumulation there
        // (ggml-cuda.cu:1205-1351) and ggml-cpu is f32 throughout.
        if (src0->type == GGML_TYPE_F32) return MM_FLOAT;
        if (dst && dst->op == GGML_OP_MUL_MAT && (enum ggml_prec)dst->op_params[0] == GGML_PREC_F32) return MM_FLOAT;
        return N > 16 && K % 64 == 0 && aligned && src1->type == GGML_TYPE_F32 ? MM_DQ : MM_FLOAT;
    }
    if (!is_weight_type(src0->type)) {      // q4_1 / q5_0 / q5_1 / q2_K / q3_K / iq4_nl / iq4_xs
        return N > 8 && K % 64 == 0 && aligned ? MM_DQ : MM_VEC;
    }
    if (!no_skinny && K % 256 == 0 && aligned && !(N > 8 && int8_mmq)) {
        // q8_K activations, integer sub-block scales on the matrix core: q4_K / q5_K digit-split (mmsk2), q6_K one 32x32x16 MFMA per
        // 16-weight sub-block (mmsk6; up to 4 columns the column-group mat-vec is still ahead: 15.8 against 20.5 us on 4096 x 14336)
        const bool kq = src0->type == GGML_TYPE_Q4_K || src0->type == GGML_TYPE_Q5_K || src0->type == GGML_TYPE_Q6_K;
        // One pass of 32 columns.  A second pass streams the weights again, and since the GEMM splits K its time is flat up to 128
        // columns (4096 x 14336 q4_K: 41 us at 33..128 columns against 37 / 56 / 74 for 2 / 3 / 4 skinny passes; 4096 x 4096: 18 against
        // 24 for 2), and the GEMM path brings the fused Q/K/V finish with it: 8B pp64 6889 -> 7545 t/s, pp128 12.3k -> 13.8k.
        // Round 3: 33..64 columns are ONE skinny launch too (two column tiles per workgroup: the weights cross HBM once, the fixed costs of
        // a launch are paid once): 8B pp33 / 40 / 48 / 64 4445 / 5394 / 6403 / 8389 t/s on the GEMM, 5216 / 6198 / 7285 / 9148 here.
        static const int64_t skinny_max = [] { const char * e = getenv("GGML_MI355_SKINNY_MAX"); return e ? atoll(e) : 64; }();
        // ... for the large matrices (ffn, 70B's wq / wo): on the 4096 x 4096 class the two tie as kernels and the GEMM brings the fused
        // Q/K/V finish and add + norm launches with it (8x7B, whose dense matrices are all of that class: pp64 4294 on the GEMM, 4069 here).
        // src0->ne[1] is the whole matrix's row count, so a row-split matrix still takes the unsplit one's path.
        static const int64_t wide_min = [] { const char * e = getenv("GGML_MI355_SKINNY_WIDE_MIN"); return e ? atoll(e) : (int64_t)32 << 20; }();
        const int64_t n_max = src0->ne[1] * K >= wide_min || (kq && N > 32 && act_read_by_kquants_only(src1)) ? skinny_max : std::min<int64_t>(skinny_max, 32);
        if (kq  && N >= (src0->type == GGML_TYPE_Q6_K ? 5 : 4) && N <= n_max) return MM_SKINNY_K;
        if (!kq && N > 8  && N <= (K >= 8192 ? 64 : 32))  return MM_SKINNY;
    }
    if (N > 8 && int8_mmq && K % 128 == 0) return MM_Q_INT8;
    if (N > 8 && K % 64 == 0 && aligned && !int8_mmq) return MM_DQ;
    return MM_VEC;
}

// planes of the activation scratch for `rows` rows of length K in the format of `path` (each 256-byte aligned)
mi355_act_layout mi355_act_planes(int path, int64_t K, int64_t rows) {
    mi355_act_layout l = {0, 0, 0};
    switch (mm_path_fmt(path)) {
        case 1: l.b0 = align_up((size_t)rows * K * 2, 256); break;
        case 2: l.b0 = align_up((size_t)rows * K, 256); l.b1 = align_up((size_t)rows * (K / 256) * 4, 256); l.b2 = align_up((size_t)rows * (K / 16) * 2, 256); break;
        default: l.b0 = align_up((size_t)mi355_act_q_bytes(K, rows), 256); l.b1 = align_up((size_t)mi355_act_ds_bytes(K, rows), 256); break;
    }
    return l;
}

// the launches of one MUL_MAT over rows [row_low, row_low + M) of the weight matrix whose first owned row is at W;
// `act` holds all of src1 in the format of `path`
// gemm_ws / gemm_ws_bytes: split-K workspace of the prefill GEMM for THIS stream (NULL: the kernel library's own, which it allocates)
static void mi355_mul_mat_launch(const ggml_tensor * dst, const char * W, int64_t row_low, int64_t M,
                                 const char * act, int path, hipStream_t stream, void * gemm_ws = nullptr, int64_t gemm_ws_bytes = 0) {
    const ggml_tensor * src0 = dst->src[0];
    const ggml_tensor * src1 = dst->src[1];
    const int64_t K = src0->ne[0], ne02 = src0->ne[2], ne03 = src0->ne[3];
    const int64_t N = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    const int64_t r2 = ne12 / ne02, r3 = ne13 / ne03;
    const mi355_act_layout L = mi355_act_planes(path, K, N * ne12 * ne13);
    const int8_t * aq = (const int8_t *)act;
    const float * ads = (const float *)(act + L.b0);
    char * dbase = (char *)dst->data + row_low * sizeof(float);
    const int64_t dcol = (int64_t)(dst->nb[1] / sizeof(float));
    if (path != MM_VEC) {
        // one launch (or pass of 32 columns) per broadcast channel
        for (int64_t i13 = 0; i13 < ne13; i13++) {
            for (int64_t i12 = 0; i12 < ne12; i12++) {
                const char * Wc = W + (i13 / r3) * src0->nb[3] + (i12 / r2) * src0->nb[2];
                const int64_t arow = (i13 * ne12 + i12) * N;
                float * d = (float *)(dbase + i13 * dst->nb[3] + i12 * dst->nb[2]);
                if (path == MM_DQ) {
                    MI_CHECK(mi355_mul_mat_dq_slice_ws((int)src0->type, Wc, (int64_t)src0->nb[1], (const uint16_t *)act + arow * K, K, d, dcol, M, src0->ne[1], K, N,
                                                       gemm_ws, gemm_ws_bytes, stream));
                } else if (path == MM_Q_INT8) {
                    MI_CHECK(mi355_mul_mat_q((int)src0->type, Wc, (int64_t)src0->nb[1], aq + arow * K, ads + arow * (K / 32) * 2, d, dcol, M, K, N, stream));
                } else {
                    const int64_t pass = path == MM_SKINNY_K ? 64 : 32;      // the K-quant kernels take 33..64 columns as two column tiles per workgroup
                    for (int64_t c0 = 0; c0 < N; c0 += pass) {
                        const int64_t r = arow + c0, nc = N - c0 < pass ? N - c0 : pass;
                        if (path == MM_SKINNY_K) {
                            const int16_t * bs = (const int16_t *)(act + L.b0 + L.b1);
                            MI_CHECK(mi355_mul_mat_skinny_k_slice((int)src0->type, Wc, (int64_t)src0->nb[1], aq + r * K, ads + r * (K / 256), bs + r * (K / 16),
                                                                  d + c0 * dcol, dcol, M, src0->ne[1], K, nc, stream));
                        } else {
                            MI_CHECK(mi355_mul_mat_skinny_slice((int)src0->type, Wc, (int64_t)src0->nb[1], aq + r * K, ads + r * (K / 32) * 2,
                                                                d + c0 * dcol, dcol, M, src0->ne[1], K, nc, stream));
                        }
                    }
                }
            }
        }
        return;
    }
    for (int64_t i13 = 0; i13 < ne13; i13++) {
        for (int64_t c0 = 0; c0 < N; c0 += 8) {
            mi355_mmvq_args a;
            memset(&a, 0, sizeof(a));
            a.type = (int32_t)src0->type;
            a.W = W + (i13 / r3) * src0->nb[3];
            a.w_row_stride = (int64_t)src0->nb[1];
            a.w_ch_stride  = (int64_t)src0->nb[2];
            a.act_q  = aq + (i13 * ne12 * N + c0) * K;
            a.act_ds = ads + (i13 * ne12 * N + c0) * (K / 32) * 2;
            a.act_ch_stride = N;
            a.dst = (float *)(dbase + i13 * dst->nb[3]) + c0 * dcol;
            a.dst_col_stride = dcol;
            a.dst_ch_stride  = (int64_t)(dst->nb[2] / sizeof(float));
            a.m = M; a.k = K;
            a.ncols = (int32_t)(N - c0 < 8 ? N - c0 : 8);
            a.nchannels = (int32_t)ne12;
            a.ch_ratio  = (int32_t)r2;
            MI_CHECK(mi355_mul_mat_vec_q(&a, stream));
        }
    }
}

static mi355_tensor to_mi355_tensor(const ggml_tensor * t);

void mi355_backend_ctx::peer_gemm_ws(int d, const ggml_tensor * src0, const mi355_split_extra * extra, int64_t K, int64_t N, int path, void ** ws, int64_t * bytes) {
    *ws = nullptr; *bytes = 0;
    if (path != MM_DQ) return;
    const int64_t own = extra->row_high[d] - extra->row_low[d];
    const int64_t need = mi355_mul_mat_dq_workspace_bytes(own > 0 ? own : 0, src0->ne[1], K, N);
    if (need <= 0) return;
    peer_t & p = peer(d);
    grow_on(d, &p.gemm_ws, &p.gemm_ws_size, (size_t)need, p.stream);
    mi355_set_device(device);
    *ws = p.gemm_ws; *bytes = need;
}

void mi355_mul_mat(mi355_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * src0 = dst->src[0];
    const ggml_tensor * src1 = dst->src[1];
    if (src0->type == GGML_TYPE_F32 && src0->ne[1] <= 1024 && src1->type == GGML_TYPE_F32) {
        // small f32 matrices (the MoE router): the kernel whose summation order the fused router reproduces
        mi355_tensor ta = to_mi355_tensor(src0), tb = to_mi355_tensor(src1), td = to_mi355_tensor(dst);
        MI_CHECK(mi355_mul_mat_f32(&ta, &tb, &td, ctx->stream));
        return;
    }
    if (is_float_weight_type(src0->type) && mi355_mul_mat_path(src0, src1, dst) == MM_FLOAT) {
        mi355_tensor ta = to_mi355_tensor(src0), tb = to_mi355_tensor(src1), td = to_mi355_tensor(dst);
        MI_CHECK(mi355_mul_mat_f(&ta, &tb, &td, ctx->stream));
        return;
    }
    const int64_t K = src0->ne[0], M = src0->ne[1];
    const int64_t N = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    const int64_t rows = N * ne12 * ne13;
    if (rows == 0 || M == 0) return;

    // Activations for all of src1 (strided / permuted src1 is handled by the converters), once per distinct src1 and
    // format: q8_1 for the mat-vec, q8_K for the K-quant skinny kernel, bf16 for the prefill GEMM.
    const int path = mi355_mul_mat_path(src0, src1, dst), fmt = mm_path_fmt(path);
    const mi355_act_layout L = mi355_act_planes(path, K, rows);
    const size_t qb = L.total(), db = 0;
    char * ws = (char *)ctx->get_scratch(0, qb);
    if (!ctx->act_cached(src1, fmt) || MI355_ENV_SET("GGML_MI355_NO_FUSE")) {
        const float * x = (const float *)src1->data;
        if (fmt == 1)      MI_CHECK(mi355_act_bf16(x, (uint16_t *)ws, K, N, ne12, ne13, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        else if (fmt == 2) MI_CHECK(mi355_quantize_act_q8k(x, (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), K, N, ne12, ne13,
                                                           src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        else               MI_CHECK(mi355_quantize_act(x, (int8_t *)ws, (float *)(ws + L.b0), K, N, ne12, ne13, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        ctx->act_remember(src1, fmt);
    }

    // split-K workspace of the prefill GEMM: from this stream's scratch, so the kernel library allocates nothing
    void * gemm_ws = nullptr;
    int64_t gemm_ws_bytes = 0;
    if (path == MM_DQ) {
        const int64_t own_rows = tensor_is_split(src0) ? ((const mi355_split_extra *)src0->extra)->row_high[ctx->device] - ((const mi355_split_extra *)src0->extra)->row_low[ctx->device] : M;
        gemm_ws_bytes = mi355_mul_mat_dq_workspace_bytes(own_rows > 0 ? own_rows : 0, M, K, N);
        if (gemm_ws_bytes > 0) gemm_ws = ctx->get_scratch(3, (size_t)gemm_ws_bytes);
        if (gemm_ws && ctx->scratch_size[3] < (size_t)gemm_ws_bytes) { gemm_ws = nullptr; gemm_ws_bytes = 0; }    // under capture the buffer could not grow
    }
    if (!tensor_is_split(src0)) {
        mi355_mul_mat_launch(dst, (const char *)src0->data, 0, M, ws, path, ctx->stream, gemm_ws, gemm_ws_bytes);
        return;
    }

    // Row split (reference: ggml_cuda_op_mul_mat, ggml-cuda.cu:1431-1749).  This stream's device is the main
    // one: it quantized src1 and owns dst.  Every other device with rows gets the quantized activations by one
    // peer copy (skipped when it already holds this version), runs the same kernels on its row slice and
    // stores its rows *directly into dst on the main device* (peer stores over xGMI) -- no staging buffer and
    // no gather copy.  One event forks the helpers off the main stream, one per helper joins them.
    const mi355_split_extra * extra = (const mi355_split_extra *)src0->extra;
    GGML_ASSERT(extra && src0->ne[2] == 1 && src0->ne[3] == 1);
    if (!ctx->ev_ready) HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ctx->ev_ready, ctx->stream));
    const int ndev = mi355_device_count();
    // all of dst's columns one after the other (nb[2] = N nb[1], ...): what the 2-D placement after the all-gather assumes
    const bool flat_cols = dst->nb[2] == (size_t)N * dst->nb[1] && dst->nb[3] == (size_t)ne12 * dst->nb[2];
    if (flat_cols && mi355_split_gather_rccl()) {
        // ---- RCCL form: every device computes its rows into a local [cols][rows_max] slice, one all-gather hands every device all
        // slices, the main device places each slice's rows into dst.  Same kernels on the same row slices as the peer-store form:
        // the same bits.
        mi355_rccl & r = mi355_rccl_state();
        int64_t rows_max = 1;
        for (int d = 0; d < ndev; d++) rows_max = std::max(rows_max, extra->row_high[d] - extra->row_low[d]);
        const size_t slice = (size_t)rows * rows_max;                 // floats per device
        for (int d = 0; d < ndev; d++) {
            const bool main = d == ctx->device;
            hipStream_t st = main ? ctx->stream : ctx->peer(d).stream;
            mi355_backend_ctx::split_buf_t & sb = ctx->split_bufs[d];
            mi355_backend_ctx::grow_on(d, (void **)&sb.part, &sb.part_size, slice * sizeof(float), st);
            mi355_backend_ctx::grow_on(d, (void **)&sb.gath, &sb.gath_size, slice * sizeof(float) * ndev, st);
            mi355_set_device(d);
            const char * act = ws;
            if (!main) {
                mi355_backend_ctx::peer_t & p = ctx->peers[d];
                HIP_CHECK(hipStreamWaitEvent(p.stream, ctx->ev_ready, 0));
                if (extra->data[d]) {
                    if (p.act_size < qb + db) {
                        HIP_CHECK(hipStreamSynchronize(p.stream));
                        if (p.act) HIP_CHECK(hipFree(p.act));
                        p.act_size = (qb + db + (1u << 20)) & ~((size_t)(1u << 20) - 1);
                        HIP_CHECK(hipMalloc(&p.act, p.act_size));
                        p.act_version = 0;
                    }
                    if (p.act_version != ctx->act_version) {
                        mi355_copy_d2d_async(p.act, d, ws, ctx->device, qb + db, p.stream);
                        p.act_version = ctx->act_version;
                    }
                    act = (const char *)p.act;
                }
            }
            if (extra->data[d]) {
                // dst as the kernels see it: column stride rows_max, rows [row_low, row_high) of the matrix at [0, rows_d)
                ggml_tensor local = *dst;
                local.nb[1] = (size_t)rows_max * sizeof(float);
                local.nb[2] = (size_t)N * local.nb[1];
                local.nb[3] = (size_t)ne12 * local.nb[2];
                local.data  = (char *)sb.part - extra->row_low[d] * (int64_t)sizeof(float);
                void * lws = gemm_ws; int64_t lws_bytes = gemm_ws_bytes;
                if (!main) ctx->peer_gemm_ws(d, src0, extra, K, N, path, &lws, &lws_bytes);
                mi355_mul_mat_launch(&local, (const char *)extra->data[d], extra->row_low[d], extra->row_high[d] - extra->row_low[d], act, path, st, lws, lws_bytes);
            }
        }
        RCCL_CHECK(r.GroupStart());
        for (int d = 0; d < ndev; d++) {
            mi355_set_device(d);
            RCCL_CHECK(r.AllGather(ctx->split_bufs[d].part, ctx->split_bufs[d].gath, slice, ncclFloat, r.comm[d], d == ctx->device ? ctx->stream : ctx->peers[d].stream));
        }
        RCCL_CHECK(r.GroupEnd());
        mi355_set_device(ctx->device);
        for (int d = 0; d < ndev; d++) {
            const int64_t nr = extra->row_high[d] - extra->row_low[d];
            if (nr <= 0) continue;
            HIP_CHECK(hipMemcpy2DAsync((char *)dst->data + extra->row_low[d] * sizeof(float), dst->nb[1],
                                       ctx->split_bufs[ctx->device].gath + (size_t)d * slice, (size_t)rows_max * sizeof(float),
                                       (size_t)nr * sizeof(float), (size_t)rows, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return;
    }
    for (int d = 0; d < ndev; d++) {
        if (d == ctx->device || !extra->data[d]) continue;
        mi355_backend_ctx::peer_t & p = ctx->peer(d);
        mi355_set_device(d);
        HIP_CHECK(hipStreamWaitEvent(p.stream, ctx->ev_ready, 0));
        if (p.act_size < qb + db && ctx->gc.capturing) {      // no allocation under capture: this capture is given up (nothing captured
            ctx->gc.capture_failed = true;                    // ever runs), the eager run that follows grows the buffer
            HIP_CHECK(hipEventRecord(p.done, p.stream));
            continue;
        }
        if (p.act_size < qb + db) {
            HIP_CHECK(hipStreamSynchronize(p.stream));
            if (p.act) HIP_CHECK(hipFree(p.act));
            p.act_size = (qb + db + (1u << 20)) & ~((size_t)(1u << 20) - 1);
            HIP_CHECK(hipMalloc(&p.act, p.act_size));
            p.act_version = 0;
        }
        if (p.act_version != ctx->act_version) {
            mi355_copy_d2d_async(p.act, d, ws, ctx->device, qb + db, p.stream);
            p.act_version = ctx->act_version;
        }
        void * lws = nullptr; int64_t lws_bytes = 0;
        ctx->peer_gemm_ws(d, src0, extra, K, N, path, &lws, &lws_bytes);
        mi355_set_device(d);
        mi355_mul_mat_launch(dst, (const char *)extra->data[d], extra->row_low[d], extra->row_high[d] - extra->row_low[d],
                             (const char *)p.act, path, p.stream, lws, lws_bytes);
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    mi355_set_device(ctx->device);
    if (extra->data[ctx->device]) {
        mi355_mul_mat_launch(dst, (const char *)extra->data[ctx->device], extra->row_low[ctx->device],
                             extra->row_high[ctx->device] - extra->row_low[ctx->device], ws, path, ctx->stream, gemm_ws, gemm_ws_bytes);
    }
    for (int d = 0; d < ndev; d++) {
        if (d == ctx->device || !extra->data[d]) continue;
        HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->peers[d].done, 0));
    }
}

// MUL_MAT_ID (contract ggml.c:3000-3036; reference ggml_cuda_mul_mat_id, ggml-cuda.cu:2065-2206;
// oracle ggml_compute_forward_mul_mat_id, ggml-cpu.c:1470-1648).  Expert routing stays on the
// device: the kernel reads ids itself, so there is no stream sync + host-side sort per call
// (the reference does both, ggml-cuda/mmq.cu:157-158,188-189).
// the skinny regime of MUL_MAT_ID (see mi355_mul_mat_id): pairs from max(12, 1.5 * n_expert) to 256, K-quants and q4_0 / q8_0
static bool moe_skinny_regime(const ggml_tensor * dst) {
    const ggml_tensor * as = dst->src[0], * b = dst->src[1], * ids = dst->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2], n_tokens = b->ne[2], n_used = ids->ne[0];
    static const int64_t sk_max = getenv("GGML_MI355_MOE_SKINNY_MAX") ? atoll(getenv("GGML_MI355_MOE_SKINNY_MAX")) : 256;
    static const int64_t sk_min = getenv("GGML_MI355_MOE_SKINNY_MIN") ? atoll(getenv("GGML_MI355_MOE_SKINNY_MIN")) : 12;
    const bool dense_dst = dst->nb[1] == (size_t)M * sizeof(float) && dst->nb[2] == (size_t)M * n_used * sizeof(float) &&
                           ((((uintptr_t)b->data) | b->nb[1] | b->nb[2]) & 15) == 0;
    return n_tokens * n_used >= std::max(sk_min, n_expert * 3 / 2) && n_tokens * n_used <= sk_max && K % 256 == 0 && n_expert <= 256 && dense_dst &&
           is_weight_type(as->type) && !MI355_ENV_SET("GGML_MI355_NO_SKINNY");
}

// MUL_MAT_ID(ffn_up_exps) and MUL_MAT_ID(ffn_gate_exps) of a MoE layer (two consecutive nodes on the same b and ids, build_moe_ffn) in
// the skinny regime: ONE launch -- one activation quantization, one bucketing, one grid over both tensors' row tiles
// (mi355_mul_mat_id_skinny_pair; each result has the bits of its own launch).  GGML_MI355_NO_MOE_PAIR=1 launches them one by one.
// What a skinny MUL_MAT_ID launch finds prepared: its activation planes in scratch[0] -- written by the launch that produced b (the norm
// or the GLU, see mi355_try_fuse_rms_norm_mul / mi355_glu_with_quant) or by an earlier MUL_MAT_ID on the same b -- and the routing tables of
// its ids in scratch[8] (the layer's up / gate launch built them; ffn_down routes by the same ids).  Returns the flags for
// mi355_mul_mat_id_skinny_ex; *tables: where the tables are or go (NULL: inside the workspace).  Call after get_scratch(0, need).
static int moe_skinny_prepared(mi355_backend_ctx * ctx, const ggml_tensor * b, const ggml_tensor * ids, enum ggml_type type, int64_t n_expert, int32_t ** tables) {
    *tables = nullptr;
    if (MI355_ENV_SET("GGML_MI355_NO_MOE_READY") || MI355_ENV_SET("GGML_MI355_NO_FUSE")) { ctx->act_cache.valid = false; ctx->moe_tables.valid = false; return 0; }
    int flags = 0;
    const bool kq = type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K;
    const int64_t K = b->ne[0], rows = b->ne[1] * b->ne[2];
    const bool dense_b = b->nb[0] == sizeof(float) && b->nb[1] == (size_t)K * sizeof(float) && b->nb[2] == (size_t)K * b->ne[1] * sizeof(float);
    if (dense_b && ctx->act_cached_rows(b->data, K, rows, kq ? 2 : 0)) flags |= MI355_IDS_ACT_READY;
    const int64_t tb = mi355_mul_mat_id_tables_bytes(ids->ne[1], ids->ne[0], n_expert);
    if (tb > 0) {
        const bool cached = ctx->moe_tables_cached(ids, n_expert);       // (asked before get_scratch, which forgets them if the block moves)
        int32_t * t = (int32_t *)ctx->get_scratch(8, (size_t)tb);
        if (ctx->scratch_size[8] >= (size_t)tb) {
            *tables = t;
            if (cached && ctx->moe_tables.valid) flags |= MI355_IDS_TABLES_READY;
        }
    }
    return flags;
}
// ... and what it leaves behind: b's planes at the start of scratch[0], the tables in scratch[8]
static void moe_skinny_remember(mi355_backend_ctx * ctx, const ggml_tensor * b, const ggml_tensor * ids, enum ggml_type type, int64_t n_expert, const int32_t * tables) {
    const bool kq = type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K;
    if (ggml_is_contiguous(b) && !MI355_ENV_SET("GGML_MI355_NO_MOE_READY") && !MI355_ENV_SET("GGML_MI355_NO_FUSE")) ctx->act_remember(b, kq ? 2 : 0);
    else ctx->act_cache.valid = false;
    if (tables) ctx->moe_tables_remember(ids, n_expert); else ctx->moe_tables.valid = false;
}

bool mi355_mul_mat_id_pair(mi355_backend_ctx * ctx, ggml_tensor * d0, ggml_tensor * d1) {
    if (d0->op != GGML_OP_MUL_MAT_ID || d1->op != GGML_OP_MUL_MAT_ID || MI355_ENV_SET("GGML_MI355_NO_MOE_PAIR")) return false;
    const ggml_tensor * a0 = d0->src[0], * a1 = d1->src[0], * b = d0->src[1], * ids = d0->src[2];
    if (d1->src[1] != b || d1->src[2] != ids || a0->type != a1->type || !ggml_are_same_shape(a0, a1) || a0->nb[1] != a1->nb[1] || a0->nb[2] != a1->nb[2]) return false;
    if (a0->type != GGML_TYPE_Q4_K && a0->type != GGML_TYPE_Q5_K && a0->type != GGML_TYPE_Q6_K) return false;
    if (tensor_is_split(a0) || tensor_is_split(a1) || a0->ne[3] != 1 || d0->type != GGML_TYPE_F32 || d1->type != GGML_TYPE_F32) return false;
    if (!moe_skinny_regime(d0) || !moe_skinny_regime(d1)) return false;
    const size_t nb = ggml_nbytes(d0);
    auto overlap = [](const void * p, size_t pn, const void * q, size_t qn) { return (const char *)p < (const char *)q + qn && (const char *)q < (const char *)p + pn; };
    if (overlap(d0->data, nb, d1->data, ggml_nbytes(d1)) || overlap(d0->data, nb, b->data, ggml_nbytes(b)) || overlap(d1->data, ggml_nbytes(d1), b->data, ggml_nbytes(b)) ||
        overlap(d0->data, nb, ids->data, ggml_nbytes(ids)) || overlap(d1->data, ggml_nbytes(d1), ids->data, ggml_nbytes(ids))) return false;
    const int64_t K = a0->ne[0], M = a0->ne[1], n_expert = a0->ne[2], b_ne1 = b->ne[1], n_tokens = b->ne[2], n_used = ids->ne[0];
    const int64_t need = mi355_mul_mat_id_skinny_workspace_bytes((int)a0->type, K, n_tokens, b_ne1, n_used, n_expert);
    void * ws = ctx->get_scratch(0, (size_t)need);
    if (ctx->scratch_size[0] < (size_t)need) return false;
    int32_t * tables = nullptr;
    const int flags = moe_skinny_prepared(ctx, b, ids, a0->type, n_expert, &tables);
    const int rc = mi355_mul_mat_id_skinny_ex((int)a0->type, a0->data, a1->data, (int64_t)a0->nb[1], (int64_t)a0->nb[2], (const float *)b->data,
                                              (int64_t)b->nb[1], (int64_t)b->nb[2], (const int32_t *)ids->data, (int64_t)(ids->nb[1] / sizeof(int32_t)),
                                              (float *)d0->data, (float *)d1->data, M, K, n_expert, n_used, n_tokens, b_ne1, ws, need, tables, flags,
                                              ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) { ctx->act_cache.valid = false; ctx->moe_tables.valid = false; return false; }
    MI_CHECK(rc);
    moe_skinny_remember(ctx, b, ids, a0->type, n_expert, tables);
    return true;
}

void mi355_mul_mat_id(mi355_backend_ctx * ctx, ggml_tensor * dst) {
    const ggml_tensor * as  = dst->src[0];
    const ggml_tensor * b   = dst->src[1];
    const ggml_tensor * ids = dst->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2];
    const int64_t b_ne1 = b->ne[1], n_tokens = b->ne[2];
    const int64_t n_used = ids->ne[0];
    if (n_tokens == 0 || M == 0) return;
    // a few to a few hundred pairs (SURVEY config 5, T = 8..64): bucket by expert on the device, then every expert's matrix
    // is streamed once per 32 of its columns through the skinny int8-MFMA kernels
    static const int64_t sk_max = getenv("GGML_MI355_MOE_SKINNY_MAX") ? atoll(getenv("GGML_MI355_MOE_SKINNY_MAX")) : 256;
    static const int64_t sk_min = getenv("GGML_MI355_MOE_SKINNY_MIN") ? atoll(getenv("GGML_MI355_MOE_SKINNY_MIN")) : 12;
    const bool dense_dst = dst->nb[1] == (size_t)M * sizeof(float) && dst->nb[2] == (size_t)M * n_used * sizeof(float) &&
                           ((((uintptr_t)b->data) | b->nb[1] | b->nb[2]) & 15) == 0;
    // (measured on 8x7B Q4_K_M, profiles/r03_moe_skinny_min.txt: 16 pairs 642 t/s here against 605 on the per-pair mat-vec since the
    // skinny kernels stage only the columns that exist -- it was 521 against 548 before --, 8 pairs 371 against 444; above 256 the GEMM)
    // Many experts: the skinny launch costs a pass per expert in use whatever its count, the per-pair mat-vec a pass per pair -- it stays
    // ahead up to about 1.5 pairs per expert (tools/moe_lab.py, q4_K: 64 experts x 2048 x 4096 top-6, 48 pairs 43 us against 66, 96 pairs
    // 74 against 70; 128 experts top-8, 128 pairs 83 against 113, 256 pairs 156 against 122; 8 experts: 16 pairs 89 against 83)
    const int64_t sk_from = std::max(sk_min, n_expert * 3 / 2);
    if (n_tokens * n_used >= sk_from && n_tokens * n_used <= sk_max && K % 256 == 0 && n_expert <= 256 && dense_dst && is_weight_type(as->type) &&
        !MI355_ENV_SET("GGML_MI355_NO_SKINNY")) {
        const int64_t need = mi355_mul_mat_id_skinny_workspace_bytes((int)as->type, K, n_tokens, b_ne1, n_used, n_expert);
        void * ws = ctx->get_scratch(0, (size_t)need);
        int32_t * tables = nullptr;
        const int flags = moe_skinny_prepared(ctx, b, ids, as->type, n_expert, &tables);
        const int rc = mi355_mul_mat_id_skinny_ex((int)as->type, as->data, nullptr, (int64_t)as->nb[1], (int64_t)as->nb[2], (const float *)b->data,
                                                  (int64_t)b->nb[1], (int64_t)b->nb[2], (const int32_t *)ids->data,
                                                  (int64_t)(ids->nb[1] / sizeof(int32_t)), (float *)dst->data, nullptr, M, K, n_expert, n_used,
                                                  n_tokens, b_ne1, ws, need, tables, flags, ctx->stream);
        if (rc == MI355_OK) { moe_skinny_remember(ctx, b, ids, as->type, n_expert, tables); return; }
        ctx->act_cache.valid = false; ctx->moe_tables.valid = false;
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    // prefill-sized batches: bucket the (token, slot) pairs by expert on the device and run one bf16-MFMA GEMM over all
    // experts, so every touched expert matrix is read once.  Below ~64 pairs the per-pair mat-vec (each pair streams
    // its expert's rows, mostly from the Infinity Cache) is faster than dequantizing whole tiles for a few columns.
    if (n_tokens * n_used >= std::max((int64_t)64, sk_from) && K % 64 == 0 && n_expert <= 256 && dense_dst && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {
        const int64_t need = mi355_mul_mat_id_dq_workspace_bytes(K, n_tokens, b_ne1, n_used, n_expert);
        void * ws = ctx->get_scratch(0, (size_t)need);
        ctx->act_cache.valid = false;     // the workspace replaces whatever activations were cached in scratch[0]
        const int rc = mi355_mul_mat_id_dq((int)as->type, as->data, (int64_t)as->nb[1], (int64_t)as->nb[2], (const float *)b->data,
                                           (int64_t)b->nb[1], (int64_t)b->nb[2], (const int32_t *)ids->data,
                                           (int64_t)(ids->nb[1] / sizeof(int32_t)), (float *)dst->data, M, K, n_expert, n_used, n_tokens,
                                           b_ne1, ws, need, ctx->stream);
        if (rc == MI355_OK) return;
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    const int64_t rows = b_ne1 * n_tokens;
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256);
    const size_t db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    int8_t * aq  = (int8_t *)ws;
    float  * ads = (float *)(ws + qb);
    if (!ctx->act_cached(b)) {   // gate / up experts read the same b
        MI_CHECK(mi355_quantize_act((const float *)b->data, aq, ads, K, b_ne1, n_tokens, 1, b->nb[1], b->nb[2], b->nb[3], ctx->stream));
        ctx->act_remember(b);
    }

    // (token, slot) pairs are the launch's channels; at most 65535 per launch
    const int64_t tok_per_launch = 65535 / n_used;
    for (int64_t t0 = 0; t0 < n_tokens; t0 += tok_per_launch) {
        const int64_t nt = n_tokens - t0 < tok_per_launch ? n_tokens - t0 : tok_per_launch;
        mi355_mmvq_args a;
        memset(&a, 0, sizeof(a));
        a.type = (int32_t)as->type;
        a.W = as->data;
        a.w_row_stride = (int64_t)as->nb[1];
        a.w_ch_stride  = (int64_t)as->nb[2];
        a.act_q  = aq + t0 * b_ne1 * K;
        a.act_ds = ads + t0 * b_ne1 * (K / 32) * 2;
        a.act_ch_stride = b_ne1;
        a.dst = (float *)((char *)dst->data + t0 * dst->nb[2]);
        a.dst_col_stride = M;
        a.dst_ch_stride  = (int64_t)(dst->nb[1] / sizeof(float));
        a.m = M; a.k = K; a.ncols = 1;
        a.nchannels = (int32_t)(nt * n_used);
        a.ch_ratio = 1;
        a.ids = (const int32_t *)((const char *)ids->data + t0 * ids->nb[1]);
        a.ids_stride = (int32_t)(ids->nb[1] / sizeof(int32_t));
        a.ids_nused  = (int32_t)n_used;
        a.act_ne1    = (int32_t)b_ne1;
        a.n_expert   = (int32_t)n_expert;
        MI_CHECK(mi355_mul_mat_vec_q(&a, ctx->stream));
    }
}

// FLASH_ATTN_EXT (contract ggml.h:2012-2033; reference dispatch ggml-cuda/fattn.cu:271-346;
// oracle ggml_compute_forward_flash_attn_ext_f16, ggml-cpu/ops.cpp:7975-8192)
static void fattn_args_from(const ggml_tensor * dst, mi355_fattn_args * a) {
    const ggml_tensor * q = dst->src[0], * k = dst->src[1], * v = dst->src[2], * mask = dst->src[3];
    memset(a, 0, sizeof(*a));
    a->q = (const float *)q->data; a->q_nb1 = q->nb[1]; a->q_nb2 = q->nb[2]; a->q_nb3 = q->nb[3];
    a->k = k->data; a->type_k = (int32_t)k->type; a->k_nb1 = k->nb[1]; a->k_nb2 = k->nb[2]; a->k_nb3 = k->nb[3];
    a->v = v->data; a->type_v = (int32_t)v->type; a->v_nb1 = v->nb[1]; a->v_nb2 = v->nb[2]; a->v_nb3 = v->nb[3];
    if (mask) { a->mask = mask->data; a->m_nb1 = mask->nb[1]; a->m_nb2 = mask->nb[2]; a->m_nb3 = mask->nb[3]; a->m_ne2 = mask->ne[2]; a->m_ne3 = mask->ne[3]; }
    a->dst = (float *)dst->data; a->d_nb1 = dst->nb[1]; a->d_nb2 = dst->nb[2]; a->d_nb3 = dst->nb[3];
    a->d = q->ne[0]; a->dv = v->ne[0]; a->n_q = q->ne[1]; a->n_head = q->ne[2]; a->ne3 = q->ne[3]; a->n_kv = k->ne[1]; a->n_head_kv = k->ne[2];
    memcpy(&a->scale, (const float *)dst->op_params + 0, 4);
    memcpy(&a->max_bias, (const float *)dst->op_params + 1, 4);
    memcpy(&a->logit_softcap, (const float *)dst->op_params + 2, 4);
}

bool mi355_flash_attn_supported(const ggml_tensor * op) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2], * mask = op->src[3];
    if (q->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
    if (k->type != v->type || (k->type != GGML_TYPE_F16 && k->type != GGML_TYPE_BF16 && k->type != GGML_TYPE_Q8_0 && k->type != GGML_TYPE_Q4_0)) return false;
    const int64_t d = q->ne[0];
    // head sizes: multiples of 8 up to 256, V's at most K's (DeepSeek's 192 / 128; not the 576 / 512 of its MLA form); quantized K/V rows are
    // whole 32-blocks by construction
    const int64_t dv = v->ne[0];
    if (d < 8 || d > 256 || d % 8 != 0 || k->ne[0] != d || dv < 8 || dv > d || dv % 8 != 0) return false;
    if (k->ne[3] != q->ne[3] || v->ne[3] != q->ne[3] || k->ne[2] != v->ne[2] || q->ne[2] % k->ne[2] != 0) return false;
    if (q->nb[0] != sizeof(float) || k->nb[0] != ggml_type_size(k->type) || v->nb[0] != ggml_type_size(v->type)) return false;
    if ((q->nb[1] | q->nb[2] | q->nb[3]) & 15) return false;
    if ((k->type == GGML_TYPE_F16 || k->type == GGML_TYPE_BF16) && ((k->nb[1] | k->nb[2] | k->nb[3] | v->nb[1] | v->nb[2] | v->nb[3]) & 15)) return false;
    if (mask && (mask->type != GGML_TYPE_F16 || mask->nb[0] != sizeof(ggml_fp16_t))) return false;
    if (!ggml_is_contiguous(op)) return false;
    if (q->ne[1] > 65535 || q->ne[3] * 64 > 65535) return false;
    return true;
}

static void mi355_flash_attn(mi355_backend_ctx * ctx, ggml_tensor * dst) {
    mi355_fattn_args a;
    fattn_args_from(dst, &a);
    // a workspace lets the KV range be split; only worth having for decode-sized batches
    void * ws = nullptr;
    int64_t ws_bytes = 0;
    // (decode-sized batches; and the 512-token ubatches of a long prompt, whose KV blocks are dealt to two workgroups per unit group)
    if ((a.n_kv >= 128 && a.n_head_kv * a.n_q * a.ne3 < 512) || (a.n_q >= 32 && a.n_kv >= 1024 && (a.n_q + 31) / 32 * a.n_head * a.ne3 < 1024)) {
        ws_bytes = mi355_flash_attn_ext_workspace_bytes(&a);
        ws = ctx->get_scratch(1, (size_t)ws_bytes);
    }
    // decode: the result (all heads of a token = one activation row) feeds the attention-output MUL_MAT; emit its
    // q8_1 form from the same kernel
    const int64_t K = a.dv * a.n_head, rows = a.n_q * a.ne3;
    if (rows <= 8 && ggml_is_contiguous(dst) && !MI355_ENV_SET("GGML_MI355_NO_FUSE")) {
        const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256), db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
        char * act = (char *)ctx->get_scratch(0, qb + db);
        const int rc = mi355_flash_attn_ext_quant(&a, ws, ws_bytes, (int8_t *)act, (float *)(act + qb), ctx->stream);
        if (rc == MI355_OK) {
            // remembered under the shape its consumer uses: kqv_out = [n_head * d, n_tokens]
            ggml_tensor as_rows = *dst;
            as_rows.ne[0] = K; as_rows.ne[1] = a.n_q; as_rows.ne[2] = a.ne3; as_rows.ne[3] = 1;
            as_rows.nb[0] = sizeof(float); as_rows.nb[1] = K * sizeof(float); as_rows.nb[2] = as_rows.nb[1] * a.n_q; as_rows.nb[3] = as_rows.nb[2] * a.ne3;
            ctx->act_remember(&as_rows);
            return;
        }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    // prefill: the bf16 rows for the attention-output GEMM from the same kernel
    if (rows > MI355_BF16_ROWS_ABOVE && ggml_is_contiguous(dst) && K % 8 == 0 && !MI355_ENV_SET("GGML_MI355_NO_FUSE") && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        const int rc = mi355_flash_attn_ext_bf16(&a, ws, ws_bytes, abf, ctx->stream);
        if (rc == MI355_OK) {
            ggml_tensor as_rows = *dst;
            as_rows.ne[0] = K; as_rows.ne[1] = a.n_q; as_rows.ne[2] = a.ne3; as_rows.ne[3] = 1;
            as_rows.nb[0] = sizeof(float); as_rows.nb[1] = K * sizeof(float); as_rows.nb[2] = as_rows.nb[1] * a.n_q; as_rows.nb[3] = as_rows.nb[2] * a.ne3;
            ctx->act_remember(&as_rows, 1);
            return;
        }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    MI_CHECK(mi355_flash_attn_ext(&a, ws, ws_bytes, ctx->stream));
}

static bool mi355_glu_with_quant(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next);

// next: the node after `node` in the graph (NULL at the end), for producers that write the activation format of their reader
bool mi355_compute_forward(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next) {
    switch (node->op) {
        case GGML_OP_MUL_MAT:        mi355_mul_mat(ctx, node);    return true;
        case GGML_OP_MUL_MAT_ID:     mi355_mul_mat_id(ctx, node); return true;
        case GGML_OP_FLASH_ATTN_EXT: mi355_flash_attn(ctx, node); return true;
        case GGML_OP_GLU:            if (mi355_glu_with_quant(ctx, node, next)) return true; break;
        default: break;
    }
    return mi355_ops_forward(node, ctx->stream);
}

// A fused kernel reads and writes per element index, so an output may share memory with an input only when the
// two coincide exactly (in-place); a shifted overlap (ggml-alloc reusing part of a freed block) would let one row's
// store land on another row's not-yet-read input.
static bool alias_safe(const ggml_tensor * out, const ggml_tensor * in) {
    const char * o0 = (const char *)out->data, * o1 = o0 + ggml_nbytes(out);
    const char * i0 = (const char *)in->data,  * i1 = i0 + ggml_nbytes(in);
    if (o1 <= i0 || i1 <= o0) return true;
    if (o0 != i0) return false;
    for (int d = 0; d < 4; d++) if (out->nb[d] != in->nb[d] || out->ne[d] != in->ne[d]) return false;
    return true;
}

static mi355_tensor to_mi355_tensor(const ggml_tensor * t) {
    mi355_tensor r;
    r.data = t->data; r.type = (int32_t)t->type;
    for (int d = 0; d < 4; d++) { r.ne[d] = t->ne[d]; r.nb[d] = (int64_t)t->nb[d]; }
    return r;
}

// RMS_NORM at node i followed by MUL with the per-feature norm weight; when `add` is given it is the ADD at node
// i-1 whose result the norm reads (residual add), and all three nodes run as one kernel.
// `gemm` (with `add`): the MUL_MAT whose product is one addend of `add` has not run yet; it runs here as mi355_mul_mat_dq_parts and the
// norm kernel adds its K-slice planes on the way in (prefill: wo / ffn_down -> residual ADD -> RMS_NORM -> MUL, no slice-sum launch).
// Nothing is launched unless the whole chain can be.
bool mi355_try_fuse_rms_norm_mul(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, ggml_tensor * add, const ggml_tensor * gemm) {
    ggml_tensor * norm = ggml_graph_node(cgraph, i);
    ggml_tensor * mul  = ggml_graph_node(cgraph, i + 1);
    if (mul->op != GGML_OP_MUL || MI355_ENV_SET("GGML_MI355_NO_FUSE")) return false;
    const ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : (mul->src[1] == norm ? mul->src[0] : nullptr);
    if (!w || w == norm) return false;
    if (w->type != GGML_TYPE_F32 || mul->type != GGML_TYPE_F32 || norm->type != GGML_TYPE_F32) return false;
    if (w->ne[0] != norm->ne[0] || ggml_nrows(w) != 1 || w->nb[0] != sizeof(float)) return false;
    if (!ggml_are_same_shape(norm, mul) || norm->nb[0] != sizeof(float) || mul->nb[0] != sizeof(float)) return false;
    if (norm->src[0]->nb[0] != sizeof(float)) return false;
    if (norm->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
    const int n = ggml_graph_n_nodes(cgraph);
    for (int j = i + 2; j < n; j++) {           // any other reader of the un-multiplied norm?
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s = 0; s < GGML_MAX_SRC; s++) if (t->src[s] == norm || (t->src[s] && t->src[s]->view_src == norm)) return false;
    }
    const ggml_tensor * x = norm->src[0];
    if (!alias_safe(mul, x) || !alias_safe(mul, w)) return false;
    if (add) {
        const ggml_tensor * a = add->src[0], * b = add->src[1];
        if (x != add || add->type != GGML_TYPE_F32 || a->type != GGML_TYPE_F32 || b->type != GGML_TYPE_F32) return false;
        if (!ggml_are_same_shape(a, add) || !ggml_are_same_shape(b, add)) return false;
        if (a->nb[0] != sizeof(float) || b->nb[0] != sizeof(float) || add->nb[0] != sizeof(float)) return false;
        if (!alias_safe(add, a) || !alias_safe(add, b) || !alias_safe(mul, a) || !alias_safe(mul, b)) return false;
        const char * m0 = (const char *)mul->data, * s0 = (const char *)add->data;
        if (m0 < s0 + ggml_nbytes(add) && s0 < m0 + ggml_nbytes(mul)) return false;   // the two outputs must be disjoint
    }
    float eps;
    memcpy(&eps, norm->op_params, sizeof(float));
    mi355_tensor tx = to_mi355_tensor(x), td = to_mi355_tensor(mul);
    const int64_t rows = ggml_nrows(mul), K = mul->ne[0];
    // batches whose MUL_MATs take the K-quant skinny kernels (5..32 rows; the first MUL_MAT that reads the result tells): also emit
    // the q8_K planes they read, instead of a quantization launch per batch and norm (126 such launches in a 16-token 8B step)
    if (!gemm && rows > 1 && rows <= 64 && K % 256 == 0 && K <= 8192 && ggml_is_contiguous(mul) && !MI355_ENV_SET("GGML_MI355_NO_NORM_Q8K")) {
        const int nn = ggml_graph_n_nodes(cgraph);
        int fmt = -1;
        for (int j = i + 2; j < nn && j < i + 16 && fmt < 0; j++) {
            const ggml_tensor * t = ggml_graph_node(cgraph, j);
            if (t->op == GGML_OP_MUL_MAT && t->src[1] == mul && !tensor_is_split(t->src[0])) {
                const int path = mi355_mul_mat_path(t->src[0], mul, t);
                if (path == MM_FLOAT) continue;          // a float-weight product (the MoE router) reads the f32 rows themselves
                fmt = mm_path_fmt(path);
            }
            // the experts of a MoE layer in the skinny regime read the q8_K planes too (the tensor they name is a reshaped view of the result)
            if (t->op == GGML_OP_MUL_MAT_ID && t->src[1]->data == mul->data && (t->src[1] == mul || t->src[1]->view_src == mul) && moe_skinny_regime(t) &&
                t->src[1]->ne[0] == K && t->src[1]->ne[1] * t->src[1]->ne[2] == rows && ggml_is_contiguous(t->src[1]) &&
                (t->src[0]->type == GGML_TYPE_Q4_K || t->src[0]->type == GGML_TYPE_Q5_K || t->src[0]->type == GGML_TYPE_Q6_K) &&
                !MI355_ENV_SET("GGML_MI355_NO_MOE_READY")) fmt = 2;
        }
        if (fmt == 2) {
            const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, rows);
            char * ws = (char *)ctx->get_scratch(0, L.total());
            mi355_tensor ta = to_mi355_tensor(add ? add->src[0] : x), tb, ts;
            if (add) { tb = to_mi355_tensor(add->src[1]); ts = to_mi355_tensor(add); }
            const int rc = mi355_rms_norm_mul_q8k_f32(&ta, add ? &tb : nullptr, add ? &ts : nullptr, (const float *)w->data, &td, eps,
                                                      (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), ctx->stream);
            if (rc == MI355_OK) {
                if (add) ctx->act_invalidate_if_written(add);
                ctx->act_invalidate_if_written(mul);
                ctx->act_remember(mul, 2);
                return true;
            }
            if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
        }
    }
    // decode-sized rows: also emit the q8_1 form for the MUL_MATs that read the result
    const bool quant = rows <= 8 && K % 32 == 0 && K <= 8192 && ggml_is_contiguous(mul);
    int8_t * aq = nullptr; float * ads = nullptr;
    if (quant) {
        const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256), db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
        char * ws = (char *)ctx->get_scratch(0, qb + db);
        aq = (int8_t *)ws; ads = (float *)(ws + qb);
    }
    // prefill-sized rows (every MUL_MAT that reads them takes the bf16 GEMM): also emit the bf16 rows, so those MUL_MATs
    // need no conversion launch
    if (gemm) {
        // everything the norm kernel could refuse is checked before the GEMM is launched
        const ggml_tensor * other = add->src[0] == gemm ? add->src[1] : add->src[0];
        uintptr_t al = (uintptr_t)other->data | (uintptr_t)add->data | (uintptr_t)mul->data | (uintptr_t)w->data;
        for (int d = 1; d < 4; d++) al |= (uintptr_t)other->nb[d] | (uintptr_t)add->nb[d] | (uintptr_t)mul->nb[d];
        if (!(rows > MI355_BF16_ROWS_ABOVE && K % 16 == 0 && K <= 8192 && ggml_is_contiguous(mul) && ggml_is_contiguous(add) && (al & 15) == 0) || MI355_ENV_SET("GGML_MI355_MMQ_INT8")) return false;
        const ggml_tensor * W = gemm->src[0], * src1 = gemm->src[1];
        const int64_t pb = mi355_mul_mat_dq_parts_bytes(W->ne[1], W->ne[1], W->ne[0], src1->ne[1]);
        if (pb <= 0 || W->ne[1] != K || src1->ne[1] != rows) return false;
        float * parts = (float *)ctx->get_scratch(3, (size_t)pb);
        if (ctx->scratch_size[3] < (size_t)pb) return false;
        const uint16_t * act = prefill_act_bf16(ctx, src1);
        int32_t kz = 0;
        const int rc0 = mi355_mul_mat_dq_parts((int)W->type, W->data, (int64_t)W->nb[1], act, W->ne[0], parts, pb, W->ne[1], W->ne[1], W->ne[0], src1->ne[1], &kz, ctx->stream);
        if (rc0 == MI355_E_UNSUPPORTED || rc0 == MI355_E_ALIGN) return false;
        MI_CHECK(rc0);
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        mi355_tensor tb = to_mi355_tensor(other), ts = to_mi355_tensor(add);
        MI_CHECK(mi355_parts_add_rms_norm_mul_bf16_f32(parts, kz, &tb, &ts, (const float *)w->data, &td, eps, abf, ctx->stream));
        ctx->act_invalidate_if_written(add);
        ctx->act_invalidate_if_written(mul);
        ctx->act_remember(mul, 1);
        return true;
    }
    if (rows > MI355_BF16_ROWS_ABOVE && K % 16 == 0 && K <= 8192 && ggml_is_contiguous(mul) && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        mi355_tensor ta = to_mi355_tensor(add ? add->src[0] : x), tb, ts;
        if (add) { tb = to_mi355_tensor(add->src[1]); ts = to_mi355_tensor(add); }
        const int rc = mi355_rms_norm_mul_bf16_f32(&ta, add ? &tb : nullptr, add ? &ts : nullptr, (const float *)w->data, &td, eps, abf, ctx->stream);
        if (rc == MI355_OK) {
            if (add) ctx->act_invalidate_if_written(add);
            ctx->act_invalidate_if_written(mul);
            ctx->act_remember(mul, 1);
            return true;
        }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    if (add) {
        mi355_tensor ta = to_mi355_tensor(add->src[0]), tb = to_mi355_tensor(add->src[1]), ts = to_mi355_tensor(add);
        int rc = mi355_add_rms_norm_mul_f32(&ta, &tb, &ts, (const float *)w->data, &td, eps, aq, ads, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED && quant) {
            aq = nullptr; ads = nullptr;
            rc = mi355_add_rms_norm_mul_f32(&ta, &tb, &ts, (const float *)w->data, &td, eps, nullptr, nullptr, ctx->stream);
        }
        if (rc == MI355_E_UNSUPPORTED) return false;
        MI_CHECK(rc);
        ctx->act_invalidate_if_written(add);
        ctx->act_invalidate_if_written(mul);
        if (aq) ctx->act_remember(mul);
        return true;
    }
    ctx->act_invalidate_if_written(mul);
    if (quant) {
        const int rc = mi355_rms_norm_mul_quant_f32(&tx, (const float *)w->data, &td, eps, aq, ads, ctx->stream);
        if (rc == MI355_OK) { ctx->act_remember(mul); return true; }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    MI_CHECK(mi355_rms_norm_mul_f32(&tx, (const float *)w->data, &td, eps, ctx->stream));
    return true;
}

// GLU whose result feeds ffn_down: the activation format ffn_down reads (q8_1 at decode, q8_K planes for 5..32 rows over a K-quant,
// bf16 rows for the prefill GEMM) is written by the same launch
static bool mi355_glu_with_quant(mi355_backend_ctx * ctx, ggml_tensor * node, const ggml_tensor * next) {
    const ggml_tensor * a = node->src[0], * b = node->src[1];
    const int64_t rows = ggml_nrows(node), K = node->ne[0];
    if (!ggml_is_contiguous(node) || MI355_ENV_SET("GGML_MI355_NO_FUSE")) return false;
    auto T = [](const ggml_tensor * t) { mi355_tensor r; r.data = t->data; r.type = (int32_t)t->type;
                                         for (int d = 0; d < 4; d++) { r.ne[d] = t->ne[d]; r.nb[d] = (int64_t)t->nb[d]; } return r; };
    mi355_tensor ta = T(a), tb, td = T(node);
    if (b) tb = T(b);
    // the experts' ffn_down of a MoE layer in the skinny regime (up to 256 (token, slot) rows) reads q8_K planes: written here, its
    // MUL_MAT_ID would quantize the rows itself otherwise (asked first: such a step has more than 32 rows from 17 tokens on)
    if (next && next->op == GGML_OP_MUL_MAT_ID && next->src[1] == node && rows > 1 && K % 256 == 0 && moe_skinny_regime(next) &&
        (next->src[0]->type == GGML_TYPE_Q4_K || next->src[0]->type == GGML_TYPE_Q5_K || next->src[0]->type == GGML_TYPE_Q6_K) &&
        !MI355_ENV_SET("GGML_MI355_NO_NORM_Q8K") && !MI355_ENV_SET("GGML_MI355_NO_MOE_READY")) {
        const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, rows);
        // sized for the MUL_MAT_ID's whole workspace already, so that its own request does not move the block
        const ggml_tensor * ids = next->src[2];
        const int64_t need = mi355_mul_mat_id_skinny_workspace_bytes((int)next->src[0]->type, K, node->ne[2], node->ne[1], ids->ne[0], next->src[0]->ne[2]);
        char * ws = (char *)ctx->get_scratch(0, std::max(L.total(), (size_t)(need > 0 ? need : 0)));
        const int rc = mi355_glu_q8k_f32((int)ggml_get_glu_op(node), &ta, b ? &tb : nullptr, &td, ((const int32_t *)node->op_params)[1],
                                         (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), ctx->stream);
        if (rc == MI355_OK) { ctx->act_remember(node, 2); return true; }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    // ... and a ffn_down that takes the K-quant skinny kernels (up to 64 rows: asked before the bf16 form, which starts at 33)
    if (next && next->op == GGML_OP_MUL_MAT && next->src[1] == node && !tensor_is_split(next->src[0]) && rows > 1 && K % 256 == 0 &&
        mi355_mul_mat_path(next->src[0], node) == MM_SKINNY_K && !MI355_ENV_SET("GGML_MI355_NO_NORM_Q8K")) {
        const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, rows);
        char * ws = (char *)ctx->get_scratch(0, L.total());
        const int rc = mi355_glu_q8k_f32((int)ggml_get_glu_op(node), &ta, b ? &tb : nullptr, &td, ((const int32_t *)node->op_params)[1],
                                         (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), ctx->stream);
        if (rc == MI355_OK) { ctx->act_remember(node, 2); return true; }
        if (rc != MI355_E_UNSUPPORTED) MI_CHECK(rc);
    }
    if (rows > MI355_BF16_ROWS_ABOVE && b && K % 4 == 0 && !MI355_ENV_SET("GGML_MI355_MMQ_INT8")) {   // prefill: bf16 rows for the ffn_down GEMM
        uint16_t * abf = (uint16_t *)ctx->get_scratch(0, align_up((size_t)rows * K * 2, 256));
        const int rc = mi355_glu_bf16_f32((int)ggml_get_glu_op(node), &ta, &tb, &td, ((const int32_t *)node->op_params)[1], abf, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED) return false;
        MI_CHECK(rc);
        ctx->act_remember(node, 1);
        return true;
    }
    if (rows > 8 || K % 32 != 0) return false;
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, rows), 256), db = align_up((size_t)mi355_act_ds_bytes(K, rows), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    const int rc = mi355_glu_quant_f32((int)ggml_get_glu_op(node), &ta, b ? &tb : nullptr, &td, ((const int32_t *)node->op_params)[1],
                                       (int8_t *)ws, (float *)(ws + qb), ctx->stream);
    if (rc == MI355_E_UNSUPPORTED) return false;
    MI_CHECK(rc);
    ctx->act_remember(node);
    return true;
}

