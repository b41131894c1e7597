// mi355-fusions.cpp -- the graph-level part of the ggml-mi355 plugin: pattern matchers over llama.cpp's node order, the fused launches they
// issue (every one bit-identical to the nodes run one by one), and mi355_issue_graph, the loop that walks a graph.
#include "mi355-backend.h"

// ------------------------------------------------------------------------------------------------
// Grouped decode mat-vecs.  wq/wk/wv (and ffn gate/up) read the same activation; llama.cpp's graph has them a few
// nodes apart (the rope of Q sits between wq and wk).  When MUL_MAT node i is a decode mat-vec, later MUL_MATs with
// the same src1, weight type and shapes are HOISTED into its launch (mi355_mul_mat_vec_q_group) -- legal when
// nothing between i and j reads or writes the bytes node j's result will occupy (ggml-alloc reuses freed blocks, so
// this is checked on addresses) and nothing in between writes src1.  Hoisted nodes are marked done and skipped.
// ------------------------------------------------------------------------------------------------
static bool ranges_overlap(const void * a, size_t an, const void * b, size_t bn) {
    const char * a0 = (const char *)a, * b0 = (const char *)b;
    return a0 < b0 + bn && b0 < a0 + an;
}

static bool is_view_op(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}

static const ggml_tensor * view_root(const ggml_tensor * t) { return t->view_src ? t->view_src : t; }

// Row-split weights (-sm row) in the fused decode launches: every fused kernel takes a row range, and the 128-row rounding of the
// split keeps rope pairs, GLU pairs and q8_1 blocks inside a slice -- so each device that owns rows runs the SAME launch on its slice
// (weights from its own memory, activations and results in the main device's over xGMI) between one event fork and one join
// (mi355_split_fan_out).  GGML_MI355_NO_SPLIT_FUSE=1: split MUL_MATs run one by one through mi355_mul_mat as before.
static bool split_fusable(const ggml_tensor * w) {
    static const bool off = MI355_ENV_SET("GGML_MI355_NO_SPLIT_FUSE");
    return !off && tensor_is_split(w) && w->extra && w->ne[2] == 1 && w->ne[3] == 1;
}

static bool matvec_groupable(const ggml_tensor * node) {
    if (node->op != GGML_OP_MUL_MAT) return false;
    const ggml_tensor * a = node->src[0], * b = node->src[1];
    if (!is_weight_type(a->type) || (tensor_is_split(a) && !split_fusable(a)) || a->op != GGML_OP_NONE || a->view_src) return false;
    if (a->ne[2] != 1 || a->ne[3] != 1 || b->ne[2] != 1 || b->ne[3] != 1 || b->ne[1] < 1 || b->ne[1] > 8) return false;
    if (mi355_mul_mat_path(a, b) != MM_VEC) return false;      // e.g. 4..8 columns of a K-quant: the skinny MFMA kernel is faster
    return ggml_is_contiguous(node) && a->ne[1] > 0;
}

static int mi355_collect_matvec_group(ggml_cgraph * cgraph, int i, const std::vector<char> & done, int * members) {
    const int n = ggml_graph_n_nodes(cgraph);
    const ggml_tensor * first = ggml_graph_node(cgraph, i);
    const ggml_tensor * src1 = first->src[1];
    int cnt = 0;
    members[cnt++] = i;
    const int horizon = i + 12 < n ? i + 12 : n;
    for (int j = i + 1; j < horizon && cnt < MI355_MMVQ_MAX_GROUP; j++) {
        const ggml_tensor * cand = ggml_graph_node(cgraph, j);
        if (done[j] || !matvec_groupable(cand) || cand->src[1] != src1 || cand->src[0]->type != first->src[0]->type ||
            cand->src[0]->ne[0] != first->src[0]->ne[0] || tensor_is_split(cand->src[0]) != tensor_is_split(first->src[0])) continue;
        const size_t dn = ggml_nbytes(cand);
        bool safe = !ranges_overlap(cand->data, dn, src1->data, ggml_nbytes(src1));
        for (int m = 0; m < cnt && safe; m++) {
            const ggml_tensor * g = ggml_graph_node(cgraph, members[m]);
            if (ranges_overlap(cand->data, dn, g->data, ggml_nbytes(g))) safe = false;
        }
        for (int t = i + 1; t < j && safe; t++) {
            const ggml_tensor * mid = ggml_graph_node(cgraph, t);
            if (done[t] || ggml_is_empty(mid)) continue;
            bool member = false;
            for (int m = 0; m < cnt; m++) if (members[m] == t) member = true;
            if (member) continue;
            if (!is_view_op(mid)) {
                if (ranges_overlap(cand->data, dn, mid->data, ggml_nbytes(mid))) safe = false;              // would clobber / be clobbered
                if (ranges_overlap(src1->data, ggml_nbytes(src1), mid->data, ggml_nbytes(mid))) safe = false; // src1 rewritten before j
            }
            for (int sidx = 0; sidx < GGML_MAX_SRC && safe; sidx++) {
                const ggml_tensor * ms = mid->src[sidx];
                if (ms && ms->data && ranges_overlap(cand->data, dn, ms->data, ggml_nbytes(ms))) safe = false; // reads the old contents
            }
        }
        if (safe) members[cnt++] = j;
    }
    return cnt;
}

// One fused decode op over row-split weights: `has[d]` says which devices own rows, `launch(d, stream)` issues device d's slice.  The
// helpers start behind everything the main stream has issued so far (one event), the main stream goes on behind all of them (one
// event each).  Under capture the helper streams join the capture through the first wait and leave it through the last.
// A slice whose device is the SAME physical GPU as the main device (virtual devices) is issued on the main stream itself: a second
// stream on one GPU buys no parallel hardware and every fork / join edge costs ~5 us on the GPU and in hipGraphLaunch (2 virtual
// devices, 8B Q4_K_M tg64: 313 t/s through helper streams, 100 before the split forms; unsplit 510).  GGML_MI355_SPLIT_STREAMS=1
// keeps the helper streams there too -- the tests run both, so that the fork / join path stays exercised on the one-GPU box.
template <class F>
static void mi355_split_fan_out(mi355_backend_ctx * ctx, const bool * has, F launch) {
    const int ndev = mi355_device_count();
    static const bool force_streams = MI355_ENV_SET("GGML_MI355_SPLIT_STREAMS");
    bool helper[MI355_MAX_DEVICES] = {};
    int n_helpers = 0;
    for (int d = 0; d < ndev; d++) {
        helper[d] = d != ctx->device && has[d] && (force_streams || mi355_phys(d) != mi355_phys(ctx->device));
        n_helpers += helper[d];
    }
    if (n_helpers == 0) {
        for (int d = 0; d < ndev; d++) if (has[d]) launch(d, ctx->stream);
        return;
    }
    if (!ctx->ev_ready) HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ctx->ev_ready, ctx->stream));
    for (int d = 0; d < ndev; d++) {
        if (!helper[d]) continue;
        mi355_backend_ctx::peer_t & p = ctx->peer(d);
        mi355_set_device(d);
        HIP_CHECK(hipStreamWaitEvent(p.stream, ctx->ev_ready, 0));
        launch(d, p.stream);
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    mi355_set_device(ctx->device);
    for (int d = 0; d < ndev; d++) if (has[d] && !helper[d]) launch(d, ctx->stream);      // the main device's slice (and same-GPU ones)
    for (int d = 0; d < ndev; d++) {
        if (!helper[d]) continue;
        HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->peers[d].done, 0));
    }
}

static const mi355_split_extra * split_extra_of(const ggml_tensor * w) { return (const mi355_split_extra *)w->extra; }

// norm_x / norm_w / norm_eps (one column only): src1 = (norm_x * rsqrt(mean(norm_x^2) + eps)) * norm_w is computed, stored and quantized in the
// launch's prologue (mi355_mmvq_args.src_f32) instead of being read from a launch of its own
static void mi355_mul_mat_group(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, const int * members, int cnt,
                                const float * residual = nullptr, float * sum = nullptr,
                                const float * norm_x = nullptr, const float * norm_w = nullptr, float norm_eps = 0.0f) {
    const ggml_tensor * first = ggml_graph_node(cgraph, members[0]);
    const ggml_tensor * src1 = first->src[1];
    const int64_t K = src1->ne[0], N = src1->ne[1];
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, N), 256);
    const size_t db = align_up((size_t)mi355_act_ds_bytes(K, N), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    if (!norm_x && !ctx->act_cached(src1)) {
        MI_CHECK(mi355_quantize_act((const float *)src1->data, (int8_t *)ws, (float *)(ws + qb), K, N, 1, 1, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        ctx->act_remember(src1);
    }
    mi355_mmvq_args a[MI355_MMVQ_MAX_GROUP];
    memset(a, 0, sizeof(a));
    for (int m = 0; m < cnt; m++) {
        const ggml_tensor * node = ggml_graph_node(cgraph, members[m]);
        const ggml_tensor * w = node->src[0];
        a[m].type = (int32_t)w->type;
        a[m].W = w->data; a[m].w_row_stride = (int64_t)w->nb[1];
        a[m].act_q = (const int8_t *)ws; a[m].act_ds = (const float *)(ws + qb); a[m].act_ch_stride = N;
        a[m].dst = (float *)node->data; a[m].dst_col_stride = (int64_t)(node->nb[1] / sizeof(float));
        a[m].m = w->ne[1]; a[m].k = K; a[m].ncols = (int32_t)N; a[m].nchannels = 1; a[m].ch_ratio = 1;
        if (norm_x) {
            a[m].act_q = nullptr; a[m].act_ds = nullptr;
            a[m].src_f32 = norm_x; a[m].norm_w = norm_w; a[m].norm_out = (float *)src1->data; a[m].norm_eps = norm_eps;
        }
    }
    a[0].residual = residual; a[0].sum = sum;      // mat-vec + ADD (cnt == 1)
    if (tensor_is_split(first->src[0])) {
        // every device launches the group over its row slices: the members it owns rows of, results (and the residual sum) at the
        // slice's place in the main device's tensors
        bool has[MI355_MAX_DEVICES] = {};
        const int ndev = mi355_device_count();
        for (int m = 0; m < cnt; m++) {
            const mi355_split_extra * ex = split_extra_of(ggml_graph_node(cgraph, members[m])->src[0]);
            for (int d = 0; d < ndev; d++) if (ex->data[d] && ex->row_high[d] > ex->row_low[d]) has[d] = true;
        }
        mi355_split_fan_out(ctx, has, [&](int d, hipStream_t st) {
            mi355_mmvq_args b[MI355_MMVQ_MAX_GROUP];
            int nb = 0;
            for (int m = 0; m < cnt; m++) {
                const mi355_split_extra * ex = split_extra_of(ggml_graph_node(cgraph, members[m])->src[0]);
                const int64_t lo = ex->row_low[d], rows = ex->row_high[d] - lo;
                if (!ex->data[d] || rows <= 0) continue;
                b[nb] = a[m];
                b[nb].W = ex->data[d]; b[nb].m = rows; b[nb].dst = a[m].dst + lo;
                if (a[m].residual) { b[nb].residual = a[m].residual + lo; b[nb].sum = a[m].sum + lo; }
                nb++;
            }
            if (nb) MI_CHECK(mi355_mul_mat_vec_q_group(b, nb, st));
        });
        return;
    }
    MI_CHECK(mi355_mul_mat_vec_q_group(a, cnt, ctx->stream));
}

// Decode mat-vec followed by the residual ADD (wo / ffn_down of every layer: cur = MUL_MAT(w, x); out = ADD(cur, inp)) in
// one launch: the lane that stores a result element also stores result + residual.  Per-element, so the ADD may be in
// place on either operand; both tensors are written unless the sum takes the result's place.  With the ADD gone from
// it, the norm that follows needs no residual input -- which is what lets it move into the next mat-vec's prologue
// (mi355_try_fuse_norm_glu_matvec): there every workgroup reads the row, so nothing may update it in that launch.
static bool mi355_try_fuse_matvec_add(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 1 >= n || MI355_ENV_SET("GGML_MI355_NO_MATVEC_ADD")) return false;
    ggml_tensor * mm = ggml_graph_node(cgraph, i), * add = ggml_graph_node(cgraph, i + 1);
    if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32) return false;
    const ggml_tensor * r = add->src[0] == mm ? add->src[1] : (add->src[1] == mm ? add->src[0] : nullptr);
    if (!r || r == mm || r->type != GGML_TYPE_F32) return false;
    if (!ggml_are_same_shape(mm, add) || !ggml_are_same_shape(r, add) || !ggml_is_contiguous(add) || !ggml_is_contiguous(r)) return false;
    const size_t nb = ggml_nbytes(add);
    if (ranges_overlap(r->data, nb, mm->data, nb)) return false;
    if (add->data != mm->data && ranges_overlap(add->data, nb, mm->data, nb)) return false;
    if (add->data != r->data && ranges_overlap(add->data, nb, r->data, nb)) return false;
    // neither output may land on the activations this mat-vec is about to be given
    const ggml_tensor * src1 = mm->src[1];
    if (ranges_overlap(add->data, nb, src1->data, ggml_nbytes(src1))) return false;
    const int members[1] = {i};
    mi355_mul_mat_group(ctx, cgraph, members, 1, (const float *)r->data, (float *)add->data);
    ctx->act_invalidate_if_written(add);
    return true;
}

// ---- norm prologue of the fused decode kernels (mi355_glu_matvec / mi355_qkv_rope_matvec with norm_x set) ----------
// RMS_NORM at node i followed by MUL with the per-feature norm weight at i+1, one token, nothing else reading the bare
// norm (the checks of mi355_try_fuse_rms_norm_mul): returns the weight tensor, or NULL.
static const ggml_tensor * decode_norm_pattern(ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 1 >= n || done[i] || done[i + 1]) return nullptr;
    const ggml_tensor * norm = ggml_graph_node(cgraph, i), * mul = ggml_graph_node(cgraph, i + 1);
    if (norm->op != GGML_OP_RMS_NORM || mul->op != GGML_OP_MUL) return nullptr;
    const ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : (mul->src[1] == norm ? mul->src[0] : nullptr);
    if (!w || w == norm) return nullptr;
    if (w->type != GGML_TYPE_F32 || mul->type != GGML_TYPE_F32 || norm->type != GGML_TYPE_F32 || norm->src[0]->type != GGML_TYPE_F32) return nullptr;
    if (w->ne[0] != norm->ne[0] || ggml_nrows(w) != 1 || w->nb[0] != sizeof(float)) return nullptr;
    if (!ggml_are_same_shape(norm, mul) || ggml_nrows(mul) != 1 || !ggml_is_contiguous(mul) || (norm->flags & GGML_TENSOR_FLAG_OUTPUT)) return nullptr;
    if (norm->src[0]->nb[0] != sizeof(float)) return nullptr;
    for (int j = i + 2; j < n; j++) {           // any other reader of the un-multiplied norm?
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s2 = 0; s2 < GGML_MAX_SRC; s2++) if (t->src[s2] == norm || (t->src[s2] && t->src[s2]->view_src == norm)) return nullptr;
    }
    return w;
}

// In such a launch every workgroup reads the norm input x and the weight w while workgroup 0 stores the MUL result and
// all of them store their share of the other outputs: x, w and the outputs must be pairwise apart (outs[0] is the MUL
// result), K within the prologue's reach and the three rows 16-byte aligned.
static bool norm_prologue_ok(const ggml_tensor * x, const ggml_tensor * w, int64_t K, int64_t max_k, const ggml_tensor * const * outs,
                             const size_t * out_bytes, int n_outs) {
    if (K > max_k || K % 32 != 0) return false;
    if (((uintptr_t)x->data | (uintptr_t)w->data | (uintptr_t)outs[0]->data) & 15) return false;
    const size_t xb = (size_t)K * sizeof(float);
    for (int o = 0; o < n_outs; o++) {
        if (ranges_overlap(outs[o]->data, out_bytes[o], x->data, xb) || ranges_overlap(outs[o]->data, out_bytes[o], w->data, xb)) return false;
        for (int q = 0; q < o; q++) if (ranges_overlap(outs[o]->data, out_bytes[o], outs[q]->data, out_bytes[q])) return false;
    }
    return true;
}

// Feed-forward input of a decode step in one launch.  build_ffn (src/llama-graph.cpp:470-640) spells it as
//     tmp = MUL_MAT(ffn_up, x)   cur = MUL_MAT(ffn_gate, x)   cur = GLU(cur, tmp)            (split SWIGLU / GEGLU / REGLU)
// When the two mat-vecs form a group (members[0..1], consecutive nodes, one token) and the GLU is the very next node,
// mi355_glu_matvec computes all three tensors (bit-identical values) plus the q8_1 form of the GLU result, which
// ffn_down then finds in the activation cache.  Every tensor of the chain is still written, so no use analysis is
// needed -- except that a GLU result placed exactly on top of one of its inputs (ggml-alloc's in-place reuse) simply
// replaces it.  Grids beyond one round of workgroups (m > 16384) stay on the grouped mat-vec, which is faster there.
static bool mi355_try_fuse_glu_matvec(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, const int * members, int norm_idx = -1) {
    const int n = ggml_graph_n_nodes(cgraph);
    const int i0 = members[0], i1 = members[1];
    if (i1 != i0 + 1 || i1 + 1 >= n || MI355_ENV_SET("GGML_MI355_NO_GLU_MATVEC")) return false;
    ggml_tensor * glu = ggml_graph_node(cgraph, i1 + 1);
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i0), * m1 = ggml_graph_node(cgraph, i1);
    if (glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * gate = glu->src[0], * up = glu->src[1];
    if (!((gate == m0 && up == m1) || (gate == m1 && up == m0))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    const ggml_tensor * src1 = m0->src[1];
    const int64_t K = src1->ne[0], M = gate->ne[0];
    if (src1->ne[1] != 1 || up->ne[0] != M || M % 32 != 0 || M > 16384 || ggml_nrows(glu) != 1 || glu->ne[0] != M) return false;
    if (!ggml_is_contiguous(glu) || ((uintptr_t)glu->data & 15)) return false;
    if ((size_t)K + (size_t)K / 32 * 8 + 16 + 256 > 64 * 1024) return false;
    // the GLU result may coincide with one of its inputs (then that input is not stored), but not straddle them,
    // nor touch the activations the mat-vecs were given
    float * gate_out = (float *)gate->data, * up_out = (float *)up->data;
    if (glu->data == gate->data) gate_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), gate->data, ggml_nbytes(gate))) return false;
    if (glu->data == up->data) up_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), up->data, ggml_nbytes(up))) return false;
    // The GLU result on the bytes of src1 (ggml-alloc does that once the two mat-vecs, src1's last readers, are behind it -- seen with
    // row-split weights): the kernel never reads src1's f32 form (the activations arrive quantized, or are normalised from the norm's
    // input in every workgroup), so the only conflict is the prologue's store of the norm result, which then goes to a row nobody reads.
    const bool glu_on_src1 = ranges_overlap(glu->data, ggml_nbytes(glu), src1->data, ggml_nbytes(src1));

    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
    const size_t oqb = align_up((size_t)mi355_act_q_bytes(M, 1), 256), odb = align_up((size_t)mi355_act_ds_bytes(M, 1), 256);
    mi355_glu_matvec_args a;
    memset(&a, 0, sizeof(a));
    char * ws = nullptr, * os;
    if (norm_idx >= 0) {
        // RMS_NORM (norm_idx) -> MUL by the norm weight (norm_idx + 1 == src1) in the prologue: every workgroup reads the
        // norm input, workgroup 0 stores src1 -- so src1, the norm input and the weight must be pairwise apart, and none
        // of the launch's other outputs may touch the two inputs
        const ggml_tensor * norm = ggml_graph_node(cgraph, norm_idx);
        const ggml_tensor * x = norm->src[0];
        const ggml_tensor * w = src1->src[0] == norm ? src1->src[1] : src1->src[0];
        // outputs that coincide (a GLU result in place on gate / up) were already reduced to one above
        ggml_tensor norm_dst = *src1;                      // where the prologue stores the norm result
        if (glu_on_src1) {
            norm_dst.data = ctx->get_scratch(6, (size_t)K * sizeof(float));
            if (ctx->scratch_size[6] < (size_t)K * sizeof(float)) return false;
        }
        const ggml_tensor * outs[4] = {&norm_dst, glu, nullptr, nullptr};
        size_t ob[4] = {ggml_nbytes(src1), ggml_nbytes(glu), 0, 0};
        int no = 2;
        if (gate_out) { outs[no] = gate; ob[no++] = ggml_nbytes(gate); }
        if (up_out)   { outs[no] = up;   ob[no++] = ggml_nbytes(up); }
        if (!norm_prologue_ok(x, w, K, 8192, outs, ob, no)) return false;
        a.norm_x = (const float *)x->data; a.norm_w = (const float *)w->data; a.norm_out = (float *)norm_dst.data;
        memcpy(&a.norm_eps, norm->op_params, sizeof(float));
        os = (char *)ctx->get_scratch(0, oqb + odb);      // scratch[0] is not read by this launch: the result goes straight in
    } else {
        ws = (char *)ctx->get_scratch(0, qb + db);
        if (!ctx->act_cached(src1)) {
            MI_CHECK(mi355_quantize_act((const float *)src1->data, (int8_t *)ws, (float *)(ws + qb), K, 1, 1, 1, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
            ctx->act_remember(src1);
        }
        os = (char *)ctx->get_scratch(2, oqb + odb);
    }
    a.type = (int32_t)gate->src[0]->type; a.glu_op = op;
    a.w_gate = gate->src[0]->data; a.w_up = up->src[0]->data;
    a.gate_row_stride = (int64_t)gate->src[0]->nb[1]; a.up_row_stride = (int64_t)up->src[0]->nb[1];
    a.m = M; a.k = K;
    if (ws) { a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb); }
    a.gate_out = gate_out; a.up_out = up_out; a.dst = (float *)glu->data;
    a.out_q = (int8_t *)os; a.out_ds = (float *)(os + oqb);
    if (tensor_is_split(gate->src[0])) {
        // row-split gate / up: both matrices are cut at the same rows (same shape, same split), a device computes rows [lo, hi) of
        // gate, up, the GLU result and its q8_1 form (128-row rounding: whole q8_1 blocks)
        const mi355_split_extra * eg = split_extra_of(gate->src[0]), * eu = split_extra_of(up->src[0]);
        if (!tensor_is_split(up->src[0])) return false;
        bool has[MI355_MAX_DEVICES] = {};
        const int ndev = mi355_device_count();
        for (int d = 0; d < ndev; d++) {
            if (eg->row_low[d] != eu->row_low[d] || eg->row_high[d] != eu->row_high[d] || (eg->row_low[d] % 32)) return false;
            const int64_t rows = eg->row_high[d] - eg->row_low[d];
            has[d] = rows > 0 && eg->data[d] && eu->data[d];
            if (has[d] && rows % 32) return false;
        }
        mi355_split_fan_out(ctx, has, [&](int d, hipStream_t st) {
            mi355_glu_matvec_args b = a;
            const int64_t lo = eg->row_low[d];
            b.w_gate = eg->data[d]; b.w_up = eu->data[d]; b.m = eg->row_high[d] - lo;
            if (a.gate_out) b.gate_out = a.gate_out + lo;
            if (a.up_out)   b.up_out   = a.up_out + lo;
            b.dst = a.dst + lo; b.out_q = a.out_q + lo; b.out_ds = a.out_ds + lo / 32 * 2;
            MI_CHECK(mi355_glu_matvec(&b, st));
        });
    } else {
        const int rc = mi355_glu_matvec(&a, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
        MI_CHECK(rc);
    }
    if (norm_idx < 0) ctx->swap_act_scratch();            // scratch[0] now holds the q8_1 form of the GLU result
    ctx->act_remember(glu);
    return true;
}

// bf16 rows of a prefill MUL_MAT's activations in scratch[0] (converted unless the producer already left them there)
uint16_t * prefill_act_bf16(mi355_backend_ctx * ctx, const ggml_tensor * src1) {
    const int64_t K = src1->ne[0], N = src1->ne[1];
    uint16_t * act = (uint16_t *)ctx->get_scratch(0, align_up((size_t)N * K * 2, 256));
    if (!ctx->act_cached(src1, 1)) {
        MI_CHECK(mi355_act_bf16((const float *)src1->data, act, K, N, 1, 1, src1->nb[1], src1->nb[2], src1->nb[3], ctx->stream));
        ctx->act_remember(src1, 1);
    }
    return act;
}

static bool prefill_mm_ok(const ggml_tensor * t);

// MUL_MAT (wo, ffn_down) -> ADD (residual) -> RMS_NORM -> MUL (norm weight) at nodes i .. i + 3 of a prefill batch: the GEMM leaves its
// K-slice planes and the norm launch adds them (see mi355_try_fuse_rms_norm_mul).  The product itself is never written, so the ADD must
// be its only reader.
static bool mi355_try_fuse_gemm_add_norm(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 3 >= n || done[i + 1] || done[i + 2] || done[i + 3] || MI355_ENV_SET("GGML_MI355_NO_GEMM_NORM")) return false;
    const ggml_tensor * mm = ggml_graph_node(cgraph, i);
    ggml_tensor * add = ggml_graph_node(cgraph, i + 1);
    const ggml_tensor * norm = ggml_graph_node(cgraph, i + 2);
    if (!prefill_mm_ok(mm) || mm->src[1]->ne[1] <= MI355_BF16_ROWS_ABOVE || mm->src[1]->ne[2] != 1 || mm->src[1]->ne[3] != 1) return false;
    if (add->op != GGML_OP_ADD || norm->op != GGML_OP_RMS_NORM || norm->src[0] != add || ggml_graph_node(cgraph, i + 3)->op != GGML_OP_MUL) return false;
    if ((add->src[0] == mm) == (add->src[1] == mm)) return false;             // exactly one addend is the product
    if (!ggml_is_contiguous(mm) || (mm->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    for (int j = 0; j < n; j++) {
        if (j == i + 1) continue;
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s = 0; s < GGML_MAX_SRC; s++) if (t->src[s] && (t->src[s] == mm || t->src[s]->view_src == mm)) return false;
    }
    return mi355_try_fuse_rms_norm_mul(ctx, cgraph, i + 2, add, mm);
}

// Feed-forward input of a prefill batch in one launch: MUL_MAT(ffn_gate) MUL_MAT(ffn_up) GLU at nodes i, i + 1, i + 2 (either order of
// the two products) -> mi355_glu_mul_mat_dq, whose tiles hold 64 rows of both matrices.  The two f32 products are NOT written, so they
// must have no other reader in this graph and must not be graph outputs (a scheduler eval callback that wants to see them gets them
// as one-node graphs, where this pattern cannot match).  The kernel library offers the fused form only where it has the bits of the
// three launches; everywhere else it answers MI355_E_UNSUPPORTED and the nodes run one by one.
static bool mi355_try_fuse_glu_gemm(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 2 >= n || done[i + 1] || done[i + 2] || MI355_ENV_SET("GGML_MI355_NO_GLU_GEMM") || MI355_ENV_SET("GGML_MI355_MMQ_INT8")) return false;
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i), * m1 = ggml_graph_node(cgraph, i + 1);
    ggml_tensor * glu = ggml_graph_node(cgraph, i + 2);
    if (m1->op != GGML_OP_MUL_MAT || glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * ga = glu->src[0], * gb = glu->src[1];       // act(ga) * gb; `swapped` only applies to the single-tensor form
    if (!((ga == m0 && gb == m1) || (ga == m1 && gb == m0))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    const ggml_tensor * src1 = m0->src[1], * wa = ga->src[0], * wb = gb->src[0];
    if (m1->src[1] != src1 || src1->type != GGML_TYPE_F32 || src1->ne[2] != 1 || src1->ne[3] != 1) return false;
    if (wa->type != wb->type || !ggml_are_same_shape(wa, wb) || wa->ne[2] != 1 || wa->ne[3] != 1 || tensor_is_split(wa) || tensor_is_split(wb)) return false;
    if (mi355_mul_mat_path(wa, src1) != MM_DQ || mi355_mul_mat_path(wb, src1) != MM_DQ) return false;
    const int64_t K = wa->ne[0], M = wa->ne[1], N = src1->ne[1];
    if (glu->ne[0] != M || ggml_nrows(glu) != N || !ggml_is_contiguous(glu) || ((uintptr_t)glu->data & 3)) return false;
    if ((m0->flags | m1->flags) & GGML_TENSOR_FLAG_OUTPUT) return false;
    for (int j = 0; j < n; j++) {                 // any other reader of the two products?
        if (j == i + 2) continue;
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        for (int s = 0; s < GGML_MAX_SRC; s++) {
            const ggml_tensor * r = t->src[s];
            if (r && (r == m0 || r == m1 || r->view_src == m0 || r->view_src == m1)) return false;
        }
    }
    if (ranges_overlap(glu->data, ggml_nbytes(glu), wa->data, ggml_nbytes(wa)) || ranges_overlap(glu->data, ggml_nbytes(glu), wb->data, ggml_nbytes(wb))) return false;

    const uint16_t * act = prefill_act_bf16(ctx, src1);
    uint16_t * hbf = (uint16_t *)ctx->get_scratch(2, align_up((size_t)N * M * 2, 256));
    const int rc = mi355_glu_mul_mat_dq((int)wa->type, op, wa->data, (int64_t)wa->nb[1], wb->data, (int64_t)wb->nb[1], act, K,
                                        (float *)glu->data, M, hbf, M, K, N, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;    // the activations stay cached for the separate launches
    MI_CHECK(rc);
    ctx->swap_act_scratch();                      // scratch[0] now holds the bf16 rows of the GLU result, for ffn_down
    ctx->act_remember(glu, 1);
    return true;
}

// RMS_NORM -> MUL(norm weight) -> MUL_MAT(ffn_up) MUL_MAT(ffn_gate) -> GLU of a decode token: five nodes, one launch
// (the norm runs in the prologue of mi355_glu_matvec).  The norm checks are those of mi355_try_fuse_rms_norm_mul.
static int mi355_collect_matvec_group(ggml_cgraph * cgraph, int i, const std::vector<char> & done, int * members);
static bool matvec_groupable(const ggml_tensor * node);
static bool mi355_try_fuse_norm_glu_matvec(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 4 >= n || MI355_ENV_SET("GGML_MI355_NO_NORM_GLU")) return false;
    for (int j = i; j <= i + 4; j++) if (done[j]) return false;
    if (!decode_norm_pattern(cgraph, i, done)) return false;
    const ggml_tensor * mul = ggml_graph_node(cgraph, i + 1);
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i + 2);
    if (!matvec_groupable(m0) || m0->src[1] != mul) return false;
    int members[MI355_MMVQ_MAX_GROUP];
    if (mi355_collect_matvec_group(cgraph, i + 2, done, members) != 2 || members[1] != i + 3) return false;
    return mi355_try_fuse_glu_matvec(ctx, cgraph, members, i);
}

// RMS_NORM -> MUL(weight) -> the group of decode mat-vecs that read it (ffn gate + up where the one-launch GLU form does not apply:
// Llama-3-70B's 28672 rows), the norm in the grouped launch's prologue.  *consumed = nodes taken after the norm (the MUL and the mat-vecs).
static bool mi355_try_fuse_norm_group(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done, int * consumed) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 2 >= n || MI355_ENV_SET("GGML_MI355_NO_NORM_GROUP")) return false;
    if (!decode_norm_pattern(cgraph, i, done)) return false;
    const ggml_tensor * norm = ggml_graph_node(cgraph, i);
    const ggml_tensor * mul = ggml_graph_node(cgraph, i + 1);
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i + 2);
    if (done[i + 2] || !matvec_groupable(m0) || m0->src[1] != mul || mul->ne[1] != 1) return false;
    int members[MI355_MMVQ_MAX_GROUP];
    const int cnt = mi355_collect_matvec_group(cgraph, i + 2, done, members);
    if (cnt < 2) return false;                       // a single mat-vec: the fused norm + quantize launch and the residual forms serve it
    for (int m = 0; m < cnt; m++) if (members[m] != i + 2 + m) return false;
    const ggml_tensor * x = norm->src[0];
    const ggml_tensor * w = mul->src[0] == norm ? mul->src[1] : mul->src[0];
    const int64_t K = mul->ne[0];
    const ggml_tensor * outs[1 + MI355_MMVQ_MAX_GROUP] = {mul};
    size_t ob[1 + MI355_MMVQ_MAX_GROUP] = {ggml_nbytes(mul)};
    for (int m = 0; m < cnt; m++) { outs[1 + m] = ggml_graph_node(cgraph, members[m]); ob[1 + m] = ggml_nbytes(outs[1 + m]); }
    if (!norm_prologue_ok(x, w, K, 16384, outs, ob, 1 + cnt)) return false;
    float eps;
    memcpy(&eps, norm->op_params, sizeof(float));
    mi355_mul_mat_group(ctx, cgraph, members, cnt, nullptr, nullptr, (const float *)x->data, (const float *)w->data, eps);
    ctx->act_invalidate_if_written(mul);
    *consumed = 1 + cnt;
    return true;
}

// Expert feed-forward input of one decode token in one launch (build_moe_ffn, src/llama-graph.cpp:660-760):
//     up = MUL_MAT_ID(ffn_up_exps, x, ids)   gate = MUL_MAT_ID(ffn_gate_exps, x, ids)   cur = GLU(gate, up)
// three consecutive nodes, one token, x broadcast over the slots -> mi355_glu_matvec with ids: all three tensors are
// written (bit-identical values) plus the q8_1 form of the GLU result, which MUL_MAT_ID(ffn_down_exps) finds in the
// activation cache.  A GLU result exactly on top of one of its inputs replaces it.
static bool mi355_try_fuse_moe_glu_matvec(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 2 >= n || done[i + 1] || done[i + 2] || MI355_ENV_SET("GGML_MI355_NO_GLU_MATVEC")) return false;
    const ggml_tensor * m0 = ggml_graph_node(cgraph, i), * m1 = ggml_graph_node(cgraph, i + 1);
    ggml_tensor * glu = ggml_graph_node(cgraph, i + 2);
    if (m1->op != GGML_OP_MUL_MAT_ID || glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * gate = glu->src[0], * up = glu->src[1];
    if (!((gate == m0 && up == m1) || (gate == m1 && up == m0))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    const ggml_tensor * b = m0->src[1], * ids = m0->src[2];
    if (m1->src[1] != b || m1->src[2] != ids) return false;
    const ggml_tensor * wg = gate->src[0], * wu = up->src[0];
    if (!is_weight_type(wg->type) || wg->type != wu->type || tensor_is_split(wg) || tensor_is_split(wu)) return false;
    const int64_t K = wg->ne[0], M = wg->ne[1], n_expert = wg->ne[2], n_used = ids->ne[0];
    if (wu->ne[0] != K || wu->ne[1] != M || wu->ne[2] != n_expert || wg->ne[3] != 1 || wu->ne[3] != 1) return false;
    if (b->ne[1] != 1 || b->ne[2] != 1 || b->ne[3] != 1 || ids->ne[1] != 1 || ids->type != GGML_TYPE_I32 || ids->nb[0] != sizeof(int32_t)) return false;
    if (n_used < 1 || n_used > 64 || M % 32 != 0 || K % ggml_blck_size(wg->type) != 0 || b->type != GGML_TYPE_F32 || b->nb[0] != sizeof(float)) return false;
    if ((size_t)K + (size_t)K / 32 * 8 + 16 + 256 > 64 * 1024) return false;
    const ggml_tensor * outs[3] = {gate, up, glu};
    for (const ggml_tensor * o : outs) {
        if (o->type != GGML_TYPE_F32 || o->ne[0] != M || o->ne[1] != n_used || ggml_nrows(o) != n_used || !ggml_is_contiguous(o)) return false;
        if (ranges_overlap(o->data, ggml_nbytes(o), b->data, ggml_nbytes(b)) || ranges_overlap(o->data, ggml_nbytes(o), ids->data, ggml_nbytes(ids))) return false;
    }
    if (ranges_overlap(gate->data, ggml_nbytes(gate), up->data, ggml_nbytes(up)) || ((uintptr_t)glu->data & 15)) return false;
    float * gate_out = (float *)gate->data, * up_out = (float *)up->data;
    if (glu->data == gate->data) gate_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), gate->data, ggml_nbytes(gate))) return false;
    if (glu->data == up->data) up_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), up->data, ggml_nbytes(up))) return false;

    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
    const size_t oqb = align_up((size_t)mi355_act_q_bytes(M, n_used), 256), odb = align_up((size_t)mi355_act_ds_bytes(M, n_used), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    if (!ctx->act_cached(b)) {
        MI_CHECK(mi355_quantize_act((const float *)b->data, (int8_t *)ws, (float *)(ws + qb), K, 1, 1, 1, b->nb[1], b->nb[2], b->nb[3], ctx->stream));
        ctx->act_remember(b);
    }
    char * os = (char *)ctx->get_scratch(2, oqb + odb);
    mi355_glu_matvec_args a;
    memset(&a, 0, sizeof(a));
    a.type = (int32_t)wg->type; a.glu_op = op;
    a.w_gate = wg->data; a.w_up = wu->data; a.gate_row_stride = (int64_t)wg->nb[1]; a.up_row_stride = (int64_t)wu->nb[1];
    a.m = M; a.k = K; a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb);
    a.gate_out = gate_out; a.up_out = up_out; a.dst = (float *)glu->data;
    a.out_q = (int8_t *)os; a.out_ds = (float *)(os + oqb);
    a.ids = (const int32_t *)ids->data; a.n_slots = (int32_t)n_used; a.n_expert = (int32_t)n_expert;
    a.gate_expert_stride = (int64_t)wg->nb[2]; a.up_expert_stride = (int64_t)wu->nb[2];
    const int rc = mi355_glu_matvec(&a, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    ctx->swap_act_scratch();            // scratch[0] now holds the q8_1 form of the GLU result ([n_used][M], the layout of a [M, n_used] src1)
    ctx->act_remember(glu);
    return true;
}

// Expert feed-forward output of one decode token in one launch (build_moe_ffn, src/llama-graph.cpp:760-795):
//     y = MUL_MAT_ID(ffn_down_exps, h, ids)  [router nodes, views]  y = MUL(y, weights)  out = ADD(view(y, 0), view(y, 1))
//     [l_out = ADD(out, residual)]
// for two used experts (Mixtral).  The nodes after the MUL_MAT_ID must be the next ones still to run (the router's
// GET_ROWS / SUM_ROWS / DIV that llama.cpp places in between were already computed by mi355_try_fuse_moe_router, so the
// weights exist); every tensor of the chain is written with bit-identical values.  Outputs may coincide exactly where
// the graph computes in place (MUL on y, l_out on out or on the residual) and must be apart otherwise.
static int next_active_node(ggml_cgraph * cgraph, int j, const std::vector<char> & done) {
    const int n = ggml_graph_n_nodes(cgraph);
    for (j++; j < n; j++) {
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j] || ggml_is_empty(t) || is_view_op(t)) continue;
        return j;
    }
    return -1;
}

static bool same_or_apart(const ggml_tensor * a, const ggml_tensor * b) {
    return a->data == b->data ? ggml_nbytes(a) == ggml_nbytes(b) : !ranges_overlap(a->data, ggml_nbytes(a), b->data, ggml_nbytes(b));
}

static bool mi355_try_fuse_moe_down(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    if (MI355_ENV_SET("GGML_MI355_NO_MOE_DOWN")) return false;
    const ggml_tensor * down = ggml_graph_node(cgraph, i);
    const ggml_tensor * as = down->src[0], * b = down->src[1], * ids = down->src[2];
    const int64_t K = as->ne[0], M = as->ne[1], n_expert = as->ne[2], n_used = ids->ne[0];
    if (n_used != 2 || !is_weight_type(as->type) || tensor_is_split(as) || as->ne[3] != 1) return false;
    if (ids->ne[1] != 1 || ids->type != GGML_TYPE_I32 || ids->nb[0] != sizeof(int32_t)) return false;
    if (b->type != GGML_TYPE_F32 || b->ne[0] != K || b->ne[1] != n_used || b->ne[2] != 1 || b->ne[3] != 1 || !ggml_is_contiguous(b)) return false;
    if (down->type != GGML_TYPE_F32 || down->ne[0] != M || down->ne[1] != n_used || ggml_nrows(down) != n_used || !ggml_is_contiguous(down)) return false;
    if (K % ggml_blck_size(as->type) != 0 || K % 32 != 0) return false;
    const int jm = next_active_node(cgraph, i, done);
    if (jm < 0) return false;
    const ggml_tensor * mul = ggml_graph_node(cgraph, jm);
    if (mul->op != GGML_OP_MUL || mul->src[0] != down || mul->type != GGML_TYPE_F32 || !ggml_are_same_shape(mul, down) || !ggml_is_contiguous(mul)) return false;
    const ggml_tensor * w = mul->src[1];
    if (w->type != GGML_TYPE_F32 || w->ne[0] != 1 || w->ne[1] != n_used || ggml_nelements(w) != n_used || w->nb[1] != sizeof(float)) return false;
    const int ja = next_active_node(cgraph, jm, done);
    if (ja < 0) return false;
    const ggml_tensor * add = ggml_graph_node(cgraph, ja);
    if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32 || add->ne[0] != M || ggml_nrows(add) != 1 || !ggml_is_contiguous(add)) return false;
    for (int s2 = 0; s2 < 2; s2++) {
        const ggml_tensor * v = add->src[s2];
        if (!v || view_root(v) != mul || v->type != GGML_TYPE_F32 || v->ne[0] != M || ggml_nrows(v) != 1 || v->nb[0] != sizeof(float)) return false;
        if (v->data != (const char *)mul->data + s2 * mul->nb[1]) return false;
    }
    // optional residual ADD right behind
    const ggml_tensor * radd = nullptr, * r = nullptr;
    const int jr = next_active_node(cgraph, ja, done);
    if (jr >= 0) {
        const ggml_tensor * t = ggml_graph_node(cgraph, jr);
        if (t->op == GGML_OP_ADD && t->type == GGML_TYPE_F32 && (t->src[0] == add || t->src[1] == add)) {
            const ggml_tensor * o = t->src[0] == add ? t->src[1] : t->src[0];
            if (o != add && o->type == GGML_TYPE_F32 && ggml_are_same_shape(o, add) && ggml_are_same_shape(t, add) && ggml_is_contiguous(o) &&
                ggml_is_contiguous(t)) { radd = t; r = o; }
        }
    }
    // placement
    if (!same_or_apart(mul, down)) return false;
    const ggml_tensor * big[2] = {down, mul};
    const ggml_tensor * small[3] = {add, radd, r};
    for (const ggml_tensor * sm : small) {
        if (!sm) continue;
        for (const ggml_tensor * bg : big) if (ranges_overlap(sm->data, ggml_nbytes(sm), bg->data, ggml_nbytes(bg))) return false;
    }
    if (radd && (!same_or_apart(radd, add) || !same_or_apart(radd, r) || !same_or_apart(add, r))) return false;
    if (radd && add->data == r->data) return false;      // the sum would overwrite the residual before it is read back
    const ggml_tensor * outs[4] = {down, mul, add, radd};
    for (const ggml_tensor * o : outs) {
        if (!o) continue;
        if (ranges_overlap(o->data, ggml_nbytes(o), ids->data, ggml_nbytes(ids)) || ranges_overlap(o->data, ggml_nbytes(o), w->data, ggml_nbytes(w)) ||
            ranges_overlap(o->data, ggml_nbytes(o), b->data, ggml_nbytes(b))) return false;
    }
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, n_used), 256), db = align_up((size_t)mi355_act_ds_bytes(K, n_used), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    if (!ctx->act_cached(b)) {
        MI_CHECK(mi355_quantize_act((const float *)b->data, (int8_t *)ws, (float *)(ws + qb), K, n_used, 1, 1, b->nb[1], b->nb[2], b->nb[3], ctx->stream));
        ctx->act_remember(b);
    }
    mi355_moe_down_args a;
    memset(&a, 0, sizeof(a));
    a.type = (int32_t)as->type; a.n_slots = (int32_t)n_used; a.n_expert = (int32_t)n_expert;
    a.W = as->data; a.w_row_stride = (int64_t)as->nb[1]; a.w_expert_stride = (int64_t)as->nb[2];
    a.m = M; a.k = K; a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb);
    a.ids = (const int32_t *)ids->data; a.weights = (const float *)w->data;
    a.out = mul->data == down->data ? nullptr : (float *)down->data;
    a.weighted_out = (float *)mul->data; a.sum_out = (float *)add->data;
    if (radd) { a.residual = (const float *)r->data; a.residual_out = (float *)radd->data; }
    const int rc = mi355_moe_down_matvec(&a, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    done[jm] = 1; done[ja] = 1;
    if (radd) done[jr] = 1;
    for (const ggml_tensor * o : outs) if (o) ctx->act_invalidate_if_written(o);
    return true;
}

// The same tail for a batch of tokens, when ffn_moe_down itself ran as a grouped GEMM: MUL(experts, weights) and the ADD of its two
// slot views as one pass (mi355_moe_combine2_f32); both tensors are written, so no use analysis is needed.
static bool mi355_try_fuse_moe_combine(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    if (MI355_ENV_SET("GGML_MI355_NO_MOE_COMBINE")) return false;
    const ggml_tensor * mul = ggml_graph_node(cgraph, i);
    const ggml_tensor * e = mul->src[0], * w = mul->src[1];
    if (!e || !w || mul->type != GGML_TYPE_F32 || e->type != GGML_TYPE_F32 || w->type != GGML_TYPE_F32) return false;
    const int64_t M = e->ne[0], T = e->ne[2];
    if (e->ne[1] != 2 || e->ne[3] != 1 || T < 1 || T > 65535 || M % 4 != 0 || !ggml_are_same_shape(mul, e) || !ggml_is_contiguous(e) || !ggml_is_contiguous(mul)) return false;
    if (w->ne[0] != 1 || w->ne[1] != 2 || w->ne[2] != T || w->ne[3] != 1) return false;
    const int ja = next_active_node(cgraph, i, done);
    if (ja < 0) return false;
    const ggml_tensor * add = ggml_graph_node(cgraph, ja);
    if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32 || add->ne[0] != M || add->ne[1] != T || ggml_nrows(add) != T || !ggml_is_contiguous(add)) return false;
    for (int s2 = 0; s2 < 2; s2++) {
        const ggml_tensor * v = add->src[s2];
        if (!v || view_root(v) != mul || v->type != GGML_TYPE_F32 || v->ne[0] != M || v->ne[1] != T || ggml_nrows(v) != T || v->nb[0] != sizeof(float) ||
            v->nb[1] != mul->nb[2]) return false;
        if (v->data != (const char *)mul->data + s2 * mul->nb[1]) return false;
    }
    if (!same_or_apart(mul, e)) return false;
    if (ranges_overlap(add->data, ggml_nbytes(add), e->data, ggml_nbytes(e)) || ranges_overlap(add->data, ggml_nbytes(add), mul->data, ggml_nbytes(mul))) return false;
    if (ranges_overlap(w->data, ggml_nbytes(w), mul->data, ggml_nbytes(mul)) || ranges_overlap(w->data, ggml_nbytes(w), add->data, ggml_nbytes(add))) return false;
    const int rc = mi355_moe_combine2_f32((const float *)e->data, (const float *)w->data, (int64_t)w->nb[2], (int64_t)w->nb[1], (float *)mul->data,
                                          (float *)add->data, M, T, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    done[ja] = 1;
    ctx->act_invalidate_if_written(mul);
    ctx->act_invalidate_if_written(add);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Fused attention input of a decode step.  llm_build_llama (src/llama-model.cpp:4898-5052) spells it as
//     Qcur = MUL_MAT(wq, x)  RESHAPE  ROPE      Kcur = MUL_MAT(wk, x)  RESHAPE  ROPE      Vcur = MUL_MAT(wv, x)  RESHAPE
//     CPY(rope(Kcur) -> view of cache_k)        CPY(Vcur -> view of cache_v)
// (Kcur and Vcur even reuse Qcur's memory, so these mat-vecs cannot simply be grouped.)  When exactly this shape
// is found for a single token, with f16/f32 cache rows and no other reader of the intermediates, one launch of
// mi355_qkv_rope_matvec produces rope(Qcur) and both cache rows; Qcur, Kcur, Vcur and rope(Kcur) are never stored.
// ------------------------------------------------------------------------------------------------

// MoE router of a decode-sized batch in one launch: MUL_MAT(f32 gate_inp, x) SOFT_MAX ARGSORT(desc) [top-k view] GET_ROWS
// SUM_ROWS DIV (build_moe_ffn, src/llama-graph.cpp:660-720) -> mi355_moe_router_f32.  Every tensor of the chain is written,
// with bit-identical values, so no use analysis is needed; the outputs only have to be laid out so that one
// workgroup's (= one token's) stores cannot land in another token's rows of a different tensor.
// norm_idx >= 0: the RMS_NORM at that node and its MUL (norm_idx + 1, the router MUL_MAT's src1, one token) run in the same launch
// (mi355_moe_router_norm_f32: the normalised row and its q8_1 form are written as the norm launch would have) -- the caller has
// checked decode_norm_pattern
static bool mi355_try_fuse_moe_router(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done, int norm_idx = -1) {
    const int n = ggml_graph_n_nodes(cgraph);
    ggml_tensor * mm = ggml_graph_node(cgraph, i);
    const ggml_tensor * g = mm->src[0], * x = mm->src[1];
    if (g->type != GGML_TYPE_F32 || x->type != GGML_TYPE_F32 || tensor_is_split(g)) return false;
    const int64_t K = g->ne[0], E = g->ne[1], T = x->ne[1];
    if (g->ne[2] != 1 || g->ne[3] != 1 || x->ne[2] != 1 || x->ne[3] != 1 || E < 1 || E > 256 || T < 1 || T > 64) return false;
    if (g->nb[0] != sizeof(float) || x->nb[0] != sizeof(float) || !ggml_is_contiguous(mm)) return false;
    // The chain's nodes in graph order; llama.cpp's graph puts the three expert MUL_MAT_IDs between the top-k and the
    // weight gathering, so other nodes may sit in between -- they are remembered and checked below.
    static const ggml_op seq[5] = {GGML_OP_SOFT_MAX, GGML_OP_ARGSORT, GGML_OP_GET_ROWS, GGML_OP_SUM_ROWS, GGML_OP_DIV};
    ggml_tensor * t5[5];
    int idx[5], cnt = 0;
    std::vector<const ggml_tensor *> between;
    for (int j = i + 1; j < n && j < i + 40 && cnt < 5; j++) {
        ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j]) return false;
        if (is_view_op(t) || ggml_is_empty(t)) continue;
        const ggml_tensor * prev = cnt == 0 ? mm : t5[cnt - 1];
        const bool link = t->op == seq[cnt] && (cnt == 2 ? view_root(t->src[0]) == t5[0]                  // GET_ROWS reads the probabilities
                                               : cnt == 3 ? view_root(t->src[0]) == t5[2]                  // SUM_ROWS / DIV read the gathered weights
                                               : cnt == 4 ? view_root(t->src[0]) == t5[2] : t->src[0] == prev);
        if (link) { t5[cnt] = t; idx[cnt++] = j; }
        else {
            if (cnt < 2) return false;      // MUL_MAT, SOFT_MAX, ARGSORT are consecutive
            between.push_back(t);
        }
    }
    if (cnt < 5) return false;
    ggml_tensor * sm = t5[0], * as = t5[1], * gr = t5[2], * sr = t5[3], * dv = t5[4];
    float scale, max_bias;
    memcpy(&scale, (const float *)sm->op_params + 0, 4); memcpy(&max_bias, (const float *)sm->op_params + 1, 4);
    if (sm->src[0] != mm || sm->src[1] || scale != 1.0f || max_bias != 0.0f || !ggml_is_contiguous(sm) || sm->type != GGML_TYPE_F32) return false;
    if (as->src[0] != sm || ((const int32_t *)as->op_params)[0] != (int32_t)GGML_SORT_ORDER_DESC || !ggml_is_contiguous(as)) return false;
    const ggml_tensor * sel = gr->src[1];      // top-k: a view of the first n_used entries of every argsort row
    const int64_t U = sel->ne[0];
    if (view_root(gr->src[0]) != sm || view_root(sel) != as || sel->data != as->data || sel->ne[1] != T || sel->ne[2] != 1 || sel->ne[3] != 1 ||
        sel->nb[0] != sizeof(int32_t) || sel->nb[1] != (size_t)E * sizeof(int32_t) || U < 1 || U > E) return false;
    if (gr->src[0]->ne[0] != 1 || gr->src[0]->ne[1] != E || gr->src[0]->ne[2] != T || gr->src[0]->data != sm->data || !ggml_is_contiguous(gr->src[0])) return false;
    if (gr->type != GGML_TYPE_F32 || !ggml_is_contiguous(gr) || ggml_nelements(gr) != U * T) return false;
    if (view_root(sr->src[0]) != gr || sr->src[0]->data != gr->data || sr->src[0]->ne[0] != U || sr->src[0]->ne[1] != T || !ggml_is_contiguous(sr->src[0]) ||
        !ggml_is_contiguous(sr) || ggml_nelements(sr) != T) return false;
    if (view_root(dv->src[0]) != gr || dv->src[0]->data != gr->data || dv->src[1] != sr || dv->ne[0] != U || dv->ne[1] != T || !ggml_is_contiguous(dv) ||
        !ggml_is_contiguous(dv->src[0])) return false;
    // GET_ROWS, SUM_ROWS and DIV run early: no node in between may read or write the bytes they produce
    for (const ggml_tensor * t : between) {
        for (const ggml_tensor * o : {(const ggml_tensor *)gr, (const ggml_tensor *)sr, (const ggml_tensor *)dv}) {
            if (ranges_overlap(t->data, ggml_nbytes(t), o->data, ggml_nbytes(o))) return false;
            for (int k = 0; k < GGML_MAX_SRC; k++)
                if (t->src[k] && ranges_overlap(t->src[k]->data, ggml_nbytes(t->src[k]), o->data, ggml_nbytes(o))) return false;
        }
        // ... and none may overwrite the probabilities or the order before the (now earlier) gathering would have read them
        if (ranges_overlap(t->data, ggml_nbytes(t), sm->data, ggml_nbytes(sm)) || ranges_overlap(t->data, ggml_nbytes(t), as->data, ggml_nbytes(as))) return false;
    }
    // Several tokens = several workgroups in flight.  The logits are dead once the probabilities exist, the probabilities once the
    // weights are gathered, the weight sums once the division is done -- so ggml-alloc hands the bytes of the first two to the third,
    // and one token's sum would land in another token's probabilities while those are still being read.  A tensor of the chain that
    // nobody outside the chain reads goes to scratch instead, and its node's bytes are left alone.
    const ggml_tensor * outs[6] = {mm, sm, as, gr, sr, dv};
    bool aside[6] = {false, false, false, false, false, false};
    if (T > 1) {
        auto inside_only = [&](const ggml_tensor * t, std::initializer_list<const ggml_tensor *> readers) {
            if (t->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
            for (int j = 0; j < n; j++) {
                const ggml_tensor * r = ggml_graph_node(cgraph, j);
                bool known = false;
                for (const ggml_tensor * k : readers) known |= r == k;
                if (known) continue;
                for (int k = 0; k < GGML_MAX_SRC; k++) if (r->src[k] && view_root(r->src[k]) == t) return false;
            }
            return true;
        };
        aside[0] = inside_only(mm, {sm});
        aside[1] = inside_only(sm, {as, gr});
        aside[4] = inside_only(sr, {dv});
    }
    // outputs: pairwise disjoint, or exactly the same rows (in-place reuse by ggml-alloc); inputs untouched
    const size_t rowb[6] = {(size_t)E * 4, (size_t)E * 4, (size_t)E * 4, (size_t)U * 4, 4, (size_t)U * 4};
    for (int p = 0; p < 6; p++) {
        if (aside[p]) continue;
        if (ranges_overlap(outs[p]->data, ggml_nbytes(outs[p]), x->data, ggml_nbytes(x)) ||
            ranges_overlap(outs[p]->data, ggml_nbytes(outs[p]), g->data, ggml_nbytes(g))) return false;
        for (int q = p + 1; q < 6; q++) {
            if (aside[q] || !ranges_overlap(outs[p]->data, ggml_nbytes(outs[p]), outs[q]->data, ggml_nbytes(outs[q]))) continue;
            if (T > 1 && (outs[p]->data != outs[q]->data || rowb[p] != rowb[q])) return false;
            // one token = one workgroup: logits, probs and order are written in phases separated by barriers, so a later
            // tensor may reuse an earlier one's memory freely; the last three are written together by one wave
            if (T == 1 && p >= 3 && !(p == 3 && q == 5 && outs[p]->data == outs[q]->data)) return false;
        }
    }
    float * side = aside[0] || aside[1] || aside[4] ? (float *)ctx->get_scratch(6, (size_t)(2 * E + 1) * T * sizeof(float)) : nullptr;
    if (norm_idx >= 0) {
        const ggml_tensor * norm = ggml_graph_node(cgraph, norm_idx), * mul = ggml_graph_node(cgraph, norm_idx + 1);
        const ggml_tensor * xr = norm->src[0], * nw = mul->src[0] == norm ? mul->src[1] : mul->src[0];
        if (x != mul || T != 1 || K > 4096 || K % 32 != 0 || MI355_ENV_SET("GGML_MI355_NO_NORM_ROUTER")) return false;
        // the row is read by this one workgroup only, but it must not be any of the launch's outputs, nor the weight
        const ggml_tensor * nouts[7] = {mul, mm, sm, as, gr, sr, dv};
        size_t nob[7];
        for (int p = 0; p < 7; p++) nob[p] = ggml_nbytes(nouts[p]);
        if (((uintptr_t)xr->data | (uintptr_t)nw->data | (uintptr_t)mul->data) & 15) return false;
        for (int p = 0; p < 7; p++) {
            if (ranges_overlap(nouts[p]->data, nob[p], xr->data, (size_t)K * 4) || ranges_overlap(nouts[p]->data, nob[p], nw->data, (size_t)K * 4)) return false;
            if (p > 0 && ranges_overlap(nouts[p]->data, nob[p], mul->data, nob[0])) return false;
        }
        float eps;
        memcpy(&eps, norm->op_params, sizeof(float));
        const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
        char * ws = (char *)ctx->get_scratch(0, qb + db);
        const int rc = mi355_moe_router_norm_f32((const float *)xr->data, (int64_t)xr->nb[1], (const float *)nw->data, eps, (float *)mul->data, (int8_t *)ws,
                                                 (float *)(ws + qb), (const float *)g->data, (int64_t)g->nb[1], K, E, U, T, (float *)mm->data, (float *)sm->data,
                                                 (int32_t *)as->data, (float *)gr->data, (float *)sr->data, (float *)dv->data, ctx->stream);
        if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN) return false;
        MI_CHECK(rc);
        for (int m = 0; m < 5; m++) done[idx[m]] = 1;
        for (int p = 0; p < 6; p++) ctx->act_invalidate_if_written(outs[p]);
        ctx->act_invalidate_if_written(mul);
        ctx->act_remember(mul);
        return true;
    }
    MI_CHECK(mi355_moe_router_f32((const float *)x->data, (int64_t)x->nb[1], (const float *)g->data, (int64_t)g->nb[1], K, E, U, T,
                                  aside[0] ? side : (float *)mm->data, aside[1] ? side + E * T : (float *)sm->data, (int32_t *)as->data, (float *)gr->data,
                                  aside[4] ? side + 2 * E * T : (float *)sr->data, (float *)dv->data, ctx->stream));
    for (int m = 0; m < 5; m++) done[idx[m]] = 1;
    for (int p = 0; p < 6; p++) if (!aside[p]) ctx->act_invalidate_if_written(outs[p]);
    return true;
}

// The attention input of a llama-style layer as build_attn spells it (src/llama-graph.cpp, llama-kv-cache-unified.cpp:817-891):
//     q = MUL_MAT(wq, x)  q = ROPE(q)  k = MUL_MAT(wk, x)  k = ROPE(k)  v = MUL_MAT(wv, x)  CPY(k -> cache rows)  CPY(v -> cache rows)
// starting at node i, for N = x->ne[1] tokens whose cache rows lie one after the other.  Only q (rotated) and the two cache views
// are materialised by the fused launches, so nobody else may read the four tensors in between.
struct qkv_nodes {
    ggml_tensor * mq, * rq, * mk, * rk, * mv, * ck, * cv;
    int idx[7];
    bool kv_q8;      // a quantized cache (q8_0 or q4_0, kv_qt): K / V leave the fused launch as f32 rows and one more launch quantizes them
    int kv_qt;
};

static bool qkv_match(ggml_cgraph * cgraph, int i, const std::vector<char> & done, bool (*mm_ok)(const ggml_tensor *), qkv_nodes & q) {
    const int n = ggml_graph_n_nodes(cgraph);
    ggml_tensor * mq = ggml_graph_node(cgraph, i);
    const ggml_tensor * x = mq->src[1];
    if (x->ne[2] != 1 || x->ne[3] != 1) return false;
    const int64_t N = x->ne[1];
    ggml_tensor * rq = nullptr, * mk = nullptr, * rk = nullptr, * mv = nullptr, * ck = nullptr, * cv = nullptr;
    int * idx = q.idx, last = -1;
    idx[0] = i;
    for (int m = 1; m < 7; m++) idx[m] = -1;
    const int horizon = i + 24 < n ? i + 24 : n;
    for (int j = i + 1; j < horizon; j++) {
        ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j]) return false;
        if (is_view_op(t) || ggml_is_empty(t)) continue;
        if (t->op == GGML_OP_ROPE) {
            const ggml_tensor * r = view_root(t->src[0]);
            if (r == mq && !rq)            { rq = t; idx[1] = j; }
            else if (mk && r == mk && !rk) { rk = t; idx[3] = j; }
            else return false;
        } else if (t->op == GGML_OP_MUL_MAT) {
            if (!mm_ok(t) || t->src[1] != x) return false;
            if (!mk)      { mk = t; idx[2] = j; }
            else if (!mv) { mv = t; idx[4] = j; }
            else return false;
        } else if (t->op == GGML_OP_CPY) {
            const ggml_tensor * r = view_root(t->src[0]);
            if (rk && r == rk && !ck)      { ck = t; idx[5] = j; }
            else if (mv && r == mv && !cv) { cv = t; idx[6] = j; }
            else return false;
        } else {
            return false;
        }
        if (rq && mk && rk && mv && ck && cv) { last = j; break; }
    }
    if (last < 0) return false;
    // shapes and types
    const ggml_tensor * wq = mq->src[0], * wk = mk->src[0], * wv = mv->src[0];
    if (wq->ne[0] != wk->ne[0] || wq->ne[0] != wv->ne[0]) return false;
    const int64_t Mq = wq->ne[1], Mk = wk->ne[1], Mv = wv->ne[1], hd = rq->ne[0];
    if (Mq % 8 || Mk % 8 || Mv % 8 || hd != rk->ne[0] || hd <= 0 || Mq % hd || Mk % hd) return false;
    for (ggml_tensor * r : {rq, rk}) {
        if (r->type != GGML_TYPE_F32 || !ggml_is_contiguous(r) || !ggml_is_contiguous(r->src[0]) || r->ne[2] != N || r->ne[3] != 1) return false;
        if (r->src[1] != rq->src[1] || r->src[2] != rq->src[2]) return false;
        if (memcmp(r->op_params, rq->op_params, 11 * sizeof(int32_t)) != 0) return false;
        if (r->src[1]->type != GGML_TYPE_I32 || (r->src[2] && r->src[2]->type != GGML_TYPE_F32)) return false;
    }
    if (rq->src[0]->data != mq->data || rk->src[0]->data != mk->data) return false;
    if (ggml_nelements(rq) != Mq * N || ggml_nelements(rk) != Mk * N) return false;
    const int32_t * rp = (const int32_t *)rq->op_params;
    if (rp[2] != 0 && rp[2] != 2) return false;
    q.kv_q8 = ck->type == cv->type && (ck->type == GGML_TYPE_Q8_0 || ck->type == GGML_TYPE_Q4_0);
    q.kv_qt = (int)ck->type;
    for (ggml_tensor * c : {ck, cv}) {
        if ((c->type != GGML_TYPE_F16 && c->type != GGML_TYPE_F32 && !q.kv_q8) || !ggml_is_contiguous(c) || !ggml_is_contiguous(c->src[0])) return false;
        if (c->src[0]->type != GGML_TYPE_F32) return false;
    }
    if (q.kv_q8 && (Mk % 32 || Mv % 32)) return false;
    if (ck->src[0]->data != rk->data || cv->src[0]->data != mv->data || ggml_nelements(ck) != Mk * N || ggml_nelements(cv) != Mv * N) return false;
    // nobody else may read the tensors that will not be materialised
    const ggml_tensor * hidden[4] = {mq, mk, mv, rk};
    for (const ggml_tensor * h : hidden) if (h->flags & GGML_TENSOR_FLAG_OUTPUT) return false;
    for (int j = i + 1; j < n; j++) {
        const ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (is_view_op(t)) { if (t->flags & GGML_TENSOR_FLAG_OUTPUT) for (const ggml_tensor * h : hidden) if (view_root(t) == h) return false; continue; }
        bool in_pattern = false;
        for (int m = 1; m < 7; m++) if (idx[m] == j) in_pattern = true;
        if (in_pattern) continue;
        for (int sidx = 0; sidx < GGML_MAX_SRC; sidx++) {
            if (!t->src[sidx]) continue;
            for (const ggml_tensor * h : hidden) if (view_root(t->src[sidx]) == h) return false;
        }
    }
    q.mq = mq; q.rq = rq; q.mk = mk; q.rk = rk; q.mv = mv; q.ck = ck; q.cv = cv;
    return true;
}

static void qkv_rope_params(const ggml_tensor * rq, int32_t * n_dims, int32_t * mode, int32_t * n_ctx_orig, float * f6) {
    const int32_t * rp = (const int32_t *)rq->op_params;
    *n_dims = rp[1]; *mode = rp[2]; *n_ctx_orig = rp[4];
    memcpy(f6, rp + 5, 6 * sizeof(float));       // freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow
}

// Prefill form: three GEMMs that leave their K-slice planes in scratch (mi355_mul_mat_dq_parts) and ONE launch that adds the planes,
// rotates q and k and stores k and v into the cache rows (mi355_qkv_finish) -- 4 launches for the 7 nodes' 10 (3 GEMMs, up to 3 slice
// sums or memsets, 2 ROPE, 2 CPY), and q / k / v cross HBM once less.  Bit-identical to the nodes run one by one.
static bool prefill_mm_ok(const ggml_tensor * t) {
    const ggml_tensor * w = t->src[0], * x = t->src[1];
    return t->op == GGML_OP_MUL_MAT && t->type == GGML_TYPE_F32 && x->type == GGML_TYPE_F32 && w->ne[2] == 1 && w->ne[3] == 1 && !tensor_is_split(w) &&
           mi355_mul_mat_path(w, x) == MM_DQ;
}

static bool mi355_try_fuse_qkv_gemm(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    ggml_tensor * mq0 = ggml_graph_node(cgraph, i);
    if (!prefill_mm_ok(mq0) || mq0->src[1]->ne[1] <= 1 || MI355_ENV_SET("GGML_MI355_NO_QKV_GEMM") || MI355_ENV_SET("GGML_MI355_MMQ_INT8")) return false;
    qkv_nodes q;
    if (!qkv_match(cgraph, i, done, prefill_mm_ok, q)) return false;
    const ggml_tensor * x = q.mq->src[1];
    const ggml_tensor * w[3] = {q.mq->src[0], q.mk->src[0], q.mv->src[0]};
    const int64_t K = x->ne[0], N = x->ne[1], hd = q.rq->ne[0];
    if (w[2]->ne[1] % 4) return false;
    // the three plane sets
    int64_t pb[3], off[3], total = 0;
    for (int m = 0; m < 3; m++) {
        pb[m] = mi355_mul_mat_dq_parts_bytes(w[m]->ne[1], w[m]->ne[1], K, N);
        if (pb[m] <= 0) return false;
        off[m] = total; total += (int64_t)align_up((size_t)pb[m], 256);
    }
    char * parts = (char *)ctx->get_scratch(3, (size_t)total);
    if (ctx->scratch_size[3] < (size_t)total) return false;
    // outputs must not alias one another or the positions (they are written by one launch, per element)
    const ggml_tensor * outs[3] = {q.rq, q.ck, q.cv};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < a; b++) if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), outs[b]->data, ggml_nbytes(outs[b]))) return false;
        if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), q.rq->src[1]->data, ggml_nbytes(q.rq->src[1]))) return false;
    }
    // a q8_0 cache: k and v leave the finish launch as f32 rows in scratch and one more launch quantizes both into the cache rows
    // (mi355_kv_store_q8_0, the arithmetic of the CPY f32 -> q8_0 it replaces), as on the decode path
    const int64_t Mk = w[1]->ne[1], Mv = w[2]->ne[1];
    float * kv_tmp = nullptr;
    if (q.kv_q8) {
        const size_t need = (size_t)N * (Mk + Mv) * sizeof(float);
        kv_tmp = (float *)ctx->get_scratch(4, need);
        if (ctx->scratch_size[4] < need || N * (Mk > Mv ? Mk : Mv) > 0x7fffffff) return false;
    }
    const uint16_t * act = prefill_act_bf16(ctx, x);
    mi355_qkv_finish_args f;
    memset(&f, 0, sizeof(f));
    int32_t kz[3] = {0, 0, 0};
    bool paired = false;
    for (int m = 0; m < 3; m++) {
        if (m == 2 && paired) break;
        if (m == 1 && Mk == Mv && pb[1] == pb[2] && !MI355_ENV_SET("GGML_MI355_NO_QKV_PAIR")) {
            // wk and wv (M = 1024 each in a GQA model) in one launch: each alone fills half the chip
            int32_t kzp = 0;
            const int rcp = mi355_mul_mat_dq_parts_pair((int)w[1]->type, w[1]->data, (int64_t)w[1]->nb[1], (float *)(parts + off[1]),
                                                        (int)w[2]->type, w[2]->data, (int64_t)w[2]->nb[1], (float *)(parts + off[2]), pb[1],
                                                        act, K, Mk, K, N, &kzp, ctx->stream);
            if (rcp == MI355_OK) { kz[1] = kz[2] = kzp; paired = true; continue; }
            if (rcp != MI355_E_UNSUPPORTED && rcp != MI355_E_ALIGN) MI_CHECK(rcp);
        }
        const int rc = mi355_mul_mat_dq_parts((int)w[m]->type, w[m]->data, (int64_t)w[m]->nb[1], act, K, (float *)(parts + off[m]), pb[m], w[m]->ne[1], w[m]->ne[1], K, N,
                                              &kz[m], ctx->stream);
        // nothing of the pattern has been written yet when the first product is refused; a later refusal cannot happen for a type
        // and K the first one took
        if (m == 0 && (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN)) return false;
        MI_CHECK(rc);
    }
    f.q_parts = (const float *)(parts + off[0]); f.k_parts = (const float *)(parts + off[1]); f.v_parts = (const float *)(parts + off[2]);
    f.kz_q = kz[0]; f.kz_k = kz[1]; f.kz_v = kz[2];
    f.m_q = w[0]->ne[1]; f.m_k = w[1]->ne[1]; f.m_v = w[2]->ne[1]; f.n = N;
    f.q_out = (float *)q.rq->data; f.k_out = q.ck->data; f.v_out = q.cv->data;
    f.k_out_type = (int32_t)q.ck->type; f.v_out_type = (int32_t)q.cv->type;
    if (kv_tmp) { f.k_out = kv_tmp; f.v_out = kv_tmp + N * Mk; f.k_out_type = f.v_out_type = (int32_t)GGML_TYPE_F32; }
    f.pos = (const int32_t *)q.rq->src[1]->data; f.freq_factors = q.rq->src[2] ? (const float *)q.rq->src[2]->data : nullptr;
    f.head_dim = (int32_t)hd;
    float f6[6];
    qkv_rope_params(q.rq, &f.n_dims, &f.mode, &f.n_ctx_orig, f6);
    f.freq_base = f6[0]; f.freq_scale = f6[1]; f.ext_factor = f6[2]; f.attn_factor = f6[3]; f.beta_fast = f6[4]; f.beta_slow = f6[5];
    MI_CHECK(mi355_qkv_finish(&f, ctx->stream));
    if (kv_tmp) MI_CHECK(mi355_kv_store_q(q.kv_qt, kv_tmp, N * Mk, q.ck->data, kv_tmp + N * Mk, N * Mv, q.cv->data, nullptr, 0, ctx->stream));
    for (int m = 1; m < 7; m++) done[q.idx[m]] = 1;
    ctx->act_invalidate_if_written(q.rq);
    return true;
}

// Small-batch form (2..32 tokens on the skinny kernels): the three products go to scratch as one "plane" each and the same finish launch
// rotates q and k and stores k and v -- 4 launches for the nodes' 7 (3 MUL_MAT, 2 ROPE, 2 CPY).  The products are the bits of the
// separate MUL_MATs (same kernels, same activations), the finish launch adds nothing to a single plane: bit-identical.
static bool small_mm_ok(const ggml_tensor * t) {
    const ggml_tensor * w = t->src[0], * x = t->src[1];
    if (!(t->op == GGML_OP_MUL_MAT && t->type == GGML_TYPE_F32 && x->type == GGML_TYPE_F32 && w->ne[2] == 1 && w->ne[3] == 1 && !tensor_is_split(w) &&
          is_weight_type(w->type) && ggml_is_contiguous(t))) return false;
    const int path = mi355_mul_mat_path(w, x);
    // (2..8 columns on the column-group mat-vec: a step of two or three tokens -- a short speculative draft -- has the same 7-node Q/K/V
    // pattern and the same launches to save)
    return path == MM_SKINNY_K || path == MM_SKINNY || (path == MM_VEC && x->ne[1] > 1 && x->ne[1] <= 8);
}

// wk / wv beside a wq that takes a skinny kernel: any quantized product will do (each lands in its plane through mi355_mul_mat, whatever
// kernel that picks) -- an 8-expert Q4_K_M model has q8_0 wk / wv, which at 33..64 columns take the GEMM while wq stays skinny
static bool small_qkv_sibling_ok(const ggml_tensor * t) {
    const ggml_tensor * w = t->src[0], * x = t->src[1];
    return t->op == GGML_OP_MUL_MAT && t->type == GGML_TYPE_F32 && x->type == GGML_TYPE_F32 && w->ne[2] == 1 && w->ne[3] == 1 && !tensor_is_split(w) &&
           is_weight_type(w->type) && ggml_is_contiguous(t);
}

static bool mi355_try_fuse_qkv_small(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    ggml_tensor * mq0 = ggml_graph_node(cgraph, i);
    if (mq0->src[1]->ne[1] <= 1 || !small_mm_ok(mq0) || MI355_ENV_SET("GGML_MI355_NO_QKV_SMALL")) return false;
    qkv_nodes q;
    if (!qkv_match(cgraph, i, done, small_qkv_sibling_ok, q)) return false;
    const ggml_tensor * x = q.mq->src[1];
    ggml_tensor * mm[3] = {q.mq, q.mk, q.mv};
    const int64_t N = x->ne[1], hd = q.rq->ne[0];
    const int64_t Mq = mm[0]->ne[0], Mk = mm[1]->ne[0], Mv = mm[2]->ne[0];
    if (Mv % 4 || (N * Mv) % 4) return false;
    size_t off[3], total = 0;
    for (int m = 0; m < 3; m++) { off[m] = total; total += align_up((size_t)N * mm[m]->ne[0] * sizeof(float), 256); }
    const ggml_tensor * outs[3] = {q.rq, q.ck, q.cv};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < a; b++) if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), outs[b]->data, ggml_nbytes(outs[b]))) return false;
        if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), q.rq->src[1]->data, ggml_nbytes(q.rq->src[1]))) return false;
        if (ranges_overlap(outs[a]->data, ggml_nbytes(outs[a]), x->data, ggml_nbytes(x))) return false;
    }
    char * parts = (char *)ctx->get_scratch(7, total);      // not [3]: a member that takes the GEMM brings its split-K workspace from there
    if (ctx->scratch_size[7] < total) return false;
    float * kv_tmp = nullptr;
    if (q.kv_q8) {
        const size_t need = (size_t)N * (Mk + Mv) * sizeof(float);
        kv_tmp = (float *)ctx->get_scratch(4, need);
        if (ctx->scratch_size[4] < need) return false;
    }
    mi355_qkv_finish_args f;
    memset(&f, 0, sizeof(f));
    f.q_parts = (const float *)(parts + off[0]); f.k_parts = (const float *)(parts + off[1]); f.v_parts = (const float *)(parts + off[2]);
    f.kz_q = f.kz_k = f.kz_v = 1;
    f.m_q = Mq; f.m_k = Mk; f.m_v = Mv; f.n = N;
    f.q_out = (float *)q.rq->data; f.k_out = q.ck->data; f.v_out = q.cv->data;
    f.k_out_type = (int32_t)q.ck->type; f.v_out_type = (int32_t)q.cv->type;
    if (kv_tmp) { f.k_out = kv_tmp; f.v_out = kv_tmp + N * Mk; f.k_out_type = f.v_out_type = (int32_t)GGML_TYPE_F32; }
    f.pos = (const int32_t *)q.rq->src[1]->data; f.freq_factors = q.rq->src[2] ? (const float *)q.rq->src[2]->data : nullptr;
    f.head_dim = (int32_t)hd;
    float f6[6];
    qkv_rope_params(q.rq, &f.n_dims, &f.mode, &f.n_ctx_orig, f6);
    f.freq_base = f6[0]; f.freq_scale = f6[1]; f.ext_factor = f6[2]; f.attn_factor = f6[3]; f.beta_fast = f6[4]; f.beta_slow = f6[5];
    // everything the finish launch could refuse, before anything is launched (mi355_qkv_finish's own checks)
    if (hd <= 0 || (hd & 1) || f.n_dims > hd || (f.n_dims & 1) || Mq % hd || Mk % hd || (f.mode != 0 && f.mode != 2)) return false;
    if ((f.k_out_type != (int32_t)GGML_TYPE_F16 && f.k_out_type != (int32_t)GGML_TYPE_F32) || (f.v_out_type != (int32_t)GGML_TYPE_F16 && f.v_out_type != (int32_t)GGML_TYPE_F32)) return false;
    if (((uintptr_t)f.v_out & 15) || ((uintptr_t)f.q_out & 3) || ((uintptr_t)f.k_out & 3)) return false;
    // The K-quant products of wq's type leave as ONE launch over all their row tiles (Q4_K_M: wq + wk, and wv where it is q4_K too):
    // 128 + 32 + 32 tiles of an 8B layer are three launches that each cut K over two workgroups behind a memset, or one launch of
    // 192 workgroups with neither.
    bool grouped[3] = {false, false, false};
    const int64_t K = x->ne[0];
    if (N <= 64 && x->ne[2] == 1 && x->ne[3] == 1 && !MI355_ENV_SET("GGML_MI355_NO_SKINNY_GROUP") &&
        mi355_mul_mat_path(mm[0]->src[0], x, mm[0]) == MM_SKINNY_K) {
        const void * Wg[3]; int64_t rs[3], dcs[3], mg[3]; float * dg[3];
        int cnt = 0, who[3];
        for (int m = 0; m < 3; m++) {
            const ggml_tensor * w = mm[m]->src[0];
            if (w->type != mm[0]->src[0]->type || mi355_mul_mat_path(w, x, mm[m]) != MM_SKINNY_K) continue;
            Wg[cnt] = w->data; rs[cnt] = (int64_t)w->nb[1]; dg[cnt] = (float *)(parts + off[m]); dcs[cnt] = mg[cnt] = w->ne[1];
            who[cnt++] = m;
        }
        if (cnt >= 2) {
            const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, N);
            char * ws = (char *)ctx->get_scratch(0, L.total());
            if (!ctx->act_cached(x, 2)) {
                MI_CHECK(mi355_quantize_act_q8k((const float *)x->data, (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), K, N, 1, 1,
                                                x->nb[1], x->nb[2], x->nb[3], ctx->stream));
                ctx->act_remember(x, 2);
            }
            const int rc = mi355_mul_mat_skinny_k_group((int)mm[0]->src[0]->type, cnt, Wg, rs, dg, dcs, mg, (const int8_t *)ws, (const float *)(ws + L.b0),
                                                        (const int16_t *)(ws + L.b0 + L.b1), K, N, ctx->stream);
            if (rc == MI355_OK) for (int c = 0; c < cnt; c++) grouped[who[c]] = true;
            else if (rc != MI355_E_UNSUPPORTED && rc != MI355_E_ALIGN && rc != MI355_E_SHAPE) MI_CHECK(rc);
        }
    }
    if (N <= 8 && x->ne[2] == 1 && x->ne[3] == 1 && mi355_mul_mat_path(mm[0]->src[0], x, mm[0]) == MM_VEC) {
        // 2..8 columns: the mat-vecs of wq's type as one grouped launch (mi355_mul_mat_vec_q_group), results into the planes
        mi355_mmvq_args a[3];
        memset(a, 0, sizeof(a));
        int cnt = 0, who[3];
        const size_t qb = align_up((size_t)mi355_act_q_bytes(K, N), 256), db = align_up((size_t)mi355_act_ds_bytes(K, N), 256);
        char * ws = (char *)ctx->get_scratch(0, qb + db);
        for (int m = 0; m < 3; m++) {
            const ggml_tensor * w = mm[m]->src[0];
            if (w->type != mm[0]->src[0]->type || mi355_mul_mat_path(w, x, mm[m]) != MM_VEC) continue;
            a[cnt].type = (int32_t)w->type; a[cnt].W = w->data; a[cnt].w_row_stride = (int64_t)w->nb[1];
            a[cnt].act_q = (const int8_t *)ws; a[cnt].act_ds = (const float *)(ws + qb); a[cnt].act_ch_stride = N;
            a[cnt].dst = (float *)(parts + off[m]); a[cnt].dst_col_stride = w->ne[1];
            a[cnt].m = w->ne[1]; a[cnt].k = K; a[cnt].ncols = (int32_t)N; a[cnt].nchannels = 1; a[cnt].ch_ratio = 1;
            who[cnt++] = m;
        }
        if (cnt >= 2) {
            if (!ctx->act_cached(x)) {
                MI_CHECK(mi355_quantize_act((const float *)x->data, (int8_t *)ws, (float *)(ws + qb), K, N, 1, 1, x->nb[1], x->nb[2], x->nb[3], ctx->stream));
                ctx->act_remember(x);
            }
            MI_CHECK(mi355_mul_mat_vec_q_group(a, cnt, ctx->stream));
            for (int c = 0; c < cnt; c++) grouped[who[c]] = true;
        }
    }
    for (int m = 0; m < 3; m++) {
        if (grouped[m]) continue;
        ggml_tensor t = *mm[m];              // the product of node m, into its plane instead of the node's own bytes
        t.data = parts + off[m];
        mi355_mul_mat(ctx, &t);
    }
    MI_CHECK(mi355_qkv_finish(&f, ctx->stream));
    if (kv_tmp) MI_CHECK(mi355_kv_store_q(q.kv_qt, kv_tmp, N * Mk, q.ck->data, kv_tmp + N * Mk, N * Mv, q.cv->data, nullptr, 0, ctx->stream));
    for (int m = 1; m < 7; m++) done[q.idx[m]] = 1;
    ctx->act_invalidate_if_written(q.rq);
    return true;
}

// Consecutive small-batch MUL_MATs of one K-quant type on the same activations (ffn_up + ffn_gate of a step of 5..32 tokens,
// llama-graph.cpp build_ffn) as one launch over all their row tiles.
static bool mi355_try_group_small(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    ggml_tensor * first = ggml_graph_node(cgraph, i);
    const ggml_tensor * x = first->src[1];
    if (!small_mm_ok(first) || x->ne[1] > 64 || x->ne[2] != 1 || x->ne[3] != 1 || MI355_ENV_SET("GGML_MI355_NO_SKINNY_GROUP")) return false;
    if (mi355_mul_mat_path(first->src[0], x, first) != MM_SKINNY_K) return false;
    const void * Wg[4]; int64_t rs[4], dcs[4], mg[4]; float * dg[4];
    int cnt = 0;
    const int n = ggml_graph_n_nodes(cgraph);
    for (int j = i; j < n && cnt < 4; j++) {
        ggml_tensor * t = ggml_graph_node(cgraph, j);
        if (done[j] || !small_mm_ok(t) || t->src[1] != x || t->src[0]->type != first->src[0]->type ||
            mi355_mul_mat_path(t->src[0], x, t) != MM_SKINNY_K) break;
        bool clash = ranges_overlap(t->data, ggml_nbytes(t), x->data, ggml_nbytes(x));
        for (int c = 0; c < cnt; c++) clash = clash || ranges_overlap(t->data, ggml_nbytes(t), dg[c], (size_t)dcs[c] * x->ne[1] * sizeof(float));
        if (clash) break;
        Wg[cnt] = t->src[0]->data; rs[cnt] = (int64_t)t->src[0]->nb[1]; dg[cnt] = (float *)t->data;
        dcs[cnt] = (int64_t)(t->nb[1] / sizeof(float)); mg[cnt] = t->src[0]->ne[1];
        cnt++;
    }
    if (cnt < 2) return false;
    const int64_t K = x->ne[0], N = x->ne[1];
    const mi355_act_layout L = mi355_act_planes(MM_SKINNY_K, K, N);
    char * ws = (char *)ctx->get_scratch(0, L.total());
    if (!ctx->act_cached(x, 2)) {
        MI_CHECK(mi355_quantize_act_q8k((const float *)x->data, (int8_t *)ws, (float *)(ws + L.b0), (int16_t *)(ws + L.b0 + L.b1), K, N, 1, 1,
                                        x->nb[1], x->nb[2], x->nb[3], ctx->stream));
        ctx->act_remember(x, 2);
    }
    const int rc = mi355_mul_mat_skinny_k_group((int)first->src[0]->type, cnt, Wg, rs, dg, dcs, mg, (const int8_t *)ws, (const float *)(ws + L.b0),
                                                (const int16_t *)(ws + L.b0 + L.b1), K, N, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_ALIGN || rc == MI355_E_SHAPE) return false;
    MI_CHECK(rc);
    for (int c = 1; c < cnt; c++) done[i + c] = 1;
    for (int c = 0; c < cnt; c++) ctx->act_invalidate_if_written(ggml_graph_node(cgraph, i + c));
    return true;
}

// Decode form.  norm_idx >= 0: nodes norm_idx (RMS_NORM) and norm_idx + 1 (MUL by the norm weight, == x) come right before node i and
// run in the launch's prologue (attn_norm of every layer): 9 nodes, one launch.
static bool mi355_try_fuse_qkv_rope(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done, int norm_idx = -1) {
    ggml_tensor * mq0 = ggml_graph_node(cgraph, i);
    if (!matvec_groupable(mq0) || mq0->src[1]->ne[1] != 1) return false;
    qkv_nodes qn;
    if (!qkv_match(cgraph, i, done, matvec_groupable, qn)) return false;
    ggml_tensor * mq = qn.mq, * rq = qn.rq, * mk = qn.mk, * mv = qn.mv, * ck = qn.ck, * cv = qn.cv;
    const int * idx = qn.idx;
    const bool kv_q8 = qn.kv_q8;
    const ggml_tensor * x = mq->src[1];
    const ggml_tensor * wq = mq->src[0], * wk = mk->src[0], * wv = mv->src[0];
    const int64_t K = wq->ne[0], Mq = wq->ne[1], Mk = wk->ne[1], Mv = wv->ne[1], hd = rq->ne[0];
    const int32_t * rp = (const int32_t *)rq->op_params;
    // activations of the token, q8_1
    const size_t qb = align_up((size_t)mi355_act_q_bytes(K, 1), 256), db = align_up((size_t)mi355_act_ds_bytes(K, 1), 256);
    char * ws = (char *)ctx->get_scratch(0, qb + db);
    mi355_qkv_rope_args a;
    memset(&a, 0, sizeof(a));
    if (norm_idx >= 0) {
        const ggml_tensor * norm = ggml_graph_node(cgraph, norm_idx);
        const ggml_tensor * w = x->src[0] == norm ? x->src[1] : x->src[0];
        const ggml_tensor * outs[4] = {x, rq, ck, cv};
        const size_t ob[4] = {ggml_nbytes(x), ggml_nbytes(rq), ggml_nbytes(ck), ggml_nbytes(cv)};
        if (!norm_prologue_ok(norm->src[0], w, K, 4096, outs, ob, 4)) return false;
        a.norm_x = (const float *)norm->src[0]->data; a.norm_w = (const float *)w->data; a.norm_out = (float *)x->data;
        memcpy(&a.norm_eps, norm->op_params, sizeof(float));
    }
    a.type_qk = (int32_t)wq->type; a.type_v = (int32_t)wv->type;
    a.type_k = wk->type != wq->type ? (int32_t)wk->type : 0;
    a.wq = wq->data; a.wk = wk->data; a.wv = wv->data;
    a.wq_row_stride = (int64_t)wq->nb[1]; a.wk_row_stride = (int64_t)wk->nb[1]; a.wv_row_stride = (int64_t)wv->nb[1];
    a.m_q = Mq; a.m_k = Mk; a.m_v = Mv; a.k = K;
    a.act_q = (const int8_t *)ws; a.act_ds = (const float *)(ws + qb);
    a.q_out = (float *)rq->data; a.k_out = ck->data; a.v_out = cv->data;
    void * const * kv_table = nullptr;
    int kv_slot = 0;
    if (ctx->gc.capturing) {
        if (ctx->gc.n_slots >= MI355_KV_SLOTS) { ctx->gc.capture_failed = true; }
        else { kv_table = ctx->gc.kv_dev; kv_slot = ctx->gc.n_slots; }
    }
    a.k_out_type = (int32_t)ck->type; a.v_out_type = (int32_t)cv->type;
    float * kv_tmp = nullptr;
    if (kv_q8) {
        kv_tmp = (float *)ctx->get_scratch(4, (size_t)(Mk + Mv) * sizeof(float));
        if (ctx->scratch_size[4] < (size_t)(Mk + Mv) * sizeof(float)) return false;       // under capture the buffer could not grow
        a.k_out = kv_tmp; a.v_out = kv_tmp + Mk;
        a.k_out_type = a.v_out_type = (int32_t)GGML_TYPE_F32;
    } else {
        a.kv_table = kv_table; a.kv_slot = kv_slot;      // the fused launch stores into the cache itself
    }
    a.pos = (const int32_t *)rq->src[1]->data; a.freq_factors = rq->src[2] ? (const float *)rq->src[2]->data : nullptr;
    a.head_dim = (int32_t)hd; a.n_dims = rp[1]; a.mode = rp[2]; a.n_ctx_orig = rp[4];
    memcpy(&a.freq_base, rp + 5, 4); memcpy(&a.freq_scale, rp + 6, 4); memcpy(&a.ext_factor, rp + 7, 4);
    memcpy(&a.attn_factor, rp + 8, 4); memcpy(&a.beta_fast, rp + 9, 4); memcpy(&a.beta_slow, rp + 10, 4);
    // dry validation first (argument errors are reported before anything is launched): an unsupported type pair or
    // shape leaves the graph to the unfused path
    const bool need_quant = norm_idx < 0 && !ctx->act_cached(x);
    if (need_quant) {
        MI_CHECK(mi355_quantize_act((const float *)x->data, (int8_t *)ws, (float *)(ws + qb), K, 1, 1, 1, x->nb[1], x->nb[2], x->nb[3], ctx->stream));
        ctx->act_remember(x);
    }
    if (tensor_is_split(wq)) {
        // row-split wq / wk / wv: a device rotates and stores the heads whose rows it owns (the split's 128-row rounding keeps a head's
        // rows together for head_dim <= 128); K and V land at the slice's place in the cache row -- under capture through a table
        // slot of the device's own, whose entries carry the slice's byte offset
        const mi355_split_extra * eq = split_extra_of(wq), * ek = split_extra_of(wk), * ev = split_extra_of(wv);
        if (kv_q8 || !tensor_is_split(wk) || !tensor_is_split(wv)) return false;
        bool has[MI355_MAX_DEVICES] = {};
        const int ndev = mi355_device_count();
        int n_has = 0;
        for (int d = 0; d < ndev; d++) {
            const int64_t rq_ = eq->row_high[d] - eq->row_low[d], rk_ = ek->row_high[d] - ek->row_low[d], rv_ = ev->row_high[d] - ev->row_low[d];
            if (rq_ <= 0 && rk_ <= 0 && rv_ <= 0) continue;
            // the kernel wants rows of all three; whole heads; 16-byte aligned slice starts in the f16 cache rows
            if (rq_ <= 0 || rk_ <= 0 || rv_ <= 0 || rq_ % hd || rk_ % hd || eq->row_low[d] % hd || ek->row_low[d] % hd || rv_ % 8 || ev->row_low[d] % 8) return false;
            has[d] = true; n_has++;
        }
        if (ctx->gc.capturing && ctx->gc.n_slots + n_has > MI355_KV_SLOTS) { ctx->gc.capture_failed = true; kv_table = nullptr; }
        const size_t ksz = ggml_type_size(ck->type), vsz = ggml_type_size(cv->type);
        int slot_of[MI355_MAX_DEVICES] = {};
        if (kv_table) {
            for (int d = 0; d < ndev; d++) {
                if (!has[d]) continue;
                slot_of[d] = ctx->gc.n_slots++;
                ctx->gc.kv_push(idx[5], (size_t)ek->row_low[d] * ksz);
                ctx->gc.kv_push(idx[6], (size_t)ev->row_low[d] * vsz);
            }
        }
        mi355_split_fan_out(ctx, has, [&](int d, hipStream_t st) {
            mi355_qkv_rope_args b = a;
            b.wq = eq->data[d]; b.wk = ek->data[d]; b.wv = ev->data[d];
            b.m_q = eq->row_high[d] - eq->row_low[d]; b.m_k = ek->row_high[d] - ek->row_low[d]; b.m_v = ev->row_high[d] - ev->row_low[d];
            b.q_out = a.q_out + eq->row_low[d];
            b.k_out = (char *)ck->data + (size_t)ek->row_low[d] * ksz;
            b.v_out = (char *)cv->data + (size_t)ev->row_low[d] * vsz;
            b.kv_table = kv_table; b.kv_slot = slot_of[d];
            MI_CHECK(mi355_qkv_rope_matvec(&b, st));
        });
        for (int m = 1; m < 7; m++) done[idx[m]] = 1;
        ctx->act_invalidate_if_written(rq);
        if (norm_idx >= 0) ctx->act_invalidate_if_written(x);
        return true;
    }
    const int rc = mi355_qkv_rope_matvec(&a, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_SHAPE || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    if (kv_q8) MI_CHECK(mi355_kv_store_q(qn.kv_qt, kv_tmp, Mk, ck->data, kv_tmp + Mk, Mv, cv->data, kv_table, kv_slot, ctx->stream));
    for (int m = 1; m < 7; m++) done[idx[m]] = 1;
    if (kv_table) { ctx->gc.kv_push(idx[5], 0); ctx->gc.kv_push(idx[6], 0); ctx->gc.n_slots++; }
    ctx->act_invalidate_if_written(rq);
    if (norm_idx >= 0) ctx->act_invalidate_if_written(x);
    return true;
}

// ---- a decode token's layer body in ONE persistent launch (mi355_decode_layer, kernels/mmvq_layer.hip) ------------------------
//     i: MUL_MAT(wo)  ADD | RMS_NORM  MUL  MUL_MAT(gate | up)  MUL_MAT(up | gate)  GLU | MUL_MAT(down)  ADD |
//     [the next layer's RMS_NORM  MUL  MUL_MAT(wq) .. CPY CPY: the pattern of mi355_try_fuse_qkv_rope]
// i.e. the launches of mi355_try_fuse_matvec_add, mi355_try_fuse_norm_glu_matvec, mi355_try_fuse_matvec_add and
// mi355_try_fuse_qkv_rope (with its norm prologue) as one, bit-identical to them (tests/test_decode_layer_gpu.py,
// test_backend_plugin_gpu.py::test_decode_layer_launch_is_bit_exact_and_runs).  OPT-IN (GGML_MI355_DECODE_LAYER=1): measured end to end
// it is worth +0.9 % on Llama-3-70B-shaped Q4_K_M (tg64 96.4 -> 97.3 t/s; the layer micro-benchmark's 119 -> 111 us per layer does
// not carry over: the plugin's 70B path already groups gate + up behind the norm) and costs 5 % on 8B (tg128 509 -> 484), and a
// launch that needs every workgroup resident at once is not something to run by default on a GPU that may be shared.
// Every tensor of the run is still written.  The launch orders its phases internally, so ggml-alloc's reuse of dead tensors' bytes
// by later nodes of the run is honoured (mi355_kernels.h); what is checked here is what the separate launches check: the ADDs are
// in place or apart, the GLU result is on one of its inputs or apart, the tensors every workgroup reads during a phase (the norm
// inputs and weights) are apart from that phase's outputs.
#define MI355_DL_MAX 256            // sync blocks (= layers) per graph
static bool mi355_try_fuse_decode_layer(mi355_backend_ctx * ctx, ggml_cgraph * cgraph, int i, std::vector<char> & done) {
    static const int mode = [] { const char * e = getenv("GGML_MI355_DECODE_LAYER"); return e ? atoi(e) : 0; }();
    if (mode != 1) return false;
    const int n = ggml_graph_n_nodes(cgraph);
    if (i + 8 >= n) return false;
    ggml_tensor * nd[9];
    for (int j = 0; j < 9; j++) { nd[j] = ggml_graph_node(cgraph, i + j); if (done[i + j]) return false; }
    ggml_tensor * mm0 = nd[0], * add0 = nd[1], * norm1 = nd[2], * mul1 = nd[3], * glu = nd[6], * mm2 = nd[7], * add2 = nd[8];
    if (!matvec_groupable(mm0) || tensor_is_split(mm0->src[0]) || mm0->src[1]->ne[1] != 1 || mm0->src[0]->ne[1] < 2048) return false;
    auto add_of = [](const ggml_tensor * add, const ggml_tensor * mm) -> const ggml_tensor * {     // the residual operand (mi355_try_fuse_matvec_add's checks)
        if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32) return nullptr;
        const ggml_tensor * r = add->src[0] == mm ? add->src[1] : (add->src[1] == mm ? add->src[0] : nullptr);
        if (!r || r == mm || r->type != GGML_TYPE_F32) return nullptr;
        if (!ggml_are_same_shape(mm, add) || !ggml_are_same_shape(r, add) || !ggml_is_contiguous(add) || !ggml_is_contiguous(r)) return nullptr;
        const size_t nb = ggml_nbytes(add);
        if (ranges_overlap(r->data, nb, mm->data, nb)) return nullptr;
        if (add->data != mm->data && ranges_overlap(add->data, nb, mm->data, nb)) return nullptr;
        if (ranges_overlap(add->data, nb, r->data, nb)) return nullptr;       // (in place on the residual: left to the separate launches)
        return r;
    };
    const ggml_tensor * r0 = add_of(add0, mm0);
    if (!r0) return false;
    const ggml_tensor * w1 = decode_norm_pattern(cgraph, i + 2, done);
    if (!w1 || norm1->src[0] != add0) return false;
    const ggml_tensor * ma = nd[4], * mb = nd[5];
    if (!matvec_groupable(ma) || !matvec_groupable(mb) || ma->src[1] != mul1 || mb->src[1] != mul1) return false;
    if (glu->op != GGML_OP_GLU || !glu->src[1] || glu->type != GGML_TYPE_F32) return false;
    const ggml_tensor * gate = glu->src[0], * up = glu->src[1];
    if (!((gate == ma && up == mb) || (gate == mb && up == ma))) return false;
    const int op = (int)ggml_get_glu_op(glu);
    if (op < 0 || op > 2) return false;
    if (!matvec_groupable(mm2) || mm2->src[1] != glu) return false;
    const ggml_tensor * r2 = add_of(add2, mm2);
    if (r2 != add0) return false;
    const ggml_tensor * wo = mm0->src[0], * wg = gate->src[0], * wu = up->src[0], * wd = mm2->src[0];
    const int64_t n_embd = wo->ne[1], k_o = wo->ne[0], n_ff = wg->ne[1];
    if (wg->type != wu->type || !ggml_are_same_shape(wg, wu) || wg->nb[1] != wu->nb[1] || wg->ne[0] != n_embd || wd->ne[0] != n_ff || wd->ne[1] != n_embd) return false;
    if (ggml_nrows(glu) != 1 || glu->ne[0] != n_ff || !ggml_is_contiguous(glu) || mul1->ne[0] != n_embd) return false;
    // the GLU result: on one of its inputs (then that input is not stored) or apart from both, as in mi355_try_fuse_glu_matvec
    float * gate_out = (float *)gate->data, * up_out = (float *)up->data;
    if (glu->data == gate->data) gate_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), gate->data, ggml_nbytes(gate))) return false;
    if (glu->data == up->data) up_out = nullptr;
    else if (ranges_overlap(glu->data, ggml_nbytes(glu), up->data, ggml_nbytes(up))) return false;
    if (ranges_overlap(gate->data, ggml_nbytes(gate), up->data, ggml_nbytes(up))) return false;
    {   // what all workgroups read while others store (per phase): ffn_inp + w1 against ffn_norm / gate / up / hidden
        const ggml_tensor * outs[4] = {mul1, glu, nullptr, nullptr};
        size_t ob[4] = {ggml_nbytes(mul1), ggml_nbytes(glu), 0, 0};
        int no = 2;
        if (gate_out) { outs[no] = gate; ob[no++] = ggml_nbytes(gate); }
        if (up_out)   { outs[no] = up;   ob[no++] = ggml_nbytes(up); }
        if (!norm_prologue_ok(add0, w1, n_embd, 8192, outs, ob, no)) return false;
    }
    // hidden is read by everybody during the down phase, ffn_inp by the stores of l_out
    if (ranges_overlap(glu->data, ggml_nbytes(glu), add2->data, ggml_nbytes(add2)) || ranges_overlap(glu->data, ggml_nbytes(glu), mm2->data, ggml_nbytes(mm2))) return false;
    if (ranges_overlap(glu->data, ggml_nbytes(glu), add0->data, ggml_nbytes(add0))) return false;

    mi355_decode_layer_args a;
    memset(&a, 0, sizeof(a));
    // ---- the tail: the next layer's attn_norm + Q/K/V + rope + KV store
    qkv_nodes qn;
    bool tail = false, kv_q8 = false;
    const ggml_tensor * x3 = nullptr;
    void * const * kv_table = nullptr;
    int kv_slot = 0;
    if (i + 11 < n && !MI355_ENV_SET("GGML_MI355_NO_NORM_QKV")) {
        const ggml_tensor * norm3 = ggml_graph_node(cgraph, i + 9);
        const ggml_tensor * w3 = decode_norm_pattern(cgraph, i + 9, done);
        ggml_tensor * mq = ggml_graph_node(cgraph, i + 11);
        x3 = ggml_graph_node(cgraph, i + 10);
        if (w3 && norm3->src[0] == add2 && mq->op == GGML_OP_MUL_MAT && mq->src[1] == x3 && matvec_groupable(mq) && qkv_match(cgraph, i + 11, done, matvec_groupable, qn) && !qn.kv_q8) {
            const ggml_tensor * wq = qn.mq->src[0], * wk = qn.mk->src[0], * wv = qn.mv->src[0];
            const ggml_tensor * outs[4] = {x3, qn.rq, qn.ck, qn.cv};
            const size_t ob[4] = {ggml_nbytes(x3), ggml_nbytes(qn.rq), ggml_nbytes(qn.ck), ggml_nbytes(qn.cv)};
            if (wq->ne[0] == n_embd && norm_prologue_ok(add2, w3, n_embd, 8192, outs, ob, 4)) {
                mi355_qkv_rope_args & q = a.qkv;
                q.norm_x = (const float *)add2->data; q.norm_w = (const float *)w3->data; q.norm_out = (float *)x3->data;
                memcpy(&q.norm_eps, norm3->op_params, sizeof(float));
                q.type_qk = (int32_t)wq->type; q.type_v = (int32_t)wv->type; q.type_k = wk->type != wq->type ? (int32_t)wk->type : 0;
                q.wq = wq->data; q.wk = wk->data; q.wv = wv->data;
                q.wq_row_stride = (int64_t)wq->nb[1]; q.wk_row_stride = (int64_t)wk->nb[1]; q.wv_row_stride = (int64_t)wv->nb[1];
                q.m_q = wq->ne[1]; q.m_k = wk->ne[1]; q.m_v = wv->ne[1]; q.k = n_embd;
                q.q_out = (float *)qn.rq->data; q.k_out = qn.ck->data; q.v_out = qn.cv->data;
                q.k_out_type = (int32_t)qn.ck->type; q.v_out_type = (int32_t)qn.cv->type;
                if (ctx->gc.capturing) {
                    if (ctx->gc.n_slots >= MI355_KV_SLOTS) ctx->gc.capture_failed = true;
                    else { kv_table = ctx->gc.kv_dev; kv_slot = ctx->gc.n_slots; }
                }
                q.kv_table = kv_table; q.kv_slot = kv_slot;
                q.pos = (const int32_t *)qn.rq->src[1]->data; q.freq_factors = qn.rq->src[2] ? (const float *)qn.rq->src[2]->data : nullptr;
                q.head_dim = (int32_t)qn.rq->ne[0];
                float f6[6];
                qkv_rope_params(qn.rq, &q.n_dims, &q.mode, &q.n_ctx_orig, f6);
                q.freq_base = f6[0]; q.freq_scale = f6[1]; q.ext_factor = f6[2]; q.attn_factor = f6[3]; q.beta_fast = f6[4]; q.beta_slow = f6[5];
                tail = true; kv_q8 = qn.kv_q8;
            }
        }
    }
    (void)kv_q8;
    a.type_o = (int32_t)wo->type; a.type_gu = (int32_t)wg->type; a.type_d = (int32_t)wd->type; a.glu_op = op;
    a.tail = tail ? MI355_TAIL_QKV : MI355_TAIL_NONE;
    a.n_embd = n_embd; a.k_o = k_o; a.n_ff = n_ff;
    a.wo = wo->data; a.wo_row_stride = (int64_t)wo->nb[1];
    const ggml_tensor * src1 = mm0->src[1];
    const size_t qb = align_up((size_t)mi355_act_q_bytes(k_o, 1), 256);
    if (ctx->act_cached(src1)) { a.attn_q = (const int8_t *)ctx->scratch[0]; a.attn_ds = (const float *)((const char *)ctx->scratch[0] + qb); }
    else {
        if (src1->type != GGML_TYPE_F32 || !ggml_is_contiguous(src1) || ((uintptr_t)src1->data & 15)) return false;
        a.attn_f32 = (const float *)src1->data;
    }
    a.residual = (const float *)r0->data; a.attn_out = (float *)mm0->data; a.ffn_inp = (float *)add0->data;
    a.ffn_norm_w = (const float *)w1->data; a.ffn_norm_out = (float *)mul1->data;
    memcpy(&a.norm_eps, norm1->op_params, sizeof(float));
    a.w_gate = wg->data; a.w_up = wu->data; a.gu_row_stride = (int64_t)wg->nb[1];
    a.gate_out = gate_out; a.up_out = up_out; a.hidden = (float *)glu->data;
    a.w_down = wd->data; a.down_row_stride = (int64_t)wd->nb[1];
    a.down_out = (float *)mm2->data; a.l_out = (float *)add2->data;
    // a sync block of this graph's memset
    const size_t sb = (size_t)mi355_decode_layer_sync_bytes();
    if (ctx->dl_used >= MI355_DL_MAX) return false;
    char * sync = (char *)ctx->get_scratch(5, sb * MI355_DL_MAX);
    if (ctx->scratch_size[5] < sb * MI355_DL_MAX) return false;          // under capture the buffer could not grow
    if (ctx->dl_used == 0) HIP_CHECK(hipMemsetAsync(sync, 0, sb * MI355_DL_MAX, ctx->stream));
    int rc = mi355_decode_layer(&a, sync + sb * ctx->dl_used, ctx->stream);
    if (rc == MI355_E_UNSUPPORTED && tail) {                             // e.g. a Q/K/V shape the fused tail does not deal: the body alone
        tail = false; a.tail = MI355_TAIL_NONE;
        rc = mi355_decode_layer(&a, sync + sb * ctx->dl_used, ctx->stream);
    }
    if (rc == MI355_E_UNSUPPORTED || rc == MI355_E_SHAPE || rc == MI355_E_ALIGN) return false;
    MI_CHECK(rc);
    ctx->dl_used++;
    for (int j = 1; j < 9; j++) done[i + j] = 1;
    if (tail) {
        done[i + 9] = 1; done[i + 10] = 1;
        for (int m = 0; m < 7; m++) done[qn.idx[m]] = 1;
        if (kv_table) { ctx->gc.kv_push(qn.idx[5], 0); ctx->gc.kv_push(qn.idx[6], 0); ctx->gc.n_slots++; }
    }
    ctx->act_cache.valid = false;        // nothing of what scratch[0] held describes a tensor the following nodes read
    return true;
}

// issue the launches of a graph on ctx->stream (eagerly, or into the stream capture in progress)
enum ggml_status mi355_issue_graph(mi355_backend_ctx * ctx, ggml_cgraph * cgraph) {
    ctx->act_cache.valid = false; // tensors may have been rewritten through set_tensor / copies between calls
    ctx->moe_tables.valid = false;
    mi355_note_graph_activations(cgraph);
    ctx->dl_used = 0;
    const int n = ggml_graph_n_nodes(cgraph);
    const bool fuse = !MI355_ENV_SET("GGML_MI355_NO_FUSE");
    if (MI355_ENV_SET("GGML_MI355_DUMP_GRAPH")) {   // diagnostics: the node list this backend was handed
        for (int i = 0; i < n; i++) {
            const ggml_tensor * t = ggml_graph_node(cgraph, i);
            fprintf(stderr, "[mi355] %4d %-14s %-28s [%lld,%lld,%lld,%lld] %p", i, ggml_op_name(t->op), t->name, (long long)t->ne[0],
                    (long long)t->ne[1], (long long)t->ne[2], (long long)t->ne[3], t->data);
            for (int s2 = 0; s2 < GGML_MAX_SRC; s2++) if (t->src[s2]) fprintf(stderr, "  <- %s(%p)", t->src[s2]->name, t->src[s2]->data);
            fprintf(stderr, "\n");
        }
    }
    std::vector<char> done(n, 0);    // nodes already computed by a hoisting fusion
    for (int i = 0; i < n; i++) {
        ggml_tensor * node = ggml_graph_node(cgraph, i);
        if (done[i]) continue;
        // empty tensors and pure view ops compute nothing (ggml-cuda.cu:2589)
        if (ggml_is_empty(node) || node->op == GGML_OP_NONE || node->op == GGML_OP_RESHAPE || node->op == GGML_OP_VIEW ||
            node->op == GGML_OP_PERMUTE || node->op == GGML_OP_TRANSPOSE) {
            continue;
        }
        // fusion: residual ADD -> RMS_NORM -> MUL(weight) in one launch (the add result stays a graph tensor)
        if (node->op == GGML_OP_ADD && i + 2 < n && ggml_graph_node(cgraph, i + 1)->op == GGML_OP_RMS_NORM &&
            ggml_graph_node(cgraph, i + 1)->src[0] == node && mi355_try_fuse_rms_norm_mul(ctx, cgraph, i + 1, node)) {
            i += 2;
            continue;
        }
        ctx->act_invalidate_if_written(node);
        // fusion: Q/K/V projections + rope + KV-cache store of one token in one launch
        if (fuse && node->op == GGML_OP_MUL_MAT && node->src[0]->type == GGML_TYPE_F32 && mi355_try_fuse_moe_router(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_decode_layer(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_qkv_rope(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_qkv_gemm(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_qkv_small(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_group_small(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_gemm_add_norm(ctx, cgraph, i, done)) {
            i += 3;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL_MAT && mi355_try_fuse_glu_gemm(ctx, cgraph, i, done)) {
            done[i + 1] = 1;
            done[i + 2] = 1;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL_MAT_ID && mi355_try_fuse_moe_glu_matvec(ctx, cgraph, i, done)) {
            done[i + 1] = 1;
            done[i + 2] = 1;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL_MAT_ID && mi355_try_fuse_moe_down(ctx, cgraph, i, done)) continue;
        if (fuse && node->op == GGML_OP_MUL_MAT_ID && i + 1 < n && !done[i + 1] && mi355_mul_mat_id_pair(ctx, node, ggml_graph_node(cgraph, i + 1))) {
            done[i + 1] = 1;
            continue;
        }
        if (fuse && node->op == GGML_OP_MUL && node->ne[1] == 2 && mi355_try_fuse_moe_combine(ctx, cgraph, i, done)) continue;
        // fusion: decode mat-vecs that share their activation run as one grouped launch
        if (fuse && matvec_groupable(node)) {
            int members[MI355_MMVQ_MAX_GROUP];
            const int cnt = mi355_collect_matvec_group(cgraph, i, done, members);
            if (cnt == 2 && mi355_try_fuse_glu_matvec(ctx, cgraph, members)) {
                done[members[1]] = 1;
                done[members[1] + 1] = 1;
                continue;
            }
            if (cnt > 1) {
                mi355_mul_mat_group(ctx, cgraph, members, cnt);
                for (int m = 1; m < cnt; m++) done[members[m]] = 1;
                continue;
            }
            if (mi355_try_fuse_matvec_add(ctx, cgraph, i)) {
                done[i + 1] = 1;
                continue;
            }
        }
        // fusion: RMS_NORM followed by MUL with the per-feature norm weight (every llama layer), when the
        // norm result has no other reader in this graph
        if (fuse && node->op == GGML_OP_RMS_NORM && mi355_try_fuse_norm_glu_matvec(ctx, cgraph, i, done)) {
            i += 4;
            continue;
        }
        // the ffn norm of a MoE decode step: into the router's launch (the experts find the q8_1 form it leaves)
        if (fuse && node->op == GGML_OP_RMS_NORM && i + 2 < n && decode_norm_pattern(cgraph, i, done)) {
            int j = i + 2;      // (the reshaped view of the norm that the experts will read sits in between)
            while (j < n && !done[j] && is_view_op(ggml_graph_node(cgraph, j))) j++;
            if (j < n && !done[j] && ggml_graph_node(cgraph, j)->op == GGML_OP_MUL_MAT && ggml_graph_node(cgraph, j)->src[0]->type == GGML_TYPE_F32 &&
                ggml_graph_node(cgraph, j)->src[1] == ggml_graph_node(cgraph, i + 1) && mi355_try_fuse_moe_router(ctx, cgraph, j, done, i)) {
                done[i + 1] = 1; done[j] = 1;      // the norm's MUL and the router's MUL_MAT; the five nodes after it are marked done already
                continue;
            }
        }
        if (fuse && node->op == GGML_OP_RMS_NORM && i + 2 < n && !MI355_ENV_SET("GGML_MI355_NO_NORM_QKV") && decode_norm_pattern(cgraph, i, done) &&
            ggml_graph_node(cgraph, i + 2)->op == GGML_OP_MUL_MAT && ggml_graph_node(cgraph, i + 2)->src[1] == ggml_graph_node(cgraph, i + 1) &&
            mi355_try_fuse_qkv_rope(ctx, cgraph, i + 2, done, i)) {
            i += 2;      // the norm, its MUL and wq; the other six nodes are marked done
            continue;
        }
        {
            int consumed = 0;
            if (fuse && node->op == GGML_OP_RMS_NORM && mi355_try_fuse_norm_group(ctx, cgraph, i, done, &consumed)) {
                i += consumed;
                continue;
            }
        }
        if (node->op == GGML_OP_RMS_NORM && i + 1 < n && mi355_try_fuse_rms_norm_mul(ctx, cgraph, i)) {
            i++;
            continue;
        }
        if (!mi355_compute_forward(ctx, node, i + 1 < n ? ggml_graph_node(cgraph, i + 1) : nullptr)) {
            MI_LOG_ERROR("%s: op not supported %s (%s)\n", __func__, node->name, ggml_op_name(node->op));
            return GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;
}

