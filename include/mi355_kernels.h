/*
 * mi355_kernels.h -- C ABI of libmi355_kernels.so: hand-written HIP (gfx950 / CDNA4) kernels for
 * llama.cpp's quantized MUL_MAT / MUL_MAT_ID / FLASH_ATTN_EXT hot path.
 *
 * This is the "thin C ABI" of BASELINE.json:north_star.  Plain pointers and sizes only: no ggml
 * types, no torch types, no C++.  All pointers except `ids_host`-style arguments are DEVICE
 * pointers valid on the current HIP device; `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Every launcher is asynchronous, allocates nothing and is graph-capture safe.
 * The ggml backend plugin (libggml-mi355.so, include/ggml_mi355.h) is the host side that calls
 * these from ggml_backend_i.graph_compute.
 *
 * Return value: 0 on success; a positive value is a hipError_t from the launch; negative values
 * are argument errors (MI355_E_*), reported before anything is launched.
 *
 * Reference citations are file:line relative to the reference checkout (gnana997/llm).
 */
#ifndef MI355_KERNELS_H
#define MI355_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_API __attribute__((visibility("default")))

/* ggml_type ids (ggml/include/ggml.h:363-400) of the types this library computes with */
enum mi355_type {
    MI355_TYPE_F32  = 0,
    MI355_TYPE_F16  = 1,
    MI355_TYPE_Q4_0 = 2,
    MI355_TYPE_Q4_1 = 3,
    MI355_TYPE_Q5_0 = 6,
    MI355_TYPE_Q5_1 = 7,
    MI355_TYPE_Q8_0 = 8,
    MI355_TYPE_Q8_1 = 9,
    MI355_TYPE_Q2_K = 10,
    MI355_TYPE_Q3_K = 11,
    MI355_TYPE_Q4_K = 12,
    MI355_TYPE_Q5_K = 13,
    MI355_TYPE_Q6_K = 14,
    MI355_TYPE_IQ4_NL = 20,
    MI355_TYPE_IQ4_XS = 23,
    MI355_TYPE_I32  = 26,
    MI355_TYPE_I64  = 27,
    MI355_TYPE_BF16 = 30,
};

enum mi355_error {
    MI355_OK            = 0,
    MI355_E_TYPE        = -1, /* unsupported block type */
    MI355_E_SHAPE       = -2, /* k not a multiple of the block size, negative dims, ncols out of range */
    MI355_E_ALIGN       = -3, /* pointer / stride alignment the kernel relies on is not met */
    MI355_E_NULL        = -4, /* required pointer is NULL */
    MI355_E_WORKSPACE   = -5, /* workspace too small */
    MI355_E_UNSUPPORTED = -6, /* valid ggml op, but a variant this library does not implement */
};
/* Every entry point returns 0, one of the negative codes above (nothing was launched), or a POSITIVE hipError_t when a HIP runtime
 * call (launch, memset, symbol lookup) failed; mi355_error_string() describes all three. */

/* library / device introspection */
#define MI355_ABI_VERSION 6                             /* bumped on any signature or struct change */
MI355_API int         mi355_abi_version(void);          /* == MI355_ABI_VERSION of the header the library was built from;
                                                         * the plugin and the Python binding refuse a library that differs */
MI355_API int64_t     mi355_args_sizeof(int which);     /* sizeof of 0 mi355_mmvq_args, 1 qkv_rope, 2 glu_matvec, 3 moe_down,
                                                         * 4 fattn_args, 5 mi355_tensor, 6 qkv_finish_args, 7 decode_layer_args; -1 otherwise */
MI355_API const char *mi355_error_string(int err);      /* MI355_E_* or hipGetErrorString */
MI355_API int64_t     mi355_blck_size(int type);        /* elements per block, 0 if unknown */
MI355_API int64_t     mi355_type_size(int type);        /* bytes per block, 0 if unknown */
MI355_API int64_t     mi355_row_size(int type, int64_t k); /* bytes of a row of k elements, -1 if k % blck */

/* ---------------------------------------------------------------------------------------------
 * Per-launch kernel timing (the backend's per-op profiler; the reference only has the offline
 * `test-backend-ops perf` mode, tests/test-backend-ops.cpp:598-749).  Off by default.  When on,
 * every kernel launch carries a start/stop event pair on its own dispatch, so the time is the
 * kernel's execution time (what rocprofv3 --kernel-trace reports), without launch gaps.
 * Not graph-capture safe while enabled.  Tags: "mmvq_<type>_n<ncols>", "quantize_q8_1", "dequantize".
 */
MI355_API void    mi355_timing_enable(int on);
MI355_API void    mi355_timing_reset(void);
MI355_API int     mi355_timing_read(const char *tag, int64_t *launches, double *total_us);
MI355_API int64_t mi355_timing_report(char *buf, int64_t cap);

/* ---------------------------------------------------------------------------------------------
 * Activation quantization, f32 -> block_q8_1 {half d, half s, int8 qs[32]} (36 B / 32 values).
 * BIT-EXACT with quantize_row_q8_1_ref (ggml/src/ggml-quants.c:214-247):
 *   d = amax/127, id = d ? 1/d : 0, q = roundf(x*id), s = fp16(sum(q) * d).
 * Replaces quantize_row_q8_1_cuda / quantize_q8_1 (ggml/src/ggml-cuda/quantize.cu:4-48,148-163).
 *
 * x is a 4-D f32 tensor with ne0 = k contiguous and byte strides nb1, nb2, nb3 (ggml layout; pass
 * nb1 = k*4, ne2 = ne3 = 1 for a plain matrix).  Row (i1,i2,i3) goes to blocks
 * y + ((i3*ne2 + i2)*ne1 + i1) * (k/32) * 36.
 */
MI355_API int mi355_quantize_row_q8_1(const float *x, void *y, int64_t k,
                                      int64_t ne1, int64_t ne2, int64_t ne3,
                                      int64_t nb1, int64_t nb2, int64_t nb3, void *stream);

/* Same arithmetic, written in the library's internal activation layout ("act"):
 *   act_q  : int8  [rows][k]        the q values, 16-byte aligned rows (k % 32 == 0)
 *   act_ds : float [rows][k/32][2]  { float(fp16(d)), float(sum(q)) } per 32-block
 * The sum is kept as the exact integer (not the fp16-rounded s) so the q4_K/q5_K min terms and the
 * q4_0 / q6_K offsets are exact.  mi355_act_bytes() gives the bytes of each part. */
MI355_API int64_t mi355_act_q_bytes(int64_t k, int64_t rows);
MI355_API int64_t mi355_act_ds_bytes(int64_t k, int64_t rows);
MI355_API int mi355_quantize_act(const float *x, int8_t *act_q, float *act_ds, int64_t k,
                                 int64_t ne1, int64_t ne2, int64_t ne3,
                                 int64_t nb1, int64_t nb2, int64_t nb3, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Block dequantization to f32, BIT-EXACT with dequantize_row_{q4_0,q8_0,q4_K,q5_K,q6_K}
 * (ggml/src/ggml-quants.c:249-267,343-357,1274-1296,1476-1501,1684-1713); also q8_1 (q*d) and
 * f16.  Replaces ggml_get_to_fp32_cuda (ggml/src/ggml-cuda/convert.cu).  n = number of elements.
 */
MI355_API int mi355_dequantize_row(int type, const void *blocks, float *y, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Decode mat-vec on block-quantized weights: dst[c][r] = sum_k W[r][k] * act[c][k], ncols <= 8.
 * Replaces ggml_cuda_mul_mat_vec_q / mul_mat_vec_q (ggml/src/ggml-cuda/mmvq.cu:137-224,491-561).
 *
 * W        : blocks of `type`, row r of channel ch at W + ch_w*w_ch_stride + r*w_row_stride (bytes)
 * act_q/ds : activations quantized by mi355_quantize_act, column c of channel ch is act row
 *            (ch*act_ch_stride + c)
 * dst      : f32, element (r, c, ch) at dst[ch*dst_ch_stride + c*dst_col_stride + r]
 * nchannels: number of independent (weight-slice, activation) problems in this launch (ggml's
 *            ne12*ne13 broadcast, or the (token, expert-slot) pairs of MUL_MAT_ID);
 *            ch_w = ch / ch_ratio when ids == NULL (ggml broadcast ne12/ne02), else
 *            ch_w = ids[(ch / ids_nused) * ids_stride + ch % ids_nused]  (expert routing on device,
 *            replaces the ids path of mmvq.cu:161-164) and the activation row becomes
 *            (ch / ids_nused) * act_ch_stride + (ch % ids_nused) % act_ne1.
 */
typedef struct mi355_mmvq_args {
    int32_t        type;
    const void    *W;
    int64_t        w_row_stride;   /* bytes */
    int64_t        w_ch_stride;    /* bytes */
    const int8_t  *act_q;
    const float   *act_ds;
    int64_t        act_ch_stride;  /* in act rows */
    float         *dst;
    int64_t        dst_col_stride; /* in floats */
    int64_t        dst_ch_stride;  /* in floats */
    int64_t        m, k;
    int32_t        ncols;          /* 1..8 */
    int32_t        nchannels;      /* >= 1 */
    int32_t        ch_ratio;       /* >= 1 */
    const int32_t *ids;            /* optional, device */
    int32_t        ids_stride;     /* int32 elements between tokens */
    int32_t        ids_nused;      /* expert slots per token */
    int32_t        act_ne1;        /* b->ne[1] of MUL_MAT_ID (1 or ids_nused) */
    int32_t        n_expert;       /* bound for ids values (out-of-range ids are skipped) */
    const float   *residual;       /* optional (nchannels == 1, no ids): f32 laid out like dst; then sum = dst + residual is */
    float         *sum;            /* stored too -- the ADD node after wo / ffn_down (ggml-cuda/binbcast.cu), bit-identical
                                    * to a separate add.  sum == dst: only the sum is stored (ADD in place on the result);
                                    * sum == residual is allowed.  Both NULL or both set. */
    /* optional (ncols == 1, nchannels == 1, no ids; k <= 16384, k % 32 == 0; 16-byte aligned): the activations given as
     * the f32 row src_f32[k] instead of act_q / act_ds (which are then not read).  Every workgroup quantizes the row for
     * itself in its prologue, bit-identical to mi355_quantize_act -- the reference's separate quantize_q8_1 launch
     * (ggml-cuda/quantize.cu:4-48) disappears.  With norm_w set the row is first normalised: q8_1((x * rsqrt(mean(x^2) +
     * norm_eps)) * norm_w), bit-identical to mi355_rms_norm_mul_quant_f32, and norm_out[k] receives the f32 result
     * (RMS_NORM + MUL nodes, ggml-cuda/norm.cu).  src_f32, norm_w and every output must not overlap. */
    const float   *src_f32;
    const float   *norm_w;
    float         *norm_out;
    float          norm_eps;
} mi355_mmvq_args;

MI355_API int mi355_mul_mat_vec_q(const mi355_mmvq_args *args, void *stream);

/* Grouped mat-vec: n_mats (1..MI355_MMVQ_MAX_GROUP) matrices of the same type and k that read the SAME quantized
 * activations (same act_q / act_ds / ncols; nchannels == 1, no ids) computed by ONE launch -- wq/wk/wv or ffn
 * gate/up at decode.  The reference launches one mul_mat_vec_q per matrix (ggml-cuda/mmvq.cu:491-561); on MI355X
 * the ~3 us floor of a launch is a large part of a 9 MB mat-vec, so the matrices share a grid instead.  Results are
 * bit-identical to separate mi355_mul_mat_vec_q calls. */
#define MI355_MMVQ_MAX_GROUP 4
MI355_API int mi355_mul_mat_vec_q_group(const mi355_mmvq_args *args, int n_mats, void *stream);

/* Attention-input stage of one decode token in ONE launch: q = rope(Wq x), k = rope(Wk x), v = Wv x, with k and v
 * written straight into their KV-cache rows (f16 or f32).  Replaces, for n_tokens == 1, the node sequence
 * MUL_MAT(wq) ROPE MUL_MAT(wk) ROPE MUL_MAT(wv) CPY(k) CPY(v) of llm_build_llama (src/llama-model.cpp:4898-5052), i.e.
 * 3 x mul_mat_vec_q + 2 x rope + 2 x cpy launches of the reference.  Bit-identical to running those kernels one
 * by one.  wq/wk share type_qk, wv has type_v (Q4_K_M gives wv Q6_K or Q5_K in some layers); rows of every matrix
 * a multiple of 8; head_dim = ne0 of the rope (rows per head); rope parameters as in mi355_rope_f32 (pos[0] is the
 * token position).  MI355_E_UNSUPPORTED for type pairs / shapes outside this, callers then run the nodes singly. */
typedef struct mi355_qkv_rope_args {
    int32_t type_qk, type_v;
    const void *wq, *wk, *wv;
    int64_t wq_row_stride, wk_row_stride, wv_row_stride;   /* bytes */
    int64_t m_q, m_k, m_v, k;
    const int8_t *act_q; const float *act_ds;              /* activations of the token (mi355_quantize_act) */
    float *q_out;                                          /* f32 [m_q] */
    void  *k_out; void *v_out;                             /* [m_k], [m_v] of k_out_type / v_out_type */
    int32_t k_out_type, v_out_type;                        /* MI355_TYPE_F16 or MI355_TYPE_F32 */
    const int32_t *pos; const float *freq_factors;         /* device; freq_factors may be NULL */
    int32_t head_dim, n_dims, mode, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    int32_t type_k;                                        /* type of wk when it differs from wq's (8-expert Q4_K_M: wq Q4_K, wk/wv
                                                            * Q8_0); 0 = same as type_qk */
    /* optional RMS_NORM -> MUL(weight) prologue, as in mi355_glu_matvec_args (k <= 4096 here): the activations are
     * q8_1((norm_x * rsqrt(mean(norm_x^2) + norm_eps)) * norm_w), act_q / act_ds are not read, norm_out receives the f32
     * result.  norm_x, norm_w and every output of the launch must not overlap. */
    const float *norm_x, *norm_w; float *norm_out; float norm_eps;
    /* optional indirection for launches replayed from a HIP graph (the plugin's decode graph, ggml-cuda's analogue:
     * ggml_cuda_cpy_dest_ptrs, ggml-cuda/cpy.cu): the KV-cache rows a token is stored to move with every token while
     * the captured kernel arguments cannot.  When kv_table is set (device array of pointers) the kernel stores K to
     * kv_table[2 * kv_slot] and V to kv_table[2 * kv_slot + 1]; k_out / v_out are then only checked, not used. */
    void *const *kv_table; int32_t kv_slot;
} mi355_qkv_rope_args;
MI355_API int mi355_qkv_rope_matvec(const mi355_qkv_rope_args *args, void *stream);

/* Feed-forward input stage of one decode token in ONE launch: g = W_gate x, u = W_up x, dst = act(g) * u, plus the q8_1
 * form of dst (mi355_quantize_act layout, one row) for the ffn_down mat-vec.  Replaces, for n_tokens == 1, the nodes
 * MUL_MAT(ffn_up) MUL_MAT(ffn_gate) GLU(split) of build_ffn (src/llama-graph.cpp:470-640), i.e. 2 x mul_mat_vec_q
 * (ggml-cuda/mmvq.cu:491-561) + the gated activation (ggml-cuda/unary.cu: ggml_cuda_op_swiglu & co) + the q8_1
 * quantization of the next MUL_MAT (ggml-cuda/quantize.cu).  Bit-identical to those kernels run one by one.  Both matrices
 * have `type` and m rows (m % 32 == 0: a workgroup owns one q8_1 block of the result); gate_out / up_out (the f32 results
 * of the two MUL_MAT nodes) may be NULL; out_q / out_ds may both be NULL, and must not overlap act_q / act_ds. */
typedef struct mi355_glu_matvec_args {
    int32_t type, glu_op;                                  /* weight type; MI355_GLU_* (0 reglu, 1 geglu, 2 swiglu) */
    const void *w_gate, *w_up;
    int64_t gate_row_stride, up_row_stride;                /* bytes */
    int64_t m, k;
    const int8_t *act_q; const float *act_ds;              /* activations of the token (mi355_quantize_act) */
    float *gate_out, *up_out;                              /* f32 [m] each, optional */
    float *dst;                                            /* f32 [m], 16-byte aligned */
    int8_t *out_q; float *out_ds;                          /* q8_1 SoA of dst: [m] int8, [m/32][2] f32; optional */
    /* optional RMS_NORM -> MUL(weight) prologue (build_norm, src/llama-graph.cpp; ggml-cuda/norm.cu rms_norm_f32 + the
     * weight MUL): when norm_x is set the activations are q8_1((norm_x * rsqrt(mean(norm_x^2) + norm_eps)) * norm_w),
     * computed by every workgroup for itself, bit-identical to mi355_rms_norm_mul_quant_f32; act_q / act_ds are not
     * read, norm_out receives the f32 result (k floats).  k <= 8192; the three pointers 16-byte aligned.  norm_x,
     * norm_w and every output of the launch must not overlap (workgroups finish at different times). */
    const float *norm_x, *norm_w; float *norm_out; float norm_eps;
    /* optional expert routing -- the MUL_MAT_ID(ffn_up_exps) MUL_MAT_ID(ffn_gate_exps) GLU nodes of one decode token
     * (build_moe_ffn, src/llama-graph.cpp:660-760; ggml_cuda_mul_mat_id, ggml-cuda.cu:2065-2206): ids = the token's
     * n_slots expert ids (device, int32); slot s uses the matrices at w_gate + ids[s] * gate_expert_stride (bytes) and
     * w_up + ids[s] * up_expert_stride, reads the same activations, and writes its m outputs at index s * m of gate_out /
     * up_out / dst / out_q (out_ds: s * m / 32 pairs).  Out-of-range ids leave the slot's outputs untouched.  Not
     * combinable with the norm prologue. */
    const int32_t *ids; int32_t n_slots, n_expert; int64_t gate_expert_stride, up_expert_stride;
} mi355_glu_matvec_args;
MI355_API int mi355_glu_matvec(const mi355_glu_matvec_args *args, void *stream);

/* Expert feed-forward output of one decode token in ONE launch: y[s] = W[ids[s]] h[s] (MUL_MAT_ID(ffn_down_exps)), y[s] *=
 * weights[s] (the MUL by the normalised router weights), sum_out = (y[0] + y[1]) + ... (the ADD chain over the slot views),
 * residual_out = sum_out + residual (the layer's residual ADD) -- build_moe_ffn, src/llama-graph.cpp:760-795; replaces
 * ggml_cuda_mul_mat_id (ggml-cuda.cu:2065-2206) + 3 binbcast launches.  Bit-identical to those kernels run one by one.
 * act_q / act_ds: q8_1 SoA of the n_slots activation rows (mi355_quantize_act layout, slot s = row s); out (optional):
 * unweighted y [n_slots][m]; weighted_out: [n_slots][m] (may equal out: MUL in place); sum_out [m]; residual /
 * residual_out [m] optional (both or neither; residual_out may equal sum_out or residual).  n_slots <= 8.  Slots with an
 * out-of-range id contribute nothing and are not stored. */
typedef struct mi355_moe_down_args {
    int32_t type, n_slots, n_expert;
    const void *W; int64_t w_row_stride, w_expert_stride;  /* bytes */
    int64_t m, k;
    const int8_t *act_q; const float *act_ds;
    const int32_t *ids; const float *weights;              /* device, n_slots each */
    float *out, *weighted_out, *sum_out;
    const float *residual; float *residual_out;
} mi355_moe_down_args;
MI355_API int mi355_moe_down_matvec(const mi355_moe_down_args *args, void *stream);

/* The body of one decode token's transformer layer in ONE persistent launch -- no counterpart in the reference, which runs a
 * launch per node (ggml_cuda_mul_mat_vec_q, ggml-cuda/mmvq.cu:491-561; quantize_q8_1, quantize.cu:4-48; rms_norm_f32, norm.cu;
 * binbcast.cu; rope.cu; cpy.cu) -- replacing, for n_tokens == 1, the node run of llm_build_llama (src/llama-model.cpp:4898-5052)
 *     MUL_MAT(wo) ADD | RMS_NORM MUL MUL_MAT(gate) MUL_MAT(up) GLU | MUL_MAT(down) ADD | tail
 * where the tail is the NEXT layer's RMS_NORM MUL MUL_MAT(wq) ROPE MUL_MAT(wk) ROPE MUL_MAT(wv) CPY CPY (MI355_TAIL_QKV: `qkv`, an
 * mi355_qkv_rope_args whose norm_x is l_out), the model's RMS_NORM MUL MUL_MAT(output) (MI355_TAIL_ROWS: `out`, an mi355_mmvq_args
 * whose src_f32 is l_out, with norm_w / norm_out), or nothing.  It is the four launches mi355_mul_mat_vec_q_group (+ residual),
 * mi355_glu_matvec (+ norm), mi355_mul_mat_vec_q_group (+ residual), mi355_qkv_rope_matvec (+ norm) fused, and bit-identical
 * to them: one workgroup per CU stays resident, the weight stream (LDS-DMA rings) runs on across the phase boundaries, and
 * the three vectors that cross workgroups (ffn_inp, hidden, l_out) are handed over inside the launch through write-through
 * stores, sharded arrival counters and L1-bypassing loads.
 *   All matrices Q4_K / Q5_K / Q6_K; n_embd, k_o (columns of wo) and n_ff multiples of 256, >= 2048; n_embd <= 8192 and a
 *   multiple of 4 x the workgroup count (the CU count, halved until it divides); otherwise MI355_E_UNSUPPORTED and the caller
 *   runs the separate launches.  attn_q / attn_ds (mi355_quantize_act layout) or attn_f32: the attention output.  attn_out,
 *   gate_out, up_out, down_out (the MUL_MAT nodes' own results) may be NULL.  All f32 pointers 16-byte aligned.  attn_out may equal
 *   ffn_inp and down_out may equal l_out (ADD in place).  Beyond that the launch keeps the node order's guarantees for memory that
 *   ggml-alloc reuses: an output may lie on the bytes of an output or input of an EARLIER node of the run (l_out on ffn_norm_out, the
 *   tail's q_out on gate_out, the tail's norm_out on residual, ...) and then holds the later node's values; outputs of the same node
 *   group (gate_out / up_out / hidden; q_out / k_out / v_out) and the weights must not overlap anything written.
 *   sync: mi355_decode_layer_sync_bytes() bytes of device memory, 128-byte aligned, ZEROED by the caller before every call on
 *   the same stream (hipMemsetAsync; one memset may clear the blocks of all of a token's calls) and not shared by calls that
 *   can overlap.  Its last word receives a non-zero code if a wait inside the launch ran out (4 ms: a workgroup was not
 *   resident, i.e. something else occupied the GPU): the launch then still ends, with undefined results. */
enum { MI355_TAIL_NONE = 0, MI355_TAIL_QKV = 1, MI355_TAIL_ROWS = 2 };
typedef struct mi355_decode_layer_args {
    int32_t type_o, type_gu, type_d, glu_op, tail;
    int64_t n_embd, k_o, n_ff;
    const void *wo; int64_t wo_row_stride;                 /* [n_embd rows][k_o] */
    const int8_t *attn_q; const float *attn_ds; const float *attn_f32;
    const float *residual; float *attn_out; float *ffn_inp;
    const float *ffn_norm_w; float *ffn_norm_out; float norm_eps;
    const void *w_gate, *w_up; int64_t gu_row_stride;      /* [n_ff rows][n_embd] each */
    float *gate_out, *up_out, *hidden;
    const void *w_down; int64_t down_row_stride;           /* [n_embd rows][n_ff] */
    float *down_out, *l_out;
    mi355_qkv_rope_args qkv;                               /* MI355_TAIL_QKV */
    mi355_mmvq_args out;                                   /* MI355_TAIL_ROWS */
} mi355_decode_layer_args;
MI355_API int64_t mi355_decode_layer_sync_bytes(void);
MI355_API int mi355_decode_layer(const mi355_decode_layer_args *args, void *sync, void *stream);

/* "Skinny" MUL_MAT, 1 <= n <= 32 activation columns per call, on the int8 matrix cores: same operands and
 * arithmetic contract as mi355_mul_mat_vec_q (one channel), but one v_mfma_i32_32x32x32_i8 covers 32 rows x 32 columns
 * of a 32-block, so the time per weight byte does not grow with n (the mat-vec kernel drops to 9 % of the HBM
 * roofline at 8 columns).  Replaces mul_mat_vec_q with ncols_dst > 1 (mmvq.cu:137-224) and the small-mmq_x end of
 * mul_mat_q (mmq.cuh:3095-3175).  k % 256 == 0 and n <= 32, else MI355_E_UNSUPPORTED (callers loop over passes of 32
 * columns or use the other kernels).  dst[c*dst_col_stride + r]. */
MI355_API int mi355_mul_mat_skinny(int type, const void *W, int64_t w_row_stride, const int8_t *act_q, const float *act_ds,
                                   float *dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n, void *stream);
/* same for m rows of a matrix of m_total rows (row split): whether K is cut over two workgroups follows the whole matrix, so a slice
 * gives the bits of the unsplit call (ggml-cuda.cu:1431-1749 runs its kernels on row slices likewise). */
MI355_API int mi355_mul_mat_skinny_slice(int type, const void *W, int64_t w_row_stride, const int8_t *act_q, const float *act_ds,
                                         float *dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n, void *stream);

/* q8_K activations, the per-256 quantization ggml-cpu pairs with the K-quants (quantize_row_q8_K_ref,
 * ggml-quants.c:2471-2508; bit-exact), as three planes: q int8 [rows][k], d f32 [rows][k/256], bsums int16 [rows][k/16].
 * x is addressed like mi355_quantize_act's (4-D, byte strides, 16-byte aligned rows); k % 256 == 0. */
MI355_API int mi355_quantize_act_q8k(const float *x, int8_t *q, float *d, int16_t *bsums, int64_t k, int64_t ne1, int64_t ne2,
                                     int64_t ne3, int64_t nb1, int64_t nb2, int64_t nb3, void *stream);

/* Skinny MUL_MAT of a K-quant (Q4_K, Q5_K, Q6_K; others MI355_E_UNSUPPORTED) on q8_K activations, n <= 64 columns (33..64: the two
 * column halves of a 32-row tile are two tiles of one workgroup, so the weights still cross HBM once and the launch's fixed costs
 * are paid once; same integers, the f32 additions over K in a different order than two calls of 32 columns): the
 * arithmetic of ggml_vec_dot_q4_K_q8_K / q5_K_q8_K (ggml-cpu/quants.c; the ggml-cpu contract of MUL_MAT,
 * ggml-cpu.c:1196-1388) with the integer sub-block scales folded into the int8 MFMA operands, so there is one
 * f32 rescale per 256 weights instead of one per 32.  dst[c*dst_col_stride + r]. */
MI355_API int mi355_mul_mat_skinny_k(int type, const void *W, int64_t w_row_stride, const int8_t *act_q, const float *act_d,
                                     const int16_t *act_bsums, float *dst, int64_t dst_col_stride, int64_t m, int64_t k,
                                     int64_t n, void *stream);
/* same for m rows of a matrix of m_total rows (row split, ggml-cuda.cu:1431-1749): the launch geometry that fixes the order of the f32
 * additions over K follows the whole matrix, so a slice's results are the bits the unsplit call gives those rows. */
MI355_API int mi355_mul_mat_skinny_k_slice(int type, const void *W, int64_t w_row_stride, const int8_t *act_q, const float *act_d,
                                           const int16_t *act_bsums, float *dst, int64_t dst_col_stride, int64_t m, int64_t m_total,
                                           int64_t k, int64_t n, void *stream);

/* Up to four weight matrices of ONE K-quant type and one k against the SAME q8_K activations in one launch: the wq / wk / wv
 * products of a step of 5..64 tokens (llama-graph.cpp build_attn's three ggml_mul_mat on one input).  Each result is what
 * mi355_mul_mat_skinny_k writes for that matrix to within the order of the f32 additions over K (here always one workgroup per 32
 * rows and a fixed order; the single call may cut K over two workgroups and add with atomics).  Replaces three launches of
 * mul_mat_q / mul_mat_vec_q (mmq.cuh:3095-3175, mmvq.cu:137-224). */
MI355_API int mi355_mul_mat_skinny_k_group(int type, int n_mats, const void *const *W, const int64_t *w_row_stride, float *const *dst,
                                           const int64_t *dst_col_stride, const int64_t *m, const int8_t *act_q, const float *act_d,
                                           const int16_t *act_bsums, int64_t k, int64_t n, void *stream);

/* MUL_MAT_ID for a few to a few hundred (token, slot) pairs -- the regime of SURVEY config 5 (8x7B, T = 8..64): the
 * pairs are bucketed by expert on the device and every expert's matrix is streamed once per 32 of its columns through
 * the skinny int8-MFMA kernels (Q4_K / Q5_K on q8_K activations, the other types on q8_1).  Same arguments as
 * mi355_mul_mat_id_dq; k % 256 == 0.  Replaces ggml_cuda_mul_mat_id (ggml-cuda.cu:2065-2206) incl. its host-side sort
 * (mmq.cu:141-223).  dst[(t*n_used + s)*m + r]. */
MI355_API int64_t mi355_mul_mat_id_skinny_workspace_bytes(int type, int64_t k, int64_t n_tokens, int64_t b_ne1, int64_t n_used,
                                                          int64_t n_expert);
MI355_API int mi355_mul_mat_id_skinny(int type, const void *as, int64_t w_row_stride, int64_t w_expert_stride, const float *b,
                                      int64_t b_nb1, int64_t b_nb2, const int32_t *ids, int64_t ids_stride, float *dst, int64_t m,
                                      int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1, void *workspace,
                                      int64_t workspace_bytes, void *stream);

/* Two expert tensors of one K-quant type, shape and strides on the same b and ids -- ffn_up_exps and ffn_gate_exps of a MoE layer
 * (build_moe_ffn, llama-graph.cpp:650-795) -- in ONE launch: one activation quantization, one bucketing, one grid over the row tiles of
 * both.  dst0 / dst1 receive exactly what two mi355_mul_mat_id_skinny calls write.  Workspace as for one call.  Q4_K / Q5_K / Q6_K. */
MI355_API int mi355_mul_mat_id_skinny_pair(int type, const void *as0, const void *as1, int64_t w_row_stride, int64_t w_expert_stride,
                                           const float *b, int64_t b_nb1, int64_t b_nb2, const int32_t *ids, int64_t ids_stride,
                                           float *dst0, float *dst1, int64_t m, int64_t k, int64_t n_expert, int64_t n_used,
                                           int64_t n_tokens, int64_t b_ne1, void *workspace, int64_t workspace_bytes, void *stream);
/* The same launches with the two preparatory steps open to the caller (ggml-cuda's MUL_MAT_ID re-sorts the ids on the host and
 * re-quantizes b in every call, ggml-cuda.cu:2065-2206, mmq.cu:141-223): as1 / dst1 may be NULL (one tensor, as mi355_mul_mat_id_skinny) or given (as
 * _pair).  `tables` (mi355_mul_mat_id_tables_bytes, 4-byte aligned; NULL: inside the workspace): where the routing tables of these ids
 * are built -- and kept, so that the next MUL_MAT_ID on the same ids (ffn_down after ffn_up / ffn_gate) can pass
 * MI355_IDS_TABLES_READY and skip the bucketing launch.  MI355_IDS_ACT_READY: the workspace already starts with b's activation
 * planes in the layout of mi355_quantize_act_q8k (K-quants: q at 0, d at up256(rows k), bsums after up256(rows k / 256 * 4)) or
 * mi355_quantize_act (other types) -- written by the launch that produced b (mi355_rms_norm_mul_q8k_f32, mi355_glu_q8k_f32). */
#define MI355_IDS_ACT_READY    1
#define MI355_IDS_TABLES_READY 2
MI355_API int64_t mi355_mul_mat_id_tables_bytes(int64_t n_tokens, int64_t n_used, int64_t n_expert);
MI355_API int mi355_mul_mat_id_skinny_ex(int type, const void *as0, const void *as1, int64_t w_row_stride, int64_t w_expert_stride,
                                         const float *b, int64_t b_nb1, int64_t b_nb2, const int32_t *ids, int64_t ids_stride, float *dst0,
                                         float *dst1, int64_t m, int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1,
                                         void *workspace, int64_t workspace_bytes, int32_t *tables, int flags, void *stream);


/* Prefill MUL_MAT of a K-quant (Q4_K, Q5_K; others MI355_E_UNSUPPORTED) on q8_K activations, any n: the tile-kernel
 * form of mi355_mul_mat_skinny_k (64 rows x 128 columns per workgroup, digit planes prepared once per weight while the
 * tile is staged, one f32 fold per 256 weights).  Replaces mul_mat_q (mmq.cuh:2606-2841) for these types.
 * k % 256 == 0.  dst[c*dst_col_stride + r]. */
MI355_API int mi355_mul_mat_qk(int type, const void *W, int64_t w_row_stride, const int8_t *act_q, const float *act_d,
                               const int16_t *act_bsums, float *dst, int64_t dst_col_stride, int64_t m, int64_t k,
                               int64_t n, void *stream);

/* Prefill tile kernel (int8 MFMA): dst[c][r] = sum_k W[r][k] * act[c][k] for any n (meant for n > 8).
 * Same operands as the mat-vec (one channel).  Returns MI355_E_UNSUPPORTED when k % 128 != 0 (callers
 * then use the mat-vec kernel over column groups).  Replaces ggml_cuda_mul_mat_q / mul_mat_q
 * (ggml/src/ggml-cuda/mmq.cu:68-224, mmq.cuh:2606-2841). */
MI355_API int mi355_mul_mat_q(int type, const void *W, int64_t w_row_stride, const int8_t *act_q, const float *act_ds,
                              float *dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n, void *stream);

/* Prefill GEMM, second form: dequantize-in-LDS + bf16 MFMA (v_mfma_f32_32x32x16_bf16), f32 accumulation.
 * The block scales are applied once per weight while a tile is staged instead of once per 32-block per output
 * element, which is what bounds the int8 kernel above on CDNA4.  Activations are rounded
 * to bf16 once per distinct src1 by mi355_act_bf16 (x: 4-D f32 with byte strides as in mi355_quantize_act; y: bf16
 * [rows][k], 16-byte aligned, k % 8 == 0) -- there is no q8_1 pass on this path.  Same contract and error budget as
 * mi355_mul_mat_q (NMSE <= 5e-4 vs ggml-cpu; measured ~3e-6).  MI355_E_UNSUPPORTED when k % 64 != 0.
 * Replaces the reference's large-batch path (mul_mat_q, or dequantize + hipBLAS GEMM: ggml-cuda.cu:1223-1340).
 *   act_bf16: n columns of k bf16, act_row_stride elements apart (multiple of 8)
 * Tiles: 128 weight rows x 256 columns, 8 multiplying + 4 dequantizing waves, activation tile by LDS-DMA.  A matrix with too few such
 * tiles for the chip (e.g. 4096 rows at n = 512) has K cut in 2 or 4 slices whose partial tiles are kept in a workspace and are
 * added in slice order (a fixed summation order, no atomics).  The workspace is the CALLER's (mi355_mul_mat_dq_slice_ws): no entry
 * point of this library allocates or synchronises.  mi355_mul_mat_dq / mi355_mul_mat_dq_slice take no workspace and return
 * MI355_E_WORKSPACE for a shape whose launch needs one (mi355_mul_mat_dq_workspace_bytes > 0); MI355_MMDQ_KZ=0 in the environment
 * keeps every launch on the forms that need none. */
MI355_API int mi355_act_bf16(const float *x, uint16_t *y, int64_t k, int64_t ne1, int64_t ne2, int64_t ne3,
                             int64_t nb1, int64_t nb2, int64_t nb3, void *stream);
MI355_API int mi355_mul_mat_dq(int type, const void *W, int64_t w_row_stride, const uint16_t *act_bf16, int64_t act_row_stride,
                               float *dst, int64_t dst_col_stride, int64_t m, int64_t k, int64_t n, void *stream);
/* same for a ROW SLICE of a matrix of m_total rows (the row-split path, ggml-cuda.cu:1431-1749): the one choice that changes
 * the f32 summation order (K cut in two for small matrices) is taken from m_total, so every slice produces the bits the
 * unsplit matrix would. */
MI355_API int mi355_mul_mat_dq_slice(int type, const void *W, int64_t w_row_stride, const uint16_t *act_bf16, int64_t act_row_stride,
                                     float *dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n, void *stream);
/* same with the split-K workspace handed in by the caller: mi355_mul_mat_dq_workspace_bytes(m, m_total, k, n) bytes (0 when this shape
 * needs none), 16-byte aligned, private to the stream.  MI355_E_WORKSPACE when the shape needs one and it is NULL or too small. */
MI355_API int64_t mi355_mul_mat_dq_workspace_bytes(int64_t m, int64_t m_total, int64_t k, int64_t n);
MI355_API int mi355_mul_mat_dq_slice_ws(int type, const void *W, int64_t w_row_stride, const uint16_t *act_bf16, int64_t act_row_stride,
                                        float *dst, int64_t dst_col_stride, int64_t m, int64_t m_total, int64_t k, int64_t n,
                                        void *workspace, int64_t workspace_bytes, void *stream);

/* The prefill GEMM stopped before its last step: the K slices of mi355_mul_mat_dq_slice for this shape are left as *kz planes
 * [kz][n][m] of f32 in `parts` (kz == 1: the finished product, one plane) -- for a caller whose next kernel reads the product once
 * anyway and can add the planes on the way (mi355_qkv_finish).  Adding the planes in order, ((p0 + p1) + p2) + p3, gives the bits of
 * mi355_mul_mat_dq_slice.  parts: mi355_mul_mat_dq_parts_bytes(m, m_total, k, n) bytes, 16-byte aligned. */
MI355_API int64_t mi355_mul_mat_dq_parts_bytes(int64_t m, int64_t m_total, int64_t k, int64_t n);
MI355_API int mi355_mul_mat_dq_parts(int type, const void *W, int64_t w_row_stride, const uint16_t *act_bf16, int64_t act_row_stride,
                                     float *parts, int64_t parts_bytes, int64_t m, int64_t m_total, int64_t k, int64_t n,
                                     int32_t *kz, void *stream);

/* mi355_mul_mat_dq_parts for TWO matrices of m rows each and the same activations in ONE launch -- wk and wv of a GQA model (M = 1024
 * at prefill), each of which alone is a split-K launch of small tiles that fills half the chip.  The planes hold the bits of the two
 * separate calls.  Same type, or q4_K / q5_K with q6_K; only where mi355_mul_mat_dq_parts would take its small-tile split form
 * (*kz >= 2), else MI355_E_UNSUPPORTED and the caller issues the two calls.  parts_a / parts_b: parts_bytes_each bytes each. */
MI355_API int mi355_mul_mat_dq_parts_pair(int type_a, const void *Wa, int64_t wa_row_stride, float *parts_a,
                                          int type_b, const void *Wb, int64_t wb_row_stride, float *parts_b, int64_t parts_bytes_each,
                                          const uint16_t *act_bf16, int64_t act_row_stride, int64_t m, int64_t k, int64_t n,
                                          int32_t *kz, void *stream);

/* Q / K / V of a prefill batch after the three GEMMs, in ONE launch: adds the K-slice planes of each product, rotates q and k
 * (ggml_rope_ext, NORMAL or NEOX), stores q as f32 [n][m_q] and k / v as the rows of the KV cache (f16 or f32, n rows one after the
 * other).  Replaces, per layer, 3 slice sums + ROPE x 2 (ggml-cuda/rope.cu) + CPY x 2 (ggml-cuda/cpy.cu; llama-kv-cache-unified.cpp:817-891
 * cpy_k / cpy_v) with the bits those produce one by one. */
typedef struct mi355_qkv_finish_args {
    const float *q_parts, *k_parts, *v_parts;              /* planes [kz][n][m] of mi355_mul_mat_dq_parts */
    int32_t kz_q, kz_k, kz_v;
    int64_t m_q, m_k, m_v, n;                              /* rows of wq / wk / wv, tokens */
    float *q_out;                                          /* [n][m_q] */
    void *k_out, *v_out;                                   /* [n][m_k], [n][m_v] of k_out_type / v_out_type (MI355_TYPE_F16 / F32) */
    int32_t k_out_type, v_out_type;
    const int32_t *pos;                                    /* [n] */
    const float *freq_factors;                             /* [head_dim / 2] or NULL */
    int32_t head_dim, n_dims, mode, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
} mi355_qkv_finish_args;
MI355_API int mi355_qkv_finish(const mi355_qkv_finish_args *args, void *stream);

/* Feed-forward input stage of a prefill batch in ONE launch: dst[n][m] = act(Wa x_n)[m] * (Wb x_n)[m] as f32 and, rounded as
 * mi355_act_bf16 rounds, as bf16 rows ([n][m]) for the ffn_down GEMM.  Replaces MUL_MAT(ffn_gate) MUL_MAT(ffn_up) GLU(split) of
 * build_ffn (src/llama-graph.cpp:470-640) for n_tokens > 128: 2 x the dequantize + GEMM of ggml-cuda.cu:1200-1260, the gated
 * activation (ggml-cuda/unary.cu ggml_cuda_op_swiglu & co) and the bf16 conversion before the next GEMM.  Each workgroup tile
 * holds 64 rows of Wa and the same 64 rows of Wb, so both products of an output element meet in one workgroup; the two f32
 * products are never written.  Offered only for shapes where mi355_mul_mat_dq runs each product as a single K range of
 * 128 x 256 tiles (m even, n >= 256, ceil(n / 256) * ceil(m / 128) >= 200, k % 64 == 0): the result then has the bits of
 * mi355_mul_mat_dq x 2 + mi355_glu_bf16_f32.  Anything else: MI355_E_UNSUPPORTED, and the caller issues the three launches. */
MI355_API int mi355_glu_mul_mat_dq(int type, int glu_op, const void *Wa, int64_t wa_row_stride, const void *Wb, int64_t wb_row_stride,
                                   const uint16_t *act_bf16, int64_t act_row_stride, float *dst, int64_t dst_col_stride,
                                   uint16_t *dst_bf16, int64_t m, int64_t k, int64_t n, void *stream);

/* MUL_MAT_ID (ggml.c:3000-3036) for prefill-sized batches: dst[t][s][:] = as[ids[t][s]] . b[t][s % b_ne1].
 * The (token, slot) pairs are bucketed by expert ON THE DEVICE (one small kernel; the reference synchronises the
 * stream and sorts on the host, ggml-cuda/mmq.cu:141-223) and all experts run in ONE launch of the bf16-MFMA GEMM
 * over gathered columns, so every touched expert matrix is read once.  as: n_expert matrices w_expert_stride bytes
 * apart; b: f32 [n_tokens][b_ne1][k] with byte strides b_nb1 (slot) and b_nb2 (token), 16-byte aligned;
 * ids: device int32, ids_stride elements between tokens; dst: f32 [n_tokens][n_used][m] contiguous.
 * Out-of-range expert ids are skipped (dst untouched).  k % 64 == 0, n_expert <= 256, else MI355_E_UNSUPPORTED / SHAPE. */
MI355_API int64_t mi355_mul_mat_id_dq_workspace_bytes(int64_t k, int64_t n_tokens, int64_t b_ne1, int64_t n_used, int64_t n_expert);
MI355_API int mi355_mul_mat_id_dq(int type, const void *as, int64_t w_row_stride, int64_t w_expert_stride, const float *b,
                                  int64_t b_nb1, int64_t b_nb2, const int32_t *ids, int64_t ids_stride, float *dst, int64_t m,
                                  int64_t k, int64_t n_expert, int64_t n_used, int64_t n_tokens, int64_t b_ne1, void *workspace,
                                  int64_t workspace_bytes, void *stream);

/* Convenience for the common case (2-D contiguous W, contiguous f32 x[n][k], dst[n][m]):
 * quantizes x into `workspace` (>= mi355_mul_mat_workspace_bytes(k, n)) and runs the mat-vec
 * (n <= 8) or the MFMA tile kernel (n > 8).  Two launches on `stream`. */
MI355_API int64_t mi355_mul_mat_workspace_bytes(int64_t k, int64_t n);
MI355_API int mi355_mul_mat(int type, const void *W, int64_t w_row_stride, const float *x, float *dst,
                            int64_t m, int64_t k, int64_t n, void *workspace, int64_t workspace_bytes,
                            void *stream);


/* ---------------------------------------------------------------------------------------------
 * The non-matmul ops of a Llama / Mixtral decode graph, so the reference's
 * scheduler finds nothing to split back to the CPU.  Tensors are described ggml-style: 4 extents
 * (ne[0] fastest) and 4 byte strides; types are ggml_type ids.
 */
typedef struct mi355_tensor {
    void   *data;
    int32_t type;
    int64_t ne[4];
    int64_t nb[4];
} mi355_tensor;

enum mi355_binary_op { MI355_ADD = 0, MI355_MUL = 1, MI355_SUB = 2, MI355_DIV = 3 };

/* dst = a (op) b with ggml broadcasting of b (b.ne[i] divides dst.ne[i]); f32.
 * Replaces ggml_cuda_op_add/mul/sub/div (ggml/src/ggml-cuda/binbcast.cu); oracle
 * ggml_compute_forward_add etc. (ggml/src/ggml-cpu/binary-ops.cpp). */
MI355_API int mi355_binary_f32(int op, const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, void *stream);

/* y = x * rsqrt(mean(x^2) + eps) per row of ne[0]; f32.  ggml_cuda_op_rms_norm (ggml-cuda/norm.cu);
 * oracle ggml_compute_forward_rms_norm_f32 (ggml-cpu/ops.cpp). */
MI355_API int mi355_rms_norm_f32(const mi355_tensor *x, const mi355_tensor *dst, float eps, void *stream);
/* fused RMS_NORM followed by MUL with a per-feature weight vector w[ne0] (the norm weight of every
 * llama layer): dst = (x * rsqrt(mean(x^2) + eps)) * w, same f32 op order as the two separate nodes. */
MI355_API int mi355_rms_norm_mul_f32(const mi355_tensor *x, const float *w, const mi355_tensor *dst, float eps, void *stream);
/* ... and additionally the q8_1 form of dst in the internal activation layout (mi355_quantize_act), so the
 * MUL_MATs that consume dst need no separate quantization launch.  Rows up to 8192 floats, ne0 % 32 == 0,
 * 16-byte aligned; otherwise MI355_E_UNSUPPORTED.  Row r of dst is activation row r. */
MI355_API int mi355_rms_norm_mul_quant_f32(const mi355_tensor *x, const float *w, const mi355_tensor *dst, float eps,
                                           int8_t *act_q, float *act_ds, void *stream);

/* residual ADD feeding the next block's norm, in one launch: sum = a + b (stored, same shape, no broadcast), then
 * dst = (sum * rsqrt(mean(sum^2) + eps)) * w, optionally also its q8_1 form (act_q / act_ds both NULL or both set).
 * Same f32 op order as the three separate graph nodes ADD, RMS_NORM, MUL.  Rows up to 8192 floats, 16-byte aligned,
 * otherwise MI355_E_UNSUPPORTED (callers then run the nodes one by one). */
MI355_API int mi355_add_rms_norm_mul_f32(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *sum, const float *w,
                                         const mi355_tensor *dst, float eps, int8_t *act_q, float *act_ds, void *stream);

/* rotary embedding, modes NORMAL (0) and NEOX (2), optional freq_factors, YaRN parameters.
 * ggml_cuda_op_rope (ggml-cuda/rope.cu); oracle ggml_compute_forward_rope_f32 (ggml-cpu/ops.cpp:5942-6110). */
MI355_API int mi355_rope_f32(const mi355_tensor *x, const int32_t *pos, const float *freq_factors, const mi355_tensor *dst,
                             int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                             float attn_factor, float beta_fast, float beta_slow, void *stream);

/* gated linear unit, split or single-tensor form: dst = act(a) * b.  glu_op: 0 REGLU, 1 GEGLU, 2 SWIGLU
 * (ggml.h:556-564); b may be NULL (then a holds both halves along ne[0]; `swapped` as in ggml).
 * ggml_cuda_op_swiglu (ggml-cuda/unary.cu); oracle ggml_compute_forward_swiglu_f32 (ggml-cpu/ops.cpp:3476-3533). */
MI355_API int mi355_glu_f32(int glu_op, const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, int swapped, void *stream);
/* same, plus the q8_1 form of dst (see mi355_rms_norm_mul_quant_f32); nc % 32 == 0, 16-byte aligned rows */
MI355_API int mi355_glu_quant_f32(int glu_op, const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, int swapped,
                                  int8_t *act_q, float *act_ds, void *stream);
/* Prefill forms of the two fused producers: the f32 result plus its bf16 rows ([rows][ne0] uint16, the rounding of
 * mi355_act_bf16) for the dequantize-in-LDS GEMM that reads it -- what ggml-cuda does with a separate convert kernel
 * before cuBLAS (ggml-cuda.cu:1200-1260).  b / sum NULL: no residual add.  ne0 % 16 == 0 (norm) / % 4 == 0 (glu),
 * rows 16-byte aligned, else MI355_E_UNSUPPORTED. */
MI355_API int mi355_rms_norm_mul_bf16_f32(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *sum, const float *w,
                                          const mi355_tensor *dst, float eps, uint16_t *act_bf16, void *stream);
MI355_API int mi355_glu_bf16_f32(int glu_op, const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, int swapped,
                                 uint16_t *act_bf16, void *stream);
/* the fused norm for batches of 4..32 rows whose consumers are the K-quant skinny kernels: dst = rms_norm(a [+ b]) * w (sum = a + b
 * also stored when b is given) plus the q8_K planes of dst (mi355_quantize_act_q8k layout and arithmetic: act_q int8 [rows][ne0],
 * act_d f32 [rows][ne0 / 256], act_bsums int16 [rows][ne0 / 16]).  ne0 % 256 == 0, rows of at most 8192 values, 16-byte aligned,
 * else MI355_E_UNSUPPORTED.  Replaces a norm launch + a quantization launch (norm.cu + quantize.cu in the reference). */
MI355_API int mi355_rms_norm_mul_q8k_f32(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *sum, const float *w,
                                         const mi355_tensor *dst, float eps, int8_t *act_q, float *act_d, int16_t *act_bsums, void *stream);

/* GLU (mi355_glu_f32's arguments) that also writes the q8_K planes of its result, as mi355_rms_norm_mul_q8k_f32 does for the norm:
 * batches of 5..32 rows, whose ffn_down (ggml_mul_mat, llama-graph.cpp build_ffn) takes mi355_mul_mat_skinny_k.  dst->ne[0] % 256 == 0,
 * 16-byte aligned rows; MI355_E_UNSUPPORTED otherwise.  Planes bit-identical to mi355_quantize_act_q8k(dst). */
MI355_API int mi355_glu_q8k_f32(int glu_op, const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, int swapped,
                                int8_t *act_q, float *act_d, int16_t *act_bsums, void *stream);
/* mi355_rms_norm_mul_bf16_f32 whose first addend is still the kz K-slice planes of the GEMM that produces it (mi355_mul_mat_dq_parts:
 * [kz][rows][ne0] f32, rows and ne0 those of `sum`): sum = ((p0 + p1) + ...) + b, dst = rms_norm(sum) * w, act_bf16 = bf16(dst).
 * MUL_MAT(wo / ffn_down) -> ADD(residual) -> RMS_NORM -> MUL without the slice-sum launch and without the product crossing HBM
 * twice; the bits of the launches run one by one.  Rows of at most 8192 values, ne0 % 16 == 0, else MI355_E_UNSUPPORTED. */
MI355_API int mi355_parts_add_rms_norm_mul_bf16_f32(const float *a_parts, int kz, const mi355_tensor *b, const mi355_tensor *sum,
                                                    const float *w, const mi355_tensor *dst, float eps, uint16_t *act_bf16, void *stream);

/* Reads `bytes` at p (16-byte aligned) with `workgroups` workgroups and drops them: pulls a weight range into the memory-side cache
 * ahead of the launch that streams it, from a second stream beside the previous launch.  No counterpart in the reference; measured
 * in bench.py (BENCH_PREFETCH_MB), see DESIGN.md section 8. */
MI355_API int mi355_prefetch(const void *p, int64_t bytes, int workgroups, void *stream);
/* Measurement aid, not part of the hot path: reads bytes (rounded down to whole KiB) from p once with the weight stream's own
 * instruction (LDS-DMA, non-temporal, 8 KiB in flight per wave, 16 waves per workgroup) and discards them -- the streaming rate
 * this box reaches, which bench.py reports beside the 8 TB/s of the data sheet.  workgroups: the grid (256 = one per CU). */
MI355_API int mi355_stream_read(const void *p, int64_t bytes, int workgroups, void *stream);

/* copy / convert with arbitrary strides and equal element counts (ggml_cpy, ggml_cont, ggml_dup):
 * f32 -> f32 | f16 | bf16 | q8_0 | q4_0 (every KV-cache type FLASH_ATTN_EXT reads), f16 -> f16 | f32.  The q8_0 / q4_0 destinations
 * follow quantize_row_q8_0_ref / quantize_row_q4_0_ref (ggml-quants.c:188-211, 25-60), bf16 ggml_compute_fp32_to_bf16
 * (ggml-impl.h:453-466), bit for bit.  ggml_cuda_cpy (ggml-cuda/cpy.cu); oracle ggml_compute_forward_dup (ggml-cpu/ops.cpp). */
MI355_API int mi355_cpy(const mi355_tensor *src, const mi355_tensor *dst, void *stream);

/* dst[:, i, b2, b3] = dequant(src[:, ids[i, b2, b3], b2, b3]); src f32 | f16 | q4_0 | q8_0 | q4_K | q5_K | q6_K, ids i32.
 * ggml_cuda_op_get_rows (ggml-cuda/getrows.cu); oracle ggml_compute_forward_get_rows (ggml-cpu/ops.cpp). */
MI355_API int mi355_get_rows(const mi355_tensor *src, const mi355_tensor *ids, const mi355_tensor *dst, void *stream);

/* dst[:, ids[i, b2 % ids.ne1, b3 % ids.ne2], b2, b3] = convert(src[:, i, b2, b3]); src f32, ids i64, dst f32 | f16 | bf16 | q8_0 | q4_0 (the
 * KV-cache store of llama.cpp with LLAMA_SET_ROWS, src/llama-kv-cache-unified.cpp:817-892).  Rows with an index outside
 * [0, dst.ne1) are skipped.  Oracle ggml_compute_forward_set_rows_f32 (ggml-cpu/ops.cpp:5179-5227). */
MI355_API int mi355_set_rows(const mi355_tensor *src, const mi355_tensor *ids, const mi355_tensor *dst, void *stream);
/* K and V rows of one decode token, f32 -> q8_0 rows of the KV cache, in one launch (the two CPY f32 -> q8_0 nodes that follow rope / the V
 * projection, ggml-cuda/cpy.cu ggml_cpy_f32_q8_0_cuda; arithmetic: quantize_row_q8_0_ref, ggml-quants.c:188-211, bit-exact).  n_k, n_v:
 * values (multiples of 32).  With kv_table the destinations are read on the device from kv_table[2 * kv_slot] (K) and
 * kv_table[2 * kv_slot + 1] (V) -- the table of mi355_qkv_rope_args, for graph replay -- and k_dst / v_dst are ignored. */
MI355_API int mi355_kv_store_q8_0(const float *k_src, int64_t n_k, void *k_dst, const float *v_src, int64_t n_v, void *v_dst,
                                  void *const *kv_table, int kv_slot, void *stream);
/* the same for a cache of `type` Q8_0 or Q4_0 (quantize_row_q4_0_ref, ggml-quants.c:25-60; ggml_cpy_f32_q4_0_cuda) */
MI355_API int mi355_kv_store_q(int type, const float *k_src, int64_t n_k, void *k_dst, const float *v_src, int64_t n_v, void *v_dst,
                               void *const *kv_table, int kv_slot, void *stream);

/* dst = x * scale + bias (f32, contiguous).  ggml_cuda_op_scale (ggml-cuda/scale.cu). */
MI355_API int mi355_scale_f32(const mi355_tensor *x, const mi355_tensor *dst, float scale, float bias, void *stream);

/* Element-wise maps, f32 -> f32 of the same shape, any byte strides: the UNARY ops in ggml's order (ggml.h:536-553; arithmetic of
 * ggml-cpu/unary-ops.cpp:3-65, vec.h:547-640), SQR / SQRT / SIN / COS / LOG, and CLAMP to [p0, p1] (p0, p1 ignored otherwise).
 * ggml_cuda_op_unary & co. (ggml-cuda/unary.cu), ggml_cuda_op_clamp (ggml-cuda/clamp.cu).  Outside the Llama / Mixtral graphs of
 * what Gemma (TANH for the final softcap), Phi / StableLM / Falcon (GELU, NORM) graphs add to them. */
enum mi355_map_op {
    MI355_MAP_ABS = 0, MI355_MAP_SGN, MI355_MAP_NEG, MI355_MAP_STEP, MI355_MAP_TANH, MI355_MAP_ELU, MI355_MAP_RELU, MI355_MAP_SIGMOID,
    MI355_MAP_GELU, MI355_MAP_GELU_QUICK, MI355_MAP_SILU, MI355_MAP_HARDSWISH, MI355_MAP_HARDSIGMOID, MI355_MAP_EXP, MI355_MAP_GELU_ERF,
    MI355_MAP_SQR, MI355_MAP_SQRT, MI355_MAP_SIN, MI355_MAP_COS, MI355_MAP_LOG, MI355_MAP_CLAMP
};
MI355_API int mi355_map_f32(int op, const mi355_tensor *x, const mi355_tensor *dst, float p0, float p1, void *stream);

/* NORM: dst = (x - mean(x)) / sqrt(var(x) + eps) per row (LayerNorm without the affine part; oracle ggml_compute_forward_norm_f32,
 * ggml-cpu/ops.cpp:3905-3960; ggml_cuda_op_norm, ggml-cuda/norm.cu); l2 != 0: L2_NORM, dst = x / max(|x|_2, eps) (ops.cpp:4314-4358).
 * Rows contiguous. */
MI355_API int mi355_norm_f32(const mi355_tensor *x, const mi355_tensor *dst, float eps, int l2, void *stream);

/* dst (i32, same shape) = indices that sort each row of x (f32, rows of at most 1024 values), ascending or descending;
 * ties by index.  ggml_cuda_op_argsort (ggml-cuda/argsort.cu); oracle ggml_compute_forward_argsort_f32 (ggml-cpu/ops.cpp).
 * ggml_top_k is this op plus a view (ggml.c: ggml_top_k). */
MI355_API int mi355_argsort_f32(const mi355_tensor *x, const mi355_tensor *dst, int descending, void *stream);

/* dst[0, i1, i2, i3] = sum over i0 of x.  ggml_cuda_op_sum_rows (ggml-cuda/sumrows.cu). */
MI355_API int mi355_sum_rows_f32(const mi355_tensor *x, const mi355_tensor *dst, void *stream);

/* MUL_MAT with f32 weights, for small matrices such as the MoE router (ffn_gate_inp, 8 x 4096; llama-graph.cpp:660-676):
 * dst[m, n, i2, i3] = sum_k a[k, m, i2 / r2, i3 / r3] * b[k, n, i2, i3], one wave per output element.  Replaces the cuBLAS
 * branch of ggml_cuda_mul_mat (ggml-cuda.cu:1988-2063) for this case only. */
MI355_API int mi355_mul_mat_f32(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, void *stream);

/* MUL_MAT on float weights, small batches, any strides (rows contiguous): a f32 | f16 | bf16 [k, m, ne02, ne03], b f32 | f16
 * [k, n, ne12, ne13] with ne12 % ne02 == 0 and ne13 % ne03 == 0 (broadcast), dst f32 [m, n, ne12, ne13], f32 accumulation.
 * One pass over the weights per 8 columns; prefill-sized batches of a plain matrix belong on mi355_mul_mat_dq.
 * mul_mat_vec (ggml-cuda/mmv.cu), ggml_cuda_op_mul_mat_cublas (ggml-cuda.cu:1205-1351); oracle ggml_compute_forward_mul_mat
 * with vec_dot_type f16 / bf16 / f32 (ggml-cpu/ggml-cpu.c:1196-1388). */
MI355_API int mi355_mul_mat_f(const mi355_tensor *a, const mi355_tensor *b, const mi355_tensor *dst, void *stream);

/* The router of a MoE layer in one launch, for decode-sized batches (one workgroup per token): logits = gate_inp . x,
 * probs = softmax(logits), order = argsort(probs, descending), w = probs[order[0..n_used)], wsum = sum(w), wn = w / wsum --
 * the chain MUL_MAT SOFT_MAX ARGSORT(top-k) GET_ROWS SUM_ROWS DIV of build_moe_ffn (src/llama-graph.cpp:660-720) with every
 * intermediate written (dense: logits/probs/order [n_tokens][n_expert], w/wn [n_tokens][n_used], wsum [n_tokens]) and each
 * value computed in the order the separate kernels use, i.e. bit-identical to running them one by one.  Row strides in
 * bytes; n_expert <= 256. */
MI355_API int mi355_moe_router_f32(const float *x, int64_t x_row_stride, const float *gate_inp, int64_t gate_row_stride, int64_t k,
                                   int64_t n_expert, int64_t n_used, int64_t n_tokens, float *logits, float *probs, int32_t *order,
                                   float *w, float *wsum, float *wn, void *stream);
/* The same with the layer's ffn norm in front, in the same launch: xn = RMS_NORM(x) * norm_w (dense rows [n_tokens][k], the bits of
 * mi355_rms_norm_mul_f32) and, if asked for, its q8_1 form (act_q / act_ds as mi355_quantize_act writes them: what the experts'
 * mat-vecs read), then the router on xn.  [RMS_NORM, MUL, MUL_MAT, SOFT_MAX, ARGSORT, GET_ROWS, SUM_ROWS, DIV] of a MoE decode step
 * in one launch.  k <= 4096 and k % 32 == 0, else MI355_E_UNSUPPORTED. */
MI355_API int mi355_moe_router_norm_f32(const float *x, int64_t x_row_stride, const float *norm_w, float eps, float *xn, int8_t *act_q,
                                        float *act_ds, const float *gate_inp, int64_t gate_row_stride, int64_t k, int64_t n_expert,
                                        int64_t n_used, int64_t n_tokens, float *logits, float *probs, int32_t *order, float *w,
                                        float *wsum, float *wn, void *stream);
/* The experts' results of a batch combined in one pass: weighted[t][s][:] = e[t][s][:] * w[t][s] and sum[t][:] = weighted[t][0][:] +
 * weighted[t][1][:] -- the MUL and the ADD behind ffn_moe_down for two experts per token (build_moe_ffn, src/llama-graph.cpp:773-790;
 * ggml-cpu binary ops), each value with the same single f32 multiply / add.  e, weighted: dense [n_tokens][2][m] (weighted may be e
 * itself); sum: dense [n_tokens][m]; w: one float at w + t * w_token_stride + s * w_slot_stride (bytes).  m % 4 == 0, else
 * MI355_E_UNSUPPORTED; n_tokens <= 65535. */
MI355_API int mi355_moe_combine2_f32(const float *e, const float *w, int64_t w_token_stride, int64_t w_slot_stride, float *weighted,
                                     float *sum, int64_t m, int64_t n_tokens, void *stream);


/* row softmax with optional f16/f32 mask and ALiBi slope: dst = softmax(x*scale + slope*mask).
 * ggml_cuda_op_soft_max (ggml-cuda/softmax.cu); oracle ggml_compute_forward_soft_max_f32 (ggml-cpu/ops.cpp). */
MI355_API int mi355_soft_max_f32(const mi355_tensor *x, const mi355_tensor *mask, const mi355_tensor *dst,
                                 float scale, float max_bias, void *stream);

/* ---------------------------------------------------------------------------------------------
 * FLASH_ATTN_EXT (ggml.h:2012-2033): dst = softmax(scale * Q K^T [softcap] + slope * mask) V with
 * f16, bf16, q8_0 or q4_0 K/V (K and V of the same type), head size d any multiple of 8 up to 256 (V's may be smaller, see dv; whole 32-blocks
 * for q8_0 / q4_0 rows, else MI355_E_SHAPE) -- the reference's kernels cover 40, 64, 72, 80, 96, 112, 128 and 256 of these.  The
 * kernels are built for 64 / 96 / 128 / 256 features per row; another size runs on the next one up with zeros past the row.
 * Replaces ggml_cuda_flash_attn_ext (ggml/src/ggml-cuda/fattn.cu:271-346); oracle
 * ggml_compute_forward_flash_attn_ext_f16 (ggml/src/ggml-cpu/ops.cpp:7975-8192).
 *   q    f32 [d, n_q, n_head, ne3]      byte strides q_nb1..3 (rows contiguous, 16-byte aligned)
 *   k, v     [d, n_kv, n_head_kv, ne3]  byte strides *_nb1..3 (views of the KV cache)
 *   mask f16 [n_kv, >= n_q, m_ne2, m_ne3] or NULL; -inf entries are skipped
 *   dst  f32 [d, n_head, n_q, ne3]      byte strides d_nb1 (head), d_nb2 (token), d_nb3
 * workspace (optional, 16-byte aligned): lets decode-sized batches split the KV range over many workgroups
 * (mi355_flash_attn_ext_workspace_bytes gives the largest size the launcher can use).  Its first 64 KiB hold the
 * slices' arrival counters: the buffer must be ZERO-FILLED before its first use; every call leaves the counters
 * at zero again.  Calls that share a workspace must be ordered on one stream.
 */
typedef struct mi355_fattn_args {
    const float *q;    int64_t q_nb1, q_nb2, q_nb3;
    const void  *k;    int32_t type_k; int64_t k_nb1, k_nb2, k_nb3;
    const void  *v;    int32_t type_v; int64_t v_nb1, v_nb2, v_nb3;
    const void  *mask; int64_t m_nb1, m_nb2, m_nb3, m_ne2, m_ne3;
    float       *dst;  int64_t d_nb1, d_nb2, d_nb3;
    int64_t d, n_q, n_kv, n_head, n_head_kv, ne3;
    float scale, max_bias, logit_softcap;
    int64_t dv;          /* V's (and dst's) head size where it differs from K's (DeepSeek: d 192, dv 128); 0 = d.  dv <= d. */
} mi355_fattn_args;

MI355_API int64_t mi355_flash_attn_ext_workspace_bytes(const mi355_fattn_args *args);
MI355_API int     mi355_flash_attn_ext(const mi355_fattn_args *args, void *workspace, int64_t workspace_bytes, void *stream);
/* same, and additionally the q8_1 form of dst in the internal activation layout (row = token, k = n_head * d), so
 * the attention-output MUL_MAT needs no quantization launch.  Decode-sized batches only (n_q < 32) and d % 32 == 0 (a 32-block
 * of the row lies inside one head), else MI355_E_UNSUPPORTED.  act_q / act_ds as in mi355_quantize_act. */
MI355_API int     mi355_flash_attn_ext_quant(const mi355_fattn_args *args, void *workspace, int64_t workspace_bytes,
                                             int8_t *act_q, float *act_ds, void *stream);
/* same for prefill-sized batches: additionally the bf16 form of dst (rows [ne3][n_q][n_head * d], the rounding of mi355_act_bf16),
 * so the attention-output GEMM needs no conversion launch.  Only where the matrix-core kernel runs (n_q >= 32, K and V
 * of one type), else MI355_E_UNSUPPORTED. */
MI355_API int     mi355_flash_attn_ext_bf16(const mi355_fattn_args *args, void *workspace, int64_t workspace_bytes,
                                            uint16_t *act_bf16, void *stream);

/* FLASH_ATTN_EXT for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  NO COUNTERPART IN THE REFERENCE (ggml-cuda takes its split-KV buffer from the backend's pool).  Same
 * arguments, arithmetic and results as mi355_flash_attn_ext called with a zeroed workspace of
 * mi355_flash_attn_ext_noalloc_max_bytes() bytes, but the split-KV scratch (arrival counters + partials) is a fixed
 * array inside the library: 64 KiB + 4 MiB per GPU, held until the process ends.  Calls no hipMalloc / hipFree or any
 * allocating routine.  MI355_E_WORKSPACE when mi355_flash_attn_ext_workspace_bytes(args) exceeds the array (e.g.
 * n_q * n_head * (d + 2) * 256 B > 4 MiB: use the workspace form).  All calls on one GPU share the array: calls that
 * overlap on one GPU (two streams) race; order them on one stream.  The device that owns `stream` must be the current device
 * (the array's copy is chosen by the current device; MI355_E_UNSUPPORTED otherwise).  The array's address on a device is looked
 * up on the first call there (hipGetSymbolAddress): a caller whose first call would be under stream capture calls
 * mi355_flash_attn_ext_noalloc_max_bytes() once per device beforehand, which performs the lookup for the current device.
 * Return codes as everywhere in this library: 0, a negative MI355_E_* for refused arguments, a positive hipError_t when a HIP
 * call failed (mi355_error_string handles all three). */
MI355_API int64_t mi355_flash_attn_ext_noalloc_max_bytes(void);
MI355_API int     mi355_flash_attn_ext_noalloc(const mi355_fattn_args *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_KERNELS_H */
