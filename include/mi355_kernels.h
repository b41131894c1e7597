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
    MI355_TYPE_Q8_0 = 8,
    MI355_TYPE_Q8_1 = 9,
    MI355_TYPE_Q4_K = 12,
    MI355_TYPE_Q5_K = 13,
    MI355_TYPE_Q6_K = 14,
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

/* library / device introspection */
MI355_API int         mi355_abi_version(void);          /* bumps on any signature change */
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
} mi355_mmvq_args;

MI355_API int mi355_mul_mat_vec_q(const mi355_mmvq_args *args, void *stream);

/* Convenience for the common case (2-D contiguous W, contiguous f32 x[n][k], dst[n][m]):
 * quantizes x into `workspace` (>= mi355_mul_mat_workspace_bytes(k, n)) and runs the mat-vec
 * (n <= 8) or the MFMA tile kernel (n > 8).  Two launches on `stream`. */
MI355_API int64_t mi355_mul_mat_workspace_bytes(int64_t k, int64_t n);
MI355_API int mi355_mul_mat(int type, const void *W, int64_t w_row_stride, const float *x, float *dst,
                            int64_t m, int64_t k, int64_t n, void *workspace, int64_t workspace_bytes,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_KERNELS_H */
