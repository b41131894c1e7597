// mul_mat.hip -- MUL_MAT convenience entry: quantize the f32 activations once, then dispatch on the
// batch size the way ggml_cuda_mul_mat does (ggml/src/ggml-cuda/ggml-cuda.cu:1988-2063): n <= 8
// columns -> mat-vec kernel (HBM-bound), larger batches -> tile kernel.
#include "common.h"

static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

extern "C" int64_t mi355_mul_mat_workspace_bytes(int64_t k, int64_t n) {
    if (k <= 0 || k % QK != 0 || n < 0) return -1;
    return align_up(mi355_act_q_bytes(k, n), 256) + align_up(mi355_act_ds_bytes(k, n), 256);
}

extern "C" int mi355_mul_mat(int type, const void * W, int64_t w_row_stride, const float * x, float * dst,
                             int64_t m, int64_t k, int64_t n, void * workspace, int64_t workspace_bytes,
                             void * stream) {
    if (!W || !x || !dst || !workspace) return MI355_E_NULL;
    const int64_t need = mi355_mul_mat_workspace_bytes(k, n);
    if (need < 0) return MI355_E_SHAPE;
    if (workspace_bytes < need) return MI355_E_WORKSPACE;
    if ((uintptr_t)workspace & 255) return MI355_E_ALIGN;
    if (n == 0 || m == 0) return MI355_OK;
    int8_t * aq  = (int8_t *)workspace;
    float  * ads = (float *)((char *)workspace + align_up(mi355_act_q_bytes(k, n), 256));
    int rc = mi355_quantize_act(x, aq, ads, k, n, 1, 1, k * 4, k * n * 4, k * n * 4, stream);
    if (rc) return rc;
    const bool tile_type = type == MI355_TYPE_Q4_0 || type == MI355_TYPE_Q8_0 || type == MI355_TYPE_Q4_K || type == MI355_TYPE_Q5_K ||
                           type == MI355_TYPE_Q6_K;       // the int8 tile kernel serves these; q4_1 / q5_0 / q5_1 go by column groups
    if (n > 8 && tile_type) {
        rc = mi355_mul_mat_q(type, W, w_row_stride, aq, ads, dst, m, m, k, n, stream);
        if (rc != MI355_E_UNSUPPORTED) return rc;
    }
    // mat-vec kernel over groups of up to 8 columns
    for (int64_t c0 = 0; c0 < n; c0 += 8) {
        mi355_mmvq_args a = {};
        a.type = type; a.W = W; a.w_row_stride = w_row_stride; a.w_ch_stride = 0;
        a.act_q = aq + c0 * k; a.act_ds = ads + c0 * (k / QK) * 2; a.act_ch_stride = 0;
        a.dst = dst + c0 * m; a.dst_col_stride = m; a.dst_ch_stride = 0;
        a.m = m; a.k = k; a.ncols = (int32_t)(n - c0 < 8 ? n - c0 : 8); a.nchannels = 1; a.ch_ratio = 1;
        rc = mi355_mul_mat_vec_q(&a, stream);
        if (rc) return rc;
    }
    return MI355_OK;
}
