// ops.cpp -- the non-matmul ggml ops the backend serves (see mi355_ops.h): translation from
// ggml_tensor to the plain-C tensor descriptors of include/mi355_kernels.h, and the supports_op
// policy for them.  No kernels here.
#include "mi355_ops.h"

#include <cstring>

#include "mi355_kernels.h"

extern "C" void ggml_log_internal(enum ggml_log_level level, const char * format, ...);

#define MI_CHECK(expr)                                                                               \
    do {                                                                                             \
        int rc_ = (expr);                                                                            \
        if (rc_ != 0) {                                                                              \
            ggml_abort(__FILE__, __LINE__, "mi355 kernel error: %s\n  in %s", mi355_error_string(rc_), #expr); \
        }                                                                                            \
    } while (0)

static mi355_tensor T(const ggml_tensor * t) {
    mi355_tensor r;
    r.data = t->data;
    r.type = (int32_t)t->type;
    for (int i = 0; i < 4; i++) { r.ne[i] = t->ne[i]; r.nb[i] = (int64_t)t->nb[i]; }
    return r;
}

static bool is_f32(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32; }

static bool cpy_types_ok(enum ggml_type s, enum ggml_type d) {
    if (s == GGML_TYPE_F32) return d == GGML_TYPE_F32 || d == GGML_TYPE_F16 || d == GGML_TYPE_BF16 || d == GGML_TYPE_Q8_0 || d == GGML_TYPE_Q4_0;   // the KV cache types FLASH_ATTN_EXT reads
    if (s == GGML_TYPE_F16) return d == GGML_TYPE_F16 || d == GGML_TYPE_F32;
    return false;
}

static bool get_rows_type_ok(enum ggml_type t) {
    switch (t) {
        case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0:
        case GGML_TYPE_Q5_1: case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: case GGML_TYPE_Q6_K:
        case GGML_TYPE_Q2_K: case GGML_TYPE_Q3_K: case GGML_TYPE_IQ4_NL: case GGML_TYPE_IQ4_XS: return true;
        default: return false;
    }
}

bool mi355_ops_supported(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0];
    const ggml_tensor * b = op->src[1];
    switch (op->op) {
        case GGML_OP_ADD:
        case GGML_OP_MUL:
        case GGML_OP_SUB:
        case GGML_OP_DIV:
            return is_f32(a) && is_f32(b) && is_f32(op) && ggml_can_repeat(b, a);
        case GGML_OP_RMS_NORM:
            return is_f32(a) && is_f32(op) && a->nb[0] == sizeof(float) && op->nb[0] == sizeof(float);
        case GGML_OP_ROPE: {
            const int mode = ((const int32_t *)op->op_params)[2];
            return is_f32(a) && is_f32(op) && (mode == 0 || mode == 2) && a->nb[0] == sizeof(float) && op->nb[0] == sizeof(float) &&
                   b && b->type == GGML_TYPE_I32 && ggml_is_contiguous(b) && (!op->src[2] || (is_f32(op->src[2]) && ggml_is_contiguous(op->src[2])));
        }
        case GGML_OP_GLU: {
            const int glu = (int)ggml_get_glu_op(op);
            return is_f32(a) && is_f32(op) && (!b || is_f32(b)) && glu <= (int)GGML_GLU_OP_SWIGLU && ggml_is_contiguous_1(a) &&
                   ggml_is_contiguous_1(op) && (!b || ggml_is_contiguous_1(b));
        }
        case GGML_OP_CPY:
            if (!cpy_types_ok(a->type, b->type)) return false;
            if (b->type == GGML_TYPE_Q8_0 || b->type == GGML_TYPE_Q4_0) return a->ne[0] % 32 == 0 && a->nb[0] == sizeof(float) && b->nb[0] == ggml_type_size(b->type);
            return true;
        case GGML_OP_CONT:
        case GGML_OP_DUP:
            return cpy_types_ok(a->type, op->type) && op->type != GGML_TYPE_Q8_0 && op->type != GGML_TYPE_Q4_0;
        case GGML_OP_GET_ROWS:
            return get_rows_type_ok(a->type) && b->type == GGML_TYPE_I32 && is_f32(op) && a->nb[0] == ggml_type_size(a->type);
        case GGML_OP_SET_ROWS:
            // the KV-cache store with LLAMA_SET_ROWS (src/llama-kv-cache-unified.cpp:817-892): f32 rows scattered by an i64 index
            return is_f32(a) && b && b->type == GGML_TYPE_I64 && a->nb[0] == sizeof(float) && op->nb[0] == ggml_type_size(op->type) &&
                   (op->type == GGML_TYPE_F32 || op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_BF16 ||
                    ((op->type == GGML_TYPE_Q8_0 || op->type == GGML_TYPE_Q4_0) && a->ne[0] % 32 == 0)) &&
                   ((b->nb[0] | b->nb[1] | b->nb[2]) & 7) == 0;
        case GGML_OP_SCALE:
            return is_f32(a) && is_f32(op) && ggml_is_contiguous(a) && ggml_is_contiguous(op);
        case GGML_OP_UNARY:
            return is_f32(a) && is_f32(op) && (int)ggml_get_unary_op(op) <= (int)GGML_UNARY_OP_GELU_ERF;
        case GGML_OP_SQR: case GGML_OP_SQRT: case GGML_OP_SIN: case GGML_OP_COS: case GGML_OP_LOG: case GGML_OP_CLAMP:
            return is_f32(a) && is_f32(op);
        case GGML_OP_NORM: case GGML_OP_L2_NORM:
            return is_f32(a) && is_f32(op) && a->nb[0] == sizeof(float) && op->nb[0] == sizeof(float);
        case GGML_OP_SOFT_MAX:
            return is_f32(a) && is_f32(op) && a->nb[0] == sizeof(float) && ggml_is_contiguous(op) &&
                   (!b || ((b->type == GGML_TYPE_F16 || b->type == GGML_TYPE_F32) && b->nb[0] == ggml_type_size(b->type)));
        case GGML_OP_ARGSORT:
            return is_f32(a) && op->type == GGML_TYPE_I32 && a->ne[0] <= 1024 && a->nb[0] == sizeof(float) && op->nb[0] == sizeof(int32_t);
        case GGML_OP_SUM_ROWS:
            return is_f32(a) && is_f32(op);
        default:
            return false;
    }
}

bool mi355_ops_forward(ggml_tensor * node, hipStream_t stream) {
    const ggml_tensor * a = node->src[0];
    const ggml_tensor * b = node->src[1];
    mi355_tensor td = T(node);
    switch (node->op) {
        case GGML_OP_ADD:
        case GGML_OP_MUL:
        case GGML_OP_SUB:
        case GGML_OP_DIV: {
            const int op = node->op == GGML_OP_ADD ? MI355_ADD : node->op == GGML_OP_MUL ? MI355_MUL : node->op == GGML_OP_SUB ? MI355_SUB : MI355_DIV;
            mi355_tensor ta = T(a), tb = T(b);
            MI_CHECK(mi355_binary_f32(op, &ta, &tb, &td, stream));
            return true;
        }
        case GGML_OP_RMS_NORM: {
            float eps;
            memcpy(&eps, node->op_params, sizeof(float));
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_rms_norm_f32(&ta, &td, eps, stream));
            return true;
        }
        case GGML_OP_ROPE: {
            const int32_t * p = (const int32_t *)node->op_params;
            float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
            memcpy(&freq_base, p + 5, 4); memcpy(&freq_scale, p + 6, 4); memcpy(&ext_factor, p + 7, 4);
            memcpy(&attn_factor, p + 8, 4); memcpy(&beta_fast, p + 9, 4); memcpy(&beta_slow, p + 10, 4);
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_rope_f32(&ta, (const int32_t *)b->data, node->src[2] ? (const float *)node->src[2]->data : nullptr, &td,
                                    p[1], p[2], p[4], freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow, stream));
            return true;
        }
        case GGML_OP_GLU: {
            mi355_tensor ta = T(a), tb;
            if (b) tb = T(b);
            MI_CHECK(mi355_glu_f32((int)ggml_get_glu_op(node), &ta, b ? &tb : nullptr, &td, ((const int32_t *)node->op_params)[1], stream));
            return true;
        }
        case GGML_OP_CPY: {
            mi355_tensor ta = T(a), tb = T(b);
            MI_CHECK(mi355_cpy(&ta, &tb, stream));
            return true;
        }
        case GGML_OP_CONT:
        case GGML_OP_DUP: {
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_cpy(&ta, &td, stream));
            return true;
        }
        case GGML_OP_GET_ROWS: {
            mi355_tensor ta = T(a), tb = T(b);
            MI_CHECK(mi355_get_rows(&ta, &tb, &td, stream));
            return true;
        }
        case GGML_OP_SET_ROWS: {
            mi355_tensor ta = T(a), tb = T(b);
            MI_CHECK(mi355_set_rows(&ta, &tb, &td, stream));
            return true;
        }
        case GGML_OP_UNARY: case GGML_OP_SQR: case GGML_OP_SQRT: case GGML_OP_SIN: case GGML_OP_COS: case GGML_OP_LOG: case GGML_OP_CLAMP: {
            // the UNARY ops map one to one (mi355_map_op follows ggml_unary_op's order, checked below); CLAMP's bounds are op_params[0..1]
            static_assert((int)GGML_UNARY_OP_ABS == (int)MI355_MAP_ABS && (int)GGML_UNARY_OP_TANH == (int)MI355_MAP_TANH &&
                          (int)GGML_UNARY_OP_SILU == (int)MI355_MAP_SILU && (int)GGML_UNARY_OP_GELU_ERF == (int)MI355_MAP_GELU_ERF, "unary op order");
            int mop; float p0 = 0.0f, p1 = 0.0f;
            switch (node->op) {
                case GGML_OP_UNARY: mop = (int)ggml_get_unary_op(node); break;
                case GGML_OP_SQR:   mop = MI355_MAP_SQR;  break;
                case GGML_OP_SQRT:  mop = MI355_MAP_SQRT; break;
                case GGML_OP_SIN:   mop = MI355_MAP_SIN;  break;
                case GGML_OP_COS:   mop = MI355_MAP_COS;  break;
                case GGML_OP_LOG:   mop = MI355_MAP_LOG;  break;
                default:            mop = MI355_MAP_CLAMP; memcpy(&p0, (const float *)node->op_params + 0, 4); memcpy(&p1, (const float *)node->op_params + 1, 4); break;
            }
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_map_f32(mop, &ta, &td, p0, p1, stream));
            return true;
        }
        case GGML_OP_NORM: case GGML_OP_L2_NORM: {
            float eps;
            memcpy(&eps, node->op_params, sizeof(float));
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_norm_f32(&ta, &td, eps, node->op == GGML_OP_L2_NORM, stream));
            return true;
        }
        case GGML_OP_SCALE: {
            float scale; // ggml_scale_impl stores the factor only (ggml.c:3068-3083)
            memcpy(&scale, (const float *)node->op_params + 0, 4);
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_scale_f32(&ta, &td, scale, 0.0f, stream));
            return true;
        }
        case GGML_OP_SOFT_MAX: {
            float scale, max_bias;
            memcpy(&scale, (const float *)node->op_params + 0, 4);
            memcpy(&max_bias, (const float *)node->op_params + 1, 4);
            mi355_tensor ta = T(a), tb;
            if (b) tb = T(b);
            MI_CHECK(mi355_soft_max_f32(&ta, b ? &tb : nullptr, &td, scale, max_bias, stream));
            return true;
        }
        case GGML_OP_ARGSORT: {
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_argsort_f32(&ta, &td, ((const int32_t *)node->op_params)[0] == (int32_t)GGML_SORT_ORDER_DESC, stream));
            return true;
        }
        case GGML_OP_SUM_ROWS: {
            mi355_tensor ta = T(a);
            MI_CHECK(mi355_sum_rows_f32(&ta, &td, stream));
            return true;
        }
        default:
            return false;
    }
}
