// api.hip -- introspection entry points of libmi355_kernels.so (include/mi355_kernels.h).
#include "common.h"

extern "C" {

int mi355_abi_version(void) { return MI355_ABI_VERSION; }

// sizeof() of the argument structs, so that a binding built against another header is caught at load time
int64_t mi355_args_sizeof(int which) {
    switch (which) {
        case 0: return (int64_t)sizeof(mi355_mmvq_args);
        case 1: return (int64_t)sizeof(mi355_qkv_rope_args);
        case 2: return (int64_t)sizeof(mi355_glu_matvec_args);
        case 3: return (int64_t)sizeof(mi355_moe_down_args);
        case 4: return (int64_t)sizeof(mi355_fattn_args);
        case 5: return (int64_t)sizeof(mi355_tensor);
        case 6: return (int64_t)sizeof(mi355_qkv_finish_args);
        case 7: return (int64_t)sizeof(mi355_decode_layer_args);
    }
    return -1;
}

const char * mi355_error_string(int err) {
    switch (err) {
        case MI355_OK:            return "ok";
        case MI355_E_TYPE:        return "unsupported block type";
        case MI355_E_SHAPE:       return "bad shape (k not a multiple of the block size, negative dim, or ncols out of range)";
        case MI355_E_ALIGN:       return "pointer or stride alignment not met";
        case MI355_E_NULL:        return "required pointer is NULL";
        case MI355_E_WORKSPACE:   return "workspace too small";
        case MI355_E_UNSUPPORTED: return "variant not implemented";
    }
    if (err > 0) return hipGetErrorString((hipError_t)err);
    return "unknown error";
}

int64_t mi355_blck_size(int type) {
    switch (type) {
        case MI355_TYPE_F32: case MI355_TYPE_F16: case MI355_TYPE_BF16: return 1;
        case MI355_TYPE_Q4_0: case MI355_TYPE_Q4_1: case MI355_TYPE_Q5_0: case MI355_TYPE_Q5_1: case MI355_TYPE_Q8_0: case MI355_TYPE_Q8_1: case MI355_TYPE_IQ4_NL: return QK;
        case MI355_TYPE_Q4_K: case MI355_TYPE_Q5_K: case MI355_TYPE_Q6_K: case MI355_TYPE_Q2_K: case MI355_TYPE_Q3_K: case MI355_TYPE_IQ4_XS: return QK_K;
    }
    return 0;
}

int64_t mi355_type_size(int type) {
    switch (type) {
        case MI355_TYPE_F32:  return 4;
        case MI355_TYPE_F16:  return 2;
        case MI355_TYPE_BF16: return 2;
        case MI355_TYPE_Q4_0: return 18;
        case MI355_TYPE_Q4_1: return 20;
        case MI355_TYPE_Q5_0: return 22;
        case MI355_TYPE_Q5_1: return 24;
        case MI355_TYPE_Q8_0: return 34;
        case MI355_TYPE_Q8_1: return 36;
        case MI355_TYPE_Q4_K: return 144;
        case MI355_TYPE_Q5_K: return 176;
        case MI355_TYPE_Q6_K: return 210;
        case MI355_TYPE_Q2_K: return 84;
        case MI355_TYPE_Q3_K: return 110;
        case MI355_TYPE_IQ4_NL: return 18;
        case MI355_TYPE_IQ4_XS: return 136;
    }
    return 0;
}

int64_t mi355_row_size(int type, int64_t k) {
    const int64_t bs = mi355_blck_size(type);
    if (bs == 0 || k < 0 || k % bs != 0) return -1;
    return k / bs * mi355_type_size(type);
}

} // extern "C"
