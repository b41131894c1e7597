"""ctypes binding of lib/libmi355_kernels.so (C ABI: include/mi355_kernels.h).

Product path: device pointers in, kernels launched on the caller's HIP stream.  There is NO
fallback of any kind -- a missing or unloadable library raises ImportError here, and every non-zero
return code raises Mi355Error.  torch is used by callers only for device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355_kernels.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is not built: run `make -C llm_amd/csrc` (or __graft_entry__.build()); "
                      "there is no CPU fallback for the MI355X kernels")
lib = C.CDLL(LIB_PATH)

_vp, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32

ABI_VERSION = 6          # MI355_ABI_VERSION of include/mi355_kernels.h this binding mirrors

EXPORTS = [
    "mi355_abi_version", "mi355_args_sizeof", "mi355_error_string", "mi355_blck_size", "mi355_type_size", "mi355_row_size",
    "mi355_quantize_row_q8_1", "mi355_act_q_bytes", "mi355_act_ds_bytes", "mi355_quantize_act",
    "mi355_timing_enable", "mi355_timing_reset", "mi355_timing_read", "mi355_timing_report",
    "mi355_binary_f32", "mi355_rms_norm_f32", "mi355_rms_norm_mul_f32", "mi355_rms_norm_mul_quant_f32", "mi355_glu_quant_f32", "mi355_rms_norm_mul_bf16_f32", "mi355_glu_bf16_f32", "mi355_prefetch", "mi355_parts_add_rms_norm_mul_bf16_f32", "mi355_rms_norm_mul_q8k_f32", "mi355_glu_q8k_f32", "mi355_add_rms_norm_mul_f32", "mi355_mul_mat_vec_q_group", "mi355_qkv_rope_matvec", "mi355_glu_matvec", "mi355_moe_down_matvec", "mi355_rope_f32", "mi355_glu_f32", "mi355_cpy", "mi355_get_rows", "mi355_set_rows", "mi355_mul_mat_f",
    "mi355_scale_f32", "mi355_soft_max_f32", "mi355_flash_attn_ext_workspace_bytes", "mi355_flash_attn_ext", "mi355_flash_attn_ext_quant", "mi355_flash_attn_ext_bf16", "mi355_flash_attn_ext_noalloc_max_bytes", "mi355_flash_attn_ext_noalloc",
    "mi355_dequantize_row", "mi355_mul_mat_vec_q", "mi355_mul_mat_q", "mi355_mul_mat_skinny", "mi355_mul_mat_skinny_slice", "mi355_quantize_act_q8k", "mi355_mul_mat_skinny_k", "mi355_mul_mat_skinny_k_slice", "mi355_mul_mat_skinny_k_group", "mi355_mul_mat_qk", "mi355_mul_mat_id_skinny_workspace_bytes", "mi355_mul_mat_id_skinny", "mi355_mul_mat_id_skinny_pair", "mi355_argsort_f32", "mi355_moe_router_f32", "mi355_sum_rows_f32", "mi355_mul_mat_f32", "mi355_mul_mat_dq", "mi355_mul_mat_dq_slice", "mi355_kv_store_q8_0", "mi355_mul_mat_dq_slice_ws", "mi355_glu_mul_mat_dq", "mi355_mul_mat_dq_parts_bytes", "mi355_mul_mat_dq_parts", "mi355_mul_mat_dq_parts_pair", "mi355_qkv_finish", "mi355_mul_mat_dq_workspace_bytes", "mi355_act_bf16", "mi355_mul_mat_id_dq_workspace_bytes", "mi355_mul_mat_id_dq", "mi355_mul_mat_workspace_bytes", "mi355_mul_mat",
    "mi355_decode_layer_sync_bytes", "mi355_decode_layer", "mi355_stream_read", "mi355_map_f32", "mi355_norm_f32",
    "mi355_mul_mat_id_tables_bytes", "mi355_mul_mat_id_skinny_ex", "mi355_moe_router_norm_f32", "mi355_kv_store_q", "mi355_moe_combine2_f32",
]


class MmvqArgs(C.Structure):
    _fields_ = [
        ("type", _i32), ("W", _vp), ("w_row_stride", _i64), ("w_ch_stride", _i64),
        ("act_q", _vp), ("act_ds", _vp), ("act_ch_stride", _i64),
        ("dst", _vp), ("dst_col_stride", _i64), ("dst_ch_stride", _i64),
        ("m", _i64), ("k", _i64), ("ncols", _i32), ("nchannels", _i32), ("ch_ratio", _i32),
        ("ids", _vp), ("ids_stride", _i32), ("ids_nused", _i32), ("act_ne1", _i32), ("n_expert", _i32),
        ("residual", _vp), ("sum", _vp),
        ("src_f32", _vp), ("norm_w", _vp), ("norm_out", _vp), ("norm_eps", C.c_float),
    ]


class Tensor(C.Structure):
    """mi355_tensor: data, ggml type id, 4 extents (ne[0] fastest), 4 byte strides"""
    _fields_ = [("data", _vp), ("type", _i32), ("ne", _i64 * 4), ("nb", _i64 * 4)]

    @staticmethod
    def of(t, type_id=0):
        """describe a contiguous torch tensor (up to 4-D, last dim fastest) ggml-style"""
        shape = list(t.shape)[::-1] + [1] * (4 - t.dim())
        r = Tensor()
        r.data, r.type = t.data_ptr(), type_id
        nb = t.element_size()
        for i in range(4):
            r.ne[i], r.nb[i] = shape[i], nb
            nb *= shape[i]
        return r


class QkvRopeArgs(C.Structure):
    """mi355_qkv_rope_args (include/mi355_kernels.h)"""
    _fields_ = [
        ("type_qk", _i32), ("type_v", _i32), ("wq", _vp), ("wk", _vp), ("wv", _vp),
        ("wq_row_stride", _i64), ("wk_row_stride", _i64), ("wv_row_stride", _i64),
        ("m_q", _i64), ("m_k", _i64), ("m_v", _i64), ("k", _i64),
        ("act_q", _vp), ("act_ds", _vp), ("q_out", _vp), ("k_out", _vp), ("v_out", _vp),
        ("k_out_type", _i32), ("v_out_type", _i32), ("pos", _vp), ("freq_factors", _vp),
        ("head_dim", _i32), ("n_dims", _i32), ("mode", _i32), ("n_ctx_orig", _i32),
        ("freq_base", C.c_float), ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float),
        ("beta_fast", C.c_float), ("beta_slow", C.c_float), ("type_k", C.c_int32),
        ("norm_x", _vp), ("norm_w", _vp), ("norm_out", _vp), ("norm_eps", C.c_float),
        ("kv_table", _vp), ("kv_slot", _i32),
    ]


class GluMatvecArgs(C.Structure):
    """mi355_glu_matvec_args (include/mi355_kernels.h)"""
    _fields_ = [
        ("type", _i32), ("glu_op", _i32), ("w_gate", _vp), ("w_up", _vp), ("gate_row_stride", _i64), ("up_row_stride", _i64),
        ("m", _i64), ("k", _i64), ("act_q", _vp), ("act_ds", _vp), ("gate_out", _vp), ("up_out", _vp), ("dst", _vp),
        ("out_q", _vp), ("out_ds", _vp), ("norm_x", _vp), ("norm_w", _vp), ("norm_out", _vp), ("norm_eps", C.c_float),
        ("ids", _vp), ("n_slots", _i32), ("n_expert", _i32), ("gate_expert_stride", _i64), ("up_expert_stride", _i64),
    ]


class MoeDownArgs(C.Structure):
    """mi355_moe_down_args (include/mi355_kernels.h)"""
    _fields_ = [
        ("type", _i32), ("n_slots", _i32), ("n_expert", _i32), ("W", _vp), ("w_row_stride", _i64), ("w_expert_stride", _i64),
        ("m", _i64), ("k", _i64), ("act_q", _vp), ("act_ds", _vp), ("ids", _vp), ("weights", _vp),
        ("out", _vp), ("weighted_out", _vp), ("sum_out", _vp), ("residual", _vp), ("residual_out", _vp),
    ]


class FattnArgs(C.Structure):
    """mi355_fattn_args (include/mi355_kernels.h)"""
    _fields_ = [
        ("q", _vp), ("q_nb1", _i64), ("q_nb2", _i64), ("q_nb3", _i64),
        ("k", _vp), ("type_k", _i32), ("k_nb1", _i64), ("k_nb2", _i64), ("k_nb3", _i64),
        ("v", _vp), ("type_v", _i32), ("v_nb1", _i64), ("v_nb2", _i64), ("v_nb3", _i64),
        ("mask", _vp), ("m_nb1", _i64), ("m_nb2", _i64), ("m_nb3", _i64), ("m_ne2", _i64), ("m_ne3", _i64),
        ("dst", _vp), ("d_nb1", _i64), ("d_nb2", _i64), ("d_nb3", _i64),
        ("d", _i64), ("n_q", _i64), ("n_kv", _i64), ("n_head", _i64), ("n_head_kv", _i64), ("ne3", _i64),
        ("scale", C.c_float), ("max_bias", C.c_float), ("logit_softcap", C.c_float),
        ("dv", _i64),
    ]


class QkvFinishArgs(C.Structure):
    """mi355_qkv_finish_args (include/mi355_kernels.h)"""
    _fields_ = [
        ("q_parts", _vp), ("k_parts", _vp), ("v_parts", _vp), ("kz_q", _i32), ("kz_k", _i32), ("kz_v", _i32),
        ("m_q", _i64), ("m_k", _i64), ("m_v", _i64), ("n", _i64), ("q_out", _vp), ("k_out", _vp), ("v_out", _vp),
        ("k_out_type", _i32), ("v_out_type", _i32), ("pos", _vp), ("freq_factors", _vp),
        ("head_dim", _i32), ("n_dims", _i32), ("mode", _i32), ("n_ctx_orig", _i32),
        ("freq_base", C.c_float), ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float),
        ("beta_fast", C.c_float), ("beta_slow", C.c_float),
    ]


class DecodeLayerArgs(C.Structure):
    """mi355_decode_layer_args (include/mi355_kernels.h)"""
    _fields_ = [
        ("type_o", _i32), ("type_gu", _i32), ("type_d", _i32), ("glu_op", _i32), ("tail", _i32),
        ("n_embd", _i64), ("k_o", _i64), ("n_ff", _i64),
        ("wo", _vp), ("wo_row_stride", _i64),
        ("attn_q", _vp), ("attn_ds", _vp), ("attn_f32", _vp),
        ("residual", _vp), ("attn_out", _vp), ("ffn_inp", _vp),
        ("ffn_norm_w", _vp), ("ffn_norm_out", _vp), ("norm_eps", C.c_float),
        ("w_gate", _vp), ("w_up", _vp), ("gu_row_stride", _i64),
        ("gate_out", _vp), ("up_out", _vp), ("hidden", _vp),
        ("w_down", _vp), ("down_row_stride", _i64),
        ("down_out", _vp), ("l_out", _vp),
        ("qkv", QkvRopeArgs), ("out", MmvqArgs),
    ]


TAIL_NONE, TAIL_QKV, TAIL_ROWS = 0, 1, 2

lib.mi355_abi_version.restype = C.c_int
lib.mi355_args_sizeof.restype = _i64
lib.mi355_args_sizeof.argtypes = [C.c_int]
if lib.mi355_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} has ABI version {lib.mi355_abi_version()}, this binding mirrors version {ABI_VERSION}: rebuild")
lib.mi355_error_string.restype = C.c_char_p
lib.mi355_error_string.argtypes = [C.c_int]
for _n in ("mi355_blck_size", "mi355_type_size"):
    getattr(lib, _n).restype = _i64
    getattr(lib, _n).argtypes = [C.c_int]
lib.mi355_row_size.restype = _i64
lib.mi355_row_size.argtypes = [C.c_int, _i64]
for _n in ("mi355_act_q_bytes", "mi355_act_ds_bytes", "mi355_mul_mat_workspace_bytes"):
    getattr(lib, _n).restype = _i64
    getattr(lib, _n).argtypes = [_i64, _i64]
lib.mi355_quantize_row_q8_1.argtypes = [_vp, _vp] + [_i64] * 7 + [_vp]
lib.mi355_quantize_act.argtypes = [_vp, _vp, _vp] + [_i64] * 7 + [_vp]
lib.mi355_dequantize_row.argtypes = [C.c_int, _vp, _vp, _i64, _vp]
lib.mi355_mul_mat_vec_q.argtypes = [C.POINTER(MmvqArgs), _vp]
lib.mi355_flash_attn_ext_workspace_bytes.restype = _i64
lib.mi355_flash_attn_ext_noalloc_max_bytes.restype = _i64
lib.mi355_flash_attn_ext_noalloc_max_bytes.argtypes = []
lib.mi355_flash_attn_ext_noalloc.argtypes = [C.POINTER(FattnArgs), _vp]
lib.mi355_flash_attn_ext_workspace_bytes.argtypes = [C.POINTER(FattnArgs)]
lib.mi355_flash_attn_ext.argtypes = [C.POINTER(FattnArgs), _vp, _i64, _vp]
lib.mi355_flash_attn_ext_quant.argtypes = [C.POINTER(FattnArgs), _vp, _i64, _vp, _vp, _vp]
lib.mi355_flash_attn_ext_bf16.argtypes = [C.POINTER(FattnArgs), _vp, _i64, _vp, _vp]
lib.mi355_mul_mat_id_skinny_workspace_bytes.restype = _i64
lib.mi355_mul_mat_id_skinny_workspace_bytes.argtypes = [C.c_int] + [_i64] * 5
lib.mi355_mul_mat_id_skinny.argtypes = [C.c_int, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp] + [_i64] * 6 + [_vp, _i64, _vp]
lib.mi355_mul_mat_id_skinny_pair.argtypes = [C.c_int, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp] + [_i64] * 6 + [_vp, _i64, _vp]
lib.mi355_mul_mat_id_skinny_ex.argtypes = [C.c_int, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp] + [_i64] * 6 + [_vp, _i64, _vp, C.c_int, _vp]
lib.mi355_mul_mat_id_tables_bytes.restype = _i64
lib.mi355_moe_router_f32.argtypes = [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64] + [_vp] * 7
lib.mi355_moe_combine2_f32.argtypes = [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp]
lib.mi355_moe_router_norm_f32.argtypes = [_vp, _i64, _vp, C.c_float, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64] + [_vp] * 7
lib.mi355_mul_mat_id_tables_bytes.argtypes = [_i64] * 3
lib.mi355_mul_mat_id_dq_workspace_bytes.restype = _i64
lib.mi355_mul_mat_id_dq_workspace_bytes.argtypes = [_i64] * 5
lib.mi355_mul_mat_id_dq.argtypes = [C.c_int, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp] + [_i64] * 6 + [_vp, _i64, _vp]
lib.mi355_act_bf16.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp]
lib.mi355_mul_mat_dq_slice.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp]
lib.mi355_mul_mat_dq_parts_bytes.restype = _i64
lib.mi355_mul_mat_dq_parts_bytes.argtypes = [C.c_int64] * 4
lib.mi355_mul_mat_dq_parts.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), _vp]
lib.mi355_mul_mat_dq_parts_pair.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), _vp]
lib.mi355_qkv_finish.argtypes = [C.POINTER(QkvFinishArgs), _vp]
lib.mi355_glu_mul_mat_dq.argtypes = [C.c_int, C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, _vp]
lib.mi355_mul_mat_dq_slice_ws.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int64, _vp]
lib.mi355_mul_mat_dq_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64]
lib.mi355_mul_mat_dq_workspace_bytes.restype = C.c_int64
lib.mi355_kv_store_q8_0.argtypes = [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, _vp, C.c_int, _vp]
lib.mi355_mul_mat_dq.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp]
_T = C.POINTER(Tensor)
lib.mi355_rms_norm_f32.argtypes = [_T, _T, C.c_float, _vp]
lib.mi355_rms_norm_mul_f32.argtypes = [_T, _vp, _T, C.c_float, _vp]
lib.mi355_rms_norm_mul_quant_f32.argtypes = [_T, _vp, _T, C.c_float, _vp, _vp, _vp]
lib.mi355_add_rms_norm_mul_f32.argtypes = [_T, _T, _T, _vp, _T, C.c_float, _vp, _vp, _vp]
lib.mi355_scale_f32.argtypes = [_T, _T, C.c_float, C.c_float, _vp]
lib.mi355_map_f32.argtypes = [C.c_int, _T, _T, C.c_float, C.c_float, _vp]
lib.mi355_norm_f32.argtypes = [_T, _T, C.c_float, C.c_int, _vp]
lib.mi355_binary_f32.argtypes = [C.c_int, _T, _T, _T, _vp]
lib.mi355_glu_f32.argtypes = [C.c_int, _T, _T, _T, C.c_int, _vp]
lib.mi355_glu_quant_f32.argtypes = [C.c_int, _T, _T, _T, C.c_int, _vp, _vp, _vp]
lib.mi355_rms_norm_mul_q8k_f32.argtypes = [_T, _T, _T, _vp, _T, C.c_float, _vp, _vp, _vp, _vp]
lib.mi355_glu_q8k_f32.argtypes = [C.c_int, _T, _T, _T, C.c_int, _vp, _vp, _vp, _vp]
lib.mi355_parts_add_rms_norm_mul_bf16_f32.argtypes = [_vp, C.c_int, _T, _T, _vp, _T, C.c_float, _vp, _vp]
lib.mi355_prefetch.argtypes = [_vp, C.c_int64, C.c_int, _vp]
lib.mi355_stream_read.argtypes = [_vp, C.c_int64, C.c_int, _vp]
lib.mi355_glu_bf16_f32.argtypes = [C.c_int, _T, _T, _T, C.c_int, _vp, _vp]
lib.mi355_rms_norm_mul_bf16_f32.argtypes = [_T, _T, _T, _vp, _T, C.c_float, _vp, _vp]
lib.mi355_cpy.argtypes = [_T, _T, _vp]
lib.mi355_set_rows.argtypes = [_T, _T, _T, _vp]
lib.mi355_get_rows.argtypes = [_T, _T, _T, _vp]
lib.mi355_mul_mat_f.argtypes = [_T, _T, _T, _vp]
lib.mi355_rope_f32.argtypes = [C.POINTER(Tensor), _vp, _vp, C.POINTER(Tensor), C.c_int, C.c_int, C.c_int] + [C.c_float] * 6 + [_vp]
lib.mi355_qkv_rope_matvec.argtypes = [C.POINTER(QkvRopeArgs), _vp]
lib.mi355_glu_matvec.argtypes = [C.POINTER(GluMatvecArgs), _vp]
lib.mi355_moe_down_matvec.argtypes = [C.POINTER(MoeDownArgs), _vp]
lib.mi355_mul_mat_vec_q_group.argtypes = [C.POINTER(MmvqArgs), C.c_int, _vp]
lib.mi355_mul_mat_skinny.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]
lib.mi355_quantize_act_q8k.argtypes = [_vp, _vp, _vp, _vp] + [_i64] * 7 + [_vp]
lib.mi355_mul_mat_skinny_k.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]
lib.mi355_mul_mat_skinny_k_slice.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]
lib.mi355_mul_mat_skinny_k_group.argtypes = [C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]
lib.mi355_mul_mat_qk.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]
lib.mi355_mul_mat_q.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]
lib.mi355_mul_mat.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]
lib.mi355_decode_layer_sync_bytes.restype = _i64
lib.mi355_decode_layer_sync_bytes.argtypes = []
lib.mi355_decode_layer.argtypes = [C.POINTER(DecodeLayerArgs), _vp, _vp]


lib.mi355_timing_enable.argtypes = [C.c_int]
lib.mi355_timing_enable.restype = None
lib.mi355_timing_reset.restype = None
lib.mi355_timing_read.argtypes = [C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_double)]
lib.mi355_timing_report.argtypes = [C.c_char_p, _i64]
lib.mi355_timing_report.restype = _i64


def timing_read(tag):
    """-> (launches, total_us) of the kernels launched under `tag` while timing was enabled."""
    n, us = _i64(0), C.c_double(0.0)
    check(lib.mi355_timing_read(tag.encode(), C.byref(n), C.byref(us)), "mi355_timing_read")
    return n.value, us.value


def timing_report():
    need = lib.mi355_timing_report(None, 0)
    buf = C.create_string_buffer(need + 64)
    lib.mi355_timing_report(buf, need + 64)
    out = {}
    for line in buf.value.decode().splitlines():
        tag, n, us = line.split()
        out[tag] = (int(n), float(us))
    return out


class Mi355Error(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: {lib.mi355_error_string(code).decode()} ({code})")


def check(code, where):
    if code != 0:
        raise Mi355Error(code, where)


# the hand-kept ctypes mirrors must have the C structs' sizes (a stale library or binding would read struct tails as garbage)
for _which, _cls in enumerate((MmvqArgs, QkvRopeArgs, GluMatvecArgs, MoeDownArgs, FattnArgs, Tensor, QkvFinishArgs, DecodeLayerArgs)):
    if lib.mi355_args_sizeof(_which) != C.sizeof(_cls):
        raise ImportError(f"{_cls.__name__}: ctypes mirror is {C.sizeof(_cls)} bytes, the library's struct {lib.mi355_args_sizeof(_which)}")


_dq_ws = {}


def mul_mat_dq(t, w, w_row_stride, x, x_row_stride, dst, dst_col_stride, m, k, n, stream, m_total=None):
    """mi355_mul_mat_dq_slice_ws with the split-K workspace owned HERE, on the caller's side (a torch buffer per device, grown
    on demand): the library itself allocates nothing.  Pointers are raw device addresses; returns the library's code."""
    import torch
    m_total = m if m_total is None else m_total
    need = lib.mi355_mul_mat_dq_workspace_bytes(m, m_total, k, n)
    ws = None
    if need > 0:
        dev = torch.cuda.current_device()
        ws = _dq_ws.get(dev)
        if ws is None or ws.numel() < need:
            torch.cuda.synchronize()          # launches in flight may still use the old block
            ws = _dq_ws[dev] = torch.empty(need + (8 << 20), dtype=torch.uint8, device="cuda")
    return lib.mi355_mul_mat_dq_slice_ws(t, w, w_row_stride, x, x_row_stride, dst, dst_col_stride, m, m_total, k, n,
                                         ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream)
