"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference's arithmetic.

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only.  Nothing under llm_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")

F32, F16, Q4_0, Q8_0, Q8_1, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 8, 9, 12, 13, 14, 15
Q4_1, Q5_0, Q5_1, BF16 = 3, 6, 7, 30
Q2_K, Q3_K, IQ4_NL, IQ4_XS = 10, 11, 20, 23
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q8_0: "q8_0", Q8_1: "q8_1", Q4_K: "q4_K",
              Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", BF16: "bf16",
              Q2_K: "q2_K", Q3_K: "q3_K", IQ4_NL: "iq4_nl", IQ4_XS: "iq4_xs"}
WEIGHT_TYPES = (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K)
WEIGHT_TYPES2 = (Q4_1, Q5_0, Q5_1)          # second batch, fixtures in *2.gold
WEIGHT_TYPES3 = (Q2_K, Q3_K, IQ4_NL, IQ4_XS)  # third batch (the remainder), fixtures in *3.gold
FLOAT_WEIGHT_TYPES = (F16, BF16, F32)


def _build():
    src = os.path.join(_ROOT, "oracle", "oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "liboracle"],
                              stdout=subprocess.DEVNULL)


_build()
_lib = C.CDLL(_SO)
_vp, _i64, _f = C.c_void_p, C.c_int64, C.c_float
_lib.orc_blck_size.restype = _i64
_lib.orc_type_size.restype = _i64
_lib.orc_row_size.restype = _i64
_lib.orc_row_size.argtypes = [C.c_int, _i64]
_lib.orc_fp32_to_fp16.restype = C.c_uint16
_lib.orc_fp32_to_fp16.argtypes = [_f]
_lib.orc_fp16_to_fp32.restype = _f
_lib.orc_fp16_to_fp32.argtypes = [C.c_uint16]
for _n in ("q8_0", "q8_1", "q8_K", "q4_0"):
    getattr(_lib, "orc_quantize_row_" + _n).argtypes = [_vp, _vp, _i64]
    getattr(_lib, "orc_quantize_row_" + _n).restype = None
_lib.orc_fp32_to_bf16_row.argtypes = [_vp, _vp, _i64]
_lib.orc_fp32_to_bf16_row.restype = None
_lib.orc_dequantize_row.argtypes = [C.c_int, _vp, _vp, _i64]
_lib.orc_mul_mat_cpu.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _i64, _i64, _i64]
_lib.orc_mul_mat_q8_1.argtypes = [C.c_int, _vp, _i64, _vp, _vp, _i64, _i64, _i64]
_lib.orc_mul_mat_id.argtypes = [C.c_int, _vp, _i64, _i64, _vp, _vp, _vp] + [_i64] * 6 + [C.c_int]
_FA = [_vp] * 5 + [C.c_int, C.c_int] + [_i64] * 6 + [_f] * 3
_lib.orc_flash_attn_ext.argtypes = _FA
_lib.orc_flash_attn_ext_f64.argtypes = _FA
_lib.orc_nmse.restype = C.c_double
_lib.orc_nmse.argtypes = [_vp, _vp, _i64]


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def row_size(t, k):
    r = _lib.orc_row_size(t, k)
    assert r >= 0, (t, k)
    return r


def blck_size(t):
    return _lib.orc_blck_size(t)


def fp32_to_fp16(x):
    return _lib.orc_fp32_to_fp16(float(np.float32(x)))


def quantize_row(t, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(row_size(t, x.size), dtype=np.uint8)
    getattr(_lib, "orc_quantize_row_" + TYPE_NAMES[t])(_p(x), _p(out), x.size)
    return out


def fp32_to_bf16_row(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(x.size, dtype=np.uint16)
    _lib.orc_fp32_to_bf16_row(_p(x), _p(out), x.size)
    return out


def dequantize_row(t, blocks, n):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    assert blocks.size >= row_size(t, n)
    y = np.empty(n, dtype=np.float32)
    assert _lib.orc_dequantize_row(t, _p(blocks), _p(y), n) == 0
    return y


def mul_mat(t, w, x, m, k, n, semantics="cpu"):
    """w: uint8 blocks [m rows]; x: [n, k] f32 -> dst [n, m] f32."""
    w = np.ascontiguousarray(w, dtype=np.uint8)
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(n, k)
    dst = np.empty((n, m), dtype=np.float32)
    fn = _lib.orc_mul_mat_cpu if semantics == "cpu" else _lib.orc_mul_mat_q8_1
    assert fn(t, _p(w), row_size(t, k), _p(x), _p(dst), m, k, n) == 0
    return dst


def mul_mat_id(t, w, b, ids, m, k, n_expert, n_used, n_tokens, b_ne1, semantics="cpu"):
    w = np.ascontiguousarray(w, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    dst = np.empty((n_tokens, n_used, m), dtype=np.float32)
    rs = row_size(t, k)
    rc = _lib.orc_mul_mat_id(t, _p(w), rs, rs * m, _p(b), _p(ids), _p(dst), m, k, n_expert, n_used,
                             n_tokens, b_ne1, 0 if semantics == "cpu" else 1)
    assert rc == 0, rc
    return dst


def flash_attn_ext(q, k, v, mask, type_k, type_v, dk, dv, n_q, n_kv, n_head, n_head_kv, scale,
                   max_bias=0.0, logit_softcap=0.0, exact=False):
    q = np.ascontiguousarray(q, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.uint8)
    v = np.ascontiguousarray(v, dtype=np.uint8)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint16)
    dst = np.empty((n_q, n_head, dv), dtype=np.float32)
    fn = _lib.orc_flash_attn_ext_f64 if exact else _lib.orc_flash_attn_ext
    rc = fn(_p(q), _p(k), _p(v), _p(mask), _p(dst), type_k, type_v, dk, dv, n_q, n_kv, n_head,
            n_head_kv, scale, max_bias, logit_softcap)
    assert rc == 0, rc
    return dst


def nmse(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    b = np.ascontiguousarray(b, dtype=np.float32).ravel()
    assert a.size == b.size
    return _lib.orc_nmse(_p(a), _p(b), a.size)
