"""Pins oracle/liboracle.so (our CPU restatement) to the reference: every fixture in tests/golden/
was produced by the reference's own code (oracle/gen_golden.cpp linked to oracle/_ref).

bit-exact: activation quantizers (*_ref) and block dequantizers.
tolerance: MUL_MAT / MUL_MAT_ID / FLASH_ATTN_EXT of the ggml-cpu *backend*, whose AVX2 kernels and
SIMD activation quantizer differ from the generic/_ref forms we restate at float-rounding level
(the reference's own test only asks RMSE < 1e-4 between them, tests/test-quantize-fns.cpp:58-71).
"""
import numpy as np
import pytest

import oracle_lib as O
from golden_io import load_gold

NAME2T = {v: k for k, v in O.TYPE_NAMES.items()}


@pytest.mark.parametrize("name", ["cos0", "cos1", "u1234", "wide", "edge"])
@pytest.mark.parametrize("t", [O.Q8_0, O.Q8_1, O.Q8_K])
def test_quantize_bit_exact(name, t):
    g = load_gold("quantize.gold")
    got = O.quantize_row(t, g["x." + name])
    want = g[O.TYPE_NAMES[t] + "." + name]
    assert got.size == want.size
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} bytes differ"


@pytest.mark.parametrize("name", ["cos0", "u1234", "wide", "edge"])
def test_kv_store_formats_bit_exact(name):
    """q4_0 blocks (quantize_row_q4_0_ref) and bf16 rows (ggml_fp32_to_bf16_row_ref) -- what CPY / SET_ROWS write into a q4_0 / bf16 KV
    cache -- of the oracle against the reference's own output (quantize2.gold)"""
    g = load_gold("quantize2.gold")
    x = g["x." + name]
    assert np.array_equal(O.quantize_row(O.Q4_0, x), g["q4_0." + name])
    assert np.array_equal(O.fp32_to_bf16_row(x), g["bf16." + name])


@pytest.mark.parametrize("src", ["rand", "quant"])
@pytest.mark.parametrize("t", O.WEIGHT_TYPES)
def test_dequantize_bit_exact(src, t):
    g = load_gold("dequantize.gold")
    tn = O.TYPE_NAMES[t]
    want = g[f"{src}.deq.{tn}"]
    got = O.dequantize_row(t, g[f"{src}.blocks.{tn}"], want.size)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("src", ["rand", "quant"])
@pytest.mark.parametrize("t", O.WEIGHT_TYPES2)
def test_dequantize2_bit_exact(src, t):
    g = load_gold("dequantize2.gold")
    tn = O.TYPE_NAMES[t]
    want = g[f"{src}.deq.{tn}"]
    got = O.dequantize_row(t, g[f"{src}.blocks.{tn}"], want.size)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("src", ["rand", "quant"])
@pytest.mark.parametrize("t", O.WEIGHT_TYPES3)
def test_dequantize3_bit_exact(src, t):
    """q2_K / q3_K / iq4_nl / iq4_xs (ggml-quants.c:706-735, 1050-1099, 2428-2468)"""
    g = load_gold("dequantize3.gold")
    tn = O.TYPE_NAMES[t]
    want = g[f"{src}.deq.{tn}"]
    got = O.dequantize_row(t, g[f"{src}.blocks.{tn}"], want.size)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_fp16_roundtrip_all_values():
    # every finite half: to_fp32 then back is the identity; also spot-check RNE on midpoints
    import ctypes as C
    for h in range(0, 0x10000, 1):
        if (h & 0x7c00) == 0x7c00:
            continue
        f = O._lib.orc_fp16_to_fp32(h)
        assert O._lib.orc_fp32_to_fp16(f) == h
    hs = np.arange(0, 0x7bff, dtype=np.uint16)
    lo = hs.view(np.float16).astype(np.float32)
    hi = (hs + 1).astype(np.uint16).view(np.float16).astype(np.float32)
    mid = ((lo.astype(np.float64) + hi.astype(np.float64)) / 2).astype(np.float32)  # exact in f32
    want = mid.astype(np.float16).view(np.uint16)  # numpy rounds to nearest even
    got = np.array([O.fp32_to_fp16(x) for x in mid[::7]], dtype=np.uint16)
    assert np.array_equal(got, want[::7])


def _mm_cases():
    g = load_gold("mul_mat.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mm_cases())
def test_mul_mat_vs_ggml_cpu(tag):
    g = load_gold("mul_mat.gold")
    t, m, k, n = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n, m)
    got = O.mul_mat(t, g["w." + tag], g["x." + tag], m, k, n, "cpu")
    assert O.nmse(want, got) < 1e-6  # SIMD from_float vs _ref differ in rare blocks
    # the q8_1 contract (what the GPU path computes) sits well inside test-backend-ops' 5e-4
    got81 = O.mul_mat(t, g["w." + tag], g["x." + tag], m, k, n, "q8_1")
    assert O.nmse(want, got81) < 2e-4


def _mm2_cases():
    g = load_gold("mul_mat2.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mm2_cases())
def test_mul_mat2_vs_ggml_cpu(tag):
    """q4_1 / q5_0 / q5_1 and float (f16 / bf16 / f32) weights: the restatement of ggml-cpu's path (activations converted
    to the weight's vec_dot_type, generic vec_dot) against the ggml-cpu backend's own output."""
    g = load_gold("mul_mat2.gold")
    t, m, k, n = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n, m)
    got = O.mul_mat(t, g["w." + tag], g["x." + tag], m, k, n, "cpu")
    assert O.nmse(want, got) < 1e-6
    if t in O.WEIGHT_TYPES2:
        got81 = O.mul_mat(t, g["w." + tag], g["x." + tag], m, k, n, "q8_1")
        assert O.nmse(want, got81) < 2e-4


def _mm3_cases():
    g = load_gold("mul_mat3.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mm3_cases())
def test_mul_mat3_vs_ggml_cpu(tag):
    """q2_K / q3_K / iq4_xs (q8_K activations) and iq4_nl (q8_0): restated generic vec_dot against the ggml-cpu backend"""
    g = load_gold("mul_mat3.gold")
    t, m, k, n = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n, m)
    got = O.mul_mat(t, g["w." + tag], g["x." + tag], m, k, n, "cpu")
    assert O.nmse(want, got) < 1e-6
    got81 = O.mul_mat(t, g["w." + tag], g["x." + tag], m, k, n, "q8_1")
    assert O.nmse(want, got81) < 2e-4


def _mmid_cases():
    g = load_gold("mul_mat_id.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mmid_cases())
def test_mul_mat_id_vs_ggml_cpu(tag):
    g = load_gold("mul_mat_id.gold")
    t, m, k, n_exp, n_used, n_tok, b1 = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n_tok, n_used, m)
    got = O.mul_mat_id(t, g["w." + tag], g["x." + tag], g["ids." + tag], m, k, n_exp, n_used, n_tok, b1)
    assert O.nmse(want, got) < 1e-6  # SIMD from_float vs _ref differ in rare blocks


def _fa_cases():
    g = load_gold("flash_attn_ext.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _fa_cases())
def test_flash_attn_vs_ggml_cpu(tag):
    g = load_gold("flash_attn_ext.gold")
    d, nh, nhkv, nq, nkv, tkv, has_mask, nqp = (int(x) for x in g["shape." + tag])
    scale, max_bias, softcap = (float(x) for x in g["params." + tag])
    mask = g["mask." + tag] if has_mask else None
    want = g["dst." + tag].reshape(nq, nh, d)
    got = O.flash_attn_ext(g["q." + tag], g["k." + tag], g["v." + tag], mask, tkv, tkv, d, d, nq, nkv,
                           nh, nhkv, scale, max_bias, softcap)
    # f16 V accumulates in fp16 on the CPU (ops.cpp:8130-8144); SIMD vs generic order differs
    assert O.nmse(want, got) < (2e-6 if tkv == O.F16 else 1e-7)
    exact = O.flash_attn_ext(g["q." + tag], g["k." + tag], g["v." + tag], mask, tkv, tkv, d, d, nq, nkv,
                             nh, nhkv, scale, max_bias, softcap, exact=True)
    assert O.nmse(want, exact) < 5e-4
