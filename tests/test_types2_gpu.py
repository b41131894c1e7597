"""GPU parity for the second batch of types: q4_1 / q5_0 / q5_1 block weights and f16 / bf16 / f32 weights
through the C ABI, against the oracle (pinned by tests/golden/*2.gold, which the reference itself produced) and against the
ggml-cpu outputs stored in those fixtures; plus GGML_OP_SET_ROWS."""
import numpy as np
import pytest

import llm_amd.blocks as B
import oracle_lib as O
from golden_io import load_gold

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    assert torch.cuda.is_available()
    import llm_amd.ops as ops
    return ops


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _float_weights(t, wf):
    """f32 array -> (device tensor holding the weights in type t, the values they represent as f64)"""
    import torch
    w = torch.from_numpy(np.ascontiguousarray(wf)).cuda()
    if t == B.F16:
        w = w.half()
    elif t == B.BF16:
        w = w.bfloat16()
    return w, w.double().cpu().numpy()


@pytest.mark.parametrize("src", ["rand", "quant"])
@pytest.mark.parametrize("t", B.WEIGHT_TYPES2)
def test_dequantize2_golden_bit_exact(ops, src, t):
    g = load_gold("dequantize2.gold")
    tn = B.TYPE_NAMES[t]
    want = g[f"{src}.deq.{tn}"]
    got = ops.dequantize(t, dev(g[f"{src}.blocks.{tn}"]), want.size).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("t", B.WEIGHT_TYPES2)
def test_dequantize2_large_vs_oracle(ops, t):
    w = B.random_blocks(t, 64, 14336, seed=t)
    got = ops.dequantize(t, dev(w), 64 * 14336).cpu().numpy()
    want = O.dequantize_row(t, w, 64 * 14336)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _mm2_cases():
    g = load_gold("mul_mat2.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mm2_cases())
def test_mul_mat2_golden(ops, tag):
    """against the ggml-cpu backend's own outputs (test-backend-ops' bar, NMSE <= 5e-4)"""
    import torch
    g = load_gold("mul_mat2.gold")
    t, m, k, n = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n, m)
    x = g["x." + tag].reshape(n, k)
    if t in B.WEIGHT_TYPES2:
        got = ops.mul_mat(t, dev(g["w." + tag]), dev(x), m).cpu().numpy()
        assert O.nmse(O.mul_mat(t, g["w." + tag], x, m, k, n, "q8_1"), got) <= 1e-10
    else:
        dt = {B.F32: torch.float32, B.F16: torch.float16, B.BF16: torch.bfloat16}[t]
        w = dev(g["w." + tag]).view(dt).reshape(m, k)
        got = ops.mul_mat_f(t, w, dev(x)).cpu().numpy()
        ref = x.astype(np.float64) @ w.double().cpu().numpy().T
        assert float(((got - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-12
    assert O.nmse(want, got) <= 5e-4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES2)
@pytest.mark.parametrize("m,k,n", [(1, 256, 1), (5, 512, 1), (130, 4096, 1), (4096, 4096, 1), (67, 2048, 2), (33, 768, 4), (16, 32, 7),
                                   (40, 1536, 8), (24, 256, 17)])
def test_mmvq2_vs_oracle_contract(ops, t, m, k, n):
    rng = np.random.default_rng(m * 131 + k + n)
    w = B.random_blocks(t, m, k, seed=m + k + n + t)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat(t, dev(w), dev(x), m).cpu().numpy()
    want = O.mul_mat(t, w, x, m, k, n, "q8_1")
    assert O.nmse(want, got) <= 1e-10
    assert O.nmse(O.mul_mat(t, w, x, m, k, n, "cpu"), got) <= 5e-4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES2 + B.FLOAT_WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n", [(128, 256, 128), (130, 1024, 33), (64, 512, 9), (300, 2048, 200), (1024, 4096, 512)])
def test_mmdq2_bf16_mfma_vs_exact(ops, t, m, k, n):
    """prefill GEMM (dequantize / convert in LDS, bf16 MFMA) vs exact W @ x in f64: two bf16 roundings per product"""
    rng = np.random.default_rng(m + k + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    if t in B.WEIGHT_TYPES2:
        w = B.random_blocks(t, m, k, seed=m * 7 + n)
        wd = dev(w)
        wf = O.dequantize_row(t, w, m * k).reshape(m, k).astype(np.float64)
    else:
        wd, wf = _float_weights(t, rng.standard_normal((m, k)).astype(np.float32) * 0.05)
        wd = wd.view(-1).view(__import__("torch").uint8)
    got = ops.mul_mat_dq(t, wd, dev(x), m).cpu().numpy()
    ref = x.astype(np.float64) @ wf.T
    err = float(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert np.isfinite(got).all() and err < 2e-5, err


@pytest.mark.parametrize("t", B.FLOAT_WEIGHT_TYPES)
@pytest.mark.parametrize("tb", [B.F32, B.F16])
@pytest.mark.parametrize("m,k,n", [(16, 256, 1), (33, 100, 3), (130, 4096, 8), (7, 1, 2), (64, 520, 9), (4096, 4096, 1)])
def test_mul_mat_f_vs_f64(ops, t, tb, m, k, n):
    import torch
    rng = np.random.default_rng(m + 3 * k + n)
    wd, wf = _float_weights(t, rng.standard_normal((m, k)).astype(np.float32))
    x = rng.standard_normal((n, k)).astype(np.float32)
    xd = dev(x)
    if tb == B.F16:
        xd = xd.half()
    got = ops.mul_mat_f(t, wd, xd, type_b=tb).cpu().numpy()
    ref = xd.double().cpu().numpy() @ wf.T
    assert float(((got - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-11


def test_mul_mat_f_broadcast_and_strides(ops):
    """the K.Q product of a graph without flash attention: f16 K cache view [n_head_kv][n_kv][d] with padded rows, Q
    [n_head][n_q][d] f32, 4 Q heads per KV head; 2-byte-aligned (not 16-byte) rows take the element path"""
    import torch
    rng = np.random.default_rng(5)
    hk, g, nkv, nq, d = 2, 4, 70, 3, 128
    cache = torch.from_numpy(rng.standard_normal((hk, nkv, d + 8)).astype(np.float32)).cuda().half()
    kview = cache[:, :, 4:4 + d]          # row stride (d + 8) * 2 bytes, start 8 bytes in
    q = dev(rng.standard_normal((hk * g, nq, d)).astype(np.float32))
    got = ops.mul_mat_f(B.F16, kview, q).cpu().numpy()
    ref = np.einsum("hqd,hkd->hqk", q.double().cpu().numpy(), np.repeat(kview.double().cpu().numpy(), g, axis=0))
    assert got.shape == (hk * g, nq, nkv)
    assert float(((got - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-11


@pytest.mark.parametrize("tdst", [B.F32, B.F16, B.Q8_0, B.Q4_0, B.BF16])
def test_set_rows(ops, tdst):
    """GGML_OP_SET_ROWS (ggml-cpu/ops.cpp:5179-5227): rows scattered by an i64 index that is broadcast over dims 2, 3;
    q8_0 / q4_0 rows are bit-exact with quantize_row_q8_0_ref / quantize_row_q4_0_ref, bf16 rows with ggml_fp32_to_bf16_row_ref"""
    import torch
    rng = np.random.default_rng(11)
    ne3, ne2, n, k, nrows = 3, 4, 5, 64, 11
    src = rng.standard_normal((ne3, ne2, n, k)).astype(np.float32)
    ids = np.stack([rng.permutation(nrows)[:n] for _ in range(2 * 1)]).reshape(1, 2, n).astype(np.int64)   # ids [1][2][n]
    if tdst in (B.Q8_0, B.Q4_0):
        dst = torch.zeros((ne3, ne2, nrows, k // 32 * B.TYPE_SIZE[tdst]), dtype=torch.uint8, device="cuda")
    elif tdst == B.BF16:
        dst = torch.full((ne3, ne2, nrows, k), 7, dtype=torch.int16, device="cuda")
    else:
        dst = torch.full((ne3, ne2, nrows, k), -7.0, dtype=torch.float32 if tdst == B.F32 else torch.float16, device="cuda")
    before = dst.clone()
    ops.set_rows(dev(src), dev(ids), dst, tdst)
    got, was = dst.cpu().numpy(), before.cpu().numpy()
    touched = np.zeros((ne3, ne2, nrows), dtype=bool)
    for i3 in range(ne3):
        for i2 in range(ne2):
            for i in range(n):
                r = ids[0, i2 % 2, i]
                touched[i3, i2, r] = True
                if tdst in (B.Q8_0, B.Q4_0):
                    assert np.array_equal(got[i3, i2, r], O.quantize_row(tdst, src[i3, i2, i]))
                elif tdst == B.BF16:
                    assert np.array_equal(got[i3, i2, r].view(np.uint16), O.fp32_to_bf16_row(src[i3, i2, i]))
                else:
                    assert np.array_equal(got[i3, i2, r], src[i3, i2, i].astype(got.dtype))
    assert np.array_equal(got[~touched], was[~touched])


@pytest.mark.parametrize("name", ["cos0", "u1234", "wide", "edge"])
def test_cpy_into_q4_0_and_bf16_golden(ops, name):
    """CPY f32 -> q4_0 / bf16 (the KV-cache stores of -ctk q4_0 / bf16; ggml-cuda/cpy.cu) against the reference's own blocks / rows
    (tests/golden/quantize2.gold, written by quantize_row_q4_0_ref and ggml_fp32_to_bf16_row_ref): bit-exact, edge blocks included."""
    import torch
    import llm_amd.kernels as K
    g = load_gold("quantize2.gold")
    x = dev(g["x." + name].reshape(4, -1))          # four rows
    s = torch.cuda.current_stream().cuda_stream
    k = x.shape[1]
    def desc(t, type_id, k_elems, unit):
        d = K.Tensor()
        d.data, d.type = t.data_ptr(), type_id
        d.ne[0], d.ne[1], d.ne[2], d.ne[3] = k_elems, t.shape[0], 1, 1
        d.nb[0] = unit; d.nb[1] = t.stride(0) * t.element_size(); d.nb[2] = d.nb[3] = d.nb[1] * t.shape[0]
        return d
    q = torch.zeros((4, k // 32 * 18), dtype=torch.uint8, device="cuda")
    K.check(K.lib.mi355_cpy(desc(x, B.F32, k, 4), desc(q, B.Q4_0, k, 18), s), "cpy q4_0")
    got, want = q.cpu().numpy().reshape(-1, 18), g["q4_0." + name].reshape(-1, 18)
    bad = [i for i in range(want.shape[0]) if not np.array_equal(got[i], want[i])]
    assert not bad, (bad[:8], [(got[i].tolist(), want[i].tolist(), g["x." + name][32 * i:32 * i + 4].tolist()) for i in bad[:2]])
    b = torch.zeros((4, k), dtype=torch.int16, device="cuda")
    K.check(K.lib.mi355_cpy(desc(x, B.F32, k, 4), desc(b, B.BF16, k, 2), s), "cpy bf16")
    assert np.array_equal(b.cpu().numpy().view(np.uint16).ravel(), g["bf16." + name])

