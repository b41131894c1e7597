"""GPU parity for the third batch of types (the remainder): q2_K / q3_K / iq4_nl / iq4_xs weights through the C ABI,
against the oracle (pinned by tests/golden/*3.gold, which the reference itself produced) and against the ggml-cpu outputs stored
in those fixtures.  These types take the mat-vec kernel up to 8 columns and the dequantize-in-LDS bf16 GEMM beyond."""
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


@pytest.mark.parametrize("src", ["rand", "quant"])
@pytest.mark.parametrize("t", B.WEIGHT_TYPES3)
def test_dequantize3_golden_bit_exact(ops, src, t):
    g = load_gold("dequantize3.gold")
    tn = B.TYPE_NAMES[t]
    want = g[f"{src}.deq.{tn}"]
    got = ops.dequantize(t, dev(g[f"{src}.blocks.{tn}"]), want.size).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("t", B.WEIGHT_TYPES3)
def test_dequantize3_large_vs_oracle(ops, t):
    w = B.random_blocks(t, 64, 14336, seed=t)
    got = ops.dequantize(t, dev(w), 64 * 14336).cpu().numpy()
    want = O.dequantize_row(t, w, 64 * 14336)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _mm3_cases():
    g = load_gold("mul_mat3.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mm3_cases())
def test_mul_mat3_golden(ops, tag):
    """against the ggml-cpu backend's own outputs (test-backend-ops' bar, NMSE <= 5e-4)"""
    g = load_gold("mul_mat3.gold")
    t, m, k, n = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n, m)
    x = g["x." + tag].reshape(n, k)
    got = ops.mul_mat(t, dev(g["w." + tag]), dev(x), m).cpu().numpy()
    assert O.nmse(O.mul_mat(t, g["w." + tag], x, m, k, n, "q8_1"), got) <= 1e-10
    assert O.nmse(want, got) <= 5e-4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES3)
@pytest.mark.parametrize("m,k,n", [(1, 256, 1), (5, 512, 1), (130, 4096, 1), (4096, 4096, 1), (67, 2048, 2), (33, 768, 4), (16, 32, 7),
                                   (40, 1536, 8), (24, 256, 17), (4096, 14336, 1)])
def test_mmvq3_vs_oracle_contract(ops, t, m, k, n):
    if k % B.BLCK_SIZE[t]:
        pytest.skip("K is not a multiple of the block size")
    rng = np.random.default_rng(m * 131 + k + n)
    w = B.random_blocks(t, m, k, seed=m + k + n + t)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat(t, dev(w), dev(x), m).cpu().numpy()
    rows = slice(0, m) if m * k <= 1 << 22 else slice(0, 96)          # the scalar oracle on a row subset of the big shapes
    want = O.mul_mat(t, w[rows], x, min(m, rows.stop), k, n, "q8_1")
    assert O.nmse(want, got[:, rows]) <= 1e-10
    if m * n >= 64:        # NMSE over a handful of outputs is one draw of the activation-rounding noise, not a bar
        assert O.nmse(O.mul_mat(t, w[rows], x, min(m, rows.stop), k, n, "cpu"), got[:, rows]) <= 5e-4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES3)
@pytest.mark.parametrize("m,k,n", [(128, 256, 128), (130, 1024, 33), (64, 512, 9), (300, 2048, 200), (1024, 4096, 512)])
def test_mmdq3_bf16_mfma_vs_exact(ops, t, m, k, n):
    """prefill GEMM (dequantize in LDS, bf16 MFMA) vs exact W @ x in f64: two bf16 roundings per product"""
    rng = np.random.default_rng(m + k + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    w = B.random_blocks(t, m, k, seed=m * 7 + n)
    wf = O.dequantize_row(t, w, m * k).reshape(m, k).astype(np.float64)
    got = ops.mul_mat_dq(t, dev(w), dev(x), m).cpu().numpy()
    ref = x.astype(np.float64) @ wf.T
    err = float(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert np.isfinite(got).all() and err < 2e-5, err


@pytest.mark.parametrize("t", B.WEIGHT_TYPES3)
def test_get_rows3(ops, t):
    """GGML_OP_GET_ROWS on the embedding table in these types: rows bit-exact with dequantize_row_*"""
    import torch
    k, nrows = 1024, 40
    w = B.random_blocks(t, nrows, k, seed=3 + t)
    ids = np.array([[3, 39, 0, 17, 17, 5]], dtype=np.int32)
    got = ops.get_rows(t, dev(w), torch.from_numpy(ids).cuda(), k).cpu().numpy().reshape(-1, k)
    want = O.dequantize_row(t, w, nrows * k).reshape(nrows, k)[ids[0]]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
