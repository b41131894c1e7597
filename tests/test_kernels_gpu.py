"""GPU parity tests: HIP kernels (through the C ABI) vs the oracle and the reference's golden vectors.

bit-exact : q8_1 activation quantization, block dequantization (BASELINE.json north_star)
tolerance : MUL_MAT / MUL_MAT_ID -- NMSE <= 5e-4 vs ggml-cpu output (tests/test-backend-ops.cpp:2254-2256,
            2357-2359) and <= 1e-10 vs the oracle's q8_1-contract evaluation (same integers, f32 sum order only)
"""
import numpy as np
import pytest
import torch

import oracle_lib as O
from golden_io import load_gold

pytestmark = pytest.mark.gpu

import llm_amd.blocks as B  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import llm_amd.ops as ops_
    return ops_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---------------------------------------------------------------- q8_1 quantization: bit-exact
@pytest.mark.parametrize("name", ["cos0", "cos1", "u1234", "wide", "edge"])
def test_quantize_q8_1_golden_bit_exact(ops, name):
    g = load_gold("quantize.gold")
    x = g["x." + name]
    got = ops.quantize_q8_1(dev(x).reshape(1, -1)).cpu().numpy().ravel()
    want = g["q8_1." + name]
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {want.size} bytes differ"


@pytest.mark.parametrize("rows,k,seed", [(1, 32, 0), (7, 4096, 1), (64, 14336, 2), (3, 28672, 3), (512, 4096, 4)])
def test_quantize_q8_1_oracle_bit_exact(ops, rows, k, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, k)) * rng.choice([1e-3, 1.0, 30.0], size=(rows, 1))).astype(np.float32)
    got = ops.quantize_q8_1(dev(x)).cpu().numpy()
    for r in range(0, rows, max(1, rows // 8)):
        assert np.array_equal(got[r], O.quantize_row(O.Q8_1, x[r])), r


def test_quantize_q8_1_strided_4d(ops):
    # permuted / padded views as test-backend-ops builds them (tests/test-backend-ops.cpp:4560-4606)
    rng = np.random.default_rng(5)
    base = rng.standard_normal((2, 3, 5, 96)).astype(np.float32)
    xt = dev(base)
    view = xt.permute(1, 0, 2, 3)[:, :, :, :64]          # ne = [64, 5, 2, 3], non-contiguous
    got = ops.quantize_q8_1(view).cpu().numpy()
    ref = base.transpose(1, 0, 2, 3)[:, :, :, :64].reshape(-1, 64)
    for r in range(ref.shape[0]):
        assert np.array_equal(got[r], O.quantize_row(O.Q8_1, np.ascontiguousarray(ref[r])))


def test_quantize_act_layout_matches_q8_1(ops):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((5, 1024)).astype(np.float32)
    aq, ads = ops.quantize_act(dev(x))
    aq, ads = aq.cpu().numpy(), ads.cpu().numpy()
    for r in range(5):
        blk = O.quantize_row(O.Q8_1, x[r]).reshape(-1, 36)
        assert np.array_equal(aq[r].view(np.uint8).reshape(-1, 32), blk[:, 4:])
        d = blk[:, 0:2].copy().view(np.float16).astype(np.float32).ravel()
        assert np.array_equal(ads[r, :, 0], d)
        assert np.array_equal(ads[r, :, 1], blk[:, 4:].view(np.int8).astype(np.int32).sum(1).astype(np.float32))


# ---------------------------------------------------------------- block dequantization: bit-exact
@pytest.mark.parametrize("src", ["rand", "quant"])
@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
def test_dequantize_golden_bit_exact(ops, src, t):
    g = load_gold("dequantize.gold")
    tn = B.TYPE_NAMES[t]
    want = g[f"{src}.deq.{tn}"]
    got = ops.dequantize(t, dev(g[f"{src}.blocks.{tn}"]), want.size).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
def test_dequantize_large_vs_oracle(ops, t):
    k = 14336 if B.BLCK_SIZE[t] == 256 else 14336
    w = B.random_blocks(t, 64, k, seed=t)
    got = ops.dequantize(t, dev(w), 64 * k).cpu().numpy()
    want = O.dequantize_row(t, w, 64 * k)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------- MUL_MAT
def _mm_cases():
    g = load_gold("mul_mat.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mm_cases())
def test_mul_mat_golden(ops, tag):
    g = load_gold("mul_mat.gold")
    t, m, k, n = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n, m)                      # ggml-cpu backend output
    x = g["x." + tag].reshape(n, k)
    got = ops.mul_mat(t, dev(g["w." + tag]), dev(x), m).cpu().numpy()
    assert O.nmse(want, got) <= 5e-4
    contract = O.mul_mat(t, g["w." + tag], x, m, k, n, "q8_1")
    assert O.nmse(contract, got) <= 1e-10


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n", [(1, 256, 1), (5, 512, 1), (130, 4096, 1), (67, 2048, 2), (64, 1024, 3), (33, 768, 4),
                                   (16, 256, 5), (16, 256, 6), (16, 256, 7), (40, 1536, 8)])
def test_mmvq_vs_oracle_contract(ops, t, m, k, n):
    if k % B.BLCK_SIZE[t]:
        pytest.skip("k not a multiple of the block size")
    rng = np.random.default_rng(m * 131 + k + n)
    w = B.random_blocks(t, m, k, seed=m + k + n + t)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat(t, dev(w), dev(x), m).cpu().numpy()
    want = O.mul_mat(t, w, x, m, k, n, "q8_1")
    assert O.nmse(want, got) <= 1e-10
    assert np.allclose(got, want, rtol=2e-4, atol=2e-4 * np.abs(want).max())


@pytest.mark.parametrize("t,m,k", [(B.Q4_K, 4096, 4096), (B.Q4_K, 1024, 14336), (B.Q6_K, 2048, 4096), (B.Q4_0, 4096, 4096),
                                   (B.Q8_0, 1024, 4096), (B.Q5_K, 1024, 8192),
                                   # Llama-3-70B (SURVEY config 4): wq / wo, gate / up, down, wv in its three types
                                   (B.Q4_K, 8192, 8192), (B.Q4_K, 28672, 8192), (B.Q4_K, 8192, 28672), (B.Q6_K, 8192, 28672),
                                   (B.Q6_K, 1024, 8192), (B.Q6_K, 128256, 4096),
                                   # two rows per wave with a row image that is not a whole number of 512-byte steps (q6_K at K = 14336)
                                   (B.Q6_K, 8192, 14336), (B.Q6_K, 4099, 14336), (B.Q4_K, 8192, 14336), (B.Q5_K, 5000, 14336)])
def test_mmvq_model_shapes_linearity(ops, t, m, k):
    # full-size property check: mat-vec is linear in x up to the activation quantization, so
    # compare against the oracle on a row subset and check W.(2x) == 2 W.x exactly (q8_1 scales by 2)
    rng = np.random.default_rng(k)
    w = B.random_blocks(t, m, k, seed=k + t)
    x = rng.standard_normal((1, k)).astype(np.float32)
    wd = dev(w)
    y1 = ops.mul_mat(t, wd, dev(x), m).cpu().numpy()
    y2 = ops.mul_mat(t, wd, dev(2 * x), m).cpu().numpy()
    assert np.array_equal(2 * y1, y2)
    rows = np.arange(0, m, max(1, m // 48) | 1)           # odd stride: rows of both parities (a wave may own 2+ consecutive rows)
    want = O.mul_mat(t, np.ascontiguousarray(w[rows]), x, rows.size, k, 1, "q8_1")
    assert O.nmse(want, y1[:, rows]) <= 1e-10


@pytest.mark.parametrize("t", [B.Q4_K, B.Q5_K, B.Q6_K, B.Q8_0, B.Q4_0, B.Q5_1])
@pytest.mark.parametrize("m,k,n", [(4096, 14336, 8), (4099, 14336, 3), (4096, 4096, 5), (3072, 8192, 7), (14336, 4096, 8)])
def test_mmvq_multi_column_model_shapes(ops, t, m, k, n):
    # 3..8 columns at model shapes take the 16-rows-per-workgroup form (all columns staged in LDS, up to 143 KB); a column of
    # the result is the same arithmetic as the one-column launch on that column, and matches the oracle on a row subset
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m + n + t)
    x = rng.standard_normal((n, k)).astype(np.float32)
    wd, xd = dev(w), dev(x)
    aq, ads = ops.quantize_act(xd)
    got = ops.mul_mat_vec_q(t, wd, m, k, aq, ads, n)
    for c in (0, n - 1):
        aq1, ads1 = ops.quantize_act(xd[c:c + 1].contiguous())
        one = ops.mul_mat_vec_q(t, wd, m, k, aq1, ads1, 1)
        assert torch.equal(got[c], one[0]), f"column {c}"
    rows = np.arange(0, m, max(1, m // 32) | 1)
    want = O.mul_mat(t, np.ascontiguousarray(w[rows]), x, rows.size, k, n, "q8_1")
    assert O.nmse(want, got.cpu().numpy()[:, rows]) <= 1e-10


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n", [(128, 256, 128), (130, 1024, 33), (64, 512, 9), (300, 2048, 200), (257, 256, 129)])
def test_mmq_int8_mfma_vs_oracle_contract(ops, t, m, k, n):
    # n > 8 and k % 128 == 0 -> the int8-MFMA tile kernel (mi355_mul_mat_q); same q8_1 contract as the mat-vec
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m * 7 + k + n + t)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat(t, dev(w), dev(x), m).cpu().numpy()
    want = O.mul_mat(t, w, x, m, k, n, "q8_1")
    assert O.nmse(want, got) <= 1e-10
    assert O.nmse(O.mul_mat(t, w, x, m, k, n, "cpu"), got) <= 5e-4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n", [(128, 256, 128), (130, 1024, 33), (64, 512, 9), (300, 2048, 200), (257, 256, 129), (1024, 4096, 512),
                                   (12810, 512, 300),         # the 128 x 256 tile (>= 200 workgroups of it), ragged in both directions
                                   # too few 128 x 256 tiles for the chip: K in 4 / 2 slices, partial tiles summed in slice order
                                   (4096, 4096, 512), (4100, 14336, 300), (8192, 2048, 512)])
def test_mmdq_bf16_mfma_vs_dequantized_reference(ops, t, m, k, n):
    """dequantize-in-LDS bf16 MFMA GEMM vs exact dequant(W) @ x in f64 (the ggml contract, ggml.c:2971-2986).
    Error model: two bf16 roundings per product -> NMSE ~ 3e-6; the op's budget is 5e-4 (test-backend-ops)."""
    if k % B.BLCK_SIZE[t]:
        pytest.skip("k not a multiple of the block size")
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m * 7 + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat_dq(t, dev(w), dev(x), m).cpu().numpy()
    wf = O.dequantize_row(t, w, m * k).reshape(m, k).astype(np.float64)
    ref = x.astype(np.float64) @ wf.T
    err = float(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert np.isfinite(got).all() and err < 2e-5, err
    if n <= 64 and m <= 300:   # and against the reference's CPU arithmetic (q8 activations) within the op's budget
        assert O.nmse(O.mul_mat(t, w, x, m, k, n, "cpu"), got) <= 5e-4


def test_mmdq_split_k_workspace_from_the_caller(ops):
    """mi355_mul_mat_dq_slice_ws: the split-K partial tiles in a workspace the caller owns (what the plugin does, so the kernel library
    allocates nothing) -- same bits as with the library's own workspace; the size query is 0 for shapes that need none; a workspace that
    is too small is refused before anything is launched."""
    import llm_amd.kernels as K
    t, m, k, n = B.Q4_K, 4096, 4096, 512
    rng = np.random.default_rng(5)
    w = dev(B.random_blocks(t, m, k, seed=9))
    x = ops.act_bf16(dev(rng.standard_normal((n, k)).astype(np.float32)))
    s = torch.cuda.current_stream().cuda_stream
    need = K.lib.mi355_mul_mat_dq_workspace_bytes(m, m, k, n)
    assert need == 4 * n * m * 4                                    # 64 tiles of 128 x 256: K in four slices
    assert K.lib.mi355_mul_mat_dq_workspace_bytes(14336, 14336, k, n) == 0
    assert K.lib.mi355_mul_mat_dq_workspace_bytes(1024, 4096, k, n) == 4 * n * 1024 * 4     # a row slice: the cut of the whole matrix
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    a = torch.empty((n, m), device="cuda")
    b = torch.full((n, m), 7.0, device="cuda")
    # without a workspace the library refuses a shape that needs one: it allocates nothing itself
    assert K.lib.mi355_mul_mat_dq_slice(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, a.data_ptr(), m, m, m, k, n, s) == -5
    assert K.lib.mi355_mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, a.data_ptr(), m, m, k, n, s) == -5
    K.check(K.mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, a.data_ptr(), m, m, k, n, s), "dq (python-side workspace)")
    K.check(K.lib.mi355_mul_mat_dq_slice_ws(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, b.data_ptr(), m, m, m, k, n, ws.data_ptr(), need, s), "dq_slice_ws")
    assert torch.equal(a, b)
    rc = K.lib.mi355_mul_mat_dq_slice_ws(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, b.data_ptr(), m, m, m, k, n, ws.data_ptr(), need - 16, s)
    assert rc == K.E_WORKSPACE if hasattr(K, "E_WORKSPACE") else rc == -5


@pytest.mark.parametrize("t", [B.Q4_K, B.Q6_K, B.Q8_0, B.Q4_0])
@pytest.mark.parametrize("op", [0, 1, 2])
def test_glu_mul_mat_dq_has_the_bits_of_the_three_launches(ops, t, op):
    """mi355_glu_mul_mat_dq (ffn_gate, ffn_up and the GLU in one launch, each tile holding 64 rows of both matrices) against
    mi355_mul_mat_dq x 2 + mi355_glu_bf16_f32: f32 result and bf16 copy bit for bit; ragged m (not a multiple of 64) and n (not
    a multiple of 256); shapes outside the contract are refused, not approximated."""
    import ctypes as C
    import llm_amd.kernels as K
    if op != 2 and t != B.Q4_K:
        pytest.skip("the activation is independent of the weight type")
    m, k, n = 14336 + 36, 1024, 512 - 3                     # m % 4 == 0: the vector path of mi355_glu_bf16_f32
    rng = np.random.default_rng(17)
    wa = dev(B.random_blocks(t, m, k, seed=3))
    wb = dev(B.random_blocks(t, m, k, seed=4))
    x = ops.act_bf16(dev(rng.standard_normal((n, k)).astype(np.float32)))
    s = torch.cuda.current_stream().cuda_stream
    rs = B.row_size(t, k)
    g = torch.empty((n, m), device="cuda"); u = torch.empty((n, m), device="cuda")
    K.check(K.mul_mat_dq(t, wa.data_ptr(), rs, x.data_ptr(), k, g.data_ptr(), m, m, k, n, s), "dq gate")
    K.check(K.mul_mat_dq(t, wb.data_ptr(), rs, x.data_ptr(), k, u.data_ptr(), m, m, k, n, s), "dq up")
    h = torch.empty((n, m), device="cuda"); hb = torch.empty((n, m), dtype=torch.int16, device="cuda")
    K.check(K.lib.mi355_glu_bf16_f32(op, C.byref(K.Tensor.of(g)), C.byref(K.Tensor.of(u)), C.byref(K.Tensor.of(h)), 0, hb.data_ptr(), s), "glu")
    h2 = torch.full((n, m), 7.0, device="cuda"); hb2 = torch.full((n, m), 7, dtype=torch.int16, device="cuda")
    K.check(K.lib.mi355_glu_mul_mat_dq(t, op, wa.data_ptr(), rs, wb.data_ptr(), rs, x.data_ptr(), k, h2.data_ptr(), m, hb2.data_ptr(), m, k, n, s), "glu_dq")
    torch.cuda.synchronize()
    assert torch.isfinite(h).all()
    assert torch.equal(h.view(torch.int32), h2.view(torch.int32))
    assert torch.equal(hb, hb2)
    # few tiles (the separate GEMMs would cut K or take small tiles) and odd m: refused
    for mm, nn in ((4096, 512), (14336, 128), (14335, 512)):
        rc = K.lib.mi355_glu_mul_mat_dq(t, op, wa.data_ptr(), rs, wb.data_ptr(), rs, x.data_ptr(), k, h2.data_ptr(), mm, hb2.data_ptr(), mm, k, nn, s)
        assert rc == -6, (mm, nn, rc)                   # MI355_E_UNSUPPORTED


@pytest.mark.parametrize("mq,mkv,k,n,hd,n_dims,mode,ff,kv16,tq,tv", [
    (4096, 1024, 4096, 512, 128, 128, 2, True, True, B.Q4_K, B.Q6_K),       # Llama-8B layer, 512 tokens: every product in 4 K slices
    (4096, 1024, 4096, 2048, 128, 128, 0, False, True, B.Q4_K, B.Q4_K),     # 2048 tokens: q in one K range, k / v split
    (1024, 256, 1024, 160, 64, 32, 0, False, False, B.Q8_0, B.Q8_0),        # small shapes, partial rotation, f32 cache
    (2048, 512, 2048, 300, 128, 128, 2, False, True, B.Q5_K, B.Q5_K),
    (4096, 1024, 4096, 16, 128, 128, 2, True, True, B.Q4_K, B.Q6_K),        # <= 64 tokens: one workgroup per token, rotations through LDS
    (4096, 1024, 4096, 33, 128, 128, 0, False, True, B.Q4_K, B.Q4_K),
    (1024, 256, 1024, 7, 64, 64, 0, True, False, B.Q8_0, B.Q8_0),
    (2048, 512, 2048, 64, 128, 128, 2, False, False, B.Q5_K, B.Q5_K),
])
def test_qkv_parts_and_finish_have_the_bits_of_the_separate_launches(ops, mq, mkv, k, n, hd, n_dims, mode, ff, kv16, tq, tv):
    """Prefill Q / K / V: mi355_mul_mat_dq_parts x 3 (the GEMMs stopped before their slice sum) + mi355_qkv_finish (slice sum, rope of
    q and k, conversion of k and v to the cache type) against mi355_mul_mat_dq x 3, mi355_rope_f32 x 2 and the f32 -> f16 conversion:
    bit for bit."""
    import ctypes as C
    import llm_amd.kernels as K
    rng = np.random.default_rng(23)
    s = torch.cuda.current_stream().cuda_stream
    x = ops.act_bf16(dev(rng.standard_normal((n, k)).astype(np.float32)))
    pos = dev((np.arange(n) * 3 + 5).astype(np.int32))
    fac = dev((1.0 + rng.random(hd // 2)).astype(np.float32)) if ff else None
    rope_kw = dict(n_ctx_orig=4096, freq_base=500000.0, freq_scale=0.5, ext_factor=0.25 if ff else 0.0, attn_factor=1.1, beta_fast=32.0, beta_slow=1.0)
    ws, ms, ts = [], (mq, mkv, mkv), (tq, tq, tv)
    want, parts, kzs = [], [], []
    for i, (m, t) in enumerate(zip(ms, ts)):
        w = dev(B.random_blocks(t, m, k, seed=40 + i)); ws.append(w)
        d = torch.empty((n, m), device="cuda")
        K.check(K.mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, d.data_ptr(), m, m, k, n, s), "dq")
        want.append(d)
        nb = K.lib.mi355_mul_mat_dq_parts_bytes(m, m, k, n)
        assert nb >= n * m * 4
        p = torch.full((nb // 4,), float("nan"), device="cuda")
        kz = C.c_int32(0)
        K.check(K.lib.mi355_mul_mat_dq_parts(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, p.data_ptr(), nb, m, m, k, n, C.byref(kz), s), "dq_parts")
        assert 1 <= kz.value <= 4 and nb == max(kz.value, 1) * n * m * 4
        parts.append(p); kzs.append(kz.value)
        planes = p.view(kz.value, n, m)
        acc = planes[0].clone()
        for z in range(1, kz.value):
            acc += planes[z]
        assert torch.equal(acc, d)                      # the planes, added in order, are the product
    q_want = ops.rope(want[0].view(n, mq // hd, hd), pos, n_dims, mode, freq_factors=fac, **rope_kw).view(n, mq)
    k_want = ops.rope(want[1].view(n, mkv // hd, hd), pos, n_dims, mode, freq_factors=fac, **rope_kw).view(n, mkv)
    v_want = want[2]
    kvt = torch.float16 if kv16 else torch.float32
    k_want, v_want = k_want.to(kvt), v_want.to(kvt)
    q_out = torch.full((n, mq), 7.0, device="cuda")
    k_out = torch.full((n, mkv), 7.0, dtype=kvt, device="cuda"); v_out = torch.full((n, mkv), 7.0, dtype=kvt, device="cuda")
    f = K.QkvFinishArgs()
    f.q_parts, f.k_parts, f.v_parts = parts[0].data_ptr(), parts[1].data_ptr(), parts[2].data_ptr()
    f.kz_q, f.kz_k, f.kz_v = kzs
    f.m_q, f.m_k, f.m_v, f.n = mq, mkv, mkv, n
    f.q_out, f.k_out, f.v_out = q_out.data_ptr(), k_out.data_ptr(), v_out.data_ptr()
    f.k_out_type = f.v_out_type = B.F16 if kv16 else B.F32
    f.pos = pos.data_ptr(); f.freq_factors = fac.data_ptr() if ff else None
    f.head_dim, f.n_dims, f.mode, f.n_ctx_orig = hd, n_dims, mode, rope_kw["n_ctx_orig"]
    f.freq_base, f.freq_scale, f.ext_factor, f.attn_factor = rope_kw["freq_base"], rope_kw["freq_scale"], rope_kw["ext_factor"], rope_kw["attn_factor"]
    f.beta_fast, f.beta_slow = rope_kw["beta_fast"], rope_kw["beta_slow"]
    K.check(K.lib.mi355_qkv_finish(C.byref(f), s), "qkv_finish")
    torch.cuda.synchronize()
    assert torch.isfinite(q_out).all()
    assert torch.equal(q_out, q_want)
    assert torch.equal(k_out, k_want)
    assert torch.equal(v_out, v_want)


@pytest.mark.parametrize("ta,tb,m,k,n", [(B.Q4_K, B.Q6_K, 1024, 4096, 512), (B.Q4_K, B.Q4_K, 1024, 4096, 200), (B.Q8_0, B.Q8_0, 1000, 2048, 300),
                                          (B.Q5_K, B.Q6_K, 512, 8192, 512)])
def test_mul_mat_dq_parts_pair_has_the_bits_of_two_calls(ops, ta, tb, m, k, n):
    """mi355_mul_mat_dq_parts_pair (wk and wv of a prefill batch in one launch) against two mi355_mul_mat_dq_parts calls: the same number
    of K slices and the same planes, bit for bit; a shape whose single launch would not be the small-tile split form is refused."""
    import ctypes as C
    import llm_amd.kernels as K
    rng = np.random.default_rng(31)
    s = torch.cuda.current_stream().cuda_stream
    x = ops.act_bf16(dev(rng.standard_normal((n, k)).astype(np.float32)))
    wa, wb = dev(B.random_blocks(ta, m, k, seed=60)), dev(B.random_blocks(tb, m, k, seed=61))
    nb = K.lib.mi355_mul_mat_dq_parts_bytes(m, m, k, n)
    single, kzs = [], []
    for t, w in ((ta, wa), (tb, wb)):
        p = torch.full((nb // 4,), float("nan"), device="cuda")
        kz = C.c_int32(0)
        K.check(K.lib.mi355_mul_mat_dq_parts(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, p.data_ptr(), nb, m, m, k, n, C.byref(kz), s), "parts")
        single.append(p); kzs.append(kz.value)
    pa = torch.full((nb // 4,), float("nan"), device="cuda"); pb = torch.full((nb // 4,), float("nan"), device="cuda")
    kz = C.c_int32(0)
    K.check(K.lib.mi355_mul_mat_dq_parts_pair(ta, wa.data_ptr(), B.row_size(ta, k), pa.data_ptr(), tb, wb.data_ptr(), B.row_size(tb, k), pb.data_ptr(), nb,
                                              x.data_ptr(), k, m, k, n, C.byref(kz), s), "pair")
    torch.cuda.synchronize()
    assert kz.value == kzs[0] == kzs[1] and kz.value >= 2
    used = kz.value * n * m
    assert torch.isfinite(pa[:used]).all() and torch.isfinite(pb[:used]).all()
    assert torch.equal(pa[:used], single[0][:used]) and torch.equal(pb[:used], single[1][:used])
    # K = 512: too short to split, the single launch would not be the split form -> refused (MI355_E_UNSUPPORTED), nothing launched
    rc = K.lib.mi355_mul_mat_dq_parts_pair(ta, wa.data_ptr(), B.row_size(ta, 512), pa.data_ptr(), tb, wb.data_ptr(), B.row_size(tb, 512), pb.data_ptr(), nb,
                                           x.data_ptr(), k, m, 512, n, C.byref(kz), s)
    assert rc == -6


@pytest.mark.parametrize("t,m,k,n", [(B.Q4_K, 4096, 4096, 512), (B.Q6_K, 4096, 14336, 512), (B.Q4_K, 4096, 4096, 2048), (B.Q8_0, 1024, 2816, 160), (B.Q4_K, 8192, 2048, 300), (B.Q4_K, 8192, 8192, 511)])
def test_gemm_planes_into_add_rms_norm_have_the_bits_of_the_separate_launches(ops, t, m, k, n):
    """wo / ffn_down of a prefill batch: mi355_mul_mat_dq_parts + mi355_parts_add_rms_norm_mul_bf16_f32 (the norm kernel adds the K-slice
    planes on the way in) against mi355_mul_mat_dq + mi355_rms_norm_mul_bf16_f32 with the residual add: the sum, the normed rows and
    their bf16 copy, bit for bit."""
    import ctypes as C
    import llm_amd.kernels as K
    rng = np.random.default_rng(29)
    s = torch.cuda.current_stream().cuda_stream
    w = dev(B.random_blocks(t, m, k, seed=50))
    x = ops.act_bf16(dev(rng.standard_normal((n, k)).astype(np.float32)))
    res = dev(rng.standard_normal((n, m)).astype(np.float32))
    nw = dev((0.5 + rng.random(m)).astype(np.float32))
    T = lambda v: C.byref(K.Tensor.of(v))
    prod = torch.empty((n, m), device="cuda")
    K.check(K.mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, prod.data_ptr(), m, m, k, n, s), "dq")
    sum_a = torch.empty_like(prod); out_a = torch.empty_like(prod); bf_a = torch.empty((n, m), dtype=torch.int16, device="cuda")
    K.check(K.lib.mi355_rms_norm_mul_bf16_f32(T(prod), T(res), T(sum_a), nw.data_ptr(), T(out_a), 1e-5, bf_a.data_ptr(), s), "norm")
    nb = K.lib.mi355_mul_mat_dq_parts_bytes(m, m, k, n)
    parts = torch.empty(nb // 4, device="cuda")
    kz = C.c_int32(0)
    K.check(K.lib.mi355_mul_mat_dq_parts(t, w.data_ptr(), B.row_size(t, k), x.data_ptr(), k, parts.data_ptr(), nb, m, m, k, n, C.byref(kz), s), "parts")
    sum_b = torch.full_like(prod, 7.0); out_b = torch.full_like(prod, 7.0); bf_b = torch.full((n, m), 7, dtype=torch.int16, device="cuda")
    K.check(K.lib.mi355_parts_add_rms_norm_mul_bf16_f32(parts.data_ptr(), kz.value, T(res), T(sum_b), nw.data_ptr(), T(out_b), 1e-5, bf_b.data_ptr(), s), "parts norm")
    torch.cuda.synchronize()
    assert torch.isfinite(out_a).all()
    assert torch.equal(sum_a, sum_b) and torch.equal(out_a, out_b) and torch.equal(bf_a, bf_b)


def test_kv_store_q8_0_bit_exact(ops):
    """mi355_kv_store_q8_0 (K and V rows of a decode token into a q8_0 cache, one launch) against the oracle's quantize_row_q8_0_ref,
    with the destinations given directly and through the pointer table used under graph replay"""
    import llm_amd.kernels as K
    rng = np.random.default_rng(11)
    nk, nv = 1024, 2048
    k = rng.standard_normal(nk).astype(np.float32) * 3
    v = rng.standard_normal(nv).astype(np.float32)
    v[:32] = 0.0                                           # an all-zero block: d = 0
    kd, vd = dev(k), dev(v)
    s = torch.cuda.current_stream().cuda_stream
    want = {id(k): O.quantize_row(B.Q8_0, k), id(v): O.quantize_row(B.Q8_0, v)}
    for table in (False, True):
        ko = torch.zeros(nk // 32 * 34, dtype=torch.uint8, device="cuda")
        vo = torch.zeros(nv // 32 * 34, dtype=torch.uint8, device="cuda")
        if table:
            tab = torch.tensor([0, 0, ko.data_ptr(), vo.data_ptr()], dtype=torch.int64, device="cuda")
            K.check(K.lib.mi355_kv_store_q8_0(kd.data_ptr(), nk, None, vd.data_ptr(), nv, None, tab.data_ptr(), 1, s), "kv_store")
        else:
            K.check(K.lib.mi355_kv_store_q8_0(kd.data_ptr(), nk, ko.data_ptr(), vd.data_ptr(), nv, vo.data_ptr(), None, 0, s), "kv_store")
        for src, out, n in ((k, ko, nk), (v, vo, nv)):
            blocks = out.cpu().numpy().reshape(n // 32, 34)
            back = O.dequantize_row(B.Q8_0, blocks, n)
            d = blocks[:, :2].copy().view(np.float16).astype(np.float32)[:, 0]
            amax = np.abs(src.reshape(-1, 32)).max(axis=1)
            assert np.array_equal(d, (amax / np.float32(127.0)).astype(np.float16).astype(np.float32))
            assert np.abs(back - src).max() <= 0.5 * d.max() + 1e-6
            assert np.array_equal(blocks, np.asarray(want[id(src)]).view(np.uint8).reshape(-1, 34))
    assert K.lib.mi355_kv_store_q8_0(kd.data_ptr(), 100, ko.data_ptr(), vd.data_ptr(), nv, vo.data_ptr(), None, 0, s) < 0      # n_k % 32


def test_mmdq_strided_activations_and_errors(ops):
    import llm_amd.kernels as K
    rng = np.random.default_rng(5)
    m, k, n = 192, 512, 40
    w = B.random_blocks(B.Q4_K, m, k, seed=3)
    xfull = rng.standard_normal((n, 2 * k)).astype(np.float32)      # row-padded view: stride 2k floats
    xd = dev(xfull)
    got = ops.mul_mat_dq(B.Q4_K, dev(w), xd[:, :k], m).cpu().numpy()
    ref = ops.mul_mat_dq(B.Q4_K, dev(w), dev(xfull[:, :k].copy()), m).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # the bf16 rounding itself is exact round-to-nearest-even
    y = ops.act_bf16(xd[:, :k]).cpu().numpy().view(np.uint16)
    exp = torch.from_numpy(xfull[:, :k].copy()).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(y, exp)
    assert K.lib.mi355_mul_mat_dq(B.Q4_K, 16, 16, 16, 288, 16, 1, 1, 256 + 32, 1, None) == -2   # k % 256
    assert K.lib.mi355_mul_mat_dq(B.Q4_0, 16, 18, 16, 32, 16, 1, 1, 32, 1, None) == -6         # k % 64 -> caller falls back
    assert K.lib.mi355_mul_mat_dq(B.Q4_K, None, 16, 16, 256, 16, 1, 1, 256, 1, None) == -4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n", [(32, 256, 1), (130, 512, 2), (4096, 4096, 8), (257, 2048, 31), (1024, 4096, 32), (96, 1024, 70), (2816, 14336, 5)])
def test_skinny_int8_mfma_vs_oracle_contract(ops, t, m, k, n):
    """small-batch MUL_MAT on the int8 matrix cores: same integers as the q8_1 contract (oracle, double accumulation),
    only the f32 summation order differs -> NMSE <= 1e-10; and within the op's budget of the ggml-cpu arithmetic."""
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat_skinny(t, dev(w), dev(x), m).cpu().numpy()
    ref = O.mul_mat(t, w, x, m, k, n, "q8_1")
    assert np.isfinite(got).all() and O.nmse(ref, got) <= 1e-10, O.nmse(ref, got)
    if m * n <= 40000:
        assert O.nmse(O.mul_mat(t, w, x, m, k, n, "cpu"), got) <= 5e-4


def _q8k_planes(x):
    """oracle q8_K blocks of x [n, k] -> (q int8 [n, k], d f32 [n, k/256], bsums int16 [n, k/16])"""
    n, k = x.shape
    blk = np.stack([O.quantize_row(O.Q8_K, x[i]) for i in range(n)]).reshape(n, k // 256, 292)
    d = blk[:, :, :4].copy().view(np.float32).reshape(n, k // 256)
    q = blk[:, :, 4:260].copy().view(np.int8).reshape(n, k)
    bs = blk[:, :, 260:].copy().view(np.int16).reshape(n, k // 16)
    return q, d, bs


@pytest.mark.parametrize("n,k", [(1, 256), (7, 4096), (33, 1024)])
def test_quantize_q8k_bit_exact(ops, n, k):
    """q8_K activations == quantize_row_q8_K_ref (oracle pinned by the reference's golden vectors), including ties
    between +max and -max (first occurrence wins), all-zero blocks and values that round to +128 (clamped to 127)."""
    rng = np.random.default_rng(n * k)
    x = rng.standard_normal((n, k)).astype(np.float32) * rng.choice([1e-3, 1.0, 300.0], size=(n, 1)).astype(np.float32)
    x[0, :256] = 0.0
    if n > 1:
        x[1, 5] = 9.0; x[1, 77] = -9.0          # tie, positive first
        x[2, 300] = -7.5; x[2, 301] = 7.5       # tie, negative first: +7.5 * (-127 / -7.5) = 127
        x[3, :256] = np.linspace(-1, 1, 256)
    q, d, bs = ops.quantize_act_q8k(dev(x))
    rq, rd, rbs = _q8k_planes(x)
    assert np.array_equal(q.cpu().numpy(), rq)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    assert np.array_equal(bs.cpu().numpy(), rbs)


@pytest.mark.parametrize("t", [B.Q4_K, B.Q5_K, B.Q6_K])      # q6_K: one 32x32x16 MFMA per 16-weight sub-block, scales applied in integers
@pytest.mark.parametrize("m,k,n", [(32, 256, 1), (130, 512, 2), (4096, 4096, 8), (257, 2048, 31), (1024, 4096, 32), (96, 1024, 70), (2816, 14336, 5)])
def test_skinny_k_digit_split_vs_q8k_contract(ops, t, m, k, n):
    """K-quant MUL_MAT with the sub-block scales folded into the MFMA operands: the integers are those of
    ggml_vec_dot_q*_K_q8_K, so against dequantize(W) . dequantize(q8_K(x)) in f64 only f32 rounding remains
    (NMSE <= 1e-10), and the ggml-cpu oracle (f32, its own summation order) agrees to the same level."""
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat_skinny_k(t, dev(w), dev(x), m).cpu().numpy()
    wf = np.stack([O.dequantize_row(t, w[i], k) for i in range(m)]).astype(np.float64)
    q, d, _ = _q8k_planes(x)
    xa = q.astype(np.float64).reshape(n, k // 256, 256) * d.astype(np.float64)[:, :, None]
    ref = (xa.reshape(n, k) @ wf.T).astype(np.float32)
    assert np.isfinite(got).all() and O.nmse(ref, got) <= 1e-10, O.nmse(ref, got)
    if m * n <= 40000:
        assert O.nmse(O.mul_mat(t, w, x, m, k, n, "cpu"), got) <= 1e-10


@pytest.mark.parametrize("t", [B.Q4_K, B.Q5_K, B.Q6_K])
@pytest.mark.parametrize("m,k,n", [(32, 256, 33), (130, 512, 64), (4096, 4096, 40), (257, 2048, 63), (14336, 4096, 48), (2816, 14336, 37), (96, 1024, 150)])
def test_skinny_k_two_column_tiles_per_workgroup(ops, t, m, k, n):
    """33..64 columns in one launch: the two column halves of a row tile are two tiles of one workgroup (the weights cross HBM once).
    The integers are still those of ggml_vec_dot_q*_K_q8_K: NMSE <= 1e-10 against dequantize(W) . dequantize(q8_K(x)) in f64, and
    <= 1e-12 against the same product done in passes of 32 (only the order of the f32 additions over K differs)."""
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat_skinny_k(t, dev(w), dev(x), m, pass_cols=64).cpu().numpy()
    by32 = ops.mul_mat_skinny_k(t, dev(w), dev(x), m).cpu().numpy()
    wf = np.stack([O.dequantize_row(t, w[i], k) for i in range(m)]).astype(np.float64)
    q, d, _ = _q8k_planes(x)
    xa = q.astype(np.float64).reshape(n, k // 256, 256) * d.astype(np.float64)[:, :, None]
    ref = (xa.reshape(n, k) @ wf.T).astype(np.float32)
    assert np.isfinite(got).all() and O.nmse(ref, got) <= 1e-10, O.nmse(ref, got)
    assert O.nmse(by32, got) <= 1e-12


@pytest.mark.parametrize("t", [B.Q4_K, B.Q5_K, B.Q6_K])
@pytest.mark.parametrize("ms,k,n", [((64, 32), 1024, 8), ((4096, 1024, 1024), 4096, 16), ((130, 1, 33, 95), 512, 32), ((257,), 2048, 5), ((96, 32, 31), 1024, 1),
                                    ((14336, 14336), 4096, 8), ((12320, 12288), 1024, 7),      # two tiles per workgroup; an odd tile count
                                    ((4096, 1024, 1024), 4096, 40), ((14336, 14336), 4096, 64), ((130, 1, 33), 512, 33)])     # 33..64 columns
def test_skinny_k_group_of_matrices_in_one_launch(ops, t, ms, k, n):
    """wq / wk / wv (or ffn_up / ffn_gate) of a small batch as one launch over all their row tiles: every matrix's result is the single
    call's BIT FOR BIT (a matrix is cut over the waves as a function of its own (m, k) in both), the q8_K contract (f64) holds for
    each, and rows past a matrix's end never leak into its neighbour."""
    rng = np.random.default_rng(sum(ms) + k + n)
    ws = [B.random_blocks(t, m, k, seed=m + n + i) for i, m in enumerate(ms)]
    x = rng.standard_normal((n, k)).astype(np.float32)
    xd = dev(x)
    outs = ops.mul_mat_skinny_k_group(t, [dev(w) for w in ws], xd, list(ms))
    q, d, _ = _q8k_planes(x)
    xa = (q.astype(np.float64).reshape(n, k // 256, 256) * d.astype(np.float64)[:, :, None]).reshape(n, k)
    for w, m, o in zip(ws, ms, outs):
        got = o.cpu().numpy()
        single = ops.mul_mat_skinny_k(t, dev(w), xd, m, pass_cols=64).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), single.view(np.uint32))
        wf = np.stack([O.dequantize_row(t, w[i], k) for i in range(m)]).astype(np.float64)
        ref = (xa @ wf.T).astype(np.float32)
        assert np.isfinite(got).all() and O.nmse(ref, got) <= 1e-10, O.nmse(ref, got)


def test_skinny_k_group_refuses_what_would_change_a_matrix_s_bits(ops):
    """Matrices whose own launches cut K differently (14336 rows: two waves per tile; 1024 rows: eight), or over two workgroups
    (K = 14336 with fewer than 256 row tiles), are not grouped: MI355_E_UNSUPPORTED, nothing launched."""
    import llm_amd.kernels as K
    for ms, k in (((14336, 1024), 4096), ((4096, 4096), 14336)):
        ws = [dev(B.random_blocks(B.Q4_K, m, k, seed=m)) for m in ms]
        x = dev(np.ones((8, k), dtype=np.float32))
        with pytest.raises(K.Mi355Error) as e:
            ops.mul_mat_skinny_k_group(B.Q4_K, ws, x, list(ms))
        assert e.value.code == -6


@pytest.mark.parametrize("t", [B.Q4_K, B.Q5_K])
@pytest.mark.parametrize("m,k,n", [(64, 256, 128), (1, 256, 1), (130, 512, 200), (4096, 4096, 9), (257, 2048, 131), (96, 1024, 512), (1024, 14336, 65)])
def test_mmqk_digit_split_tile_kernel_vs_q8k_contract(ops, t, m, k, n):
    """prefill form of the digit-split kernel: same integers as ggml_vec_dot_q*_K_q8_K -> against
    dequantize(W) . dequantize(q8_K(x)) in f64 only f32 rounding remains (NMSE <= 1e-10)."""
    rng = np.random.default_rng(m + k + n)
    w = B.random_blocks(t, m, k, seed=m + n)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat_qk(t, dev(w), dev(x), m).cpu().numpy()
    wf = np.stack([O.dequantize_row(t, w[i], k) for i in range(m)]).astype(np.float64)
    q, d, _ = _q8k_planes(x)
    xa = q.astype(np.float64).reshape(n, k // 256, 256) * d.astype(np.float64)[:, :, None]
    ref = (xa.reshape(n, k) @ wf.T).astype(np.float32)
    assert np.isfinite(got).all() and O.nmse(ref, got) <= 1e-10, O.nmse(ref, got)
    if m * n <= 40000:
        assert O.nmse(O.mul_mat(t, w, x, m, k, n, "cpu"), got) <= 1e-10


def test_mmq_matches_matvec_path_at_model_shape(ops):
    # pp512-shaped launch vs the decode kernel on the same data (first 8 columns)
    t, m, k, n = B.Q4_K, 1024, 4096, 512
    rng = np.random.default_rng(3)
    w = B.random_blocks(t, m, k, seed=11)
    x = rng.standard_normal((n, k)).astype(np.float32)
    wd, xd = dev(w), dev(x)
    big = ops.mul_mat(t, wd, xd, m).cpu().numpy()
    small = ops.mul_mat(t, wd, xd[:8].contiguous(), m).cpu().numpy()
    assert O.nmse(small, big[:8]) <= 1e-10


def test_mul_mat_large_batch_falls_back_to_column_groups(ops):
    t, m, k, n = B.Q4_0, 96, 96, 19   # k % 128 != 0: no tile kernel
    rng = np.random.default_rng(9)
    w = B.random_blocks(t, m, k, seed=9)
    x = rng.standard_normal((n, k)).astype(np.float32)
    got = ops.mul_mat(t, dev(w), dev(x), m).cpu().numpy()
    assert O.nmse(O.mul_mat(t, w, x, m, k, n, "q8_1"), got) <= 1e-10


# ---------------------------------------------------------------- MUL_MAT_ID (decode-sized)
@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("ms,k,n", [((4096, 1024, 1024), 4096, 1), ((130, 7, 64, 1), 512, 3), ((300,), 2048, 8), ((2816, 2816), 1024, 2)])
def test_mmvq_group_is_bit_identical_to_separate_launches(ops, t, ms, k, n):
    """grouped QKV / gate+up launch (mi355_mul_mat_vec_q_group) == one launch per matrix, to the bit"""
    rng = np.random.default_rng(k + n)
    x = dev(rng.standard_normal((n, k)).astype(np.float32))
    aq, ads = ops.quantize_act(x)
    ws = [dev(B.random_blocks(t, m, k, seed=100 + i)) for i, m in enumerate(ms)]
    sep = [ops.mul_mat_vec_q(t, w, m, k, aq, ads, n).cpu().numpy() for w, m in zip(ws, ms)]
    grp = [o.cpu().numpy() for o in ops.mul_mat_vec_q_group(t, ws, list(ms), k, aq, ads, n)]
    for a, b in zip(sep, grp):
        assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_mmvq_group_argument_errors(ops):
    import llm_amd.kernels as K
    arr = (K.MmvqArgs * 5)()
    assert K.lib.mi355_mul_mat_vec_q_group(arr, 5, None) == -2          # more than MI355_MMVQ_MAX_GROUP
    assert K.lib.mi355_mul_mat_vec_q_group(arr, 1, None) == -4          # NULL pointers
    x = dev(np.zeros((1, 512), np.float32))
    aq, ads = ops.quantize_act(x)
    w = dev(B.random_blocks(B.Q4_K, 8, 512, seed=1))
    w2 = dev(B.random_blocks(B.Q4_K, 8, 256, seed=2))
    d = dev(np.zeros((1, 8), np.float32))
    for i, (ww, kk) in enumerate(((w, 512), (w2, 256))):
        a = arr[i]
        a.type, a.W, a.w_row_stride = B.Q4_K, ww.data_ptr(), B.row_size(B.Q4_K, kk)
        a.act_q, a.act_ds, a.dst, a.dst_col_stride = aq.data_ptr(), ads.data_ptr(), d.data_ptr(), 8
        a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = 8, kk, 1, 1, 1
    assert K.lib.mi355_mul_mat_vec_q_group(arr, 2, None) == -2          # different k inside one group


@pytest.mark.parametrize("tqk,tv,tk", [(B.Q4_K, B.Q4_K, None), (B.Q4_K, B.Q6_K, None), (B.Q4_K, B.Q5_K, None), (B.Q8_0, B.Q8_0, None),
                                       (B.Q4_0, B.Q4_0, None), (B.Q6_K, B.Q6_K, None), (B.Q4_K, B.Q8_0, B.Q8_0)])
@pytest.mark.parametrize("mode,n_dims,kv", [(0, 128, "f16"), (2, 128, "f16"), (2, 64, "f32"), (0, 96, "f16")])
def test_qkv_rope_fused_is_bit_identical_to_separate_kernels(ops, tqk, tv, tk, mode, n_dims, kv):
    """mi355_qkv_rope_matvec == mul_mat_vec_q x3 + rope x2 + f32->f16 store, to the bit (Llama-3-8B head layout
    scaled down: 8 heads, 2 KV heads, head_dim 128; NORMAL and NEOX rope, partial rotary dims, YaRN off/on)."""
    k, hd, nh, nkv = 1024, 128, 8, 2
    rng = np.random.default_rng(mode * 7 + n_dims)
    x = dev(rng.standard_normal((1, k)).astype(np.float32))
    aq, ads = ops.quantize_act(x)
    tk = tqk if tk is None else tk      # 8-expert Q4_K_M: wq Q4_K, wk / wv Q8_0
    wq, wk, wv = (dev(B.random_blocks(t, m, k, seed=s)) for t, m, s in ((tqk, nh * hd, 1), (tk, nkv * hd, 2), (tv, nkv * hd, 3)))
    pos = torch.tensor([37], dtype=torch.int32, device="cuda")
    kw = dict(freq_base=500000.0, freq_scale=0.5 if n_dims == 96 else 1.0, ext_factor=1.0 if n_dims == 96 else 0.0, n_ctx_orig=4096)
    qf, kf, vf = ops.qkv_rope_matvec(tqk, tv, wq, wk, wv, nh * hd, nkv * hd, nkv * hd, k, aq, ads, pos, hd, n_dims, mode,
                                     kv_dtype=torch.float16 if kv == "f16" else torch.float32, type_k=tk, **kw)
    q0 = ops.mul_mat_vec_q(tqk, wq, nh * hd, k, aq, ads, 1)
    k0 = ops.mul_mat_vec_q(tk, wk, nkv * hd, k, aq, ads, 1)
    v0 = ops.mul_mat_vec_q(tv, wv, nkv * hd, k, aq, ads, 1)
    qr = ops.rope(q0.view(1, nh, hd), pos, n_dims, mode, **kw).view(-1)
    kr = ops.rope(k0.view(1, nkv, hd), pos, n_dims, mode, **kw).view(-1)
    vr = v0.view(-1)
    if kv == "f16":
        kr, vr = kr.to(torch.float16), vr.to(torch.float16)
    assert torch.isfinite(qf).all() and float(qf.abs().max()) > 0
    assert torch.equal(qf.view(torch.int32), qr.view(torch.int32))
    it = torch.int16 if kv == "f16" else torch.int32
    assert torch.equal(kf.view(it), kr.view(it)) and torch.equal(vf.view(it), vr.view(it))


@pytest.mark.parametrize("tqk,tv", [(B.Q4_K, B.Q4_K), (B.Q4_K, B.Q6_K), (B.Q8_0, B.Q8_0)])
@pytest.mark.parametrize("k", [4096, 1024, 96])
def test_qkv_rope_norm_prologue_equals_separate_norm(ops, tqk, tv, k):
    """mi355_qkv_rope_matvec with the RMS_NORM -> MUL(weight) prologue == mi355_rms_norm_mul_quant_f32 followed by the
    plain fused launch, to the bit."""
    import llm_amd.kernels as K
    if k % B.BLCK_SIZE[tqk] != 0 or k % B.BLCK_SIZE[tv] != 0:
        pytest.skip("k not a multiple of the block size")
    hd, nh, nkv = 128, 8, 2
    rng = np.random.default_rng(k)
    x, nw = (dev(rng.standard_normal(sh).astype(np.float32)) for sh in ((1, k), (k,)))
    wq, wk, wv = (dev(B.random_blocks(t, m, k, seed=s_)) for t, m, s_ in ((tqk, nh * hd, 1), (tqk, nkv * hd, 2), (tv, nkv * hd, 3)))
    pos = torch.tensor([11], dtype=torch.int32, device="cuda")
    T, s = K.Tensor.of, torch.cuda.current_stream().cuda_stream
    n0 = torch.empty_like(x)
    aq = torch.empty((1, k), dtype=torch.int8, device="cuda")
    ads = torch.empty((1, k // 32, 2), dtype=torch.float32, device="cuda")
    K.check(K.lib.mi355_rms_norm_mul_quant_f32(T(x), nw.data_ptr(), T(n0), 1e-5, aq.data_ptr(), ads.data_ptr(), s), "rms_norm_mul_quant")
    q0, k0, v0 = ops.qkv_rope_matvec(tqk, tv, wq, wk, wv, nh * hd, nkv * hd, nkv * hd, k, aq, ads, pos, hd, 128, 0, freq_base=500000.0)
    n1 = torch.full_like(x, float("nan"))
    q1, k1, v1 = ops.qkv_rope_matvec(tqk, tv, wq, wk, wv, nh * hd, nkv * hd, nkv * hd, k, None, None, pos, hd, 128, 0, freq_base=500000.0,
                                     norm=(x, nw, 1e-5, n1))
    assert torch.isfinite(q1).all() and float(q1.abs().max()) > 0
    assert torch.equal(n1.view(torch.int32), n0.view(torch.int32)) and torch.equal(q1.view(torch.int32), q0.view(torch.int32))
    assert torch.equal(k1.view(torch.int16), k0.view(torch.int16)) and torch.equal(v1.view(torch.int16), v0.view(torch.int16))


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,op,keep", [(14336, 4096, 2, True), (96, 512, 2, False), (352, 8192, 1, True), (64, 1024, 0, True), (28672, 2048, 2, True)])
def test_glu_matvec_fused_is_bit_identical_to_separate_kernels(ops, t, m, k, op, keep):
    """mi355_glu_matvec == grouped gate/up mat-vec + glu_quant (act(gate) * up and its q8_1 form), to the bit
    (Llama-3-8B ffn shape, a ragged number of 32-blocks, K = 8192 with two units per lane, all three activations)."""
    import llm_amd.kernels as K
    rng = np.random.default_rng(m + k + op)
    x = dev(rng.standard_normal((1, k)).astype(np.float32))
    aq, ads = ops.quantize_act(x)
    wg, wu = dev(B.random_blocks(t, m, k, seed=11)), dev(B.random_blocks(t, m, k, seed=12))
    g0, u0 = ops.mul_mat_vec_q_group(t, [wg, wu], [m, m], k, aq, ads, 1)
    d0 = torch.empty(1, m, device="cuda")
    q0 = torch.empty((1, m), dtype=torch.int8, device="cuda")
    s0 = torch.empty((1, m // 32, 2), dtype=torch.float32, device="cuda")
    T = K.Tensor.of
    K.check(K.lib.mi355_glu_quant_f32(op, T(g0.view(1, m)), T(u0.view(1, m)), T(d0), 0, q0.data_ptr(), s0.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "glu_quant")
    d1, g1, u1, q1, s1 = ops.glu_matvec(t, wg, wu, m, k, aq, ads, glu_op=op, quant=True, keep=keep)
    assert torch.isfinite(d1).all() and float(d1.abs().max()) > 0
    assert torch.equal(d1.view(torch.int32), d0.view(-1).view(torch.int32))
    assert torch.equal(q1, q0.view(-1)) and torch.equal(s1.view(torch.int32), s0.view(-1, 2).view(torch.int32))
    if keep:
        assert torch.equal(g1.view(torch.int32), g0.view(-1).view(torch.int32)) and torch.equal(u1.view(torch.int32), u0.view(-1).view(torch.int32))
    d2 = ops.glu_matvec(t, wg, wu, m, k, aq, ads, glu_op=op, quant=False, keep=False)[0]
    assert torch.equal(d2.view(torch.int32), d1.view(torch.int32))


@pytest.mark.parametrize("t", [B.Q4_K, B.Q6_K, B.Q8_0])
@pytest.mark.parametrize("m,k,n_exp,ids", [(14336, 4096, 8, (5, 2)), (96, 512, 4, (3, 3, 0)), (64, 1024, 8, (7, 9))])
def test_glu_matvec_expert_routing(ops, t, m, k, n_exp, ids):
    """mi355_glu_matvec with ids (MUL_MAT_ID gate / up of one decode token + GLU): every slot equals the plain launch on
    its expert's matrices, to the bit; an out-of-range id leaves its slot untouched."""
    rng = np.random.default_rng(m + k)
    x = dev(rng.standard_normal((1, k)).astype(np.float32))
    aq, ads = ops.quantize_act(x)
    rs = B.row_size(t, k)
    wg, wu = dev(B.random_blocks(t, n_exp * m, k, seed=41)), dev(B.random_blocks(t, n_exp * m, k, seed=42))
    idt = torch.tensor(ids, dtype=torch.int32, device="cuda")
    d1, g1, u1, q1, s1 = ops.glu_matvec(t, wg, wu, m, k, aq, ads, ids=idt, n_expert=n_exp)
    for s_, e in enumerate(ids):
        sl = slice(s_ * m, (s_ + 1) * m)
        if e >= n_exp:
            assert torch.isnan(d1[sl]).all() and int(q1[sl].abs().max()) == 0
            continue
        d0, g0, u0, q0, s0 = ops.glu_matvec(t, wg.view(n_exp, m * rs)[e], wu.view(n_exp, m * rs)[e], m, k, aq, ads)
        assert torch.isfinite(d0).all() and float(d0.abs().max()) > 0
        for a, b in ((d0, d1[sl]), (g0, g1[sl]), (u0, u1[sl]), (s0, s1[s_ * m // 32:(s_ + 1) * m // 32])):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert torch.equal(q0, q1[sl])


@pytest.mark.parametrize("t", [B.Q4_K, B.Q6_K, B.Q8_0, B.Q4_0])
@pytest.mark.parametrize("m,k,n_exp,ids,res", [(4096, 14336, 8, (5, 2), True), (130, 512, 4, (3, 3, 0), False), (64, 1024, 8, (7, 1), True),
                                                (7, 32768, 3, (2, 0), True)])
def test_moe_down_matvec_equals_separate_ops(ops, t, m, k, n_exp, ids, res):
    """mi355_moe_down_matvec == MUL_MAT_ID (per-pair mat-vec) + MUL by the router weights + ADD over the slots + residual
    ADD, every tensor to the bit (Mixtral ffn_down shape, three slots, a K too long for the LDS activation copy)."""
    if k % B.BLCK_SIZE[t]:
        pytest.skip("k not a multiple of the block size")
    rng = np.random.default_rng(m + k)
    ns = len(ids)
    h = dev(rng.standard_normal((1, ns, k)).astype(np.float32))
    w = dev(B.random_blocks(t, n_exp * m, k, seed=51))
    idt = torch.tensor([ids], dtype=torch.int32, device="cuda")
    wts = dev(rng.random(ns).astype(np.float32))
    r = dev(rng.standard_normal(m).astype(np.float32)) if res else None
    y0 = ops.mul_mat_id(t, w, h, idt, m, n_exp)[0]                 # [ns, m]
    yw0 = y0 * wts[:, None]
    sm0 = yw0[0] + yw0[1]
    for s_ in range(2, ns):
        sm0 = sm0 + yw0[s_]
    aq, ads = ops.quantize_act(h.view(ns, k))
    for in_place in (False, True):
        y1, yw1, sm1, r1 = ops.moe_down_matvec(t, w, m, k, n_exp, aq, ads, idt.view(-1), wts, residual=r, in_place=in_place)
        assert torch.isfinite(sm1).all() and float(sm1.abs().max()) > 0
        if not in_place:
            assert torch.equal(y1.view(torch.int32), y0.view(torch.int32))
        assert torch.equal(yw1.view(torch.int32), yw0.view(torch.int32)) and torch.equal(sm1.view(torch.int32), sm0.view(torch.int32))
        if res:
            assert torch.equal(r1.view(torch.int32), (sm0 + r).view(torch.int32))


def test_glu_matvec_argument_errors(ops):
    import llm_amd.kernels as K
    a = K.GluMatvecArgs()
    assert K.lib.mi355_glu_matvec(a, None) == -4                       # NULL pointers
    x = dev(np.zeros((1, 512), np.float32))
    aq, ads = ops.quantize_act(x)
    w = dev(B.random_blocks(B.Q4_K, 64, 512, seed=1))
    d = dev(np.zeros(64, np.float32))
    a.type, a.glu_op, a.w_gate, a.w_up = B.Q4_K, 2, w.data_ptr(), w.data_ptr()
    a.gate_row_stride = a.up_row_stride = B.row_size(B.Q4_K, 512)
    a.m, a.k, a.act_q, a.act_ds, a.dst = 48, 512, aq.data_ptr(), ads.data_ptr(), d.data_ptr()
    assert K.lib.mi355_glu_matvec(a, None) == -2                       # m not a multiple of 32
    a.m, a.out_q, a.out_ds = 64, aq.data_ptr(), ads.data_ptr()
    assert K.lib.mi355_glu_matvec(a, None) == -2                       # q8_1 output on top of the activations being read
    a.out_ds = None
    assert K.lib.mi355_glu_matvec(a, None) == -4                       # out_q without out_ds
    a.out_q, a.glu_op = None, 7
    assert K.lib.mi355_glu_matvec(a, None) == -6                       # unknown activation


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n,mode", [(4096, 4096, 1, "separate"), (4096, 14336, 1, "in_place"), (130, 512, 3, "on_residual"), (7, 1024, 8, "separate")])
def test_mmvq_residual_epilogue_equals_separate_add(ops, t, m, k, n, mode):
    """mat-vec + ADD in one launch (mi355_mmvq_args.residual / .sum): the sum equals result + residual to the bit, the
    result itself is still stored unless the sum takes its place (ADD in place on the mat-vec result)."""
    rng = np.random.default_rng(m + k + n)
    x = dev(rng.standard_normal((n, k)).astype(np.float32))
    aq, ads = ops.quantize_act(x)
    w = dev(B.random_blocks(t, m, k, seed=21))
    res = dev(rng.standard_normal((n, m)).astype(np.float32))
    ref = ops.mul_mat_vec_q(t, w, m, k, aq, ads, n)
    exp = ref + res
    dst = torch.full((n, m), float("nan"), device="cuda")
    if mode == "separate":
        out = torch.full((n, m), float("nan"), device="cuda")
        ops.mul_mat_vec_q(t, w, m, k, aq, ads, n, dst=dst, residual=res, sum_out=out)
        assert torch.equal(dst.view(torch.int32), ref.view(torch.int32))
    elif mode == "in_place":
        ops.mul_mat_vec_q(t, w, m, k, aq, ads, n, dst=dst, residual=res, sum_out=dst)
        out = dst
    else:
        out = res.clone()
        ops.mul_mat_vec_q(t, w, m, k, aq, ads, n, dst=dst, residual=out, sum_out=out)
        assert torch.equal(dst.view(torch.int32), ref.view(torch.int32))
    assert torch.isfinite(out).all() and torch.equal(out.view(torch.int32), exp.view(torch.int32))


@pytest.mark.parametrize("t", [B.Q4_K, B.Q6_K, B.Q8_0])
@pytest.mark.parametrize("m,k", [(14336, 4096), (96, 1024), (2816, 8192), (64, 96)])
def test_glu_matvec_norm_prologue_equals_separate_norm(ops, t, m, k):
    """mi355_glu_matvec with the RMS_NORM -> MUL(weight) prologue == mi355_rms_norm_mul_quant_f32 followed by the plain
    mi355_glu_matvec, to the bit (rows of 4096 and fewer: the 256-thread norm kernel; 8192: its 512-thread form)."""
    import llm_amd.kernels as K
    if k % B.BLCK_SIZE[t] != 0:
        pytest.skip("k not a multiple of the block size")
    rng = np.random.default_rng(m + k)
    x, nw = (dev(rng.standard_normal(sh).astype(np.float32)) for sh in ((1, k), (k,)))
    wg, wu = dev(B.random_blocks(t, m, k, seed=31)), dev(B.random_blocks(t, m, k, seed=32))
    T, s = K.Tensor.of, torch.cuda.current_stream().cuda_stream
    n0 = torch.empty_like(x)
    aq = torch.empty((1, k), dtype=torch.int8, device="cuda")
    ads = torch.empty((1, k // 32, 2), dtype=torch.float32, device="cuda")
    K.check(K.lib.mi355_rms_norm_mul_quant_f32(T(x), nw.data_ptr(), T(n0), 1e-5, aq.data_ptr(), ads.data_ptr(), s), "rms_norm_mul_quant")
    d0, g0, u0, q0, s0 = ops.glu_matvec(t, wg, wu, m, k, aq, ads)
    n1 = torch.full_like(x, float("nan"))
    d1, g1, u1, q1, s1 = ops.glu_matvec(t, wg, wu, m, k, None, None, norm=(x, nw, 1e-5, n1))
    assert torch.isfinite(d1).all() and float(d1.abs().max()) > 0
    assert torch.equal(n1.view(torch.int32), n0.view(torch.int32))
    for a, b in ((d0, d1), (g0, g1), (u0, u1), (s0, s1)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert torch.equal(q0, q1)


@pytest.mark.parametrize("mode,n_dims,nh", [(0, 128, 32), (2, 128, 8), (0, 96, 8), (2, 64, 32)])
def test_rope_head_walking_kernel_is_bit_identical(ops, mode, n_dims, nh):
    """prefill-sized ROPE (cos / sin computed once per (pair, token), heads walked by the lane) == the per-pair kernel that small
    launches take, to the bit: 512 tokens in one call vs the same tokens 64 at a time."""
    hd, nt = 128, 512
    rng = np.random.default_rng(mode + n_dims + nh)
    x = dev(rng.standard_normal((nt, nh, hd)).astype(np.float32))
    pos = torch.arange(1000, 1000 + nt, dtype=torch.int32, device="cuda")
    kw = dict(freq_base=500000.0, freq_scale=0.5 if n_dims == 96 else 1.0, ext_factor=1.0 if n_dims == 96 else 0.0, n_ctx_orig=4096)
    big = ops.rope(x, pos, n_dims, mode, **kw)
    small = torch.cat([ops.rope(x[i:i + 64].contiguous(), pos[i:i + 64].contiguous(), n_dims, mode, **kw) for i in range(0, nt, 64)])
    assert torch.isfinite(big).all() and torch.equal(big.view(torch.int32), small.view(torch.int32))


@pytest.mark.parametrize("rows,n,with_add", [(130, 4096, True), (200, 1024, False), (129, 8192, True), (3, 96, False)])
def test_norm_and_glu_bf16_forms(ops, rows, n, with_add):
    """prefill producers: mi355_rms_norm_mul_bf16_f32 / mi355_glu_bf16_f32 == the plain kernels bit for bit, and their bf16
    rows == mi355_act_bf16 of that f32 result."""
    import llm_amd.kernels as K
    rng = np.random.default_rng(rows + n)
    x, y, w = (dev(rng.standard_normal(sh).astype(np.float32)) for sh in ((rows, n), (rows, n), (n,)))
    T, s = K.Tensor.of, torch.cuda.current_stream().cuda_stream
    su0, d0, su1, d1 = (torch.empty_like(x) for _ in range(4))
    bf = torch.empty((rows, n), dtype=torch.int16, device="cuda")
    if with_add:
        K.check(K.lib.mi355_add_rms_norm_mul_f32(T(x), T(y), T(su0), w.data_ptr(), T(d0), 1e-5, None, None, s), "add_rms_norm_mul")
        K.check(K.lib.mi355_rms_norm_mul_bf16_f32(T(x), T(y), T(su1), w.data_ptr(), T(d1), 1e-5, bf.data_ptr(), s), "rms_norm_mul_bf16")
        assert torch.equal(su0.view(torch.int32), su1.view(torch.int32))
    else:
        K.check(K.lib.mi355_rms_norm_mul_f32(T(x), w.data_ptr(), T(d0), 1e-5, s), "rms_norm_mul")
        K.check(K.lib.mi355_rms_norm_mul_bf16_f32(T(x), None, None, w.data_ptr(), T(d1), 1e-5, bf.data_ptr(), s), "rms_norm_mul_bf16")
    assert torch.isfinite(d1).all() and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
    assert torch.equal(bf, ops.act_bf16(d1))
    g0, g1 = torch.empty_like(x), torch.empty_like(x)
    K.check(K.lib.mi355_glu_f32(2, T(x), T(y), T(g0), 0, s), "glu")
    K.check(K.lib.mi355_glu_bf16_f32(2, T(x), T(y), T(g1), 0, bf.data_ptr(), s), "glu_bf16")
    assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)) and torch.equal(bf, ops.act_bf16(g1))


@pytest.mark.parametrize("rows,n,with_add", [(16, 4096, True), (5, 4096, False), (32, 8192, True), (12, 256, False), (7, 1024, True)])
def test_norm_q8k_form(ops, rows, n, with_add):
    """5..32-row batches: mi355_rms_norm_mul_q8k_f32 == the plain kernels bit for bit, and its q8_K planes (quants, block
    scales, sums of 16) == mi355_quantize_act_q8k of that f32 result. One row holds a zero block and a repeated maximum."""
    import llm_amd.kernels as K
    rng = np.random.default_rng(rows * 7 + n)
    xh = rng.standard_normal((rows, n)).astype(np.float32)
    yh = rng.standard_normal((rows, n)).astype(np.float32)
    xh[1, :256] = 0.0; yh[1, :256] = 0.0                 # a block of zeros: d = 0, all quants 0
    xh[2, 5] = xh[2, 77] = 9.0; yh[2, 5] = yh[2, 77] = 0.0    # |max| twice in one block: the first occurrence gives the sign
    if n > 256:
        xh[2, 300] = 9.0; xh[2, 400] = -9.0; yh[2, 300] = yh[2, 400] = 0.0
    x, y = dev(xh), dev(yh)
    w = dev(np.ones(n, dtype=np.float32) if rows == 7 else rng.standard_normal(n).astype(np.float32))
    T, s = K.Tensor.of, torch.cuda.current_stream().cuda_stream
    su0, d0, su1, d1 = (torch.empty_like(x) for _ in range(4))
    q = torch.empty((rows, n), dtype=torch.int8, device="cuda")
    d = torch.empty((rows, n // 256), dtype=torch.float32, device="cuda")
    bs = torch.empty((rows, n // 16), dtype=torch.int16, device="cuda")
    if with_add:
        K.check(K.lib.mi355_add_rms_norm_mul_f32(T(x), T(y), T(su0), w.data_ptr(), T(d0), 1e-5, None, None, s), "add_rms_norm_mul")
        K.check(K.lib.mi355_rms_norm_mul_q8k_f32(T(x), T(y), T(su1), w.data_ptr(), T(d1), 1e-5, q.data_ptr(), d.data_ptr(), bs.data_ptr(), s), "q8k")
        assert torch.equal(su0.view(torch.int32), su1.view(torch.int32))
    else:
        K.check(K.lib.mi355_rms_norm_mul_f32(T(x), w.data_ptr(), T(d0), 1e-5, s), "rms_norm_mul")
        K.check(K.lib.mi355_rms_norm_mul_q8k_f32(T(x), None, None, w.data_ptr(), T(d1), 1e-5, q.data_ptr(), d.data_ptr(), bs.data_ptr(), s), "q8k")
    assert torch.isfinite(d1).all() and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
    rq, rd, rbs = ops.quantize_act_q8k(d1)
    assert torch.equal(q, rq.view(rows, n))
    assert torch.equal(d.view(torch.int32), rd.view(rows, n // 256).view(torch.int32))
    assert torch.equal(bs, rbs.view(rows, n // 16))
    # the GLU producer of the same planes (ffn_down's input)
    g0, g1 = torch.empty_like(x), torch.empty_like(x)
    for op in (2, 1):
        K.check(K.lib.mi355_glu_f32(op, T(x), T(y), T(g0), 0, s), "glu")
        K.check(K.lib.mi355_glu_q8k_f32(op, T(x), T(y), T(g1), 0, q.data_ptr(), d.data_ptr(), bs.data_ptr(), s), "glu_q8k")
        assert torch.equal(g0.view(torch.int32), g1.view(torch.int32))
        rq, rd, rbs = ops.quantize_act_q8k(g1)
        assert torch.equal(q, rq.view(rows, n)) and torch.equal(bs, rbs.view(rows, n // 16))
        assert torch.equal(d.view(torch.int32), rd.view(rows, n // 256).view(torch.int32))


@pytest.mark.parametrize("rows,n", [(1, 4096), (3, 1024), (8, 8192), (2, 96), (1, 4128)])
@pytest.mark.parametrize("with_add", [False, True])
def test_fused_norm_matches_separate_ops(ops, rows, n, with_add):
    """[ADD ->] RMS_NORM -> MUL(weight) [-> q8_1] in one launch: the f32 outputs equal the separate kernels' bit for
    bit and the fused q8_1 form equals mi355_quantize_act of that output; values agree with numpy (ops.cpp rms_norm)."""
    import llm_amd.kernels as K
    rng = np.random.default_rng(rows * n)
    x, y, w = (dev(rng.standard_normal(sh).astype(np.float32)) for sh in ((rows, n), (rows, n), (n,)))
    T, s = K.Tensor.of, torch.cuda.current_stream().cuda_stream
    su, d, d2, nrm = (torch.empty_like(x) for _ in range(4))
    aq = torch.empty((rows, n), dtype=torch.int8, device="cuda")
    ads = torch.empty((rows, n // 32, 2), dtype=torch.float32, device="cuda")
    quant = n % 32 == 0
    if with_add:
        K.check(K.lib.mi355_add_rms_norm_mul_f32(T(x), T(y), T(su), w.data_ptr(), T(d), 1e-5, aq.data_ptr() if quant else None,
                                                 ads.data_ptr() if quant else None, s), "add_rms_norm_mul")
        K.check(K.lib.mi355_binary_f32(0, T(x), T(y), T(d2), s), "add")
        assert torch.equal(su.view(torch.int32), d2.view(torch.int32))
        xin = su
    else:
        if quant:
            K.check(K.lib.mi355_rms_norm_mul_quant_f32(T(x), w.data_ptr(), T(d), 1e-5, aq.data_ptr(), ads.data_ptr(), s), "rms_norm_mul_quant")
        else:
            K.check(K.lib.mi355_rms_norm_mul_f32(T(x), w.data_ptr(), T(d), 1e-5, s), "rms_norm_mul")
        xin = x
    # separate kernels: RMS_NORM, then MUL with the broadcast weight
    K.check(K.lib.mi355_rms_norm_f32(T(xin), T(nrm), 1e-5, s), "rms_norm")
    K.check(K.lib.mi355_binary_f32(1, T(nrm), T(w), T(d2), s), "mul")
    assert torch.equal(d.view(torch.int32), d2.view(torch.int32))
    xi = xin.cpu().numpy().astype(np.float64)
    ref = xi / np.sqrt((xi ** 2).mean(axis=1, keepdims=True) + 1e-5) * w.cpu().numpy()
    assert np.allclose(d.cpu().numpy(), ref, rtol=2e-5, atol=1e-6)
    if quant:
        aq2, ads2 = ops.quantize_act(d)
        assert torch.equal(aq, aq2.view(rows, n)) and torch.equal(ads.view(-1).view(torch.int32), ads2.view(-1).view(torch.int32))


def _mmid_cases():
    g = load_gold("mul_mat_id.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _mmid_cases())
def test_mul_mat_id_golden(ops, tag):
    g = load_gold("mul_mat_id.gold")
    t, m, k, n_exp, n_used, n_tok, b1 = (int(x) for x in g["shape." + tag])
    want = g["dst." + tag].reshape(n_tok, n_used, m)
    b = g["x." + tag].reshape(n_tok, b1, k)
    ids = g["ids." + tag].reshape(n_tok, n_used)
    got = ops.mul_mat_id(t, dev(g["w." + tag]), dev(b), dev(ids), m, n_exp).cpu().numpy()
    assert O.nmse(want, got) <= 5e-4
    contract = O.mul_mat_id(t, g["w." + tag], b, ids, m, k, n_exp, n_used, n_tok, b1, "q8_1")
    assert O.nmse(contract, got) <= 1e-10


# ---------------------------------------------------------------- error behaviour through the C ABI
def _fa_cases():
    g = load_gold("flash_attn_ext.gold")
    return [k[len("shape."):] for k in g if k.startswith("shape.")]


@pytest.mark.parametrize("tag", _fa_cases())
def test_flash_attn_golden(ops, tag):
    """FLASH_ATTN_EXT through the C ABI vs the ggml-cpu outputs held in tests/golden (NMSE <= 5e-4,
    tests/test-backend-ops.cpp:3599-3601) and vs the oracle's f64 evaluation."""
    g = load_gold("flash_attn_ext.gold")
    d, nh, nhkv, nq, nkv, tkv, has_mask, nqp = (int(x) for x in g["shape." + tag])
    scale, max_bias, softcap = (float(x) for x in g["params." + tag])
    q = dev(np.frombuffer(g["q." + tag], np.float32).reshape(nh, nq, d))
    k = dev(np.frombuffer(g["k." + tag], np.uint8).reshape(nhkv, nkv, -1))
    v = dev(np.frombuffer(g["v." + tag], np.uint8).reshape(nhkv, nkv, -1))
    mask = dev(np.frombuffer(g["mask." + tag], np.int16).reshape(nqp, nkv)) if has_mask else None
    want = np.frombuffer(g["dst." + tag], np.float32).reshape(nq, nh, d)
    for use_ws in (True, False):
        got = ops.flash_attn_ext(q, k, v, mask, tkv, nkv, scale, max_bias, softcap, use_workspace=use_ws).cpu().numpy()
        assert np.isfinite(got).all() and O.nmse(want, got) <= 5e-4, (tag, use_ws)


@pytest.mark.parametrize("d,nh,nhkv,nq,nkv,causal", [(128, 8, 2, 96, 200, True), (128, 4, 4, 33, 64, False), (64, 8, 1, 130, 333, True),
                                                      (128, 32, 8, 512, 512, True), (128, 8, 2, 1, 300, False), (128, 8, 2, 4, 4096, False),
                                                      # long decode contexts: ragged slice ends, 4 heads per workgroup from n_kv 16384
                                                      # (GQA 4), 8 per workgroup (GQA 8), no GQA, two tokens
                                                      (128, 32, 8, 1, 20000, False), (128, 8, 2, 2, 16411, False), (128, 64, 8, 1, 5000, False),
                                                      (64, 8, 8, 1, 9000, False), (128, 32, 8, 1, 1000, True),
                                                      # prefill tile kernel: GQA 2 (two heads x two query blocks per workgroup), GQA 3 (one head x four),
                                                      # a long causal prompt whose upper-triangle blocks are never staged
                                                      (128, 4, 2, 70, 130, True), (128, 6, 2, 50, 90, True), (128, 8, 2, 700, 700, True),
                                                      # a 512-token ubatch late in a long prompt: KV blocks dealt to two workgroups per unit group + merge
                                                      (128, 32, 8, 512, 2048, True), (64, 8, 2, 100, 1100, True), (128, 6, 2, 33, 1029, False),
                                                      # head sizes between the ones the kernels are built for (the reference's list, fattn.cu:271-346, and
                                                      # test-backend-ops' hs values 40 / 72 / 80 / 96 / 112; 192 and 24 for the rule itself): decode and
                                                      # prefill shapes, with and without the KV split, ragged query blocks, the short-context form
                                                      (80, 8, 8, 1, 300, False), (80, 8, 2, 70, 130, True), (80, 32, 32, 1, 5000, False), (80, 8, 8, 512, 1536, True),
                                                      (96, 8, 2, 96, 200, True), (96, 32, 32, 1, 5000, False), (96, 6, 2, 3, 700, True), (96, 32, 32, 512, 2048, True),
                                                      (112, 4, 2, 33, 64, True), (112, 8, 2, 2, 9000, False), (40, 4, 4, 3, 100, False), (40, 4, 2, 64, 64, True),
                                                      (40, 8, 8, 1, 4000, False), (72, 6, 2, 50, 90, True), (72, 6, 2, 1, 1000, False), (72, 8, 8, 130, 1333, True),
                                                      (192, 4, 2, 40, 100, True), (192, 4, 2, 1, 600, False), (24, 4, 4, 2, 77, False), (24, 4, 4, 40, 77, True),
                                                      # head size 256 on the matrix cores (Gemma): GQA 2 and 1, the KV shares + merge, ragged ends
                                                      (256, 4, 2, 96, 200, True), (256, 16, 8, 512, 512, True), (256, 8, 8, 70, 1100, True), (256, 16, 8, 512, 1536, True)])
def test_flash_attn_f16_vs_f64_oracle(ops, d, nh, nhkv, nq, nkv, causal):
    """prefill shapes take the matrix-core tile kernel (n_q >= 32), decode shapes the split-KV vec kernel; both
    against the oracle's double-precision evaluation of the op on the same f16 K/V."""
    rng = np.random.default_rng(d + nq + nkv)
    q = rng.standard_normal((nh, nq, d)).astype(np.float32)
    k = rng.standard_normal((nhkv, nkv, d)).astype(np.float16)
    v = rng.standard_normal((nhkv, nkv, d)).astype(np.float16)
    nqp = (nq + 63) // 64 * 64
    mask = np.zeros((nqp, nkv), np.float16)
    if causal:
        off = nkv - nq
        for i in range(nq):
            mask[i, max(0, i + off + 1):] = -np.inf
    scale = 1.0 / np.sqrt(d)
    ku, vu = k.view(np.uint8).reshape(nhkv, nkv, -1), v.view(np.uint8).reshape(nhkv, nkv, -1)
    got = ops.flash_attn_ext(dev(q), dev(ku), dev(vu), dev(mask.view(np.int16)), B.F16, nkv, scale).cpu().numpy()
    ref = O.flash_attn_ext(q, ku, vu, mask.view(np.uint16), O.F16, O.F16, d, d, nq, nkv, nh, nhkv, scale, exact=True)
    assert np.isfinite(got).all() and O.nmse(ref, got) <= 2e-5, O.nmse(ref, got)


@pytest.mark.parametrize("tkv", [B.Q8_0, B.Q4_0])
@pytest.mark.parametrize("d,nh,nhkv,nq,nkv", [(128, 8, 2, 96, 200), (64, 4, 4, 33, 130), (128, 8, 2, 3, 300), (128, 32, 8, 1, 17000),
                                              (128, 8, 2, 512, 2048),       # SURVEY config 3: a 512-token ubatch at n_kv 2048 (GQA 4)
                                              (96, 8, 2, 96, 200), (96, 8, 2, 1, 3000), (96, 8, 8, 3, 100), (256, 4, 2, 40, 300), (256, 4, 2, 2, 3000),
                                              (32, 4, 4, 40, 100), (32, 4, 4, 1, 100), (160, 4, 2, 33, 130)])
def test_flash_attn_quantized_kv_vs_f64_oracle(ops, tkv, d, nh, nhkv, nq, nkv):
    """q8_0 / q4_0 KV cache: the tile kernel dequantizes K/V blocks to f16 while staging (n_q >= 32), the vec
    kernel in registers (decode); both vs the oracle's f64 evaluation on the same quantized rows."""
    rng = np.random.default_rng(d + nq + nkv + tkv)
    q = rng.standard_normal((nh, nq, d)).astype(np.float32)
    scale_blocks = 20.0   # random_blocks' scales are weight-sized; attention rows are O(1)
    k = B.random_blocks(tkv, nhkv * nkv, d, seed=11).reshape(nhkv, nkv, -1)
    v = B.random_blocks(tkv, nhkv * nkv, d, seed=12).reshape(nhkv, nkv, -1)
    nqp = (nq + 63) // 64 * 64
    mask = np.zeros((nqp, nkv), np.float16)
    off = nkv - nq
    for i in range(nq):
        mask[i, i + off + 1:] = -np.inf
    got = ops.flash_attn_ext(dev(q), dev(k), dev(v), dev(mask.view(np.int16)), tkv, nkv, scale_blocks).cpu().numpy()
    ref = O.flash_attn_ext(q, k, v, mask.view(np.uint16), tkv, tkv, d, d, nq, nkv, nh, nhkv, scale_blocks, exact=True)
    assert np.isfinite(got).all() and O.nmse(ref, got) <= 5e-5, O.nmse(ref, got)


@pytest.mark.parametrize("d,nh,nhkv,nq,nkv", [(128, 8, 2, 1, 300), (64, 4, 4, 33, 130), (128, 8, 2, 3, 5000), (256, 4, 2, 2, 100), (128, 32, 8, 70, 200),
                                              (80, 8, 2, 1, 300), (96, 4, 4, 33, 130), (112, 8, 2, 3, 5000), (40, 4, 2, 70, 200)])
def test_flash_attn_bf16_kv_vs_f64_oracle(ops, d, nh, nhkv, nq, nkv):
    """bf16 KV cache (test-backend-ops.cpp:4866-4906 type_KV list; the vec kernel for any n_q): K / V values that both bf16 and f16
    hold exactly, so the oracle's f64 evaluation on the f16 rows is the reference; the query is rounded to bf16 as the CPU path
    does (ops.cpp:8061-8066), which is all the difference there is (NMSE <= 5e-5; the reference's own bar is 5e-4)."""
    rng = np.random.default_rng(d + nq + nkv)
    q = rng.standard_normal((nh, nq, d)).astype(np.float32)
    k = (rng.integers(-64, 65, (nhkv, nkv, d)) / 32.0).astype(np.float32)
    v = (rng.integers(-64, 65, (nhkv, nkv, d)) / 32.0).astype(np.float32)
    kb, vb = ((x.view(np.uint32) >> 16).astype(np.uint16) for x in (k, v))
    assert np.array_equal((kb.astype(np.uint32) << 16).view(np.float32), k)
    nqp = (nq + 63) // 64 * 64
    mask = np.zeros((nqp, nkv), np.float16)
    off = nkv - nq
    for i in range(nq):
        mask[i, i + off + 1:] = -np.inf
    scale = 1.0 / np.sqrt(d)
    ku, vu = (x.view(np.uint8).reshape(nhkv, nkv, -1) for x in (kb, vb))
    kh, vh = (x.astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1) for x in (k, v))
    ref = O.flash_attn_ext(q, kh, vh, mask.view(np.uint16), O.F16, O.F16, d, d, nq, nkv, nh, nhkv, scale, exact=True)
    for use_ws in (True, False):
        got = ops.flash_attn_ext(dev(q), dev(ku), dev(vu), dev(mask.view(np.int16)), B.BF16, nkv, scale, use_workspace=use_ws).cpu().numpy()
        assert np.isfinite(got).all() and O.nmse(ref, got) <= 5e-5, O.nmse(ref, got)


@pytest.mark.parametrize("d", [128, 96])
@pytest.mark.parametrize("nq,nkv,use_ws", [(1, 300, True), (1, 300, False), (4, 4096, True), (8, 96, True)])
def test_flash_attn_fused_q8_1_output(ops, nq, nkv, use_ws, d):
    """mi355_flash_attn_ext_quant: dst equals the plain call bit for bit and the fused q8_1 form equals
    mi355_quantize_act(dst); run twice on one workspace (the arrival counters re-arm themselves)."""
    nh, nhkv = 8, 2
    rng = np.random.default_rng(nq + nkv)
    q = dev(rng.standard_normal((nh, nq, d)).astype(np.float32))
    k = dev(rng.standard_normal((nhkv, nkv, d)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1))
    v = dev(rng.standard_normal((nhkv, nkv, d)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1))
    mask = dev(np.zeros((64, nkv), np.float16).view(np.int16))
    plain = ops.flash_attn_ext(q, k, v, mask, B.F16, nkv, 0.088, use_workspace=use_ws)
    dst, aq, ads = ops.flash_attn_ext(q, k, v, mask, B.F16, nkv, 0.088, use_workspace=use_ws, quant=True)
    assert torch.equal(plain.view(torch.int32), dst.view(torch.int32))
    aq2, ads2 = ops.quantize_act(dst.view(nq, nh * d))
    assert torch.equal(aq, aq2) and torch.equal(ads.view(torch.int32), ads2.view(torch.int32))


@pytest.mark.parametrize("tkv", [B.F16, B.Q8_0])
@pytest.mark.parametrize("dk,dv,nh,nhkv,nq,nkv", [(192, 128, 8, 2, 1, 700), (192, 128, 4, 4, 3, 100), (192, 128, 8, 2, 70, 130), (192, 128, 16, 16, 512, 1536),
                                                  (128, 64, 8, 2, 40, 100), (96, 32, 4, 2, 2, 5000)])
def test_flash_attn_v_head_size_smaller_than_k(ops, tkv, dk, dv, nh, nhkv, nq, nkv):
    """K and V rows of different sizes (DeepSeek-V2's 192 / 128; test-backend-ops.cpp hsk / hsv): decode and prefill kernels vs the
    oracle's f64 evaluation on the same rows."""
    rng = np.random.default_rng(dk + dv + nq + nkv)
    q = rng.standard_normal((nh, nq, dk)).astype(np.float32)
    if tkv == B.F16:
        k = rng.standard_normal((nhkv, nkv, dk)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1)
        v = rng.standard_normal((nhkv, nkv, dv)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1)
        scale = 1.0 / np.sqrt(dk)
    else:
        k = B.random_blocks(tkv, nhkv * nkv, dk, seed=11).reshape(nhkv, nkv, -1)
        v = B.random_blocks(tkv, nhkv * nkv, dv, seed=12).reshape(nhkv, nkv, -1)
        scale = 20.0
    mask = np.zeros(((nq + 63) // 64 * 64, nkv), np.float16)
    for i in range(nq):
        mask[i, i + nkv - nq + 1:] = -np.inf
    ref = O.flash_attn_ext(q, k, v, mask.view(np.uint16), tkv, tkv, dk, dv, nq, nkv, nh, nhkv, scale, exact=True)
    for use_ws in (True, False):
        got = ops.flash_attn_ext(dev(q), dev(k), dev(v), dev(mask.view(np.int16)), tkv, nkv, scale, dv=dv, use_workspace=use_ws).cpu().numpy()
        assert got.shape == ref.shape and np.isfinite(got).all() and O.nmse(ref, got) <= 5e-5, O.nmse(ref, got)


@pytest.mark.parametrize("max_bias,softcap", [(8.0, 0.0), (0.0, 10.0), (4.0, 30.0)])
@pytest.mark.parametrize("d,nh,nhkv,nq,nkv", [(128, 8, 2, 96, 200), (256, 8, 4, 70, 300), (64, 12, 12, 33, 64), (128, 32, 8, 512, 1536)])
def test_flash_attn_prefill_alibi_and_softcap(ops, d, nh, nhkv, nq, nkv, max_bias, softcap):
    """ALiBi slopes (max_bias; the mask carries the position bias, ggml.c:4733-4786 / ops.cpp:8041-8047) and the logit softcap (Gemma-2,
    ops.cpp:8094-8098) on the matrix-core kernel -- prefill-sized batches -- and, with the same inputs cut to 3 queries, on the vec kernel."""
    rng = np.random.default_rng(d + nq + nkv)
    k = rng.standard_normal((nhkv, nkv, d)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1)
    v = rng.standard_normal((nhkv, nkv, d)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1)
    scale = 1.0 / np.sqrt(d)
    for n in (nq, 3):
        q = (3.0 * rng.standard_normal((nh, n, d))).astype(np.float32)      # scores that reach the cap
        mask = np.zeros(((n + 63) // 64 * 64, nkv), np.float16)
        for i in range(n):
            mask[i, :] = -0.25 * np.abs(np.arange(nkv) - (i + nkv - n))      # a position bias
            mask[i, i + nkv - n + 1:] = -np.inf
        ref = O.flash_attn_ext(q, k, v, mask.view(np.uint16), O.F16, O.F16, d, d, n, nkv, nh, nhkv, scale, max_bias, softcap, exact=True)
        got = ops.flash_attn_ext(dev(q), dev(k), dev(v), dev(mask.view(np.int16)), B.F16, nkv, scale, max_bias, softcap).cpu().numpy()
        assert np.isfinite(got).all() and O.nmse(ref, got) <= 5e-5, (n, O.nmse(ref, got))


def test_flash_attn_head_size_rules(ops):
    """What the op refuses, by return code: head sizes that are not a multiple of 8 or above 256 (MI355_E_UNSUPPORTED: the plugin's supports_op
    says no to the same), quantized K/V rows that are not whole 32-blocks (MI355_E_SHAPE), and the fused q8_1 output for a head size whose
    32-blocks would straddle heads (MI355_E_UNSUPPORTED: the caller quantizes the row itself)."""
    import llm_amd.kernels as K
    def run(d, tkv=B.F16, rb=None, **kw):
        q = torch.zeros(4, 1, d, device="cuda")
        kv = torch.zeros(4, 64, rb if rb else d * 2, dtype=torch.uint8, device="cuda")
        return ops.flash_attn_ext(q, kv, kv, None, tkv, 64, 1.0, **kw)
    for d in (36, 260, 4, 512):
        with pytest.raises(K.Mi355Error) as e:
            run(d)
        assert e.value.code == -6      # MI355_E_UNSUPPORTED
    with pytest.raises(K.Mi355Error) as e:
        run(80, B.Q8_0, rb=85)
    assert e.value.code == -2          # MI355_E_SHAPE
    with pytest.raises(K.Mi355Error) as e:
        run(80, quant=True)
    assert e.value.code == -6
    assert torch.isfinite(run(80)).all() and torch.isfinite(run(8)).all()


@pytest.mark.parametrize("tkv,d,nh,nhkv,nq,nkv", [(B.F16, 128, 32, 8, 512, 512), (B.F16, 128, 6, 2, 50, 90), (B.Q8_0, 64, 4, 4, 130, 333), (B.F16, 128, 8, 2, 8, 300),
                                                  (B.F16, 128, 32, 8, 512, 1536), (B.F16, 80, 8, 2, 100, 300), (B.Q8_0, 96, 4, 4, 130, 333), (B.F16, 256, 8, 4, 512, 1536),
                                                  (B.F16, 72, 6, 2, 33, 64)])
def test_flash_attn_fused_bf16_output(ops, tkv, d, nh, nhkv, nq, nkv):
    """mi355_flash_attn_ext_bf16 (prefill): dst equals the plain call bit for bit and the bf16 rows equal mi355_act_bf16(dst); ragged
    query blocks; a decode-sized batch (vec kernel) is refused, not served by something else."""
    import llm_amd.kernels as K
    rng = np.random.default_rng(nq + nkv + d)
    q = dev(rng.standard_normal((nh, nq, d)).astype(np.float32))
    if tkv == B.F16:
        k = dev(rng.standard_normal((nhkv, nkv, d)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1))
        v = dev(rng.standard_normal((nhkv, nkv, d)).astype(np.float16).view(np.uint8).reshape(nhkv, nkv, -1))
    else:
        k = dev(B.random_blocks(tkv, nhkv * nkv, d, seed=11).reshape(nhkv, nkv, -1))
        v = dev(B.random_blocks(tkv, nhkv * nkv, d, seed=12).reshape(nhkv, nkv, -1))
    m = np.zeros(((nq + 63) // 64 * 64, nkv), np.float16)
    for i in range(nq):
        m[i, i + nkv - nq + 1:] = -np.inf
    mask = dev(m.view(np.int16))
    if nq < 32:
        with pytest.raises(K.Mi355Error):
            ops.flash_attn_ext(q, k, v, mask, tkv, nkv, 0.088, bf16=True)
        return
    plain = ops.flash_attn_ext(q, k, v, mask, tkv, nkv, 0.088)
    dst, abf = ops.flash_attn_ext(q, k, v, mask, tkv, nkv, 0.088, bf16=True)
    assert torch.isfinite(dst).all()
    assert torch.equal(plain.view(torch.int32), dst.view(torch.int32))
    assert torch.equal(abf, ops.act_bf16(dst.view(nq, nh * d)).view(torch.int16).view(nq, nh * d))


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n_exp,n_used,n_tok,b1", [(192, 256, 8, 2, 40, 1), (130, 512, 4, 2, 200, 2), (64, 256, 8, 3, 33, 3), (256, 1024, 8, 2, 512, 1),
                                                       # >= 1024 rows: the role-split 128 x 256 tile in MUL_MAT_ID mode (tiles of 256 pairs: several per expert,
                                                       # ragged last tiles, ragged row tile)
                                                       (1024, 512, 8, 2, 300, 1), (1100, 256, 4, 2, 700, 2), (2048, 1024, 8, 2, 64, 1)])
def test_mul_mat_id_grouped_gemm(ops, t, m, k, n_exp, n_used, n_tok, b1):
    """prefill-sized MUL_MAT_ID: device-side bucketing by expert + one bf16-MFMA GEMM launch, vs exact
    dequant(as[ids]) . b in f64 (budget of the op: NMSE 5e-4, tests/test-backend-ops.cpp:2357-2359), and vs the
    per-pair mat-vec path; skewed routing (one expert takes half the pairs) and an unused expert included."""
    if k % B.BLCK_SIZE[t]:
        pytest.skip("k not a multiple of the block size")
    rng = np.random.default_rng(m + k + n_tok)
    w = B.random_blocks(t, n_exp * m, k, seed=5)
    b = rng.standard_normal((n_tok, b1, k)).astype(np.float32)
    ids = np.stack([rng.permutation(n_exp - 1)[:n_used] for _ in range(n_tok)]).astype(np.int32)   # expert n_exp-1 is never used
    ids[: n_tok // 2, 0] = 0; ids[: n_tok // 2, 1] = 1 + (np.arange(n_tok // 2) % max(1, n_exp - 2))  # skew towards expert 0
    got = ops.mul_mat_id_dq(t, dev(w), dev(b), dev(ids), m, n_exp).cpu().numpy()
    wf = O.dequantize_row(t, w, n_exp * m * k).reshape(n_exp, m, k).astype(np.float64)
    ref = np.empty((n_tok, n_used, m))
    for tt in range(n_tok):
        for s_ in range(n_used):
            ref[tt, s_] = wf[ids[tt, s_]] @ b[tt, s_ % b1].astype(np.float64)
    err = float(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert np.isfinite(got).all() and err < 2e-5, err
    vec = ops.mul_mat_id(t, dev(w), dev(b), dev(ids), m, n_exp).cpu().numpy()
    assert O.nmse(vec, got) <= 5e-4


@pytest.mark.parametrize("t", B.WEIGHT_TYPES)
@pytest.mark.parametrize("m,k,n_exp,n_used,n_tok,b1", [(96, 256, 4, 2, 1, 1), (130, 512, 8, 2, 8, 1), (257, 1024, 8, 2, 64, 2), (64, 2048, 8, 3, 100, 1),
                                                       (512, 4096, 8, 2, 16, 1)])
def test_mul_mat_id_skinny(ops, t, m, k, n_exp, n_used, n_tok, b1):
    """MUL_MAT_ID at config-5 batch sizes (T = 8..64): bucketing on the device, every expert streamed once per 32 of its
    columns through the skinny int8-MFMA kernels.  vs exact dequant(as[ids]) . b in f64 within the op's budget (the
    activations are q8_K / q8_1 quantized: ~1e-5), and vs the per-pair mat-vec path; skewed routing (one expert takes
    half the pairs, more than one pass of 32) and an unused expert included; rows not written stay NaN-free."""
    rng = np.random.default_rng(m + k + n_tok)
    w = B.random_blocks(t, n_exp * m, k, seed=5)
    b = rng.standard_normal((n_tok, b1, k)).astype(np.float32)
    ids = np.stack([rng.permutation(n_exp - 1)[:n_used] for _ in range(n_tok)]).astype(np.int32)   # expert n_exp-1 is never used
    ids[: n_tok // 2, 0] = 0; ids[: n_tok // 2, 1] = 1 + (np.arange(n_tok // 2) % max(1, n_exp - 2))  # skew towards expert 0
    got = ops.mul_mat_id_skinny(t, dev(w), dev(b), dev(ids), m, n_exp).cpu().numpy()
    wf = O.dequantize_row(t, w, n_exp * m * k).reshape(n_exp, m, k).astype(np.float64)
    ref = np.empty((n_tok, n_used, m))
    for tt in range(n_tok):
        for s_ in range(n_used):
            ref[tt, s_] = wf[ids[tt, s_]] @ b[tt, s_ % b1].astype(np.float64)
    err = float(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert np.isfinite(got).all() and err < 5e-4, err
    vec = ops.mul_mat_id(t, dev(w), dev(b), dev(ids), m, n_exp).cpu().numpy()
    assert O.nmse(vec, got) <= 5e-4


@pytest.mark.parametrize("t", [B.Q4_K, B.Q6_K, B.Q8_0])
def test_mul_mat_id_skinny_with_planes_and_tables_from_the_caller(ops, t):
    """mi355_mul_mat_id_skinny_ex: the up / gate pair builds the routing tables in a buffer of the caller's, the down product on the same
    ids reuses them (MI355_IDS_TABLES_READY) and finds its activation planes already in the workspace (MI355_IDS_ACT_READY; written here by
    mi355_quantize_act_q8k / mi355_quantize_act, in the plugin by the GLU launch): every result has the bits of the plain call."""
    import llm_amd.kernels as K
    m, k, n_exp, n_used, n_tok = 256, 512, 8, 2, 13
    rng = np.random.default_rng(5)
    w0, w1 = (dev(B.random_blocks(t, n_exp * m, k, seed=s)) for s in (1, 2))
    wd = dev(B.random_blocks(t, n_exp * k, m, seed=3))           # "down": k rows of m values per expert
    x = dev(rng.standard_normal((n_tok, 1, k)).astype(np.float32))
    h = dev(rng.standard_normal((n_tok, n_used, m)).astype(np.float32))
    ids = dev(np.stack([rng.permutation(n_exp)[:n_used] for _ in range(n_tok)]).astype(np.int32))
    tb = K.lib.mi355_mul_mat_id_tables_bytes(n_tok, n_used, n_exp)
    tables = torch.zeros(tb, dtype=torch.uint8, device="cuda")
    need_up = K.lib.mi355_mul_mat_id_skinny_workspace_bytes(t, k, n_tok, 1, n_used, n_exp)
    need_dn = K.lib.mi355_mul_mat_id_skinny_workspace_bytes(t, m, n_tok, n_used, n_used, n_exp)
    ws = torch.zeros(max(need_up, need_dn), dtype=torch.uint8, device="cuda")
    if t == B.Q8_0:      # no pair form outside the K-quants
        up = ops.mul_mat_id_skinny_ex(t, w0, None, x, ids, m, n_exp, ws, tables, 0)
        assert torch.equal(up.view(torch.int32), ops.mul_mat_id_skinny(t, w0, x, ids, m, n_exp).view(torch.int32))
    else:
        up, gate = ops.mul_mat_id_skinny_ex(t, w0, w1, x, ids, m, n_exp, ws, tables, 0)
        want = ops.mul_mat_id_skinny_pair(t, w0, w1, x, ids, m, n_exp)
        assert torch.equal(up.view(torch.int32), want[0].view(torch.int32)) and torch.equal(gate.view(torch.int32), want[1].view(torch.int32))
    # the down product: planes of h written into the workspace by the caller, tables from the call above
    rows = n_tok * n_used
    up256 = lambda v: (v + 255) // 256 * 256
    if t == B.Q8_0:
        aq, ads = ops.quantize_act(h.view(rows, m))
        ws[:rows * m] = aq.view(torch.uint8).view(-1)
        ws[up256(rows * m):up256(rows * m) + ads.numel() * 4] = ads.view(-1).view(torch.uint8)
    else:
        q, d, bs = ops.quantize_act_q8k(h.view(rows, m))
        b0, b1 = up256(rows * m), up256(rows * (m // 256) * 4)
        ws[:rows * m] = q.view(torch.uint8).view(-1)
        ws[b0:b0 + d.numel() * 4] = d.view(-1).view(torch.uint8)
        ws[b0 + b1:b0 + b1 + bs.numel() * 2] = bs.view(-1).view(torch.uint8)
    ids.add_(0)         # (the ids are untouched; the tables would be stale otherwise)
    down = ops.mul_mat_id_skinny_ex(t, wd, None, h, ids, k, n_exp, ws, tables, 3)
    assert torch.equal(down.view(torch.int32), ops.mul_mat_id_skinny(t, wd, h, ids, k, n_exp).view(torch.int32))
    with pytest.raises(K.Mi355Error):
        ops.mul_mat_id_skinny_ex(t, wd, None, h, ids, k, n_exp, ws, None, 2)      # ready tables need a place


@pytest.mark.parametrize("t", [B.Q4_K, B.Q5_K, B.Q6_K])
@pytest.mark.parametrize("m,k,n_exp,n_used,n_tok", [(96, 256, 4, 2, 6), (130, 512, 8, 2, 8), (2080, 1024, 8, 2, 40), (14336, 4096, 8, 2, 16)])
def test_mul_mat_id_skinny_pair_has_the_bits_of_two_calls(ops, t, m, k, n_exp, n_used, n_tok):
    """ffn_up_exps and ffn_gate_exps of a MoE layer in one skinny launch (one quantization, one bucketing, one grid over the row tiles
    of both): each result is its own call's, bit for bit -- odd tile counts, an out-of-range id and the full Mixtral shape included."""
    from test_decode_layer_gpu import gpu_blocks
    rng = np.random.default_rng(m + k + n_tok)
    w0, w1 = gpu_blocks(t, n_exp * m, k, seed=70 + t), gpu_blocks(t, n_exp * m, k, seed=71 + t)
    b = dev(rng.standard_normal((n_tok, 1, k)).astype(np.float32))
    ids = np.stack([rng.permutation(n_exp)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    ids[n_tok // 2, 1] = n_exp + 1                                      # pair skipped: both results keep their NaN there
    idd = dev(ids)
    d0, d1 = ops.mul_mat_id_skinny_pair(t, w0, w1, b, idd, m, n_exp)
    s0 = ops.mul_mat_id_skinny(t, w0, b, idd, m, n_exp)
    s1 = ops.mul_mat_id_skinny(t, w1, b, idd, m, n_exp)
    assert torch.equal(d0.view(torch.int32), s0.view(torch.int32)) and torch.equal(d1.view(torch.int32), s1.view(torch.int32))
    ok = torch.from_numpy((ids >= 0) & (ids < n_exp)).cuda()
    assert torch.isfinite(d0[ok]).all() and torch.isfinite(d1[ok]).all() and torch.isnan(d0[~ok]).all() and torch.isnan(d1[~ok]).all()


@pytest.mark.parametrize("t", [B.Q4_K, B.Q6_K])
@pytest.mark.parametrize("m,k", [(14336, 4096), (4096, 14336)])
@pytest.mark.parametrize("n_tok", [8, 32, 64])
def test_mul_mat_id_skinny_mixtral_shapes(ops, t, m, k, n_tok):
    """BASELINE.json config 5 at FULL size on its batched path: Mixtral-8x7B's expert matrices (8 x 14336 x 4096 gate / up,
    8 x 4096 x 14336 down), 2 experts per token, T = 8 / 32 / 64 tokens through mi355_mul_mat_id_skinny (device-side bucketing,
    col_map, K split, the q6_K mmsk6 kernel in IDS mode).  Checked against exact dequant(as[ids]) . b in f64 on a random subset
    of rows of every expert (the oracle's dequantizer on those rows), and against the per-pair mat-vec path on all rows."""
    from test_decode_layer_gpu import gpu_blocks
    n_exp, n_used = 8, 2
    rng = np.random.default_rng(m + n_tok + t)
    w = gpu_blocks(t, n_exp * m, k, seed=40 + t)
    b = rng.standard_normal((n_tok, 1, k)).astype(np.float32)
    ids = np.stack([rng.permutation(n_exp)[:n_used] for _ in range(n_tok)]).astype(np.int32)
    ids[: n_tok // 2, 0] = 3; ids[: n_tok // 2, 1] = (4 + np.arange(n_tok // 2)) % n_exp       # expert 3 takes more than its share
    ids[: n_tok // 2, 1][ids[: n_tok // 2, 1] == 3] = 0
    got = ops.mul_mat_id_skinny(t, w, dev(b), dev(ids), m, n_exp).cpu().numpy()
    assert got.shape == (n_tok, n_used, m) and np.isfinite(got).all()
    rows = np.sort(rng.choice(m, size=48, replace=False))
    rs = B.row_size(t, k)
    num = den = 0.0
    for e in range(n_exp):
        sub = w.view(n_exp, m, rs)[e][torch.from_numpy(rows).cuda()].cpu().numpy()
        wf = O.dequantize_row(t, np.ascontiguousarray(sub), len(rows) * k).reshape(len(rows), k).astype(np.float64)
        for tt in range(n_tok):
            for s_ in range(n_used):
                if ids[tt, s_] == e:
                    ref = wf @ b[tt, 0].astype(np.float64)
                    num += float(((got[tt, s_, rows] - ref) ** 2).sum()); den += float((ref ** 2).sum())
    assert den > 0 and num / den < 5e-4, num / den
    vec = ops.mul_mat_id(t, w, dev(b), dev(ids), m, n_exp).cpu().numpy()
    assert O.nmse(vec, got) <= 5e-4


def test_errors_are_reported_not_launched(ops):
    import llm_amd.kernels as K
    w = torch.zeros(144 * 4 + 16, dtype=torch.uint8, device="cuda")
    aq, ads = ops.quantize_act(torch.zeros((1, 256), device="cuda"))
    with pytest.raises(K.Mi355Error) as e:      # q4_K blocks must be 16-byte aligned
        ops.mul_mat_vec_q(B.Q4_K, w[2:2 + 144 * 4], 4, 256, aq, ads, 1)
    assert e.value.code == -3
    with pytest.raises(K.Mi355Error) as e:      # ncols > 8 is not a mat-vec
        ops.mul_mat_vec_q(B.Q4_K, w[:144 * 4], 4, 256, aq, ads, 9)
    assert e.value.code == -2


def test_map_and_norm_vs_numpy(ops):
    """mi355_map_f32 (the UNARY ops, SQR .. LOG, CLAMP) and mi355_norm_f32 (NORM, L2_NORM) against float64 numpy on a strided view
    (every op also runs under the reference's test-backend-ops in tests/test_backend_plugin_gpu.py); refusals by return code."""
    import math
    import llm_amd.kernels as K
    s = torch.cuda.current_stream().cuda_stream
    def T(t):                                 # a strided view, ggml-style
        r = K.Tensor()
        r.data, r.type = t.data_ptr(), 0
        sh, st = list(t.shape)[::-1], list(t.stride())[::-1]
        for i in range(4):
            r.ne[i] = sh[i] if i < len(sh) else 1
            r.nb[i] = st[i] * 4 if i < len(sh) else r.nb[i - 1] * r.ne[i - 1]
        return r
    rng = np.random.default_rng(3)
    base = dev(rng.uniform(-4, 4, (3, 5, 40)).astype(np.float32))
    x = base[:, :, 3:36]                      # rows of 33 values, 160-byte pitch
    xn = x.cpu().numpy().astype(np.float64)
    erf = np.vectorize(math.erf)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    hs = lambda v: np.minimum(1.0, np.maximum(0.0, (v + 3.0) / 6.0))
    want = {0: np.abs(xn), 1: np.sign(xn), 2: -xn, 3: (xn > 0) * 1.0, 4: np.tanh(xn), 5: np.where(xn > 0, xn, np.expm1(xn)), 6: np.maximum(xn, 0), 7: sig(xn),
            8: 0.5 * xn * (1 + np.tanh(0.7978845608028654 * xn * (1 + 0.044715 * xn * xn))), 9: xn * sig(1.702 * xn), 10: xn * sig(xn), 11: xn * hs(xn),
            12: hs(xn), 13: np.exp(xn), 14: 0.5 * xn * (1 + erf(xn * 0.7071067811865476)), 15: xn * xn, 17: np.sin(xn), 18: np.cos(xn), 20: np.clip(xn, -1.5, 2.25)}
    for op, w in want.items():
        y = torch.full_like(base, float("nan"))[:, :, 3:36]
        K.check(K.lib.mi355_map_f32(op, T(x), T(y), -1.5, 2.25, s), "map")
        assert np.allclose(y.cpu().numpy(), w, rtol=2e-6, atol=2e-6), op
    xp = x.abs() + 0.1
    for op, w in ((16, np.sqrt), (19, np.log)):
        y = torch.empty_like(base)[:, :, 3:36]
        K.check(K.lib.mi355_map_f32(op, T(xp), T(y), 0.0, 0.0, s), "map")
        assert np.allclose(y.cpu().numpy(), w(xp.cpu().numpy().astype(np.float64)), rtol=2e-6, atol=2e-6), op
    for l2 in (0, 1):
        y = torch.empty_like(base)[:, :, 3:36]
        K.check(K.lib.mi355_norm_f32(T(x), T(y), 1e-5, l2, s), "norm")
        w = xn / np.maximum(np.sqrt((xn ** 2).sum(-1, keepdims=True)), 1e-5) if l2 else (xn - xn.mean(-1, keepdims=True)) / np.sqrt(xn.var(-1, keepdims=True) + 1e-5)
        assert np.allclose(y.cpu().numpy(), w, rtol=1e-5, atol=1e-6), l2
    y = torch.empty_like(base)[:, :, 3:36]
    assert K.lib.mi355_map_f32(21, T(x), T(y), 0.0, 0.0, s) == -6          # no such op
    assert K.lib.mi355_norm_f32(T(x), T(base), 1e-5, 0, s) == -2            # shapes differ


@pytest.mark.parametrize("k,n_exp,n_used,n_tok", [(4096, 8, 2, 1), (1024, 8, 2, 3), (2048, 64, 6, 2), (4096, 8, 2, 16)])
def test_moe_router_and_its_norm_form(ops, k, n_exp, n_used, n_tok):
    """mi355_moe_router_f32 against numpy (probabilities, the order of the top n_used, normalised weights), and
    mi355_moe_router_norm_f32 -- the ffn norm in front, in the same launch -- against mi355_rms_norm_mul_quant_f32 followed by the
    router: the normalised row, its q8_1 planes and every router tensor bit for bit."""
    import llm_amd.kernels as K
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(k + n_exp + n_tok)
    x = dev(rng.standard_normal((n_tok, k)).astype(np.float32) * 3)
    nw = dev(rng.uniform(0.5, 1.5, k).astype(np.float32))
    gate = dev(rng.standard_normal((n_exp, k)).astype(np.float32) * 0.05)
    def outs():
        f = lambda *sh: torch.full(sh, float("nan"), device="cuda")
        return [f(n_tok, n_exp), f(n_tok, n_exp), torch.full((n_tok, n_exp), -1, dtype=torch.int32, device="cuda"), f(n_tok, n_used), f(n_tok), f(n_tok, n_used)]
    xn = torch.empty_like(x)
    aq = torch.empty((n_tok, k), dtype=torch.int8, device="cuda"); ads = torch.empty((n_tok, k // 32, 2), device="cuda")
    if n_tok <= 8:
        K.check(K.lib.mi355_rms_norm_mul_quant_f32(K.Tensor.of(x), nw.data_ptr(), K.Tensor.of(xn), 1e-5, aq.data_ptr(), ads.data_ptr(), s), "norm")
    else:
        K.check(K.lib.mi355_rms_norm_mul_f32(K.Tensor.of(x), nw.data_ptr(), K.Tensor.of(xn), 1e-5, s), "norm")
        aq, ads = ops.quantize_act(xn)
    a = outs()
    K.check(K.lib.mi355_moe_router_f32(xn.data_ptr(), k * 4, gate.data_ptr(), k * 4, k, n_exp, n_used, n_tok, *[t.data_ptr() for t in a], s), "router")
    # vs numpy
    xd = xn.cpu().numpy().astype(np.float64)
    lg = xd @ gate.cpu().numpy().astype(np.float64).T
    pr = np.exp(lg - lg.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
    assert np.allclose(a[1].cpu().numpy(), pr, rtol=1e-4, atol=1e-6)
    top = np.argsort(-pr, axis=1, kind="stable")[:, :n_used]
    assert np.array_equal(a[2].cpu().numpy()[:, :n_used], top)
    wsel = np.take_along_axis(pr, top, 1)
    assert np.allclose(a[5].cpu().numpy(), wsel / wsel.sum(1, keepdims=True), rtol=1e-4)
    # the norm form
    b = outs()
    xn2 = torch.full_like(x, float("nan")); aq2 = torch.zeros_like(aq); ads2 = torch.zeros_like(ads)
    K.check(K.lib.mi355_moe_router_norm_f32(x.data_ptr(), k * 4, nw.data_ptr(), 1e-5, xn2.data_ptr(), aq2.data_ptr(), ads2.data_ptr(), gate.data_ptr(), k * 4,
                                            k, n_exp, n_used, n_tok, *[t.data_ptr() for t in b], s), "router_norm")
    assert torch.equal(xn.view(torch.int32), xn2.view(torch.int32)) and torch.equal(aq, aq2) and torch.equal(ads.view(torch.int32), ads2.view(torch.int32))
    for u, v in zip(a, b):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))
    assert K.lib.mi355_moe_router_norm_f32(x.data_ptr(), 8192 * 4, nw.data_ptr(), 1e-5, xn2.data_ptr(), None, None, gate.data_ptr(), 8192 * 4,
                                           8192, n_exp, n_used, 1, *[t.data_ptr() for t in b], s) == -6



@pytest.mark.parametrize("m,n_tok,in_place", [(4096, 1, True), (1024, 12, True), (4100, 7, False), (14336, 512, True)])
def test_moe_combine_vs_numpy(ops, m, n_tok, in_place):
    """mi355_moe_combine2_f32: weighted = e * w and sum = weighted[:, 0] + weighted[:, 1], each value one f32 multiply / add as ggml-cpu's
    MUL and ADD do them -> bit-exact against numpy; the weights are read through strides (llama.cpp hands a [1, 2, T] view)."""
    import llm_amd.kernels as K
    rng = np.random.default_rng(m + n_tok)
    e = rng.standard_normal((n_tok, 2, m)).astype(np.float32)
    w = rng.uniform(0.1, 0.9, (n_tok, 3)).astype(np.float32)        # row stride 3 floats: slot s at [t, s]
    e_d, w_d = dev(e), dev(w)
    weighted = e_d if in_place else torch.full((n_tok, 2, m), float("nan"), device="cuda")
    total = torch.full((n_tok, m), float("nan"), device="cuda")
    rc = K.lib.mi355_moe_combine2_f32(e_d.data_ptr(), w_d.data_ptr(), 12, 4, weighted.data_ptr(), total.data_ptr(), m, n_tok,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    want = e * w[:, :2, None]
    assert np.array_equal(weighted.cpu().numpy(), want)
    assert np.array_equal(total.cpu().numpy(), want[:, 0] + want[:, 1])
    assert K.lib.mi355_moe_combine2_f32(e_d.data_ptr(), w_d.data_ptr(), 12, 4, weighted.data_ptr(), total.data_ptr(), 4098, 1, 0) == -6      # MI355_E_UNSUPPORTED
