"""mi355_decode_layer (the persistent decode-layer launch, llm_amd/csrc/kernels/mmvq_layer.hip) against the four launches it
fuses -- mi355_mul_mat_vec_q (+ residual), mi355_glu_matvec (+ norm), mi355_mul_mat_vec_q (+ residual), mi355_qkv_rope_matvec
(+ norm) or the output mat-vec (+ norm) -- which are themselves pinned to the oracle / goldens in test_kernels_gpu.py.  Bit for bit,
every tensor the launch writes, at Llama-3-8B / 70B shapes and at shapes that exercise uneven dealing and a smaller grid."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_amd.blocks as B  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import llm_amd.ops as o
    return o


def gpu_blocks(t, m, k, seed):
    """random valid blocks of type t made on the GPU (random quant bytes, finite fp16 scales), as bench.py does"""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    nb = k // B.BLCK_SIZE[t]
    w = torch.randint(0, 256, (m * nb, B.TYPE_SIZE[t]), dtype=torch.uint8, device="cuda", generator=gen)
    scale = {B.Q4_K: 0.003, B.Q5_K: 0.002, B.Q6_K: 0.0005}[t]
    for off in B._HALF_FIELDS[t]:
        d = ((torch.rand(m * nb, device="cuda", generator=gen) * 0.75 + 0.25) * scale).to(torch.float16)
        w[:, off:off + 2] = d.view(torch.uint8).view(m * nb, 2)
    return w.view(m, B.row_size(t, k))


def bits(x):
    return x.contiguous().view(-1).view(torch.int16 if x.dtype == torch.float16 else torch.int32)


CASES = [
    # n_embd, k_o, n_ff, (type_o, type_gu, type_d), tail, tail types, heads (n_head, n_head_kv, head_dim)
    (4096, 4096, 14336, (B.Q4_K, B.Q4_K, B.Q4_K), "qkv", (B.Q4_K, B.Q4_K), (32, 8, 128)),
    (4096, 4096, 14336, (B.Q4_K, B.Q4_K, B.Q6_K), "qkv", (B.Q4_K, B.Q6_K), (32, 8, 128)),
    (4096, 4096, 14336, (B.Q4_K, B.Q4_K, B.Q6_K), "rows", (B.Q6_K, 128256), None),
    (4096, 4096, 14336, (B.Q5_K, B.Q5_K, B.Q5_K), None, None, None),
    (8192, 8192, 28672, (B.Q4_K, B.Q4_K, B.Q6_K), "qkv", (B.Q4_K, B.Q6_K), (64, 8, 128)),
    (2048, 2048, 5120, (B.Q5_K, B.Q4_K, B.Q6_K), "qkv", (B.Q5_K, B.Q5_K), (16, 16, 128)),
    (3072, 4096, 8192, (B.Q6_K, B.Q6_K, B.Q4_K), "rows", (B.Q4_K, 1000), None),
    (5120, 5120, 13312, (B.Q4_K, B.Q4_K, B.Q4_K), "qkv", (B.Q4_K, B.Q4_K), (40, 8, 128)),      # 20 / 52 rows per workgroup: uneven dealing
]


@pytest.mark.parametrize("n_embd,k_o,n_ff,types,tail,tt,heads", CASES)
@pytest.mark.parametrize("attn_quantized", [False, True])
def test_decode_layer_has_the_bits_of_the_four_launches(ops, n_embd, k_o, n_ff, types, tail, tt, heads, attn_quantized):
    import llm_amd.kernels as K
    if attn_quantized and n_embd not in (4096, 5120):
        pytest.skip("the quantized-input form is covered at two shapes")
    wo = gpu_blocks(types[0], n_embd, k_o, 1)
    wg, wu = gpu_blocks(types[1], n_ff, n_embd, 2), gpu_blocks(types[1], n_ff, n_embd, 3)
    wd = gpu_blocks(types[2], n_embd, n_ff, 4)
    eps = 1e-5
    pos = torch.tensor([41], dtype=torch.int32, device="cuda")
    if tail == "qkv":
        nh, nkv, hd = heads
        wq, wk, wv = gpu_blocks(tt[0], nh * hd, n_embd, 5), gpu_blocks(tt[0], nkv * hd, n_embd, 6), gpu_blocks(tt[1], nkv * hd, n_embd, 7)
    elif tail == "rows":
        wout = gpu_blocks(tt[0], tt[1], n_embd, 8)
    for rep in range(3):         # the same buffers again with other inputs: nothing may come from an earlier call
        g = torch.Generator(device="cuda")
        g.manual_seed(100 + rep)
        attn = torch.randn(k_o, device="cuda", generator=g)
        x = torch.randn(n_embd, device="cuda", generator=g)
        nw1 = torch.rand(n_embd, device="cuda", generator=g) + 0.5
        nw3 = torch.rand(n_embd, device="cuda", generator=g) + 0.5
        # ---- the separate launches
        aq, ads = ops.quantize_act(attn.view(1, -1))
        ffn_inp0 = torch.empty(1, n_embd, device="cuda")
        attn_out0 = ops.mul_mat_vec_q(types[0], wo, n_embd, k_o, aq, ads, 1, residual=x.view(1, -1), sum_out=ffn_inp0)
        n1_0 = torch.empty(n_embd, device="cuda")
        h0, g0, u0, hq, hds = ops.glu_matvec(types[1], wg, wu, n_ff, n_embd, None, None, glu_op=2, quant=True, keep=True,
                                             norm=(ffn_inp0, nw1, eps, n1_0))
        l_out0 = torch.empty(1, n_embd, device="cuda")
        down0 = ops.mul_mat_vec_q(types[2], wd, n_embd, n_ff, hq.view(1, -1), hds.view(1, -1, 2), 1, residual=ffn_inp0, sum_out=l_out0)
        n3_0 = torch.full((n_embd,), float("nan"), device="cuda")
        t_arg = None
        if tail == "qkv":
            q0, k0, v0 = ops.qkv_rope_matvec(tt[0], tt[1], wq, wk, wv, nh * hd, nkv * hd, nkv * hd, n_embd, None, None, pos, hd, hd, 0,
                                             freq_base=500000.0, norm=(l_out0.view(-1), nw3, eps, n3_0))
            n3_1 = torch.full((n_embd,), float("nan"), device="cuda")
            q1 = torch.full((nh * hd,), float("nan"), device="cuda")
            k1 = torch.zeros(nkv * hd, dtype=torch.float16, device="cuda")
            v1 = torch.zeros(nkv * hd, dtype=torch.float16, device="cuda")
            a = K.QkvRopeArgs()
            a.type_qk, a.type_v, a.wq, a.wk, a.wv = tt[0], tt[1], wq.data_ptr(), wk.data_ptr(), wv.data_ptr()
            a.wq_row_stride = a.wk_row_stride = B.row_size(tt[0], n_embd)
            a.wv_row_stride = B.row_size(tt[1], n_embd)
            a.m_q, a.m_k, a.m_v, a.k = nh * hd, nkv * hd, nkv * hd, n_embd
            a.q_out, a.k_out, a.v_out = q1.data_ptr(), k1.data_ptr(), v1.data_ptr()
            a.k_out_type = a.v_out_type = B.F16
            a.pos = pos.data_ptr()
            a.head_dim, a.n_dims, a.mode, a.n_ctx_orig = hd, hd, 0, 8192
            a.freq_base, a.freq_scale, a.ext_factor, a.attn_factor, a.beta_fast, a.beta_slow = 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0
            a.norm_w, a.norm_out, a.norm_eps = nw3.data_ptr(), n3_1.data_ptr(), eps
            t_arg = ("qkv", a)
        elif tail == "rows":
            m_out = tt[1]
            a0 = K.MmvqArgs()
            logits0 = torch.full((m_out,), float("nan"), device="cuda")
            a0.type, a0.W, a0.w_row_stride = tt[0], wout.data_ptr(), B.row_size(tt[0], n_embd)
            a0.dst, a0.dst_col_stride, a0.m, a0.k, a0.ncols, a0.nchannels, a0.ch_ratio = logits0.data_ptr(), m_out, m_out, n_embd, 1, 1, 1
            a0.src_f32, a0.norm_w, a0.norm_out, a0.norm_eps = l_out0.data_ptr(), nw3.data_ptr(), n3_0.data_ptr(), eps
            K.check(K.lib.mi355_mul_mat_vec_q(a0, torch.cuda.current_stream().cuda_stream), "mi355_mul_mat_vec_q")
            n3_1 = torch.full((n_embd,), float("nan"), device="cuda")
            logits1 = torch.full((m_out,), float("nan"), device="cuda")
            a = K.MmvqArgs()
            a.type, a.W, a.w_row_stride = tt[0], wout.data_ptr(), B.row_size(tt[0], n_embd)
            a.dst, a.dst_col_stride, a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = logits1.data_ptr(), m_out, m_out, n_embd, 1, 1, 1
            a.norm_w, a.norm_out, a.norm_eps = nw3.data_ptr(), n3_1.data_ptr(), eps
            t_arg = ("rows", a)
        # ---- the one launch
        o = ops.decode_layer(types, (wo, wg, wu, wd), n_embd, k_o, n_ff, (aq, ads) if attn_quantized else attn, x, nw1, eps=eps, glu_op=2,
                             tail=t_arg, keep=True)
        torch.cuda.synchronize()
        err = int(o["sync"].view(torch.int32)[-32].item())
        assert err == 0, f"a wait inside the launch ran out: code {err:#x}"
        for name, ref in (("attn_out", attn_out0), ("ffn_inp", ffn_inp0), ("ffn_norm", n1_0), ("gate", g0), ("up", u0), ("hidden", h0),
                          ("down", down0), ("l_out", l_out0)):
            assert torch.isfinite(o[name]).all(), name
            assert torch.equal(bits(o[name]), bits(ref)), f"{name} differs (rep {rep})"
        if tail == "qkv":
            assert torch.equal(bits(n3_1), bits(n3_0)), "attn_norm differs"
            assert torch.isfinite(q1).all() and float(q1.abs().max()) > 0
            assert torch.equal(bits(q1), bits(q0)) and torch.equal(bits(k1), bits(k0)) and torch.equal(bits(v1), bits(v0)), f"q / k / v differ (rep {rep})"
        elif tail == "rows":
            assert torch.equal(bits(n3_1), bits(n3_0)), "output norm differs"
            assert torch.isfinite(logits1).all() and torch.equal(bits(logits1), bits(logits0)), f"logits differ (rep {rep})"


@pytest.mark.parametrize("n_embd,n_ff,heads", [(4096, 14336, (32, 8, 128)), (8192, 28672, (64, 8, 128)), (2048, 5120, (16, 4, 128))])
def test_decode_layer_on_buffers_placed_like_ggml_alloc(ops, n_embd, n_ff, heads):
    """ggml-alloc puts the tensors of later nodes on the bytes of dead earlier ones.  In a llama decode graph (GGML_MI355_DUMP_GRAPH):
    ffn_inp in place on attn_out; ffn_out (down) and l_out on ffn_norm's bytes; the next layer's rotated Qcur on ffn_gate's bytes; the
    next layer's attn_norm on the bytes of this layer's residual input.  Inside ONE launch those are stores of different phases,
    from different XCDs, to the same addresses: whatever survives must still carry the bits of the separate launches."""
    import llm_amd.kernels as K
    types, tt = (B.Q4_K, B.Q4_K, B.Q6_K), (B.Q4_K, B.Q6_K)
    nh, nkv, hd = heads
    wo = gpu_blocks(types[0], n_embd, n_embd, 1)
    wg, wu = gpu_blocks(types[1], n_ff, n_embd, 2), gpu_blocks(types[1], n_ff, n_embd, 3)
    wd = gpu_blocks(types[2], n_embd, n_ff, 4)
    wq, wk, wv = gpu_blocks(tt[0], nh * hd, n_embd, 5), gpu_blocks(tt[0], nkv * hd, n_embd, 6), gpu_blocks(tt[1], nkv * hd, n_embd, 7)
    eps = 1e-5
    pos = torch.tensor([7], dtype=torch.int32, device="cuda")
    buf_a = torch.empty(n_embd, device="cuda")       # ffn_norm, then ffn_out / l_out
    buf_b = torch.empty(n_embd, device="cuda")       # attn_out / ffn_inp
    buf_g = torch.empty(n_ff, device="cuda")         # ffn_gate, then the next layer's Qcur
    for rep in range(4):
        g = torch.Generator(device="cuda")
        g.manual_seed(500 + rep)
        attn = torch.randn(n_embd, device="cuda", generator=g)
        x = torch.randn(n_embd, device="cuda", generator=g)          # the residual input, then the next layer's attn_norm
        nw1 = torch.rand(n_embd, device="cuda", generator=g) + 0.5
        nw3 = torch.rand(n_embd, device="cuda", generator=g) + 0.5
        aq, ads = ops.quantize_act(attn.view(1, -1))
        ffn_inp0 = torch.empty(1, n_embd, device="cuda")
        ops.mul_mat_vec_q(types[0], wo, n_embd, n_embd, aq, ads, 1, residual=x.view(1, -1), sum_out=ffn_inp0)
        n1_0 = torch.empty(n_embd, device="cuda")
        h0, g0, u0, hq, hds = ops.glu_matvec(types[1], wg, wu, n_ff, n_embd, None, None, glu_op=2, quant=True, keep=True, norm=(ffn_inp0, nw1, eps, n1_0))
        l_out0 = torch.empty(1, n_embd, device="cuda")
        ops.mul_mat_vec_q(types[2], wd, n_embd, n_ff, hq.view(1, -1), hds.view(1, -1, 2), 1, residual=ffn_inp0, sum_out=l_out0)
        n3_0 = torch.full((n_embd,), float("nan"), device="cuda")
        q0, k0, v0 = ops.qkv_rope_matvec(tt[0], tt[1], wq, wk, wv, nh * hd, nkv * hd, nkv * hd, n_embd, None, None, pos, hd, hd, 0,
                                         freq_base=500000.0, norm=(l_out0.view(-1), nw3, eps, n3_0))
        k1 = torch.zeros(nkv * hd, dtype=torch.float16, device="cuda")
        v1 = torch.zeros(nkv * hd, dtype=torch.float16, device="cuda")
        a = K.QkvRopeArgs()
        a.type_qk, a.type_v, a.wq, a.wk, a.wv = tt[0], tt[1], wq.data_ptr(), wk.data_ptr(), wv.data_ptr()
        a.wq_row_stride = a.wk_row_stride = B.row_size(tt[0], n_embd)
        a.wv_row_stride = B.row_size(tt[1], n_embd)
        a.m_q, a.m_k, a.m_v, a.k = nh * hd, nkv * hd, nkv * hd, n_embd
        a.q_out, a.k_out, a.v_out = buf_g.data_ptr(), k1.data_ptr(), v1.data_ptr()
        a.k_out_type = a.v_out_type = B.F16
        a.pos = pos.data_ptr()
        a.head_dim, a.n_dims, a.mode, a.n_ctx_orig = hd, hd, 0, 8192
        a.freq_base, a.freq_scale, a.ext_factor, a.attn_factor, a.beta_fast, a.beta_slow = 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0
        a.norm_w, a.norm_out, a.norm_eps = nw3.data_ptr(), x.data_ptr(), eps
        for t in (buf_a, buf_b, buf_g):
            t.fill_(float("nan"))
        o = ops.decode_layer(types, (wo, wg, wu, wd), n_embd, n_embd, n_ff, (aq, ads), x, nw1, eps=eps, glu_op=2, tail=("qkv", a), keep=True,
                             bufs={"ffn_norm": buf_a, "down": buf_a, "l_out": buf_a, "attn_out": buf_b, "ffn_inp": buf_b, "gate": buf_g})
        torch.cuda.synchronize()
        err = int(o["sync"].view(torch.int32)[-32].item())
        assert err == 0, f"a wait inside the launch ran out: code {err:#x}"
        for name, got, ref in (("ffn_inp", buf_b, ffn_inp0), ("up", o["up"], u0), ("hidden", o["hidden"], h0), ("l_out", buf_a, l_out0),
                               ("attn_norm of the next layer", x, n3_0), ("q", buf_g[:nh * hd], q0), ("k", k1, k0), ("v", v1, v0)):
            assert torch.equal(bits(got), bits(ref)), f"{name} differs (rep {rep})"
        assert torch.equal(bits(buf_g[nh * hd:]), bits(g0.view(-1)[nh * hd:])), "the rest of ffn_gate differs"


def test_decode_layer_argument_errors(ops):
    import llm_amd.kernels as K
    a = K.DecodeLayerArgs()
    sync = torch.zeros(K.lib.mi355_decode_layer_sync_bytes(), dtype=torch.uint8, device="cuda")
    assert K.lib.mi355_decode_layer(a, sync.data_ptr(), None) == -4                     # NULL pointers
    assert K.lib.mi355_decode_layer(None, sync.data_ptr(), None) == -4
    w = gpu_blocks(B.Q4_K, 4096, 4096, 1)
    x = torch.zeros(14336, device="cuda")
    a.type_o = a.type_gu = a.type_d = B.Q4_K
    a.glu_op, a.n_embd, a.k_o, a.n_ff = 2, 4096, 4096, 14336
    for n in ("wo", "w_gate", "w_up", "w_down", "residual", "ffn_inp", "ffn_norm_w", "ffn_norm_out", "hidden", "l_out", "attn_f32"):
        setattr(a, n, x.data_ptr() if not n.startswith("w") else w.data_ptr())
    a.wo_row_stride = a.gu_row_stride = B.row_size(B.Q4_K, 4096)
    a.down_row_stride = B.row_size(B.Q4_K, 14336)
    a.type_d = B.Q8_0
    assert K.lib.mi355_decode_layer(a, sync.data_ptr(), None) == -6                     # not a K-quant
    a.type_d = B.Q4_K
    a.n_ff = 14336 + 256
    assert K.lib.mi355_decode_layer(a, sync.data_ptr(), None) in (-6, -2)               # rows do not deal evenly / stride too short
    a.n_ff = 14336
    a.tail = 7
    assert K.lib.mi355_decode_layer(a, sync.data_ptr(), None) == -2
    a.tail = 0
    assert K.lib.mi355_decode_layer(a, sync.data_ptr() + 4, None) == -3                 # sync block alignment
