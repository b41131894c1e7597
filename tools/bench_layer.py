#!/usr/bin/env python3
"""Per-layer time of a decode token's mat-vecs: the persistent launch (mi355_decode_layer) against the four launches it
fuses, over a rotating set of layers (cold weights), replayed from a HIP graph.  usage: bench_layer.py [n_layers] [8b|70b]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import llm_amd.blocks as B  # noqa: E402
import llm_amd.kernels as K  # noqa: E402
import llm_amd.ops as ops  # noqa: E402
from test_decode_layer_gpu import gpu_blocks  # noqa: E402

NL = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = sys.argv[2] if len(sys.argv) > 2 else "8b"
n_embd, n_ff, nh, nkv, hd = (4096, 14336, 32, 8, 128) if model == "8b" else (8192, 28672, 64, 8, 128)
eps = 1e-5
pos = torch.tensor([41], dtype=torch.int32, device="cuda")
g = torch.Generator(device="cuda")
g.manual_seed(1)
attn = torch.randn(n_embd, device="cuda", generator=g)
x = torch.randn(n_embd, device="cuda", generator=g)
nw = torch.rand(n_embd, device="cuda", generator=g) + 0.5
aq, ads = ops.quantize_act(attn.view(1, -1))
layers = []
nbytes = 0
for l in range(NL):
    td, tv = (B.Q6_K, B.Q6_K) if l % 2 == 0 else (B.Q4_K, B.Q4_K)
    w = dict(wo=gpu_blocks(B.Q4_K, n_embd, n_embd, 10 * l), wg=gpu_blocks(B.Q4_K, n_ff, n_embd, 10 * l + 1), wu=gpu_blocks(B.Q4_K, n_ff, n_embd, 10 * l + 2),
             wd=gpu_blocks(td, n_embd, n_ff, 10 * l + 3), wq=gpu_blocks(B.Q4_K, nh * hd, n_embd, 10 * l + 4), wk=gpu_blocks(B.Q4_K, nkv * hd, n_embd, 10 * l + 5),
             wv=gpu_blocks(tv, nkv * hd, n_embd, 10 * l + 6), td=td, tv=tv)
    nbytes += sum(v.numel() for v in w.values() if torch.is_tensor(v))
    layers.append(w)
print(f"{NL} layers, {nbytes / 1e6:.1f} MB of weights ({nbytes / NL / 1e6:.1f} MB per layer)", flush=True)
f32 = lambda n: torch.empty(n, device="cuda")
buf = dict(ffn_inp=f32(n_embd), n1=f32(n_embd), hidden=f32(n_ff), l_out=f32(n_embd), n3=f32(n_embd), q=f32(nh * hd), attn_out=f32(n_embd), down=f32(n_embd),
           k=torch.empty(nkv * hd, dtype=torch.float16, device="cuda"), v=torch.empty(nkv * hd, dtype=torch.float16, device="cuda"),
           hq=torch.empty(n_ff, dtype=torch.int8, device="cuda"), hds=torch.empty((n_ff // 32, 2), device="cuda"))
nb = K.lib.mi355_decode_layer_sync_bytes()
sync = torch.zeros(NL * nb, dtype=torch.uint8, device="cuda")


def qkv_args(w, norm_x):
    a = K.QkvRopeArgs()
    a.type_qk, a.type_v, a.wq, a.wk, a.wv = B.Q4_K, w["tv"], w["wq"].data_ptr(), w["wk"].data_ptr(), w["wv"].data_ptr()
    a.wq_row_stride = a.wk_row_stride = B.row_size(B.Q4_K, n_embd)
    a.wv_row_stride = B.row_size(w["tv"], n_embd)
    a.m_q, a.m_k, a.m_v, a.k = nh * hd, nkv * hd, nkv * hd, n_embd
    a.q_out, a.k_out, a.v_out = buf["q"].data_ptr(), buf["k"].data_ptr(), buf["v"].data_ptr()
    a.k_out_type = a.v_out_type = B.F16
    a.pos = pos.data_ptr()
    a.head_dim, a.n_dims, a.mode, a.n_ctx_orig = hd, hd, 0, 8192
    a.freq_base, a.freq_scale, a.ext_factor, a.attn_factor, a.beta_fast, a.beta_slow = 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0
    a.norm_x, a.norm_w, a.norm_out, a.norm_eps = norm_x, nw.data_ptr(), buf["n3"].data_ptr(), eps
    return a


def mm_args(t, w, m, k, dst, res, sum_, aq_, ads_):
    a = K.MmvqArgs()
    a.type, a.W, a.w_row_stride = t, w.data_ptr(), B.row_size(t, k)
    a.act_q, a.act_ds = aq_.data_ptr(), ads_.data_ptr()
    a.dst, a.dst_col_stride, a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = dst.data_ptr(), m, m, k, 1, 1, 1
    a.residual, a.sum = res.data_ptr(), sum_.data_ptr()
    return a


sep, fused = [], []
for w in layers:
    ga = K.GluMatvecArgs()
    ga.type, ga.glu_op, ga.w_gate, ga.w_up = B.Q4_K, 2, w["wg"].data_ptr(), w["wu"].data_ptr()
    ga.gate_row_stride = ga.up_row_stride = B.row_size(B.Q4_K, n_embd)
    ga.m, ga.k, ga.dst, ga.out_q, ga.out_ds = n_ff, n_embd, buf["hidden"].data_ptr(), buf["hq"].data_ptr(), buf["hds"].data_ptr()
    ga.norm_x, ga.norm_w, ga.norm_out, ga.norm_eps = buf["ffn_inp"].data_ptr(), nw.data_ptr(), buf["n1"].data_ptr(), eps
    sep.append((mm_args(B.Q4_K, w["wo"], n_embd, n_embd, buf["attn_out"], x, buf["ffn_inp"], aq, ads), ga,
                mm_args(w["td"], w["wd"], n_embd, n_ff, buf["down"], buf["ffn_inp"], buf["l_out"], buf["hq"], buf["hds"]),
                qkv_args(w, buf["l_out"].data_ptr())))
    a = K.DecodeLayerArgs()
    a.type_o, a.type_gu, a.type_d, a.glu_op, a.tail = B.Q4_K, B.Q4_K, w["td"], 2, K.TAIL_QKV
    a.n_embd, a.k_o, a.n_ff = n_embd, n_embd, n_ff
    a.wo, a.wo_row_stride = w["wo"].data_ptr(), B.row_size(B.Q4_K, n_embd)
    a.attn_q, a.attn_ds = aq.data_ptr(), ads.data_ptr()
    a.residual, a.ffn_inp, a.attn_out = x.data_ptr(), buf["ffn_inp"].data_ptr(), buf["attn_out"].data_ptr()
    a.ffn_norm_w, a.ffn_norm_out, a.norm_eps = nw.data_ptr(), buf["n1"].data_ptr(), eps
    a.w_gate, a.w_up, a.gu_row_stride = w["wg"].data_ptr(), w["wu"].data_ptr(), B.row_size(B.Q4_K, n_embd)
    a.hidden, a.w_down, a.down_row_stride = buf["hidden"].data_ptr(), w["wd"].data_ptr(), B.row_size(w["td"], n_ff)
    a.l_out, a.down_out = buf["l_out"].data_ptr(), buf["down"].data_ptr()
    a.qkv = qkv_args(w, buf["l_out"].data_ptr())
    fused.append(a)


def run_sep():
    s = torch.cuda.current_stream().cuda_stream
    for wo_a, ga, d_a, q_a in sep:
        K.check(K.lib.mi355_mul_mat_vec_q(wo_a, s), "wo")
        K.check(K.lib.mi355_glu_matvec(ga, s), "glu")
        K.check(K.lib.mi355_mul_mat_vec_q(d_a, s), "down")
        K.check(K.lib.mi355_qkv_rope_matvec(q_a, s), "qkv")


def run_fused():
    s = torch.cuda.current_stream().cuda_stream
    sync.zero_()
    for i, a in enumerate(fused):
        K.check(K.lib.mi355_decode_layer(a, sync.data_ptr() + i * nb, s), "decode_layer")


def timeit(fn, name):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps / NL
    print(f"{name}: {us:.2f} us per layer, {nbytes / NL / us * 1e-3:.0f} GB/s", flush=True)
    return us


timeit(run_sep, "four launches")
timeit(run_fused, "persistent   ")
print("error words:", [int(sync.view(torch.int32)[(i + 1) * (nb // 4) - 32].item()) for i in range(NL)])
K.lib.mi355_timing_reset()
K.lib.mi355_timing_enable(1)
for _ in range(3):
    run_sep()
    run_fused()
torch.cuda.synchronize()
K.lib.mi355_timing_enable(0)
for t_, (n, us) in K.timing_report().items():
    if n:
        print(f"  {t_}: {n} launches, {us / n:.2f} us")
