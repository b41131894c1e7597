#!/usr/bin/env python3
"""Fused gate/up/GLU mat-vec vs grouped mat-vec + glu_quant, per-launch time from the library's timing mode over a
rotating pool of weight copies (cold: the pool is larger than the Infinity Cache)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

def timed(fn, tags, reps=40):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    for i in range(reps): fn(i)
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    return {tag: K.timing_read(tag)[1] / reps for tag in tags}

s = torch.cuda.current_stream().cuda_stream
T = K.Tensor.of
for t, m, k in ((B.Q4_K, 14336, 4096), (B.Q4_K, 28672, 8192), (B.Q6_K, 14336, 4096)):
    name = B.TYPE_NAMES[t]
    rs = B.row_size(t, k)
    ncopy = max(2, int(1.2e9 // (2 * m * rs)))
    pool = [(torch.randint(0, 255, (m * rs,), dtype=torch.uint8, device="cuda"), torch.randint(0, 255, (m * rs,), dtype=torch.uint8, device="cuda")) for _ in range(ncopy)]
    x = torch.randn(1, k, device="cuda")
    aq, ads = ops.quantize_act(x)
    d0 = torch.empty(1, m, device="cuda"); q0 = torch.empty((1, m), dtype=torch.int8, device="cuda"); s0 = torch.empty((1, m // 32, 2), device="cuda")
    def sep(i):
        wg, wu = pool[i % ncopy]
        g, u = ops.mul_mat_vec_q_group(t, [wg, wu], [m, m], k, aq, ads, 1)
        K.check(K.lib.mi355_glu_quant_f32(2, T(g.view(1, m)), T(u.view(1, m)), T(d0), 0, q0.data_ptr(), s0.data_ptr(), s), "glu")
    def fused(i):
        wg, wu = pool[i % ncopy]
        ops.glu_matvec(t, wg, wu, m, k, aq, ads)
    a = timed(sep, ["mmvq_%s_n1" % name, "glu_quant"])
    b = timed(fused, ["mmvq_glu"])
    wb = 2 * m * rs
    print(f"{name} 2x{m}x{k}: grouped {a['mmvq_%s_n1' % name]:.2f} + glu {a['glu_quant']:.2f} us   fused {b['mmvq_glu']:.2f} us ({wb / b['mmvq_glu'] / 1e3:.0f} GB/s)", flush=True)
    del pool

# expert-routed form (Mixtral-8x7B decode: 8 experts, 2 used): two MUL_MAT_ID mat-vecs + glu_quant against one launch
t, m, k, n_exp = B.Q4_K, 14336, 4096, 8
rs = B.row_size(t, k)
pool = [(torch.randint(0, 255, (n_exp * m * rs,), dtype=torch.uint8, device="cuda"), torch.randint(0, 255, (n_exp * m * rs,), dtype=torch.uint8, device="cuda")) for _ in range(2)]
x = torch.randn(1, 1, k, device="cuda")
aq, ads = ops.quantize_act(x.view(1, k))
d0 = torch.empty(2, m, device="cuda"); q0 = torch.empty((2, m), dtype=torch.int8, device="cuda"); s0 = torch.empty((2, m // 32, 2), device="cuda")
def sep(i):
    wg, wu = pool[i % 2]
    ids = torch.tensor([[(i * 3) % 8, (i * 3 + 5) % 8]], dtype=torch.int32, device="cuda")
    g = ops.mul_mat_id(t, wg, x, ids, m, n_exp)
    u = ops.mul_mat_id(t, wu, x, ids, m, n_exp)
    K.check(K.lib.mi355_glu_quant_f32(2, T(g.view(2, m)), T(u.view(2, m)), T(d0), 0, q0.data_ptr(), s0.data_ptr(), s), "glu")
def fused(i):
    wg, wu = pool[i % 2]
    ids = torch.tensor([(i * 3) % 8, (i * 3 + 5) % 8], dtype=torch.int32, device="cuda")
    ops.glu_matvec(t, wg, wu, m, k, aq, ads, ids=ids, n_expert=n_exp)
K.lib.mi355_timing_reset()
a = timed(sep, [])
rep = {tg: (n, us) for tg, (n, us) in K.timing_report().items() if n}
print("MoE separate:", {tg: round(us / n, 2) for tg, (n, us) in rep.items()}, flush=True)
K.lib.mi355_timing_reset()
b = timed(fused, ["mmvq_glu"])
print(f"MoE fused: {b['mmvq_glu']:.2f} us", flush=True)
