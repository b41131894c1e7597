"""mi355_glu_mul_mat_dq against the three launches it replaces (ffn_gate GEMM, ffn_up GEMM, GLU + bf16 copy), Llama-8B prefill shape"""
import sys, torch, numpy as np, ctypes as C
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import llm_amd.kernels as K
from llm_amd import blocks as B, ops
t, m, k, n = B.Q4_K, 14336, 4096, 512
rng = np.random.default_rng(1)
wa = torch.from_numpy(B.random_blocks(t, m, k, seed=3)).cuda(); wb = torch.from_numpy(B.random_blocks(t, m, k, seed=4)).cuda()
x = ops.act_bf16(torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32)).cuda())
s = torch.cuda.current_stream().cuda_stream
rs = B.row_size(t, k)
g = torch.empty((n, m), device='cuda'); u = torch.empty_like(g); h = torch.empty_like(g); hb = torch.empty((n, m), dtype=torch.int16, device='cuda')
def three():
    K.mul_mat_dq(t, wa.data_ptr(), rs, x.data_ptr(), k, g.data_ptr(), m, m, k, n, s)
    K.mul_mat_dq(t, wb.data_ptr(), rs, x.data_ptr(), k, u.data_ptr(), m, m, k, n, s)
    K.lib.mi355_glu_bf16_f32(2, C.byref(K.Tensor.of(g)), C.byref(K.Tensor.of(u)), C.byref(K.Tensor.of(h)), 0, hb.data_ptr(), s)
def one():
    K.lib.mi355_glu_mul_mat_dq(t, 2, wa.data_ptr(), rs, wb.data_ptr(), rs, x.data_ptr(), k, h.data_ptr(), m, hb.data_ptr(), m, k, n, s)
for f in (three, one, three, one):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f.__name__, e0.elapsed_time(e1) / 50 * 1000, 'us')
