"""33..128 columns: skinny passes of 32, the bf16 GEMM and the digit-split int8 tile kernel (mmqk) side by side, cold (rotating over
>= 1 GB of matrices).  us per product, q4_K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

def bench(fn, n, reps=3):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3

t = B.Q4_K
s = torch.cuda.current_stream().cuda_stream
for m, k in [(4096, 4096), (14336, 4096), (4096, 14336)]:
    rs = B.row_size(t, k)
    nmat = max(2, (1 << 30) // (m * rs))
    ws = [torch.randint(0, 255, (m * rs,), dtype=torch.uint8, device="cuda") for _ in range(nmat)]
    for n in (32, 40, 48, 64, 96, 128):
        x = torch.randn(n, k, device="cuda")
        q8, d8, bs8 = ops.quantize_act_q8k(x)
        xb = ops.act_bf16(x)
        dst = torch.empty(n, m, device="cuda")
        def sk(i):
            w = ws[i]
            for c0 in range(0, n, 32):
                K.check(K.lib.mi355_mul_mat_skinny_k(t, w.data_ptr(), rs, q8[c0:].data_ptr(), d8[c0:].data_ptr(), bs8[c0:].data_ptr(),
                                                     dst[c0:].data_ptr(), m, m, k, min(32, n - c0), s), "sk")
        def dq(i):
            K.check(K.mul_mat_dq(t, ws[i].data_ptr(), rs, xb.data_ptr(), k, dst.data_ptr(), m, m, k, n, s), "dq")
        def qk(i):
            K.check(K.lib.mi355_mul_mat_qk(t, ws[i].data_ptr(), rs, q8.data_ptr(), d8.data_ptr(), bs8.data_ptr(), dst.data_ptr(), m, m, k, n, s), "qk")
        print(f"q4_K {m:6d}x{k:6d} N={n:4d}  skinny {bench(sk, nmat):7.1f}  mmdq {bench(dq, nmat):7.1f}  mmqk {bench(qk, nmat):7.1f} us", flush=True)
