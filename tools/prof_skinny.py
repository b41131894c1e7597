#!/usr/bin/env python3
"""One skinny-kernel configuration launched repeatedly (profiling target for rocprofv3 --pmc / --kernel-trace).
usage: prof_skinny.py <type> <m> <k> <n> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B
t, m, k, n = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
rs = B.row_size(t, k)
w = torch.randint(0, 255, (m * rs,), dtype=torch.uint8, device="cuda")
x = torch.randn(n, k, device="cuda")
dst = torch.empty(n, m, device="cuda")
s = torch.cuda.current_stream().cuda_stream
if t in (B.Q4_K, B.Q5_K, B.Q6_K) and not os.environ.get("SK_V1"):
    q8, d8, bs8 = ops.quantize_act_q8k(x)
    fn = lambda: K.check(K.lib.mi355_mul_mat_skinny_k(t, w.data_ptr(), rs, q8.data_ptr(), d8.data_ptr(), bs8.data_ptr(), dst.data_ptr(), m, m, k, n, s), "sk2")
else:
    aq, ads = ops.quantize_act(x)
    fn = lambda: K.check(K.lib.mi355_mul_mat_skinny(t, w.data_ptr(), rs, aq.data_ptr(), ads.data_ptr(), dst.data_ptr(), m, m, k, n, s), "sk")
for _ in range(reps): fn()
torch.cuda.synchronize()
print("done", flush=True)
