"""Cold skinny launches for a PMC pass: q4_K 14336 x 4096 and 4096 x 14336, 8 and 32 columns, rotating over 1 GB of matrices.
usage (GPU box): cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out> -- python3 <repo>/tools/skinny_traffic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B
t = B.Q4_K
s = torch.cuda.current_stream().cuda_stream
for m, k in [(14336, 4096), (4096, 14336)]:
    rs = B.row_size(t, k)
    nmat = (1 << 30) // (m * rs)
    ws = [torch.randint(0, 255, (m * rs,), dtype=torch.uint8, device="cuda") for _ in range(nmat)]
    for n in (8, 32):
        x = torch.randn(n, k, device="cuda")
        q8, d8, bs8 = ops.quantize_act_q8k(x)
        dst = torch.empty(n, m, device="cuda")
        for rep in range(2):
            for w in ws:
                K.check(K.lib.mi355_mul_mat_skinny_k(t, w.data_ptr(), rs, q8.data_ptr(), d8.data_ptr(), bs8.data_ptr(), dst.data_ptr(), m, m, k, n, s), "sk")
        torch.cuda.synchronize()
print("done")
