"""MUL_MAT_ID at small batches, op level: the per-pair mat-vec (an expert streamed once per pair) against the skinny int8-MFMA
kernels in IDS mode (an expert streamed once per 32 of its pairs), for few- and many-expert shapes.  us per call incl. the wrappers'
quantization and allocations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import llm_amd.ops as ops, llm_amd.blocks as B
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = B.Q4_K
for (n_exp, n_used, m, k) in [(64, 6, 2048, 4096), (64, 6, 4096, 2048), (8, 2, 14336, 4096), (128, 8, 1536, 4096)]:
    rs = B.row_size(t, k)
    w = torch.randint(0, 255, (n_exp * m * rs,), dtype=torch.uint8, device='cuda')
    for n_tok in (2, 4, 8, 16, 32):
        b = torch.randn(n_tok, 1, k, device='cuda')
        ids = torch.stack([torch.randperm(n_exp)[:n_used] for _ in range(n_tok)]).to(torch.int32).cuda()
        r = {}
        r['pair'] = bench(lambda: ops.mul_mat_id(t, w, b, ids, m, n_exp))
        r['skinny'] = bench(lambda: ops.mul_mat_id_skinny(t, w, b, ids, m, n_exp))
        print(f"E={n_exp} top{n_used} {m}x{k} tok={n_tok:3d} pairs={n_tok*n_used:4d}  " + "  ".join(f"{k_} {v:8.1f} us" for k_, v in r.items()), flush=True)
