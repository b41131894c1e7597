"""How the prefill-sized MUL_MAT_ID launch depends on the routing: ffn_down of an 8x7B layer (8 experts x 4096 x 14336 q4_K, 512 tokens, 2 experts
each) and gate / up (14336 x 4096), with every expert taking exactly 128 pairs, and with the routing skewed so that one / two experts pass 256
pairs (a second column tile each).  Whole op (activation conversion + bucketing + GEMM), torch events, median of 20."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import llm_amd.blocks as B
import llm_amd.ops as ops

def routing(kind, n_tok, n_exp):
    t = np.arange(n_tok)
    ids = np.stack([t % n_exp, (t + 1 + (t // n_exp) % (n_exp - 1)) % n_exp], 1)
    ids[:, 1] = np.where(ids[:, 1] == ids[:, 0], (ids[:, 0] + 1) % n_exp, ids[:, 1])
    if kind >= 1: ids[: 200, 0] = 0; ids[: 200, 1] = 1 + t[:200] % (n_exp - 1)          # expert 0: > 256 pairs
    if kind >= 2: ids[200 : 400, 0] = 1; ids[200 : 400, 1] = 2 + t[200:400] % (n_exp - 2)   # expert 1 as well
    return ids.astype(np.int32)

SHAPES = [("down", 4096, 14336, 2), ("gate/up", 14336, 4096, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "shapes":      # which of the two differences matters: the depth of K or the number of row tiles
    SHAPES = [("m4096 k4096", 4096, 4096, 2), ("m4096 k8192", 4096, 8192, 2), ("m4096 k14336", 4096, 14336, 2), ("m8192 k14336", 8192, 14336, 2),
              ("m14336 k14336", 14336, 14336, 2), ("m14336 k4096", 14336, 4096, 2)]
for name, m, k, b1 in SHAPES:
    w = torch.from_numpy(B.random_blocks(B.Q4_K, 8 * m, k, seed=1)).cuda()
    b = torch.randn((512, b1, k), device="cuda")
    for kind in ((0,) if len(SHAPES) > 2 else (0, 1, 2)):
        ids = torch.from_numpy(routing(kind, 512, 8)).cuda()
        counts = np.bincount(ids.cpu().numpy().ravel(), minlength=8)
        for _ in range(3): ops.mul_mat_id_dq(B.Q4_K, w, b, ids, m, 8)
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.mul_mat_id_dq(B.Q4_K, w, b, ids, m, 8); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000)
        stages = (m // 128) * int(sum((c + 255) // 256 for c in counts)) * (k // 64)
        print(f"{name:14s} {np.median(ts) * 256 / stages:6.2f} us per stage at 256 workgroups in flight; pairs per expert {counts.tolist()} tiles {int(sum((c + 255) // 256 for c in counts))}: {np.median(ts):7.1f} us", flush=True)
