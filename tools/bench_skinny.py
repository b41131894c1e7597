#!/usr/bin/env python3
"""Small-batch MUL_MAT (1 <= N <= 256): mat-vec (ncols), skinny int8-MFMA and dequantize+bf16 GEMM kernels side by side.
Per-launch execution time from the library's timing mode; GB/s = weight bytes / time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

def timed(fn, tags, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    tot = 0.0
    for tag in tags:
        c, us = K.timing_read(tag)
        tot += us
    return tot / reps

types = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [B.Q4_K, B.Q6_K, B.Q8_0]
shapes = [(4096, 4096), (14336, 4096), (4096, 14336)]
s = torch.cuda.current_stream().cuda_stream
for t in types:
    name = B.TYPE_NAMES[t]
    for m, k in shapes:
        rs = B.row_size(t, k)
        w = torch.randint(0, 255, (m * rs,), dtype=torch.uint8, device="cuda")
        for n in (int(v) for v in os.environ.get("SK_N", "1,2,4,8,16,32,64,128,256").split(",")):
            x = torch.randn(n, k, device="cuda")
            aq, ads = ops.quantize_act(x)
            xb = ops.act_bf16(x)
            dst = torch.empty(n, m, device="cuda")
            res = {}
            if n <= 8:
                res["mmvq"] = timed(lambda: ops.mul_mat_vec_q(t, w, m, k, aq, ads, n, dst=dst), ["mmvq_%s_n%d" % (name, n)])
            def sk():
                for c0 in range(0, n, 32):
                    K.check(K.lib.mi355_mul_mat_skinny(t, w.data_ptr(), rs, aq[c0:].data_ptr(), ads[c0:].data_ptr(), dst[c0:].data_ptr(), m, m, k, min(32, n - c0), s), "sk")
            res["skinny"] = timed(sk, ["mmsk_%s" % name])
            if t in (B.Q4_K, B.Q5_K, B.Q6_K):
                q8, d8, bs8 = ops.quantize_act_q8k(x)
                def sk2():
                    for c0 in range(0, n, 32):
                        K.check(K.lib.mi355_mul_mat_skinny_k(t, w.data_ptr(), rs, q8[c0:].data_ptr(), d8[c0:].data_ptr(), bs8[c0:].data_ptr(),
                                                             dst[c0:].data_ptr(), m, m, k, min(32, n - c0), s), "sk2")
                res["skinny_k"] = timed(sk2, ["mmsk2_%s" % name, "mmsk6_%s" % name])
            if n > 8:
                res["mmdq"] = timed(lambda: K.check(K.mul_mat_dq(t, w.data_ptr(), rs, xb.data_ptr(), k, dst.data_ptr(), m, m, k, n, s), "dq"), ["mmdq_%s" % name, "mmdq_kz_sum"])
            wb = m * rs
            print(f"{name:5s} {m:6d}x{k:6d} N={n:4d}  " + "  ".join(f"{kk} {v:8.2f} us ({wb / v / 1e3:6.0f} GB/s)" for kk, v in res.items()), flush=True)
