#!/usr/bin/env python3
"""Prefill FLASH_ATTN_EXT (causal, f16 KV) per head size: per-launch time from the library's timing mode and the matrix-core rate
(4 d n_q n_kv / 2 flops per head for a causal prompt that starts at 0).  FA_SHAPES="d,nh,nhkv,nq,nkv;..." overrides the list."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

shapes = os.environ.get("FA_SHAPES", "128,32,8,512,512;128,32,8,512,2048;256,16,8,512,512;256,16,8,512,2048;96,32,32,512,512;96,32,32,512,2048;"
                                     "80,32,32,512,512;64,32,8,512,512;112,32,8,512,512")
for sh in shapes.split(";"):
    d, nh, nhkv, nq, nkv = (int(x) for x in sh.split(","))
    q = torch.randn(nh, nq, d, device="cuda")
    k = torch.randn(nhkv, nkv, d, device="cuda").half().view(torch.uint8).view(nhkv, nkv, -1)
    v = torch.randn(nhkv, nkv, d, device="cuda").half().view(torch.uint8).view(nhkv, nkv, -1)
    m = np.zeros(((nq + 63) // 64 * 64, nkv), np.float16)
    for i in range(nq):
        m[i, i + nkv - nq + 1:] = -np.inf
    mask = torch.from_numpy(m.view(np.int16)).cuda()
    reps = 10
    for _ in range(2):
        ops.flash_attn_ext(q, k, v, mask, B.F16, nkv, d ** -0.5)
    torch.cuda.synchronize()
    K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    for _ in range(reps):
        ops.flash_attn_ext(q, k, v, mask, B.F16, nkv, d ** -0.5)
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    rep = {t: (n, us) for t, (n, us) in K.timing_report().items() if n}
    us = sum(u for _, u in rep.values()) / reps
    vis = nq * (nkv - nq) + nq * (nq + 1) / 2           # visible (query, kv) pairs per head
    fl = 4.0 * d * vis * nh
    print(f"d {d:3d} heads {nh}/{nhkv} n_q {nq} n_kv {nkv}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {list(rep)}", flush=True)
