#!/usr/bin/env python3
"""Prefill FLASH_ATTN_EXT as llama.cpp issues it for a long prompt: one ubatch of 512 queries against n_kv cached rows (causal on the
last 512), Llama-3-8B heads (32 x 128, 8 KV heads, f16 KV).  Time per launch: fixed cost and slope per 128 KV rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

nh, nkv, d, nq = 32, 8, 128, 512
for n in (int(v) for v in os.environ.get("FA_NKV", "512,1024,2048,4096,8192").split(",")):
    q = torch.randn(nh, nq, d, device="cuda")
    k = torch.randn(nkv, n, d, device="cuda").half()
    v = torch.randn(nkv, n, d, device="cuda").half()
    mask = torch.full((nq, n), float("-inf"), device="cuda").triu(n - nq + 1).half().view(torch.int16)
    for _ in range(2): ops.flash_attn_ext(q, k, v, mask, B.F16, n, d ** -0.5)
    torch.cuda.synchronize()
    K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    reps = 10
    for _ in range(reps): ops.flash_attn_ext(q, k, v, mask, B.F16, n, d ** -0.5)
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    rep = {t: (c, us) for t, (c, us) in K.timing_report().items() if c}
    us = sum(u for _, u in rep.values()) / reps
    fl = 4.0 * nq * (n - nq / 2) * d * nh
    print(f"n_q {nq} n_kv {n:5d}: {us:9.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s", flush=True)
