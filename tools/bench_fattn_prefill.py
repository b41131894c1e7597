#!/usr/bin/env python3
"""Prefill FLASH_ATTN_EXT (causal, Llama-3-8B heads: 32 x 128, 8 KV heads, f16 KV): time and TFLOP/s counting the causal
half (2 * 2 * n^2/2 * d * heads)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

nh, nkv, d = 32, 8, 128
for n in (512, 2048, 4096):
    q = torch.randn(nh, n, d, device="cuda")
    k = torch.randn(nkv, n, d, device="cuda").half()
    v = torch.randn(nkv, n, d, device="cuda").half()
    mask = torch.full((n, n), float("-inf"), device="cuda").triu(1).half().view(torch.int16)
    for _ in range(2): ops.flash_attn_ext(q, k, v, mask, B.F16, n, d ** -0.5)
    torch.cuda.synchronize()
    K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
    reps = 5
    for _ in range(reps): ops.flash_attn_ext(q, k, v, mask, B.F16, n, d ** -0.5)
    torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
    rep = {t: (c, us) for t, (c, us) in K.timing_report().items() if c}
    us = sum(u for _, u in rep.values()) / reps
    fl = 2.0 * n * n * d * nh          # causal half of 4 n^2 d h
    print(f"n {n:5d}: {us:9.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s  {list(rep)}", flush=True)
