#!/usr/bin/env python3
"""Decode FLASH_ATTN_EXT (one query token, Llama-3-8B heads: 32 x 128, 8 KV heads) over context lengths: per-launch time
from the library's timing mode and the K/V bytes it streams (HBM roofline of the op)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_amd.kernels as K, llm_amd.ops as ops, llm_amd.blocks as B

nh, nkv, d = 32, 8, 128
for tkv, name, bpe in ((B.F16, "f16", 2.0), (B.Q8_0, "q8_0", 34 / 32)):
    for n_kv in (int(v) for v in os.environ.get("FA_NKV", "256,1024,4096,16384,65536").split(",")):
        q = torch.randn(nh, 1, d, device="cuda")
        if tkv == B.F16:
            # rotate over several caches so that long contexts are read from HBM, not the Infinity Cache
            ncopy = max(1, int(600e6 // (2 * nkv * n_kv * d * 2)))
            caches = [(torch.randn(nkv, n_kv, d, device="cuda").half(), torch.randn(nkv, n_kv, d, device="cuda").half()) for _ in range(min(ncopy, 16))]
        else:
            rs = B.row_size(tkv, d)
            caches = [(torch.randint(0, 255, (nkv, n_kv, rs), dtype=torch.uint8, device="cuda"), torch.randint(0, 255, (nkv, n_kv, rs), dtype=torch.uint8, device="cuda")) for _ in range(4)]
        reps = 30
        for i in range(3):
            ops.flash_attn_ext(q, *caches[i % len(caches)], None, tkv, n_kv, d ** -0.5)
        torch.cuda.synchronize()
        K.lib.mi355_timing_reset(); K.lib.mi355_timing_enable(1)
        for i in range(reps):
            ops.flash_attn_ext(q, *caches[i % len(caches)], None, tkv, n_kv, d ** -0.5)
        torch.cuda.synchronize(); K.lib.mi355_timing_enable(0)
        rep = {t: (n, us) for t, (n, us) in K.timing_report().items() if n}
        us = sum(u for _, u in rep.values()) / reps
        by = 2 * nkv * n_kv * d * bpe
        print(f"{name:5s} n_kv {n_kv:6d}: {us:8.2f} us  {by / us / 1e3:7.0f} GB/s  ({by / 1e6:.1f} MB)  {list(rep)}", flush=True)
        del caches
