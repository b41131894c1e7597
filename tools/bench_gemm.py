#!/usr/bin/env python3
"""Prefill GEMM micro-benchmark: TFLOP/s of mi355_mul_mat_dq (bf16 MFMA) and mi355_mul_mat_q (int8 MFMA) / mi355_mul_mat_qk (digit-split int8 MFMA, "qk") on the
Llama-3-8B shapes at N = 512 (flops = 2*M*N*K, tests/test-backend-ops.cpp:2262-2265).  usage: bench_gemm.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import llm_amd.blocks as B, llm_amd.kernels as K, llm_amd.ops as ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
which = sys.argv[2] if len(sys.argv) > 2 else "dq"
only = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else None      # "type,m,k": one shape only
s = torch.cuda.current_stream().cuda_stream
for t, m, k in ((B.Q4_K, 1024, 4096), (B.Q4_K, 4096, 4096), (B.Q4_K, 14336, 4096), (B.Q4_K, 4096, 14336), (B.Q6_K, 4096, 14336), (B.Q6_K, 128256, 4096),
                (B.Q8_0, 14336, 4096), (B.Q4_0, 14336, 4096), (B.Q5_K, 1024, 8192)):
    if only and (t, m, k) != only: continue
    w = torch.from_numpy(B.random_blocks(t, min(m, 16384), k, seed=1)).cuda()
    mm = min(m, 16384)
    x = torch.randn((n, k), device="cuda")
    dst = torch.empty((n, mm), device="cuda")
    if which == "dq":
        y = ops.act_bf16(x)
        run = lambda: K.check(K.mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), y.data_ptr(), k, dst.data_ptr(), mm, mm, k, n, s), "dq")
    elif which == "qk":
        if t not in (B.Q4_K, B.Q5_K): continue
        q8, d8, bs8 = ops.quantize_act_q8k(x)
        run = lambda: K.check(K.lib.mi355_mul_mat_qk(t, w.data_ptr(), B.row_size(t, k), q8.data_ptr(), d8.data_ptr(), bs8.data_ptr(), dst.data_ptr(), mm, mm, k, n, s), "qk")
    else:
        aq, ads = ops.quantize_act(x)
        run = lambda: K.check(K.lib.mi355_mul_mat_q(t, w.data_ptr(), B.row_size(t, k), aq.data_ptr(), ads.data_ptr(), dst.data_ptr(), mm, mm, k, n, s), "q")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{which} {B.TYPE_NAMES[t]:5s} m={mm:6d} k={k:6d} n={n:4d}  {us:9.1f} us  {2.0 * mm * n * k / us * 1e-6:8.1f} TFLOP/s", flush=True)
