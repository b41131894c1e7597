#!/usr/bin/env python3
"""Fixed cost and per-K-stage cost of the prefill GEMM: times mi355_mul_mat_dq at several K for one (type, M, N).
usage: bench_gemm_k.py <type id> <M> <N> <K,K,...>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_amd.blocks as B, llm_amd.kernels as K, llm_amd.ops as ops

t, m, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = torch.cuda.current_stream().cuda_stream
for k in (int(v) for v in sys.argv[4].split(",")):
    if t in B.FLOAT_WEIGHT_TYPES:      # f16 / bf16 / f32 weights: random bits of small magnitude
        w = (torch.randn((m, k), device="cuda") * 0.02).to({B.F16: torch.float16, B.BF16: torch.bfloat16}.get(t, torch.float32)).view(torch.uint8).reshape(m, -1)
    else:
        w = torch.from_numpy(B.random_blocks(t, m, k, seed=1)).cuda()
    y = ops.act_bf16(torch.randn((n, k), device="cuda"))
    dst = torch.empty((n, m), device="cuda")
    run = lambda: K.check(K.mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), y.data_ptr(), k, dst.data_ptr(), m, m, k, n, s), "dq")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{B.TYPE_NAMES[t]:5s} m={m:6d} n={n:4d} k={k:6d}  {us:9.1f} us  {2.0 * m * n * k / us * 1e-6:8.1f} TFLOP/s", flush=True)
