"""The kernel library allocates no device memory of its own (every workspace comes from the caller, so every entry
point can be captured into a graph and called from a serving loop).  Checked by accounting: across a battery of entry points at shapes
the process has not seen before -- decode mat-vecs and their fused forms, the skinny kernels (single, group, 33..64 columns, MUL_MAT_ID
single and pair), the prefill GEMM with split K, MUL_MAT_ID GEMM, flash attention with and without KV split, the elementwise producers
-- the driver's free memory may only fall by what torch's allocator reserved for the tensors and workspaces the test itself made."""
import numpy as np
import pytest
import torch

import llm_amd.blocks as B
import llm_amd.ops as ops

pytestmark = pytest.mark.gpu


def _battery(scale):
    """one pass over the entry points; `scale` changes every shape so that nothing can be served from a size seen before"""
    g = torch.Generator(device="cuda"); g.manual_seed(7 + scale)
    k, m = 1024 * scale, 768 * scale
    outs = []
    for t in (B.Q4_K, B.Q6_K, B.Q8_0):
        w = torch.from_numpy(B.random_blocks(t, m, k, seed=scale + t)).cuda()
        for n in (1, 3, 8, 24, 48, 200):
            x = torch.randn((n, k), device="cuda", generator=g)
            outs.append(ops.mul_mat(t, w, x, m))
        x = torch.randn((40, k), device="cuda", generator=g)
        if t != B.Q8_0:
            outs.append(ops.mul_mat_skinny_k(t, w, x, m, pass_cols=64))
            outs += ops.mul_mat_skinny_k_group(t, [w, w], x, [m, m])
        outs.append(ops.mul_mat_dq(t, w, torch.randn((300, k), device="cuda", generator=g), m))
    # MUL_MAT_ID: per-pair mat-vec, skinny (single and pair), grouped GEMM
    n_exp, n_used = 8, 2
    we = torch.from_numpy(B.random_blocks(B.Q4_K, n_exp * 256 * scale, k, seed=50 + scale)).cuda()
    for n_tok in (2, 16, 80):
        b = torch.randn((n_tok, 1, k), device="cuda", generator=g)
        ids = torch.stack([torch.randperm(n_exp, device="cuda", generator=g)[:n_used] for _ in range(n_tok)]).to(torch.int32)
        outs.append(ops.mul_mat_id(B.Q4_K, we, b, ids, 256 * scale, n_exp))
        outs.append(ops.mul_mat_id_skinny(B.Q4_K, we, b, ids, 256 * scale, n_exp))
        outs += list(ops.mul_mat_id_skinny_pair(B.Q4_K, we, we, b, ids, 256 * scale, n_exp))
        outs.append(ops.mul_mat_id_dq(B.Q4_K, we, b, ids, 256 * scale, n_exp))
    # flash attention: short context (no split), long context (KV split through the caller's workspace), a prefill batch
    for n_q, n_kv in ((1, 256), (1, 4096 * scale), (64, 512 * scale)):
        q = torch.randn((8, n_q, 128), device="cuda", generator=g)
        kk = (torch.randn((2, n_kv, 128), device="cuda", generator=g) * 0.5).to(torch.float16).view(torch.uint8).contiguous()
        vv = (torch.randn((2, n_kv, 128), device="cuda", generator=g) * 0.5).to(torch.float16).view(torch.uint8).contiguous()
        mask = torch.zeros(((n_q + 63) // 64 * 64, n_kv), dtype=torch.float16, device="cuda").view(torch.int16)
        outs.append(ops.flash_attn_ext(q, kk, vv, mask, B.F16, n_kv, 0.088))
    # ... other head sizes (the kernels built for the next size up; 256 on the matrix cores; V smaller than K; softcap)
    for d, dv, n_q, n_kv, cap in ((96, None, 1, 3000 * scale, 0.0), (80, None, 40, 300 * scale, 0.0), (256, None, 64, 400 * scale, 30.0), (192, 128, 2, 2000 * scale, 0.0)):
        q = torch.randn((8, n_q, d), device="cuda", generator=g)
        kk = (torch.randn((2, n_kv, d), device="cuda", generator=g) * 0.5).to(torch.float16).view(torch.uint8).contiguous()
        vv = (torch.randn((2, n_kv, dv or d), device="cuda", generator=g) * 0.5).to(torch.float16).view(torch.uint8).contiguous()
        mask = torch.zeros(((n_q + 63) // 64 * 64, n_kv), dtype=torch.float16, device="cuda").view(torch.int16)
        outs.append(ops.flash_attn_ext(q, kk, vv, mask, B.F16, n_kv, 0.088, 0.0, cap, dv=dv))
    # MUL_MAT_ID with the caller's tables / ready planes, and the element-wise ops added at the end of round 3
    b = torch.randn((12, 1, k), device="cuda", generator=g)
    ids = torch.stack([torch.randperm(n_exp, device="cuda", generator=g)[:n_used] for _ in range(12)]).to(torch.int32)
    import llm_amd.kernels as K
    ws = torch.zeros(K.lib.mi355_mul_mat_id_skinny_workspace_bytes(B.Q4_K, k, 12, 1, n_used, n_exp), dtype=torch.uint8, device="cuda")
    tables = torch.zeros(K.lib.mi355_mul_mat_id_tables_bytes(12, n_used, n_exp), dtype=torch.uint8, device="cuda")
    outs += list(ops.mul_mat_id_skinny_ex(B.Q4_K, we, we, b, ids, 256 * scale, n_exp, ws, tables, 0))
    outs.append(ops.mul_mat_id_skinny_ex(B.Q4_K, we, None, b, ids, 256 * scale, n_exp, ws, tables, 3))
    x = torch.rand((7 * scale, 333), device="cuda", generator=g) + 0.5
    for op in (4, 8, 10, 19):                      # TANH, GELU, SILU, LOG
        y = torch.empty_like(x)
        K.check(K.lib.mi355_map_f32(op, K.Tensor.of(x), K.Tensor.of(y), 0.0, 0.0, torch.cuda.current_stream().cuda_stream), "map")
        outs.append(y)
    for l2 in (0, 1):
        y = torch.empty_like(x)
        K.check(K.lib.mi355_norm_f32(K.Tensor.of(x), K.Tensor.of(y), 1e-5, l2, torch.cuda.current_stream().cuda_stream), "norm")
        outs.append(y)
    e = torch.randn((5 * scale, 2, 512 * scale), device="cuda", generator=g)       # the experts' rows of a batch, combined
    wts = torch.rand((5 * scale, 2), device="cuda", generator=g)
    tot = torch.empty((5 * scale, 512 * scale), device="cuda")
    K.check(K.lib.mi355_moe_combine2_f32(e.data_ptr(), wts.data_ptr(), 8, 4, e.data_ptr(), tot.data_ptr(), 512 * scale, 5 * scale,
                                         torch.cuda.current_stream().cuda_stream), "moe_combine2")
    outs.append(tot)
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in outs)


def test_library_allocates_no_device_memory():
    _battery(1)                                   # code objects loaded, torch's pools warm, lazily created runtime state in place
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    res0 = torch.cuda.memory_reserved()
    _battery(2)                                   # every shape new and larger
    _battery(3)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    res1 = torch.cuda.memory_reserved()
    outside_torch = (free0 - free1) - (res1 - res0)
    assert outside_torch <= (2 << 20), f"{outside_torch / 2**20:.1f} MiB of device memory were allocated outside torch's allocator"
