"""mi355_flash_attn_ext_noalloc (the entry point that allocates nothing: its split-KV scratch is a fixed device array
inside the library) against mi355_flash_attn_ext with a caller-allocated workspace of the same size: bit-identical
outputs, on two calls in a row, the second with a smaller input (the array keeps the first call's contents).
Collected last (tests/test_zz_*): the round's end runs pytest -x."""
import numpy as np
import pytest
import torch

import llm_amd.blocks as B
import llm_amd.kernels as K
import llm_amd.ops as ops

pytestmark = pytest.mark.gpu


def _case(n_head, n_head_kv, d, n_q, n_kv, type_kv, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    q = torch.randn((n_head, n_q, d), device="cuda", generator=g)
    if type_kv == B.F16:
        k = (torch.randn((n_head_kv, n_kv, d), device="cuda", generator=g) * 0.5).to(torch.float16).view(torch.uint8).contiguous()
        v = (torch.randn((n_head_kv, n_kv, d), device="cuda", generator=g) * 0.5).to(torch.float16).view(torch.uint8).contiguous()
    else:
        rows = [torch.from_numpy(B.random_blocks(type_kv, n_head_kv * n_kv, d, seed=seed + i)).cuda().view(n_head_kv, n_kv, -1) for i in (1, 2)]
        k, v = rows
    n_q_pad = (n_q + 63) // 64 * 64
    mask = torch.zeros((n_q_pad, n_kv), dtype=torch.float16, device="cuda")
    mask[:, n_kv - 7:] = float("-inf")                       # a masked tail, as the padded KV cache has
    return q, k, v, mask.view(torch.int16), 1.0 / float(np.sqrt(d))


def test_noalloc_matches_workspace_entry_point_twice():
    wsb = K.lib.mi355_flash_attn_ext_noalloc_max_bytes()
    assert wsb == 65536 + (4 << 20)
    # first call: 32 heads, long context (many KV slices -> the counters and most of the partial area are used);
    # second call: smaller in every dimension, another KV type
    for n_head, n_head_kv, d, n_q, n_kv, t, seed in ((32, 8, 128, 1, 4096, B.F16, 1), (8, 2, 64, 2, 512, B.Q8_0, 2),
                                                      (64, 8, 128, 1, 1024, B.Q4_0, 3)):
        q, k, v, mask, scale = _case(n_head, n_head_kv, d, n_q, n_kv, t, seed)
        want = ops.flash_attn_ext(q, k, v, mask, t, n_kv, scale, workspace_bytes=wsb)
        got = ops.flash_attn_ext(q, k, v, mask, t, n_kv, scale, noalloc=True)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (n_head, n_kv, t)


def test_noalloc_rejects_sizes_beyond_the_array():
    q, k, v, mask, scale = _case(32, 8, 128, 64, 256, B.F16, 4)       # 64 x 32 heads x 130 floats x 64 slices > 4 MiB
    with pytest.raises(K.Mi355Error) as e:
        ops.flash_attn_ext(q, k, v, mask, B.F16, 256, scale, noalloc=True)
    assert e.value.code == -5                                          # MI355_E_WORKSPACE


def test_noalloc_is_capturable_into_a_graph():
    """The callers the entry point exists for: under stream capture nothing may allocate or synchronise.  The call is captured
    (memset node + kernel nodes), replayed twice, and gives the bits of the workspace form.  mi355_flash_attn_ext_noalloc_max_bytes()
    has resolved the array's address on this device before the capture starts, as the header asks."""
    wsb = K.lib.mi355_flash_attn_ext_noalloc_max_bytes()
    q, k, v, mask, scale = _case(32, 8, 128, 1, 2048, B.F16, 5)
    want = ops.flash_attn_ext(q, k, v, mask, B.F16, 2048, scale, workspace_bytes=wsb)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = ops.flash_attn_ext(q, k, v, mask, B.F16, 2048, scale, noalloc=True)
    for _ in range(2):
        got.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
