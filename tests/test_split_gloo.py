"""Row split (`-sm row`) across ranks, world_size 2 and 3 over gloo on the CPU.

The sharding logic (llm_amd/split.py: boundaries per get_row_split, ggml-cuda.cu:768-781, padded
all-gather of unequal slices) is device independent; here each rank computes its row slice with the
oracle (tests may use it) and the gathered result must equal the unsplit oracle result bit for bit.
On the GPU box the same code runs with backend nccl (= RCCL) from bench.py --gpus N.
"""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from llm_amd.split import max_slice_rows, row_split  # noqa: E402


def test_row_split_boundaries():
    # equal split, rounding 128, remainder on the last rank (ggml-cuda.cu:768-781)
    assert row_split(4096, 1) == [(0, 4096)]
    assert row_split(4096, 2) == [(0, 2048), (2048, 4096)]
    assert row_split(128256, 8)[0] == (0, 16000 - 16000 % 128)
    for m in (128, 1000, 4096, 14336, 128256):
        for ws in (1, 2, 3, 4, 8):
            sp = row_split(m, ws)
            assert sp[0][0] == 0 and sp[-1][1] == m
            for (l0, h0), (l1, h1) in zip(sp, sp[1:]):
                assert h0 == l1 and l0 <= h0
            for lo, hi in sp[:-1]:
                assert lo % 128 == 0 and hi % 128 == 0
            assert max_slice_rows(m, ws) == max(h - l for l, h in sp)


def test_row_split_tensor_split_weights():
    sp = row_split(4096, 2, tensor_split=[3, 1])
    assert sp == [(0, 3072), (3072, 4096)]
    sp = row_split(1024, 4, tensor_split=[0, 1, 1, 0])   # empty slices are legal
    assert sp[0] == (0, 0) and sp[-1][1] == 1024
    assert sum(h - l for l, h in sp) == 1024


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, k, ncols, type_id, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        import llm_amd.blocks as B
        import oracle_lib as O
        from llm_amd.split import all_gather_rows
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        w = B.random_blocks(type_id, m, k, seed=7)                       # same on every rank
        x = np.random.default_rng(3).standard_normal((ncols, k)).astype(np.float32)
        lo, hi = row_split(m, world)[rank]
        rs = B.row_size(type_id, k)
        wl = np.ascontiguousarray(w.reshape(m, rs)[lo:hi])
        if hi > lo:
            local = O.mul_mat(type_id, wl, x, hi - lo, k, ncols, "q8_1").reshape(ncols, hi - lo)
        else:
            local = np.zeros((ncols, 0), np.float32)
        full = all_gather_rows(torch.from_numpy(local.copy()), m, rank, world, dist).numpy()
        ref = O.mul_mat(type_id, w, x, m, k, ncols, "q8_1").reshape(ncols, m)
        ok = bool(np.array_equal(full.view(np.uint32), ref.view(np.uint32)))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok, None))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("world,m,k,ncols,tname", [
    (2, 512, 256, 1, "Q4_K"),      # equal slices
    (2, 384, 256, 3, "Q6_K"),      # 128 / 256: unequal, padded gather
    (3, 640, 128, 2, "Q8_0"),      # 128 / 256(->128,128..) remainder on the last rank
    (2, 100, 64, 1, "Q4_0"),       # fewer rows than the rounding: rank 0 owns nothing
])
def test_row_split_all_gather_gloo(world, m, k, ncols, tname):
    import torch.multiprocessing as mp
    import llm_amd.blocks as B
    import oracle_lib as O  # noqa: F401  (builds liboracle once, before the ranks race for it)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, m, k, ncols, getattr(B, tname), q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert err is None, f"rank {rank}: {err}"
        assert ok, f"rank {rank}: gathered result differs from the unsplit result"
