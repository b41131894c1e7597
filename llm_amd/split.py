"""Row split of weight matrices across the GPUs of one node (the reference's `-sm row`).

Boundaries follow get_row_split (ggml/src/ggml-cuda/ggml-cuda.cu:768-781): rank r owns rows
[round_down(M*cum[r], R), round_down(M*cum[r+1], R)), the last rank takes the remainder; R is the
row rounding (128 = the MMQ tile height on AMD, ggml-cuda/mmq.cuh:126-129).  Every rank needs all
of x and produces a disjoint slice of dst; the slices are combined with an all-gather.
"""
ROW_ROUNDING = 128


def row_split(nrows, world_size, tensor_split=None, rounding=ROW_ROUNDING):
    """-> list of (row_low, row_high) per rank."""
    if tensor_split is None:
        tensor_split = [1.0] * world_size
    assert len(tensor_split) == world_size and world_size >= 1
    total = float(sum(tensor_split))
    cum, acc = [], 0.0
    for s in tensor_split:
        cum.append(acc / total)
        acc += s
    out = []
    for r in range(world_size):
        lo = 0 if r == 0 else int(nrows * cum[r])
        lo -= lo % rounding
        if r == world_size - 1:
            hi = nrows
        else:
            hi = int(nrows * cum[r + 1])
            hi -= hi % rounding
        out.append((lo, hi))
    return out


def max_slice_rows(nrows, world_size, tensor_split=None, rounding=ROW_ROUNDING):
    return max(hi - lo for lo, hi in row_split(nrows, world_size, tensor_split, rounding))


def all_gather_rows(local_out, nrows, rank, world_size, dist, pad_rows=None):
    """All-gather the row slices of a mat-vec output. local_out: [ncols, rows_of_this_rank] tensor.

    Slices can be unequal (remainder on the last rank), so every rank contributes a buffer padded to
    the largest slice and the padding is dropped on assembly.  `dist` is torch.distributed
    (backend nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
    import torch
    splits = row_split(nrows, world_size)
    pad = pad_rows or max(hi - lo for lo, hi in splits)
    ncols = local_out.shape[0]
    send = torch.zeros((ncols, pad), dtype=local_out.dtype, device=local_out.device)
    lo, hi = splits[rank]
    send[:, : hi - lo] = local_out
    # flat buffers: the concatenating form both RCCL and gloo accept
    recv = torch.empty(world_size * ncols * pad, dtype=local_out.dtype, device=local_out.device)
    dist.all_gather_into_tensor(recv, send.view(-1))
    recv = recv.view(world_size, ncols, pad)
    full = torch.empty((ncols, nrows), dtype=local_out.dtype, device=local_out.device)
    for r, (l, h) in enumerate(splits):
        full[:, l:h] = recv[r, :, : h - l]
    return full
