// moe_bucket.h -- device-side routing for MUL_MAT_ID at batch sizes > 1, shared by the bf16 GEMM (mmdq.hip, column
// tiles of 128) and the skinny int8-MFMA kernels (mmsk.hip, passes of 32 columns).
// One workgroup counts the (token, slot) pairs per expert in LDS, prefix-sums them into `off` (first entry of every
// expert in `sorted`) and `tile_off` (first column tile of every expert, tiles of `tile` columns), and scatters the
// pair ids.  The reference does this sort on the host after a stream synchronisation (ggml-cuda/mmq.cu:141-223).
// Pairs with an out-of-range expert id are dropped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ void __launch_bounds__(1024)
moe_bucket_kernel(const int32_t * ids, int64_t ids_stride, int n_tokens, int n_used, int n_expert, int tile, int * sorted, int * off, int * tile_off) {
    __shared__ int cnt[257], cur[257];
    const int tid = threadIdx.x, np = n_tokens * n_used;
    for (int e = tid; e <= n_expert; e += 1024) cnt[e] = 0;
    __syncthreads();
    for (int p = tid; p < np; p += 1024) {
        const int e = ids[(int64_t)(p / n_used) * ids_stride + p % n_used];
        if (e >= 0 && e < n_expert) atomicAdd(&cnt[e], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0, t = 0;
        for (int e = 0; e < n_expert; e++) {
            off[e] = o; tile_off[e] = t; cur[e] = o;
            o += cnt[e]; t += (cnt[e] + tile - 1) / tile;
        }
        off[n_expert] = o; tile_off[n_expert] = t;
    }
    __syncthreads();
    for (int p = tid; p < np; p += 1024) {
        const int e = ids[(int64_t)(p / n_used) * ids_stride + p % n_used];
        if (e >= 0 && e < n_expert) sorted[atomicAdd(&cur[e], 1)] = p;
    }
}

// MUL_MAT_ID mode of a kernel whose workgroups own a tile of columns: the routing tables made by moe_bucket_kernel
struct IdsArgs {
    const int * sorted;      // [n_pairs] pair ids (token * n_used + slot) grouped by expert; nullptr = plain MUL_MAT
    const int * off;         // [n_expert + 1]
    const int * tile_off;    // [n_expert + 1]
    int n_expert, n_used, act_ne1;
    int64_t w_expert_stride;
};

} // namespace
