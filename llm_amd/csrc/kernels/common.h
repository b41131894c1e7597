// common.h -- device helpers shared by the gfx950 kernels (wave64, v_dot4c_i32_i8, 128-bit loads).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "mi355_kernels.h"
#include "timing.h"
#include <string.h>

#define QK   32
#define QK_K 256

typedef int   v4i __attribute__((ext_vector_type(4)));
typedef int   v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// 16-byte vectors at the 2-byte alignment GGUF blocks have (18 B q4_0, 34 B q8_0, 210 B q6_K);
// gfx950 runs in unaligned-access mode, so these still compile to one global_load_dwordx4.
struct __attribute__((packed, aligned(2))) v4i_a2 { v4i v; };
struct __attribute__((packed, aligned(2))) v2i_a2 { v2i v; };

static __device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }

// c + sum_i a.i8[i] * b.i8[i]  (v_dot4c_i32_i8)
static __device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// nontemporal 16-byte load.  NOT used for GGUF weight blocks: a block's header shares its 128-B
// line with the quants, and an nt line is dropped before the header load gets to it (measured
// -12 % on the q4_K mat-vec, llm_amd/csrc/lab/q4k_lab.hip).
static __device__ __forceinline__ v4i ld_nt16(const void * p) { return __builtin_nontemporal_load((const v4i *)p); }
static __device__ __forceinline__ v4i ld16_a2(const void * p) {
    v4i r; __builtin_memcpy(&r, __builtin_assume_aligned(p, 2), 16); return r;
}
static __device__ __forceinline__ uint16_t ld_u16(const void * p) { return *(const uint16_t *)p; }

// Sum over the 64 lanes of a wave, the same value in every lane.  Four DPP steps inside each row of 16 lanes (xor 1,
// xor 2, mirror within 8, mirror within 16: a few cycles each, against ~100 for the ds_bpermute behind __shfl_xor), then
// the four row sums are read with v_readlane and added in row order.  Every reduction in this library goes through
// here, so fused and unfused forms of an op add in the same order.
template <int CTRL>
static __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
static __device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// sum over the NW waves of a block (sh: NW floats of LDS): wave butterflies, then the wave sums added in wave order
template <int NW>
static __device__ __forceinline__ float block_sum_n(float v, float * sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < NW; i++) r += sh[i];
    return r;
}

// the same with the wave count known at run time (nw <= 16; sh: 16 floats)
static __device__ __forceinline__ float block_sum_rt(float v, float * sh, int nw) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; i++) r += sh[i];
    return r;
}

// per-byte (x - c) for bytes x in [0,127], c in [0,127]: no borrow can cross a byte
static __device__ __forceinline__ int sub_bytes(int x, int c4) { return ((x | 0x80808080) - c4) ^ 0x80808080; }

// q8_1 quantization of 4 consecutive values per lane, 8 consecutive lanes = one 32-block
// (quantize_row_q8_1_ref semantics, see quantize.hip).  Writes the 4 int8 and, from the block's
// first lane, {float(fp16(d)), float(sum q)} of the internal SoA layout.  All 8 lanes must be active.
static __device__ __forceinline__ void quant4_soa(const float v[4], int8_t * q_dst, float * ds_dst, bool leader) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float d  = __fdiv_rn(amax, 127.0f);
    const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
    int q[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float r = roundf(__fmul_rn(v[j], id));
        q[j] = fabsf(r) < 2147483648.0f ? (int)(int8_t)(int)r : 0;
        sum += q[j];
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o, 64);
    *(uint32_t *)q_dst = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
    if (leader) { v2f ds; ds.x = __half2float(__float2half_rn(d)); ds.y = (float)sum; *(v2f *)ds_dst = ds; }
}

// 16 consecutive values of a row (half a 32-block; the partner lane, lane ^ 1, holds the other half) -> q8_1 SoA.
// Same arithmetic as quant4_soa / quantize_row_q8_1_ref; `leader` (the even lane) writes {fp16(d), sum}.
static __device__ __forceinline__ void quant16_soa(const float v[16], int8_t * q_dst, float * ds_dst, bool leader) {
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) amax = fmaxf(amax, fabsf(v[j]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    const float d  = __fdiv_rn(amax, 127.0f);
    const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
    int sum = 0;
    v4i packed;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float r = roundf(__fmul_rn(v[4 * w + j], id));
            const int q = fabsf(r) < 2147483648.0f ? (int)(int8_t)(int)r : 0;
            sum += q;
            word |= (uint32_t)(q & 0xff) << (8 * j);
        }
        packed[w] = (int)word;
    }
    sum += __shfl_xor(sum, 1, 64);
    *(v4i *)q_dst = packed;
    if (leader) { v2f ds; ds.x = __half2float(__float2half_rn(d)); ds.y = (float)sum; *(v2f *)ds_dst = ds; }
}

// Optional in-kernel timeline stamps (lab builds only: -DMI355_STAMPS, llm_amd/csrc/lab/timeline.cpp).  Up to eight 100 MHz
// s_memrealtime stamps per wave: 0 entry, 1 activations staged, 2 weight loop done, 3 results stored, 4 activations fetched.
#ifdef MI355_STAMPS
extern unsigned long long * mi355_stamp_host_ptr;
#define STAMP_FIELD unsigned long long * stamps;
#define STAMP_SET(a) ((a).stamps = mi355_stamp_host_ptr)
#define STAMP_PTR(a) ((a).stamps)
#define STAMP_AT(p, i)                                                                                                         \
    do {                                                                                                                       \
        if ((p) && (threadIdx.x & 63) == 0)                                                                                    \
            ((unsigned long long *)(p))[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (i)] = \
                __builtin_amdgcn_s_memrealtime();                                                                              \
    } while (0)
#else
#define STAMP_FIELD
#define STAMP_SET(a) ((void)0)
#define STAMP_PTR(a) nullptr
#define STAMP_AT(p, i) ((void)0)
#endif

#define MI355_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
