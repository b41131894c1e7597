// valu_rate.hip -- issue cost of a few VALU instructions on gfx950 (one / two waves per SIMD), by s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int OP> __device__ __forceinline__ uint32_t op(uint32_t x, uint32_t y) {
    if (OP == 0) return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, x) * __builtin_bit_cast(us2, y)));   // v_pk_mul_lo_u16
    if (OP == 1) return x & y;                                                                                           // v_and_b32
    if (OP == 2) return __builtin_amdgcn_perm(x, y, 0x0c000c01u);                                                        // v_perm_b32
    if (OP == 3) return x * y;                                                                                           // v_mul_lo_u32
    if (OP == 4) return (x & 0xffffff) * (y & 0xffffff);                                                                 // and + v_mul_u32_u24
    if (OP == 5) return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, x) + __builtin_bit_cast(us2, y)));   // v_pk_add_u16
    return x;
}
template <int OP> __global__ void k(uint32_t * out, long long * cyc, int iters) {
    uint32_t v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 2654435761u + i;
    uint32_t y = threadIdx.x | 0x00030003;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = op<OP>(v[i], y);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s ^= v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char * name, int threads) {
    uint32_t * out; long long * cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 1000;
    k<OP><<<256, threads>>>(out, cyc, iters); hipDeviceSynchronize();
    k<OP><<<256, threads>>>(out, cyc, iters); hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; i++) m += h[i]; m /= 256;
    printf("%-16s %d waves/SIMD: %.2f cycles per instruction per wave\n", name, threads / 256, m / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("v_pk_mul_lo_u16", th); run<1>("v_and_b32", th); run<2>("v_perm_b32", th); run<3>("v_mul_lo_u32", th);
        run<4>("and+mul_u32_u24", th); run<5>("v_pk_add_u16", th);
    }
    return 0;
}
