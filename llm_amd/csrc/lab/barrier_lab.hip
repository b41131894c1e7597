// barrier_lab.hip -- what does one workgroup barrier cost on gfx950?  A loop of `iters` barriers in workgroups of
// T threads (one workgroup per CU), with optional LDS traffic between them.  build: hipcc --offload-arch=gfx950 -O3
// usage: barrier_lab [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void loop_kernel(int iters, float * out, int lds_bytes) {
    extern __shared__ float lds[];
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 1) { lds[threadIdx.x] = acc; }
        if (MODE == 2) { asm volatile("s_nop 0"); }
        __syncthreads();
        if (MODE == 1) { acc += lds[(threadIdx.x + 64) % blockDim.x]; }
    }
    if (acc == 123.0f) out[threadIdx.x] = acc;
}

template <int MODE>
static void run(const char * name, int threads, int lds, int iters, int blocks) {
    float * out; hipMalloc(&out, 4096 * 4);
    hipFuncSetAttribute((const void *)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int r = 0; r < 2; r++) {
        const int it = r == 0 ? iters : 2 * iters;
        loop_kernel<MODE><<<blocks, threads, lds>>>(it, out, lds);
        hipEventRecord(e0);
        for (int j = 0; j < 10; j++) loop_kernel<MODE><<<blocks, threads, lds>>>(it, out, lds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[r], e0, e1); ms[r] /= 10;
    }
    printf("%-8s threads %4d lds %6d blocks %4d: %7.1f us at %d iters, %7.1f us at %d -> %6.1f ns per barrier (%4.0f cycles at 2.4 GHz)\n",
           name, threads, lds, blocks, ms[0] * 1e3, iters, ms[1] * 1e3, 2 * iters, (ms[1] - ms[0]) * 1e6 / iters, (ms[1] - ms[0]) * 1e6 / iters * 2.4);
    hipFree(out);
}

int main(int argc, char ** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    for (int threads : {64, 128, 256, 384, 512, 768, 1024}) run<0>("empty", threads, 1024, iters, 256);
    for (int threads : {256, 768}) run<0>("empty", threads, 144 * 1024, iters, 256);
    for (int threads : {256, 384}) run<0>("2 per CU", threads, 72 * 1024, iters, 512);
    for (int threads : {256, 768}) run<1>("lds", threads, 16 * 1024, iters, 256);
    for (int threads : {256, 768}) run<2>("nop", threads, 1024, iters, 256);
    return 0;
}
