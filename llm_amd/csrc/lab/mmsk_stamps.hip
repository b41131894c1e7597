// mmsk_stamps.hip -- diagnostic build of the digit-split skinny kernel with s_memtime stamps around its phases.
// (The stamps go to a buffer of their own; the product build has none.)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../../include -I../kernels -o mmsk_stamps mmsk_stamps.hip -L../../lib -lmi355_kernels -Wl,-rpath,'$ORIGIN/../../lib'
#define SK_STAMPS 1
#include "../kernels/mmsk.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
namespace mi355 { bool timing_on() { return false; } void timing_events(const char *, hipEvent_t *, hipEvent_t *) {} }

int main(int argc, char ** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 14336, k = argc > 2 ? atoi(argv[2]) : 4096, n = argc > 3 ? atoi(argv[3]) : 8;
    const int tpw = argc > 4 ? atoi(argv[4]) : 1;       // row tiles per workgroup
    const int WAVES = 8;
    const int64_t rs = (int64_t)k / 256 * 144;
    std::vector<uint8_t> hw((size_t)m * rs);
    srand(1);
    for (auto & b : hw) b = (uint8_t)rand();
    for (int64_t i = 0; i < (int64_t)m * (k / 256); i++) { hw[i * 144 + 1] = 0x2c; hw[i * 144 + 3] = 0x2c; }   // sane f16 d, dmin
    uint8_t * W; int8_t * aq; float * ad; int16_t * bs; float * dst; long long * stamps;
    hipMalloc(&W, hw.size()); hipMemcpy(W, hw.data(), hw.size(), hipMemcpyHostToDevice);
    hipMalloc(&aq, (size_t)32 * k); hipMemset(aq, 1, (size_t)32 * k);
    hipMalloc(&ad, (size_t)32 * (k / 256) * 4); hipMemset(ad, 0, (size_t)32 * (k / 256) * 4);
    hipMalloc(&bs, (size_t)32 * (k / 16) * 2); hipMemset(bs, 0, (size_t)32 * (k / 16) * 2);
    hipMalloc(&dst, (size_t)32 * m * 4);
    const unsigned nb = tpw == 2 ? (m + 63) / 64 : (m + 31) / 32, ky = tpw == 1 && nb < 256 && k >= 2048 ? 2 : 1;
    const size_t nst = (size_t)nb * ky * WAVES * 8;
    hipMalloc(&stamps, nst * 8);
    S2Args a = {}; a.W = W; a.w_row_stride = rs; a.aq = aq; a.ad = ad; a.bs = bs; a.dst = dst; a.dst_col_stride = m; a.m = m; a.n = n; a.k = k; a.stamps = stamps;
    for (int it = 0; it < 5; it++) {
        hipMemset(dst, 0, (size_t)32 * m * 4);
        if (tpw == 2) hipLaunchKernelGGL((mmsk2_kernel<MI355_TYPE_Q4_K, 8, 2>), dim3(nb, ky), dim3(512), 0, 0, a);
        else          hipLaunchKernelGGL((mmsk2_kernel<MI355_TYPE_Q4_K, 8>), dim3(nb, ky), dim3(512), 0, 0, a);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(nst);
    hipMemcpy(h.data(), stamps, nst * 8, hipMemcpyDeviceToHost);
    const char * names[5] = {"setup+first issue", "first step (incl. wait for first data)", "remaining steps", "wait at barrier", "reduce+store"};
    for (int p = 0; p < 5; p++) {
        std::vector<long long> d;
        for (size_t w = 0; w < nst / 8; w++) d.push_back(h[w * 8 + p + 1] - h[w * 8 + p]);
        std::sort(d.begin(), d.end());
        printf("%-42s median %7lld  p10 %7lld  p90 %7lld cycles\n", names[p], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    // wall: s_memrealtime (100 MHz) of every wave's start and end relative to the first start
    long long t0 = h[7], t1 = 0;
    for (size_t w = 0; w < nst / 8; w++) { t0 = std::min(t0, h[w * 8 + 7]); t1 = std::max(t1, h[w * 8 + 6]); }
    std::vector<long long> starts, life;
    for (size_t w = 0; w < nst / 8; w++) { starts.push_back(h[w * 8 + 7] - t0); life.push_back(h[w * 8 + 6] - h[w * 8 + 7]); }
    std::sort(starts.begin(), starts.end()); std::sort(life.begin(), life.end());
    printf("kernel span %.2f us; wave start: median %.2f us p90 %.2f us max %.2f us; wave lifetime median %.2f us p90 %.2f us\n", (t1 - t0) / 100.0,
           starts[starts.size() / 2] / 100.0, starts[starts.size() * 9 / 10] / 100.0, starts.back() / 100.0, life[life.size() / 2] / 100.0, life[life.size() * 9 / 10] / 100.0);
    long long cyc = 0, rt = 0;
    for (size_t w = 0; w < nst / 8; w++) { cyc += h[w * 8 + 5] - h[w * 8 + 0]; rt += h[w * 8 + 6] - h[w * 8 + 7]; }
    printf("clock %.2f GHz\n", (double)cyc / rt * 0.1);
    return 0;
}
