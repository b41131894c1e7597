// fattn_stamps.hip -- diagnostic build of the decode attention kernel with s_memtime stamps around its phases.
// (The stamps go to a buffer of their own; the product build has none.)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../../include -I../kernels -o fattn_stamps fattn_stamps.hip
// usage: fattn_stamps [n_kv] [nsplit]
#define FA_STAMPS 1
#include "../kernels/fattn.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
namespace mi355 { bool timing_on() { return false; } void timing_events(const char *, hipEvent_t *, hipEvent_t *) {} }

int main(int argc, char ** argv) {
    const int n_kv = argc > 1 ? atoi(argv[1]) : 256, nsplit = argc > 2 ? atoi(argv[2]) : 8;
    const int nh = 32, nkv = 8, D = 128, R = 2;
    const int ncopy = 16;
    float * q, * dst, * part; char * k, * v; int * cnt; long long * stamps; uint16_t * mask;
    hipMalloc(&q, nh * D * 4); hipMemset(q, 0, nh * D * 4);
    const size_t kvb = (size_t)nkv * n_kv * D * 2;
    hipMalloc(&k, kvb * ncopy); hipMemset(k, 0, kvb * ncopy);
    hipMalloc(&v, kvb * ncopy); hipMemset(v, 0, kvb * ncopy);
    hipMalloc(&mask, (size_t)n_kv * 2 * 64); hipMemset(mask, 0, (size_t)n_kv * 2 * 64);
    hipMalloc(&dst, nh * D * 4);
    hipMalloc(&part, (size_t)nh * 64 * (D + 2) * 4);
    hipMalloc(&cnt, 65536); hipMemset(cnt, 0, 65536);
    const int hgroups = 4 / R, nwg = nkv * hgroups * nsplit;
    hipMalloc(&stamps, (size_t)nwg * 64);
    FArgs a = {};
    a.q = (const char *)q; a.q_nb1 = D * 4; a.q_nb2 = D * 4; a.q_nb3 = nh * D * 4;
    a.k_nb1 = D * 2; a.k_nb2 = (int64_t)n_kv * D * 2; a.k_nb3 = kvb;
    a.v_nb1 = D * 2; a.v_nb2 = (int64_t)n_kv * D * 2; a.v_nb3 = kvb;
    a.mask = (const char *)mask; a.m_nb1 = n_kv * 2; a.m_nb2 = 0; a.m_nb3 = 0; a.m_ne2 = 1; a.m_ne3 = 1;
    a.dst = (char *)dst; a.d_nb1 = D * 4; a.d_nb2 = nh * D * 4; a.d_nb3 = nh * D * 4;
    a.n_q = 1; a.n_kv = n_kv; a.n_head = nh; a.n_head_kv = nkv; a.ne3 = 1;
    a.scale = 0.088f; a.max_bias = 0; a.logit_softcap = 0; a.m0 = a.m1 = 1; a.n_head_log2 = 32;
    a.nsplit = nsplit; a.part = part; a.counters = cnt; a.act_q = nullptr; a.act_ds = nullptr; a.stamps = stamps;
    const int G_ = FA_THREADS / (D / 8); const size_t sh = ((size_t)R * G_ * (D + 2) + (size_t)R * G_ + 2 * R) * sizeof(float);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 12; it++) {
        a.k = k + (size_t)(it % ncopy) * kvb; a.v = v + (size_t)(it % ncopy) * kvb;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fattn_vec_kernel<128, 2, MI355_TYPE_F16, MI355_TYPE_F16>), dim3(nkv * hgroups, 1, nsplit), dim3(FA_THREADS), sh, 0, a);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("n_kv %d nsplit %d: %d workgroups, last launch %.2f us (events)\n", n_kv, nsplit, nwg, ms * 1e3);
    std::vector<long long> h((size_t)nwg * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    const char * names[5] = {"entry -> q + first k/v loads issued", "kv loop (incl. wait for the first rows)", "merge in the workgroup + partial store", "arrival (barrier + release atomic)", "merge of the slices + store (last workgroup)"};
    for (int p = 0; p < 5; p++) {
        std::vector<long long> d;
        for (int w = 0; w < nwg; w++) if (h[w * 8 + p + 1] && h[w * 8 + p]) d.push_back(h[w * 8 + p + 1] - h[w * 8 + p]);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("%-48s n %4zu median %7lld  p10 %7lld  p90 %7lld cycles\n", names[p], d.size(), d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    long long t0 = h[7], t1 = 0;
    for (int w = 0; w < nwg; w++) { t0 = std::min(t0, h[w * 8 + 7]); t1 = std::max(t1, h[w * 8 + 6]); }
    std::vector<long long> starts, life;
    for (int w = 0; w < nwg; w++) { starts.push_back(h[w * 8 + 7] - t0); life.push_back(h[w * 8 + 6] - h[w * 8 + 7]); }
    std::sort(starts.begin(), starts.end()); std::sort(life.begin(), life.end());
    printf("first wave start -> last workgroup end %.2f us; workgroup start: median %.2f us max %.2f us; workgroup lifetime median %.2f us max %.2f us\n",
           (t1 - t0) / 100.0, starts[starts.size() / 2] / 100.0, starts.back() / 100.0, life[life.size() / 2] / 100.0, life.back() / 100.0);
    return 0;
}
