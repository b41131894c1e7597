// lab: latency of small synchronous host -> device copies (the 4 input uploads between a decode token's graphs)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    void * dev; CK(hipMalloc(&dev, 1 << 20));
    void * pin; CK(hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault));
    std::vector<char> host(1 << 20, 1);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t sz : {4ul, 4096ul, 16384ul, 65536ul, 524288ul}) {
        auto run = [&](int mode) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 200; i++) {
                if (mode == 0) (void)hipMemcpy(dev, host.data(), sz, hipMemcpyHostToDevice);
                else if (mode == 1) { memcpy(pin, host.data(), sz); (void)hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
                else if (mode == 2) { (void)hipMemcpyAsync(dev, host.data(), sz, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
                else if (mode == 3) { (void)hipMemcpy(host.data(), dev, sz, hipMemcpyDeviceToHost); }
                else { (void)hipMemcpyAsync(pin, dev, sz, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); memcpy(host.data(), pin, sz); }
            }
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
        };
        for (int m = 0; m < 5; m++) run(m);
        printf("%7zu B: h2d hipMemcpy %.1f us | pinned staging + async + sync %.1f us | async pageable + sync %.1f us | d2h hipMemcpy %.1f us | d2h via pinned %.1f us\n",
               sz, run(0), run(1), run(2), run(3), run(4));
    }
    return 0;
}
