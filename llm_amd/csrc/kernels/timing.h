// timing.h -- optional per-launch kernel timing (off by default; mi355_timing_enable).
// When on, every kernel of this library is launched through hipExtLaunchKernelGGL with a start/stop
// event pair, so the elapsed time is the dispatch's own execution time as the hardware stamps it
// (the same clock rocprofv3 --kernel-trace reports), not host or inter-kernel gaps.
// This is the backend's per-op profiler (the reference has test-backend-ops perf only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace mi355 {
bool timing_on();
void timing_events(const char * tag, hipEvent_t * start, hipEvent_t * stop);
}

// launch `kernel<<<grid, block, shmem, stream>>>(args...)`, timed under `tag` when timing is on
#define MI355_LAUNCH(tag, kernel, grid, block, shmem, stream, ...)                                        \
    do {                                                                                                  \
        if (mi355::timing_on()) {                                                                         \
            hipEvent_t ev0_, ev1_;                                                                        \
            mi355::timing_events(tag, &ev0_, &ev1_);                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev0_, ev1_, 0, __VA_ARGS__);        \
        } else {                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                          \
        }                                                                                                 \
    } while (0)
