// timing.hip -- see timing.h
#include "timing.h"

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {
std::atomic<int> g_on{0};
std::mutex g_mu;
struct Pair { hipEvent_t a, b; };
std::map<std::string, std::vector<Pair>> g_pending;
struct Acc { int64_t n = 0; double us = 0; };
std::map<std::string, Acc> g_acc;

void drain_locked() {
    for (auto & kv : g_pending) {
        Acc & acc = g_acc[kv.first];
        for (Pair & p : kv.second) {
            if (hipEventSynchronize(p.b) == hipSuccess) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { acc.n++; acc.us += (double)ms * 1e3; }
            }
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        kv.second.clear();
    }
}
} // namespace

namespace mi355 {
bool timing_on() { return g_on.load(std::memory_order_relaxed) != 0; }
void timing_events(const char * tag, hipEvent_t * start, hipEvent_t * stop) {
    Pair p;
    (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending[tag].push_back(p);
    *start = p.a; *stop = p.b;
}
} // namespace mi355

extern "C" {

void mi355_timing_enable(int on) { g_on.store(on ? 1 : 0); }

void mi355_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    g_acc.clear();
}

int mi355_timing_read(const char * tag, int64_t * launches, double * total_us) {
    if (!tag || !launches || !total_us) return MI355_E_NULL;
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    auto it = g_acc.find(tag);
    *launches = it == g_acc.end() ? 0 : it->second.n;
    *total_us = it == g_acc.end() ? 0.0 : it->second.us;
    return MI355_OK;
}

// writes "tag launches total_us\n" lines for every tag seen; returns bytes needed
int64_t mi355_timing_report(char * buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    std::string s;
    for (auto & kv : g_acc) s += kv.first + " " + std::to_string(kv.second.n) + " " + std::to_string(kv.second.us) + "\n";
    if (buf && cap > 0) {
        const int64_t n = (int64_t)s.size() < cap - 1 ? (int64_t)s.size() : cap - 1;
        memcpy(buf, s.data(), (size_t)n); buf[n] = 0;
    }
    return (int64_t)s.size() + 1;
}

} // extern "C"
