// ThreadSanitizer harness for the device group's persistent workers (rayaccel_amd/csrc/racc_group_worker.h), no GPU needed:
// several caller threads post jobs to three workers, drain, post a "wait" job behind them and collect failures — the call pattern of
// racc_hip_group_intersect / _intersect_device / _wait.  Built by `make -C rayaccel_amd/csrc tsan`; tests/test_host_build.py runs it.
#include <atomic>
#include <cstdio>
#include <memory>
#include <thread>
#include <vector>

#include "racc_group_worker.h"

// the one C-ABI symbol the header refers to: a local stand-in, so that this harness links neither libracc_hip.so nor the ROCm runtime
extern "C" const char* racc_hip_last_error(void) { return "job failed (stub)"; }

int main() {
    constexpr int kWorkers = 3, kCallers = 4, kRounds = 400;
    std::vector<std::unique_ptr<GroupWorker>> w;
    std::vector<long long> done(kWorkers, 0);          // each slot is written by its worker thread only, read after drain()
    for (int i = 0; i < kWorkers; ++i) { w.emplace_back(new GroupWorker()); GroupWorker* p = w.back().get(); p->th = std::thread([p] { p->run(); }); }
    std::atomic<int> failuresPosted{0};
    std::mutex groupCall;                              // a group's entry points are called by one host thread at a time per group
    std::vector<std::thread> callers;
    int collected = 0;
    for (int c = 0; c < kCallers; ++c)
        callers.emplace_back([&, c] {
            for (int r = 0; r < kRounds; ++r) {
                std::lock_guard<std::mutex> g(groupCall);
                for (int i = 0; i < kWorkers; ++i) {
                    GroupWorker* p = w[i].get();
                    long long* slot = &done[i];
                    const bool bad = (r % 97 == 13) && i == c % kWorkers;
                    if (bad) ++failuresPosted;
                    p->post([=] { ++*slot; p->note(bad ? RACC_HIP_ERR_DEVICE : RACC_HIP_OK); });
                }
                if (r % 8 == 7) {                      // ≙ racc_hip_group_wait: a job behind everything issued, drain, collect
                    for (int i = 0; i < kWorkers; ++i) { GroupWorker* p = w[i].get(); p->post([=] { p->note(RACC_HIP_OK); }); }
                    for (auto& p : w) p->drain();
                    for (auto& p : w) { std::string t; if (p->collect(t) != RACC_HIP_OK) ++collected; }
                }
            }
        });
    for (auto& t : callers) t.join();
    for (auto& p : w) p->drain();
    for (auto& p : w) { std::string t; if (p->collect(t) != RACC_HIP_OK) ++collected; }
    long long total = 0;
    for (long long d : done) total += d;
    for (auto& p : w) { { std::lock_guard<std::mutex> lk(p->m); p->stop = true; } p->cv.notify_all(); p->th.join(); }
    const long long expect = 1LL * kCallers * kRounds * kWorkers;
    std::printf("jobs %lld of %lld, failures posted %d, collected (first per worker per wait) %d\n", total, expect, failuresPosted.load(), collected);
    return (total == expect && collected > 0 && collected <= failuresPosted.load()) ? 0 : 1;
}
