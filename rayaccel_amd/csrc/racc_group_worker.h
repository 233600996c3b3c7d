// racc_group_worker.h — the persistent per-GPU host thread of a device group (racc_group.inc): a job queue with in-order execution,
// drain() and first-failure collection.  A header of its own so that tests/cpp/group_worker_tsan.cpp can run it under
// -fsanitize=thread without a GPU (`make tsan`); libracc_hip.so includes it through racc_group.inc.
#ifndef RACC_GROUP_WORKER_H
#define RACC_GROUP_WORKER_H

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "racc_hip.h"

namespace {
struct GroupWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t posted = 0, finished = 0;
    bool stop = false;
    int rc = RACC_HIP_OK;              // first failure of a posted job since the last collect()
    std::string msg;
    void run() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;      // stop, and nothing left
            std::function<void()> job = std::move(q.front());
            q.pop_front();
            lk.unlock();
            job();
            lk.lock();
            ++finished;
            cv.notify_all();
        }
    }
    void post(std::function<void()> job) {
        { std::lock_guard<std::mutex> g(m); q.push_back(std::move(job)); ++posted; }
        cv.notify_all();
    }
    void drain() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return finished == posted; }); }
    void note(int code) {               // called from a job, on the worker thread: keeps the first failure and its thread-local text
        if (code == RACC_HIP_OK) return;
        std::lock_guard<std::mutex> g(m);
        if (rc == RACC_HIP_OK) { rc = code; msg = racc_hip_last_error(); }
    }
    int collect(std::string& text) { std::lock_guard<std::mutex> g(m); const int r = rc; if (r != RACC_HIP_OK) text = msg; rc = RACC_HIP_OK; msg.clear(); return r; }
};
}  // namespace

#endif
