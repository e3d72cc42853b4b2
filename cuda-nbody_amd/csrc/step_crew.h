// step_crew.h -- the crew of persistent threads that enqueues the local ranks of a multi-GPU step in parallel (nbody_comm.hip; the why
// is in the comment above its use there).  Plain C++: no HIP, no RCCL -- so that its synchronisation can be run under ThreadSanitizer on
// a host without a GPU (tests/step_crew_tsan.cpp, tests/test_capi_symbols.py::test_step_crew_under_thread_sanitizer).
//
// One job at a time: run(n, fn) calls fn(k) for k = 0 .. n-1 -- k = 0 on the calling thread, k >= 1 on worker k -- and returns when all
// have returned.  A job is announced by a TICKET (an atomic counter, bumped under the mutex so that a worker about to sleep cannot miss
// it); EVERY worker answers every ticket, also one with nothing to do in that job, so a worker that wakes late can never meet the next
// job's description under this job's ticket.  Workers spin for ~0.3 ms after a job (a step's phases and the next step follow within
// microseconds), then sleep on a condition variable.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace nbc {

class StepCrew {
 public:
    // `workers` threads (a crew for workers + 1 ranks); `on_start(k)` runs once on worker k before its first job (nbody_comm.hip: hipSetDevice)
    explicit StepCrew(size_t workers, std::function<void(size_t)> on_start = {}) : on_start_(std::move(on_start)) {
        for (size_t k = 1; k <= workers; ++k) workers_.emplace_back([this, k] { work(k); });
    }
    ~StepCrew() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            quit_ = true;
            ++ticket_;
        }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
    StepCrew(const StepCrew&)            = delete;
    StepCrew& operator=(const StepCrew&) = delete;

    static constexpr int kTooManyRanks = 10001;  // (= NB_ERR_INVALID_ARGUMENT)
    // fn(k) for k = 0 .. n-1, k = 0 on the calling thread, the others on the crew; returns the first non-zero result
    int run(size_t n, const std::function<int(size_t)>& fn) {
        if (n > workers_.size() + 1) return kTooManyRanks;
        results_.assign(n, 0);
        job_ = &fn, job_size_ = n;
        // EVERY thread of the crew answers every ticket, also one with nothing to do in this job: a thread that woke late must not meet
        // the NEXT job's description under the ticket of this one
        pending_.store(static_cast<int>(workers_.size()), std::memory_order_release);
        {
            std::lock_guard<std::mutex> lock(mutex_);  // (the ticket changes under the lock: a worker about to sleep cannot miss it)
            ticket_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) != 0) wake_.notify_all();
        results_[0] = fn(0);
        for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins)
            if (spins > 2000) std::this_thread::yield();
        for (int r : results_)
            if (r != 0) return r;
        return 0;
    }

 private:
    void work(size_t k) {
        if (on_start_) on_start_(k);
        unsigned long long seen = 0;
        for (;;) {
            const auto idle_since = std::chrono::steady_clock::now();
            while (ticket_.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lock(mutex_);
                    sleepers_.fetch_add(1, std::memory_order_acq_rel);
                    wake_.wait(lock, [&] { return ticket_.load(std::memory_order_acquire) != seen; });
                    sleepers_.fetch_sub(1, std::memory_order_acq_rel);
                }
            }
            seen = ticket_.load(std::memory_order_acquire);
            if (quit_) return;
            if (k < job_size_) results_[k] = (*job_)(k);
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

    std::function<void(size_t)>        on_start_;
    std::vector<std::thread>           workers_;
    std::mutex                         mutex_;
    std::condition_variable            wake_;
    std::atomic<unsigned long long>    ticket_{0};
    std::atomic<int>                   pending_{0}, sleepers_{0};
    const std::function<int(size_t)>*  job_      = nullptr;
    size_t                             job_size_ = 0;
    std::vector<int>                   results_;
    bool                               quit_ = false;
};

}  // namespace nbc
