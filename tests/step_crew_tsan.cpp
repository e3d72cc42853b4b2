// step_crew_tsan.cpp -- TEST of cuda-nbody_amd/csrc/step_crew.h under ThreadSanitizer, on a host without a GPU
// (tests/test_capi_symbols.py::test_step_crew_under_thread_sanitizer builds and runs it: -fsanitize=thread, exit status 0 and no report).
//
// The crew enqueues the local ranks of a multi-GPU step in parallel; what must hold, whatever the timing:
//   * every k of a job runs exactly once, on its own thread, and run() returns only after all of them have;
//   * jobs of DIFFERENT sizes may follow one another (a worker with nothing to do in a job still answers its ticket: it must never run the
//     next job's function under this job's ticket);
//   * what a job writes is visible to the caller after run(), what the caller writes before run() is visible to the job (no data race
//     on the job's description or on plain memory handed through it);
//   * the first non-zero result of a job is reported;
//   * workers that went to sleep (longer pauses than their 0.3 ms of spinning) wake for the next job; destruction joins them.
#include "../cuda-nbody_amd/csrc/step_crew.h"

#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
    const int    jobs    = argc > 1 ? std::atoi(argv[1]) : 20000;
    const size_t workers = 7;
    std::vector<int> started(workers + 1, 0);
    std::vector<long long> plain(workers + 1, 0);  // plain memory written by the jobs, read by the caller: a race here is a bug in the crew
    long long              handed = 0;             // ... and written by the caller, read by the jobs
    std::mt19937           rng(12345);
    {
        nbc::StepCrew crew(workers, [&](size_t k) { started[k] = 1; });
        if (crew.run(workers + 2, [](size_t) { return 0; }) != nbc::StepCrew::kTooManyRanks) return 10;
        for (int j = 0; j < jobs; ++j) {
            const size_t n = 1 + rng() % (workers + 1);
            handed         = j;
            std::vector<std::atomic<int>> ran(n);
            const int fail_at = (j % 97 == 0) ? static_cast<int>(rng() % n) : -1;
            const int rc      = crew.run(n, [&](size_t k) {
                if (handed != j) return 99;  // the caller's write before run() must be visible
                ran[k].fetch_add(1);
                plain[k] += static_cast<long long>(k) + 1;
                return static_cast<int>(k) == fail_at ? 7 : 0;
            });
            for (size_t k = 0; k < n; ++k)
                if (ran[k].load() != 1) {
                    std::fprintf(stderr, "job %d: k = %zu ran %d times\n", j, k, ran[k].load());
                    return 1;
                }
            if (rc != (fail_at >= 0 ? 7 : 0)) {
                std::fprintf(stderr, "job %d: result %d\n", j, rc);
                return 2;
            }
            if (j % 2500 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));  // the crew goes to sleep; the next job must wake it
        }
        long long sum = 0;
        for (long long v : plain) sum += v;  // (read after run(): everything the jobs wrote is visible)
        if (sum <= 0) return 3;
        for (size_t k = 1; k <= workers; ++k)
            if (!started[k]) return 4;  // (every worker ran its on_start; written before its first job, read after many)
    }  // joins
    std::puts("step crew ok");
    return 0;
}
