// rand_stream_guard.h -- keeps the caller's libc rand() stream intact across calls into the HIP / RCCL runtimes.
#pragma once

#include <cstdlib>
#include <mutex>

namespace nb {

// The reference's initial conditions are drawn from the process-global libc rand() stream
// (randomise_bodies.cpp:37-43), so a drop-in must not disturb that stream.  The HIP runtime does: its first
// pageable host-to-device copy consumes rand() draws (tools/rand_probe.cpp).  Every entry point that reaches
// the runtime therefore parks the caller's random()/rand() state and lends the runtime a scratch one.
// initstate/setstate swap a process-global pointer, so the swap-call-restore sequence is serialised across threads.
class RandStreamGuard {
 public:
    RandStreamGuard() : lock_(mutex()) {
        State& st = state();
        if (!st.seeded) {
            prev_     = initstate(0x9e3779b9u, st.scratch, sizeof(st.scratch));
            st.seeded = true;
        } else {
            prev_ = setstate(st.scratch);
        }
    }
    ~RandStreamGuard() {
        if (prev_ != nullptr) (void)setstate(prev_);
    }
    RandStreamGuard(const RandStreamGuard&)            = delete;
    RandStreamGuard& operator=(const RandStreamGuard&) = delete;

 private:
    struct State {
        char scratch[128];
        bool seeded = false;
    };
    static State& state() {  // one instance per shared library (inline function, local static), shared by every translation unit
        static State st;
        return st;
    }
    static std::recursive_mutex& mutex() {
        static std::recursive_mutex m;  // recursive: nb_graph_create_* calls the launch path under its own guard
        return m;
    }
    std::lock_guard<std::recursive_mutex> lock_;
    char*                       prev_ = nullptr;
};
#define NB_KEEP_RAND_STREAM ::nb::RandStreamGuard nb_rand_stream_guard_

}  // namespace nb
