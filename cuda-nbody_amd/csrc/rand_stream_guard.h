// rand_stream_guard.h -- keeps the caller's libc rand() stream intact across calls into the HIP / RCCL runtimes.
#pragma once

#include <cstdlib>
#include <mutex>

namespace nb {

// The reference's initial conditions are drawn from the process-global libc rand() stream
// (randomise_bodies.cpp:37-43), so a drop-in must not disturb that stream.  The HIP runtime does: its first
// pageable host-to-device copy consumes rand() draws (tools/rand_probe.cpp).  Every entry point that reaches
// the runtime therefore parks the caller's random()/rand() state and lends the runtime a scratch one.
//
// initstate/setstate swap a process-global pointer, so the guard is a COUNT of the calls currently inside the runtime:
// the first one in parks the caller's state, the last one out puts it back.  The mutex covers the swap only, never the
// guarded call itself (round 3: a rank per THREAD may block inside an RCCL call until its peers -- other threads that
// need this guard too -- have arrived; holding a lock across the call would deadlock them).
class RandStreamGuard {
 public:
    RandStreamGuard() {
        State&                      st = state();
        std::lock_guard<std::mutex> lock(st.mutex);
        if (st.inside++ == 0) {
            if (!st.seeded) {
                st.parked = initstate(0x9e3779b9u, st.scratch, sizeof(st.scratch));
                st.seeded = true;
            } else {
                st.parked = setstate(st.scratch);
            }
        }
    }
    ~RandStreamGuard() {
        State&                      st = state();
        std::lock_guard<std::mutex> lock(st.mutex);
        if (--st.inside == 0 && st.parked != nullptr) {
            (void)setstate(st.parked);
            st.parked = nullptr;
        }
    }
    RandStreamGuard(const RandStreamGuard&)            = delete;
    RandStreamGuard& operator=(const RandStreamGuard&) = delete;

 private:
    struct State {
        std::mutex mutex;
        char       scratch[128];
        char*      parked = nullptr;
        int        inside = 0;
        bool       seeded = false;
    };
    static State& state() {  // one instance per shared library (inline function, local static), shared by every translation unit
        static State st;
        return st;
    }
};
#define NB_KEEP_RAND_STREAM ::nb::RandStreamGuard nb_rand_stream_guard_

}  // namespace nb
