// nbody_capi.hip -- implementation of include/nbody_hip.h (the C-ABI drop-in boundary).  gfx950 only.
//
// Thin by design: argument validation, per-precision softening state, launch-plan selection and the
// HIP runtime calls the reference makes through CUDA/thrust/cuda-api-wrappers.  No allocation, no
// synchronisation and no host<->device traffic inside the integrate entry points (graph-capturable).
#include "../../include/nbody_hip.h"
#include "../../include/nbody_hip_tuning.h"

#include "nbody_kernels.h"
#include "rand_stream_guard.h"

#include <algorithm>
#include <atomic>
#include <utility>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

// The reference keeps softening^2 in two __constant__ symbols (bodysystemcuda.cu:43-44); here it is host
// state passed to every launch as a kernel argument (no hipMemcpyToSymbol, nothing to synchronise).
std::atomic<float>  g_softening_sq_f32{0.0f};
std::atomic<double> g_softening_sq_f64{0.0};

std::atomic<int> g_ovr_i{0}, g_ovr_s{0}, g_ovr_tile{0};
std::atomic<int> g_pair_r{0}, g_pair_s{0}, g_pair_c{0}, g_pair_min{0};  // overrides of the pairwise plan (0 = automatic)
std::atomic<void*> g_pair_probe{nullptr};                               // nb_set_pair_probe_event
std::atomic<unsigned long long*> g_clock_words{nullptr};                // nb_set_pair_clock_words
std::atomic<size_t>              g_clock_bytes{0};
std::atomic<int>   g_pair_slices{0};                                    // nb_set_pair_slices_override (0 = automatic)


// The HIP runtime sets parts of itself up lazily, on the first call that needs them (the null stream, the first event,
// the staging path of the first pageable copy), and some of that set-up draws from libc rand().  Do all of it ONCE per
// device, under the guard, so that the launch path (kernel launches, event records, stream waits) can stay lock-free
// without ever being the call that triggers a lazy initialisation.
std::atomic<size_t> g_total_memory[64] = {};  // per device, filled by the one-time warm-up below (0: unknown)
std::atomic<size_t> g_memory_budget{0};       // nb_set_memory_budget: what to assume instead (0: the device's own figure)
static_assert(NB_ERR_OUT_OF_MEMORY == hipErrorOutOfMemory, "the header names the runtime's own value");

int current_device_ready() {
    static std::atomic<int> cu_count[64] = {};
    int                     dev          = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cu_count[dev].load(std::memory_order_acquire);
    if (v == 0) {
        NB_KEEP_RAND_STREAM;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) g_total_memory[dev].store(total_bytes);
        void* scratch = nullptr;
        if (hipMalloc(&scratch, 256) == hipSuccess) {
            const unsigned word  = 0;
            hipEvent_t     event = nullptr;
            (void)hipMemcpy(scratch, &word, sizeof(word), hipMemcpyHostToDevice);  // pageable copy: the known consumer of rand() draws
            (void)hipMemsetAsync(scratch, 0, 256, nullptr);
            if (hipEventCreate(&event) == hipSuccess) {
                (void)hipEventRecord(event, nullptr);
                (void)hipStreamWaitEvent(nullptr, event, 0);
                (void)hipEventSynchronize(event);
                (void)hipEventDestroy(event);
            }
            (void)hipStreamSynchronize(nullptr);
            (void)hipFree(scratch);
        }
        cu_count[dev].store(v, std::memory_order_release);
    }
    return v;
}
inline int cu_count_cached() { return current_device_ready(); }

inline hipStream_t as_stream(nb_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline hipEvent_t  as_event(nb_event_t e) { return reinterpret_cast<hipEvent_t>(e); }

template <typename T> bool aligned_vec4(const void* p) { return (reinterpret_cast<std::uintptr_t>(p) % (4 * sizeof(T))) == 0; }

template <typename T>
int integrate_shard(T* new_pos, const T* old_pos, T* vel, T* acc, unsigned i_begin, unsigned i_count, unsigned j_begin, unsigned j_count, unsigned flags, T dt, T damping, T eps2, int block_size, int mode, nb_stream_t stream) {
    // (no rand()-stream guard here: a kernel launch draws nothing -- tests/test_gpu_parity.py checks it -- and the hot path
    // stays free of the guard's process-wide lock)
    const bool acc_in   = (flags & NB_SHARD_ACC_IN) != 0;
    const bool finalize = (flags & NB_SHARD_FINALIZE) != 0;
    if (!old_pos || i_count == 0) return NB_ERR_INVALID_ARGUMENT;
    if (finalize && (!new_pos || !vel || new_pos == old_pos)) return NB_ERR_INVALID_ARGUMENT;
    if ((acc_in || !finalize) && !acc) return NB_ERR_INVALID_ARGUMENT;
    if (!aligned_vec4<T>(old_pos) || !aligned_vec4<T>(new_pos) || !aligned_vec4<T>(vel) || !aligned_vec4<T>(acc)) return NB_ERR_INVALID_ARGUMENT;
    if (static_cast<unsigned long long>(i_begin) + i_count > 0xFFFFFFFFull || static_cast<unsigned long long>(j_begin) + j_count > 0xFFFFFFFFull) return NB_ERR_INVALID_ARGUMENT;
    {
        // old_pos is read-only for the whole launch (the kernels read it through the scalar cache and the constant address
        // space): nothing the launch WRITES -- bodies [i_begin, i_begin+i_count) of new_pos, vel, acc -- may overlap what it
        // READS of old_pos -- the bodies i and the bodies j.
        const auto span = [](const T* base, unsigned first, unsigned count) {
            const auto lo = reinterpret_cast<std::uintptr_t>(base) + static_cast<std::uintptr_t>(first) * 4 * sizeof(T);
            return std::pair<std::uintptr_t, std::uintptr_t>(lo, lo + static_cast<std::uintptr_t>(count) * 4 * sizeof(T));
        };
        const auto overlaps = [](std::pair<std::uintptr_t, std::uintptr_t> a, std::pair<std::uintptr_t, std::uintptr_t> b) { return a.first < b.second && b.first < a.second; };
        const auto read_i = span(old_pos, i_begin, i_count), read_j = span(old_pos, j_begin, j_count);
        for (const T* written : {static_cast<const T*>(finalize ? new_pos : nullptr), static_cast<const T*>(finalize ? vel : nullptr), static_cast<const T*>(finalize ? nullptr : acc)}) {
            if (written == nullptr) continue;
            const auto w = span(written, i_begin, i_count);
            if (overlaps(w, read_i) || overlaps(w, read_j)) return NB_ERR_INVALID_ARGUMENT;
        }
    }

    nb::Shard<T> s;
    s.new_pos = new_pos, s.old_pos = old_pos, s.vel = vel, s.acc = acc;
    s.i_begin = i_begin, s.i_count = i_count, s.j_begin = j_begin, s.j_count = j_count;
    s.acc_in = acc_in, s.finalize = finalize;
    s.dt = dt, s.damping = damping, s.eps2 = eps2;

    const int cu_count = current_device_ready();
    if (mode == NB_MODE_STRICT) {
        if (block_size <= 0) block_size = 256;  // the reference's default --blockSize (nbody.cpp:285)
        if (block_size % 64 != 0 || block_size > 1024) return NB_ERR_INVALID_ARGUMENT;
        return static_cast<int>(nb::launch_strict<T>(s, block_size, cu_count, as_stream(stream)));
    }
    if (mode == NB_MODE_FAST) {
        const nb::Plan p = nb::plan_fast<T>(i_count, j_count, cu_count, g_ovr_i.load(), g_ovr_s.load(), g_ovr_tile.load());
        return static_cast<int>(nb::launch_fast<T>(s, p, as_stream(stream)));
    }
    return NB_ERR_INVALID_ARGUMENT;
}

struct StepGraph {
    hipGraph_t     graph = nullptr;
    hipGraphExec_t exec  = nullptr;
};

// Fewer bodies than this and the pairwise layout loses to the one-sided kernels (measured: profiles/round3_pair_crossover_*.jsonl
// and the finer sweep in docs/history.md section 5: fp32 8 192 bodies 29.4 against 27.1 us, 10 240 bodies 44.8 against 51.5 us;
// fp64 4 096 bodies 24.3 / 18.9 us, 6 144 bodies 38.5 / 40.8 us; round 4, with the wave-split tiles of 512 bodies: exactly 6 144 bodies
// -- 384 workgroups -- 43.7 against 33.9 us, 6 145 bodies 43.3 / 43.8, 7 000 bodies 44.9 / 48.4, 8 193 bodies 50.3 / 76.4)
template <typename T> constexpr unsigned kPairMinBodies = sizeof(T) == 4 ? 8193u : 6145u;

// Which form of the pairwise layout a system of n bodies takes when `have` bytes of workspace are on offer: the single tournament
// (slices == 1) if its workspace fits, else the tournament cut into the FEWEST slices (2 .. 15) whose workspace fits -- fewer
// slices = fewer, larger launches and less folding.  "Fits" = within `have` and within a third of the device's memory (the
// workspace grows with N^2: 6.4 GB at 1 Mi bodies, 103 GB at 4 Mi in one tournament; 7 GB at 4 Mi in eight slices).
struct PairChoice {
    unsigned        slices = 0;  // 0: the pairwise layout does not apply (the step is the one-sided kernel)
    nb::PairPlan    plan{};      // slices == 1
    nb::PairSlicing sliced{};    // slices >= 2
    size_t          bytes = 0;
};

template <typename T> PairChoice choose_pair_layout(unsigned n, int mode, size_t have) {
    PairChoice c;
    const int  floor_bodies = g_pair_min.load();
    if (mode != NB_MODE_FAST || n == 0 || n < (floor_bodies > 0 ? static_cast<unsigned>(floor_bodies) : kPairMinBodies<T>)) return c;
    const size_t budget = nb::device_memory_budget();
    const size_t limit  = budget == 0 ? have : std::min(have, budget / 3);
    const int    forced = g_pair_slices.load();
    const int    r = g_pair_r.load(), w = g_pair_s.load(), g = g_pair_c.load();
    if (forced <= 1) {
        c.plan = nb::plan_pair<T>(n, cu_count_cached(), r, w, g);
        if (c.plan.workspace_bytes <= limit) {
            c.slices = 1, c.bytes = c.plan.workspace_bytes;
            return c;
        }
        if (forced == 1) return c;
    }
    // The fewest slices that fit -- tried as 2, 4, 8 first: a number of slices that is a power of two steps as fast as one
    // tournament (1 Mi bodies in 2 / 4 / 8 slices: 157.7 / 158.5 / 158.6 ms against 158.3), any other count leaves ragged rounds
    // of workgroups and costs 6-10 % (3 slices of 524 288 bodies 42.9 ms, 4 slices 39.8; profiles/round4_shard_plan_times.txt).
    static constexpr unsigned kOrder[] = {2, 4, 8, 3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15};
    for (unsigned k : kOrder) {
        if (forced > 1 && k != static_cast<unsigned>(forced)) continue;
        const nb::PairSlicing sl = nb::plan_pair_sliced<T>(n, k, r, w, g);
        if (sl.slices < 2 || sl.workspace_bytes > limit) continue;
        c.slices = sl.slices, c.sliced = sl, c.bytes = sl.workspace_bytes;
        return c;
    }
    return c;
}

// nb_integrate_ws_*: the whole system in one step, FAST, with a caller-owned workspace -> the pairwise layout when it
// applies and the workspace is large enough; in every other case exactly what nb_integrate_* does.
template <typename T>
int integrate_ws(T* new_pos, const T* old_pos, T* vel, T dt, T damping, T eps2, unsigned n, int block_size, int mode, void* workspace, size_t workspace_bytes, nb_stream_t stream, bool prepare_only = false) {
    const PairChoice choice = workspace != nullptr ? choose_pair_layout<T>(n, mode, workspace_bytes) : PairChoice{};
    if (choice.slices != 0) {
        if (!new_pos || !old_pos || !vel || new_pos == old_pos) return NB_ERR_INVALID_ARGUMENT;
        if (!aligned_vec4<T>(old_pos) || !aligned_vec4<T>(new_pos) || !aligned_vec4<T>(vel) || (reinterpret_cast<std::uintptr_t>(workspace) % sizeof(T)) != 0) return NB_ERR_INVALID_ARGUMENT;
        {   // nothing the launch writes may overlap the bodies it reads (old_pos is read-only for the whole launch), and the four
            // things it writes -- new positions, velocities, and the workspace in between -- are four separate ranges
            const auto lo = [](const void* q) { return reinterpret_cast<std::uintptr_t>(q); };
            const std::uintptr_t bytes = static_cast<std::uintptr_t>(n) * 4 * sizeof(T);
            auto overlap = [&](const void* a, std::uintptr_t a_len, const void* b, std::uintptr_t b_len) { return lo(a) < lo(b) + b_len && lo(b) < lo(a) + a_len; };
            if (overlap(new_pos, bytes, old_pos, bytes) || overlap(vel, bytes, old_pos, bytes) || overlap(new_pos, bytes, vel, bytes)) return NB_ERR_INVALID_ARGUMENT;
            for (const void* body_array : {static_cast<const void*>(old_pos), static_cast<const void*>(new_pos), static_cast<const void*>(vel)}) {
                if (overlap(workspace, choice.bytes, body_array, bytes)) return NB_ERR_INVALID_ARGUMENT;
            }
        }
        nb::Shard<T> s{};
        s.new_pos = new_pos, s.old_pos = old_pos, s.vel = vel, s.acc = nullptr;
        s.i_begin = 0, s.i_count = n, s.j_begin = 0, s.j_count = n;
        s.acc_in = false, s.finalize = true;
        s.dt = dt, s.damping = damping, s.eps2 = eps2;
        (void)current_device_ready();
        if (choice.slices == 1) return static_cast<int>(nb::launch_pair<T>(s, choice.plan, workspace, as_stream(stream), prepare_only));
        return static_cast<int>(nb::launch_pair_sliced<T>(s, choice.sliced, workspace, as_stream(stream), prepare_only));
    }
    if (prepare_only) return 0;
    return integrate_shard<T>(new_pos, old_pos, vel, nullptr, 0, n, 0, n, NB_SHARD_FINALIZE, dt, damping, eps2, block_size, mode, stream);
}

template <typename T> int graph_create(nb_graph_t* out, T* pos_a, T* pos_b, T* vel, T dt, T damping, T eps2, unsigned n, int block_size, int mode, unsigned steps, void* workspace = nullptr, size_t workspace_bytes = 0) {
    if (!out || !pos_a || !pos_b || !vel || pos_a == pos_b || n == 0 || steps < 2 || (steps & 1u)) return NB_ERR_INVALID_ARGUMENT;
    NB_KEEP_RAND_STREAM;
    *out = nullptr;
    (void)current_device_ready();  // the one-time device warm-up allocates and copies: never inside a capture
    {  // arm the >64 KiB dynamic-LDS attribute outside the capture
        nb::Shard<T> probe{};
        probe.i_count = n, probe.j_count = n;
        hipError_t err = hipSuccess;
        if (mode == NB_MODE_FAST) {
            const nb::Plan p = nb::plan_fast<T>(n, n, cu_count_cached(), g_ovr_i.load(), g_ovr_s.load(), g_ovr_tile.load());
            err              = nb::launch_fast<T>(probe, p, nullptr, /*prepare_only=*/true);
            if (err == hipSuccess && workspace != nullptr) {
                const int rc = integrate_ws<T>(pos_b, pos_a, vel, dt, damping, eps2, n, block_size, mode, workspace, workspace_bytes, nullptr, /*prepare_only=*/true);
                if (rc != 0) return rc;
            }
        } else if (mode == NB_MODE_STRICT) {
            err = nb::launch_strict<T>(probe, block_size, cu_count_cached(), nullptr, /*prepare_only=*/true);
        }
        if (err != hipSuccess) return static_cast<int>(err);
    }
    hipStream_t capture = nullptr;
    auto        err     = hipStreamCreateWithFlags(&capture, hipStreamNonBlocking);
    if (err != hipSuccess) return static_cast<int>(err);
    StepGraph* g  = new StepGraph;
    int        rc = 0;
    err           = hipStreamBeginCapture(capture, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) rc = static_cast<int>(err);
    for (unsigned k = 0; rc == 0 && k < steps; ++k) {
        T* from = (k & 1u) ? pos_b : pos_a;
        T* to   = (k & 1u) ? pos_a : pos_b;
        rc      = workspace != nullptr ? integrate_ws<T>(to, from, vel, dt, damping, eps2, n, block_size, mode, workspace, workspace_bytes, capture)
                                       : integrate_shard<T>(to, from, vel, nullptr, 0, n, 0, n, NB_SHARD_FINALIZE, dt, damping, eps2, block_size, mode, capture);
    }
    if (err == hipSuccess) {  // always close an opened capture
        const auto end = hipStreamEndCapture(capture, &g->graph);
        if (rc == 0 && end != hipSuccess) rc = static_cast<int>(end);
    }
    if (rc == 0) {
        err = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (err != hipSuccess) rc = static_cast<int>(err);
    }
    (void)hipStreamDestroy(capture);
    if (rc != 0) {
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return rc;
    }
    *out = g;
    return 0;
}

template <typename T> int pair_plan_query(unsigned n, nb_pair_plan_t* out) {
    if (out == nullptr || n == 0) return NB_ERR_INVALID_ARGUMENT;
    const PairChoice   chosen = choose_pair_layout<T>(n, NB_MODE_FAST, ~size_t{0});
    const nb::PairPlan p      = nb::plan_pair<T>(n, cu_count_cached(), g_pair_r.load(), g_pair_s.load(), g_pair_c.load());  // the single tournament, whether or not it is affordable
    out->applies          = chosen.slices != 0 ? 1 : 0;
    out->bodies_per_lane  = p.vectors_per_lane * (sizeof(T) == 4 ? 2 : 1);
    out->waves_per_block  = p.waves;
    out->splits           = p.splits;
    out->blocks           = p.blocks;
    out->block_bodies     = p.block_bodies;
    out->reaction_slots   = p.slots;
    out->grid_blocks      = p.grid_blocks;
    out->lds_bytes        = p.lds_bytes;
    out->workspace_bytes  = chosen.slices >= 2 ? chosen.bytes : p.workspace_bytes;
    out->slices           = chosen.slices >= 2 ? chosen.slices : 1u;
    return 0;
}
template <typename T> int plan_query(unsigned i_count, unsigned j_count, nb_launch_plan_t* out) {
    if (!out || i_count == 0) return NB_ERR_INVALID_ARGUMENT;
    const nb::Plan p     = nb::plan_fast<T>(i_count, j_count, cu_count_cached(), g_ovr_i.load(), g_ovr_s.load(), g_ovr_tile.load());
    out->bodies_per_lane = p.bodies_per_lane;
    out->lanes_per_body  = p.lanes_per_body;
    out->tile_bodies     = p.tile_bodies;
    out->block_threads   = p.block_threads;
    out->grid_blocks     = p.grid_blocks;
    out->lds_bytes       = p.lds_bytes;
    return 0;
}

}  // namespace

namespace nb {
size_t device_memory_budget() {
    if (const size_t forced = g_memory_budget.load(); forced != 0) return forced;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    (void)current_device_ready();
    return g_total_memory[dev].load();
}
hipEvent_t pair_probe_event() { return static_cast<hipEvent_t>(g_pair_probe.load(std::memory_order_relaxed)); }
unsigned long long* pair_clock_words(size_t* bytes) {
    *bytes = g_clock_bytes.load(std::memory_order_relaxed);
    return g_clock_words.load(std::memory_order_relaxed);
}
std::atomic<size_t>& alloc_limit() {  // nb_alloc requests above this are made to fail IN THE RUNTIME (0: none); only the lab library has a setter
    static std::atomic<size_t> limit{0};
    return limit;
}
void pair_plan_overrides(int* vectors_per_lane, int* waves, int* splits) {
    *vectors_per_lane = g_pair_r.load(), *waves = g_pair_s.load(), *splits = g_pair_c.load();
}
}  // namespace nb

extern "C" {

const char* nb_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case NB_ERR_INVALID_ARGUMENT: return "NB_ERR_INVALID_ARGUMENT";
        case NB_ERR_UNSUPPORTED: return "NB_ERR_UNSUPPORTED";
        case NB_ERR_RCCL_BASE + 1: return "RCCL: unhandled HIP error";
        case NB_ERR_RCCL_BASE + 2: return "RCCL: system error";
        case NB_ERR_RCCL_BASE + 3: return "RCCL: internal error";
        case NB_ERR_RCCL_BASE + 4: return "RCCL: invalid argument";
        case NB_ERR_RCCL_BASE + 5: return "RCCL: invalid usage";
        case NB_ERR_RCCL_BASE + 6: return "RCCL: remote error";
        default: return hipGetErrorName(static_cast<hipError_t>(code));
    }
}

const char* nb_version(void) { return "mi355x-nbody 0.1 (gfx950)"; }

int nb_device_count(int* count) {
    NB_KEEP_RAND_STREAM;
    if (!count) return NB_ERR_INVALID_ARGUMENT;
    return static_cast<int>(hipGetDeviceCount(count));
}
int nb_set_device(int device) {
    NB_KEEP_RAND_STREAM;
    const auto err = hipSetDevice(device);
    if (err == hipSuccess) (void)current_device_ready();
    return static_cast<int>(err);
}
int nb_get_device(int* device) {
    NB_KEEP_RAND_STREAM;
    if (!device) return NB_ERR_INVALID_ARGUMENT;
    return static_cast<int>(hipGetDevice(device));
}

int nb_device_info(int device, nb_device_info_t* out) {
    NB_KEEP_RAND_STREAM;
    if (!out) return NB_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    const auto      err = hipGetDeviceProperties(&prop, device);
    if (err != hipSuccess) return static_cast<int>(err);
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->name, prop.name, sizeof(out->name) - 1);
    std::strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    out->compute_units       = prop.multiProcessorCount;
    out->wavefront_size      = prop.warpSize;
    out->clock_khz           = prop.clockRate;
    out->can_map_host_memory = prop.canMapHostMemory;
    out->lds_bytes_per_cu    = static_cast<int>(prop.maxSharedMemoryPerMultiProcessor);
    out->total_memory        = prop.totalGlobalMem;
    return 0;
}

int nb_alloc(void** device_ptr, size_t bytes) {
    NB_KEEP_RAND_STREAM;
    if (!device_ptr) return NB_ERR_INVALID_ARGUMENT;
    (void)current_device_ready();
    // (tests of the out-of-memory fall-backs: a request above the limit is turned into one the runtime itself refuses, so that what
    // follows -- the status, the thread's last error -- is exactly what a real refusal leaves behind)
    if (const size_t limit = nb::alloc_limit().load(); limit != 0 && bytes > limit) bytes = ~size_t{0} >> 4;
    const auto err = hipMalloc(device_ptr, bytes);
    // The caller gets the status; the thread's "last error" is cleared, so that the fall-backs built on a refused allocation
    // (halve the workspace and try again, step without one) do not see it again as the status of their next launch.
    if (err != hipSuccess) {
        (void)hipGetLastError();
        *device_ptr = nullptr;
    }
    return static_cast<int>(err);
}
int nb_free(void* device_ptr) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipFree(device_ptr)); }
int nb_memset(void* device_ptr, int value, size_t bytes, nb_stream_t stream) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipMemsetAsync(device_ptr, value, bytes, as_stream(stream))); }

int nb_h2d(void* device_dst, const void* host_src, size_t bytes, nb_stream_t stream) {
    NB_KEEP_RAND_STREAM;
    auto err = hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, as_stream(stream));
    if (err != hipSuccess) return static_cast<int>(err);
    return static_cast<int>(hipStreamSynchronize(as_stream(stream)));
}
int nb_d2h(void* host_dst, const void* device_src, size_t bytes, nb_stream_t stream) {
    NB_KEEP_RAND_STREAM;
    auto err = hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, as_stream(stream));
    if (err != hipSuccess) return static_cast<int>(err);
    return static_cast<int>(hipStreamSynchronize(as_stream(stream)));
}
int nb_d2d(void* device_dst, const void* device_src, size_t bytes, nb_stream_t stream) {
    NB_KEEP_RAND_STREAM;
    return static_cast<int>(hipMemcpyAsync(device_dst, device_src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
}

int nb_host_alloc_mapped(void** host_ptr, void** device_ptr, size_t bytes) {
    NB_KEEP_RAND_STREAM;
    if (!host_ptr || !device_ptr) return NB_ERR_INVALID_ARGUMENT;
    (void)current_device_ready();
    auto err = hipHostMalloc(host_ptr, bytes, hipHostMallocMapped | hipHostMallocPortable);
    if (err != hipSuccess) {
        (void)hipGetLastError();  // (as nb_alloc: the status goes to the caller, not to the next launch)
        return static_cast<int>(err);
    }
    err = hipHostGetDevicePointer(device_ptr, *host_ptr, 0);
    if (err != hipSuccess) {
        (void)hipHostFree(*host_ptr);
        *host_ptr = nullptr;
    }
    return static_cast<int>(err);
}
int nb_host_free(void* host_ptr) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipHostFree(host_ptr)); }

int nb_stream_create(nb_stream_t* stream) {
    NB_KEEP_RAND_STREAM;
    if (!stream) return NB_ERR_INVALID_ARGUMENT;
    (void)current_device_ready();
    hipStream_t s   = nullptr;
    const auto  err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *stream         = s;
    return static_cast<int>(err);
}
int nb_stream_destroy(nb_stream_t stream) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipStreamDestroy(as_stream(stream))); }
int nb_stream_synchronize(nb_stream_t stream) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipStreamSynchronize(as_stream(stream))); }
int nb_stream_wait_event(nb_stream_t stream, nb_event_t event) {
    (void)current_device_ready();
    return static_cast<int>(hipStreamWaitEvent(as_stream(stream), as_event(event), 0)); }

int nb_event_create(nb_event_t* event) {
    NB_KEEP_RAND_STREAM;
    if (!event) return NB_ERR_INVALID_ARGUMENT;
    (void)current_device_ready();
    hipEvent_t e   = nullptr;
    const auto err = hipEventCreate(&e);
    *event         = e;
    return static_cast<int>(err);
}
int nb_event_destroy(nb_event_t event) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipEventDestroy(as_event(event))); }
int nb_event_record(nb_event_t event, nb_stream_t stream) {
    (void)current_device_ready();
    return static_cast<int>(hipEventRecord(as_event(event), as_stream(stream))); }
int nb_event_synchronize(nb_event_t event) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipEventSynchronize(as_event(event))); }
int nb_event_elapsed_ms(float* ms, nb_event_t start, nb_event_t stop) {
    if (!ms) return NB_ERR_INVALID_ARGUMENT;
    return static_cast<int>(hipEventElapsedTime(ms, as_event(start), as_event(stop)));
}
int nb_device_synchronize(void) {
    NB_KEEP_RAND_STREAM; return static_cast<int>(hipDeviceSynchronize()); }

int nb_set_softening_sq_f32(float v) {
    g_softening_sq_f32.store(v);
    return 0;
}
int nb_set_softening_sq_f64(double v) {
    g_softening_sq_f64.store(v);
    return 0;
}
int nb_get_softening_sq_f32(float* v) {
    if (!v) return NB_ERR_INVALID_ARGUMENT;
    *v = g_softening_sq_f32.load();
    return 0;
}
int nb_get_softening_sq_f64(double* v) {
    if (!v) return NB_ERR_INVALID_ARGUMENT;
    *v = g_softening_sq_f64.load();
    return 0;
}

int nb_integrate_shard_f32(float* new_positions, const float* old_positions, float* velocities, float* acc, unsigned i_begin, unsigned i_count, unsigned j_begin, unsigned j_count, unsigned flags, float dt, float damping, int block_size, int mode,
                           nb_stream_t stream) {
    return integrate_shard<float>(new_positions, old_positions, velocities, acc, i_begin, i_count, j_begin, j_count, flags, dt, damping, g_softening_sq_f32.load(), block_size, mode, stream);
}
int nb_integrate_shard_f64(double* new_positions, const double* old_positions, double* velocities, double* acc, unsigned i_begin, unsigned i_count, unsigned j_begin, unsigned j_count, unsigned flags, double dt, double damping, int block_size,
                           int mode, nb_stream_t stream) {
    return integrate_shard<double>(new_positions, old_positions, velocities, acc, i_begin, i_count, j_begin, j_count, flags, dt, damping, g_softening_sq_f64.load(), block_size, mode, stream);
}

int nb_integrate_f32(float* new_positions, const float* old_positions, float* velocities, float dt, float damping, unsigned num_bodies, int block_size, int mode, nb_stream_t stream) {
    return integrate_shard<float>(new_positions, old_positions, velocities, nullptr, 0, num_bodies, 0, num_bodies, NB_SHARD_FINALIZE, dt, damping, g_softening_sq_f32.load(), block_size, mode, stream);
}
int nb_integrate_f64(double* new_positions, const double* old_positions, double* velocities, double dt, double damping, unsigned num_bodies, int block_size, int mode, nb_stream_t stream) {
    return integrate_shard<double>(new_positions, old_positions, velocities, nullptr, 0, num_bodies, 0, num_bodies, NB_SHARD_FINALIZE, dt, damping, g_softening_sq_f64.load(), block_size, mode, stream);
}

int nb_graph_create_f32(nb_graph_t* graph, float* position_a, float* position_b, float* velocities, float dt, float damping, unsigned num_bodies, int block_size, int mode, unsigned steps) {
    return graph_create<float>(graph, position_a, position_b, velocities, dt, damping, g_softening_sq_f32.load(), num_bodies, block_size, mode, steps);
}
int nb_graph_create_f64(nb_graph_t* graph, double* position_a, double* position_b, double* velocities, double dt, double damping, unsigned num_bodies, int block_size, int mode, unsigned steps) {
    return graph_create<double>(graph, position_a, position_b, velocities, dt, damping, g_softening_sq_f64.load(), num_bodies, block_size, mode, steps);
}
int nb_graph_create_ws_f32(nb_graph_t* graph, float* position_a, float* position_b, float* velocities, float dt, float damping, unsigned num_bodies, int block_size, int mode, unsigned steps, void* workspace,
                           size_t workspace_bytes) {
    return graph_create<float>(graph, position_a, position_b, velocities, dt, damping, g_softening_sq_f32.load(), num_bodies, block_size, mode, steps, workspace, workspace_bytes);
}
int nb_graph_create_ws_f64(nb_graph_t* graph, double* position_a, double* position_b, double* velocities, double dt, double damping, unsigned num_bodies, int block_size, int mode, unsigned steps, void* workspace,
                           size_t workspace_bytes) {
    return graph_create<double>(graph, position_a, position_b, velocities, dt, damping, g_softening_sq_f64.load(), num_bodies, block_size, mode, steps, workspace, workspace_bytes);
}

int nb_workspace_bytes_f32(unsigned num_bodies, int mode, size_t* bytes) { return nb_workspace_bytes_capped_f32(num_bodies, mode, ~size_t{0}, bytes); }
int nb_workspace_bytes_f64(unsigned num_bodies, int mode, size_t* bytes) { return nb_workspace_bytes_capped_f64(num_bodies, mode, ~size_t{0}, bytes); }
int nb_workspace_bytes_capped_f32(unsigned num_bodies, int mode, size_t max_bytes, size_t* bytes) {
    if (bytes == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *bytes = choose_pair_layout<float>(num_bodies, mode, max_bytes).bytes;
    return 0;
}
int nb_workspace_bytes_capped_f64(unsigned num_bodies, int mode, size_t max_bytes, size_t* bytes) {
    if (bytes == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *bytes = choose_pair_layout<double>(num_bodies, mode, max_bytes).bytes;
    return 0;
}
int nb_integrate_ws_f32(float* new_positions, const float* old_positions, float* velocities, float dt, float damping, unsigned num_bodies, int block_size, int mode, void* workspace, size_t workspace_bytes, nb_stream_t stream) {
    return integrate_ws<float>(new_positions, old_positions, velocities, dt, damping, g_softening_sq_f32.load(), num_bodies, block_size, mode, workspace, workspace_bytes, stream);
}
int nb_integrate_ws_f64(double* new_positions, const double* old_positions, double* velocities, double dt, double damping, unsigned num_bodies, int block_size, int mode, void* workspace, size_t workspace_bytes,
                        nb_stream_t stream) {
    return integrate_ws<double>(new_positions, old_positions, velocities, dt, damping, g_softening_sq_f64.load(), num_bodies, block_size, mode, workspace, workspace_bytes, stream);
}

int nb_pair_plan_f32(unsigned num_bodies, nb_pair_plan_t* plan) { return pair_plan_query<float>(num_bodies, plan); }
int nb_pair_plan_f64(unsigned num_bodies, nb_pair_plan_t* plan) { return pair_plan_query<double>(num_bodies, plan); }

int nb_set_pair_plan_override(int vectors_per_lane, int waves_per_block, int splits, int min_bodies) {
    auto ok = [](int v, std::initializer_list<int> allowed) {
        for (int a : allowed)
            if (v == a) return true;
        return false;
    };
    if (!ok(vectors_per_lane, {0, 1, 2, 4, 8}) || !ok(waves_per_block, {0, 4, 8, 12, 16}) || splits < 0 || splits > 64 || min_bodies < 0) return NB_ERR_INVALID_ARGUMENT;
    g_pair_r.store(vectors_per_lane);
    g_pair_s.store(waves_per_block);
    g_pair_c.store(splits);
    g_pair_min.store(min_bodies);
    return 0;
}

int nb_set_pair_slices_override(int slices) {
    if (slices < 0 || slices > 15) return NB_ERR_INVALID_ARGUMENT;
    g_pair_slices.store(slices);
    return 0;
}

int nb_set_memory_budget(size_t bytes) {
    g_memory_budget.store(bytes);
    return 0;
}


int nb_set_pair_probe_event(nb_event_t event) {
    g_pair_probe.store(event);
    return 0;
}

int nb_set_pair_clock_words(void* device_words, size_t bytes) {
    if ((device_words == nullptr) != (bytes == 0) || (reinterpret_cast<std::uintptr_t>(device_words) % sizeof(unsigned long long)) != 0) return NB_ERR_INVALID_ARGUMENT;
    g_clock_bytes.store(0);
    g_clock_words.store(static_cast<unsigned long long*>(device_words));
    g_clock_bytes.store(bytes);
    return 0;
}

int nb_graph_launch(nb_graph_t graph, nb_stream_t stream) {
    if (!graph) return NB_ERR_INVALID_ARGUMENT;
    (void)current_device_ready();
    return static_cast<int>(hipGraphLaunch(static_cast<StepGraph*>(graph)->exec, as_stream(stream)));
}
int nb_graph_destroy(nb_graph_t graph) {
    if (!graph) return NB_ERR_INVALID_ARGUMENT;
    NB_KEEP_RAND_STREAM;
    auto* g = static_cast<StepGraph*>(graph);
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

int nb_plan_f32(unsigned i_count, unsigned j_count, nb_launch_plan_t* plan) { return plan_query<float>(i_count, j_count, plan); }
int nb_plan_f64(unsigned i_count, unsigned j_count, nb_launch_plan_t* plan) { return plan_query<double>(i_count, j_count, plan); }

int nb_lds_optin_count(int* count) {
    if (count == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *count = nb::lds_optins.load();
    return 0;
}

int nb_set_plan_override(int bodies_per_lane, int lanes_per_body, int tile_bodies) {
    auto ok = [](int v, std::initializer_list<int> allowed) {
        for (int a : allowed)
            if (v == a) return true;
        return false;
    };
    if (!ok(bodies_per_lane, {0, 1, 2, 4}) || !ok(lanes_per_body, {0, 4, 8, 16, 64}) || !ok(tile_bodies, {0, 256, 512, 1024, 2048})) return NB_ERR_INVALID_ARGUMENT;
    g_ovr_i.store(bodies_per_lane);
    g_ovr_s.store(lanes_per_body);
    g_ovr_tile.store(tile_bodies);
    return 0;
}

}  // extern "C"
