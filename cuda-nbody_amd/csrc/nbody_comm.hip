// nbody_comm.hip -- multi-GPU body sharding behind the C-ABI (include/nbody_hip.h, section "multi-GPU").  gfx950 only.
//
// New design: the reference is single-GPU (no NCCL / peer copies anywhere in /root/reference).  Rank r of
// G owns the contiguous slice of bodies [r*N/G, (r+1)*N/G): their velocities and their slice of each new position
// array; every array is full-size and indexed by global body id.  The one exchange step of the path is the all-gather of
// the new positions, issued as its G-1 position TILES over RCCL (= xGMI inside a node): in round s = 1..G-1 every rank
// sends its slice to rank r-s and receives the slice of rank r+s -- ncclSend/ncclRecv pairs on the communicator's own
// high-priority stream, an event per tile (a group and an event per round by default since round 5, all rounds of a step in ONE
// RCCL group with NBODY_EXCHANGE_ONE_GROUP=1 / nb_comm_set_exchange_grouping: see exchange_tiles).  Accumulation is additive over j chunks, so a step starts with the chunk
// that is already local (j in the rank's own slice) and then takes the tiles in arrival order, the kernel of tile k waiting
// only for tile k's event: the exchange runs under the force compute of the chunks already there.
// STRICT keeps the CPU path's summation order (ascending j): tiles in rank order, each waiting for its own round,
// bit-identical to one GPU.
//
// FAST with a workspace lent to every rank (round 3): each pair of bodies is evaluated once across the ranks too -- see
// pair_sharded_step below; the position exchange stays as described, a second leg carries reaction sums to their owners.
//
// What round 5 measured with the REAL RCCL on one GPU (profiles/round5_*; the hooks are the lab library's, nbody_hip_lab.h: a self-loop with every
// byte checked, a LOOPBACK rank that steps as rank r of a nominal G-rank communicator) and what it changed here: a group per round
// is the default (exchange_tiles); the tiles no kernel waits for travel as one more group; a rank's diagonal is two launches, the
// second one LAST, with every reaction round enqueued before it (pair_rank_tiles); the second compute stream is probed against the
// caller's for a shared hardware queue (settle_side_stream).  Layout of this file: the RCCL binding, a rank's resources, the
// position exchange, the pairwise plan and step, the one-sided step, then the extern "C" entry points -- the tuning header's at the end.
//
// Process models, one code path: one process per GPU (nb_comm_init_rank; a group of 1 local rank) or one process
// driving several GPUs (nb_comm_init_all; every RCCL round is then one ncclGroup over the local ranks); a thread per GPU works too.
// RCCL is dlopen'ed on first use (librccl.so.1): a single-GPU run never pays for loading it, and inside a torch process
// the copy torch already loaded is the one that gets bound (same SONAME).
#include "../../include/nbody_hip.h"
#include "../../include/nbody_hip_tuning.h"

#include "nbody_comm_internal.h"
#include "rand_stream_guard.h"
#include "step_crew.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace nbc {

Rccl* rccl() {
    static Rccl           lib;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL to bind is the one that belongs to the HIP runtime this process already runs on: a torch process carries
        // its own copies of both (torch/lib), a plain process the ones under /opt/rocm; the two generations do not mix.
        std::string beside_hip;
        Dl_info     info{};
        if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) != 0 && info.dli_fname != nullptr) {
            beside_hip = info.dli_fname;
            const auto slash = beside_hip.rfind('/');
            beside_hip       = slash == std::string::npos ? std::string() : beside_hip.substr(0, slash + 1);
        }
        const std::string a = beside_hip.empty() ? std::string() : beside_hip + "librccl.so.1", b = beside_hip.empty() ? std::string() : beside_hip + "librccl.so";
        const char* override_path = std::getenv("NBODY_RCCL_LIB");
        for (const char* name : {override_path, a.empty() ? nullptr : a.c_str(), b.empty() ? nullptr : b.c_str(), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            if (name == nullptr) continue;
            lib.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib.handle != nullptr) break;
        }
        if (lib.handle == nullptr) return;
        auto sym = [](const char* n) { return dlsym(lib.handle, n); };
        lib.GetVersion     = reinterpret_cast<decltype(lib.GetVersion)>(sym("ncclGetVersion"));
        lib.GetUniqueId    = reinterpret_cast<decltype(lib.GetUniqueId)>(sym("ncclGetUniqueId"));
        lib.CommInitRank   = reinterpret_cast<decltype(lib.CommInitRank)>(sym("ncclCommInitRank"));
        lib.CommInitAll    = reinterpret_cast<decltype(lib.CommInitAll)>(sym("ncclCommInitAll"));
        lib.CommDestroy    = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
        lib.Send           = reinterpret_cast<decltype(lib.Send)>(sym("ncclSend"));
        lib.Recv           = reinterpret_cast<decltype(lib.Recv)>(sym("ncclRecv"));
        lib.AllGather      = reinterpret_cast<decltype(lib.AllGather)>(sym("ncclAllGather"));
        lib.GroupStart     = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
        lib.GroupEnd       = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
        if (!lib.GetUniqueId || !lib.CommInitRank || !lib.CommInitAll || !lib.CommDestroy || !lib.Send || !lib.Recv || !lib.AllGather || !lib.GroupStart || !lib.GroupEnd) {
            dlclose(lib.handle);
            lib.handle = nullptr;
            return;
        }
        Dl_info bound{};
        if (dladdr(reinterpret_cast<const void*>(lib.Send), &bound) != 0 && bound.dli_fname != nullptr) lib.path = bound.dli_fname;
    });
    return lib.handle != nullptr ? &lib : nullptr;
}

bool default_one_group() {  // NBODY_EXCHANGE_ONE_GROUP=1 flips the default of nb_comm_set_exchange_grouping (a group per round since round 5)
    const char* v = std::getenv("NBODY_EXCHANGE_ONE_GROUP");
    return v != nullptr && v[0] == '1';
}

// ---- the second compute stream of a pairwise step (every other rectangle runs there, next to the caller's stream) ----------------
// The HIP runtime maps the streams of one priority onto a small pool of hardware queues (four by default, GPU_MAX_HW_QUEUES) and
// lets a fifth stream SHARE a queue; two streams on one queue run their kernels strictly one after the other.  RCCL brings
// streams of its own, so a stream made after a communicator is up can land on the caller's queue -- measured (round 5, gpurun
// calls r5k / r5l): one rank's kernels of an 8-rank step 2.31 ms instead of 1.25, every kernel starting exactly where the other
// stream's ended.  Another priority is no way out: the pools are per priority, but a queue of lower OR higher priority does not
// run side by side with the caller's (the same step 1.77 ms either way); a CU-masked stream is one more hardware queue each.  So
// the stream is made at the caller's priority and PROBED the first time the two meet -- against the caller's stream, and against
// the NULL stream (RCCL works there: a second stream on the null stream's queue costs 25 %, a caller on it 40 %: see
// note_stream below): two ~40 us spin kernels, one on each, started together; had they run one after the other, another
// stream is made (the collided ones are kept until the communicator goes, so that the pool moves on) -- up to eight times.
__global__ void queue_probe_spin(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, whatever the shader clock
    unsigned                 x  = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0x9e3779b9u) *sink = x;
}

// one candidate for a compute stream of the library's own: non-blocking, at the caller's (normal) priority.  (A CU-masked stream -- a
// hardware queue of its own -- was measured too: the same placement lottery, and every one of them is one more hardware queue, the
// chip oversubscribed from the twelfth on; profiles/round5_hw_queue_collision.txt.)
hipError_t create_side_stream(hipStream_t* stream) { return hipStreamCreateWithFlags(stream, hipStreamNonBlocking); }

// do kernels on `a` and `b` overlap?  (both streams are synchronised first: a one-off cost, the first time a pair of streams meets)
bool streams_run_side_by_side(hipStream_t a, hipStream_t b) {
    static unsigned*  sink[64] = {};
    static std::mutex guard;  // (a thread per rank: several ranks of one device may meet here)
    int               dev      = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    {
        std::lock_guard<std::mutex> lock(guard);
        if (sink[dev] == nullptr && hipMalloc(reinterpret_cast<void**>(&sink[dev]), sizeof(unsigned)) != hipSuccess) {
            (void)hipGetLastError();
            return true;  // (cannot tell: keep what there is)
        }
    }
    hipEvent_t begin = nullptr, end_a = nullptr, end_b = nullptr;
    bool       side_by_side = true;
    if (hipEventCreate(&begin) == hipSuccess && hipEventCreate(&end_a) == hipSuccess && hipEventCreate(&end_b) == hipSuccess &&
        hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess) {
        constexpr unsigned long long kTicks = 4000;  // 40 us
        float best = 1e30f;
        for (int attempt = 0; attempt < 3; ++attempt) {  // (the shortest of three: a preempted attempt must not read as a shared queue)
            (void)hipEventRecord(begin, a);
            (void)hipStreamWaitEvent(b, begin, 0);
            hipLaunchKernelGGL(queue_probe_spin, dim3(1), dim3(64), 0, a, kTicks, sink[dev]);
            hipLaunchKernelGGL(queue_probe_spin, dim3(1), dim3(64), 0, b, kTicks, sink[dev]);
            (void)hipEventRecord(end_a, a);
            (void)hipEventRecord(end_b, b);
            float ms_a = 0, ms_b = 0;
            if (hipEventSynchronize(end_a) != hipSuccess || hipEventSynchronize(end_b) != hipSuccess || hipEventElapsedTime(&ms_a, begin, end_a) != hipSuccess ||
                hipEventElapsedTime(&ms_b, begin, end_b) != hipSuccess) {
                best = 0;
                break;
            }
            best = std::min(best, std::max(ms_a, ms_b));
        }
        side_by_side = best < 0.070f;  // two 40 us kernels one after the other take 80 us and more
    }
    (void)hipGetLastError();
    for (hipEvent_t e : {begin, end_a, end_b})
        if (e != nullptr) (void)hipEventDestroy(e);
    return side_by_side;
}

bool stream_is_capturing(hipStream_t s) {
    if (s == nullptr) return false;  // (the null stream cannot capture)
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &status) != hipSuccess) {
        (void)hipGetLastError();
        return true;  // (cannot tell: behave as if it were -- no probe, no synchronisation)
    }
    return status != hipStreamCaptureStatusNone;
}

// NBODY_AUX_PROBE=0 switches every probe of this file off (for A/B timings of the collisions themselves)
bool probing_enabled() {
    static const bool probing = [] {
        const char* v = std::getenv("NBODY_AUX_PROBE");
        return v == nullptr || v[0] != '0';
    }();
    return probing;
}

// `side` runs next to `beside` from now on: make sure it can (see above).  `retired`: streams that collided, destroyed with their owner.
hipError_t settle_side_stream(hipStream_t* side, hipStream_t beside, std::vector<hipStream_t>* retired, int* collisions) {
    if (*side == nullptr) {
        if (const auto err = create_side_stream(side); err != hipSuccess) return err;
    }
    // (the null stream too: RCCL puts work of its own there, and a stream on the null stream's hardware queue waits behind it)
    auto fits = [&](hipStream_t candidate) { return streams_run_side_by_side(beside, candidate) && (beside == nullptr || streams_run_side_by_side(nullptr, candidate)); };
    for (int attempt = 0; probing_enabled() && attempt < 8 && !fits(*side); ++attempt) {
        if (collisions != nullptr) ++*collisions;
        retired->push_back(*side);
        *side = nullptr;
        if (const auto err = create_side_stream(side); err != hipSuccess) return err;
    }
    return hipSuccess;
}

// The stream a rank's kernels run on is the caller's; WHICH stream that is matters more than it should.  Measured with the real RCCL
// next to the kernels that ship (round 5, profiles/round5_hw_queue_collision.txt): a rank that computes on the NULL stream -- or on
// a stream that shares the null stream's hardware queue: one created stream in three or four -- steps in 1.80 ms instead of 1.29
// (8 ranks, 262 144 bodies).  RCCL puts work of its own on the null stream, and whatever shares that queue waits behind it.
// Computing on a stream of the communicator's own, ordered after / before the caller's with events, was built and measured: no
// help (1.75 ms) -- the events themselves sit on the bad queue.  So the library only LOOKS and says so
// (nb_comm_caller_stream_placement; bench.py's ranks_seen), and nb_comm_stream_create hands out a stream that is well placed: what
// bench.py and BodySystemHIPSharded step on.
//
// Round 6 (advisor): the look is taken ONCE PER STREAM -- a rank keeps a note per stream it has stepped on (a handful: a caller
// that alternates two streams is probed twice in all, not twice per step) --, never while the stream is capturing (the probe
// synchronises streams), never for a stream nb_comm_stream_create handed out (probed when it was made), and not at all with
// NBODY_AUX_PROBE=0.  `need_aux`: the step is about to use the second compute stream beside this one.  A verdict is a timing
// (two 40 us kernels side by side or not) and steers speed only, never results.  A stream handle the runtime recycles after
// hipStreamDestroy inherits its predecessor's note; nb_comm_settle_side_stream takes a fresh look.
StreamNote* find_note(Comm* c, hipStream_t s) {
    for (StreamNote& n : c->seen)
        if (n.stream == s) return &n;
    return nullptr;
}
void note_stream(Comm* c, hipStream_t caller, bool need_aux, bool fresh_look = false) {
    if (c->world < 2) return;
    c->last_caller   = caller;
    StreamNote* note = find_note(c, caller);
    if (note == nullptr) {
        if (c->seen.size() >= 8) c->seen.erase(c->seen.begin());
        c->seen.push_back(StreamNote{});
        note         = &c->seen.back();
        note->stream = caller;
    } else if (fresh_look) {
        note->placement = -1, note->aux_beside = false;
    }
    const bool may_probe = probing_enabled() && !stream_is_capturing(caller);
    if (note->placement < 0) {
        const bool handed_out = std::find(c->placed.begin(), c->placed.end(), caller) != c->placed.end();
        if (caller == nullptr) note->placement = 1;
        else if (handed_out && !fresh_look) note->placement = 0;
        else if (may_probe) note->placement = streams_run_side_by_side(nullptr, caller) ? 0 : 1;
    }
    if (need_aux && !note->aux_beside && may_probe) {
        hipStream_t before = c->aux;
        if (settle_side_stream(&c->aux, caller, &c->aux_retired, &c->aux_collisions) != hipSuccess) {
            (void)hipGetLastError();
            return;  // (the step goes on with the stream there is)
        }
        if (c->aux != before)
            for (StreamNote& n : c->seen) n.aux_beside = false;  // (a new second stream has met none of the others)
        note             = find_note(c, caller);
        note->aux_beside = true;
        c->aux_probed    = true;
    }
}

int make_resources(Comm* c) {
    DeviceScope scope(c->device);
    int         lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // "greatest" priority is the numerically lowest
    auto err = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    if (err != hipSuccess) return static_cast<int>(err);
    err = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (err != hipSuccess) return static_cast<int>(err);
    err = create_side_stream(&c->aux);  // (probed against a caller's stream the first time the two meet: note_stream)
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->aux_begin, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->aux_done, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&c->cut_ready, hipEventDisableTiming);
    if (err != hipSuccess) return static_cast<int>(err);
    c->one_group = default_one_group();
    if (c->world > 1) {
        err = hipMalloc(reinterpret_cast<void**>(&c->notes), static_cast<size_t>(c->world) * kNoteWords * sizeof(unsigned long long));
        if (err != hipSuccess) return static_cast<int>(err);
    }
    c->arrived.assign(static_cast<size_t>(c->world), nullptr);
    c->react_ready.assign(static_cast<size_t>(c->world / 2 + 1), nullptr);
    c->react_arrived.assign(static_cast<size_t>(c->world / 2 + 1), nullptr);
    for (auto* events : {&c->arrived, &c->react_ready, &c->react_arrived}) {
        for (auto& e : *events) {
            err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (err != hipSuccess) return static_cast<int>(err);
        }
    }
    return 0;
}

void free_resources(Comm* c) {
    DeviceScope scope(c->device);
    for (auto* events : {&c->arrived, &c->react_ready, &c->react_arrived})
        for (auto e : *events)
            if (e) (void)hipEventDestroy(e);
    if (c->notes) (void)hipFree(c->notes);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->aux_begin) (void)hipEventDestroy(c->aux_begin);
    if (c->aux_done) (void)hipEventDestroy(c->aux_done);
    if (c->cut_ready) (void)hipEventDestroy(c->cut_ready);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    for (hipStream_t s : c->aux_retired) (void)hipStreamDestroy(s);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
}

// `locals` is a permutation of the local ranks of one communicator (and nothing else)
bool same_group(const std::vector<Comm*>& locals) {
    const std::vector<Comm*>& group = locals.front()->group;
    if (locals.size() != group.size()) return false;
    for (size_t k = 0; k < locals.size(); ++k) {
        if (locals[k]->world != locals.front()->world) return false;
        bool member = false;
        for (Comm* g : group) member = member || g == locals[k];
        if (!member) return false;
        for (size_t m = 0; m < k; ++m)
            if (locals[m] == locals[k]) return false;
    }
    return true;
}

// the local ranks in ascending rank order (indices into `locals`)
std::vector<size_t> by_rank(const std::vector<Comm*>& locals) {
    std::vector<size_t> order(locals.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return locals[a]->rank < locals[b]->rank; });
    return order;
}

// The G-1 rounds for every local rank of a communicator.  `bytes_per_body` = 4 * sizeof(T).
//
// Inside a round's RCCL group the calls are issued in ONE canonical order: every local rank's send, in ascending rank order, then
// every local rank's receive, in ascending order of the rank the data comes FROM.  Between different peers the order inside a
// group means nothing to RCCL; between the same two peers it matches sends and receives first in, first out -- so the canonical
// order is also what carries a world whose ranks share one RCCL communicator (the lab's in-process world: every transfer a
// self-transfer, the k-th send meeting the k-th receive) without a line of its own here.
int exchange_tiles(const std::vector<Comm*>& locals, void* const* positions, unsigned num_bodies, size_t bytes_per_body, int nccl_type, const hipStream_t* after, int waited_for = 0) {
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    NB_KEEP_RAND_STREAM;  // RCCL calls below; a world of one never gets here
    const int G = locals.front()->world;
    if (num_bodies % static_cast<unsigned>(G)) return NB_ERR_INVALID_ARGUMENT;
    const size_t slice_bodies = num_bodies / static_cast<unsigned>(G);
    const size_t slice_values = slice_bodies * 4;
    for (size_t k = 0; k < locals.size(); ++k) {  // the exchange may only start once the producers of the own slice have run
        Comm*       c = locals[k];
        DeviceScope scope(c->device);
        auto        err = hipEventRecord(c->ready, after[k]);
        if (err == hipSuccess) err = hipStreamWaitEvent(c->stream, c->ready, 0);
        if (err != hipSuccess) return static_cast<int>(err);
        c->in_flight = positions[k];
    }
    // A group -- one RCCL kernel, one event -- PER ROUND by default (round 5, measured with the real RCCL on one GPU, a loopback rank:
    // profiles/round5_exchange_contention.jsonl, round5_exchange_timeline.txt): tile k's event fires with round k, so the kernels that
    // need the first tiles start ~30 us earlier than when one kernel moves all G-1 tiles before any event fires (8 ranks, 262 144 bodies:
    // 1.290 against 1.357 ms per step; never slower at 2 / 4 ranks or 1 Mi bodies).  An RCCL kernel that becomes ready while the force
    // kernels hold every CU does wait for one of them to retire -- a pair_forces<float,8,8> workgroup owns its CU's register file -- but
    // with a group per round those waits fall on tiles that are not needed yet.  nb_comm_set_exchange_grouping(comm, 1) issues all
    // rounds of a step as ONE group instead: a per-communicator setting, so that one job can time both (bench.py's diagnostics do).
    // Same data, same bits either way (tested with the transport double).  PROVISIONAL: both defaults (a group per round; the tail
    // group below) were chosen on one GPU, where a transfer has no link latency and no peer to be late; the first run on a node
    // times the alternatives (bench.py's diagnostics) and may flip them.
    // `waited_for` (the pairwise step passes G/2): only the tiles of the first `waited_for` rounds are waited for by a kernel of
    // the next step -- a rank's rectangles run against ranks r+1 .. r+G/2 --, the others complete the position array for the caller:
    // those travel as ONE more group (a group costs the host and the chip a kernel launch: 8 ranks 11 -> 9 RCCL launches per step).
    const bool one_group = locals.front()->one_group;
    for (Comm* c : locals)
        if (c->one_group != one_group) return NB_ERR_INVALID_ARGUMENT;  // the local ranks of a group must agree (all ranks must)
    const std::vector<size_t> order = by_rank(locals);
    std::vector<size_t>       by_source(order);
    int first = 1;
    while (first < G) {
        const int last = one_group ? G - 1 : ((waited_for > 0 && first > waited_for) ? G - 1 : first);  // rounds first .. last form one group
        int       rc   = lib->GroupStart();
        for (int s = first; s <= last && rc == 0; ++s) {
            for (size_t k : order) {
                Comm*     c    = locals[k];
                const int dst  = (c->rank - s + G) % G;
                char*     base = static_cast<char*>(positions[k]);
                rc             = lib->Send(base + static_cast<size_t>(c->rank) * slice_bodies * bytes_per_body, slice_values, nccl_type, peer_of(c, dst), c->nccl, c->stream);
                if (rc != 0) break;
            }
            std::sort(by_source.begin(), by_source.end(), [&](size_t a, size_t b) { return (locals[a]->rank + s) % G < (locals[b]->rank + s) % G; });
            for (size_t k : by_source) {
                if (rc != 0) break;
                Comm*     c    = locals[k];
                const int src  = (c->rank + s) % G;
                char*     base = static_cast<char*>(positions[k]);
                rc             = lib->Recv(base + static_cast<size_t>(src) * slice_bodies * bytes_per_body, slice_values, nccl_type, peer_of(c, src), c->nccl, c->stream);
            }
        }
        const int end = lib->GroupEnd();
        if (rc == 0) rc = end;
        if (rc != 0) return nccl_status(rc);
        for (Comm* c : locals) {
            DeviceScope scope(c->device);
            for (int s = first; s <= last; ++s) {
                const auto err = hipEventRecord(c->arrived[static_cast<size_t>((c->rank + s) % G)], c->stream);
                if (err != hipSuccess) return static_cast<int>(err);
            }
        }
        first = last + 1;
    }
    return 0;
}

template <typename T> struct Api;
template <> struct Api<float> {
    static constexpr int nccl_type = ncclFloat32;
    static int whole(float* np, const float* op, float* v, float dt, float damping, unsigned n, int bs, int mode, void* ws, size_t bytes, nb_stream_t s) {
        return nb_integrate_ws_f32(np, op, v, dt, damping, n, bs, mode, ws, bytes, s);
    }
    static int shard(float* np, const float* op, float* v, float* a, unsigned i0, unsigned ni, unsigned j0, unsigned nj, unsigned flags, float dt, float damping, int bs, int mode, nb_stream_t s) {
        return nb_integrate_shard_f32(np, op, v, a, i0, ni, j0, nj, flags, dt, damping, bs, mode, s);
    }
};
template <> struct Api<double> {
    static constexpr int nccl_type = ncclFloat64;
    static int whole(double* np, const double* op, double* v, double dt, double damping, unsigned n, int bs, int mode, void* ws, size_t bytes, nb_stream_t s) {
        return nb_integrate_ws_f64(np, op, v, dt, damping, n, bs, mode, ws, bytes, s);
    }
    static int shard(double* np, const double* op, double* v, double* a, unsigned i0, unsigned ni, unsigned j0, unsigned nj, unsigned flags, double dt, double damping, int bs, int mode, nb_stream_t s) {
        return nb_integrate_shard_f64(np, op, v, a, i0, ni, j0, nj, flags, dt, damping, bs, mode, s);
    }
};

// ---- FAST with a workspace: every PAIR of bodies once, across the ranks too (round 3) -----------------------------------------
// One GPU evaluates each pair of bodies once (nbody_pair.hip).  A shard that evaluates rectangles of the pair matrix
// one-sidedly does twice that arithmetic, so the same tournament runs across the ranks: rank r evaluates
//   * its own slice against itself, pairwise (the diagonal);
//   * its slice against the slices of ranks r+1 .. r+H (H = G/2), pairwise: it keeps the sums of its own bodies and SENDS the
//     reaction sums -- folded to one value per body and component, N/G * 12 B, as little as a position tile -- to the owner;
//   * for an even G the partner at distance G/2 lists the pair too: the two split that rectangle, the lower rank taking
//     (its slice) x (the first half of the partner's blocks), the higher one (the second half of its own blocks) x (the
//     partner's slice), and reaction sums travel both ways;
// and the finish kernel adds a body's own sums, its diagonal reaction slots and the H arrays it RECEIVED, in a fixed order.
// Half the arithmetic per rank, one more exchange leg of the size of the existing one (same stream, same rounds, after the
// tiles' kernels; the position exchange is unchanged, so every rank still ends a step with all positions).
std::atomic<int> g_late_diagonal{1};  // the lab's nb_set_late_diagonal: 0 = the diagonal as ONE launch, first (the order up to round 4), for A/B timings

template <typename T> PairShard plan_pair_shard(unsigned num_bodies, int G, int min_slice, size_t budget) {
    const int  diagonal_mode = g_late_diagonal.load();
    const bool late_diagonal = diagonal_mode != 0;
    constexpr unsigned W = sizeof(T) == 4 ? 2 : 1;
    PairShard          p;
    if (G < 2 || num_bodies % static_cast<unsigned>(G)) return p;
    p.ni = num_bodies / static_cast<unsigned>(G);
    if (p.ni < static_cast<unsigned>(min_slice > 0 ? min_slice : 2048) || G / 2 > nb::kMaxRecv || G / 2 + 4 > nb::kMaxSelfSets) return p;
    int ovr_r = 0, ovr_s = 0, ovr_c = 0;
    nb::pair_plan_overrides(&ovr_r, &ovr_s, &ovr_c);  // (tuning sweeps: tools/pair_rank_probe.py)
    // R = 8 from slices of 32 768 bodies (round 4, one rank's kernels alone on one GPU: 65 536-body slices 2.54 -> 2.45 ms, 32 768-body
    // slices 1.356 -> 1.284 ms together with the workgroup count below; half the blocks, so half the reaction planes and workspace).
    const int R = ovr_r > 0 ? ovr_r : (p.ni >= 32768u ? 8 : ((sizeof(T) == 4 ? p.ni >= 16384 : p.ni >= 8192) ? 4 : 2));
    const int S = ovr_s > 0 ? ovr_s : 8;
    // Workgroups per launch: a launch of eight-wave workgroups costs ceil(grid / 256) rounds whatever the residency (nbody_pair.hip),
    // and from two partners on, two rectangles run at once (they alternate between two streams): 128 workgroups each fill the
    // chip together, 256 each only queue behind one another (32 768-body slices, R = 8: C = 4 1.284 ms, C = 8 1.334 ms).
    const unsigned chip = (static_cast<unsigned>(G) / 2 >= 2) ? 128u : 256u;
    p.block  = 64u * static_cast<unsigned>(R) * W;
    p.blocks = (p.ni + p.block - 1) / p.block;
    p.plane  = (p.ni + 63u) / 64u * 64u;
    p.H      = static_cast<unsigned>(G) / 2;
    p.even   = (G % 2) == 0;
    p.half   = (p.blocks / 2) * p.block;
    p.diag_slots = p.blocks < 2 ? 0u : ((p.blocks & 1u) ? p.blocks / 2 : p.blocks / 2 - 1);
    auto splits = [&](unsigned units) {  // workgroups per block: `chip` workgroups per launch, or whole multiples (nb::splits_to_fill)
        unsigned C = ovr_c > 0 ? static_cast<unsigned>(ovr_c) : nb::splits_to_fill(p.blocks, units, S, chip);
        while (C > 1 && units < C * static_cast<unsigned>(S)) C /= 2;
        return C;
    };
    // The diagonal -- the one piece of a rank's work that needs nothing from another rank and owes nothing to one -- goes out as
    // TWO launches (round 5): the block offsets q = 0 .. q_split-1 first, under which the position tiles arrive, and the rest as
    // the rank's LAST force kernel, under which the last rectangle's reaction sums travel to their owner (the finish kernel of
    // that owner waits for them: with the diagonal whole and first, that hop sat bare at the end of every step).  Whole offsets
    // per launch keep their reaction slots apart; each launch has its own i-side sums.  nb_set_late_diagonal(0) (tuning header): one launch, first.
    {
        const unsigned tiles = static_cast<unsigned>(R) * W, offsets = p.blocks / 2 + 1;
        // ... for slices up to 65 536 bodies: the hop is ~40 us whatever the size, the second launch costs a rank ~0.3 % of its kernel
        // time -- measured (profiles/round5_exchange_contention.jsonl): 32 768-body slices -1.1 %, 65 536 -0.5 %, 131 072 +0.2 ... +0.7 %
        const unsigned first = late_diagonal && offsets >= 2 && p.ni <= 65536u ? (offsets + 1) / 2 : offsets;
        p.early_units = first * tiles, p.late_units = (offsets - first) * tiles;
        // Round 6, nb_set_late_diagonal(2): with two balanced streams one of them always ended on a rectangle, whose fold and send then
        // left bare (~0.04 ms of a 1.28 ms step at 8 ranks).  No deal of WHOLE pieces ends both streams on local work (8 ranks: own =
        // 1.5 + d1, second = 2 + d2 rectangles with d1 + d2 = 0.5 of diagonal: balance forces d2 = 0), so one rectangle is cut: the
        // second stream takes half of the late offsets as ITS last kernel, and gives the same amount of work -- the first cut_tiles
        // tiles of bodies j of its last rectangle -- to the step's own stream.  An offset is `tiles` units per block, a tile of bodies j
        // of a rectangle one unit per block: cut_tiles = offsets moved x tiles.  Even G with the SPLIT rectangle last on the second
        // stream (6 ranks) and odd G (its second stream carries less and ends early anyway) keep round 5's deal.  Built, correct, and
        // measured SLOWER on one GPU (profiles/round6_cut_rectangle_ab.txt: 8 ranks 1.312 against 1.294 ms per step, 4 ranks 2.474 /
        // 2.449 -- two more launches cost the kernels 0.7-0.9 %, and ~40 us pieces do not hide the hop): NOT the default; bench.py's
        // N > 1 diagnostics time it on real links, where the hop is longer than a loopback transfer's.
        const unsigned late_offsets = offsets - first, last_on_second = (p.H & 1u) ? p.H : p.H - 1;
        if (diagonal_mode == 2 && p.even && p.H >= 2 && late_offsets >= 2 && last_on_second != p.H) {  // (an odd world's second stream ends early anyway)
            const unsigned moved = late_offsets / 2;
            p.late_aux_units = moved * tiles, p.late_units -= p.late_aux_units;
            p.cut_round = last_on_second, p.cut_tiles = moved * tiles;
            if (p.cut_tiles * 2 > (p.ni + 63) / 64) p.late_units += p.late_aux_units, p.late_aux_units = 0, p.cut_round = 0, p.cut_tiles = 0;  // (a rectangle too small to cut)
        }
    }
    p.diag      = {R, S, splits(p.early_units)};
    p.diag_late = {R, S, p.late_units != 0 ? splits(p.late_units) : 0u};
    p.rect = {R, S, splits((p.ni + 63) / 64)};
    if (p.cut_round != 0) p.diag_late_aux = {R, S, splits(p.late_aux_units)}, p.rect_cut = {R, S, std::min(splits(p.cut_tiles), p.rect.splits)};
    p.rect_upper = p.rect;
    if (p.even && p.blocks >= 2 && (p.ni + 63) / 64 >= p.rect.splits * 2 * static_cast<unsigned>(S) * 2) p.rect_upper.splits = p.rect.splits * 2;
    {   // The reaction rounds leave in the order their sums become ready, not in the order of the partners: the exchange stream is a
        // FIFO, and a round that waits for a fold late on the second stream would hold up one whose fold finished long before (8
        // ranks: round 4's sums are ready 170 us before round 3's -- the half rectangle runs on the step's own stream before the late
        // diagonal).  Expected completion = the work queued on the rectangle's stream up to and including it, in units of a full
        // rectangle (the step's own stream starts with the early diagonal: a quarter, or a half when the diagonal is one launch; odd
        // partners run on the second stream from two partners on).  A function of G alone: the same order on every rank.
        double at_own = p.late_units != 0 ? 0.25 : 0.5, at_second = 0.0, done[nb::kMaxRecv + 1] = {};
        const double cut = p.cut_round != 0 ? static_cast<double>(p.cut_tiles) / ((p.ni + 63) / 64) : 0.0;  // (the part of the cut rectangle that runs on the step's own stream)
        for (unsigned s = 1; s <= p.H; ++s) {
            const double cost = (p.even && s == p.H) ? 0.5 : 1.0;
            double&      at   = (p.H >= 2 && (s & 1u) != 0) ? at_second : at_own;
            if (s == p.cut_round) at_own += cut, at += cost - cut, done[s] = std::max(at, at_own);
            else at += cost, done[s] = at;
        }
        for (unsigned k = 0; k < p.H; ++k) p.send_order[k] = k + 1;
        std::stable_sort(p.send_order, p.send_order + p.H, [&](unsigned a, unsigned b) { return done[a] < done[b]; });
    }
    const size_t plane3 = 3 * static_cast<size_t>(p.plane);
    p.self_at    = 0;
    p.extra_self_first = p.diag.splits + p.diag_late.splits + (p.H - 1) * p.rect.splits + p.rect_upper.splits;  // (the last rectangle is the one that may be split)
    p.react_d_at = p.self_at + (static_cast<size_t>(p.extra_self_first) + p.diag_late_aux.splits + p.rect_cut.splits) * plane3;
    p.react_r_at = p.react_d_at + p.diag_slots * plane3;
    p.send_at    = p.react_r_at + 2 * p.blocks * plane3;  // (two regions: the rectangles alternate between two streams)
    p.recv_at    = p.send_at + p.H * plane3;
    p.elements   = p.recv_at + p.H * plane3;
    // as on one GPU: a workspace beyond a third of the device's memory is never asked for (it grows with the square of the slice:
    // ~16 GB per rank at 1 Mi bodies over 2 ranks) -- the step is then the one-sided tile schedule.  `budget` is the SMALLEST
    // memory of any rank's device (budget_everywhere): the answer must be the communicator's, not this rank's.
    p.applies = budget == 0 || p.elements * sizeof(T) <= budget / 3;
    return p;
}

std::atomic<int> g_pair_shard_min{0};  // nb_comm_set_pair_min_slice: tests run the pairwise step on small slices

// The kernels ONE rank launches in a pairwise step, up to (not including) the finish kernel: the diagonal, then per partner
// the rectangle and the fold of its reaction sums into the send buffer.  `c` == nullptr: no communicator (nb_emulate_pair_rank_*:
// kernel-time projection of a rank of a G-rank system on one GPU) -- no waits for tiles, no events.
// `part`: kWholeStep = everything; kBeforeSends = up to the folds of the rectangles (what the reaction sends wait for), kAfterSends =
// the late diagonal launch and the join of the second stream -- a communicator enqueues its reaction rounds between the two, so that
// the last of them is under way before the rank's last force kernel.
template <typename T>
int pair_rank_tiles(Comm* c, unsigned r, int G, const PairShard& plan, T* work, T* new_pos, const T* old_pos, T* vel, unsigned num_bodies, T dt, T damping, T eps2, hipStream_t stream, bool waiting, nb::FinishArgs<T>& f,
                    hipStream_t aux, hipEvent_t aux_begin, hipEvent_t aux_done, RankPart part) {
    const unsigned ni     = plan.ni, own = r * ni;
    const size_t   plane3 = 3 * static_cast<size_t>(plan.plane);
    nb::PairArgs<T> a{};
    a.old_pos = old_pos, a.self = work + plan.self_at, a.n = num_bodies, a.eps2 = eps2;
    a.self_origin = own, a.self_plane = plan.plane, a.react_plane = plan.plane;
    // the diagonal: the rank's own slice against itself (its positions are local: nothing to wait for), as two launches (plan_pair_shard)
    a.react = work + plan.react_d_at, a.react_origin = own;
    a.i_begin = a.j_begin = own, a.i_count = a.j_count = ni, a.diag = 1, a.keep = 1;
    auto late_diagonal_and_join = [&]() -> int {
        if (plan.late_units != 0) {
            a.self_first = plan.diag.splits, a.unit_begin = plan.early_units, a.unit_count = plan.late_units;
            if (const auto err = nb::launch_pair_tile<T>(a, plan.diag_late, stream); err != hipSuccess) return static_cast<int>(err);
            if (c != nullptr) c->trace += "forces diagonal-late\n";
            f.self_set[f.n_self++] = {plan.diag.splits, plan.diag_late.splits, 0u, ni};
            if (plan.late_aux_units != 0) {  // ... and the second stream's last kernel is local work too (round 6: plan_pair_shard; without a second stream: here)
                a.self_first = plan.extra_self_first, a.unit_begin = plan.early_units + plan.late_units, a.unit_count = plan.late_aux_units;
                if (const auto err = nb::launch_pair_tile<T>(a, plan.diag_late_aux, aux != nullptr ? aux : stream); err != hipSuccess) return static_cast<int>(err);
                if (c != nullptr) c->trace += "forces diagonal-late second stream\n";
                f.self_set[f.n_self++] = {plan.extra_self_first, plan.diag_late_aux.splits, 0u, ni};
            }
        }
        if (aux != nullptr) {  // the finish kernel (on `stream`) needs the second stream's sums too
            auto err = hipEventRecord(aux_done, aux);
            if (err == hipSuccess) err = hipStreamWaitEvent(stream, aux_done, 0);
            if (err != hipSuccess) return static_cast<int>(err);
        }
        return 0;
    };
    if (part == kAfterSends) return late_diagonal_and_join();

    f = {};
    f.old_pos = old_pos, f.new_pos = new_pos, f.vel = vel;
    f.self = work + plan.self_at, f.react = work + plan.react_d_at, f.recv = work + plan.recv_at, f.extra = nullptr;
    f.origin = own, f.count = ni;
    f.self_plane = f.react_plane = f.recv_plane = plan.plane;
    f.react_slots = plan.diag_slots;
    f.dt = dt, f.damping = damping;
    if (aux != nullptr) {  // the second stream joins in here: after everything the step's own stream has done so far
        auto err = hipEventRecord(aux_begin, stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(aux, aux_begin, 0);
        if (err != hipSuccess) return static_cast<int>(err);
    }
    a.self_first = 0, a.unit_begin = 0, a.unit_count = plan.late_units != 0 ? plan.early_units : 0u;
    if (const auto err = nb::launch_pair_tile<T>(a, plan.diag, stream); err != hipSuccess) return static_cast<int>(err);
    if (c != nullptr) c->trace = plan.late_units != 0 ? "forces diagonal-early\n" : "forces diagonal\n";
    f.self_set[f.n_self++] = {0u, plan.diag.splits, 0u, ni};
    // the rectangles against the partners r+1 .. r+H, each as its positions arrive
    for (unsigned s = 1; s <= plan.H; ++s) {
        const unsigned p    = (r + s) % static_cast<unsigned>(G);
        // odd rectangles on the second stream (own region of reaction planes) once there are at least two: measured on one rank's
        // kernels (tools/pair_rank_probe.py) 8 ranks 1.34 against 1.45 ms, 4 ranks 2.55 / 2.67; with one rectangle it only
        // competes with the diagonal (2 ranks: 5.2 / 5.05)
        const bool     other = aux != nullptr && plan.H >= 2 && (s & 1u) != 0;
        hipStream_t    on   = other ? aux : stream;
        if (c != nullptr && waiting) {
            if (const auto err = hipStreamWaitEvent(on, c->arrived[p], 0); err != hipSuccess) return static_cast<int>(err);
        }
        a.diag = 0, a.keep = 1, a.unit_begin = a.unit_count = 0, a.react = work + plan.react_r_at + (other ? plan.blocks * plane3 : size_t{0});
        a.i_begin = own, a.i_count = ni, a.j_begin = p * ni, a.j_count = ni;
        if (plan.even && s == plan.H) {  // both partners list this pair of ranks: split the rectangle
            if (r < p) a.j_count = plan.half;
            else a.i_begin = own + plan.half, a.i_count = ni - plan.half;
        }
        const unsigned blocks_i = (a.i_count + plan.block - 1) / plan.block;
        unsigned       sent     = 0;  // bodies j of this rectangle whose sums are in the send buffer already (the cut-off part)
        if (s == plan.cut_round && other) {
            // the CUT rectangle: its first cut_tiles tiles of bodies j run on the step's own stream (own region of reaction planes, own
            // i-side planes, own fold into the first part of the send array); the rest follows below on the second stream
            const unsigned part = std::min(plan.cut_tiles * 64u, a.j_count);
            if (c != nullptr && waiting) {
                if (const auto err = hipStreamWaitEvent(stream, c->arrived[p], 0); err != hipSuccess) return static_cast<int>(err);
            }
            nb::PairArgs<T> b = a;
            b.react = work + plan.react_r_at, b.j_count = part, b.react_origin = b.j_begin;
            b.self_first = plan.extra_self_first + plan.diag_late_aux.splits;
            if (const auto err = nb::launch_pair_tile<T>(b, plan.rect_cut, stream); err != hipSuccess) return static_cast<int>(err);
            f.self_set[f.n_self++] = {b.self_first, plan.rect_cut.splits, b.i_begin - own, b.i_count};
            if (const auto err = nb::launch_pair_reduce<T>(b.react, plan.plane, blocks_i, work + plan.send_at + (s - 1) * plane3, plan.plane, part, stream); err != hipSuccess) return static_cast<int>(err);
            if (c != nullptr) {
                if (const auto err = hipEventRecord(c->cut_ready, stream); err != hipSuccess) return static_cast<int>(err);
                c->trace += "forces rectangle " + std::to_string(s) + " cut-off part\nfold " + std::to_string(s) + " cut-off part\n";
            }
            sent = part, a.j_begin += part, a.j_count -= part;
        }
        a.react_origin = a.j_begin;
        a.self_first   = plan.diag.splits + plan.diag_late.splits + (s - 1) * plan.rect.splits;
        const nb::PairGeom& geom = (plan.even && s == plan.H && !(r < p)) ? plan.rect_upper : plan.rect;
        if (const auto err = nb::launch_pair_tile<T>(a, geom, on); err != hipSuccess) return static_cast<int>(err);
        f.self_set[f.n_self++] = {a.self_first, geom.splits, a.i_begin - own, a.i_count};
        if (const auto err = nb::launch_pair_reduce<T>(a.react, plan.plane, blocks_i, work + plan.send_at + (s - 1) * plane3 + sent, plan.plane, a.j_count, on); err != hipSuccess) return static_cast<int>(err);
        if (c != nullptr) {
            if (const auto err = hipEventRecord(c->react_ready[s], on); err != hipSuccess) return static_cast<int>(err);
            c->trace += "forces rectangle " + std::to_string(s) + "\nfold " + std::to_string(s) + "\n";
        }
        // what arrives in round s comes from rank r-s, which covered: all of this slice -- or, splitting the rectangle as the
        // lower rank, only the first half of its blocks
        const unsigned q = (r + static_cast<unsigned>(G) - s) % static_cast<unsigned>(G);
        f.recv_set[f.n_recv++] = {0u, (plan.even && s == plan.H && q < r) ? plan.half : ni};
    }
    if (part == kBeforeSends) return 0;
    // (no sends to place: the diagonal's second launch is simply the last force kernel)
    a.diag = 1, a.keep = 1, a.react = work + plan.react_d_at, a.react_origin = own;
    a.i_begin = a.j_begin = own, a.i_count = a.j_count = ni;
    return late_diagonal_and_join();
}

// Pair evaluations rank r makes in one pairwise step (what pair_rank_tiles launches, counted the way the kernel loops: whole
// 64-body tiles against whole blocks of bodies i, the half-kept offsets q = 0 and q = NB/2 included) and its force launches.
inline void pair_rank_work(const PairShard& plan, unsigned r, int G, unsigned long long* evaluations, int* launches) {
    const unsigned long long block = plan.block;
    unsigned long long       sum   = static_cast<unsigned long long>(plan.blocks) * (plan.blocks / 2 + 1) * block * block;  // the diagonal's tournament
    int                      count = (plan.late_units != 0 ? 2 : 1) + (plan.cut_round != 0 ? 2 : 0);
    for (unsigned s = 1; s <= plan.H; ++s) {
        const unsigned p = (r + s) % static_cast<unsigned>(G);
        unsigned       i_count = plan.ni, j_count = plan.ni;
        if (plan.even && s == plan.H) {
            if (r < p) j_count = plan.half;
            else i_count = plan.ni - plan.half;
        }
        const unsigned long long blocks_i = (i_count + plan.block - 1) / plan.block, tiles_j = (j_count + 63) / 64;
        sum += blocks_i * block * tiles_j * 64;
        ++count;
    }
    *evaluations = sum, *launches = count;
}

// The reaction leg of a pairwise step: round s = send to rank r+s what was summed for its bodies, receive from r-s what it
// summed for ours; one RCCL group per round on the exchange stream, round s waiting for react_ready[s] (the fold of rectangle s)
// and signalling react_arrived[s].  Calls inside a group in the canonical order of exchange_tiles: sends by ascending rank, then
// receives by ascending source rank.
template <typename T> int reaction_exchange(const std::vector<Comm*>& locals, const PairShard& plan) {
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    const int    G      = locals.front()->world;
    const size_t plane3 = 3 * static_cast<size_t>(plan.plane);
    NB_KEEP_RAND_STREAM;
    const std::vector<size_t> order = by_rank(locals);
    std::vector<size_t>       by_source(order);
    for (unsigned k = 0; k < plan.H; ++k) {
        const unsigned s = plan.send_order[k];
        for (Comm* c : locals) {
            DeviceScope scope(c->device);
            auto err = hipStreamWaitEvent(c->stream, c->react_ready[s], 0);
            if (err == hipSuccess && s == plan.cut_round) err = hipStreamWaitEvent(c->stream, c->cut_ready, 0);  // (the part of the cut rectangle folded on the step's own stream)
            if (err != hipSuccess) return static_cast<int>(err);
        }
        const int shift = static_cast<int>(s);
        int       rc    = lib->GroupStart();
        for (size_t m : order) {
            if (rc != 0) break;
            Comm* c = locals[m];
            rc      = lib->Send(static_cast<T*>(c->workspace) + plan.send_at + (s - 1) * plane3, plane3, Api<T>::nccl_type, peer_of(c, (c->rank + shift) % G), c->nccl, c->stream);
        }
        std::sort(by_source.begin(), by_source.end(), [&](size_t a, size_t b) { return (locals[a]->rank - shift + G) % G < (locals[b]->rank - shift + G) % G; });
        for (size_t m : by_source) {
            if (rc != 0) break;
            Comm* c = locals[m];
            rc      = lib->Recv(static_cast<T*>(c->workspace) + plan.recv_at + (s - 1) * plane3, plane3, Api<T>::nccl_type, peer_of(c, (c->rank - shift + G) % G), c->nccl, c->stream);
        }
        const int end = lib->GroupEnd();
        if (rc == 0) rc = end;
        if (rc != 0) return nccl_status(rc);
        for (Comm* c : locals) {
            DeviceScope scope(c->device);
            if (const auto err = hipEventRecord(c->react_arrived[s], c->stream); err != hipSuccess) return static_cast<int>(err);
            c->trace += "send reaction " + std::to_string(s) + "\n";
        }
    }
    return 0;
}

// ---- the crew: who enqueues a step over several local ranks ---------------------------------------------------------------------
// One process driving G devices (nb_comm_init_all: `nbody --numdevices G`, BodySystemHIPSharded -- the process model SURVEY 8(e)
// names) used to enqueue every rank's kernels, events and RCCL calls from the calling thread: 0.13-0.19 ms per rank and step
// (measured, a loopback rank of 8 at 262 144 bodies: profiles/round6_graph_capture_and_host_enqueue.txt; eight ranks in one process:
// 0.7-0.9 ms) against a 1.3 ms step -- host-bound.  Capturing a rank's step into a hipGraph was tried first (same file): RCCL's send/recv groups ARE
// captured and replay (the position exchange alone: 3 us of host time per replay; the one-sided step: 4 us per step at the same
// stream time), but the pairwise step replays in 2.41 ms instead of 1.28 -- the graph runs the two compute streams' branches one
// after the other, the very figure of two streams on one hardware queue -- and two pairwise steps in one capture end in a
// segmentation fault inside hipStreamEndCapture.  So the step stays eager and its per-rank parts are enqueued IN PARALLEL:
// a crew of persistent threads, one per local rank, made the first time a group of several ranks steps and kept until the last
// of its communicators goes.  Two forms (sharded_step_locals): ranks that each own an RCCL communicator -- a real node -- are
// stepped INDEPENDENTLY, a thread taking one rank's whole step including its RCCL groups (RCCL's thread-per-device model: what
// one process per GPU does, in threads); ranks that SHARE a communicator (the lab's in-process world, whose transfers are matched
// by the order of one thread's calls) are stepped in PHASES, the kernels and events of a phase by the crew, the RCCL groups of a
// round -- one group over all local ranks -- by the calling thread between the phases.  The crew spins for up to ~0.3 ms between
// phases and steps, then sleeps on a condition variable.
// NBODY_STEP_THREADS=0: the calling thread does everything (A/B timings; the same calls in the same per-rank order, so the
// same bits -- tested).  The crew itself is csrc/step_crew.h: plain C++, run under ThreadSanitizer on the CPU (tests/step_crew_tsan.cpp).
// no two of these ranks share an RCCL communicator (ranks that do -- the lab's in-process world -- cannot call into it concurrently,
// and their transfers are matched by the order of ONE thread's calls)
bool ranks_own_their_transport(const std::vector<Comm*>& locals) {
    for (const Comm* c : locals)
        if (c->shared_nccl) return false;
    return true;
}

static_assert(StepCrew::kTooManyRanks == NB_ERR_INVALID_ARGUMENT, "step_crew.h spells the code without the header");

bool crew_enabled() {
    static const bool on = [] {
        const char* v = std::getenv("NBODY_STEP_THREADS");
        return v == nullptr || v[0] != '0';
    }();
    return on;
}

// fn(k) for every local rank: on the group's crew when there is more than one, else (or with NBODY_STEP_THREADS=0) in turn
int for_each_rank(const std::vector<Comm*>& locals, const std::function<int(size_t)>& fn) {
    if (locals.size() > 1 && crew_enabled()) {
        Comm* owner = locals.front()->group.front();
        if (!owner->crew) {
            std::vector<int> devices;
            for (const Comm* c : owner->group) devices.push_back(c->device);
            auto crew = std::make_shared<StepCrew>(owner->group.size() - 1, [devices](size_t k) { (void)hipSetDevice(devices[k]); });
            for (Comm* c : owner->group) c->crew = crew;
        }
        return static_cast<StepCrew*>(owner->crew.get())->run(locals.size(), fn);
    }
    for (size_t k = 0; k < locals.size(); ++k)
        if (const int rc = fn(k); rc != 0) return rc;
    return 0;
}

// NBODY_ENQUEUE_TRACE=1: the host time of each phase of a pairwise step, appended to the first local rank's trace as "host <ms> <phase>"
// lines (tools/graph_capture_probe.py --what none prints them): where does the enqueue time of a step go?
class PhaseClock {
 public:
    explicit PhaseClock(const std::vector<Comm*>& locals) : to_(enabled() ? locals.front() : nullptr), at_(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        if (to_ == nullptr) return;
        const auto now = std::chrono::steady_clock::now();
        to_->trace += "host " + std::to_string(std::chrono::duration<double, std::milli>(now - at_).count()) + " " + what + "\n";
        at_ = now;
    }
    static bool enabled() {
        static const bool on = [] {
            const char* v = std::getenv("NBODY_ENQUEUE_TRACE");
            return v != nullptr && v[0] == '1';
        }();
        return on;
    }

 private:
    Comm*                                 to_;
    std::chrono::steady_clock::time_point at_;
};

template <typename T>
int pair_sharded_step(const std::vector<Comm*>& locals, const PairShard& plan, T* const* new_pos, const T* const* old_pos, T* const* vel, unsigned num_bodies, T dt, T damping, T eps2, const nb_stream_t* streams) {
    if (rccl() == nullptr) return NB_ERR_UNSUPPORTED;
    const int G = locals.front()->world;
    std::vector<nb::FinishArgs<T>> finish(locals.size());
    PhaseClock clock(locals);
    // phase 1, every rank at once: the early diagonal, the rectangles (each waiting for its tile) and their folds
    int rc = for_each_rank(locals, [&](size_t k) {
        Comm*       c = locals[k];
        DeviceScope scope(c->device);
        const bool  waiting = c->in_flight == static_cast<const void*>(old_pos[k]);
        note_stream(c, reinterpret_cast<hipStream_t>(streams[k]), /*need_aux=*/plan.H >= 2);  // (the second stream is used from two partners on)
        return pair_rank_tiles<T>(c, static_cast<unsigned>(c->rank), G, plan, static_cast<T*>(c->workspace), new_pos[k], old_pos[k], vel[k], num_bodies, dt, damping, eps2,
                                  reinterpret_cast<hipStream_t>(streams[k]), waiting, finish[k], c->aux, c->aux_begin, c->aux_done, kBeforeSends);
    });
    if (rc != 0) return rc;
    clock.lap("kernels before the sends");
    // every reaction round is enqueued (each waits for the fold of its rectangle) BEFORE the ranks' last force kernel: the late half of the diagonal
    if (rc = reaction_exchange<T>(locals, plan); rc != 0) return rc;
    clock.lap("reaction rounds");
    // phase 2, every rank at once: the late diagonal, the join of the second stream, the waits for what was received, the finish kernel
    rc = for_each_rank(locals, [&](size_t k) {
        Comm*       c = locals[k];
        DeviceScope scope(c->device);
        hipStream_t stream = reinterpret_cast<hipStream_t>(streams[k]);
        if (const int r = pair_rank_tiles<T>(c, static_cast<unsigned>(c->rank), G, plan, static_cast<T*>(c->workspace), new_pos[k], old_pos[k], vel[k], num_bodies, dt, damping, eps2, stream, false, finish[k], c->aux,
                                             c->aux_begin, c->aux_done, kAfterSends);
            r != 0)
            return r;
        for (unsigned s = 1; s <= plan.H; ++s) {
            if (const auto err = hipStreamWaitEvent(stream, c->react_arrived[s], 0); err != hipSuccess) return static_cast<int>(err);
        }
        if (const auto err = nb::launch_pair_finish<T>(finish[k], stream); err != hipSuccess) return static_cast<int>(err);
        c->trace += "finish\n";
        return 0;
    });
    clock.lap("late diagonal and finish");
    return rc;
}

// What every rank of the communicator is known to have been lent: one process driving all ranks sees them all; one process
// per rank knows it from nb_comm_set_workspace's exchange (0 before that: the one-sided schedule).
size_t lent_everywhere(const std::vector<Comm*>& locals) {
    const Comm* first = locals.front();
    if (static_cast<int>(first->group.size()) == first->world) {
        size_t least = ~size_t{0};
        for (const Comm* c : first->group) least = std::min(least, c->workspace != nullptr ? c->workspace_bytes : size_t{0});
        return least;
    }
    size_t least = ~size_t{0};
    for (const Comm* c : locals) least = std::min(least, c->workspace != nullptr ? std::min(c->agreed_bytes, c->workspace_bytes) : size_t{0});
    return least;
}

// The device memory a workspace is held against (a third of it at most): the smallest figure of any rank of the communicator, so
// that every rank reaches the same verdict -- ranks on devices of different sizes, or with different nb_set_memory_budget figures,
// would otherwise disagree about the layout of a step, which is the one thing they must never do.
size_t own_budget(const Comm* c) {
    DeviceScope scope(c->device);
    return nb::device_memory_budget();
}
size_t budget_everywhere(const std::vector<Comm*>& locals) {
    const Comm* first = locals.front();
    size_t      least = ~size_t{0};
    if (static_cast<int>(first->group.size()) == first->world) {
        for (const Comm* c : first->group) least = std::min(least, own_budget(c));
    } else {
        for (const Comm* c : locals) least = std::min(least, c->agreed_budget != 0 ? std::min(c->agreed_budget, own_budget(c)) : own_budget(c));
    }
    return least == ~size_t{0} ? 0 : least;
}

template <typename T> bool step_is_pairwise(const std::vector<Comm*>& locals, unsigned num_bodies, int mode, PairShard* plan_out) {
    const int G = locals.front()->world;
    if (G < 2 || mode != NB_MODE_FAST) return false;
    const PairShard plan = plan_pair_shard<T>(num_bodies, G, g_pair_shard_min.load(), budget_everywhere(locals));
    if (!plan.applies || lent_everywhere(locals) < plan.elements * sizeof(T)) return false;
    if (plan_out != nullptr) *plan_out = plan;
    return true;
}

// One step for every local rank: kernels of the own slice and of each tile as it arrives, integrate, start the next exchange.
template <typename T>
int sharded_step_locals(const std::vector<Comm*>& locals, T* const* new_pos, const T* const* old_pos, T* const* vel, T* const* acc, unsigned num_bodies, T dt, T damping, int block_size, int mode, const nb_stream_t* streams) {
    const int      G       = locals.front()->world;
    const size_t   n_local = locals.size();
    const unsigned ni      = num_bodies / static_cast<unsigned>(G);
    if (n_local > 1 && crew_enabled() && ranks_own_their_transport(locals)) {
        // Every local rank has an RCCL communicator of its own (nb_comm_init_all on G devices): each thread of the crew takes ONE rank's
        // whole step -- kernels, events AND its RCCL groups, a group per rank and round exactly as with one process per GPU (RCCL's
        // thread-per-device model; the sends and receives of the ranks meet inside RCCL).  The host then needs what ONE rank needs
        // (0.13-0.19 ms at 8 ranks and 262 144 bodies), whatever the number of devices.
        return for_each_rank(locals, [&](size_t k) {
            return sharded_step_locals<T>(std::vector<Comm*>{locals[k]}, new_pos + k, old_pos + k, vel + k, acc + k, num_bodies, dt, damping, block_size, mode, streams + k);
        });
    }
    bool           done_pairwise = false;
    {   // every rank of the COMMUNICATOR lent a large enough workspace (decided identically on every rank): pairs once, across the ranks too
        PairShard plan;
        if (step_is_pairwise<T>(locals, num_bodies, mode, &plan)) {
            T eps2 = 0;
            if constexpr (sizeof(T) == 4) {
                float e = 0;
                (void)nb_get_softening_sq_f32(&e);
                eps2 = e;
            } else {
                double e = 0;
                (void)nb_get_softening_sq_f64(&e);
                eps2 = e;
            }
            for (size_t k = 0; k < n_local; ++k) {  // the same argument rules as nb_integrate_ws_*: four separate ranges
                if (new_pos[k] == nullptr || old_pos[k] == nullptr || vel[k] == nullptr || new_pos[k] == old_pos[k]) return NB_ERR_INVALID_ARGUMENT;
                const auto lo = [](const void* q) { return reinterpret_cast<size_t>(q); };
                const size_t body_bytes = static_cast<size_t>(num_bodies) * 4 * sizeof(T), work_bytes = plan.elements * sizeof(T);
                const void*  work       = locals[k]->workspace;
                for (const void* body_array : {static_cast<const void*>(old_pos[k]), static_cast<const void*>(new_pos[k]), static_cast<const void*>(vel[k])}) {
                    if (lo(work) < lo(body_array) + body_bytes && lo(body_array) < lo(work) + work_bytes) return NB_ERR_INVALID_ARGUMENT;
                }
            }
            const int rc = pair_sharded_step<T>(locals, plan, new_pos, old_pos, vel, num_bodies, dt, damping, eps2, streams);
            if (rc != 0) return rc;
            done_pairwise = true;
        }
    }
    if (!done_pairwise) {  // the one-sided tile schedule, every rank at once
        const int rc = for_each_rank(locals, [&](size_t k) -> int {
            Comm*          c = locals[k];
            DeviceScope    scope(c->device);
            hipStream_t    stream  = reinterpret_cast<hipStream_t>(streams[k]);
            const bool     waiting = c->in_flight == static_cast<const void*>(old_pos[k]);  // else: every rank holds the whole array already
            const unsigned i0      = static_cast<unsigned>(c->rank) * ni;
            note_stream(c, stream, /*need_aux=*/false);
            if (G == 1 && c->workspace != nullptr)  // one rank holds every body: the single-GPU step, with its workspace
                return Api<T>::whole(new_pos[k], old_pos[k], vel[k], dt, damping, num_bodies, block_size, mode, c->workspace, c->workspace_bytes, streams[k]);
            for (int t = 0; t < G; ++t) {
                // FAST: own slice, then the tiles in arrival order (rank+1, rank+2, ...); STRICT: ascending rank = ascending j
                const int peer = mode == NB_MODE_STRICT ? t : (c->rank + t) % G;
                if (peer != c->rank && waiting) {
                    const auto err = hipStreamWaitEvent(stream, c->arrived[static_cast<size_t>(peer)], 0);
                    if (err != hipSuccess) return static_cast<int>(err);
                }
                const unsigned flags = (t > 0 ? NB_SHARD_ACC_IN : 0u) | (t == G - 1 ? NB_SHARD_FINALIZE : 0u);
                const int      rc    = Api<T>::shard(new_pos[k], old_pos[k], vel[k], acc[k], i0, ni, static_cast<unsigned>(peer) * ni, ni, flags, dt, damping, block_size, mode, streams[k]);
                if (rc != 0) return rc;
            }
            return 0;
        });
        if (rc != 0) return rc;
    }
    if (G == 1) return 0;
    std::vector<void*>       arrays(n_local);
    std::vector<hipStream_t> after(n_local);
    for (size_t k = 0; k < n_local; ++k) arrays[k] = new_pos[k], after[k] = reinterpret_cast<hipStream_t>(streams[k]);
    PhaseClock clock(locals);
    const int  rc = exchange_tiles(locals, arrays.data(), num_bodies, 4 * sizeof(T), Api<T>::nccl_type, after.data(), done_pairwise ? G / 2 : 0);
    clock.lap("position rounds");
    return rc;
}

template <typename T>
int sharded_step(nb_comm_t const* comms, int n_local, T* const* new_pos, const T* const* old_pos, T* const* vel, T* const* acc, unsigned num_bodies, T dt, T damping, int block_size, int mode, const nb_stream_t* streams) {
    if (comms == nullptr || n_local < 1 || !new_pos || !old_pos || !vel || !acc || !streams) return NB_ERR_INVALID_ARGUMENT;
    const auto         t0 = std::chrono::steady_clock::now();
    std::vector<Comm*> locals(static_cast<size_t>(n_local));
    for (int k = 0; k < n_local; ++k) {
        locals[static_cast<size_t>(k)] = as_comm(comms[k]);
        if (locals[static_cast<size_t>(k)] == nullptr) return NB_ERR_INVALID_ARGUMENT;
    }
    // The comms must be exactly the local ranks of ONE communicator (any order): a round is one RCCL group over all of
    // them, and a group that misses a peer hangs inside RCCL instead of failing.
    if (!same_group(locals)) return NB_ERR_INVALID_ARGUMENT;
    const int G = locals.front()->world;
    if (num_bodies == 0 || num_bodies % static_cast<unsigned>(G)) return NB_ERR_INVALID_ARGUMENT;  // pad with zero-mass bodies (as tipsy.cpp:111-119 does)
    const int rc = sharded_step_locals<T>(locals, new_pos, old_pos, vel, acc, num_bodies, dt, damping, block_size, mode, streams);
    // what the HOST needed to enqueue this step for all its local ranks (nb_comm_last_enqueue_ms): a step whose enqueue takes longer
    // than its kernels is bound by the host
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (Comm* c : locals) c->last_enqueue_ms = ms;
    return rc;
}

template <typename T> int comm_workspace_bytes(nb_comm_t comm, unsigned num_bodies, int mode, size_t* bytes) {
    Comm* c = as_comm(comm);
    if (c == nullptr || bytes == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *bytes = 0;
    if (c->world == 1) return sizeof(T) == 4 ? nb_workspace_bytes_f32(num_bodies, mode, bytes) : nb_workspace_bytes_f64(num_bodies, mode, bytes);
    if (mode != NB_MODE_FAST) return 0;
    const PairShard plan = plan_pair_shard<T>(num_bodies, c->world, g_pair_shard_min.load(), budget_everywhere(std::vector<Comm*>{c}));
    if (plan.applies) *bytes = plan.elements * sizeof(T);
    return 0;
}
// Tuning / projection hook (bench.py --emulate-gpus): the kernels rank `rank` of a `world`-rank pairwise step launches, with
// no communicator and no exchange (what would arrive from other ranks is whatever the workspace holds), on one GPU.
template <typename T> int emulate_pair_rank(T* new_pos, const T* old_pos, T* vel, void* workspace, size_t* workspace_bytes, unsigned num_bodies, int world, int rank, T dt, T damping, T eps2, nb_stream_t stream) {
    if (workspace_bytes == nullptr || world < 2 || rank < 0 || rank >= world) return NB_ERR_INVALID_ARGUMENT;
    const PairShard plan = plan_pair_shard<T>(num_bodies, world, g_pair_shard_min.load(), nb::device_memory_budget());
    if (!plan.applies) return NB_ERR_UNSUPPORTED;
    const size_t need = plan.elements * sizeof(T);
    if (workspace == nullptr || *workspace_bytes < need) {  // a size query
        *workspace_bytes = need;
        return workspace == nullptr ? 0 : NB_ERR_INVALID_ARGUMENT;
    }
    if (!new_pos || !old_pos || !vel || new_pos == old_pos) return NB_ERR_INVALID_ARGUMENT;
    nb::FinishArgs<T> f{};
    hipStream_t       s  = reinterpret_cast<hipStream_t>(stream);
    // the second stream of the step, as a communicator would own it (one per device, created on first use, never destroyed)
    static std::mutex  guard;
    static hipStream_t aux[64]   = {}, beside[64] = {};
    static bool        probed[64] = {};
    static hipEvent_t  begin[64] = {}, done[64] = {};
    static std::vector<hipStream_t> retired;
    int                dev       = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return NB_ERR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lock(guard);
        if (std::getenv("NBODY_PAIR_ONE_STREAM") == nullptr) {
            NB_KEEP_RAND_STREAM;
            if (begin[dev] == nullptr && (hipEventCreateWithFlags(&begin[dev], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done[dev], hipEventDisableTiming) != hipSuccess))
                return NB_ERR_UNSUPPORTED;
            if (aux[dev] == nullptr || (plan.H >= 2 && (!probed[dev] || beside[dev] != s))) {  // (as a communicator does: a second stream that really runs beside this one)
                if (settle_side_stream(&aux[dev], s, &retired, nullptr) != hipSuccess) return NB_ERR_UNSUPPORTED;
                probed[dev] = plan.H >= 2, beside[dev] = s;
            }
        }
    }
    const int rc = pair_rank_tiles<T>(nullptr, static_cast<unsigned>(rank), world, plan, static_cast<T*>(workspace), new_pos, old_pos, vel, num_bodies, dt, damping, eps2, s, false, f, aux[dev], begin[dev], done[dev]);
    if (rc != 0) return rc;
    return static_cast<int>(nb::launch_pair_finish<T>(f, s));
}

// Tuning hook: the reaction leg alone (what a pairwise step adds to the exchange), on whatever the workspace holds.
template <typename T> int reaction_exchange_only(nb_comm_t comm, unsigned num_bodies, nb_stream_t stream) {
    Comm* c = as_comm(comm);
    if (c == nullptr || c->group.size() != 1) return NB_ERR_INVALID_ARGUMENT;
    std::vector<Comm*> locals{c};
    PairShard          plan;
    if (!step_is_pairwise<T>(locals, num_bodies, NB_MODE_FAST, &plan)) return NB_ERR_UNSUPPORTED;
    DeviceScope scope(c->device);
    hipStream_t on = reinterpret_cast<hipStream_t>(stream);
    for (unsigned s = 1; s <= plan.H; ++s) {
        if (const auto err = hipEventRecord(c->react_ready[s], on); err != hipSuccess) return static_cast<int>(err);
    }
    if (const int rc = reaction_exchange<T>(locals, plan); rc != 0) return rc;
    for (unsigned s = 1; s <= plan.H; ++s) {
        if (const auto err = hipStreamWaitEvent(on, c->react_arrived[s], 0); err != hipSuccess) return static_cast<int>(err);
    }
    return 0;
}

template <typename T> int comm_layout(nb_comm_t comm, unsigned num_bodies, int mode, int* pairwise) {
    Comm* c = as_comm(comm);
    if (c == nullptr || pairwise == nullptr || num_bodies == 0) return NB_ERR_INVALID_ARGUMENT;
    if (c->world == 1) {
        // the step hands nb_integrate_ws_* whatever was lent, and that takes the form with the fewest slices that FIT (one tournament
        // or the sliced one): the same question, asked the same way -- not "is there room for one tournament"
        size_t fits = 0;
        if (c->workspace == nullptr || c->workspace_bytes == 0) {
            *pairwise = 0;
            return 0;
        }
        const int rc = sizeof(T) == 4 ? nb_workspace_bytes_capped_f32(num_bodies, mode, c->workspace_bytes, &fits) : nb_workspace_bytes_capped_f64(num_bodies, mode, c->workspace_bytes, &fits);
        *pairwise    = (rc == 0 && fits != 0) ? 1 : 0;
        return rc;
    }
    *pairwise = step_is_pairwise<T>(std::vector<Comm*>{c}, num_bodies, mode, nullptr) ? 1 : 0;
    return 0;
}
template <typename T> int comm_pair_work(nb_comm_t comm, unsigned num_bodies, unsigned long long* evaluations, int* launches) {
    Comm* c = as_comm(comm);
    if (c == nullptr || evaluations == nullptr || launches == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *evaluations = 0, *launches = 0;
    PairShard plan;
    if (!step_is_pairwise<T>(std::vector<Comm*>{c}, num_bodies, NB_MODE_FAST, &plan)) return NB_ERR_UNSUPPORTED;  // (one-sided: N/G x N directed interactions, nothing to ask)
    pair_rank_work(plan, static_cast<unsigned>(c->rank), c->world, evaluations, launches);
    return 0;
}
// `alone_too`: bind RCCL and make a real communicator even for a world of one (the lab's self-loop check);
// `self_peers`: the ncclComm has ONE rank whatever `world` says -- rank r of a nominal world whose every peer is itself (the lab's loopback rank)
int init_rank(nb_comm_t* comm, const void* id, int world, int rank, bool alone_too, bool self_peers) {
    NB_KEEP_RAND_STREAM;
    if (!comm || (!id && (world > 1 || alone_too)) || world < 1 || rank < 0 || rank >= world) return NB_ERR_INVALID_ARGUMENT;
    *comm     = nullptr;
    const bool transport = world > 1 || alone_too;
    Rccl* lib = transport ? rccl() : nullptr;  // a world of one never exchanges anything: no RCCL needed, none loaded
    if (transport && lib == nullptr) return NB_ERR_UNSUPPORTED;
    auto* c  = new Comm;
    c->rank  = rank;
    c->world = world;
    if (const auto err = hipGetDevice(&c->device); err != hipSuccess) {
        delete c;
        return static_cast<int>(err);
    }
    int rc = 0;
    if (transport) {
        ncclUniqueId uid;
        std::memcpy(uid.internal, id, sizeof(uid.internal));
        rc = nccl_status(self_peers ? lib->CommInitRank(&c->nccl, 1, uid, 0) : lib->CommInitRank(&c->nccl, world, uid, rank));
    }
    c->self_peers = self_peers;
    if (rc == 0) rc = make_resources(c);
    if (rc != 0) {
        if (c->nccl) (void)lib->CommDestroy(c->nccl);
        free_resources(c);
        delete c;
        return rc;
    }
    c->group = {c};
    *comm    = c;
    return 0;
}


}  // namespace nbc

using namespace nbc;

extern "C" {

int nb_comm_unique_id(void* id) {
    NB_KEEP_RAND_STREAM;
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    if (id == nullptr) return NB_ERR_INVALID_ARGUMENT;
    ncclUniqueId uid;
    const int    rc = lib->GetUniqueId(&uid);
    if (rc == 0) std::memcpy(id, uid.internal, sizeof(uid.internal));
    return nccl_status(rc);
}

int nb_comm_init_rank(nb_comm_t* comm, const void* id, int world, int rank) { return init_rank(comm, id, world, rank, false); }

int nb_comm_init_all(nb_comm_t* comms, int num_devices, const int* devices) {
    NB_KEEP_RAND_STREAM;
    if (!comms || num_devices < 1) return NB_ERR_INVALID_ARGUMENT;
    Rccl* lib = num_devices > 1 ? rccl() : nullptr;  // one device: nothing to exchange, RCCL stays unloaded
    if (num_devices > 1 && lib == nullptr) return NB_ERR_UNSUPPORTED;
    std::vector<int> devs(static_cast<size_t>(num_devices));
    for (int k = 0; k < num_devices; ++k) devs[static_cast<size_t>(k)] = devices ? devices[k] : k;
    int visible = 0;
    if (const auto err = hipGetDeviceCount(&visible); err != hipSuccess) return static_cast<int>(err);
    for (int d : devs)
        if (d < 0 || d >= visible) return NB_ERR_INVALID_ARGUMENT;
    std::vector<ncclComm_t> raw(static_cast<size_t>(num_devices), nullptr);
    int                     rc = lib != nullptr ? nccl_status(lib->CommInitAll(raw.data(), num_devices, devs.data())) : 0;
    if (rc != 0) return rc;
    std::vector<Comm*> made;
    for (int k = 0; k < num_devices && rc == 0; ++k) {
        auto* c   = new Comm;
        c->nccl   = raw[static_cast<size_t>(k)];
        c->rank   = k;
        c->world  = num_devices;
        c->device = devs[static_cast<size_t>(k)];
        made.push_back(c);
        rc = make_resources(c);
    }
    if (rc != 0) {
        for (size_t k = 0; k < raw.size(); ++k)
            if (lib != nullptr && raw[k] != nullptr) (void)lib->CommDestroy(raw[k]);
        for (Comm* c : made) {
            free_resources(c);
            delete c;
        }
        return rc;
    }
    for (Comm* c : made) c->group = made;
    for (int k = 0; k < num_devices; ++k) comms[k] = made[static_cast<size_t>(k)];
    return 0;
}

int nb_comm_destroy(nb_comm_t comm) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr) return NB_ERR_INVALID_ARGUMENT;
    {
        DeviceScope scope(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    if (c->shared_nccl) c->shared_nccl.reset();  // (ranks that share an ncclComm: the last of them destroys it)
    else if (c->nccl != nullptr)
        if (Rccl* lib = rccl(); lib != nullptr) (void)lib->CommDestroy(c->nccl);
    free_resources(c);
    delete c;
    return 0;
}

int nb_comm_set_workspace(nb_comm_t comm, void* workspace, size_t workspace_bytes) {
    Comm* c = as_comm(comm);
    if (c == nullptr || (workspace == nullptr && workspace_bytes != 0) || (reinterpret_cast<size_t>(workspace) % sizeof(double)) != 0) return NB_ERR_INVALID_ARGUMENT;
    c->workspace       = workspace;
    c->workspace_bytes = workspace_bytes;
    c->agreed_bytes    = workspace_bytes;
    if (c->world == 1 || static_cast<int>(c->group.size()) == c->world) return 0;  // this process sees every rank: nothing to exchange
    // One process per rank: the layout of a step must be the same on every rank of the communicator, so the ranks tell each other
    // what they were lent (and the process-global plan overrides their plans depend on): G-1 send/recv rounds of one small note
    // each on the exchange stream, in one RCCL group; every rank keeps the smallest amount.  A collective: it returns once every
    // rank of the communicator has called it.
    c->agreed_bytes = 0;  // (until the exchange below has succeeded: one-sided)
    Rccl* lib = rccl();
    if (lib == nullptr || c->notes == nullptr) return NB_ERR_UNSUPPORTED;
    NB_KEEP_RAND_STREAM;
    DeviceScope scope(c->device);
    const int   G = c->world;
    int         ovr_r = 0, ovr_s = 0, ovr_c = 0;
    nb::pair_plan_overrides(&ovr_r, &ovr_s, &ovr_c);
    unsigned long long mine[kNoteWords] = {static_cast<unsigned long long>(workspace_bytes), static_cast<unsigned long long>(g_pair_shard_min.load()), static_cast<unsigned long long>(ovr_r),
                                           static_cast<unsigned long long>(ovr_s), static_cast<unsigned long long>(ovr_c), static_cast<unsigned long long>(nb::device_memory_budget()),
                                           static_cast<unsigned long long>(g_late_diagonal.load()), 0};
    auto err = hipMemcpyAsync(c->notes + static_cast<size_t>(c->rank) * kNoteWords, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);  // (`mine` is pageable stack memory: the copy must be over before it goes away)
    if (err != hipSuccess) return static_cast<int>(err);
    int rc = lib->GroupStart();
    for (int s = 1; s < G && rc == 0; ++s) {
        const int dst = (c->rank - s + G) % G, src = (c->rank + s) % G;
        rc              = lib->Send(c->notes + static_cast<size_t>(c->rank) * kNoteWords, kNoteWords * 2, ncclFloat32, peer_of(c, dst), c->nccl, c->stream);  // (8 bytes = two 4-byte values)
        if (rc == 0) rc = lib->Recv(c->notes + static_cast<size_t>(src) * kNoteWords, kNoteWords * 2, ncclFloat32, peer_of(c, src), c->nccl, c->stream);
    }
    const int end = lib->GroupEnd();
    if (rc == 0) rc = end;
    if (rc != 0) return nccl_status(rc);
    std::vector<unsigned long long> all(static_cast<size_t>(G) * kNoteWords);
    err = hipMemcpyAsync(all.data(), c->notes, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
    if (err != hipSuccess) return static_cast<int>(err);
    unsigned long long least = ~0ull, least_budget = ~0ull;
    bool               same  = true;
    for (int r = 0; r < G; ++r) {
        const unsigned long long* note = all.data() + static_cast<size_t>(r) * kNoteWords;
        least = std::min(least, note[0]);
        if (note[5] != 0) least_budget = std::min(least_budget, note[5]);  // (0: that rank knows no figure -- no bound from it)
        for (int w = 1; w < 5; ++w) same = same && note[w] == mine[w];
        same = same && note[6] == mine[6];
    }
    if (!same) return NB_ERR_INVALID_ARGUMENT;  // (every rank sees the same notes, so every rank returns this)
    c->agreed_bytes  = static_cast<size_t>(least);
    c->agreed_budget = least_budget == ~0ull ? 0 : static_cast<size_t>(least_budget);
    return 0;
}

int nb_comm_layout_f32(nb_comm_t comm, unsigned num_bodies, int mode, int* pairwise) { return comm_layout<float>(comm, num_bodies, mode, pairwise); }
int nb_comm_layout_f64(nb_comm_t comm, unsigned num_bodies, int mode, int* pairwise) { return comm_layout<double>(comm, num_bodies, mode, pairwise); }

int nb_comm_set_exchange_grouping(nb_comm_t comm, int one_group) {
    Comm* c = as_comm(comm);
    if (c == nullptr || (one_group != 0 && one_group != 1)) return NB_ERR_INVALID_ARGUMENT;
    c->one_group = one_group != 0;
    return 0;
}
int nb_comm_get_exchange_grouping(nb_comm_t comm, int* one_group) {
    Comm* c = as_comm(comm);
    if (c == nullptr || one_group == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *one_group = c->one_group ? 1 : 0;
    return 0;
}

int nb_comm_workspace_bytes_f32(nb_comm_t comm, unsigned num_bodies, int mode, size_t* bytes) { return comm_workspace_bytes<float>(comm, num_bodies, mode, bytes); }
int nb_comm_workspace_bytes_f64(nb_comm_t comm, unsigned num_bodies, int mode, size_t* bytes) { return comm_workspace_bytes<double>(comm, num_bodies, mode, bytes); }
int nb_emulate_pair_rank_f32(float* new_positions, const float* old_positions, float* velocities, void* workspace, size_t* workspace_bytes, unsigned num_bodies, int world_size, int rank, float dt, float damping,
                             nb_stream_t stream) {
    float eps2 = 0;
    (void)nb_get_softening_sq_f32(&eps2);
    return emulate_pair_rank<float>(new_positions, old_positions, velocities, workspace, workspace_bytes, num_bodies, world_size, rank, dt, damping, eps2, stream);
}
int nb_emulate_pair_rank_f64(double* new_positions, const double* old_positions, double* velocities, void* workspace, size_t* workspace_bytes, unsigned num_bodies, int world_size, int rank, double dt, double damping,
                             nb_stream_t stream) {
    double eps2 = 0;
    (void)nb_get_softening_sq_f64(&eps2);
    return emulate_pair_rank<double>(new_positions, old_positions, velocities, workspace, workspace_bytes, num_bodies, world_size, rank, dt, damping, eps2, stream);
}
int nb_comm_reaction_exchange_f32(nb_comm_t comm, unsigned num_bodies, nb_stream_t stream) { return reaction_exchange_only<float>(comm, num_bodies, stream); }
int nb_comm_reaction_exchange_f64(nb_comm_t comm, unsigned num_bodies, nb_stream_t stream) { return reaction_exchange_only<double>(comm, num_bodies, stream); }
int nb_set_late_diagonal(int on) {
    if (on < 0 || on > 2) return NB_ERR_INVALID_ARGUMENT;
    g_late_diagonal.store(on);
    return 0;
}
int nb_comm_set_pair_min_slice(int min_bodies_per_rank) {
    if (min_bodies_per_rank < 0) return NB_ERR_INVALID_ARGUMENT;
    g_pair_shard_min.store(min_bodies_per_rank);
    return 0;
}

int nb_comm_transport_info(nb_comm_t comm, int* version, char* library_path, size_t path_bytes) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr) return NB_ERR_INVALID_ARGUMENT;
    if (version) *version = 0;
    if (library_path && path_bytes) library_path[0] = 0;
    if (c->nccl == nullptr) return 0;  // a world of one: no transport bound, nothing to report
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    if (version && lib->GetVersion) (void)lib->GetVersion(version);
    if (library_path && path_bytes) {
        std::strncpy(library_path, lib->path.c_str(), path_bytes - 1);
        library_path[path_bytes - 1] = 0;
    }
    return 0;
}

int nb_comm_settle_side_stream(nb_comm_t comm, nb_stream_t beside) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr) return NB_ERR_INVALID_ARGUMENT;
    DeviceScope scope(c->device);
    hipStream_t with = reinterpret_cast<hipStream_t>(beside);
    if (stream_is_capturing(with)) return NB_ERR_INVALID_ARGUMENT;  // (the probe synchronises streams)
    const StreamNote* known = find_note(c, with);
    note_stream(c, with, /*need_aux=*/true, /*fresh_look=*/known != nullptr && known->aux_beside);  // (asked again for a stream already settled: look afresh -- a recycled handle)
    return 0;
}

int nb_comm_caller_stream_placement(nb_comm_t comm, int* badly_placed) {
    Comm* c = as_comm(comm);
    if (c == nullptr || badly_placed == nullptr) return NB_ERR_INVALID_ARGUMENT;
    const StreamNote* note = c->seen.empty() ? nullptr : find_note(c, c->last_caller);
    *badly_placed = note == nullptr ? -1 : note->placement;
    return 0;
}

int nb_comm_last_enqueue_ms(nb_comm_t comm, double* milliseconds) {
    Comm* c = as_comm(comm);
    if (c == nullptr || milliseconds == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *milliseconds = c->last_enqueue_ms;
    return 0;
}

int nb_comm_side_stream_collisions(nb_comm_t comm, int* collisions) {
    Comm* c = as_comm(comm);
    if (c == nullptr || collisions == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *collisions = c->aux_probed ? c->aux_collisions : -1;
    return 0;
}

int nb_comm_last_step_trace(nb_comm_t comm, char* text, size_t bytes) {
    Comm* c = as_comm(comm);
    if (c == nullptr || text == nullptr || bytes == 0) return NB_ERR_INVALID_ARGUMENT;
    std::strncpy(text, c->trace.c_str(), bytes - 1);
    text[bytes - 1] = 0;
    return 0;
}

int nb_comm_pair_work_f32(nb_comm_t comm, unsigned num_bodies, unsigned long long* pair_evaluations, int* force_launches) { return comm_pair_work<float>(comm, num_bodies, pair_evaluations, force_launches); }
int nb_comm_pair_work_f64(nb_comm_t comm, unsigned num_bodies, unsigned long long* pair_evaluations, int* force_launches) { return comm_pair_work<double>(comm, num_bodies, pair_evaluations, force_launches); }

int nb_stream_create_placed(nb_stream_t* stream) {  // (nb_comm_stream_create without a communicator: for hosts that bring their own RCCL, e.g. torch.distributed)
    NB_KEEP_RAND_STREAM;
    if (stream == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *stream = nullptr;
    static std::mutex               guard;
    static std::vector<hipStream_t> retired;  // candidates on the null stream's queue: kept, so that the pool moves on
    std::lock_guard<std::mutex>     lock(guard);
    hipStream_t                     made = nullptr;
    if (const auto err = settle_side_stream(&made, nullptr, &retired, nullptr); err != hipSuccess) return static_cast<int>(err);
    *stream = made;
    return 0;
}

int nb_comm_stream_create(nb_comm_t comm, nb_stream_t* stream) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr || stream == nullptr) return NB_ERR_INVALID_ARGUMENT;
    *stream = nullptr;
    DeviceScope scope(c->device);
    hipStream_t made = nullptr;
    // (a non-blocking stream that does not share the null stream's hardware queue: candidates that do are kept until the communicator
    // goes, so that the pool moves on; with one rank there is no RCCL and nothing to avoid)
    const auto err = c->world > 1 ? settle_side_stream(&made, nullptr, &c->aux_retired, nullptr) : hipStreamCreateWithFlags(&made, hipStreamNonBlocking);
    if (err != hipSuccess) return static_cast<int>(err);
    // the step takes this stream's placement from here (no probe inside a step); a note left by an earlier stream of the same handle goes
    c->seen.erase(std::remove_if(c->seen.begin(), c->seen.end(), [&](const StreamNote& n) { return n.stream == made; }), c->seen.end());
    if (std::find(c->placed.begin(), c->placed.end(), made) == c->placed.end()) c->placed.push_back(made);
    *stream = made;
    return 0;
}

int nb_comm_info(nb_comm_t comm, int* rank, int* world, int* device) {
    Comm* c = as_comm(comm);
    if (c == nullptr) return NB_ERR_INVALID_ARGUMENT;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return 0;
}

static int exchange_one(nb_comm_t comm, void* positions, unsigned num_bodies, size_t bytes_per_body, int type, nb_stream_t after) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr || positions == nullptr) return NB_ERR_INVALID_ARGUMENT;
    if (c->group.size() != 1) return NB_ERR_INVALID_ARGUMENT;  // several local ranks: use the *_all form (one ncclGroup per round)
    if (c->world == 1) return 0;
    void*       arrays[1] = {positions};
    hipStream_t streams[1] = {reinterpret_cast<hipStream_t>(after)};
    return exchange_tiles(c->group, arrays, num_bodies, bytes_per_body, type, streams);
}
int nb_exchange_tiles_f32(nb_comm_t comm, float* positions, unsigned num_bodies, nb_stream_t after) { return exchange_one(comm, positions, num_bodies, 16, ncclFloat32, after); }
int nb_exchange_tiles_f64(nb_comm_t comm, double* positions, unsigned num_bodies, nb_stream_t after) { return exchange_one(comm, positions, num_bodies, 32, ncclFloat64, after); }

int nb_exchange_wait_tile(nb_comm_t comm, int peer, nb_stream_t stream) {
    Comm* c = as_comm(comm);
    if (c == nullptr || peer < 0 || peer >= c->world) return NB_ERR_INVALID_ARGUMENT;
    if (peer == c->rank || c->in_flight == nullptr) return 0;
    DeviceScope scope(c->device);
    return static_cast<int>(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->arrived[static_cast<size_t>(peer)], 0));
}

int nb_exchange_wait_all(nb_comm_t comm, nb_stream_t stream) {
    Comm* c = as_comm(comm);
    if (c == nullptr) return NB_ERR_INVALID_ARGUMENT;
    for (int p = 0; p < c->world; ++p) {
        const int rc = nb_exchange_wait_tile(comm, p, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

static int allgather_one(nb_comm_t comm, void* positions, unsigned num_bodies, size_t bytes_per_body, int type, nb_stream_t after) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr || positions == nullptr || c->group.size() != 1 || num_bodies % static_cast<unsigned>(c->world)) return NB_ERR_INVALID_ARGUMENT;
    if (c->world == 1) return 0;
    if (c->self_peers) return NB_ERR_UNSUPPORTED;  // (a collective over a nominal world has no one-rank form)
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    DeviceScope  scope(c->device);
    const size_t slice = num_bodies / static_cast<unsigned>(c->world);
    auto         err   = hipEventRecord(c->ready, reinterpret_cast<hipStream_t>(after));
    if (err == hipSuccess) err = hipStreamWaitEvent(c->stream, c->ready, 0);
    if (err != hipSuccess) return static_cast<int>(err);
    c->in_flight = positions;
    char* base   = static_cast<char*>(positions);
    // in place: the own slice already sits at its offset, RCCL moves the others
    const int rc = lib->AllGather(base + static_cast<size_t>(c->rank) * slice * bytes_per_body, base, slice * 4, type, c->nccl, c->stream);
    if (rc != 0) return nccl_status(rc);
    for (int p = 0; p < c->world; ++p) {  // one collective: every tile "arrives" with it
        err = hipEventRecord(c->arrived[static_cast<size_t>(p)], c->stream);
        if (err != hipSuccess) return static_cast<int>(err);
    }
    return 0;
}
int nb_allgather_f32(nb_comm_t comm, float* positions, unsigned num_bodies, nb_stream_t after) { return allgather_one(comm, positions, num_bodies, 16, ncclFloat32, after); }
int nb_allgather_f64(nb_comm_t comm, double* positions, unsigned num_bodies, nb_stream_t after) { return allgather_one(comm, positions, num_bodies, 32, ncclFloat64, after); }

int nb_sharded_step_f32(nb_comm_t comm, float* new_positions, const float* old_positions, float* velocities, float* acc, unsigned num_bodies, float dt, float damping, int block_size, int mode, nb_stream_t stream) {
    Comm* c = as_comm(comm);
    if (c == nullptr || c->group.size() != 1) return NB_ERR_INVALID_ARGUMENT;
    return sharded_step<float>(&comm, 1, &new_positions, &old_positions, &velocities, &acc, num_bodies, dt, damping, block_size, mode, &stream);
}
int nb_sharded_step_f64(nb_comm_t comm, double* new_positions, const double* old_positions, double* velocities, double* acc, unsigned num_bodies, double dt, double damping, int block_size, int mode, nb_stream_t stream) {
    Comm* c = as_comm(comm);
    if (c == nullptr || c->group.size() != 1) return NB_ERR_INVALID_ARGUMENT;
    return sharded_step<double>(&comm, 1, &new_positions, &old_positions, &velocities, &acc, num_bodies, dt, damping, block_size, mode, &stream);
}
int nb_sharded_step_all_f32(const nb_comm_t* comms, int num_local, float* const* new_positions, const float* const* old_positions, float* const* velocities, float* const* acc, unsigned num_bodies, float dt, float damping,
                            int block_size, int mode, const nb_stream_t* streams) {
    return sharded_step<float>(comms, num_local, new_positions, old_positions, velocities, acc, num_bodies, dt, damping, block_size, mode, streams);
}
int nb_sharded_step_all_f64(const nb_comm_t* comms, int num_local, double* const* new_positions, const double* const* old_positions, double* const* velocities, double* const* acc, unsigned num_bodies, double dt,
                            double damping, int block_size, int mode, const nb_stream_t* streams) {
    return sharded_step<double>(comms, num_local, new_positions, old_positions, velocities, acc, num_bodies, dt, damping, block_size, mode, streams);
}

}  // extern "C"
