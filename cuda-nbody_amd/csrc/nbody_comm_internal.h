// nbody_comm_internal.h -- what the two translation units of the multi-GPU layer share: nbody_comm.hip (the product: the RCCL
// binding, a rank's resources, the exchange, the pairwise and one-sided steps, the crew of threads that enqueues them, the extern "C" entry points of
// include/nbody_hip.h) and nbody_comm_lab.hip (the lab bench of include/nbody_hip_lab.h: self-test, loopback rank, in-process
// world, kernel-time projections, A/B switches -- linked into libnbody_hip_lab.so only, never into libnbody_hip.so).
// Everything here has hidden visibility (-fvisibility=hidden): none of it is part of any ABI.
#pragma once

#include "../../include/nbody_hip.h"

#include "nbody_kernels.h"
#include "rccl_api.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <string>
#include <vector>

namespace nbc {

// ---- the few RCCL entry points used, resolved at run time: their types live in rccl_api.h, which `make check-rccl-abi` holds
// against /opt/rocm/include/rccl/rccl.h at compile time ----------------------------------------------------------------------
using ncclComm_t   = nb_rccl::Comm;
using ncclUniqueId = nb_rccl::UniqueId;
static_assert(sizeof(ncclUniqueId) == NB_COMM_ID_BYTES, "nb_comm_unique_id hands out exactly one ncclUniqueId");
enum { ncclFloat32 = nb_rccl::kFloat32, ncclFloat64 = nb_rccl::kFloat64 };

struct Rccl {
    void*       handle = nullptr;
    std::string path;  // the file the entry points were bound from (dladdr), for the record
    nb_rccl::GetVersionFn     GetVersion     = nullptr;  // (optional: only reported)
    nb_rccl::GetUniqueIdFn    GetUniqueId    = nullptr;
    nb_rccl::CommInitRankFn   CommInitRank   = nullptr;
    nb_rccl::CommInitAllFn    CommInitAll    = nullptr;
    nb_rccl::CommDestroyFn    CommDestroy    = nullptr;
    nb_rccl::SendFn           Send           = nullptr;
    nb_rccl::RecvFn           Recv           = nullptr;
    nb_rccl::AllGatherFn      AllGather      = nullptr;
    nb_rccl::GroupStartFn     GroupStart     = nullptr;
    nb_rccl::GroupEndFn       GroupEnd       = nullptr;
    nb_rccl::GetErrorStringFn GetErrorString = nullptr;
};
Rccl* rccl();  // nullptr: no RCCL to bind (NB_ERR_UNSUPPORTED)

// what the library knows about a stream a caller stepped on (looked at ONCE per stream, outside any capture: note_stream)
struct StreamNote {
    hipStream_t stream     = nullptr;
    int         placement  = -1;     // 1: the null stream, or a stream that shares its hardware queue (RCCL works there: ~40 % slower steps); 0: fine; -1: not looked at
    bool        aux_beside = false;  // the rank's second compute stream has been probed to run beside this one
};

// ---- one local rank ------------------------------------------------------------------------------------------------------
struct Comm {
    ncclComm_t  nccl   = nullptr;
    int         rank   = 0;
    int         world  = 1;
    int         device = 0;
    hipStream_t stream = nullptr;          // the exchange runs here (high priority: its few workgroups must not queue behind a force kernel)
    hipEvent_t  ready  = nullptr;          // "what the exchange has to wait for has been enqueued" (recorded on the compute stream)
    std::vector<hipEvent_t> arrived;       // [world]: arrived[p] = the round that brings rank p's tile is done
    const void* in_flight = nullptr;       // the array whose tiles are (or were last) being exchanged
    void*       workspace = nullptr;       // caller-owned scratch memory (nb_comm_set_workspace)
    size_t      workspace_bytes = 0;
    size_t      agreed_bytes    = 0;       // one process per rank: the SMALLEST amount any rank of the communicator was lent (set_workspace's exchange)
    size_t      agreed_budget   = 0;       // ... and the smallest device memory budget of any rank (0 until that exchange: this rank's own)
    bool        one_group       = false;   // nb_comm_set_exchange_grouping: all G-1 position rounds of a step in one RCCL group (default: a group per round)
    unsigned long long* notes   = nullptr; // [world][kNoteWords] device memory of the communicator: what set_workspace's ranks tell each other
    hipStream_t aux       = nullptr;       // pairwise step: every other rectangle runs here, so that the tails and launch gaps of
    hipEvent_t  aux_begin = nullptr;       // one stream's kernels are filled by the other's (events: aux may start / aux is done)
    hipEvent_t  aux_done  = nullptr;
    std::vector<StreamNote> seen;          // the streams this rank has stepped on (a handful: a caller alternating two streams is probed twice, not per step)
    hipStream_t last_caller = nullptr;     // the stream of the rank's last step (what nb_comm_caller_stream_placement talks about)
    int         aux_collisions = 0;        // how many candidates shared a hardware queue with a caller's stream (nb_comm_side_stream_collisions reports it)
    bool        aux_probed = false;
    std::vector<hipStream_t> aux_retired;  // ... kept until the communicator goes
    std::vector<hipStream_t> placed;       // streams handed out by nb_comm_stream_create: probed to be clear of the null stream's queue when they were made
    std::vector<hipEvent_t> react_ready;   // [world/2 + 1] pairwise step: "the reaction sums for partner s are in the send buffer" (compute stream)
    std::vector<hipEvent_t> react_arrived; // [world/2 + 1] ... "round s of the reaction exchange is done" (exchange stream)
    hipEvent_t  cut_ready = nullptr;       // ... "the cut-off part of the cut rectangle is folded into the send buffer too" (the step's own stream; PairShard::cut_round)
    std::vector<Comm*> group;              // all local ranks of this communicator (just {this} with one process per GPU)
    std::string trace;                     // what the rank's last pairwise step enqueued, in host order (nb_comm_last_step_trace: tests read the order)
    double      last_enqueue_ms = 0;       // host wall clock of the last nb_sharded_step_* call this rank took part in (the whole call: what the HOST needs to enqueue a step)
    // resource ownership (a lab world shares them between ranks: nbody_comm_lab.hip)
    bool        self_peers  = false;       // every peer of this rank is RCCL rank 0 of its ncclComm (a one-rank ncclComm behind a nominal world)
    bool        owns_stream = true;
    std::shared_ptr<void> shared_nccl;     // set: the ncclComm goes with the LAST rank that holds it
    std::shared_ptr<void> shared_stream;   // ... and so does the exchange stream
    std::shared_ptr<void> crew;            // the threads that enqueue this group's steps, one per local rank (nbody_comm.hip: StepCrew); goes with the group's last rank
};

// the RCCL rank behind rank `logical` of the communicator
inline int   peer_of(const Comm* c, int logical) { return c->self_peers ? 0 : logical; }
inline Comm* as_comm(nb_comm_t c) { return static_cast<Comm*>(c); }
inline int   nccl_status(int r) { return r == 0 ? 0 : NB_ERR_RCCL_BASE + r; }

constexpr int kNoteWords = 8;  // {workspace bytes, min slice, pair plan overrides R S C, device memory budget, late diagonal, spare}

class DeviceScope {  // switch device for a few calls, restore on exit (single-process multi-GPU)
 public:
    explicit DeviceScope(int device) {
        (void)hipGetDevice(&saved_);
        if (saved_ != device) (void)hipSetDevice(device);
    }
    ~DeviceScope() {
        int now = 0;
        (void)hipGetDevice(&now);
        if (now != saved_) (void)hipSetDevice(saved_);
    }

 private:
    int saved_ = 0;
};

int  make_resources(Comm* c);
void free_resources(Comm* c);
// `alone_too`: bind RCCL and make a real communicator even for a world of one; `self_peers`: the ncclComm has ONE rank whatever `world` says
int  init_rank(nb_comm_t* comm, const void* id, int world, int rank, bool alone_too, bool self_peers = false);

hipError_t create_side_stream(hipStream_t* stream);
bool       streams_run_side_by_side(hipStream_t a, hipStream_t b);
hipError_t settle_side_stream(hipStream_t* side, hipStream_t beside, std::vector<hipStream_t>* retired, int* collisions);
bool       stream_is_capturing(hipStream_t s);

// ---- the pairwise step across the ranks ----------------------------------------------------------------------------------------
struct PairShard {
    bool         applies = false;
    nb::PairGeom diag{}, diag_late{}, rect{}, rect_upper{};  // rect_upper: the split rectangle as the HIGHER partner runs it (half of its blocks of bodies i: twice the workgroups per block)
    unsigned     ni = 0, block = 0, blocks = 0, plane = 0, half = 0, H = 0, diag_slots = 0;
    unsigned     send_order[nb::kMaxRecv] = {};  // the reaction rounds s = 1 .. H in the order their rectangles' folds are expected to complete (send_order[k] = s)
    unsigned     early_units = 0, late_units = 0;  // the diagonal's units per block as two launches: block offsets q < q_split first, the rest LAST (late_units == 0: one launch)
    // round 6 (nb_set_late_diagonal(2)): BOTH compute streams end on local work -- the late units are dealt to the two streams (late_units
    // = what the step's own stream takes, late_aux_units = the second stream's), and the last rectangle that runs on the second stream is
    // CUT by bodies j: its first cut_tiles tiles run on the step's own stream (as much work as the second stream took over)
    unsigned     late_aux_units = 0, cut_round = 0, cut_tiles = 0;
    nb::PairGeom diag_late_aux{}, rect_cut{};
    unsigned     extra_self_first = 0;  // first i-side plane of {the second stream's late diagonal, the cut-off part of the rectangle}
    bool         even = false;
    size_t       self_at = 0, react_d_at = 0, react_r_at = 0, send_at = 0, recv_at = 0, elements = 0;  // offsets in T
};
extern std::atomic<int> g_late_diagonal;   // nb_set_late_diagonal: 0 = one launch, first; 1 = two launches, the second one last; 2 = 1 + both streams end on local work (cut rectangle)
extern std::atomic<int> g_pair_shard_min;  // nb_comm_set_pair_min_slice: tests run the pairwise step on small slices

enum RankPart { kWholeStep = 0, kBeforeSends = 1, kAfterSends = 2 };
template <typename T> PairShard plan_pair_shard(unsigned num_bodies, int G, int min_slice, size_t budget);
template <typename T>
int pair_rank_tiles(Comm* c, unsigned r, int G, const PairShard& plan, T* work, T* new_pos, const T* old_pos, T* vel, unsigned num_bodies, T dt, T damping, T eps2, hipStream_t stream, bool waiting, nb::FinishArgs<T>& f,
                    hipStream_t aux, hipEvent_t aux_begin, hipEvent_t aux_done, RankPart part = kWholeStep);
template <typename T> bool step_is_pairwise(const std::vector<Comm*>& locals, unsigned num_bodies, int mode, PairShard* plan_out);
template <typename T> int  reaction_exchange(const std::vector<Comm*>& locals, const PairShard& plan);

}  // namespace nbc
