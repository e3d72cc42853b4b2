// nbody_kernels.h -- internal launch interface between the C-ABI translation unit (nbody_capi.hip) and
// the two kernel translation units.  They are separate TUs on purpose: nbody_strict.hip is compiled
// with -ffp-contract=off (bit-reproduction of the CPU path), nbody_fast.hip with contraction on.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

namespace nb {

// Dynamic LDS above 64 KiB needs an opt-in per KERNEL and per DEVICE (gfx950 has 160 KiB per CU).  The state is keyed on the
// kernel itself -- it is a non-type template parameter, so every instantiation of a kernel template owns its own bit mask
// (round 2 keyed it on the kernel's pointer TYPE, which all integrate_bodies_fast<T,...> of one precision share: once one
// was armed the others skipped the call).  `lds_optins` counts the (kernel, device) pairs armed so far (nb_lds_optin_count).
inline std::atomic<int> lds_optins{0};
template <auto Kernel> hipError_t allow_large_lds() {
    static std::atomic<unsigned long long> armed{0};  // one bit per device
    int device = 0;
    if (const auto err = hipGetDevice(&device); err != hipSuccess) return err;
    const unsigned long long bit = 1ull << (device & 63);
    if (armed.load(std::memory_order_acquire) & bit) return hipSuccess;
    const auto err = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess && !(armed.fetch_or(bit, std::memory_order_acq_rel) & bit)) lds_optins.fetch_add(1, std::memory_order_relaxed);
    return err;
}

template <typename T> struct Shard {
    T*       new_pos;  // vec4[N], written for i in the shard when finalize
    const T* old_pos;  // vec4[N]
    T*       vel;      // vec4[N], in/out when finalize
    T*       acc;      // vec4[N] partial accelerations (in when acc_in, out when !finalize); may be null otherwise
    unsigned i_begin, i_count;
    unsigned j_begin, j_count;
    bool     acc_in, finalize;
    T        dt, damping, eps2;
};

// geometry of the fast path (see nb_launch_plan_t in include/nbody_hip.h)
struct Plan {
    int      bodies_per_lane;  // I
    int      lanes_per_body;   // S
    int      tile_bodies;      // LDS tile
    int      block_threads;
    unsigned grid_blocks;
    unsigned lds_bytes;
};

// geometry of the pairwise layout (nbody_pair.hip; nb_pair_plan_t in include/nbody_hip.h)
struct PairPlan {
    int      vectors_per_lane;  // R: fp32 packed pairs / fp64 bodies i per lane
    int      waves;             // S: waves per workgroup (they share the bodies i, split the tiles of bodies j)
    unsigned splits;            // C: workgroups per block of bodies i
    unsigned blocks;            // NB
    unsigned block_bodies;      // 64 * I
    unsigned slots;             // reaction slots per body in the workspace
    unsigned grid_blocks;
    unsigned lds_bytes;
    size_t   workspace_bytes;
};

// Workgroups per block for ONE launch of the slice / shard plans (diagonal or rectangle): all workgroups of a launch carry the same
// work, and a launch of eight-wave workgroups costs ceil(grid / 256) rounds whatever the residency (nbody_pair.hip), so the C
// (any number up to 16) that fills `chip` workgroups -- or whole multiples of it -- best, while every wave keeps two units; among
// the choices within 2 % of the best fill the smallest C (fewer planes of partial sums).  chip: 256, or 128 where two launches
// run at once.  (69 blocks x 2 = 138 workgroups fill 54 % of a round; x 11 = 759 fill 99 % of three.)
inline unsigned splits_to_fill(unsigned blocks, unsigned units_per_block, int waves, unsigned chip) {
    auto fill = [&](unsigned C) {
        const double r = static_cast<double>(blocks) * C / chip;
        const double whole = static_cast<double>(static_cast<unsigned long long>(r) + ((r > static_cast<double>(static_cast<unsigned long long>(r))) ? 1 : 0));
        return r <= 1.0 ? r : r / whole;
    };
    double best = 0;
    for (unsigned C = 1; C <= 16 && (C == 1 || units_per_block >= 2 * C * static_cast<unsigned>(waves)); ++C) best = fill(C) > best ? fill(C) : best;
    for (unsigned C = 1; C <= 16 && (C == 1 || units_per_block >= 2 * C * static_cast<unsigned>(waves)); ++C) {
        if (fill(C) >= best - 0.02) return C;
    }
    return 1;
}

// LDS of one pair_forces workgroup: the waves' second-level sums / the fold buffer, the progress words, and -- when the units left
// over after an equal deal are run as quarters (PairArgs::deal == 2) -- the quarters' sums of the workgroup's tail units.
inline constexpr unsigned kPairLdsLimit = 160u * 1024u;
inline unsigned pair_tail_units(unsigned units, unsigned splits, int waves) {  // tail units of the busiest workgroup of a block
    const unsigned rem = units % (splits * static_cast<unsigned>(waves));
    return (rem + splits - 1) / splits;
}
inline unsigned pair_lds_bytes(int vectors_per_lane, int lane_width, int waves, unsigned element_bytes, unsigned tail_units) {
    return static_cast<unsigned>(static_cast<size_t>(waves) * 3 * vectors_per_lane * lane_width * 64 * element_bytes) + 256u + tail_units * 4u * 3u * 64u * element_bytes;
}

struct PairGeom {
    int      vectors_per_lane;  // R
    int      waves;             // S
    unsigned splits;            // C
};

// The pairwise layout with a BOUNDED workspace (round 4): the reaction slots of one tournament over the whole system grow with N^2
// (6.4 GB at 1 Mi bodies, 103 GB at 4 Mi).  Cut into K slices of bodies, the tournament runs slice by slice -- every slice against
// itself (diag) and against the next K/2 slices (rectangles; for an even K the two partners at distance K/2 split theirs) -- through
// ONE reusable region of reaction planes, each launch's planes folded straight away (pair_reduce) into one array per receiving
// slice; the finish kernel of a slice adds its own sums and the arrays it received.  The same pieces, in the same roles, as one
// rank's part of a multi-GPU pairwise step (nbody_comm.hip), only that "sending" is a pointer.  slices == 1: the single tournament.
struct PairSlicing {
    unsigned slices;        // K (after rounding: every slice holds at least one body)
    unsigned slice_bodies;  // bodies per slice, a multiple of the block (the last slice may hold fewer)
    unsigned partners;      // H = K / 2
    bool     even;
    PairGeom diag, rect, rect_upper;  // rect_upper: a split rectangle as the higher slice runs it (half of its blocks: twice the workgroups per block)
    unsigned block_bodies, plane;
    size_t   self_per_slice, react_elements, recv_per_slice, elements;  // in units of T
    size_t   workspace_bytes;
};

// One launch of the forces kernel: the bodies i of [i_begin, i_begin + i_count), cut into blocks of 64*I, against
//   diag = 1: themselves -- the tournament of the header comment within that range (one GPU: the whole system; a rank of a
//             multi-GPU system: its own slice), reaction slot q-1;
//   diag = 0: the bodies j of [j_begin, j_begin + j_count), every tile symmetric -- a rectangle of the pair matrix (a tile of
//             another rank's bodies); the reaction sums of block a go to slot a (keep = 0: they are dropped).
template <typename T> struct PairArgs {
    const T* old_pos;
    T*       self;          // i-side sums: slot (self_first + c), plane self_plane, body index i - self_origin
    T*       react;         // reaction sums: [slot][3][react_plane], body index j - react_origin
    unsigned n;             // bodies in the arrays
    unsigned i_begin, i_count, j_begin, j_count;
    unsigned blocks;        // NB = ceil(i_count / (64*I))
    unsigned splits;        // C workgroups per block
    unsigned diag, keep;
    unsigned unit_begin, unit_count;  // diag = 1: the launch takes units [unit_begin, unit_begin + unit_count) of the tournament's (blocks/2 + 1) * I per block
                                      // (unit u = tile u % I of block a + u / I); unit_count = 0: all of them.  Whole offsets q (multiples of I) keep the
                                      // reaction slots of two such launches apart.  diag = 0: both 0.
    unsigned deal;          // how the units reach the waves (set by launch_pair_tile; see pair_forces): 0 whole units, slots blocked; 1 whole units, slots
                            // interleaved; 2 whole units in equal numbers and the units left over as quarters, one per SIMD of one workgroup
    unsigned self_first, self_origin, self_plane;
    unsigned react_origin, react_plane;
    T        eps2;
};

// The finish kernel's view of one rank's sums (one GPU: everything with origin 0 and full coverage).
inline constexpr int kMaxSelfSets = 12, kMaxRecv = 8;
template <typename T> struct FinishArgs {
    const T* old_pos;
    T*       new_pos;
    T*       vel;
    const T* self;          // [slot][3][self_plane]
    const T* react;         // the diagonal launch's reaction slots [react_slots][3][react_plane]
    const T* recv;          // reaction sums received from other ranks [n_recv][3][recv_plane]
    const T* extra;         // optional partial accelerations vec4[N] (one-sided tiles), added
    unsigned origin, count; // the bodies this rank integrates
    unsigned self_plane, react_plane, react_slots, recv_plane;
    unsigned n_self, n_recv;
    struct Cover {
        unsigned slot, slots, first, count;  // `slots` consecutive slots from `slot` cover bodies [first, first + count) (relative to origin)
    } self_set[kMaxSelfSets];
    struct Window {
        unsigned first, count;
    } recv_set[kMaxRecv];
    T dt, damping;
};

// total memory of the current device, or what nb_set_memory_budget says instead; 0 = unknown (no device): no guard applies
size_t device_memory_budget();

// nb_set_pair_probe_event (tuning header): an event recorded between the forces kernel and the finish kernel of a one-GPU pairwise
// step, so that a benchmark can time the two separately; nullptr = none (defined in nbody_capi.hip)
hipEvent_t pair_probe_event();

// Tests of the out-of-memory fall-backs: nb_alloc hands the runtime a request no device can serve for anything above this many
// bytes (0 = no limit).  libnbody_hip.so has no way to set it; the lab library's nb_set_alloc_limit does (defined in nbody_capi.hip).
std::atomic<size_t>& alloc_limit();

// nb_set_pair_clock_words (tuning header): device memory for the in-kernel clock reading (pair_forces_clocked: any launch of the
// R = 8, S = 8 geometry whose workgroups fit), and its size in bytes; nullptr = none (defined in nbody_capi.hip)
unsigned long long* pair_clock_words(size_t* bytes);

// nb_set_pair_plan_override: 0 = automatic (defined in nbody_capi.hip; the multi-GPU layer honours it for its tiles too)
void pair_plan_overrides(int* vectors_per_lane, int* waves, int* splits);


template <typename T> PairPlan   plan_pair(unsigned n, int cu_count, int ovr_r, int ovr_s, int ovr_c);
// the pieces of a pairwise step, for callers that compose them themselves (nbody_comm.hip: one rank of a multi-GPU system)
template <typename T> hipError_t launch_pair_tile(const PairArgs<T>& args, const PairGeom& g, hipStream_t stream, bool prepare_only = false);  // fills blocks / splits
template <typename T> hipError_t launch_pair_reduce(const T* react, unsigned react_plane, unsigned slots, T* out, unsigned out_plane, unsigned count, hipStream_t stream);
template <typename T> hipError_t launch_pair_finish(const FinishArgs<T>& args, hipStream_t stream);
template <typename T> hipError_t launch_pair(const Shard<T>& s, const PairPlan& p, void* workspace, hipStream_t stream, bool prepare_only = false);
template <typename T> PairSlicing plan_pair_sliced(unsigned n, unsigned slices, int ovr_r, int ovr_s, int ovr_c);
template <typename T> hipError_t  launch_pair_sliced(const Shard<T>& s, const PairSlicing& p, void* workspace, hipStream_t stream, bool prepare_only = false);
template <typename T> Plan       plan_fast(unsigned i_count, unsigned j_count, int cu_count, int ovr_i, int ovr_s, int ovr_tile);
template <typename T> hipError_t launch_fast(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only = false);
template <typename T> hipError_t launch_strict(const Shard<T>& s, int block_size, int cu_count, hipStream_t stream, bool prepare_only = false);

}  // namespace nb
