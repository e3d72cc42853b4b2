// nbody_kernels.h -- internal launch interface between the C-ABI translation unit (nbody_capi.hip) and
// the two kernel translation units.  They are separate TUs on purpose: nbody_strict.hip is compiled
// with -ffp-contract=off (bit-reproduction of the CPU path), nbody_fast.hip with contraction on.
#pragma once

#include <hip/hip_runtime.h>

namespace nb {

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

template <typename T> Plan       plan_fast(unsigned i_count, unsigned j_count, int cu_count, int ovr_i, int ovr_s, int ovr_tile);
template <typename T> hipError_t launch_fast(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only = false);
template <typename T> hipError_t launch_strict(const Shard<T>& s, int block_size, int cu_count, hipStream_t stream);

}  // namespace nb
