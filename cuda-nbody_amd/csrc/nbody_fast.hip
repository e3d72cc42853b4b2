// nbody_fast.hip -- the production all-pairs kernels (NB_MODE_FAST).  gfx950 (CDNA4) only.
//
// What it computes is the reference kernel's bodyBodyInteraction / computeBodyAccel / integrateBodies
// (/root/reference/src/nbody/bodysystemcuda.cu:98-184); how it is laid out is MI355X-first and follows
// what tools/valu_microbench.hip measured on the chip (profiles/round1_valu_microbench.txt):
//
//   * The loop is VALU-issue bound: no MFMA (a 3-vector accumulate is not a contraction), HBM traffic is
//     64 B per body per step.  On gfx950 the PACKED fp32 ops (v_pk_add/mul/fma_f32) deliver 1.2-1.4x the
//     lane-throughput of their scalar forms, and v_rsq_f32 costs ~2.6 FMA slots.  So for fp32 each lane
//     carries its bodies i in PAIRS, one pair per 64-bit VGPR pair: an interaction pair is
//     3 v_pk_add + 6 v_pk_fma + 3 v_pk_mul + 2 v_rsq_f32 (14 issue slots for 2 interactions instead of 26),
//     the j body's x/y/z/m being broadcast into both halves with op_sel, not moved.
//     fp64 has no packed form: one body per "vector"; m*d2^(-3/2) from the v_rsq_f64 seed by a 2-term series (Lane<double>::coupling).
//   * Tile layout (large shards): each lane register-tiles R vectors (I = R*W bodies i, W = 2 fp32 / 1 fp64), so
//     one broadcast ds_read_b128 of a body j feeds I interactions.  j bodies stream HBM/L2 -> registers -> LDS in
//     tiles of TILE = block*LPT bodies (coalesced 16 B/lane global_load_dwordx4), double-buffered: tile t+1 is in
//     flight in registers while tile t is consumed from LDS; ONE barrier per tile.
//   * j-split: the S wave groups of a workgroup (L = block/S lanes each, whole waves) walk disjoint 1/S slices of
//     every tile for the SAME bodies i and are folded through LDS in a fixed order at the end (deterministic).
//     Production geometry is S = 16: a 1024-thread workgroup = 4 waves on every SIMD of its CU (plan_fast below).
//   * Wave-split layout (small shards, fewer bodies i than lanes on the chip): a wave owns the bodies i, its 64
//     lanes split j, wavefront-64 butterfly fold at the end (integrate_bodies_wavesplit below).
//   * softening^2 lives in VGPRs: a VALU op with an SGPR source issues ~35 % slower on this chip.
//   * The shard form (i-range x j-range, optional partial sums in/out) is the same kernel; the single-GPU
//     step is the shard i = j = [0,N) with finalize.
//
// Compiled with FMA contraction ON (default) -- results differ from the CPU path in rounding only; the
// bit-exact path is nbody_strict.hip.
#include "nbody_kernels.h"

#include <algorithm>

namespace nb {
namespace {

// workgroup size by j-split factor: S <= 4 -> 256 threads, S = 8 -> 512, S = 16 -> 1024 (a lane group stays >= one wave)
constexpr int block_threads_for(int S) { return S <= 4 ? 256 : 64 * S; }
// lanes_per_body value that selects the wave-split layout (all 64 lanes of a wave split j for the wave's bodies i)
constexpr int kWaveSplit = 64;

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- per-precision traits: the "vector" a lane computes with -------------------------------------------
template <typename T> struct Lane;

template <> struct Lane<float> {
    using vec4                 = float4;
    using vec                  = v2f;  // two bodies i per vector -> v_pk_*_f32
    static constexpr int W     = 2;
    static __device__ __forceinline__ vec  splat(float a) { return vec{a, a}; }
    static __device__ __forceinline__ vec  fma(vec a, vec b, vec c) { return __builtin_elementwise_fma(a, b, c); }
    // s = m * d2^(-3/2): 2 x v_rsq_f32 (1 ulp, what the reference's rsqrtf is) + 3 v_pk_mul_f32   (bodysystemcuda.cu:110-115)
    static __device__ __forceinline__ vec coupling(vec m, vec d2) {
        const vec inv  = vec{__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
        const vec inv2 = inv * inv;
        return (m * inv) * inv2;
    }
    static __device__ __forceinline__ float get(vec a, int w) { return w == 0 ? a.x : a.y; }
    static __device__ __forceinline__ void  set(vec& a, int w, float v) {
        if (w == 0) a.x = v; else a.y = v;
    }
    static __device__ __forceinline__ void  keep_in_vgpr(vec& a) { asm volatile("" : "+v"(a)); }
};

template <> struct Lane<double> {
    using vec4                 = double4;
    using vec                  = double;
    static constexpr int W     = 1;
    static __device__ __forceinline__ vec splat(double a) { return a; }
    static __device__ __forceinline__ vec fma(vec a, vec b, vec c) { return __builtin_fma(a, b, c); }
    // s = m * d2^(-3/2) in full double precision from the v_rsq_f64 seed y0 (relative error <= 2^-23) WITHOUT
    // iterating on y: with r = 1 - d2*y0^2 (|r| <= 2^-22),  d2^(-3/2) = y0^3 (1-r)^(-3/2) = y0^3 (1 + 3/2 r + 15/8 r^2 + O(r^3)),
    // truncation 35/16 r^3 < 2^-64.  7 DP ops + the seed, against 10 for two Newton steps on y followed by the cube
    // (the reference calls CUDA's <= 1 ulp rsqrt(double) here, bodysystemcuda.cu:82-84,110-115).
    static __device__ __forceinline__ vec coupling(vec m, vec d2) {
        const double y0 = __builtin_amdgcn_rsq(d2);
        const double t0 = y0 * y0;
        const double r  = __builtin_fma(-d2, t0, 1.0);
        const double mc = m * (y0 * t0);
        const double w  = r * __builtin_fma(r, 1.875, 1.5);
        return __builtin_fma(mc, w, mc);
    }
    static __device__ __forceinline__ double get(vec a, int) { return a; }
    static __device__ __forceinline__ void   set(vec& a, int, double v) { a = v; }
    static __device__ __forceinline__ void   keep_in_vgpr(vec& a) { asm volatile("" : "+v"(a)); }
};

// bodyBodyInteraction, bodysystemcuda.cu:98-123, for one body j against the R vectors of bodies i of this lane.
template <typename T, int R>
__device__ __forceinline__ void interact(const typename Lane<T>::vec4 bj, const typename Lane<T>::vec (&px)[R], const typename Lane<T>::vec (&py)[R], const typename Lane<T>::vec (&pz)[R], typename Lane<T>::vec (&ax)[R],
                                         typename Lane<T>::vec (&ay)[R], typename Lane<T>::vec (&az)[R], const typename Lane<T>::vec eps2) {
    using L   = Lane<T>;
    using vec = typename L::vec;
    const vec bx = L::splat(bj.x), by = L::splat(bj.y), bz = L::splat(bj.z), bm = L::splat(bj.w);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const vec dx   = bx - px[r];
        const vec dy   = by - py[r];
        const vec dz   = bz - pz[r];
        vec       d2   = L::fma(dx, dx, eps2);
        d2             = L::fma(dy, dy, d2);
        d2             = L::fma(dz, dz, d2);
        const vec s    = L::coupling(bm, d2);
        ax[r]          = L::fma(dx, s, ax[r]);
        ay[r]          = L::fma(dy, s, ay[r]);
        az[r]          = L::fma(dz, s, az[r]);
    }
}

// T: float|double   R: vectors per lane (I = R*W bodies i)   S: lane groups splitting j   LPT: vec4 loads per thread per tile
template <typename T, int R, int S, int LPT> __global__ __launch_bounds__(block_threads_for(S)) void integrate_bodies_fast(Shard<T> s) {
    constexpr int kBlock = block_threads_for(S);
    using LT            = Lane<T>;
    using vec4          = typename LT::vec4;
    using vec           = typename LT::vec;
    constexpr int W     = LT::W;
    constexpr int I     = R * W;         // bodies i per lane
    constexpr int TILE  = kBlock * LPT;  // bodies j per LDS tile
    constexpr int L     = kBlock / S;    // lanes per group
    constexpr int SLICE = TILE / S;      // bodies j per group per tile
    constexpr int BODIES_PER_BLOCK = L * I;
    static_assert(L % 64 == 0, "a lane group must be whole waves so the LDS read stays a broadcast");
    // j bodies in flight per lane: 8 independent interaction chains (R vectors x U bodies j) hide the VALU latency.
    // The 512/1024-thread workgroups are capped at 128 VGPRs (4 waves/SIMD), so with R >= 2 they unroll less instead
    // of spilling (an R = 4 body at U = 8 spilled 1.2 KB/lane to scratch: 1.2 GB of HBM writes per launch).
    constexpr int U = (kBlock >= 512 && R >= 2) ? 8 / R : 8;
    static_assert(SLICE % U == 0, "inner loop is unrolled by U");

    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
    vec4* tile = reinterpret_cast<vec4*>(smem_raw);  // [2][TILE]

    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const int tid   = threadIdx.x;
    const int group = tid / L;
    const int lane  = tid - group * L;

    // bodies i of this lane: block_base + k*L + lane, k = r*W + w  (coalesced across the lanes of a group)
    const unsigned block_base = blockIdx.x * BODIES_PER_BLOCK;
    vec      px[R], py[R], pz[R], ax[R], ay[R], az[R];
    unsigned idx[I];
    bool     active[I];
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const unsigned local = block_base + k * L + lane;
        active[k]            = local < s.i_count;
        idx[k]               = s.i_begin + (active[k] ? local : s.i_count - 1);
        const vec4 p         = old_pos[idx[k]];
        LT::set(px[k / W], k % W, p.x);
        LT::set(py[k / W], k % W, p.y);
        LT::set(pz[k / W], k % W, p.z);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ax[r] = ay[r] = az[r] = LT::splat(0);
    if (s.acc_in && group == 0) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const vec4 a = reinterpret_cast<const vec4*>(s.acc)[idx[k]];
            LT::set(ax[k / W], k % W, a.x);
            LT::set(ay[k / W], k % W, a.y);
            LT::set(az[k / W], k % W, a.z);
        }
    }
    vec eps2 = LT::splat(s.eps2);
    LT::keep_in_vgpr(eps2);

    const unsigned j_end   = s.j_begin + s.j_count;
    const unsigned n_tiles = (s.j_count + TILE - 1) / TILE;

    // out-of-range j slots become zero-mass bodies at the origin: they add exactly 0 (eps2 > 0)
    auto load_tile = [&](unsigned t, vec4 (&regs)[LPT]) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const unsigned j = s.j_begin + t * TILE + r * kBlock + tid;
            vec4           v;
            v.x = v.y = v.z = v.w = 0;
            if (j < j_end) v = old_pos[j];
            regs[r] = v;
        }
    };
    auto store_tile = [&](int buf, const vec4 (&regs)[LPT]) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) tile[buf * TILE + r * kBlock + tid] = regs[r];
    };

    vec4 regs[LPT];
    load_tile(0, regs);
    store_tile(0, regs);
    __syncthreads();

    for (unsigned t = 0; t < n_tiles; ++t) {
        const int  cur       = t & 1;
        const bool have_next = (t + 1) < n_tiles;
        if (have_next) load_tile(t + 1, regs);  // global loads in flight across the compute below

        const vec4* __restrict__ slice = tile + cur * TILE + group * SLICE;
#pragma unroll 1
        for (int jj = 0; jj < SLICE; jj += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) interact<T, R>(slice[jj + u], px, py, pz, ax, ay, az, eps2);
        }

        if (have_next) store_tile(cur ^ 1, regs);
        __syncthreads();
    }

    // fold the S partial sums (groups 1..S-1 -> group 0) through LDS, fixed order
    if constexpr (S > 1) {
        T* red = reinterpret_cast<T*>(smem_raw);  // [(S-1)][3][I][L]; the tiles are dead after the last barrier
        if (group > 0) {
#pragma unroll
            for (int k = 0; k < I; ++k) {
                red[(((group - 1) * 3 + 0) * I + k) * L + lane] = LT::get(ax[k / W], k % W);
                red[(((group - 1) * 3 + 1) * I + k) * L + lane] = LT::get(ay[k / W], k % W);
                red[(((group - 1) * 3 + 2) * I + k) * L + lane] = LT::get(az[k / W], k % W);
            }
        }
        __syncthreads();
        if (group == 0) {
#pragma unroll 1  // (fully unrolled, the S = 16 fold hoists 45*I LDS loads and spills)
            for (int g = 1; g < S; ++g) {
#pragma unroll
                for (int k = 0; k < I; ++k) {
                    LT::set(ax[k / W], k % W, LT::get(ax[k / W], k % W) + red[(((g - 1) * 3 + 0) * I + k) * L + lane]);
                    LT::set(ay[k / W], k % W, LT::get(ay[k / W], k % W) + red[(((g - 1) * 3 + 1) * I + k) * L + lane]);
                    LT::set(az[k / W], k % W, LT::get(az[k / W], k % W) + red[(((g - 1) * 3 + 2) * I + k) * L + lane]);
                }
            }
        }
    }
    if (group != 0) return;

#pragma unroll
    for (int k = 0; k < I; ++k) {
        if (!active[k]) continue;
        const unsigned i  = idx[k];
        const T        fx = LT::get(ax[k / W], k % W), fy = LT::get(ay[k / W], k % W), fz = LT::get(az[k / W], k % W);
        if (s.finalize) {
            // integrateBodies, bodysystemcuda.cu:166-183
            vec4 v  = reinterpret_cast<const vec4*>(s.vel)[i];
            vec4 pn = old_pos[i];
            v.x     = __builtin_fma(fx, s.dt, v.x) * s.damping;
            v.y     = __builtin_fma(fy, s.dt, v.y) * s.damping;
            v.z     = __builtin_fma(fz, s.dt, v.z) * s.damping;
            pn.x    = __builtin_fma(v.x, s.dt, pn.x);
            pn.y    = __builtin_fma(v.y, s.dt, pn.y);
            pn.z    = __builtin_fma(v.z, s.dt, pn.z);
            reinterpret_cast<vec4*>(s.new_pos)[i] = pn;
            reinterpret_cast<vec4*>(s.vel)[i]     = v;
        } else {
            vec4 a;
            a.x = fx, a.y = fy, a.z = fz, a.w = 0;
            reinterpret_cast<vec4*>(s.acc)[i] = a;
        }
    }
}

// ---- wave-split layout for small shards ------------------------------------------------------------------
// When there are fewer bodies i than the chip has lanes (i_count < 64*W*#CUs: 32 768 fp32 bodies on 256 CUs) the
// layout above leaves CUs idle.  Here the roles turn: a WAVE owns R vectors of bodies i (wave-uniform registers)
// and its 64 lanes split the bodies j -- lane l takes tile entries l, l+64, ... (one conflict-free ds_read_b128 per
// lane, stride 16 B) -- and the 64 partial sums are folded with a wavefront-64 butterfly (ds_swizzle/bpermute
// shuffles, fixed order => deterministic).  Same interaction code, same LDS staging; 4 waves (4 x I bodies) per
// 256-thread workgroup, so 1 024 bodies already make 128-256 workgroups.
template <typename F> __device__ __forceinline__ F wave64_sum(F v) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) v += __shfl_xor(v, offset, 64);
    return v;
}

template <typename T, int R, int LPT> __global__ __launch_bounds__(256) void integrate_bodies_wavesplit(Shard<T> s) {
    using LT            = Lane<T>;
    using vec4          = typename LT::vec4;
    using vec           = typename LT::vec;
    constexpr int kBlock = 256;
    constexpr int WAVES = kBlock / 64;
    constexpr int W     = LT::W;
    constexpr int I     = R * W;          // bodies i per WAVE
    constexpr int TILE  = kBlock * LPT;   // bodies j per LDS tile
    constexpr int U     = 8 / R;          // j bodies in flight per lane (8 independent interaction chains)
    static_assert(TILE % (64 * U) == 0, "a tile is consumed in rounds of 64*U bodies");

    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
    vec4* tile = reinterpret_cast<vec4*>(smem_raw);  // [2][TILE]

    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const int tid  = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;

    // bodies i of this wave (every lane holds the same values)
    const unsigned wave_base = (blockIdx.x * WAVES + wave) * I;
    vec      px[R], py[R], pz[R], ax[R], ay[R], az[R];
    unsigned idx[I];
    bool     active[I];
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const unsigned local = wave_base + k;
        active[k]            = local < s.i_count;
        idx[k]               = s.i_begin + (active[k] ? local : s.i_count - 1);
        const vec4 p         = old_pos[idx[k]];
        LT::set(px[k / W], k % W, p.x);
        LT::set(py[k / W], k % W, p.y);
        LT::set(pz[k / W], k % W, p.z);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ax[r] = ay[r] = az[r] = LT::splat(0);
    vec eps2 = LT::splat(s.eps2);
    LT::keep_in_vgpr(eps2);

    const unsigned j_end   = s.j_begin + s.j_count;
    const unsigned n_tiles = (s.j_count + TILE - 1) / TILE;

    auto load_tile = [&](unsigned t, vec4 (&regs)[LPT]) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const unsigned j = s.j_begin + t * TILE + r * kBlock + tid;
            vec4           v;
            v.x = v.y = v.z = v.w = 0;  // out-of-range slots: zero-mass bodies, contribute exactly 0
            if (j < j_end) v = old_pos[j];
            regs[r] = v;
        }
    };
    auto store_tile = [&](int buf, const vec4 (&regs)[LPT]) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) tile[buf * TILE + r * kBlock + tid] = regs[r];
    };

    vec4 regs[LPT];
    load_tile(0, regs);
    store_tile(0, regs);
    __syncthreads();

    for (unsigned t = 0; t < n_tiles; ++t) {
        const int  cur       = t & 1;
        const bool have_next = (t + 1) < n_tiles;
        if (have_next) load_tile(t + 1, regs);
        const vec4* __restrict__ mine = tile + cur * TILE + lane;  // this lane's column of the tile
#pragma unroll 1
        for (int jj = 0; jj < TILE; jj += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) interact<T, R>(mine[jj + 64 * u], px, py, pz, ax, ay, az, eps2);
        }
        if (have_next) store_tile(cur ^ 1, regs);
        __syncthreads();
    }

    // wavefront-64 fold; lane k then finishes body k of the wave
#pragma unroll
    for (int k = 0; k < I; ++k) {
        T fx = wave64_sum(LT::get(ax[k / W], k % W));
        T fy = wave64_sum(LT::get(ay[k / W], k % W));
        T fz = wave64_sum(LT::get(az[k / W], k % W));
        if (lane != k || !active[k]) continue;
        const unsigned i = idx[k];
        if (s.acc_in) {
            const vec4 a = reinterpret_cast<const vec4*>(s.acc)[i];
            fx += a.x, fy += a.y, fz += a.z;
        }
        if (s.finalize) {
            vec4 v  = reinterpret_cast<const vec4*>(s.vel)[i];
            vec4 pn = old_pos[i];
            v.x     = __builtin_fma(fx, s.dt, v.x) * s.damping;
            v.y     = __builtin_fma(fy, s.dt, v.y) * s.damping;
            v.z     = __builtin_fma(fz, s.dt, v.z) * s.damping;
            pn.x    = __builtin_fma(v.x, s.dt, pn.x);
            pn.y    = __builtin_fma(v.y, s.dt, pn.y);
            pn.z    = __builtin_fma(v.z, s.dt, pn.z);
            reinterpret_cast<vec4*>(s.new_pos)[i] = pn;
            reinterpret_cast<vec4*>(s.vel)[i]     = v;
        } else {
            vec4 a;
            a.x = fx, a.y = fy, a.z = fz, a.w = 0;
            reinterpret_cast<vec4*>(s.acc)[i] = a;
        }
    }
}

template <typename T, int R, int LPT> hipError_t launch_wavesplit(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    if (prepare_only) return hipSuccess;
    hipLaunchKernelGGL((integrate_bodies_wavesplit<T, R, LPT>), dim3(p.grid_blocks), dim3(256), p.lds_bytes, stream, s);
    return hipGetLastError();
}

template <typename T, int R> hipError_t dispatch_wavesplit(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    switch (p.tile_bodies / 256) {
        case 2: return launch_wavesplit<T, R, 2>(s, p, stream, prepare_only);
        case 4: return launch_wavesplit<T, R, 4>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template <typename T, int R, int S, int LPT> hipError_t launch_one(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    if (p.lds_bytes > 64u * 1024u) {  // above the default dynamic-LDS ceiling: opt in once per kernel (gfx950 has 160 KiB per CU)
        static hipError_t opted = hipFuncSetAttribute(reinterpret_cast<const void*>(&integrate_bodies_fast<T, R, S, LPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (opted != hipSuccess) return opted;
    }
    if (prepare_only) return hipSuccess;  // graph capture arms the attribute before hipStreamBeginCapture
    hipLaunchKernelGGL((integrate_bodies_fast<T, R, S, LPT>), dim3(p.grid_blocks), dim3(block_threads_for(S)), p.lds_bytes, stream, s);
    return hipGetLastError();
}

template <typename T, int R, int S> hipError_t dispatch_lpt(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    constexpr int kBlock = block_threads_for(S);
    if (p.tile_bodies % kBlock) return hipErrorInvalidValue;
    switch (p.tile_bodies / kBlock) {
        case 1: return launch_one<T, R, S, 1>(s, p, stream, prepare_only);
        case 2: return launch_one<T, R, S, 2>(s, p, stream, prepare_only);
        case 4: return launch_one<T, R, S, 4>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template <typename T, int R> hipError_t dispatch_s(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    switch (p.lanes_per_body) {
        case 1: return dispatch_lpt<T, R, 1>(s, p, stream, prepare_only);
        case 2: return dispatch_lpt<T, R, 2>(s, p, stream, prepare_only);
        case 4: return dispatch_lpt<T, R, 4>(s, p, stream, prepare_only);
        case 8: return dispatch_lpt<T, R, 8>(s, p, stream, prepare_only);
        case 16: return dispatch_lpt<T, R, 16>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// Geometry (measured: profiles/round1_sweep_*.txt, the shard sweeps included).
//   * S = 16: one 1024-thread workgroup = 16 waves = 4 per SIMD, all working on the SAME 64*I bodies i and each
//     walking 1/16 of every LDS tile.  One resident workgroup per CU fills the chip for any shard with >= 256
//     workgroups, which is what keeps a 32 768-body shard (8-GPU strong scaling of 262 144 bodies) at 94 % of the
//     full-size rate; at full size it is as fast as any smaller split.
//   * I (bodies i per lane, a multiple of W: fp32 bodies travel in packed pairs; at most 4, what fits 128 VGPRs):
//     register tiling amortises the LDS broadcast and the per-tile barrier, but what matters more is how the
//     resulting workgroup count fills whole rounds of 256 CUs.
//   * LDS tile 2048 bodies (fp32, 2 x 32 KiB double-buffered) / 1024 (fp64): 128 bodies j per wave between barriers.
template <typename T> Plan plan_fast(unsigned i_count, unsigned j_count, int cu_count, int ovr_i, int ovr_s, int ovr_tile) {
    constexpr int W     = Lane<T>::W;
    constexpr int kMaxI = 4;  // fp32: 2 packed pairs, fp64: 4 bodies -- the most a 1024-thread workgroup holds in 128 VGPRs without spilling
    // Tile layout, S = 16: one workgroup per CU at a time, so its efficiency is the fill of the last round of
    // workgroups.  Pick the I whose workgroup count quantises best (larger I is ~3 % faster per interaction).
    int    I        = W;
    double best_eff = 0.0;
    for (int cand = W; cand <= kMaxI; cand *= 2) {
        const long   blocks = (static_cast<long>(i_count) + 64L * cand - 1) / (64L * cand);
        const long   rounds = (blocks + cu_count - 1) / cu_count;
        const double eff    = static_cast<double>(blocks) / static_cast<double>(rounds * cu_count) * (cand == kMaxI ? 1.0 : 0.97);
        if (eff >= best_eff) best_eff = eff, I = cand;
    }
    int S = 16;
    // The wave-split layout has no such quantisation (its workgroups are 64-256x smaller) but runs at ~0.78 (fp32) /
    // ~0.62 (fp64) of the tile layout's full rate (tools/layout_crossover.py): take it when the tile layout would
    // fill the chip worse than that.
    const double wave_split_eff = sizeof(T) == 4 ? 0.78 : 0.62;
    if (best_eff < wave_split_eff) {
        S = kWaveSplit;
        I = (static_cast<long>(i_count) / (4 * 2 * W) >= 4L * cu_count) ? 2 * W : W;  // 2 vectors per wave while that leaves >= 4 workgroups per CU
    }
    if (ovr_i > 0) I = std::max(ovr_i / W * W, W);
    if (ovr_s > 0) S = ovr_s;
    if (S == kWaveSplit) {
        if (I > 2 * W) I = 2 * W;
        Plan p;
        p.bodies_per_lane = I;  // per WAVE in this layout
        p.lanes_per_body  = kWaveSplit;
        p.tile_bodies     = (ovr_tile == 512 || ovr_tile == 1024) ? ovr_tile : (j_count > 512 ? 1024 : 512);
        p.block_threads   = 256;
        const unsigned bodies_per_block = 4u * static_cast<unsigned>(I);
        p.grid_blocks     = (i_count + bodies_per_block - 1) / bodies_per_block;
        p.lds_bytes       = static_cast<unsigned>(2ull * p.tile_bodies * 4 * sizeof(T));
        return p;
    }
    const int block = block_threads_for(S);
    int       tile  = (sizeof(T) == 4 && j_count >= 8192) ? 2048 : 1024;
    if (ovr_tile > 0) tile = ovr_tile;
    if (tile < block) tile = block;

    Plan p;
    p.bodies_per_lane = I;
    p.lanes_per_body  = S;
    p.tile_bodies     = tile;
    p.block_threads   = block;
    const unsigned bodies_per_block = static_cast<unsigned>(block / S * I);
    p.grid_blocks     = (i_count + bodies_per_block - 1) / bodies_per_block;
    const size_t tile_bytes = 2ull * tile * 4 * sizeof(T);
    const size_t red_bytes  = static_cast<size_t>(S - 1) * 3 * I * (block / S) * sizeof(T);
    p.lds_bytes             = static_cast<unsigned>(std::max(tile_bytes, red_bytes));
    return p;
}

template <typename T> hipError_t launch_fast(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    constexpr int W = Lane<T>::W;
    if (p.lanes_per_body == kWaveSplit) {
        switch (p.bodies_per_lane / W) {
            case 1: return dispatch_wavesplit<T, 1>(s, p, stream, prepare_only);
            case 2: return dispatch_wavesplit<T, 2>(s, p, stream, prepare_only);
            default: return hipErrorInvalidValue;
        }
    }
    switch (p.bodies_per_lane / W) {
        case 1: return dispatch_s<T, 1>(s, p, stream, prepare_only);
        case 2: return dispatch_s<T, 2>(s, p, stream, prepare_only);
        case 4: return dispatch_s<T, 4>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template Plan       plan_fast<float>(unsigned, unsigned, int, int, int, int);
template Plan       plan_fast<double>(unsigned, unsigned, int, int, int, int);
template hipError_t launch_fast<float>(const Shard<float>&, const Plan&, hipStream_t, bool);
template hipError_t launch_fast<double>(const Shard<double>&, const Plan&, hipStream_t, bool);

}  // namespace nb
