// nbody_fast.hip -- the production all-pairs kernels (NB_MODE_FAST).  gfx950 (CDNA4) only.
//
// What it computes is the reference kernel's bodyBodyInteraction / computeBodyAccel / integrateBodies
// (/root/reference/src/nbody/bodysystemcuda.cu:98-184); how it is laid out is MI355X-first:
//
//   * VALU-issue bound (13 VALU ops + 1 double-cost v_rsq_f32 per interaction, no MFMA: a 3-vector
//     accumulate is not a contraction).  Everything else is arranged so the VALU never waits.
//   * 256-thread workgroups = 4 wave64.  Each lane register-tiles I bodies i (I x 6 VGPRs of state), so one
//     broadcast ds_read_b128 of a body j feeds I x 13 VALU ops.
//   * j bodies stream HBM/L2 -> registers -> LDS in tiles of TILE = 256*LPT float4 (coalesced 16 B/lane
//     global_load_dwordx4), double-buffered: tile t+1 is in flight in registers while tile t is consumed
//     from LDS; ONE barrier per tile.
//   * j-split: the S = 256/L lane groups of a workgroup (whole waves, L = 64..256 lanes) walk disjoint
//     1/S slices of every tile for the SAME bodies i and are reduced through LDS in a fixed order at the
//     end (deterministic).  This is what fills 256 CUs x 4 SIMDs x >=2 waves when N/I < 131072 lanes.
//   * The shard form (i-range x j-range, optional partial sums in/out) is the same kernel; the single-GPU
//     step is the shard i = j = [0,N) with finalize.
//
// Compiled with FMA contraction ON (default) -- results differ from the CPU path in rounding only; the
// bit-exact path is nbody_strict.hip.
#include "nbody_kernels.h"

#include <algorithm>

namespace nb {
namespace {

constexpr int kBlock = 256;

template <typename T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

// 1/sqrt(x): fp32 = one v_rsq_f32 (1 ulp); fp64 = v_rsq_f64 seed + 2 Newton-Raphson steps
// (the reference calls CUDA's rsqrtf / rsqrt here, bodysystemcuda.cu:74-84).
__device__ __forceinline__ float rsqrt_T(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsqrt_T(double x) {
    double y = __builtin_amdgcn_rsq(x);
    // y <- y * (1.5 - 0.5*x*y*y), written as y + y*(0.5 - 0.5*x*y*y) for a smaller final rounding error
    const double hx = 0.5 * x;
    double       e  = __builtin_fma(-hx * y, y, 0.5);
    y               = __builtin_fma(y, e, y);
    e               = __builtin_fma(-hx * y, y, 0.5);
    y               = __builtin_fma(y, e, y);
    return y;
}

__device__ __forceinline__ float  fma_T(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_T(double a, double b, double c) { return __builtin_fma(a, b, c); }

// bodyBodyInteraction, bodysystemcuda.cu:98-123, for one body j against the I bodies i of this lane.
template <typename T, int I> __device__ __forceinline__ void interact(const typename V4<T>::type bj, const T (&px)[I], const T (&py)[I], const T (&pz)[I], T (&ax)[I], T (&ay)[I], T (&az)[I], const T eps2) {
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const T dx   = bj.x - px[k];
        const T dy   = bj.y - py[k];
        const T dz   = bj.z - pz[k];
        T       d2   = fma_T(dx, dx, eps2);
        d2           = fma_T(dy, dy, d2);
        d2           = fma_T(dz, dz, d2);
        const T inv  = rsqrt_T(d2);
        const T inv2 = inv * inv;
        const T s    = (bj.w * inv) * inv2;
        ax[k]        = fma_T(dx, s, ax[k]);
        ay[k]        = fma_T(dy, s, ay[k]);
        az[k]        = fma_T(dz, s, az[k]);
    }
}

// T: float|double   I: bodies i per lane   S: lane groups splitting j (1,2,4)   LPT: float4 loads per thread per tile
template <typename T, int I, int S, int LPT> __global__ __launch_bounds__(kBlock) void integrate_bodies_fast(Shard<T> s) {
    using vec4          = typename V4<T>::type;
    constexpr int TILE  = kBlock * LPT;      // bodies j per LDS tile
    constexpr int L     = kBlock / S;        // lanes per group = bodies i per "row"
    constexpr int SLICE = TILE / S;          // bodies j per group per tile
    constexpr int BODIES_PER_BLOCK = L * I;  // bodies i per workgroup
    static_assert(L % 64 == 0, "a lane group must be whole waves so the LDS read stays a broadcast");
    static_assert(SLICE % 8 == 0, "inner loop is unrolled by 8");

    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
    vec4* tile = reinterpret_cast<vec4*>(smem_raw);  // [2][TILE]

    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const int tid   = threadIdx.x;
    const int group = tid / L;
    const int lane  = tid - group * L;

    // bodies i of this lane: block_base + k*L + lane  (coalesced across the lanes of a group)
    const unsigned block_base = blockIdx.x * BODIES_PER_BLOCK;
    T        px[I], py[I], pz[I], ax[I], ay[I], az[I];
    unsigned idx[I];
    bool     active[I];
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const unsigned local = block_base + k * L + lane;
        active[k]            = local < s.i_count;
        idx[k]               = s.i_begin + (active[k] ? local : s.i_count - 1);
        const vec4 p         = old_pos[idx[k]];
        px[k] = p.x, py[k] = p.y, pz[k] = p.z;
        ax[k] = ay[k] = az[k] = 0;
    }
    if (s.acc_in && group == 0) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const vec4 a = reinterpret_cast<const vec4*>(s.acc)[idx[k]];
            ax[k] = a.x, ay[k] = a.y, az[k] = a.z;
        }
    }
    const T eps2 = s.eps2;

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
        for (int jj = 0; jj < SLICE; jj += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) interact<T, I>(slice[jj + u], px, py, pz, ax, ay, az, eps2);
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
                red[(((group - 1) * 3 + 0) * I + k) * L + lane] = ax[k];
                red[(((group - 1) * 3 + 1) * I + k) * L + lane] = ay[k];
                red[(((group - 1) * 3 + 2) * I + k) * L + lane] = az[k];
            }
        }
        __syncthreads();
        if (group == 0) {
#pragma unroll
            for (int g = 1; g < S; ++g) {
#pragma unroll
                for (int k = 0; k < I; ++k) {
                    ax[k] += red[(((g - 1) * 3 + 0) * I + k) * L + lane];
                    ay[k] += red[(((g - 1) * 3 + 1) * I + k) * L + lane];
                    az[k] += red[(((g - 1) * 3 + 2) * I + k) * L + lane];
                }
            }
        }
    }
    if (group != 0) return;

#pragma unroll
    for (int k = 0; k < I; ++k) {
        if (!active[k]) continue;
        const unsigned i = idx[k];
        if (s.finalize) {
            // integrateBodies, bodysystemcuda.cu:166-183
            vec4 v  = reinterpret_cast<const vec4*>(s.vel)[i];
            vec4 pn = old_pos[i];
            v.x     = fma_T(ax[k], s.dt, v.x) * s.damping;
            v.y     = fma_T(ay[k], s.dt, v.y) * s.damping;
            v.z     = fma_T(az[k], s.dt, v.z) * s.damping;
            pn.x    = fma_T(v.x, s.dt, pn.x);
            pn.y    = fma_T(v.y, s.dt, pn.y);
            pn.z    = fma_T(v.z, s.dt, pn.z);
            reinterpret_cast<vec4*>(s.new_pos)[i] = pn;
            reinterpret_cast<vec4*>(s.vel)[i]     = v;
        } else {
            vec4 a;
            a.x = ax[k], a.y = ay[k], a.z = az[k], a.w = 0;
            reinterpret_cast<vec4*>(s.acc)[i] = a;
        }
    }
}

template <typename T, int I, int S, int LPT> hipError_t launch_one(const Shard<T>& s, const Plan& p, hipStream_t stream) {
    hipLaunchKernelGGL((integrate_bodies_fast<T, I, S, LPT>), dim3(p.grid_blocks), dim3(kBlock), p.lds_bytes, stream, s);
    return hipGetLastError();
}

template <typename T, int I, int S> hipError_t dispatch_lpt(const Shard<T>& s, const Plan& p, hipStream_t stream) {
    switch (p.tile_bodies / kBlock) {
        case 1: return launch_one<T, I, S, 1>(s, p, stream);
        case 2: return launch_one<T, I, S, 2>(s, p, stream);
        case 4: return launch_one<T, I, S, 4>(s, p, stream);
        default: return hipErrorInvalidValue;
    }
}

template <typename T, int I> hipError_t dispatch_s(const Shard<T>& s, const Plan& p, hipStream_t stream) {
    switch (p.lanes_per_body) {
        case 1: return dispatch_lpt<T, I, 1>(s, p, stream);
        case 2: return dispatch_lpt<T, I, 2>(s, p, stream);
        case 4: return dispatch_lpt<T, I, 4>(s, p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// Geometry: fill 256 CUs x 4 SIMDs with >= kWavesPerSimd waves if the shard is big enough, preferring
// register tiling (I) over j-splitting (S) because I amortises the LDS broadcast and S costs a reduction.
template <typename T> Plan plan_fast(unsigned i_count, unsigned j_count, int cu_count, int ovr_i, int ovr_s, int ovr_tile) {
    (void)j_count;
    const long lanes_wanted = static_cast<long>(cu_count) * 4 * 64 * 4;  // 4 waves per SIMD
    int        I = 1, S = 1;
    // largest I (<=2 fp32, 1 fp64) that still leaves enough lanes; then S to make up the rest
    const int max_i = sizeof(T) == 4 ? 2 : 1;
    for (int cand = max_i; cand >= 1; cand /= 2) {
        if (static_cast<long>(i_count) / cand >= lanes_wanted || cand == 1) {
            I = cand;
            break;
        }
    }
    while (S < 4 && static_cast<long>(i_count) / I * S < lanes_wanted) S *= 2;
    int tile = 1024;
    if (ovr_i > 0) I = ovr_i;
    if (ovr_s > 0) S = ovr_s;
    if (ovr_tile > 0) tile = ovr_tile;

    Plan p;
    p.bodies_per_lane = I;
    p.lanes_per_body  = S;
    p.tile_bodies     = tile;
    p.block_threads   = kBlock;
    const unsigned bodies_per_block = static_cast<unsigned>(kBlock / S * I);
    p.grid_blocks     = (i_count + bodies_per_block - 1) / bodies_per_block;
    const size_t tile_bytes = 2ull * tile * 4 * sizeof(T);
    const size_t red_bytes  = static_cast<size_t>(S - 1) * 3 * I * (kBlock / S) * sizeof(T);
    p.lds_bytes             = static_cast<unsigned>(std::max(tile_bytes, red_bytes));
    return p;
}

template <typename T> hipError_t launch_fast(const Shard<T>& s, const Plan& p, hipStream_t stream) {
    switch (p.bodies_per_lane) {
        case 1: return dispatch_s<T, 1>(s, p, stream);
        case 2: return dispatch_s<T, 2>(s, p, stream);
        case 4: return dispatch_s<T, 4>(s, p, stream);
        default: return hipErrorInvalidValue;
    }
}

template Plan       plan_fast<float>(unsigned, unsigned, int, int, int, int);
template Plan       plan_fast<double>(unsigned, unsigned, int, int, int, int);
template hipError_t launch_fast<float>(const Shard<float>&, const Plan&, hipStream_t);
template hipError_t launch_fast<double>(const Shard<double>&, const Plan&, hipStream_t);

}  // namespace nb
