// nbody_strict.hip -- the bit-reproducing kernels (NB_MODE_STRICT).  gfx950 only.
//
// MUST be compiled with -ffp-contract=off (csrc/Makefile does): every arithmetic op below has to stay
// the separate IEEE-754 mul/add/sub/div/sqrt the reference's CPU path executes
// (/root/reference/src/nbody/bodysystemcpu.cpp:140-303).  hipcc's defaults supply the rest:
// correctly rounded fp32 divide/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt is on by default),
// fp32 denormals kept, IEEE mode on.
//
// Mapping: one lane = one body i (bodysystemcuda.cu:151 uses the same mapping); every wave streams ALL bodies j of the
// range in ascending order through its own double-buffered 128-body LDS ring (no workgroup barrier in the loop: a wave
// reads only what it wrote itself), so each body i sees j = j_begin .. j_begin+j_count-1 in exactly the CPU path's order
// (bodysystemcpu.cpp:156 / :251).  Results do not depend on the launch geometry, so launch_strict picks it; the
// reference's --blockSize is validated and otherwise a hint.
//
// Two arithmetic forms of the same IEEE operations:
//   * generic : `/` and sqrtf as hipcc expands them (v_div_scale/v_rcp/fma chain/v_div_fmas/v_div_fixup; v_sqrt + the
//     +-1 ulp residual checks + denormal-range scaling): correct for every input, ~37 VALU per interaction.
//   * fast (fp32): two CONSECUTIVE bodies j travel as a PACKED pair against the lane's body i (v_pk_add/mul/fma_f32 are
//     the same IEEE operations, two lanes' worth per instruction; only the three running sums take the two results
//     one after the other, in j order), and divide / sqrt run WITHOUT the scaling and fix-up steps: those only act
//     on operands outside a window that is checked up front -- all coordinates |c| <= 2^18, softening^2 in
//     [2^-39, 2^38], masses +0 or 2^-40 <= |m| <= 2^40 -- per wave (bodies i) and per 128-body chunk (bodies j), and any
//     chunk outside it takes the generic form.  Inside the window r2 is in [2^-39, 2^40], r2^2 in [2^-78, 2^80], and
//       sqrt : r = rsq(x); s = x*r; h = r/2; d = fma(-s,s,x); s = fma(d,h,s)
//              == sqrtf(x) for EVERY float in [2^-100, 2^127)   (exhaustive: tools/strict_unit_mass_check.hip; so is LLVM's
//              longer form with the extra e = fma(-h,s,1/2); h = fma(h,e,h); s = fma(s,e,s) step, which rounds 1-3 used)
//       div  : r = rcp(d); e = fma(-d,r,1); r = fma(e,r,r); q = n*r; e = fma(-d,q,n); q = fma(e,r,q)
//              == n/d for EVERY pair of significands (2^46 quotients, tools/strict_divide_exhaustive.hip,
//              profiles/round2_strict_divide_exhaustive.txt; every step scales exactly with the operands' exponents inside the
//              window, v_rcp_f32 included, so that covers all operands).  hipcc's own sequence corrects the quotient a second
//              time (e = fma(-d,q,n); q = fma(e,r,q) again): also exact, never needed.
//     23 packed ops + 6 adds + 4 transcendentals per two interactions.
//       unit : a chunk whose masses are all exactly 1.0f (every start-up configuration of the reference) needs 1/d, not m/d:
//              r = rcp(d); e = fma(-d,r,1); r = fma(e,r,r) == 1.0f/d for EVERY float d in [2^-100, 2^101) (exhaustive, same
//              tool, profiles/round2_strict_unit_mass_check.txt): 20 packed ops + 6 adds + 4 transcendentals.
//   * fast (fp64): one interaction at a time (no packed fp64), the same scaling-free divide and sqrt with its own window.
#include "nbody_kernels.h"

namespace nb {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

// sqrtf/sqrt lower to llvm.sqrt and `/` to fdiv, both expanded correctly rounded under hipcc's defaults.
// (NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that is the 1-ulp native v_sqrt_f32.)
__device__ __forceinline__ float  sqrt_T(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_T(double x) { return sqrt(x); }

// r2 as the CPU path forms it:
//   fp32  bodysystemcpu.cpp:186-188   ((eps2 + dx2) + dy2) + dz2
//   fp64  bodysystemcpu.cpp:262-266   (dx2 + dy2) + (dz2 + eps2)
__device__ __forceinline__ float  r2_T(float dx2, float dy2, float dz2, float eps2) { return ((eps2 + dx2) + dy2) + dz2; }
__device__ __forceinline__ double r2_T(double dx2, double dy2, double dz2, double eps2) { return (dx2 + dy2) + (dz2 + eps2); }

// one interaction, generic form (any operand values)
template <typename T> __device__ __forceinline__ void interact_generic(const typename V4<T>::type bj, T pix, T piy, T piz, T& ax, T& ay, T& az, T eps2) {
    const T dx  = bj.x - pix;
    const T dy  = bj.y - piy;
    const T dz  = bj.z - piz;
    const T dx2 = dx * dx;
    const T dy2 = dy * dy;
    const T dz2 = dz * dz;
    const T r2  = r2_T(dx2, dy2, dz2, eps2);
    const T r   = sqrt_T(r2);
    const T mr4 = bj.w / (r2 * r2);
    const T mr3 = mr4 * r;
    ax          = ax + mr3 * dx;  // contraction is off: mul, then add (bodysystemcpu.cpp:200-210 / :278-280)
    ay          = ay + mr3 * dy;
    az          = az + mr3 * dz;
}

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// 2*U interactions: the lane's body i against U pairs {j, j+1} of consecutive bodies, fast form; valid inside the operand
// window only.  Written stage by stage over the U independent pairs so that their dependent chains interleave (the
// divide and sqrt chains are ~20 dependent operations long); the running sums take the 2*U results in j order.
// UNIT: every mass of the pairs is exactly 1.0f (bm is not read).
template <int U, bool UNIT>
__device__ __forceinline__ void interact_jpairs_fast(const v2f (&bx)[U], const v2f (&by)[U], const v2f (&bz)[U], const v2f (&bm)[U], float pix, float piy, float piz, float& ax, float& ay, float& az, v2f eps2) {
    const v2f half = {0.5f, 0.5f}, one = {1.0f, 1.0f};
    v2f dx[U], dy[U], dz[U], x[U], r[U], mr3[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        dx[u] = bx[u] - v2f{pix, pix};
        dy[u] = by[u] - v2f{piy, piy};
        dz[u] = bz[u] - v2f{piz, piz};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = ((eps2 + dx[u] * dx[u]) + dy[u] * dy[u]) + dz[u] * dz[u];  // r2
    {   // r = sqrt(r2), correctly rounded
        v2f s[U], h[U], dd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const v2f rs = v2f{__builtin_amdgcn_rsqf(x[u].x), __builtin_amdgcn_rsqf(x[u].y)};
            s[u]         = x[u] * rs;
            h[u]         = rs * half;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) dd[u] = pk_fma(-s[u], s[u], x[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = pk_fma(dd[u], h[u], s[u]);
    }
    {   // mr4 = m / (r2*r2), correctly rounded; mr3 = mr4 * r
        v2f d[U], rc[U], q[U], e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u]  = x[u] * x[u];
            rc[u] = v2f{__builtin_amdgcn_rcpf(d[u].x), __builtin_amdgcn_rcpf(d[u].y)};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = pk_fma(-d[u], rc[u], one);
#pragma unroll
        for (int u = 0; u < U; ++u) rc[u] = pk_fma(e[u], rc[u], rc[u]);
        if constexpr (UNIT) {  // 1/d: the Newton step above already gave the correctly rounded reciprocal
#pragma unroll
            for (int u = 0; u < U; ++u) mr3[u] = rc[u] * r[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = bm[u] * rc[u];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = pk_fma(-d[u], q[u], bm[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) mr3[u] = pk_fma(e[u], rc[u], q[u]) * r[u];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const v2f tx = mr3[u] * dx[u], ty = mr3[u] * dy[u], tz = mr3[u] * dz[u];
        ax = (ax + tx.x) + tx.y;
        ay = (ay + ty.x) + ty.y;
        az = (az + tz.x) + tz.y;
    }
}

// fp64 has no packed form, but the same scaling-free divide and sqrt apply: hipcc's own sequences (v_rcp_f64 + two Newton
// steps + quotient + one residual correction; v_rsq_f64 + Goldschmidt step + two residual corrections) without
// v_div_scale / v_div_fmas' scaling / v_div_fixup / the 2^256 pre-scaling and the class check of sqrt, all of which are the
// identity inside the window: |coordinate| <= 2^100, softening^2 in [2^-100, 2^100], mass +0 or 2^-100 <= |m| <= 2^100
// (r2 in [2^-100, 2^203], r2^2 in [2^-200, 2^406], quotient in [2^-506, 2^300]).  Checked on 1.7e10 random + structured
// operands and by asking v_div_scale_f64 itself over the window's exponent range (tools/strict_fastpath_check.hip).
__device__ __forceinline__ double fast_sqrt_f64(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double       g = x * y;
    double       h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g              = __builtin_fma(g, r, g);
    h              = __builtin_fma(h, r, h);
    double d       = __builtin_fma(-g, g, x);
    g              = __builtin_fma(d, h, g);
    d              = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double fast_div_f64(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r        = __builtin_fma(r, e, r);
    e        = __builtin_fma(-d, r, 1.0);
    r        = __builtin_fma(r, e, r);
    const double q = n * r;
    e              = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}
// UNIT: the body's mass is exactly 1.0 -- the same sequence with n = 1, where q = n*r is r itself (bj.w is not read)
template <bool UNIT> __device__ __forceinline__ void interact_fast_f64(const double4 bj, double pix, double piy, double piz, double& ax, double& ay, double& az, double eps2) {
    const double dx  = bj.x - pix;
    const double dy  = bj.y - piy;
    const double dz  = bj.z - piz;
    const double r2  = r2_T(dx * dx, dy * dy, dz * dz, eps2);
    const double r   = fast_sqrt_f64(r2);
    const double mr4 = fast_div_f64(UNIT ? 1.0 : bj.w, r2 * r2);
    const double mr3 = mr4 * r;
    ax               = ax + mr3 * dx;
    ay               = ay + mr3 * dy;
    az               = az + mr3 * dz;
}
__device__ __forceinline__ bool coord_in_window(double c) { return __builtin_fabs(c) <= 0x1p100; }  // false for NaN / inf
__device__ __forceinline__ bool mass_in_window(double m) {
    const double a = __builtin_fabs(m);
    return __double_as_longlong(m) == 0ll || (a >= 0x1p-100 && a <= 0x1p100);
}
__device__ __forceinline__ bool softening_in_window(float e2) { return e2 >= 0x1p-39f && e2 <= 0x1p38f; }
__device__ __forceinline__ bool softening_in_window(double e2) { return e2 >= 0x1p-100 && e2 <= 0x1p100; }

// operand window of the fast form (see the header)
__device__ __forceinline__ bool coord_in_window(float c) { return __builtin_fabsf(c) <= 0x1p18f; }  // false for NaN / inf
__device__ __forceinline__ bool mass_in_window(float m) {
    const float a = __builtin_fabsf(m);
    return __float_as_uint(m) == 0u || (a >= 0x1p-40f && a <= 0x1p40f);  // -0 excluded: the sequence returns +0 for it
}

// A wave's LDS traffic is ordered, so data a wave writes for ITSELF needs no s_barrier.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

constexpr int kChunk = 128;  // bodies j per wave and ring slot, two per lane (64: +2.3 % time, 256: -0.7 % but fp64 rings would halve the occupancy)
constexpr int kPerLane = kChunk / 64;

// The ring holds the chunk as x[64] y[64] z[64] m[64] (so that {j, j+1} of one component is one aligned 8-byte broadcast read).
template <typename T> __global__ __launch_bounds__(512, 4) void integrate_bodies_strict(Shard<T> s) {
    using vec4 = typename V4<T>::type;
    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];

    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const unsigned p    = blockDim.x;
    const unsigned tid  = threadIdx.x;
    const unsigned lane = tid & 63u;
    T* ring = reinterpret_cast<T*>(smem_raw) + (tid >> 6) * (2 * 4 * kChunk);  // this wave's [2][4][kChunk]

    const unsigned local  = blockIdx.x * p + tid;
    const bool     active = local < s.i_count;
    const unsigned i      = s.i_begin + (active ? local : s.i_count - 1);
    const vec4     pi     = old_pos[i];
    T              ax = 0, ay = 0, az = 0;
    if (s.acc_in) {
        const vec4 a = reinterpret_cast<const vec4*>(s.acc)[i];
        ax = a.x, ay = a.y, az = a.z;
    }
    const T eps2 = s.eps2;

    // fast form: decided per wave for the bodies i ...
    bool wave_in_window;
    {
        const bool mine = coord_in_window(pi.x) && coord_in_window(pi.y) && coord_in_window(pi.z);
        wave_in_window  = __builtin_amdgcn_ballot_w64(!mine) == 0 && softening_in_window(eps2);
    }

    // The SIMD arbiter is strictly oldest-first, and one wave alone reaches only 3/4 of a SIMD's issue rate: without help
    // the older of the two waves a 512-thread workgroup puts on each SIMD finishes well before the younger, which runs
    // the rest alone.  As in the FAST kernel (nbody_fast.hip), each wave publishes its chunk count and the one that is not
    // ahead of its SIMD mates (HW_ID.SIMD_ID) runs at priority 3, the other at 0.
    unsigned* const    balance    = reinterpret_cast<unsigned*>(smem_raw + static_cast<size_t>(p / 64) * 2 * 4 * kChunk * sizeof(T));
    unsigned* const    simd_count = balance;                                            // [4]
    volatile unsigned* progress   = reinterpret_cast<volatile unsigned*>(balance + 4);  // [4][8]
    if (tid < 36) balance[tid] = tid < 4 ? 0u : 0xffffffffu;
    __syncthreads();
    const unsigned simd = static_cast<unsigned>(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4));  // HW_REG_HW_ID[5:4]
    unsigned       slot = 0;
    if (lane == 0) slot = atomicAdd(&simd_count[simd], 1u);
    slot                = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(slot))) & 7u;
    volatile unsigned* const mine = progress + simd * 8;
    if (lane == 0) mine[slot] = 0;

    const unsigned n_chunks = (s.j_count + kChunk - 1) / kChunk;
    struct Loaded {
        vec4 v[kPerLane];
    };
    auto load_chunk = [&](unsigned c) -> Loaded {
        Loaded out;
#pragma unroll
        for (int r = 0; r < kPerLane; ++r) {
            const unsigned j = c * kChunk + r * 64 + lane;
            vec4           v;
            v.x = v.y = v.z = v.w = 0;
            if (j < s.j_count) v = old_pos[s.j_begin + j];
            out.v[r] = v;
        }
        return out;
    };
    // ... and per chunk for the bodies j (slots past the end of the range hold zeros and are never visited)
    // (returns 0: outside the window, 1: inside, 2: inside and every mass of the chunk is exactly 1)
    auto store_chunk = [&](int buf, unsigned c, const Loaded& loaded) -> int {
        bool ok = true, unit = true;
#pragma unroll
        for (int r = 0; r < kPerLane; ++r) {
            const vec4 v   = loaded.v[r];
            T*         dst = ring + buf * (4 * kChunk) + r * 64 + lane;
            dst[0 * kChunk] = v.x, dst[1 * kChunk] = v.y, dst[2 * kChunk] = v.z, dst[3 * kChunk] = v.w;
            ok   = ok && coord_in_window(v.x) && coord_in_window(v.y) && coord_in_window(v.z) && mass_in_window(v.w);
            unit = unit && (v.w == T(1) || c * kChunk + r * 64 + lane >= s.j_count);
        }
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) return 0;
        return __builtin_amdgcn_ballot_w64(!unit) == 0 ? 2 : 1;
    };

    int    chunk_form = 0;
    Loaded next;
    if (n_chunks > 0) {
        next       = load_chunk(0);
        chunk_form = store_chunk(0, 0, next);
    }
    wave_lds_sync();

    int cur = 0;
    for (unsigned c = 0; c < n_chunks; ++c) {
        const bool have_next = (c + 1) < n_chunks;
        if (have_next) next = load_chunk(c + 1);  // in flight across the compute below
        {
            unsigned least = c;
#pragma unroll
            for (int q = 0; q < 8; ++q) least = min(least, mine[q]);  // unsynchronised reads: a stale value only delays a priority change
            if (static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(least))) >= c) {
                __builtin_amdgcn_s_setprio(3);
            } else {
                __builtin_amdgcn_s_setprio(0);
            }
        }
        const unsigned cnt = min(static_cast<unsigned>(kChunk), s.j_count - c * kChunk);
        const T* __restrict__ cx = ring + cur * (4 * kChunk);
        const T* __restrict__ cy = cx + kChunk;
        const T* __restrict__ cz = cy + kChunk;
        const T* __restrict__ cm = cz + kChunk;

        unsigned k = 0;
        if constexpr (sizeof(T) == 4) {
            if (wave_in_window && chunk_form != 0) {
                const v2f e2 = {eps2, eps2};
                constexpr int U = 4;  // pairs in flight
                if (chunk_form == 2) {  // unit masses: the reciprocal form, the masses are not even read
#pragma unroll 1
                    for (; k + 2 * U <= cnt; k += 2 * U) {
                        v2f bx[U], by[U], bz[U], bm[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            bx[u] = *reinterpret_cast<const v2f*>(cx + k + 2 * u), by[u] = *reinterpret_cast<const v2f*>(cy + k + 2 * u);
                            bz[u] = *reinterpret_cast<const v2f*>(cz + k + 2 * u), bm[u] = v2f{1.0f, 1.0f};
                        }
                        interact_jpairs_fast<U, true>(bx, by, bz, bm, pi.x, pi.y, pi.z, ax, ay, az, e2);
                    }
                } else {
#pragma unroll 1
                    for (; k + 2 * U <= cnt; k += 2 * U) {
                        v2f bx[U], by[U], bz[U], bm[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            bx[u] = *reinterpret_cast<const v2f*>(cx + k + 2 * u), by[u] = *reinterpret_cast<const v2f*>(cy + k + 2 * u);
                            bz[u] = *reinterpret_cast<const v2f*>(cz + k + 2 * u), bm[u] = *reinterpret_cast<const v2f*>(cm + k + 2 * u);
                        }
                        interact_jpairs_fast<U, false>(bx, by, bz, bm, pi.x, pi.y, pi.z, ax, ay, az, e2);
                    }
                }
#pragma unroll 1
                for (; k + 2 <= cnt; k += 2) {  // ragged chunk: pair by pair
                    const v2f bx[1] = {*reinterpret_cast<const v2f*>(cx + k)}, by[1] = {*reinterpret_cast<const v2f*>(cy + k)};
                    const v2f bz[1] = {*reinterpret_cast<const v2f*>(cz + k)}, bm[1] = {*reinterpret_cast<const v2f*>(cm + k)};
                    interact_jpairs_fast<1, false>(bx, by, bz, bm, pi.x, pi.y, pi.z, ax, ay, az, e2);
                }
            }
        }
        if constexpr (sizeof(T) == 8) {
            if (wave_in_window && chunk_form == 2) {
#pragma unroll 4
                for (; k < cnt; ++k) {
                    vec4 bj;
                    bj.x = cx[k], bj.y = cy[k], bj.z = cz[k], bj.w = 1;
                    interact_fast_f64<true>(bj, pi.x, pi.y, pi.z, ax, ay, az, eps2);
                }
            } else if (wave_in_window && chunk_form != 0) {
#pragma unroll 4
                for (; k < cnt; ++k) {
                    vec4 bj;
                    bj.x = cx[k], bj.y = cy[k], bj.z = cz[k], bj.w = cm[k];
                    interact_fast_f64<false>(bj, pi.x, pi.y, pi.z, ax, ay, az, eps2);
                }
            }
        }
        // generic form: the whole chunk, or the odd body at the end of a ragged one
#pragma unroll 4
        for (; k < cnt; ++k) {
            vec4 bj;
            bj.x = cx[k], bj.y = cy[k], bj.z = cz[k], bj.w = cm[k];
            interact_generic<T>(bj, pi.x, pi.y, pi.z, ax, ay, az, eps2);
        }

        if (have_next) chunk_form = store_chunk(cur ^ 1, c + 1, next);
        if (lane == 0) mine[slot] = c + 1;
        wave_lds_sync();
        cur ^= 1;
    }
    if (lane == 0) mine[slot] = 0xffffffffu;  // finished: never the one the others defer to
    __builtin_amdgcn_s_setprio(0);

    if (!active) return;
    if (s.finalize) {
        // bodysystemcpu.cpp:228-234 (fp32) / :283-298 (fp64): dv = acc*dt; v = (v + dv)*damping; p += v*dt
        vec4 v  = reinterpret_cast<const vec4*>(s.vel)[i];
        vec4 pn = pi;
        const T dvx = ax * s.dt, dvy = ay * s.dt, dvz = az * s.dt;
        v.x = (v.x + dvx) * s.damping;
        v.y = (v.y + dvy) * s.damping;
        v.z = (v.z + dvz) * s.damping;
        pn.x = pn.x + v.x * s.dt;
        pn.y = pn.y + v.y * s.dt;
        pn.z = pn.z + v.z * s.dt;
        reinterpret_cast<vec4*>(s.new_pos)[i] = pn;
        reinterpret_cast<vec4*>(s.vel)[i]     = v;
    } else {
        vec4 a;
        a.x = ax, a.y = ay, a.z = az, a.w = 0;
        reinterpret_cast<vec4*>(s.acc)[i] = a;
    }
}

}  // namespace

// Geometry: results do not depend on it, so the library picks it (the caller's --blockSize is validated by the C-ABI and
// otherwise a hint).  512-thread workgroups (two waves per SIMD each, kept level by the priority scheme; two of them
// share a CU) while that still gives every CU at least one; smaller workgroups for smaller shards so that the bodies
// spread over all CUs.
template <typename T> hipError_t launch_strict(const Shard<T>& s, int block_size, int cu_count, hipStream_t stream, bool prepare_only) {
    (void)block_size;
    unsigned p = 512;
    while (p > 64 && (s.i_count + p - 1) / p < static_cast<unsigned>(cu_count)) p /= 2;
    const unsigned blocks = (s.i_count + p - 1) / p;
    const size_t   smem   = static_cast<size_t>(p / 64) * 2 * kChunk * 4 * sizeof(T) + 256;  // the waves' rings + progress words
    if (smem > 64u * 1024u) {  // fp64 at p = 512: 65 792 B
        if (const auto err = allow_large_lds<&integrate_bodies_strict<T>>(); err != hipSuccess) return err;
    }
    if (prepare_only) return hipSuccess;  // graph capture arms the attribute before hipStreamBeginCapture
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
    hipLaunchKernelGGL(integrate_bodies_strict<T>, dim3(blocks), dim3(p), smem, stream, s);
    return hipGetLastError();
}

template hipError_t launch_strict<float>(const Shard<float>&, int, int, hipStream_t, bool);
template hipError_t launch_strict<double>(const Shard<double>&, int, int, hipStream_t, bool);

}  // namespace nb
