// nbody_pair.hip -- FAST, pairwise layout (round 3): every unordered pair of bodies is evaluated ONCE and applied to both
// bodies (Newton's third law), instead of once per direction as the reference kernel does
// (/root/reference/src/nbody/bodysystemcuda.cu:98-146 evaluates all N^2 directed interactions).  gfx950 (CDNA4) only.
//
// Why: the one-sided loop of nbody_fast.hip sits at its instruction floor -- 11 v_pk_* + 2 v_rsq_f32 per packed pair of bodies
// i and body j = 61.5 SIMD cycles per 2 directed interactions and lane (30.75 each), 62 % of the "20 flop" fp32 peak.  The
// only lever left is the number of evaluations.  A pair evaluation that also feeds the reaction sums of the body j costs
// 3 more v_pk_fma; what it needs is a place for those sums that a lane can reach:
//
//   * A wave holds 64*I bodies i for good (I = R*W per lane: R packed pairs in fp32, R doubles in fp64), exactly as the
//     one-sided kernel does.  The bodies j come 64 at a time, ONE PER LANE (a coalesced vector load), together with three
//     reaction sums per lane, and ROTATE through the wave: after each step everything that belongs to the body j moves on
//     by one lane with DPP wave_ror:1 (a full 64-lane rotation exists on gfx9/CDNA; checked on the chip,
//     tools/wave_ror_check.hip).  After 64 steps every body i of the wave has met every body j of the tile and the
//     reaction sums are back in their home lanes.  Per step and lane: R x (14 v_pk_* + 2 v_rsq_f32) + 9 v_mov_b32_dpp for
//     4R directed interactions; measured in isolation (tools/sym_microbench.hip, profiles/round3_pairwise_loop_microbench.txt)
//     275 SIMD cycles per step at R = 4 with 4 waves per SIMD = 17.2 cycles per directed interaction against 30.75.
//     Round 4: R = 8 from 65 536 bodies on -- the nine moves amortised over twice the arithmetic (4.30 instead of 4.56 vector
//     instructions per directed interaction), 256 VGPRs at two waves per SIMD, 5 % faster than R = 4 at four.
//   * Work: the bodies are cut into blocks of 64*I; block pair (a, a+q) is evaluated by the workgroup(s) of block a for
//     q = 0 .. NB/2 (indices mod NB) -- a round-robin tournament, every workgroup gets the same amount.  q = 0 (the block with
//     itself) and, for an even block count, q = NB/2 (which both partners list) run the same loop but keep only the i side.
//     The S waves of a workgroup (and the C workgroups of a block, for small systems) share the bodies i and split the
//     64-body tiles of those blocks: unit u -> slot u mod (C*S), slot = wave * C + workgroup, static, so every sum is formed in the
//     same order in every run.
//   * The reaction sums of a tile leave the wave once, after its 64 steps: 3 coalesced stores into the caller's WORKSPACE,
//     slot q-1 of body j (each (slot, body) is written by exactly one wave per step: no atomics, no zeroing).  The i-side
//     sums are folded over the S waves through LDS in a fixed order (as in nbody_fast.hip) and stored to the workspace too.
//     A second kernel adds a body's slots in a fixed order and integrates (integrateBodies, bodysystemcuda.cu:166-183).
//     Workspace traffic: 12 B per tile visit and body = N^2 / (128 I) * 12 B per step (0.8 GB at 262 144 bodies with I = 8, 0.4 GB with I = 16, written once,
//     read once: ~0.3 ms of a ~8 ms step).
//   * Masses: a tile whose 64 bodies j are ONE species (one mass) multiplies nothing on the i side -- the wave's i-side sums are
//     kept in units of the mass of the species it is working through (re-expressed once when that changes); a block whose 64*I
//     bodies i are one species multiplies nothing on the reaction side -- the block's mass is applied when the sums are stored.
//     Every unit of an equal-mass system such as the reference's start-up configurations, and all but the border tiles of a
//     galaxy file (species in contiguous blocks), run the loop without any mass multiply; mixed bodies j carry m_j / unit along
//     (+1 v_pk_mul per pair, one more rotation), mixed bodies i multiply the reaction side (+1).  Bodies beyond N (ragged last
//     block) are zero-mass bodies at a real body's place: they pull nothing, and what they feel is dropped.
//   * Several GPUs (nbody_comm.hip): the same kernel takes a RANGE of bodies i and either that range again (diag: the
//     tournament within a rank's slice) or a range of bodies j (a rectangle of the pair matrix against another rank's slice,
//     every tile symmetric, reaction slot = the block of bodies i); pair_reduce folds a rectangle's reaction planes into the
//     one array that travels to the owner of those bodies, and pair_finish also adds the arrays a rank received.
//
// Results differ from the one-sided FAST kernel in summation order only; both are held to an fp64 direct sum by the tests.
#include "nbody_kernels.h"

#include <algorithm>
#include <cmath>
#include <vector>

#ifdef NB_PAIR_STAMPS
// Diagnostic build only (tools/build_pair_variant.sh stamps "-DNB_PAIR_STAMPS", read by tools/pair_stamps.py): per wave of the LAST
// pair_forces launch, {start, entry of the unit loop, its exit} in s_memtime ticks + where the hardware put the wave + (round 6) start
// and exit on the constant 100 MHz counter (s_memrealtime): exit - start on the two counters is the IN-KERNEL clock (tools/inkernel_clock.py).
__device__ unsigned long long nb_pair_stamps[8192 * 8];
extern "C" __attribute__((visibility("default"))) int nb_debug_read_pair_stamps(void* host, size_t bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(nb_pair_stamps), bytes, 0, hipMemcpyDeviceToHost));
}
#endif

// The workspace planes (reaction sums, i-side sums) are written once by pair_forces and read once by pair_finish / pair_reduce.  With
// -DNB_PAIR_NONTEMPORAL (tools/build_pair_variant.sh; the A/B of round 6: profiles/round6_nontemporal_ab.txt) those accesses carry the
// non-temporal hint, so that the streaming planes do not evict the 4 MiB position array from each XCD's L2.  Without the flag: plain accesses.
#ifdef NB_PAIR_NONTEMPORAL
#define NB_WS_STORE(ptr, value) __builtin_nontemporal_store((value), (ptr))
#define NB_WS_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define NB_WS_STORE(ptr, value) (*(ptr) = (value))
#define NB_WS_LOAD(ptr) (*(ptr))
#endif

namespace nb {
namespace {

#include "nbody_lane.h"

// ---- rotation of whatever belongs to the body j: one lane onward (lane l reads lane l-1; lane 0 reads lane 63) --------------
constexpr int kWaveRor1 = 0x13C;
// (The same rotation on the LDS crossbar -- ds_bpermute_b32, off the vector ALU -- was tried: 10.36 against 10.08 ms, same box.)
__device__ __forceinline__ int rotate_bits(int v) { return __builtin_amdgcn_update_dpp(v, v, kWaveRor1, 0xf, 0xf, false); }
__device__ __forceinline__ float rotate(float x) { return __builtin_bit_cast(float, rotate_bits(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ double rotate(double x) {
    const unsigned long long b  = __builtin_bit_cast(unsigned long long, x);
    const unsigned           l2 = static_cast<unsigned>(rotate_bits(static_cast<int>(b & 0xffffffffull)));
    const unsigned           h2 = static_cast<unsigned>(rotate_bits(static_cast<int>(b >> 32)));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(h2) << 32) | l2);
}
__device__ __forceinline__ v2f rotate(v2f x) { return v2f{rotate(x.x), rotate(x.y)}; }

// lane 0's value, as a wave-uniform (scalar) value
__device__ __forceinline__ float first_lane(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ double first_lane(double x) {
    const unsigned long long b  = __builtin_bit_cast(unsigned long long, x);
    const unsigned           lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(b & 0xffffffffull)));
    const unsigned           hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32)));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}

__device__ __forceinline__ float  both_halves(v2f a) { return a.x + a.y; }
__device__ __forceinline__ double both_halves(double a) { return a; }

// A mass the sums may be expressed in units of: 1/m is a well-behaved number (false for 0, NaN, infinities)
template <typename T> __device__ __forceinline__ bool usable_unit(T m) {
    const T a = m < 0 ? -m : m;
    return a >= T(0x1p-60) && a <= T(0x1p60);
}

// T: float|double   R: vectors per lane (I = R*W bodies i)   S: waves of a workgroup
// Registers and occupancy: up to R = 4 vectors per lane (fp32: eight bodies i) fit 128 VGPRs -> four waves per SIMD.  R = 8 (round 4:
// sixteen bodies i per lane in fp32, the nine rotation moves of a step amortised over twice the arithmetic -- 4.30 instead of 4.56
// vector instructions per directed interaction) takes 256 VGPRs at two waves per SIMD (one 8-wave workgroup per CU), or ~168 with
// spills outside the rotation loops at three (one 12-wave workgroup per CU: used where twelve waves divide a block's units evenly).
template <int R, int S> constexpr int kPairWavesPerSimd = R <= 4 ? 4 : (S == 12 ? 3 : 2);
#define NB_PAIR_KERNEL pair_forces
#define NB_PAIR_CLOCK_PARAM
#define NB_PAIR_CLOCKED 0
#include "nbody_pair_forces.inc"
#undef NB_PAIR_KERNEL
#undef NB_PAIR_CLOCK_PARAM
#undef NB_PAIR_CLOCKED
#define NB_PAIR_KERNEL pair_forces_clocked
#define NB_PAIR_CLOCK_PARAM , unsigned long long* clock_words
#define NB_PAIR_CLOCKED 1
#include "nbody_pair_forces.inc"
#undef NB_PAIR_KERNEL
#undef NB_PAIR_CLOCK_PARAM
#undef NB_PAIR_CLOCKED

// The three component sums over `slots` reaction planes, q ascending, as four interleaved running sums (shorter chains, smaller
// rounding error), wave w of the 256-thread workgroup taking q = w, w+4, ...; the caller combines (t0 + t1) + (t2 + t3).
template <typename T> __device__ __forceinline__ void quarter_sums(const T* r, size_t plane, unsigned slots, unsigned wave, T (&t)[3]) {
    // r: component 0 of slot 0 for this body; component c of slot q at r[(q * 3 + c) * plane]
    const unsigned full = slots & ~3u;
    t[0] = t[1] = t[2] = 0;
    // 24 loads in flight per lane, also for short slot lists (a missing slot adds +0, which changes no bit); the adds keep their order
    for (unsigned q = wave; q < full; q += 32) {
        T v[3][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool there = (q + 4 * u) < full;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c][u] = there ? NB_WS_LOAD(r + (static_cast<size_t>(q + 4 * u) * 3 + c) * plane) : T(0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] += v[c][u];
        }
    }
    if (wave == 0) {
        for (unsigned k = full; k < slots; ++k) {  // the odd slots join t0
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] += NB_WS_LOAD(r + (static_cast<size_t>(k) * 3 + c) * plane);
        }
    }
}

// The last kernel of a step: a body's i-side sums and reaction sums, added in a fixed order, then integrateBodies
// (bodysystemcuda.cu:166-183): v = (v + a*dt)*damping; p += v*dt.  A 256-thread workgroup takes 64 bodies (four waves = the
// four interleaved sums over the reaction slots, combined through LDS: the same order for any launch geometry).
template <typename T> __global__ __launch_bounds__(256) void pair_finish(FinishArgs<T> s) {
    using vec4 = typename Lane<T>::vec4;
    __shared__ T   part[3][3][64];  // [wave 1..3][component][body]
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned k    = blockIdx.x * 64 + lane;  // relative to the rank's first body
    const bool     live = k < s.count;
    T              t[3] = {0, 0, 0};
    // Wave 0 finishes the body: what it needs besides the reaction sums -- the body's own sums, its velocity and position -- is
    // requested BEFORE the slot sums and the barrier, so that those loads are in flight together with the slot loads instead of one
    // latency after the other (a finish launch at 16 384 bodies is latency, not bandwidth: 6 MB in 8 us).
    const unsigned body   = s.origin + k;
    const bool     closer = wave == 0 && live;
    vec4           v{}, pn{}, e{};
    T              own[3] = {0, 0, 0};
    if (closer) {
        v  = reinterpret_cast<const vec4*>(s.vel)[body];
        pn = reinterpret_cast<const vec4*>(s.old_pos)[body];
        if (s.extra != nullptr) e = reinterpret_cast<const vec4*>(s.extra)[body];
        // the body's own sums: one plane per workgroup that shared its block (up to 16), added in slot order; the loads go out four
        // slots x three components at a time (a chain of dependent loads cost 10 us of a 34 us finish at C = 13)
        for (unsigned m = 0; m < s.n_self; ++m) {
            const auto& set = s.self_set[m];
            if (k < set.first || k - set.first >= set.count) continue;
            for (unsigned c0 = 0; c0 < set.slots; c0 += 4) {
                T x[4][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int comp = 0; comp < 3; ++comp) x[i][comp] = (c0 + i) < set.slots ? NB_WS_LOAD(s.self + (static_cast<size_t>(set.slot + c0 + i) * 3 + comp) * s.self_plane + k) : T(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((c0 + i) < set.slots) {  // (a missing slot must not add +0 to a -0 sum: keep the additions exactly those of the plain loop)
#pragma unroll
                        for (int comp = 0; comp < 3; ++comp) own[comp] += x[i][comp];
                    }
                }
            }
        }
    }
    if (live) quarter_sums(s.react + k, s.react_plane, s.react_slots, wave, t);
    if (wave != 0) {
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) part[wave - 1][comp][lane] = t[comp];
    }
    __syncthreads();
    if (!closer) return;
    // what arrived from the other ranks / slices (up to kMaxRecv arrays): all loads first, then the additions in array order
    T    got[kMaxRecv][3];
    bool has[kMaxRecv];
#pragma unroll
    for (int m = 0; m < kMaxRecv; ++m) {
        has[m] = static_cast<unsigned>(m) < s.n_recv && k >= s.recv_set[m].first && k - s.recv_set[m].first < s.recv_set[m].count;
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) got[m][comp] = has[m] ? s.recv[(static_cast<size_t>(m) * 3 + comp) * s.recv_plane + k] : T(0);
    }
    T f[3];
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        T others = (t[comp] + part[0][comp][lane]) + (part[1][comp][lane] + part[2][comp][lane]);
#pragma unroll
        for (int m = 0; m < kMaxRecv; ++m) {
            if (has[m]) others += got[m][comp];
        }
        f[comp] = own[comp] - others;  // d = p_j - p_i: what body j feels from body i is -m_i d w
    }
    if (s.extra != nullptr) f[0] += e.x, f[1] += e.y, f[2] += e.z;
    v.x     = __builtin_fma(f[0], s.dt, v.x) * s.damping;
    v.y     = __builtin_fma(f[1], s.dt, v.y) * s.damping;
    v.z     = __builtin_fma(f[2], s.dt, v.z) * s.damping;
    pn.x    = __builtin_fma(v.x, s.dt, pn.x);
    pn.y    = __builtin_fma(v.y, s.dt, pn.y);
    pn.z    = __builtin_fma(v.z, s.dt, pn.z);
    reinterpret_cast<vec4*>(s.new_pos)[body] = pn;
    reinterpret_cast<vec4*>(s.vel)[body]     = v;
}

// Multi-GPU: the reaction sums of a rectangle, one plane per block of bodies i, folded into ONE array per body j (what
// travels to the rank that owns those bodies): out[comp][j] = sum over the slots, same fixed order as above.
template <typename T> __global__ __launch_bounds__(256) void pair_reduce(const T* react, unsigned react_plane, unsigned slots, T* out, unsigned out_plane, unsigned count) {
    __shared__ T   part[3][3][64];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned k    = blockIdx.x * 64 + lane;
    const bool     live = k < count;
    T              t[3] = {0, 0, 0};
    if (live) quarter_sums(react + k, react_plane, slots, wave, t);
    if (wave != 0) {
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) part[wave - 1][comp][lane] = t[comp];
    }
    __syncthreads();
    if (wave != 0 || !live) return;
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) out[static_cast<size_t>(comp) * out_plane + k] = (t[comp] + part[0][comp][lane]) + (part[1][comp][lane] + part[2][comp][lane]);
}

template <typename T, int R, int S> hipError_t launch_rs(const PairArgs<T>& args, unsigned grid, unsigned lds_bytes, hipStream_t stream, bool prepare_only) {
    if (lds_bytes > 64u * 1024u) {
        if (const auto err = allow_large_lds<&pair_forces<T, R, S>>(); err != hipSuccess) return err;
    }
    unsigned long long* clock_words = nullptr;
    if constexpr (R == 8 && S == 8) {  // nb_set_pair_clock_words: two 64-bit words per workgroup of this launch, when there is room for them
        size_t capacity = 0;
        unsigned long long* lent = pair_clock_words(&capacity);
        if (lent != nullptr && capacity >= static_cast<size_t>(grid) * 16) clock_words = lent;
        if (lds_bytes > 64u * 1024u && clock_words != nullptr) {
            if (const auto err = allow_large_lds<&pair_forces_clocked<T, R, S>>(); err != hipSuccess) return err;
        }
    }
    if (prepare_only) return hipSuccess;
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
    if constexpr (R == 8 && S == 8) {
        if (clock_words != nullptr) {
            hipLaunchKernelGGL((pair_forces_clocked<T, R, S>), dim3(grid), dim3(64 * S), lds_bytes, stream, args, clock_words);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((pair_forces<T, R, S>), dim3(grid), dim3(64 * S), lds_bytes, stream, args);
    return hipGetLastError();
}

template <typename T, int R> hipError_t launch_r(const PairArgs<T>& args, int waves, unsigned grid, unsigned lds_bytes, hipStream_t stream, bool prepare_only) {
    switch (waves) {
        case 4: return launch_rs<T, R, 4>(args, grid, lds_bytes, stream, prepare_only);
        case 8: return launch_rs<T, R, 8>(args, grid, lds_bytes, stream, prepare_only);
        case 12: return launch_rs<T, R, 12>(args, grid, lds_bytes, stream, prepare_only);
        case 16: return launch_rs<T, R, 16>(args, grid, lds_bytes, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// How a launch fills the chip (what the two functions below model).  A workgroup of eight waves puts two waves on every SIMD of its
// CU, and two runnable waves already take ~97 % of a SIMD's issue slots: a second workgroup on the same CU (R <= 4 fits two)
// does not add throughput, the two share it.  So whatever the residency, a launch costs  max over CUs of (workgroups on it) x (one
// workgroup's time) = ceil(grid / 256) "rounds" for eight waves (measured, round 4: 36 000 bodies as 71 blocks x 4 = 284 workgroups of
// R = 4 took 359 us -- two rounds -- against 181 us for the 256 workgroups of 32 768 bodies; profiles/round4_plan_sweep.txt).
inline double resident_workgroups(int S) { return 256.0 * std::max(1, 8 / S); }  // (four waves: two workgroups saturate a CU)

// The units the busiest SIMD of a block's workgroups works through (eight waves: SIMD s holds waves s and s + 4): U units over C*S
// slots -- floor(U / (C*S)) whole units per wave, and the units left over as quarters, one per SIMD of the workgroup that takes the
// unit (pair_forces, PairArgs::deal == 2): a workgroup with t tail units adds t / 4 to each of its SIMDs.  Without room in the LDS
// for the quarters' sums the units stay whole: the first U mod (C*S) slots take one more, interleaved (slot = wave*C + c: the longer
// waves are the low wave ids of every workgroup) or blocked (slot = c*S + wave), as launch_pair_tile decides.
inline double busiest_simd_units(unsigned U, unsigned C, int S, bool quarters) {
    const unsigned slots = C * static_cast<unsigned>(S), base = U / slots, rem = U % slots;
    const double   per_simd = static_cast<double>(S) / 4;  // waves of a workgroup per SIMD
    if (quarters) return per_simd * base + pair_tail_units(U, C, S) / 4.0;
    const bool     inter = C > 1 && rem != 0 && base < 34;  // (the rule of launch_pair_tile)
    if (rem == 0) return per_simd * base;
    const unsigned longer = (inter || C == 1) ? (rem + C - 1) / C : std::min(rem, static_cast<unsigned>(S));  // longer waves in the worst workgroup
    return per_simd * base + (longer <= 4 ? 1.0 : 2.0);
}

// Geometry of the single-GPU tournament: R (vectors per lane: blocks of 64*R*W bodies) and C (workgroups sharing a block, any
// number up to 16) by a cost model fitted to sweeps of 83 body counts x up to 15 geometries (fp32; 31 x 15 fp64;
// profiles/round4_plan_sweep.txt; tools/pair_plan_times.py):
//     us ~ rounds * ((units of the busiest SIMD) * unit[R] + wg[R])  +  slot * n * (reaction slots + C) * 1e-6  (+ a constant)
// -- rounds and units as above; unit[R] = what 64 rotation steps of two waves cost a SIMD (16R + 9 vector instructions per step);
// wg[R] = a workgroup's set-up and fold (it grows with the bodies i a lane holds); the last term is everything that grows with the
// reaction planes (pair_finish, the stores).  It reproduces the sweeps to 2.6 % (worst 10 %), and fitted on one half of the fp32
// sweep its choice on the other half is within 1.5 % of the best measured geometry on average (worst 8 %) -- the fixed table it
// replaces: 6 % on average and 44 % at 34 000 bodies, where 4 x 67 = 268 workgroups cost two rounds.  The powers of two come out
// as the table had them: R = 2 up to 16 384 bodies, 4 at 32 768, 8 from 65 536 (256 VGPRs, two waves per SIMD, the 9 rotation
// moves amortised over twice the arithmetic); C = 4 / 4 / 4 / 2 / 1 at 16 384 ... 262 144.  In between, C is whatever fills whole
// rounds: 50 000 bodies are 49 blocks of 1 024 -- x 5 = 245 workgroups, one round (0.67 -> 0.80 of the peak); 33 000 bodies
// 0.59 -> 0.71.  Only the ratios of the constants matter (box clocks differ).
template <typename T> struct PairCost;
template <> struct PairCost<float> {  // by R = 1, 2, 4, 8
    // (R = 1 -- two bodies i per lane, 25 instructions for 4 pair evaluations -- pays only where nothing else fills a round: 8 500-10 000
    // bodies, three workgroups per block of 128 bodies, 32 us against 40; its constant is from those sizes alone)
    static constexpr double unit[4] = {2.2, 4.53, 9.17, 17.72}, wg[4] = {0.0, 0.0, 1.45, 6.86}, slot = 14.1;
};
template <> struct PairCost<double> {
    static constexpr double unit[4] = {0.0, 5.12, 10.39, 20.72}, wg[4] = {0.0, 0.0, 0.0, 0.0}, slot = 21.9;  // (unit 0: not a candidate; the sweep does not resolve wg: fitted freely it comes out negative)
};

template <typename T> PairPlan plan_pair(unsigned n, int cu_count, int ovr_r, int ovr_s, int ovr_c) {
    constexpr int W = sizeof(T) == 4 ? 2 : 1;
    (void)cu_count;
    const bool fixed_r = ovr_r == 1 || ovr_r == 2 || ovr_r == 4 || ovr_r == 8;  // (R = 6 was tried: 74 KB of LDS per workgroup and 342 ragged blocks -- 11.8 against 10.2 ms)
    const int  S       = (ovr_s == 4 || ovr_s == 8 || ovr_s == 12 || ovr_s == 16) ? ovr_s : 8;  // (twelve waves -- <float, 8, 12>, 168 VGPRs -- through the override only: with the units interleaved eight do better)
    auto geometry = [&](int R, unsigned C) {
        PairPlan p{};
        const unsigned block  = 64u * static_cast<unsigned>(R * W);
        const unsigned blocks = (n + block - 1) / block;
        p.vectors_per_lane = R;
        p.waves            = S;
        p.splits           = C;
        p.blocks           = blocks;
        p.block_bodies     = block;
        p.slots            = blocks < 2 ? 0u : ((blocks & 1u) ? blocks / 2 : blocks / 2 - 1);
        p.grid_blocks      = blocks * C;
        p.lds_bytes        = pair_lds_bytes(R, W, S, sizeof(T), pair_tail_units((blocks / 2 + 1) * static_cast<unsigned>(R * W), C, S));
        if (p.lds_bytes > kPairLdsLimit) p.lds_bytes = pair_lds_bytes(R, W, S, sizeof(T), 0);  // (no room for the quarters' sums: whole units only)
        p.workspace_bytes  = (static_cast<size_t>(C) + p.slots) * 3 * static_cast<size_t>(blocks) * block * sizeof(T);
        return p;
    };
    auto units_of = [&](const PairPlan& p) { return (p.blocks / 2 + 1) * static_cast<unsigned>(p.vectors_per_lane * W); };
    auto with_units = [&](int R, unsigned C) {  // no wave without a unit
        PairPlan p = geometry(R, C);
        while (p.splits > 1 && units_of(p) < p.splits * static_cast<unsigned>(S)) p = geometry(R, p.splits / 2);
        return p;
    };
    PairPlan best{};
    double   best_cost = 0;
    for (int R : {1, 2, 4, 8}) {  // (near-ties go to the smaller R -- 32 768 bodies: R = 4, C = 4 178 us, R = 8, C = 8 184 us -- and the smaller C)
        const int k = R == 1 ? 0 : (R == 2 ? 1 : (R == 4 ? 2 : 3));
        if (fixed_r ? R != ovr_r : PairCost<T>::unit[k] == 0.0) continue;
        for (unsigned C = 1; C <= 16; ++C) {  // (any C, not only powers of two: 50 000 bodies are 49 blocks of 1 024 -- x 5 = 245 workgroups, one round)
            const PairPlan p = ovr_c > 0 ? with_units(R, static_cast<unsigned>(ovr_c)) : geometry(R, C);
            const unsigned U = units_of(p);
            if (ovr_c <= 0 && C > 1 && U < 2 * C * static_cast<unsigned>(S)) continue;  // every wave at least two units (a workgroup's set-up and fold cost about one)
            const double rounds = std::ceil(p.grid_blocks / resident_workgroups(S));
            const double unit   = PairCost<T>::unit[k] > 0.0 ? PairCost<T>::unit[k] : 0.5 * PairCost<T>::unit[1];  // (an R the search would not take, forced by the override)
            const bool   quarters = pair_lds_bytes(R, W, S, sizeof(T), pair_tail_units(U, p.splits, S)) <= kPairLdsLimit;
            const double cost   = rounds * (busiest_simd_units(U, p.splits, S, quarters) * unit + PairCost<T>::wg[k]) +
                                PairCost<T>::slot * 1e-6 * static_cast<double>(n) * (p.slots + p.splits);
            if (best.blocks == 0 || cost < 0.995 * best_cost) best = p, best_cost = cost;
            if (ovr_c > 0) break;
        }
    }
    return best;
}

template <typename T> hipError_t launch_pair_tile(const PairArgs<T>& args, const PairGeom& g, hipStream_t stream, bool prepare_only) {
    constexpr int  W         = sizeof(T) == 4 ? 2 : 1;
    PairArgs<T>    a         = args;
    const unsigned block     = 64u * static_cast<unsigned>(g.vectors_per_lane * W);
    a.blocks                 = (a.i_count + block - 1) / block;
    a.splits                 = g.splits;
    const unsigned units = a.diag ? (a.unit_count != 0 ? a.unit_count : (a.blocks / 2 + 1) * static_cast<unsigned>(g.vectors_per_lane * W)) : (a.j_count + 63) / 64;
    unsigned       lds_bytes;
    {   // how the units reach the waves (see the kernel): equal whole units + quarters of the rest when the LDS has room for the quarters' sums
        const unsigned slots = g.splits * static_cast<unsigned>(g.waves), each = units / slots;
        const unsigned tail  = pair_tail_units(units, g.splits, g.waves);
        lds_bytes            = pair_lds_bytes(g.vectors_per_lane, W, g.waves, sizeof(T), tail);
#ifdef NB_PAIR_NO_QUARTERS
        const bool quarters = false;
#else
        const bool quarters = lds_bytes <= kPairLdsLimit;
#endif
        if (quarters) {
            a.deal = 2u;
        } else {
            a.deal    = (g.splits > 1 && units % slots != 0 && each < 34) ? 1u : 0u;  // (interleaved where the remainder is worth spreading)
            lds_bytes = pair_lds_bytes(g.vectors_per_lane, W, g.waves, sizeof(T), 0);
        }
    }
    if (a.blocks == 0) return hipSuccess;
    switch (g.vectors_per_lane) {
        case 1: return launch_r<T, 1>(a, g.waves, a.blocks * a.splits, lds_bytes, stream, prepare_only);
        case 2: return launch_r<T, 2>(a, g.waves, a.blocks * a.splits, lds_bytes, stream, prepare_only);
        case 4: return launch_r<T, 4>(a, g.waves, a.blocks * a.splits, lds_bytes, stream, prepare_only);
        case 8: return launch_r<T, 8>(a, g.waves, a.blocks * a.splits, lds_bytes, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template <typename T> hipError_t launch_pair_reduce(const T* react, unsigned react_plane, unsigned slots, T* out, unsigned out_plane, unsigned count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
    hipLaunchKernelGGL(pair_reduce<T>, dim3((count + 63) / 64), dim3(256), 0, stream, react, react_plane, slots, out, out_plane, count);
    return hipGetLastError();
}

template <typename T> hipError_t launch_pair_finish(const FinishArgs<T>& args, hipStream_t stream) {
    if (args.count == 0) return hipSuccess;
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
    hipLaunchKernelGGL(pair_finish<T>, dim3((args.count + 63) / 64), dim3(256), 0, stream, args);
    return hipGetLastError();
}

// One GPU: the tournament over the whole system, then the finish kernel.  Workspace: [C][3][npad] i-side sums, [slots][3][npad] reaction sums.
template <typename T> hipError_t launch_pair(const Shard<T>& s, const PairPlan& p, void* workspace, hipStream_t stream, bool prepare_only) {
    const unsigned npad = p.blocks * p.block_bodies;
    T* const       work = static_cast<T*>(workspace);
    PairArgs<T>    a{};
    a.old_pos = s.old_pos, a.self = work, a.react = work + static_cast<size_t>(p.splits) * 3 * npad;
    a.n = s.i_count, a.i_begin = 0, a.i_count = s.i_count, a.j_begin = 0, a.j_count = s.i_count;
    a.diag = 1, a.keep = 1;
    a.self_first = 0, a.self_origin = 0, a.self_plane = npad, a.react_origin = 0, a.react_plane = npad;
    a.eps2 = s.eps2;
    const PairGeom g{p.vectors_per_lane, p.waves, p.splits};
    if (const auto err = launch_pair_tile<T>(a, g, stream, prepare_only); err != hipSuccess || prepare_only) return err;
    if (hipEvent_t probe = pair_probe_event(); probe != nullptr) {
        if (const auto err = hipEventRecord(probe, stream); err != hipSuccess) return err;
    }
    FinishArgs<T> f{};
    f.old_pos = s.old_pos, f.new_pos = s.new_pos, f.vel = s.vel;
    f.self = a.self, f.react = a.react, f.recv = nullptr, f.extra = nullptr;
    f.origin = 0, f.count = s.i_count;
    f.self_plane = npad, f.react_plane = npad, f.react_slots = p.slots, f.recv_plane = 0;
    f.n_self = 1, f.n_recv = 0;
    f.self_set[0] = {0u, p.splits, 0u, s.i_count};
    f.dt = s.dt, f.damping = s.damping;
    return launch_pair_finish<T>(f, stream);
}

// ---- the same step through a bounded workspace: K slices (see PairSlicing in nbody_kernels.h) ----------------------------------
template <typename T> PairSlicing plan_pair_sliced(unsigned n, unsigned slices, int ovr_r, int ovr_s, int ovr_c) {
    constexpr unsigned W = sizeof(T) == 4 ? 2 : 1;
    PairSlicing        p{};
    if (n == 0 || slices < 2) return p;
    // slices of 32 768 bodies and more take R = 8 (sixteen fp32 / eight fp64 bodies i per lane, one 8-wave workgroup per CU), as the
    // slices of a multi-GPU step do (measured there: profiles/round4_shard_plan_times.txt)
    const int R = (ovr_r == 1 || ovr_r == 2 || ovr_r == 4 || ovr_r == 8) ? ovr_r : (n / slices >= 32768 ? 8 : 4);
    const int S = (ovr_s == 4 || ovr_s == 8 || ovr_s == 12 || ovr_s == 16) ? ovr_s : 8;
    const unsigned chip = 256u;  // a launch costs ceil(grid / 256) rounds whatever the residency; the launches of a step run one after the other
    p.block_bodies          = 64u * static_cast<unsigned>(R) * W;
    const unsigned blocks   = (n + p.block_bodies - 1) / p.block_bodies;
    const unsigned per      = (blocks + slices - 1) / slices;            // blocks per slice
    p.slice_bodies          = per * p.block_bodies;
    p.slices                = (n + p.slice_bodies - 1) / p.slice_bodies;  // (rounding can leave fewer slices than asked for)
    if (p.slices < 2) return PairSlicing{};
    p.partners = p.slices / 2;
    p.even     = (p.slices % 2) == 0;
    if (p.partners + 1 > static_cast<unsigned>(kMaxRecv) || p.partners + 1 > static_cast<unsigned>(kMaxSelfSets)) return PairSlicing{};
    p.plane = p.slice_bodies;  // (a multiple of 64 already)
    auto splits = [&](unsigned units) {  // workgroups per block: whole rounds of 256 workgroups (splits_to_fill)
        unsigned C = ovr_c > 0 ? static_cast<unsigned>(ovr_c) : splits_to_fill(per, units, S, chip);
        while (C > 1 && units < C * static_cast<unsigned>(S)) C /= 2;
        return C;
    };
    p.diag = {R, S, splits((per / 2 + 1) * static_cast<unsigned>(R) * W)};
    p.rect = {R, S, splits(p.slice_bodies / 64)};
    p.rect_upper = p.rect;
    if (p.even && per >= 2 && p.slice_bodies / 64 >= p.rect.splits * 2 * static_cast<unsigned>(S) * 2) p.rect_upper.splits = p.rect.splits * 2;
    const size_t   plane3     = 3 * static_cast<size_t>(p.plane);
    const unsigned diag_slots = per < 2 ? 0u : ((per & 1u) ? per / 2 : per / 2 - 1);
    p.self_per_slice  = (p.diag.splits + static_cast<size_t>(p.partners - 1) * p.rect.splits + p.rect_upper.splits) * plane3;  // (the last rectangle may be split)
    p.react_elements  = static_cast<size_t>(std::max(diag_slots, per)) * plane3;  // one region, reused launch after launch (stream order)
    p.recv_per_slice  = (1 + static_cast<size_t>(p.partners)) * plane3;          // [0]: the slice's own folded diagonal, [s]: from slice r - s
    p.elements        = p.slices * (p.self_per_slice + p.recv_per_slice) + p.react_elements;
    p.workspace_bytes = p.elements * sizeof(T);
    return p;
}

template <typename T> hipError_t launch_pair_sliced(const Shard<T>& s, const PairSlicing& p, void* workspace, hipStream_t stream, bool prepare_only) {
    const unsigned n = s.i_count, K = p.slices, ni = p.slice_bodies, H = p.partners;
    const size_t   plane3 = 3 * static_cast<size_t>(p.plane);
    T* const       work   = static_cast<T*>(workspace);
    T* const       react  = work;                                                  // the reusable region of reaction planes
    auto self_of = [&](unsigned r) { return work + p.react_elements + static_cast<size_t>(r) * p.self_per_slice; };
    auto recv_of = [&](unsigned r) { return work + p.react_elements + static_cast<size_t>(K) * p.self_per_slice + static_cast<size_t>(r) * p.recv_per_slice; };
    auto first_of = [&](unsigned r) { return r * ni; };
    auto count_of = [&](unsigned r) { return std::min(ni, n - r * ni); };
    auto blocks_of = [&](unsigned count) { return (count + p.block_bodies - 1) / p.block_bodies; };
    auto half_of = [&](unsigned r) { return (blocks_of(count_of(r)) / 2) * p.block_bodies; };  // where a split rectangle cuts slice r

    std::vector<FinishArgs<T>> finish(K);
    for (unsigned r = 0; r < K; ++r) {
        FinishArgs<T>& f = finish[r];
        f                = {};
        f.old_pos = s.old_pos, f.new_pos = s.new_pos, f.vel = s.vel;
        f.self = self_of(r), f.react = react, f.recv = recv_of(r), f.extra = nullptr;
        f.origin = first_of(r), f.count = count_of(r);
        f.self_plane = f.react_plane = f.recv_plane = p.plane;
        f.react_slots = 0;  // (the diagonal's slots are folded into recv[0] right after its launch: the region is reused)
        f.dt = s.dt, f.damping = s.damping;
        f.n_recv = 1 + H;
        for (unsigned m = 0; m <= H; ++m) f.recv_set[m] = {0u, 0u};
    }
    for (unsigned r = 0; r < K; ++r) {
        const unsigned own = first_of(r), cnt = count_of(r);
        PairArgs<T>    a{};
        a.old_pos = s.old_pos, a.self = self_of(r), a.n = n, a.eps2 = s.eps2;
        a.self_origin = own, a.self_plane = p.plane, a.react = react, a.react_plane = p.plane;
        // the slice against itself
        a.react_origin = own, a.i_begin = a.j_begin = own, a.i_count = a.j_count = cnt, a.diag = 1, a.keep = 1, a.self_first = 0;
        if (const auto err = launch_pair_tile<T>(a, p.diag, stream, prepare_only); err != hipSuccess) return err;
        finish[r].self_set[finish[r].n_self++] = {0u, p.diag.splits, 0u, cnt};
        const unsigned b = blocks_of(cnt), diag_slots = b < 2 ? 0u : ((b & 1u) ? b / 2 : b / 2 - 1);
        if (diag_slots != 0) {
            if (!prepare_only) {
                if (const auto err = launch_pair_reduce<T>(react, p.plane, diag_slots, recv_of(r), p.plane, cnt, stream); err != hipSuccess) return err;
            }
            finish[r].recv_set[0] = {0u, cnt};
        }
        // the rectangles against the next H slices
        for (unsigned q = 1; q <= H; ++q) {
            const unsigned partner = (r + q) % K, pfirst = first_of(partner), pcnt = count_of(partner);
            a.diag = 0, a.keep = 1;
            a.i_begin = own, a.i_count = cnt, a.j_begin = pfirst, a.j_count = pcnt;
            const bool split = p.even && q == H;  // both partners list this pair of slices: the HIGHER slice is the one that is cut
            if (split) {
                if (r < partner) a.j_count = half_of(partner);                              // all of this slice x the first half of the partner
                else a.i_begin = own + half_of(r), a.i_count = cnt - half_of(r);            // the second half of this slice x all of the partner
            }
            a.react_origin = a.j_begin;
            a.self_first   = p.diag.splits + (q - 1) * p.rect.splits;
            if (a.i_count != 0 && a.j_count != 0) {
                const PairGeom& geom = (split && !(r < partner)) ? p.rect_upper : p.rect;
                if (const auto err = launch_pair_tile<T>(a, geom, stream, prepare_only); err != hipSuccess) return err;
                finish[r].self_set[finish[r].n_self++] = {a.self_first, geom.splits, a.i_begin - own, a.i_count};
                if (!prepare_only) {
                    if (const auto err = launch_pair_reduce<T>(react, p.plane, blocks_of(a.i_count), recv_of(partner) + q * plane3, p.plane, a.j_count, stream); err != hipSuccess) return err;
                }
                finish[partner].recv_set[q] = {0u, a.j_count};
            }
        }
    }
    if (prepare_only) return hipSuccess;
    for (unsigned r = 0; r < K; ++r) {
        if (const auto err = launch_pair_finish<T>(finish[r], stream); err != hipSuccess) return err;
    }
    return hipSuccess;
}

template PairPlan   plan_pair<float>(unsigned, int, int, int, int);
template PairPlan   plan_pair<double>(unsigned, int, int, int, int);
template hipError_t launch_pair<float>(const Shard<float>&, const PairPlan&, void*, hipStream_t, bool);
template hipError_t launch_pair<double>(const Shard<double>&, const PairPlan&, void*, hipStream_t, bool);
template PairSlicing plan_pair_sliced<float>(unsigned, unsigned, int, int, int);
template PairSlicing plan_pair_sliced<double>(unsigned, unsigned, int, int, int);
template hipError_t  launch_pair_sliced<float>(const Shard<float>&, const PairSlicing&, void*, hipStream_t, bool);
template hipError_t  launch_pair_sliced<double>(const Shard<double>&, const PairSlicing&, void*, hipStream_t, bool);
template hipError_t launch_pair_tile<float>(const PairArgs<float>&, const PairGeom&, hipStream_t, bool);
template hipError_t launch_pair_tile<double>(const PairArgs<double>&, const PairGeom&, hipStream_t, bool);
template hipError_t launch_pair_reduce<float>(const float*, unsigned, unsigned, float*, unsigned, unsigned, hipStream_t);
template hipError_t launch_pair_reduce<double>(const double*, unsigned, unsigned, double*, unsigned, unsigned, hipStream_t);
template hipError_t launch_pair_finish<float>(const FinishArgs<float>&, hipStream_t);
template hipError_t launch_pair_finish<double>(const FinishArgs<double>&, hipStream_t);

}  // namespace nb
