// nbody_fast.hip -- the production all-pairs kernels (NB_MODE_FAST).  gfx950 (CDNA4) only.
//
// What it computes is the reference kernel's bodyBodyInteraction / computeBodyAccel / integrateBodies
// (/root/reference/src/nbody/bodysystemcuda.cu:98-184); how it is laid out is MI355X-first and follows
// what the microbenchmarks measured on the chip (profiles/round1_valu_microbench.txt, profiles/round2_loop_microbench.txt):
//
//   * The loop is VALU-issue bound: no MFMA (a 3-vector accumulate is not a contraction), HBM traffic is
//     64 B per body per step.  A packed fp32 op (v_pk_add/mul/fma_f32) issues in 4.1 cycles per wave64 on a SIMD with 4
//     resident waves and does two lanes' worth of work; v_rsq_f32 costs 8.4.  So for fp32 each lane carries its bodies i
//     in PAIRS, one pair per 64-bit VGPR pair: an interaction pair is 3 v_pk_add + 6 v_pk_fma + 2-3 v_pk_mul + 2 v_rsq_f32,
//     the j body's x/y/z/m being broadcast into both halves with op_sel, not moved.
//     fp64 has no packed form: one body per "vector"; m*d2^(-3/2) from the v_rsq_f64 seed by a 2-term series (Lane<double>::coupling).
//   * Wave-stream layout (large shards): each lane register-tiles R vectors (I = R*W bodies i, W = 2 fp32 / 1 fp64).  The
//     S waves of a workgroup own the SAME 64*I bodies i and split the bodies j: wave w streams chunks w, w+S, w+2S, ...
//     of CH = 64*LPT consecutive bodies.  Every lane of a wave meets the same body j, so a body j is wave-UNIFORM data:
//     the wave reads its chunk with SCALAR loads (s_load_dwordx16 = 4 bodies, through the constant address space and the
//     scalar cache, one group ahead of the one being computed) into scalar registers, and x/y/z/m enter the packed
//     subtractions as scalar operands, broadcast into both halves by op_sel.  Rounds 1-2 staged the chunk in a per-wave
//     LDS ring and read it back with broadcast ds_read_b128: same VALU count, but the loop also carried the LDS
//     instructions and 4 more vector-register reads per body j -- measured in isolation 65.4 -> 61.8 cycles per
//     interaction pair (profiles/round2_loop_microbench.txt), in the kernel +2.9 % fp32 / +2.7 % fp64 on the same box
//     (profiles/round2_scalar_stream_ab.txt).  No workgroup barrier, no LDS traffic at all in the main loop; the waves
//     of a SIMD drift apart freely and keep its VALU issuing.  The S partial sums are folded through LDS in a fixed
//     order at the end (deterministic).  The LDS-tile form lives on in the wave-split layout below, where the bodies j
//     differ from lane to lane.
//   * Mass factorisation: sums are accumulated in units of m_ref (the mass of the first body j of the range), and a
//     chunk whose bodies ALL have mass m_ref -- every chunk of an equal-mass system such as the reference's start-up
//     configurations -- takes a loop without the mass multiply (11 packed ops + 2 v_rsq per interaction pair instead of
//     12 + 2; a chunk of mixed masses multiplies by the raw mass in the loop and by 1/m_ref once, when its sums join the others).  So does a chunk whose bodies all share ANOTHER mass (a species of
//     a galaxy file): into sums of its own, which join the running sums once, scaled by that mass.
//   * Wave-split layout (small shards, fewer bodies i than lanes on the chip): a wave owns the bodies i, its 64
//     lanes split j, wavefront-64 butterfly fold at the end (integrate_bodies_wavesplit below).
//   * softening^2 stays in a vector register pair: as a scalar operand of the v_pk_fma it changes nothing (same microbenchmark).
//   * The shard form (i-range x j-range, optional partial sums in/out) is the same kernel; the single-GPU
//     step is the shard i = j = [0,N) with finalize.
//
// Compiled with FMA contraction ON (default) -- results differ from the CPU path in rounding only; the
// bit-exact path is nbody_strict.hip.
#include "nbody_kernels.h"

#include <algorithm>
#include <atomic>

namespace nb {
namespace {

// workgroup size by j-split factor: S waves (S = 4, 8, 16 -> 256, 512, 1024 threads)
constexpr int block_threads_for(int S) { return 64 * S; }
// lanes_per_body value that selects the wave-split layout (all 64 lanes of a wave split j for the wave's bodies i)
constexpr int kWaveSplit = 64;

#include "nbody_lane.h"

// bodyBodyInteraction, bodysystemcuda.cu:98-123, for one body j against the R vectors of bodies i of this lane.
// bj.w is the body's mass relative to the range's reference mass; UNIT: it is exactly 1 and never read.
template <typename T, int R, bool UNIT>
__device__ __forceinline__ void interact(const typename Lane<T>::vec4 bj, const typename Lane<T>::vec (&px)[R], const typename Lane<T>::vec (&py)[R], const typename Lane<T>::vec (&pz)[R], typename Lane<T>::vec (&ax)[R],
                                         typename Lane<T>::vec (&ay)[R], typename Lane<T>::vec (&az)[R], const typename Lane<T>::vec eps2, const typename Lane<T>::Consts& consts) {
    using L   = Lane<T>;
    using vec = typename L::vec;
    const vec bx = L::splat(bj.x), by = L::splat(bj.y), bz = L::splat(bj.z), bm = L::z_and_mass(bj);
    if constexpr (UNIT && sizeof(T) == 4) {
        // Keep the whole 16-byte body in registers although the unit-mass loop never reads .w: z then stays the low half
        // of the aligned pair {z, w} it was loaded into and is broadcast from there by op_sel.  Reading only 12 bytes frees
        // the upper register, the allocator reuses it, and z gets copied out first (3 v_mov_b32 per 4 bodies j): +2.1 %
        // measured with the full load (A/B on one box, profiles/round2_keep_w_ab.txt).  (Storing {x, y, m, z} instead, to
        // rid the generic loop of its 4 mass copies per 4 bodies, made both loops worse.)
        asm volatile("" : : "v"(bj.w));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const vec dx   = bx - px[r];
        const vec dy   = by - py[r];
        const vec dz   = bz - pz[r];
        vec       d2   = L::fma(dx, dx, eps2);
        d2             = L::fma(dy, dy, d2);
        d2             = L::fma(dz, dz, d2);
        const vec s    = L::template coupling<UNIT>(bm, d2, consts);
        ax[r]          = L::fma(dx, s, ax[r]);
        ay[r]          = L::fma(dy, s, ay[r]);
        az[r]          = L::fma(dz, s, az[r]);
    }
    // generic loop: keep {z, w} where it was loaded until the body is done with (rids the loop of 26 hazard s_nop per 4 bodies j)
    if constexpr (!UNIT && sizeof(T) == 4) asm volatile("" : : "v"(bj.z), "v"(bj.w));
}

// The same for a body j held in SCALAR registers (every lane of the wave meets the same body): its coordinates enter the
// subtractions as scalar operands, broadcast to both halves of a packed pair by op_sel.  `mrel`: relative mass, a vector.
template <typename T, int R, bool UNIT>
__device__ __forceinline__ void interact_uniform(const typename Lane<T>::raw4 bj, const typename Lane<T>::vec mrel, const typename Lane<T>::vec (&px)[R], const typename Lane<T>::vec (&py)[R], const typename Lane<T>::vec (&pz)[R],
                                                 typename Lane<T>::vec (&ax)[R], typename Lane<T>::vec (&ay)[R], typename Lane<T>::vec (&az)[R], const typename Lane<T>::vec eps2, const typename Lane<T>::Consts& consts) {
    using L   = Lane<T>;
    using vec = typename L::vec;
    const vec bx = L::splat(bj.x), by = L::splat(bj.y), bz = L::splat(bj.z);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const vec dx = bx - px[r];
        const vec dy = by - py[r];
        const vec dz = bz - pz[r];
        vec       d2 = L::fma(dx, dx, eps2);
        d2           = L::fma(dy, dy, d2);
        d2           = L::fma(dz, dz, d2);
        const vec s  = L::template coupling_rel<UNIT>(mrel, d2, consts);
        ax[r]        = L::fma(dx, s, ax[r]);
        ay[r]        = L::fma(dy, s, ay[r]);
        az[r]        = L::fma(dz, s, az[r]);
    }
}

// The mass every sum of a j range is expressed in units of: the first body's, when 1/m is a well-behaved number
// (then an equal-mass range never multiplies by a mass inside the loop), otherwise 1.
template <typename T, typename Stream> __device__ __forceinline__ T reference_mass(const Shard<T>& s, Stream bodies) {
    if (s.j_count == 0) return T(1);
    const T m = bodies[s.j_begin].w;  // (a scalar load: the value is compared with scalar registers)
    const T a = m < 0 ? -m : m;
    return (a >= T(0x1p-60) && a <= T(0x1p60)) ? m : T(1);  // false for NaN too
}

// T: float|double   R: vectors per lane (I = R*W bodies i)   S: waves splitting j   LPT: vec4 loads per lane per chunk
template <typename T, int R, int S, int LPT> __global__ __launch_bounds__(block_threads_for(S)) __attribute__((amdgpu_waves_per_eu(4, 4))) void integrate_bodies_fast(Shard<T> s) {
    using LT            = Lane<T>;
    using vec4          = typename LT::vec4;
    using vec           = typename LT::vec;
    constexpr int W     = LT::W;
    constexpr int I     = R * W;     // bodies i per lane
    constexpr int CH    = 64 * LPT;  // bodies j per wave per chunk
    constexpr int BODIES_PER_BLOCK = 64 * I;
    // j bodies in flight per lane: 8 independent interaction chains (R vectors x U bodies j) hide the VALU latency.
    // Every geometry is capped at 128 VGPRs (4 waves/SIMD: 16 waves per CU in 1, 2 or 4 workgroups), so with R >= 2 the
    // loop unrolls less instead of spilling (an R = 4 body at U = 8 spilled 1.2 KB/lane to scratch: 1.2 GB of HBM writes per launch).
    constexpr int U = sizeof(T) == 8 && R == 1 ? 4 : (R >= 2 ? 8 / R : 8);  // (fp64 bodies take 8 scalar registers each)
    static_assert(CH % U == 0, "inner loop is unrolled by U");

    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];

    using raw4 = typename LT::raw4;
    typedef const raw4 __attribute__((address_space(4)))* stream_ptr;  // read-only for the whole launch -> s_load_dwordx4/x8/x16
    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const stream_ptr         bodies  = reinterpret_cast<stream_ptr>(reinterpret_cast<unsigned long long>(s.old_pos));
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // bodies i of this lane: block_base + k*64 + lane, k = r*W + w  (coalesced across the lanes of a wave)
    const unsigned block_base = blockIdx.x * BODIES_PER_BLOCK;
    vec      px[R], py[R], pz[R], ax[R], ay[R], az[R];
    unsigned idx[I];
    bool     active[I];
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const unsigned local = block_base + k * 64 + lane;
        active[k]            = local < s.i_count;
        idx[k]               = s.i_begin + (active[k] ? local : s.i_count - 1);
        const vec4 p         = old_pos[idx[k]];
        LT::set(px[k / W], k % W, p.x);
        LT::set(py[k / W], k % W, p.y);
        LT::set(pz[k / W], k % W, p.z);
    }
    const T m_ref    = reference_mass(s, bodies);
    const T inv_mref = T(1) / m_ref;
#pragma unroll
    for (int r = 0; r < R; ++r) ax[r] = ay[r] = az[r] = LT::splat(0);
    if (s.acc_in && wave == 0) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            const vec4 a = reinterpret_cast<const vec4*>(s.acc)[idx[k]];
            LT::set(ax[k / W], k % W, a.x * inv_mref);
            LT::set(ay[k / W], k % W, a.y * inv_mref);
            LT::set(az[k / W], k % W, a.z * inv_mref);
        }
    }
    vec eps2 = LT::splat(s.eps2);
    LT::keep_in_vgpr(eps2);
    vec inv_mref_v = LT::splat(inv_mref);
    LT::keep_in_vgpr(inv_mref_v);
    const typename LT::Consts consts = LT::make_consts();
    const typename LT::bits   unit_bits = __builtin_bit_cast(typename LT::bits, m_ref);

    const unsigned j_end    = s.j_begin + s.j_count;
    const unsigned n_chunks = (s.j_count + CH - 1) / CH;

#ifdef NB_STAMPS  // diagnostic build only (tools/stamp_probe.py): when does each wave start / finish streaming?
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();  // (100 MHz, one clock for the whole chip)
#endif

    // The SIMD arbiter is strictly oldest-first: left alone, the four waves that share a SIMD finish equal shares of work
    // at 30 % / 53 % / 76 % / 100 % of the workgroup's time (profiles/round2_wave_finish_times.txt), and for the last
    // quarter each SIMD is down to ONE wave, which only reaches 76 % of the issue rate four waves sustain (80.6 against 61.5
    // cycles per interaction pair).  s_setprio outranks age, so progress is equalised instead: each wave publishes how many
    // chunks it has done; a wave that is level with the slowest wave of ITS SIMD (HW_ID.SIMD_ID) runs at priority 3, one that is ahead
    // at 0.  With two waves per SIMD and workgroup (S = 8, the production geometry: two 512-thread workgroups per CU) they
    // finish within 0.5 % of each other; with four (S = 16) the two youngest still trail (a starved wave cannot re-evaluate
    // itself; graded levels made it worse, and letting a yielding wave look again every 8 bodies cost 1 % at S = 8 and 27 % on
    // one-workgroup-per-CU shards), which is why S = 8 is the default.  The chunk -> wave assignment stays static, so the summation order (and every result bit) is the same
    // from run to run.  (Putting the leaders to sleep instead equalises too, but costs 15 %: a SIMD needs 3-4 runnable waves.)
    constexpr size_t kFoldBytes = static_cast<size_t>(S - 1) * 3 * I * 64 * sizeof(T);
    unsigned* const   balance    = reinterpret_cast<unsigned*>(smem_raw + kFoldBytes);
    unsigned* const   simd_count = balance;                                     // [4] waves of this workgroup per SIMD
    volatile unsigned* progress  = reinterpret_cast<volatile unsigned*>(balance + 4);  // [4][8] chunks done, by SIMD and slot
    if (tid < 36) balance[tid] = tid < 4 ? 0u : 0xffffffffu;
    __syncthreads();
    const unsigned simd = static_cast<unsigned>(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4));  // HW_REG_HW_ID[5:4] = SIMD_ID
    unsigned       slot = 0;
    if (lane == 0) slot = atomicAdd(&simd_count[simd], 1u);
    slot                = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(slot))) & 7u;
    volatile unsigned* const mine = progress + simd * 8;
    unsigned done = 0;
    if (lane == 0) mine[slot] = 0;

    // fp32: a register sum only ever collects kFlush chunks (1 024 bodies j); it is then added to the lane's own second-level
    // sum in LDS.  One running fp32 sum over N/S terms loses ~sqrt(N/S) ulp (1.6e-5 relative at 1 Mi bodies against an fp64
    // direct sum); two levels of <= 1 024 and <= N/(1024 S) terms keep it at a few 1e-6 for any N.  (fp64 has the bits to spare.)
    constexpr bool     kTwoLevel = sizeof(T) == 4;
    constexpr unsigned kFlush    = 1024 / CH;
    T* const second = reinterpret_cast<T*>(balance + 64) + static_cast<size_t>(wave) * (3 * I * 64) + lane;  // [S][3][I][64]
    if constexpr (kTwoLevel) {
#pragma unroll
        for (int q = 0; q < 3 * I; ++q) second[q * 64] = 0;
    }
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            second[(0 * I + k) * 64] += LT::get(ax[k / W], k % W);
            second[(1 * I + k) * 64] += LT::get(ay[k / W], k % W);
            second[(2 * I + k) * 64] += LT::get(az[k / W], k % W);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ax[r] = ay[r] = az[r] = LT::splat(0);
    };

    // Every lane of a wave meets the same body j, so the bodies j are not staged anywhere: the wave reads them U at a time with
    // scalar loads (through the constant address space: `old_pos` is read-only for the whole launch) straight into scalar
    // registers, one group ahead of the one it is computing, and they enter the packed subtractions as scalar operands.
    // Against a per-wave LDS ring read back with ds_read_b128 (rounds 1-2) the loop loses its LDS instructions and 4
    // vector-register reads per body j.
    // Whether a chunk takes the loop without the mass multiply (every mass == m_ref) is found one chunk ahead: each lane
    // looks at the masses of two bodies of the wave's next chunk with an ordinary vector load.
    // The masses of chunk c, one or more per lane; wave-uniform answer.  kUnit: every mass is m_ref.  kUniform: the bodies of
    // the chunk all have the SAME mass (a species of a galaxy file) -- the chunk then runs the loop without the mass
    // multiply into sums of its own, which join the running sums scaled by that mass.  kMixed: the mass-multiplying loop.
    // (A ragged chunk -- the last of the range -- is only judged by the bodies it has; its odd bodies go one by one anyway.)
    enum : int { kMixed = 0, kUnit = 1, kUniform = 2 };
    auto chunk_form = [&](unsigned c, T& common_mass) -> int {
        const unsigned first_j = s.j_begin + c * CH;
        using bits = typename LT::bits;
        const bits first_bits = __builtin_bit_cast(bits, s.old_pos[4 * static_cast<size_t>(first_j) + 3]);  // (uniform address)
        bool same = true;
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const unsigned j = first_j + r * 64 + lane;
            same             = same && (j >= j_end || __builtin_bit_cast(bits, s.old_pos[4 * static_cast<size_t>(j < j_end ? j : first_j) + 3]) == first_bits);
        }
        common_mass = __builtin_bit_cast(T, first_bits);
        if (__builtin_amdgcn_ballot_w64(!same) != 0 || !(common_mass == common_mass)) return kMixed;  // (NaN masses take the plain loop)
        return first_bits == unit_bits ? kUnit : kUniform;
    };
    auto group = [&](stream_ptr from, raw4 (&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = from[u];
    };

    // U bodies j against the R vectors of bodies i, written stage by stage (all differences, all squared distances, all
    // reciprocal square roots, ...): U*R independent chains in flight whatever the instruction scheduler makes of it
    auto compute = [&]<bool UNIT>(const raw4 (&b)[U], vec (&ax)[R], vec (&ay)[R], vec (&az)[R]) {
        constexpr int UB = (4 / R > 0 ? 4 / R : 1) < U ? (4 / R > 0 ? 4 / R : 1) : U;  // bodies j per stage block: >= 4 chains
#pragma unroll
        for (int h = 0; h < U; h += UB) {
            vec dx[UB][R], dy[UB][R], dz[UB][R], w[UB][R];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const vec bx = LT::splat(b[h + u].x), by = LT::splat(b[h + u].y), bz = LT::splat(b[h + u].z);
#pragma unroll
                for (int r = 0; r < R; ++r) dx[u][r] = bx - px[r], dy[u][r] = by - py[r], dz[u][r] = bz - pz[r];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = LT::fma(dx[u][r], dx[u][r], eps2);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = LT::fma(dy[u][r], dy[u][r], w[u][r]);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = LT::fma(dz[u][r], dz[u][r], w[u][r]);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                vec mrel = inv_mref_v;
                if constexpr (!UNIT) mrel = LT::splat(b[h + u].w);  // (the raw mass, a scalar operand; the chunk's sums are scaled once)
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = LT::template coupling_rel<UNIT>(mrel, w[u][r], consts);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    ax[r] = LT::fma(dx[u][r], w[u][r], ax[r]);
                    ay[r] = LT::fma(dy[u][r], w[u][r], ay[r]);
                    az[r] = LT::fma(dz[u][r], w[u][r], az[r]);
                }
            }
        }
    };
    auto arrived = [](const raw4 (&b)[U]) { asm volatile("" : : "s"(b[0]) : "memory"); };  // first use of the set: what follows is issued after its wait
    // b0 holds (or is loading) group 0 of the chunk; on return it is loading the first group at `next` (the wave's next chunk)
    auto stream = [&]<bool UNIT>(stream_ptr chunk, unsigned groups, stream_ptr next, raw4 (&b0)[U], raw4 (&b1)[U], vec (&sx)[R], vec (&sy)[R], vec (&sz)[R]) {
        unsigned g = 0;
#pragma unroll 1
        for (; g + 2 <= groups; g += 2) {
            arrived(b0);
            group(chunk + (g + 1) * U, b1);
            __builtin_amdgcn_sched_barrier(0);  // (the load stays ahead of the compute it overlaps)
            compute.template operator()<UNIT>(b0, sx, sy, sz);
            arrived(b1);
            group(g + 2 < groups ? chunk + (g + 2) * U : next, b0);
            __builtin_amdgcn_sched_barrier(0);
            compute.template operator()<UNIT>(b1, sx, sy, sz);
        }
        if (g < groups) {  // (odd count: the ragged last chunk of the range, nothing follows it)
            compute.template operator()<UNIT>(b0, sx, sy, sz);
        }
    };

    unsigned c    = wave;  // wave w streams chunks w, w+S, w+2S, ...
    T        common_mass = 0, next_mass = 0;
    int      form = c < n_chunks ? chunk_form(c, common_mass) : kMixed;
    raw4     b0[U], b1[U];  // two register sets: while one group is computed the next one is in flight.  (Scalar loads return in
                            // any order, so a wait is for everything outstanding: a set is loaded only once the other has been waited for.)
    if (c < n_chunks && j_end - (s.j_begin + c * CH) >= static_cast<unsigned>(U)) group(bodies + (s.j_begin + c * CH), b0);
    for (; c < n_chunks; c += S) {
        const int next_form = (c + S) < n_chunks ? chunk_form(c + S, next_mass) : kMixed;  // (its loads are in flight across the compute below)
#ifndef NB_NO_BALANCE  // (diagnostic builds switch it off: tools/stamp_probe.py)
        // a wave that is not ahead of any wave of its SIMD (same workgroup) runs at priority 3, the others at 0
        {
            unsigned least = done;
#pragma unroll
            for (int q = 0; q < 8; ++q) least = min(least, mine[q]);  // unsynchronised reads: a stale value only delays a priority change
            if (static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(least))) >= done) {
                __builtin_amdgcn_s_setprio(3);
            } else {
                __builtin_amdgcn_s_setprio(0);
            }
        }
#endif
        const unsigned   first  = s.j_begin + c * CH;
        const unsigned   count  = min(static_cast<unsigned>(CH), j_end - first);
        const unsigned   groups = count / U;
        const stream_ptr chunk  = bodies + first;
        // the wave's next chunk, when it has a whole group (else anything readable: the set is not used again)
        const stream_ptr next = ((c + S) < n_chunks && j_end - (first + S * CH) >= static_cast<unsigned>(U)) ? chunk + S * CH : chunk;
        if (groups > 0) {
            if (form == kUnit) {
                stream.template operator()<true>(chunk, groups, next, b0, b1, ax, ay, az);
            } else if (form == kUniform) {
                vec cx[R], cy[R], cz[R];
#pragma unroll
                for (int r = 0; r < R; ++r) cx[r] = cy[r] = cz[r] = LT::splat(0);
                stream.template operator()<true>(chunk, groups, next, b0, b1, cx, cy, cz);
                const vec scale = LT::splat(common_mass) * inv_mref_v;
#pragma unroll
                for (int r = 0; r < R; ++r) ax[r] = LT::fma(cx[r], scale, ax[r]), ay[r] = LT::fma(cy[r], scale, ay[r]), az[r] = LT::fma(cz[r], scale, az[r]);
            } else {  // mixed masses: the raw mass multiplies inside the loop, 1/m_ref once per chunk
                vec cx[R], cy[R], cz[R];
#pragma unroll
                for (int r = 0; r < R; ++r) cx[r] = cy[r] = cz[r] = LT::splat(0);
                stream.template operator()<false>(chunk, groups, next, b0, b1, cx, cy, cz);
#pragma unroll
                for (int r = 0; r < R; ++r) ax[r] = LT::fma(cx[r], inv_mref_v, ax[r]), ay[r] = LT::fma(cy[r], inv_mref_v, ay[r]), az[r] = LT::fma(cz[r], inv_mref_v, az[r]);
            }
        }
#pragma unroll 1
        for (unsigned jj = groups * U; jj < count; ++jj) {  // ragged end of the range
            const raw4 b = chunk[jj];
            interact_uniform<T, R, false>(b, LT::splat(b.w) * inv_mref_v, px, py, pz, ax, ay, az, eps2, consts);
        }
        form = next_form, common_mass = next_mass;
        ++done;
        if constexpr (kTwoLevel) {
            if (done % kFlush == 0) flush();
        }
        if (lane == 0) mine[slot] = done;
    }
    if (lane == 0) mine[slot] = 0xffffffffu;  // finished: never the one the others defer to
    __builtin_amdgcn_s_setprio(0);
    if constexpr (kTwoLevel) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            LT::set(ax[k / W], k % W, second[(0 * I + k) * 64] + LT::get(ax[k / W], k % W));
            LT::set(ay[k / W], k % W, second[(1 * I + k) * 64] + LT::get(ay[k / W], k % W));
            LT::set(az[k / W], k % W, second[(2 * I + k) * 64] + LT::get(az[k / W], k % W));
        }
    }
#ifdef NB_STAMPS
    if (lane == 0 && s.acc != nullptr && s.finalize && !s.acc_in) {
        unsigned long long* stamps = reinterpret_cast<unsigned long long*>(s.acc) + (static_cast<size_t>(blockIdx.x) * S + wave) * 2;
        stamps[0] = stamp_t0, stamps[1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // fold the S partial sums (waves 1..S-1 -> wave 0) through LDS, fixed order
    T* red = reinterpret_cast<T*>(smem_raw);  // [(S-1)][3][I][64]
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            red[(((wave - 1) * 3 + 0) * I + k) * 64 + lane] = LT::get(ax[k / W], k % W);
            red[(((wave - 1) * 3 + 1) * I + k) * 64 + lane] = LT::get(ay[k / W], k % W);
            red[(((wave - 1) * 3 + 2) * I + k) * 64 + lane] = LT::get(az[k / W], k % W);
        }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1  // (fully unrolled, the S = 16 fold hoists 45*I LDS loads and spills)
    for (int g = 1; g < S; ++g) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            LT::set(ax[k / W], k % W, LT::get(ax[k / W], k % W) + red[(((g - 1) * 3 + 0) * I + k) * 64 + lane]);
            LT::set(ay[k / W], k % W, LT::get(ay[k / W], k % W) + red[(((g - 1) * 3 + 1) * I + k) * 64 + lane]);
            LT::set(az[k / W], k % W, LT::get(az[k / W], k % W) + red[(((g - 1) * 3 + 2) * I + k) * 64 + lane]);
        }
    }

#pragma unroll
    for (int k = 0; k < I; ++k) {
        // (index and activity are worked out again from a lane number the compiler cannot tie to the one above: held across
        // the streaming loop they cost I registers that the loop's stage blocks need)
        unsigned lane_again = lane;
        asm volatile("" : "+v"(lane_again));
        const unsigned local = block_base + k * 64 + lane_again;
        if (local >= s.i_count) continue;
        const unsigned i  = s.i_begin + local;
        const T        fx = LT::get(ax[k / W], k % W) * m_ref, fy = LT::get(ay[k / W], k % W) * m_ref, fz = LT::get(az[k / W], k % W) * m_ref;
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

template <typename T, int R, int LPT, int BLOCK> __global__ __launch_bounds__(BLOCK) void integrate_bodies_wavesplit(Shard<T> s) {
    using LT            = Lane<T>;
    using vec4          = typename LT::vec4;
    using vec           = typename LT::vec;
    constexpr int kBlock = BLOCK;  // (every workgroup stages ALL bodies j through its LDS: the more waves share a tile, the less L2 traffic)
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
    const typename LT::Consts consts = LT::make_consts();

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
            for (int u = 0; u < U; ++u) interact<T, R, false>(mine[jj + 64 * u], px, py, pz, ax, ay, az, eps2, consts);
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

template <typename T, int R, int LPT, int BLOCK> hipError_t launch_wavesplit(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    if (p.lds_bytes > 64u * 1024u) {
        if (const auto err = allow_large_lds<&integrate_bodies_wavesplit<T, R, LPT, BLOCK>>(); err != hipSuccess) return err;
    }
    if (prepare_only) return hipSuccess;
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
    hipLaunchKernelGGL((integrate_bodies_wavesplit<T, R, LPT, BLOCK>), dim3(p.grid_blocks), dim3(BLOCK), p.lds_bytes, stream, s);
    return hipGetLastError();
}

template <typename T, int R> hipError_t dispatch_wavesplit(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    switch (p.block_threads * 16 + p.tile_bodies / p.block_threads) {  // (workgroup size, vec4 loads per lane and tile)
        case 256 * 16 + 2: return launch_wavesplit<T, R, 2, 256>(s, p, stream, prepare_only);
        case 256 * 16 + 4: return launch_wavesplit<T, R, 4, 256>(s, p, stream, prepare_only);
        case 512 * 16 + 1: return launch_wavesplit<T, R, 1, 512>(s, p, stream, prepare_only);
        case 512 * 16 + 2: return launch_wavesplit<T, R, 2, 512>(s, p, stream, prepare_only);
        case 1024 * 16 + 1: return launch_wavesplit<T, R, 1, 1024>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template <typename T, int R, int S, int LPT> hipError_t launch_one(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    if (p.lds_bytes > 64u * 1024u) {
        if (const auto err = allow_large_lds<&integrate_bodies_fast<T, R, S, LPT>>(); err != hipSuccess) return err;
    }
    if (prepare_only) return hipSuccess;  // graph capture arms the attribute before hipStreamBeginCapture
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
    hipLaunchKernelGGL((integrate_bodies_fast<T, R, S, LPT>), dim3(p.grid_blocks), dim3(block_threads_for(S)), p.lds_bytes, stream, s);
    return hipGetLastError();
}

template <typename T, int R, int S> hipError_t dispatch_lpt(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    constexpr int kBlock = block_threads_for(S);
    if (p.tile_bodies % kBlock) return hipErrorInvalidValue;
    switch (p.tile_bodies / kBlock) {  // = bodies j per wave per chunk / 64
        case 1: return launch_one<T, R, S, 1>(s, p, stream, prepare_only);
        case 2: return launch_one<T, R, S, 2>(s, p, stream, prepare_only);
        case 4: return launch_one<T, R, S, 4>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template <typename T, int R> hipError_t dispatch_s(const Shard<T>& s, const Plan& p, hipStream_t stream, bool prepare_only) {
    switch (p.lanes_per_body) {
        case 4: return dispatch_lpt<T, R, 4>(s, p, stream, prepare_only);
        case 8: return dispatch_lpt<T, R, 8>(s, p, stream, prepare_only);
        case 16: return dispatch_lpt<T, R, 16>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// Geometry (measured: profiles/round2_plan_sweep_*.txt; round 1's sweeps for the wave-split crossover).
//   * S = 8: a 512-thread workgroup = 8 waves = 2 per SIMD, all working on the SAME 64*I bodies i and each streaming every
//     8th chunk of the bodies j.  Two workgroups share a CU (128 VGPRs -> 4 waves per SIMD); a SIMD sustains its full issue
//     rate with 3-4 runnable waves and 96.5 % of it with 2 (61.5 / 63.7 cycles per interaction pair), so a shard that only
//     has one workgroup per CU (65 536 bodies at I = 4) still runs near full rate.  S = 16 (one 1024-thread workgroup per
//     CU) is 1-2 % slower: its four waves per SIMD cannot be kept level (see the kernel), S = 4 is 8 % slower.
//   * I (bodies i per lane, a multiple of W: fp32 bodies travel in packed pairs; at most 4, what fits 128 VGPRs):
//     register tiling amortises the per-body scalar work, but what matters more is how the resulting workgroup count
//     divides over the 256 CUs.
//   * 128 bodies j per wave and chunk (`tile_bodies` = S * 128 = the bodies j a workgroup takes per round): a chunk is the
//     unit of the unit-mass decision, of the priority balancing and of the fp32 two-level sums; nothing is staged, so
//     the only LDS is the fold buffer, the progress words and the second-level sums.
template <typename T> Plan plan_fast(unsigned i_count, unsigned j_count, int cu_count, int ovr_i, int ovr_s, int ovr_tile) {
    constexpr int W     = Lane<T>::W;
    constexpr int kMaxI = 4;  // fp32: 2 packed pairs, fp64: 4 bodies -- the most a 1024-thread workgroup holds in 128 VGPRs without spilling
    // Wave-stream layout: a CU works through ceil(blocks / CUs) workgroups (two at a time), so its efficiency is the fill
    // of the last round.  Pick the I whose workgroup count quantises best (larger I is ~3 % faster per interaction).
    int    I        = W;
    double best_eff = 0.0;
    for (int cand = W; cand <= kMaxI; cand *= 2) {
        const long   blocks = (static_cast<long>(i_count) + 64L * cand - 1) / (64L * cand);
        const long   rounds = (blocks + cu_count - 1) / cu_count;
        const double eff    = static_cast<double>(blocks) / static_cast<double>(rounds * cu_count) * (cand == kMaxI ? 1.0 : 0.97);
        if (eff >= best_eff) best_eff = eff, I = cand;
    }
    // S = 8 (two 512-thread workgroups per CU) once there are more than two rounds of workgroups per CU; with fewer, one
    // 1024-thread workgroup per CU (S = 16) keeps four waves on every SIMD where S = 8 would leave two
    // (profiles/round2_plan_sweep.txt: 65 536 bodies 0.92 vs 0.94-0.97 ms, 32 768-body shard 0.246 vs 0.26-0.28 ms).
    const long blocks_at_i = (static_cast<long>(i_count) + 64L * I - 1) / (64L * I);
    int        S           = blocks_at_i <= 2L * cu_count ? 16 : 8;
    // The wave-split layout runs at ~0.8 of the tile layout's full rate (round 2: 0.78 fp32 / 0.62 fp64, tools/layout_crossover.py; fp64 caught up with tiles of 512 bodies) and its
    // workgroups are 4-16x smaller, so its last round is fuller -- but not full: round 4's sweep of 33 body counts
    // (profiles/round4_one_sided_plan_sweep.txt) showed the old rule (wave-split whenever the tile layout fills under 0.78, two vectors
    // per wave from 16 384 bodies) losing up to 45 %: 18 000 bodies as 282 workgroups of 64 bodies are two rounds for the work of
    // 1.1; with one vector per wave they are 563 workgroups, 2.2 of 3.  Both layouts are now held to the same fill estimate --
    // a launch costs ceil(workgroups / CUs) rounds (nbody_pair.hip) -- and the better one is taken.
    // (round 4, both precisions: 32 768 bodies fp32 0.244 against 0.300 ms, 16 384 bodies fp64 145 against 178 us with tiles of 512 bodies; past 40 000
    // bodies every one of thousands of workgroups staging every tile starts to show: 50 000 bodies fp64 1.70 against 1.64 ms at equal estimates)
    const double wave_split_rate = i_count > 40000 ? 0.76 : 0.80;
    // Wave-split geometry: a wave takes `vectors` bodies i, a workgroup of b waves b * vectors.  The layout is issue-bound from 16
    // waves per CU on, so a launch costs max over CUs of (waves on it) x vectors = ceil(workgroups / CUs) x b x vectors: 8 193
    // bodies as 257 workgroups of 16 waves put 32 waves on one CU (41.0 us against 23.4 for 8 192 bodies), as 1 025 workgroups of 4
    // waves 20 (30.5 us).  Smaller workgroups stage every tile more often (+2 % at 8 waves, +6 % at 4: L2 traffic); two vectors per
    // wave are 2.5 % cheaper per body from 16 384 bodies (8 192 fp64).  (profiles/round4_one_sided_plan_sweep.txt, part 2.)
    const int  wave_split_tile = (j_count > 512 && sizeof(T) == 4) ? 1024 : 512;  // (fp64: 512 bodies j -- 32 KB of LDS instead of 64 -- by 3-8 % at every size)
    const bool two_vectors_ok  = static_cast<long>(i_count) / (4 * 2 * W) >= 4L * cu_count;
    int        wave_split_i = W, wave_split_block = 256;
    double     wave_split_cost = 0;
    for (int vectors : {W, 2 * W}) {
        if (vectors != W && !two_vectors_ok) continue;
        for (int block : {1024, 512, 256}) {
            if (wave_split_tile % block) continue;
            if (block == 256 && i_count > 20480) continue;  // (every workgroup stages every tile: the traffic of four-wave workgroups was measured up to 20 000 bodies only)
            const long   per    = block / 64 * vectors;
            const long   blocks = (static_cast<long>(i_count) + per - 1) / per;
            const long   rounds = (blocks + cu_count - 1) / cu_count;
            const double cost   = static_cast<double>(rounds * per) * (vectors == W ? 1.0 : 0.975) * (block == 1024 ? 1.0 : (block == 512 ? 1.02 : (sizeof(T) == 4 ? 1.06 : 1.08)));
            if (wave_split_cost == 0 || cost < wave_split_cost) wave_split_cost = cost, wave_split_i = vectors, wave_split_block = block;
        }
    }
    const double wave_split_fill = static_cast<double>(i_count) / cu_count / wave_split_cost;  // bodies per CU over what the busiest CU is charged
    // (up to 8 192 bodies the wave-split layout always: the tile layout has under half a round of workgroups there)
    if (i_count <= 8192 ? best_eff < wave_split_rate : best_eff < wave_split_rate * wave_split_fill) {
        S = kWaveSplit;
        I = wave_split_i;
    }
    if (ovr_i > 0) I = std::min(std::max(ovr_i / W * W, W), kMaxI);
    if (ovr_s > 0) S = ovr_s;
    if (S == kWaveSplit) {
        if (I > 2 * W) I = 2 * W;
        Plan p;
        p.bodies_per_lane = I;  // per WAVE in this layout
        p.lanes_per_body  = kWaveSplit;
        p.tile_bodies     = (ovr_tile == 512 || ovr_tile == 1024) ? ovr_tile : wave_split_tile;
        // the workgroup size of the search above; with an override (another I or tile): the largest workgroup that still leaves one per CU
        p.block_threads = 256;
        if (I == wave_split_i && p.tile_bodies == wave_split_tile) {
            p.block_threads = wave_split_block;
        } else {
            for (int cand : {1024, 512}) {
                if (p.tile_bodies % cand == 0 && (static_cast<long>(i_count) + cand / 64 * I - 1) / (cand / 64 * I) >= cu_count) {
                    p.block_threads = cand;
                    break;
                }
            }
        }
        const unsigned bodies_per_block = static_cast<unsigned>(p.block_threads / 64 * I);
        p.grid_blocks     = (i_count + bodies_per_block - 1) / bodies_per_block;
        p.lds_bytes       = static_cast<unsigned>(2ull * p.tile_bodies * 4 * sizeof(T));
        return p;
    }
    const int block = block_threads_for(S);
    int       tile  = 128 * S;
    if (S == 8 && j_count >= 2048) tile = 2048;  // 256 bodies j per wave and chunk: 1-3 % over 128 at every size of round 4's sweep (262 144 bodies 14.72 -> 14.58 ms)
    if (ovr_tile > 0) tile = ovr_tile;
    if (tile < block) tile = block;

    Plan p;
    p.bodies_per_lane = I;
    p.lanes_per_body  = S;
    p.tile_bodies     = tile;
    p.block_threads   = block;
    const unsigned bodies_per_block = static_cast<unsigned>(block / S * I);
    p.grid_blocks     = (i_count + bodies_per_block - 1) / bodies_per_block;
    const size_t red_bytes  = static_cast<size_t>(S - 1) * 3 * I * (block / S) * sizeof(T);       // the fold of the S partial sums
    p.lds_bytes             = static_cast<unsigned>(red_bytes) + 256u;                            // + the waves' progress words
    if (sizeof(T) == 4) p.lds_bytes += static_cast<unsigned>(S) * 3 * I * 64 * sizeof(T);       // + fp32: the lanes' second-level sums
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
    switch (p.bodies_per_lane) {  // at most 4 bodies i per lane: fp32 R = 1, 2 packed pairs; fp64 R = 1, 2, 4
        case 1: if constexpr (W == 1) return dispatch_s<T, 1>(s, p, stream, prepare_only); else return hipErrorInvalidValue;
        case 2: return dispatch_s<T, 2 / W>(s, p, stream, prepare_only);
        case 4: return dispatch_s<T, 4 / W>(s, p, stream, prepare_only);
        default: return hipErrorInvalidValue;
    }
}

template Plan       plan_fast<float>(unsigned, unsigned, int, int, int, int);
template Plan       plan_fast<double>(unsigned, unsigned, int, int, int, int);
template hipError_t launch_fast<float>(const Shard<float>&, const Plan&, hipStream_t, bool);
template hipError_t launch_fast<double>(const Shard<double>&, const Plan&, hipStream_t, bool);

}  // namespace nb
