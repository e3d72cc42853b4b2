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
// pair_forces launch, {start, entry of the unit loop, its exit} in s_memtime ticks + where the hardware put the wave.
__device__ unsigned long long nb_pair_stamps[8192 * 6];
extern "C" __attribute__((visibility("default"))) int nb_debug_read_pair_stamps(void* host, size_t bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(nb_pair_stamps), bytes, 0, hipMemcpyDeviceToHost));
}
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
template <typename T, int R, int S> __global__ __launch_bounds__(64 * S) __attribute__((amdgpu_waves_per_eu(kPairWavesPerSimd<R, S>, kPairWavesPerSimd<R, S>))) void pair_forces(PairArgs<T> s) {
    using LT            = Lane<T>;
    using vec4          = typename LT::vec4;
    using vec           = typename LT::vec;
    using bits          = typename LT::bits;
    constexpr int W     = LT::W;
    constexpr int I     = R * W;    // bodies i per lane
    constexpr int BLOCK = 64 * I;   // bodies per block
    constexpr int TB    = I;        // 64-body tiles per block
#ifndef NB_PAIR_UNR
#define NB_PAIR_UNR 4
#endif
#ifndef NB_PAIR_RB
#define NB_PAIR_RB 2
#endif
    constexpr int UNR   = NB_PAIR_UNR;  // steps per trip of the rotation loop

    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
#ifdef NB_PAIR_STAMPS
    const unsigned long long stamp_start = __builtin_amdgcn_s_memtime();
#endif

    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const int      tid  = threadIdx.x;
    const int      wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int      lane = tid & 63;
    const unsigned a    = blockIdx.x / s.splits;  // the block whose bodies i this workgroup holds
    const unsigned c    = blockIdx.x % s.splits;


    // bodies i of this lane: block_base + k*64 + lane (coalesced across the wave); a body beyond the range sits on its last body with mass 0
    const unsigned block_base = s.i_begin + a * BLOCK;
    const unsigned i_end      = s.i_begin + s.i_count;
    vec            px[R], py[R], pz[R], ax[R], ay[R], az[R];
    // Is the block ONE species -- every body i real and of the same mass m_block?  (Wave-uniform answer: each wave holds the whole block.)
    const T    m_block    = old_pos[block_base < i_end ? block_base : i_end - 1].w;  // (uniform address)
    const bits block_bits = __builtin_bit_cast(bits, m_block);
    bool       same       = true;
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const unsigned i = block_base + k * 64 + lane;
        const vec4     p = old_pos[i < i_end ? i : i_end - 1];
        LT::set(px[k / W], k % W, p.x);
        LT::set(py[k / W], k % W, p.y);
        LT::set(pz[k / W], k % W, p.z);
        same = same && i < i_end && __builtin_bit_cast(bits, p.w) == block_bits;
    }
    const bool block_uniform = __builtin_amdgcn_ballot_w64(!same) == 0 && usable_unit(m_block);
#pragma unroll
    for (int r = 0; r < R; ++r) ax[r] = ay[r] = az[r] = LT::splat(0);
    vec eps2 = LT::splat(s.eps2);
    LT::keep_in_vgpr(eps2);
    const typename LT::Consts consts = LT::make_consts();

    // ---- LDS: [S][3*I][64] second-level sums (fp32) -- later the fold buffer -- then the progress words ---------------------
    constexpr size_t kSumBytes = static_cast<size_t>(S) * 3 * I * 64 * sizeof(T);
    T* const         sums      = reinterpret_cast<T*>(smem_raw);
    unsigned* const  balance   = reinterpret_cast<unsigned*>(smem_raw + kSumBytes);
    unsigned* const  simd_count = balance;                                            // [4] waves of this workgroup per SIMD
    volatile unsigned* progress = reinterpret_cast<volatile unsigned*>(balance + 4);  // [4][8] units done, by SIMD and slot
    if (tid < 36) balance[tid] = tid < 4 ? 0u : 0xffffffffu;
    __syncthreads();
    // SIMD-mate balancing as in nbody_fast.hip: the arbiter is oldest-first, s_setprio outranks age -- a wave level with the
    // slowest wave of its SIMD (same workgroup) runs at priority 3, one that is ahead at 0; the unit -> wave map stays static.
    const unsigned simd = static_cast<unsigned>(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4));  // HW_REG_HW_ID[5:4] = SIMD_ID
    unsigned       slot = 0;
    if (lane == 0) slot = atomicAdd(&simd_count[simd], 1u);
    slot = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(slot))) & 7u;
    volatile unsigned* const mine = progress + simd * 8;
    unsigned                 done = 0;
    if (lane == 0) mine[slot] = 0;

    // fp32: a register sum collects at most kFlush tiles (1 024 bodies j), then joins the lane's second-level sum in LDS
    constexpr bool     kTwoLevel = sizeof(T) == 4;
    constexpr unsigned kFlush    = 16;
    T* const           second    = sums + static_cast<size_t>(wave) * (3 * I * 64) + lane;
    if constexpr (kTwoLevel) {
#pragma unroll
        for (int q = 0; q < 3 * I; ++q) second[q * 64] = 0;
    }
    // The register sums are kept in units of `unit`: the mass of the species whose tiles the wave is working through (a tile of
    // ONE species runs the loop without a mass multiply; when the species changes the sums are re-expressed once: 3R multiplies
    // per change, a handful per step for a galaxy file).  The second-level sums and everything stored are in absolute units.
    T unit = T(1);
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            second[(0 * I + k) * 64] = __builtin_fma(LT::get(ax[k / W], k % W), unit, second[(0 * I + k) * 64]);
            second[(1 * I + k) * 64] = __builtin_fma(LT::get(ay[k / W], k % W), unit, second[(1 * I + k) * 64]);
            second[(2 * I + k) * 64] = __builtin_fma(LT::get(az[k / W], k % W), unit, second[(2 * I + k) * 64]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ax[r] = ay[r] = az[r] = LT::splat(0);
    };
    auto change_unit = [&](T to) {  // (wave-uniform)
        const vec ratio = LT::splat(unit / to);
#pragma unroll
        for (int r = 0; r < R; ++r) ax[r] = ax[r] * ratio, ay[r] = ay[r] * ratio, az[r] = az[r] * ratio;
        unit = to;
    };

    // ---- the units of this wave ------------------------------------------------------------------------------------------
    const unsigned NB      = s.blocks;
    const unsigned Q       = NB / 2;
    const bool     even    = (NB & 1u) == 0;
    const bool     diag    = s.diag != 0;
    const unsigned j_end   = s.j_begin + s.j_count;
    const unsigned n_units = diag ? (s.unit_count != 0 ? s.unit_count : (Q + 1) * TB) : (s.j_count + 63) / 64;  // (a diagonal launch may take a RANGE of the tournament's units: PairArgs::unit_begin)
    const unsigned G       = s.splits * S;
    // Which units are this wave's.  s.deal == 2 (what launch_pair_tile takes when the LDS allows): the first floor(n_units / (C*S)) * C*S
    // units are dealt whole, unit u to slot u mod (C*S) = c*S + wave -- every wave the same number --, and each of the units left over
    // (the "tail": fewer than C*S) goes to ONE workgroup (tail unit t to workgroup t mod C) which runs it as FOUR QUARTERS of sixteen
    // rotation steps, one per SIMD (quarter p on a wave with wave % 4 == p, the workgroup's tail units taking turns between the
    // S/4 waves of a SIMD): a quarter's lanes load the tile rotated by 16p lanes, so that sixteen steps bring lane l exactly the
    // bodies that the steps 16p .. 16p+15 of a whole unit would have, and the four partial reaction sums of a body meet in LDS
    // (fixed order) before one store.  (The waves 0-3 of a workgroup sit on four different SIMDs and wave w + 4 on the SIMD of
    // wave w -- in all 256 workgroups of a launch, by the stamps of tools/pair_stamps.py, profiles/round4_wave_exit_stamps.txt; only the
    // balance rests on that, not the result.)  Every SIMD of a block's workgroups then carries the same load to a quarter of a unit:
    // 16 384 bodies are 33 units per 8-wave workgroup -- 8.25 per SIMD instead of 9 for the one that held the wave with five.
    // s.deal == 0 / 1: every unit whole, unit u to slot u mod (C*S), the slots blocked (c*S + wave) or interleaved (wave*C + c: the
    // waves with one unit more are the low wave ids of EVERY workgroup; round 4's first answer to the remainder, kept for launches
    // whose LDS has no room for the quarters' sums).
    const bool     quartered = s.deal == 2;
    const unsigned g         = s.deal == 1 ? static_cast<unsigned>(wave) * s.splits + c : c * S + static_cast<unsigned>(wave);
    const unsigned dealt     = quartered ? n_units / G * G : n_units;                                   // units below this are dealt whole
    const unsigned n_whole   = g < dealt ? (dealt - g + G - 1) / G : 0;                                 // ... u = g, g + G, ...
    const unsigned wg_tail   = (n_units - dealt) > c ? (n_units - dealt - c + s.splits - 1) / s.splits : 0;  // tail units of this workgroup (<= S)
    constexpr unsigned kPerSimd = S / 4;                                                              // waves of a workgroup per SIMD
    const unsigned my_quarter = static_cast<unsigned>(wave) & 3u, my_turn = static_cast<unsigned>(wave) >> 2;
    const unsigned n_quarters = wg_tail > my_turn ? (wg_tail - my_turn + kPerSimd - 1) / kPerSimd : 0;
    const unsigned n_items    = n_whole + n_quarters;
    auto item_tail = [&](unsigned it) { return my_turn + (it - n_whole) * kPerSimd; };                  // which of the workgroup's tail units
    auto item_unit = [&](unsigned it) { return s.unit_begin + (it < n_whole ? g + it * G : dealt + c + item_tail(it) * s.splits); };  // (absolute: the tile and the offset q follow from it)
    auto item_shift = [&](unsigned it) { return it < n_whole ? 0u : 16u * my_quarter; };

    auto tile_first = [&](unsigned u) {
        if (!diag) return s.j_begin + u * 64;
        unsigned jb = a + u / TB;
        if (jb >= NB) jb -= NB;
        return s.j_begin + jb * BLOCK + (u % TB) * 64;
    };
    auto load_tile = [&](unsigned u, unsigned shift) {  // lane l takes the tile's body (l - shift) mod 64
        const unsigned j = tile_first(u) + ((static_cast<unsigned>(lane) - shift) & 63u);
        vec4           p = old_pos[j < j_end ? j : j_end - 1];
        if (j >= j_end) p.w = 0;
        return p;
    };

    // one rotation step: the lane's bodies i against the body j it holds right now; then the body j and its sums move on.
    // Written stage by stage over the R vectors (all differences, all squared distances, ...): R independent chains.
    // MJ: the bodies j of the tile differ in mass (m_j / unit travels with the body and multiplies the i side);
    // MI: the bodies i of the block differ in mass (their masses multiply the reaction side)
    auto step = [&]<bool MI, bool MJ>(T& jx, T& jy, T& jz, T& jm, vec& rx, vec& ry, vec& rz, const vec (&mi)[R]) {
        constexpr int RB = R < NB_PAIR_RB ? R : NB_PAIR_RB;  // vectors per stage block (more in flight at once spills at R = 4)
        const vec     bx = LT::splat(jx), by = LT::splat(jy), bz = LT::splat(jz);
        vec           mj = bx;
        if constexpr (MJ) mj = LT::splat(jm);  // m_j / unit
#pragma unroll
        for (int h = 0; h < R; h += RB) {
            vec dx[RB], dy[RB], dz[RB], w[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) dx[r] = bx - px[h + r], dy[r] = by - py[h + r], dz[r] = bz - pz[h + r];
#pragma unroll
            for (int r = 0; r < RB; ++r) w[r] = LT::fma(dx[r], dx[r], eps2);
#pragma unroll
            for (int r = 0; r < RB; ++r) w[r] = LT::fma(dy[r], dy[r], w[r]);
#pragma unroll
            for (int r = 0; r < RB; ++r) w[r] = LT::fma(dz[r], dz[r], w[r]);
#pragma unroll
            for (int r = 0; r < RB; ++r) w[r] = LT::template coupling_rel<true>(eps2, w[r], consts);  // d2^(-3/2)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                vec wi = w[r], wj = w[r];
                if constexpr (MJ) wi = mj * w[r];
                if constexpr (MI) wj = mi[h + r] * w[r];
                ax[h + r] = LT::fma(dx[r], wi, ax[h + r]), ay[h + r] = LT::fma(dy[r], wi, ay[h + r]), az[h + r] = LT::fma(dz[r], wi, az[h + r]);
                rx = LT::fma(dx[r], wj, rx), ry = LT::fma(dy[r], wj, ry), rz = LT::fma(dz[r], wj, rz);
            }
        }
        // the body j and everything that belongs to it move on by one lane
        jx = rotate(jx), jy = rotate(jy), jz = rotate(jz);
        if constexpr (MJ) jm = rotate(jm);
        rx = rotate(rx), ry = rotate(ry), rz = rotate(rz);
    };
    auto rotation_steps = [&]<bool MI, bool MJ>(int trips, T& jx, T& jy, T& jz, T& jm, vec& rx, vec& ry, vec& rz, const vec (&mi)[R]) {  // trips x UNR steps: 64 for a unit, 16 for a quarter
#pragma unroll 1
        for (int it = 0; it < trips; ++it) {
#pragma unroll
            for (int v = 0; v < UNR; ++v) step.template operator()<MI, MJ>(jx, jy, jz, jm, rx, ry, rz, mi);
        }
    };

#ifdef NB_PAIR_STAMPS
    const unsigned long long stamp_loop = __builtin_amdgcn_s_memtime();
#endif
    // the sums of the quarters, per tail unit of the workgroup: [wg_tail <= S][quarter][component][body of the tile]
    T* const tail_sums = reinterpret_cast<T*>(smem_raw + kSumBytes + 256);
    vec4     cur       = n_items > 0 ? load_tile(item_unit(0), item_shift(0)) : vec4{};
    for (unsigned item = 0; item < n_items; ++item) {
        const unsigned u     = item_unit(item);
        const unsigned shift = item_shift(item);
        const bool     whole = item < n_whole;
        const vec4     next  = (item + 1) < n_items ? load_tile(item_unit(item + 1), item_shift(item + 1)) : cur;  // in flight across the steps below
#ifndef NB_NO_BALANCE
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
        const unsigned q     = u / TB;  // (diag: the block offset)
        const unsigned first = tile_first(u);
        const unsigned j     = first + ((static_cast<unsigned>(lane) - shift) & 63u);  // the body this lane holds at the start
        // Is the tile ONE species -- every body j real and of one usable mass?  Then no mass multiplies the i side: the wave's
        // i-side sums are in units of the tile's mass.  Likewise the block: its mass multiplies the reaction sums when they are stored.
        const T    m_tile       = first_lane(cur.w);
        const bool tile_uniform = usable_unit(m_tile) && __builtin_amdgcn_ballot_w64(!(j < j_end && __builtin_bit_cast(bits, cur.w) == __builtin_bit_cast(bits, m_tile))) == 0;
        T   jx = cur.x, jy = cur.y, jz = cur.z, jm = 0;
        vec rx = LT::splat(0), ry = LT::splat(0), rz = LT::splat(0);
        if (tile_uniform && __builtin_bit_cast(bits, m_tile) != __builtin_bit_cast(bits, unit)) change_unit(m_tile);
        if (!tile_uniform) jm = cur.w / unit;  // mixed masses: m_j / unit travels with the body j
        const T   scale = block_uniform ? m_block : T(1);  // what the reaction sums are still to be multiplied by
        const int trips = whole ? 64 / UNR : 16 / UNR;
        if (block_uniform) {
            vec none[R] = {};  // (these loops never read the masses of the bodies i)
            if (tile_uniform) rotation_steps.template operator()<false, false>(trips, jx, jy, jz, jm, rx, ry, rz, none);
            else rotation_steps.template operator()<false, true>(trips, jx, jy, jz, jm, rx, ry, rz, none);
        } else {
            vec mi[R];  // the masses of the bodies i: only these paths hold them, and only while they run
#pragma unroll
            for (int k = 0; k < I; ++k) {
                const unsigned i = block_base + k * 64 + lane;
                LT::set(mi[k / W], k % W, i < i_end ? old_pos[i].w : T(0));
            }
            if (tile_uniform) rotation_steps.template operator()<true, false>(trips, jx, jy, jz, jm, rx, ry, rz, mi);
            else rotation_steps.template operator()<true, true>(trips, jx, jy, jz, jm, rx, ry, rz, mi);
        }
        // 64 steps on: every sum is back in the lane of its body j.  Keep the reaction only when the partner does not list the pair too.
        const bool     symmetric = diag ? (q != 0 && !(even && q == Q)) : (s.keep != 0);
        const unsigned slot_of   = diag ? q - 1 : a;
        if (whole) {
            if (symmetric && j < j_end) {
                T* const out = s.react + static_cast<size_t>(slot_of) * 3 * s.react_plane + (j - s.react_origin);
                out[0]                                      = both_halves(rx) * scale;
                out[static_cast<size_t>(s.react_plane)]     = both_halves(ry) * scale;
                out[2 * static_cast<size_t>(s.react_plane)] = both_halves(rz) * scale;
            }
        } else if (symmetric) {
            // a quarter: sixteen steps on the lane holds the body that started sixteen lanes back; its sums wait in LDS for the other three
            const unsigned m   = (static_cast<unsigned>(lane) - shift - 16u) & 63u;
            T* const       out = tail_sums + (static_cast<size_t>(item_tail(item)) * 4 + my_quarter) * 3 * 64 + m;
            out[0] = both_halves(rx), out[64] = both_halves(ry), out[128] = both_halves(rz);
        }
        cur = next;
        ++done;
        if constexpr (kTwoLevel) {
            if (done % kFlush == 0) flush();
        }
        if (lane == 0) mine[slot] = done;
    }
    if (lane == 0) mine[slot] = 0xffffffffu;  // finished: never the one the others defer to
    __builtin_amdgcn_s_setprio(0);
#ifdef NB_PAIR_STAMPS
    {
        const unsigned long long stamp_done = __builtin_amdgcn_s_memtime();
        const unsigned           hw         = static_cast<unsigned>(__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4));   // HW_ID[15:0]
        const unsigned           xcc        = static_cast<unsigned>(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20));   // XCC_ID
        const unsigned           w          = blockIdx.x * S + static_cast<unsigned>(wave);
        if (lane == 0 && w < 8192) {
            nb_pair_stamps[w * 6 + 0] = stamp_start, nb_pair_stamps[w * 6 + 1] = stamp_loop, nb_pair_stamps[w * 6 + 2] = stamp_done;
            nb_pair_stamps[w * 6 + 3] = (static_cast<unsigned long long>(xcc) << 32) | hw;
            nb_pair_stamps[w * 6 + 4] = (static_cast<unsigned long long>(wave) << 32) | done;
            nb_pair_stamps[w * 6 + 5] = (static_cast<unsigned long long>(simd) << 32) | slot;
        }
    }
#endif
    {  // absolute units from here on
        const vec to_absolute = LT::splat(unit);
#pragma unroll
        for (int r = 0; r < R; ++r) ax[r] = ax[r] * to_absolute, ay[r] = ay[r] * to_absolute, az[r] = az[r] * to_absolute;
    }
    if constexpr (kTwoLevel) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            LT::set(ax[k / W], k % W, second[(0 * I + k) * 64] + LT::get(ax[k / W], k % W));
            LT::set(ay[k / W], k % W, second[(1 * I + k) * 64] + LT::get(ay[k / W], k % W));
            LT::set(az[k / W], k % W, second[(2 * I + k) * 64] + LT::get(az[k / W], k % W));
        }
    }

    // fold the S partial sums (waves 1..S-1 -> wave 0) through LDS, fixed order; the second-level sums are done with
    __syncthreads();
    // the tail units of the workgroup: a body's four quarter sums, in quarter order, then the one store a whole unit would have made
    for (unsigned k = static_cast<unsigned>(wave); k < wg_tail; k += S) {
        const unsigned u         = s.unit_begin + dealt + c + k * s.splits;
        const unsigned q         = u / TB;
        const bool     symmetric = diag ? (q != 0 && !(even && q == Q)) : (s.keep != 0);
        const unsigned j         = tile_first(u) + lane;
        if (!symmetric || j >= j_end) continue;
        const T        scale     = block_uniform ? m_block : T(1);
        const T* const in        = tail_sums + static_cast<size_t>(k) * 4 * 3 * 64 + lane;
        T* const       out       = s.react + static_cast<size_t>(diag ? q - 1 : a) * 3 * s.react_plane + (j - s.react_origin);
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) out[static_cast<size_t>(comp) * s.react_plane] = (((in[comp * 64] + in[(3 + comp) * 64]) + in[(6 + comp) * 64]) + in[(9 + comp) * 64]) * scale;
    }
    T* const red = sums;  // [(S-1)][3][I][64]
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
#pragma unroll 1
    for (int w = 1; w < S; ++w) {
#pragma unroll
        for (int k = 0; k < I; ++k) {
            LT::set(ax[k / W], k % W, LT::get(ax[k / W], k % W) + red[(((w - 1) * 3 + 0) * I + k) * 64 + lane]);
            LT::set(ay[k / W], k % W, LT::get(ay[k / W], k % W) + red[(((w - 1) * 3 + 1) * I + k) * 64 + lane]);
            LT::set(az[k / W], k % W, LT::get(az[k / W], k % W) + red[(((w - 1) * 3 + 2) * I + k) * 64 + lane]);
        }
    }
    T* const self = s.self + static_cast<size_t>(s.self_first + c) * 3 * s.self_plane;
#pragma unroll
    for (int k = 0; k < I; ++k) {
        const unsigned i = block_base + k * 64 + lane;
        if (i >= i_end) continue;
        const size_t at = i - s.self_origin;
        self[at]                                         = LT::get(ax[k / W], k % W);
        self[static_cast<size_t>(s.self_plane) + at]     = LT::get(ay[k / W], k % W);
        self[2 * static_cast<size_t>(s.self_plane) + at] = LT::get(az[k / W], k % W);
    }
}

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
            for (int c = 0; c < 3; ++c) v[c][u] = there ? r[(static_cast<size_t>(q + 4 * u) * 3 + c) * plane] : T(0);
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
            for (int c = 0; c < 3; ++c) t[c] += r[(static_cast<size_t>(k) * 3 + c) * plane];
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
                    for (int comp = 0; comp < 3; ++comp) x[i][comp] = (c0 + i) < set.slots ? s.self[(static_cast<size_t>(set.slot + c0 + i) * 3 + comp) * s.self_plane + k] : T(0);
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
    if (prepare_only) return hipSuccess;
    (void)hipGetLastError();  // a launch reports ITS OWN error: the call returns, and clears, the thread's last error whatever left it (a refused allocation, say)
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
