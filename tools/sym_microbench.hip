// tools/sym_microbench.hip -- round 3: what would a PAIRWISE (Newton's third law) inner loop cost on gfx950?
//
// The production FAST loop evaluates every directed interaction: per packed pair of bodies i and one body j
// 11 v_pk_* + 2 v_rsq_f32 = 61.5 cycles per SIMD for 2 directed interactions per lane (profiles/round2_loop_microbench.txt).
// Evaluating each unordered pair once and applying it to both bodies needs the reaction sums of the bodies j somewhere a
// lane can reach.  Here: 64 bodies j sit one per lane and meet the lanes' bodies i by ROTATING within each row of 16 lanes
// (DPP row_ror), three ways:
//   B  : the bodies j and their packed reaction sums {from i0, from i1} physically rotate (9 v_mov_b32_dpp per step)
//   A  : nothing rotates: differences read the body j through a DPP source (v_sub_f32_dpp row_ror:k), reaction terms are
//        added into the home lane through the complementary rotation (v_add_f32_dpp row_ror:16-k)
//   H  : positions read through DPP sources, reaction sums rotate (6 v_mov_b32_dpp)
//   ONE: the one-sided loop on the same data layout (body j per lane, no reaction) -- the wave-split kernel's loop
//   L  : (round 4) nothing rotates on the vector ALU: the 64 bodies j of a tile sit in a per-wave LDS buffer (twice, so that no
//        index wraps); at step k lane l reads body l+k (one ds_read_b128, requested a step ahead) and adds its packed reaction
//        term, halves summed (3 v_add_f32), into the body's LDS sums (3 ds_add_f32, no return) -- 9 DPP moves less per step
// Reports true shader cycles (s_memtime) per rotation step and per DIRECTED interaction, per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 tools/sym_microbench.hip -o tools/sym_microbench && tools/sym_microbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int CTRL> __device__ __forceinline__ float dpp(float x) {
    if constexpr (CTRL == 0x120 || CTRL == 0x130) return x;  // rotate by 0 / 16
    else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f inv3_of(v2f d2) {
    const v2f inv = v2f{__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
    return inv * (inv * inv);
}

struct Out {
    unsigned long long cycles;
    float              check;
};

enum { kB = 0, kA = 1, kH = 2, kOne = 3, kBW = 4, kL = 5, kL6 = 6 };  // kBW: as B but a full 64-lane rotation (DPP wave_ror:1); kL6: L with 6 ds_add (no v_add)

template <int R> struct State {
    v2f   px[R], py[R], pz[R], ax[R], ay[R], az[R];
    float jx, jy, jz;
    v2f   rx, ry, rz;  // reaction sums of the body j: {from the low bodies i, from the high ones}
    float sx, sy, sz;  // ... or unpacked (variant A)
    v2f   eps2;
};

// one rotation step, variant B: everything that belongs to the body j moves on by one lane afterwards
template <int R, int ROT = 0x121> __device__ __forceinline__ void step_b(State<R>& s) {
    const v2f bx = v2f{s.jx, s.jx}, by = v2f{s.jy, s.jy}, bz = v2f{s.jz, s.jz};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const v2f dx = bx - s.px[r], dy = by - s.py[r], dz = bz - s.pz[r];
        v2f       d2 = fma2(dx, dx, s.eps2);
        d2           = fma2(dy, dy, d2);
        d2           = fma2(dz, dz, d2);
        const v2f w  = inv3_of(d2);
        s.ax[r] = fma2(dx, w, s.ax[r]), s.ay[r] = fma2(dy, w, s.ay[r]), s.az[r] = fma2(dz, w, s.az[r]);
        s.rx = fma2(dx, w, s.rx), s.ry = fma2(dy, w, s.ry), s.rz = fma2(dz, w, s.rz);
    }
    s.jx = dpp<ROT>(s.jx), s.jy = dpp<ROT>(s.jy), s.jz = dpp<ROT>(s.jz);
    s.rx = v2f{dpp<ROT>(s.rx.x), dpp<ROT>(s.rx.y)};
    s.ry = v2f{dpp<ROT>(s.ry.x), dpp<ROT>(s.ry.y)};
    s.rz = v2f{dpp<ROT>(s.rz.x), dpp<ROT>(s.rz.y)};
}

// variant H: the body j is read through a DPP source (rotation K), its reaction sums rotate
template <int R, int K> __device__ __forceinline__ void step_h(State<R>& s) {
    const float jx = dpp<0x120 + K>(s.jx), jy = dpp<0x120 + K>(s.jy), jz = dpp<0x120 + K>(s.jz);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const v2f dx = v2f{jx - s.px[r].x, jx - s.px[r].y}, dy = v2f{jy - s.py[r].x, jy - s.py[r].y}, dz = v2f{jz - s.pz[r].x, jz - s.pz[r].y};
        v2f       d2 = fma2(dx, dx, s.eps2);
        d2           = fma2(dy, dy, d2);
        d2           = fma2(dz, dz, d2);
        const v2f w  = inv3_of(d2);
        s.ax[r] = fma2(dx, w, s.ax[r]), s.ay[r] = fma2(dy, w, s.ay[r]), s.az[r] = fma2(dz, w, s.az[r]);
        s.rx = fma2(dx, w, s.rx), s.ry = fma2(dy, w, s.ry), s.rz = fma2(dz, w, s.rz);
    }
    s.rx = v2f{dpp<0x121>(s.rx.x), dpp<0x121>(s.rx.y)};
    s.ry = v2f{dpp<0x121>(s.ry.x), dpp<0x121>(s.ry.y)};
    s.rz = v2f{dpp<0x121>(s.rz.x), dpp<0x121>(s.rz.y)};
}

// variant A: nothing rotates
template <int R, int K> __device__ __forceinline__ void step_a(State<R>& s) {
    const float jx = dpp<0x120 + K>(s.jx), jy = dpp<0x120 + K>(s.jy), jz = dpp<0x120 + K>(s.jz);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const v2f dx = v2f{jx - s.px[r].x, jx - s.px[r].y}, dy = v2f{jy - s.py[r].x, jy - s.py[r].y}, dz = v2f{jz - s.pz[r].x, jz - s.pz[r].y};
        v2f       d2 = fma2(dx, dx, s.eps2);
        d2           = fma2(dy, dy, d2);
        d2           = fma2(dz, dz, d2);
        const v2f w  = inv3_of(d2);
        const v2f cx = dx * w, cy = dy * w, cz = dz * w;
        s.ax[r] += cx, s.ay[r] += cy, s.az[r] += cz;
        constexpr int back = 0x120 + (16 - K) % 16;
        s.sx += dpp<back>(cx.x), s.sx += dpp<back>(cx.y);
        s.sy += dpp<back>(cy.x), s.sy += dpp<back>(cy.y);
        s.sz += dpp<back>(cz.x), s.sz += dpp<back>(cz.y);
    }
}

// variant L: the body j comes from LDS (already in `j`), the reaction term goes to LDS
template <int R, bool SIX> __device__ __forceinline__ void step_l(State<R>& s, const float4 j, float* react /* this lane's slot at step k: &sums[lane + k] */) {
    const v2f bx = v2f{j.x, j.x}, by = v2f{j.y, j.y}, bz = v2f{j.z, j.z};
    v2f       tx, ty, tz;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const v2f dx = bx - s.px[r], dy = by - s.py[r], dz = bz - s.pz[r];
        v2f       d2 = fma2(dx, dx, s.eps2);
        d2           = fma2(dy, dy, d2);
        d2           = fma2(dz, dz, d2);
        const v2f w  = inv3_of(d2);
        s.ax[r] = fma2(dx, w, s.ax[r]), s.ay[r] = fma2(dy, w, s.ay[r]), s.az[r] = fma2(dz, w, s.az[r]);
        if (r == 0) tx = dx * w, ty = dy * w, tz = dz * w;
        else tx = fma2(dx, w, tx), ty = fma2(dy, w, ty), tz = fma2(dz, w, tz);
    }
    if constexpr (SIX) {
        __hip_atomic_fetch_add(react, tx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT), __hip_atomic_fetch_add(react, tx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(react + 128, ty.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT), __hip_atomic_fetch_add(react + 128, ty.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(react + 256, tz.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT), __hip_atomic_fetch_add(react + 256, tz.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
        __hip_atomic_fetch_add(react, tx.x + tx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(react + 128, ty.x + ty.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(react + 256, tz.x + tz.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

// the one-sided loop on this data layout
template <int R> __device__ __forceinline__ void step_one(State<R>& s) {
    const v2f bx = v2f{s.jx, s.jx}, by = v2f{s.jy, s.jy}, bz = v2f{s.jz, s.jz};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const v2f dx = bx - s.px[r], dy = by - s.py[r], dz = bz - s.pz[r];
        v2f       d2 = fma2(dx, dx, s.eps2);
        d2           = fma2(dy, dy, d2);
        d2           = fma2(dz, dz, d2);
        const v2f w  = inv3_of(d2);
        s.ax[r] = fma2(dx, w, s.ax[r]), s.ay[r] = fma2(dy, w, s.ay[r]), s.az[r] = fma2(dz, w, s.az[r]);
    }
    s.jx = dpp<0x121>(s.jx), s.jy = dpp<0x121>(s.jy), s.jz = dpp<0x121>(s.jz);
}

template <int R, int K> __device__ __forceinline__ void sixteen_a(State<R>& s) {
    if constexpr (K < 16) {
        step_a<R, K>(s);
        sixteen_a<R, K + 1>(s);
    }
}
template <int R, int K> __device__ __forceinline__ void sixteen_h(State<R>& s) {
    if constexpr (K < 16) {
        step_h<R, K>(s);
        sixteen_h<R, K + 1>(s);
    }
}

template <int VARIANT, int R> __global__ __launch_bounds__(1024) void bench(Out* out, const float4* bodies, int rounds) {
    State<R>  s;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 a = bodies[(blockIdx.x * 1024 + threadIdx.x) % 4096], b = bodies[(blockIdx.x * 1024 + threadIdx.x + 977 * (r + 1)) % 4096];
        s.px[r] = v2f{a.x, b.x}, s.py[r] = v2f{a.y, b.y}, s.pz[r] = v2f{a.z, b.z};
        s.ax[r] = s.ay[r] = s.az[r] = v2f{0, 0};
    }
    s.eps2 = v2f{0.01f, 0.01f};
    s.rx = s.ry = s.rz = v2f{0, 0};
    s.sx = s.sy = s.sz = 0;
    if constexpr (VARIANT == kL || VARIANT == kL6) {
        // per wave: 128 bodies (the tile twice) and 3 x 128 reaction sums
        __shared__ float4 tile_pos[16][128];
        __shared__ float  tile_sum[16][3 * 128];
        const int wave = threadIdx.x >> 6;
        float4* const pos = tile_pos[wave];
        float* const  sum = tile_sum[wave];
        const unsigned long long t0 = __builtin_readcyclecounter();
        float kept = 0;
#pragma unroll 1
        for (int tile = 0; tile < rounds / 4; ++tile) {
            const float4 mine = bodies[(tile * 64 + lane) & 4095];
            pos[lane] = mine, pos[lane + 64] = mine;
#pragma unroll
            for (int c = 0; c < 3; ++c) sum[c * 128 + lane] = 0, sum[c * 128 + lane + 64] = 0;
            float4 j = pos[lane];
#pragma unroll 1
            for (int k0 = 0; k0 < 64; k0 += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 next = pos[lane + k0 + q + 1 < 128 ? lane + k0 + q + 1 : 127];  // (requested a step ahead)
                    step_l<R, VARIANT == kL6>(s, j, sum + lane + k0 + q);
                    j = next;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) kept += sum[c * 128 + lane] + sum[c * 128 + lane + 64];
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float check = kept;
#pragma unroll
        for (int r = 0; r < R; ++r) check += s.ax[r].x + s.ax[r].y + s.ay[r].x + s.ay[r].y + s.az[r].x + s.az[r].y;
        if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = Out{t1 - t0, check};
        return;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int k = 0; k < rounds; ++k) {
        const float4 j = bodies[(k * 64 + lane) & 4095];  // the next 64 bodies j, one per lane
        s.jx = j.x, s.jy = j.y, s.jz = j.z;
        if constexpr (VARIANT == kB) {
#pragma unroll
            for (int q = 0; q < 16; ++q) step_b<R>(s);
        } else if constexpr (VARIANT == kBW) {
#pragma unroll
            for (int q = 0; q < 16; ++q) step_b<R, 0x13C>(s);
        } else if constexpr (VARIANT == kA) {
            sixteen_a<R, 0>(s);
        } else if constexpr (VARIANT == kH) {
            sixteen_h<R, 0>(s);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) step_one<R>(s);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float check = s.rx.x + s.rx.y + s.ry.x + s.ry.y + s.rz.x + s.rz.y + s.sx + s.sy + s.sz + s.jx;
#pragma unroll
    for (int r = 0; r < R; ++r) check += s.ax[r].x + s.ax[r].y + s.ay[r].x + s.ay[r].y + s.az[r].x + s.az[r].y;
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = Out{t1 - t0, check};
}

template <int VARIANT, int R> void run(const char* name, const float4* bodies, Out* out, int block) {
    const int rounds = 8000, grid = 256, waves = grid * block / 64;
    std::vector<Out> host(waves);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((bench<VARIANT, R>), dim3(grid), dim3(block), 0, nullptr, out, bodies, rounds);
        hipDeviceSynchronize();
    }
    hipMemcpy(host.data(), out, sizeof(Out) * waves, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (auto& o : host) c.push_back(static_cast<double>(o.cycles));
    std::sort(c.begin(), c.end());
    const double per_simd  = block / 256.0;                        // waves per SIMD
    const double steps     = rounds * 16.0;                        // rotation steps per wave
    const double step_cyc  = c[c.size() / 2] / steps / per_simd;   // SIMD cycles per step (all its waves ran interleaved)
    const double directed  = (VARIANT == kOne ? 1.0 : 2.0) * 2 * R;  // directed interactions per lane and step
    std::printf("%-42s R=%d %d waves/SIMD: %7.1f cycles/step  %6.2f cycles per directed interaction and lane  (x%.2f vs 30.75)\n", name, R, static_cast<int>(per_simd), step_cyc,
                step_cyc / directed, 30.75 / (step_cyc / directed));
}

int main(int argc, char**) {
    float4* bodies = nullptr;
    Out*    out    = nullptr;
    hipMalloc(&bodies, sizeof(float4) * 4096);
    hipMalloc(&out, sizeof(Out) * 256 * 16);
    std::vector<float4> host(4096);
    unsigned seed = 12345;
    auto     rnd  = [&] { seed = seed * 1664525u + 1013904223u; return static_cast<float>(seed >> 8) / 16777216.0f * 10.0f - 5.0f; };
    for (auto& b : host) b = float4{rnd(), rnd(), rnd(), 1.0f};
    hipMemcpy(bodies, host.data(), sizeof(float4) * 4096, hipMemcpyHostToDevice);
    if (argc > 1) {  // quick mode (for rocprofv3 --pmc): the production candidate only, long run
        run<kBW, 4>("pairwise B, wave_ror:1 (64-lane rotation)", bodies, out, 1024);
        run<kL, 4>("pairwise L (j from LDS, sums to LDS, 3 adds)", bodies, out, 1024);
        run<kL6, 4>("pairwise L6 (j from LDS, 6 ds_add)", bodies, out, 1024);
        run<kBW, 2>("pairwise B, wave_ror:1 (64-lane rotation)", bodies, out, 1024);
        run<kL, 2>("pairwise L (j from LDS, sums to LDS, 3 adds)", bodies, out, 1024);
        run<kBW, 4>("pairwise B, wave_ror:1 (64-lane rotation)", bodies, out, 512);
        run<kL, 4>("pairwise L (j from LDS, sums to LDS, 3 adds)", bodies, out, 512);
        run<kOne, 1>("one-sided, body j per lane", bodies, out, 1024);
        return 0;
    }
    for (int block : {256, 512, 1024}) {
        run<kOne, 1>("one-sided, body j per lane", bodies, out, block);
        run<kOne, 2>("one-sided, body j per lane", bodies, out, block);
        run<kB, 1>("pairwise B (j + reaction sums rotate)", bodies, out, block);
        run<kB, 2>("pairwise B (j + reaction sums rotate)", bodies, out, block);
        run<kB, 4>("pairwise B (j + reaction sums rotate)", bodies, out, block);
        run<kBW, 2>("pairwise B, wave_ror:1 (64-lane rotation)", bodies, out, block);
        run<kBW, 4>("pairwise B, wave_ror:1 (64-lane rotation)", bodies, out, block);
        run<kH, 1>("pairwise H (DPP-source j, sums rotate)", bodies, out, block);
        run<kH, 2>("pairwise H (DPP-source j, sums rotate)", bodies, out, block);
        run<kA, 1>("pairwise A (nothing rotates)", bodies, out, block);
        run<kA, 2>("pairwise A (nothing rotates)", bodies, out, block);
    }
    return 0;
}
