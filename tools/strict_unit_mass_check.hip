// tools/strict_unit_mass_check.hip -- which SHORTER sequences are still the correctly rounded IEEE results, exhaustively?
//
//   sqrt(x)  for every float x in [2^-100, 2^127): candidates with fewer refinement steps than the STRICT kernel's 7-op form
//   1 / d    for every float d in [2^-100, 2^100]: the reciprocal a unit-mass system needs (m / (r2*r2) with m == 1.0f),
//            candidates of 2 and 4 ops after v_rcp_f32, against `1.0f / d` as hipcc emits it (correctly rounded)
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/strict_unit_mass_check.hip -o tools/strict_unit_mass_check
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

template <int V> __device__ __forceinline__ float cand_sqrt(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float       s = x * r;
    float       h = r * 0.5f;
    if constexpr (V == 7) {  // the kernel's form (LLVM's no-denormal lowering)
        const float e = __builtin_fmaf(-h, s, 0.5f);
        h             = __builtin_fmaf(h, e, h);
        s             = __builtin_fmaf(s, e, s);
    } else if constexpr (V == 6) {  // without refining h
        const float e = __builtin_fmaf(-h, s, 0.5f);
        s             = __builtin_fmaf(s, e, s);
    } else if constexpr (V == 61) {  // without refining s
        const float e = __builtin_fmaf(-h, s, 0.5f);
        h             = __builtin_fmaf(h, e, h);
    }  // V == 4: neither
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}

template <int V> __device__ __forceinline__ float cand_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r       = __builtin_fmaf(e, r, r);
    if constexpr (V >= 4) {
        e = __builtin_fmaf(-d, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    if constexpr (V >= 6) {
        e = __builtin_fmaf(-d, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return r;
}
// the kernel's general divide with n = 1 (must be 0 mismatches: it is hipcc's own sequence)
__device__ __forceinline__ float div7_one(float d) {
    float       r  = __builtin_amdgcn_rcpf(d);
    const float e  = __builtin_fmaf(-d, r, 1.0f);
    r              = __builtin_fmaf(e, r, r);
    float       q  = r;
    const float e2 = __builtin_fmaf(-d, q, 1.0f);
    q              = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(e3, r, q);
}

// 1 / RN(x*x) for the r2 window WITHOUT v_rcp_f32: the seed comes from the v_rsq_f32 the sqrt needs anyway (rs^4 ~ 1/x^2)
template <int NEWTON> __device__ __forceinline__ float cand_rcp_of_square_from_rsq(float x) {
    const float rs = __builtin_amdgcn_rsqf(x);
    const float d  = x * x;
    const float y  = rs * rs;
    float       q  = y * y;
#pragma unroll
    for (int k = 0; k < NEWTON; ++k) {
        const float e = __builtin_fmaf(-d, q, 1.0f);
        q             = __builtin_fmaf(e, q, q);
    }
    return q;
}
// the unit-mass chain as the kernel runs it: r = sqrt(x) (4-op form), q = 1/RN(x*x) (rcp + 1 Newton), mr3 = q*r
__device__ __forceinline__ float chain_kernel(float x) {
    const float rs = __builtin_amdgcn_rsqf(x);
    const float s = x * rs, h = rs * 0.5f;
    const float r = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    const float d = x * x;
    float       q = __builtin_amdgcn_rcpf(d);
    q             = __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), q, q);
    return q * r;
}
template <int NEWTON> __device__ __forceinline__ float chain_from_rsq(float x) {
    const float rs = __builtin_amdgcn_rsqf(x);
    const float s = x * rs, h = rs * 0.5f;
    const float r = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    return cand_rcp_of_square_from_rsq<NEWTON>(x) * r;
}

struct Report {
    unsigned long long mismatches, tested;
    uint32_t           first[8], got[8], want[8];
};

template <typename F, typename W> __global__ void sweep(Report* rep, uint32_t lo_bits, uint32_t hi_bits, F candidate, W reference) {
    const uint64_t     stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long n = 0, bad = 0;
    for (uint64_t b = lo_bits + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < hi_bits; b += stride) {
        const float x = __uint_as_float(static_cast<uint32_t>(b));
        const float want = reference(x), got = candidate(x);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const unsigned long long k = atomicAdd(&rep->mismatches, 1ull);
            if (k < 8) rep->first[k] = static_cast<uint32_t>(b), rep->got[k] = __float_as_uint(got), rep->want[k] = __float_as_uint(want);
            ++bad;
        }
        ++n;
    }
    atomicAdd(&rep->tested, n);
}

static uint32_t bits_of_pow2(int e) { return static_cast<uint32_t>(e + 127) << 23; }

template <typename F, typename W> static void run(const char* what, int lo_exp, int hi_exp, F candidate, W reference) {
    Report* rep;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    hipLaunchKernelGGL((sweep<F, W>), dim3(256 * 8), dim3(256), 0, 0, rep, bits_of_pow2(lo_exp), bits_of_pow2(hi_exp), candidate, reference);
    CHECK(hipDeviceSynchronize());
    Report r;
    CHECK(hipMemcpy(&r, rep, sizeof(r), hipMemcpyDeviceToHost));
    printf("%-66s [2^%d, 2^%d)  tested %llu  mismatches %llu\n", what, lo_exp, hi_exp, r.tested, r.mismatches);
    for (unsigned k = 0; k < 4 && k < r.mismatches; ++k) printf("    x=%08x got=%08x want=%08x\n", r.first[k], r.got[k], r.want[k]);
    CHECK(hipFree(rep));
}

int main() {
    auto ref_sqrt = [] __device__(float x) { return sqrtf(x); };
    auto ref_rcp  = [] __device__(float d) { return 1.0f / d; };
    run("sqrt: rsq + 7 ops (kernel)", -100, 127, [] __device__(float x) { return cand_sqrt<7>(x); }, ref_sqrt);
    run("sqrt: rsq + 6 ops (h not refined)", -100, 127, [] __device__(float x) { return cand_sqrt<6>(x); }, ref_sqrt);
    run("sqrt: rsq + 6 ops (s not refined)", -100, 127, [] __device__(float x) { return cand_sqrt<61>(x); }, ref_sqrt);
    run("sqrt: rsq + 4 ops (one residual correction only)", -100, 127, [] __device__(float x) { return cand_sqrt<4>(x); }, ref_sqrt);
    run("1/d: general divide with n = 1 (kernel, 7 ops)", -100, 101, [] __device__(float d) { return div7_one(d); }, ref_rcp);
    run("1/d: rcp + 1 Newton step (2 ops)", -100, 101, [] __device__(float d) { return cand_rcp<2>(d); }, ref_rcp);
    run("1/d: rcp + 2 Newton steps (4 ops)", -100, 101, [] __device__(float d) { return cand_rcp<4>(d); }, ref_rcp);
    run("1/d: rcp + 3 Newton steps (6 ops)", -100, 101, [] __device__(float d) { return cand_rcp<6>(d); }, ref_rcp);
    // the whole unit-mass chain as a function of r2 alone, over the kernel's r2 window [2^-39, 2^40] (and a margin)
    auto ref_chain = [] __device__(float x) { return (1.0f / (x * x)) * sqrtf(x); };
    auto ref_rsq   = [] __device__(float x) { return 1.0f / (x * x); };
    run("chain mr3(r2) = (1/(r2*r2))*sqrt(r2): kernel form", -45, 46, [] __device__(float x) { return chain_kernel(x); }, ref_chain);
    run("1/(r2*r2): seed rs^4 from the sqrt's v_rsq, 1 Newton step (no v_rcp)", -45, 46, [] __device__(float x) { return cand_rcp_of_square_from_rsq<1>(x); }, ref_rsq);
    run("1/(r2*r2): seed rs^4 from the sqrt's v_rsq, 2 Newton steps (no v_rcp)", -45, 46, [] __device__(float x) { return cand_rcp_of_square_from_rsq<2>(x); }, ref_rsq);
    run("chain with the rs^4 seed, 1 Newton step", -45, 46, [] __device__(float x) { return chain_from_rsq<1>(x); }, ref_chain);
    run("chain with the rs^4 seed, 2 Newton steps", -45, 46, [] __device__(float x) { return chain_from_rsq<2>(x); }, ref_chain);
    // negative denominators cannot occur (d = r2*r2 > 0)
    return 0;
}
