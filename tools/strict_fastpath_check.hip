// tools/strict_fastpath_check.hip -- is the scaling-free sqrt / divide used by the STRICT kernel's fast path
// (csrc/nbody_strict.hip) bit-identical to the IEEE-754 correctly rounded sqrtf and `/` hipcc emits by default?
//
//   sqrt:  EXHAUSTIVE over every positive normal float in [2^-100, 2^100]  (the fast path's guard range is narrower)
//   div :  2^34 random (numerator, denominator) pairs inside the guard ranges + structured mantissas (all-ones,
//          single-bit, near powers of two) x exponents, against `/`
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/strict_fastpath_check.hip -o tools/strict_fastpath_check
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

// The two sequences under test -- keep in sync with csrc/nbody_strict.hip (fast_sqrt / fast_div).
__device__ __forceinline__ float fast_sqrt(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float       s = x * r;
    float       h = r * 0.5f;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h             = __builtin_fmaf(h, e, h);
    s             = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float fast_div(float n, float d) {
    float       r  = __builtin_amdgcn_rcpf(d);
    const float e  = __builtin_fmaf(-d, r, 1.0f);
    r              = __builtin_fmaf(e, r, r);
    float       q  = n * r;
    const float e2 = __builtin_fmaf(-d, q, n);
    q              = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e3, r, q);
}

struct Report {
    unsigned long long mismatches;
    unsigned long long tested;
    uint32_t           first_a[8], first_b[8], got[8], want[8];
};

__device__ void record(Report* rep, uint32_t a, uint32_t b, float got, float want) {
    const unsigned long long k = atomicAdd(&rep->mismatches, 1ull);
    if (k < 8) {
        rep->first_a[k] = a, rep->first_b[k] = b;
        rep->got[k] = __float_as_uint(got), rep->want[k] = __float_as_uint(want);
    }
}

__global__ void check_sqrt(Report* rep, uint32_t lo_bits, uint32_t hi_bits) {
    const uint64_t     stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long n      = 0;
    for (uint64_t b = lo_bits + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < hi_bits; b += stride) {
        const float x    = __uint_as_float(static_cast<uint32_t>(b));
        const float want = sqrtf(x);
        const float got  = fast_sqrt(x);
        if (__float_as_uint(want) != __float_as_uint(got)) record(rep, static_cast<uint32_t>(b), 0, got, want);
        ++n;
    }
    atomicAdd(&rep->tested, n);
}

__device__ __forceinline__ uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z          = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z          = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// numerator exponent in [-nexp, nexp] (or exactly 0), denominator exponent in [-dexp, dexp]
__global__ void check_div_random(Report* rep, int nexp, int dexp, unsigned per_thread, uint64_t seed) {
    uint64_t           s = seed + (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 0x632be59bd9b4e019ull;
    unsigned long long n = 0;
    for (unsigned k = 0; k < per_thread; ++k) {
        const uint64_t r  = splitmix(s);
        const uint64_t r2 = splitmix(s);
        const uint32_t nm = static_cast<uint32_t>(r) & 0x7fffffu, dm = static_cast<uint32_t>(r >> 32) & 0x7fffffu;
        const int      ne = static_cast<int>(r2 % (2 * nexp + 1)) - nexp, de = static_cast<int>((r2 >> 20) % (2 * dexp + 1)) - dexp;
        const uint32_t sign = static_cast<uint32_t>(r2 >> 63) << 31;
        uint32_t       nb = sign | (static_cast<uint32_t>(ne + 127) << 23) | nm;
        const uint32_t db = (static_cast<uint32_t>(de + 127) << 23) | dm;
        if ((r2 >> 50 & 0xff) == 0) nb = sign;  // a zero numerator now and then (zero-mass padding bodies)
        const float a = __uint_as_float(nb), d = __uint_as_float(db);
        const float want = a / d;
        const float got  = fast_div(a, d);
        if (__float_as_uint(want) != __float_as_uint(got)) record(rep, nb, db, got, want);
        ++n;
    }
    atomicAdd(&rep->tested, n);
}

// structured mantissas: every pair of (pattern_i, pattern_j) x exponents
__device__ uint32_t pattern(unsigned i) {
    // 0..22: single bit; 23..45: all ones above bit; 46..68: all ones below bit; 69: 0; 70: 0x7fffff; 71..: small odd values
    if (i < 23) return 1u << i;
    if (i < 46) return (0x7fffffu << (i - 23)) & 0x7fffffu;
    if (i < 69) return (1u << (i - 46)) - 1u;
    if (i == 69) return 0;
    if (i == 70) return 0x7fffffu;
    return (i - 70) * 2 + 1;
}
constexpr unsigned kPatterns = 71 + 57;  // 128

__global__ void check_div_structured(Report* rep, int nexp, int dexp) {
    const unsigned pi = blockIdx.x, pj = threadIdx.x;  // 128 x 128
    if (pi >= kPatterns || pj >= kPatterns) return;
    unsigned long long n = 0;
    for (int ne = -nexp; ne <= nexp; ne += 3)
        for (int de = -dexp; de <= dexp; de += 5) {
            const uint32_t nb = (static_cast<uint32_t>(ne + 127) << 23) | pattern(pi);
            const uint32_t db = (static_cast<uint32_t>(de + 127) << 23) | pattern(pj);
            const float    a = __uint_as_float(nb), d = __uint_as_float(db);
            const float    want = a / d, got = fast_div(a, d);
            if (__float_as_uint(want) != __float_as_uint(got)) record(rep, nb, db, got, want);
            ++n;
        }
    atomicAdd(&rep->tested, n);
}

static void show(const char* what, Report* dev) {
    Report r;
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&r, dev, sizeof(r), hipMemcpyDeviceToHost));
    printf("%-48s tested %llu  mismatches %llu\n", what, r.tested, r.mismatches);
    for (unsigned k = 0; k < 8 && k < r.mismatches; ++k) printf("    a=%08x b=%08x got=%08x want=%08x\n", r.first_a[k], r.first_b[k], r.got[k], r.want[k]);
    fflush(stdout);
    CHECK(hipMemset(dev, 0, sizeof(r)));
}

static uint32_t bits_of_pow2(int e) { return static_cast<uint32_t>(e + 127) << 23; }

int main() {
    Report* rep;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMemset(rep, 0, sizeof(Report)));

    hipLaunchKernelGGL(check_sqrt, dim3(4096), dim3(256), 0, 0, rep, bits_of_pow2(-100), bits_of_pow2(100));
    show("sqrt exhaustive [2^-100, 2^100)", rep);
    hipLaunchKernelGGL(check_sqrt, dim3(4096), dim3(256), 0, 0, rep, bits_of_pow2(-126), bits_of_pow2(-100));
    show("sqrt exhaustive [2^-126, 2^-100) (outside guard)", rep);
    hipLaunchKernelGGL(check_sqrt, dim3(4096), dim3(256), 0, 0, rep, bits_of_pow2(100), bits_of_pow2(127));
    show("sqrt exhaustive [2^100, 2^127) (outside guard)", rep);

    hipLaunchKernelGGL(check_div_structured, dim3(kPatterns), dim3(kPatterns), 0, 0, rep, 40, 80);
    show("div structured |n| 2^+-40, d 2^+-80", rep);
    for (int round = 0; round < 4; ++round) {
        hipLaunchKernelGGL(check_div_random, dim3(4096), dim3(256), 0, 0, rep, 40, 80, 4096u, 0x1234567ull + round);
        show("div random |n| in {0} U 2^+-40, d in 2^+-80", rep);
    }
    hipLaunchKernelGGL(check_div_random, dim3(4096), dim3(256), 0, 0, rep, 60, 120, 4096u, 99ull);
    show("div random |n| 2^+-60, d 2^+-120 (outside guard)", rep);
    return 0;
}
