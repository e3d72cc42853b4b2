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
    const float s = x * r;
    const float h = r * 0.5f;
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);  // (rounds 1-3 refined s and h once more first; both forms are exact: strict_unit_mass_check.hip)
}
__device__ __forceinline__ float fast_div(float n, float d) {
    float       r  = __builtin_amdgcn_rcpf(d);
    const float e  = __builtin_fmaf(-d, r, 1.0f);
    r              = __builtin_fmaf(e, r, r);
    const float q  = n * r;
    const float e2 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e2, r, q);  // (one correction: exact for every pair of significands, strict_divide_exhaustive.hip)
}

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

__device__ void record64(Report* rep, unsigned long long a, unsigned long long b, double got, double want) {
    const unsigned long long k = atomicAdd(&rep->mismatches, 1ull);
    if (k < 8) {
        rep->first_a[k] = static_cast<uint32_t>(a >> 32), rep->first_b[k] = static_cast<uint32_t>(b >> 32);
        rep->got[k] = static_cast<uint32_t>(__double_as_longlong(got) & 0xffffffff), rep->want[k] = static_cast<uint32_t>(__double_as_longlong(want) & 0xffffffff);
    }
}
__device__ __forceinline__ unsigned long long f64_bits(int exponent, unsigned long long mantissa, unsigned long long sign) {
    return (sign << 63) | (static_cast<unsigned long long>(exponent + 1023) << 52) | (mantissa & 0xfffffffffffffull);
}
__device__ unsigned long long pattern64(unsigned i) {  // 128 structured mantissas
    if (i < 52) return 1ull << i;
    if (i < 104) return (0xfffffffffffffull << (i - 52)) & 0xfffffffffffffull;
    if (i == 104) return 0;
    if (i == 105) return 0xfffffffffffffull;
    return (i - 105) * 2 + 1;
}

// fp64: sqrt over x in [2^-100, 2^203], divide over n in {0} U 2^+-nexp, d in 2^[dlo, dhi]
__global__ void check_f64_random(Report* rep, int nexp, int dlo, int dhi, int slo, int shi, unsigned per_thread, uint64_t seed) {
    uint64_t           s = seed + (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 0x632be59bd9b4e019ull;
    unsigned long long n = 0;
    for (unsigned k = 0; k < per_thread; ++k) {
        const uint64_t r1 = splitmix(s), r2 = splitmix(s), r3 = splitmix(s);
        const int      ne = static_cast<int>(r3 % (2 * nexp + 1)) - nexp, de = dlo + static_cast<int>((r3 >> 20) % (dhi - dlo + 1)), se = slo + static_cast<int>((r3 >> 40) % (shi - slo + 1));
        unsigned long long nb = f64_bits(ne, r1, r3 >> 63);
        if (((r3 >> 52) & 0xff) == 0) nb = 0;  // +0 numerator now and then
        const unsigned long long db = f64_bits(de, r2, 0), sb = f64_bits(se, r1 ^ r2, 0);
        const double a = __longlong_as_double(nb), d = __longlong_as_double(db), x = __longlong_as_double(sb);
        const double wq = a / d, gq = fast_div_f64(a, d);
        if (__double_as_longlong(wq) != __double_as_longlong(gq)) record64(rep, nb, db, gq, wq);
        const double ws = sqrt(x), gs = fast_sqrt_f64(x);
        if (__double_as_longlong(ws) != __double_as_longlong(gs)) record64(rep, sb, 0, gs, ws);
        n += 2;
    }
    atomicAdd(&rep->tested, n);
}
__global__ void check_f64_structured(Report* rep, int nexp, int dlo, int dhi) {
    const unsigned pi = blockIdx.x, pj = threadIdx.x;  // 128 x 128 mantissa patterns
    unsigned long long n = 0;
    for (int ne = -nexp; ne <= nexp; ne += 7)
        for (int de = dlo; de <= dhi; de += 13) {
            const unsigned long long nb = f64_bits(ne, pattern64(pi), 0), db = f64_bits(de, pattern64(pj), 0);
            const double a = __longlong_as_double(nb), d = __longlong_as_double(db);
            const double wq = a / d, gq = fast_div_f64(a, d);
            if (__double_as_longlong(wq) != __double_as_longlong(gq)) record64(rep, nb, db, gq, wq);
            const double ws = sqrt(d), gs = fast_sqrt_f64(d);
            if (__double_as_longlong(ws) != __double_as_longlong(gs)) record64(rep, db, 0, gs, ws);
            n += 2;
        }
    atomicAdd(&rep->tested, n);
}
// does v_div_scale_f64 leave operands of the window alone (no scaling, VCC clear)?  every exponent pair x 128 mantissas
__global__ void check_div_scale_identity(Report* rep, int nexp, int dlo, int dhi) {
    const unsigned pat = threadIdx.x;
    unsigned long long n = 0;
    for (int ne = -nexp + static_cast<int>(blockIdx.x); ne <= nexp; ne += static_cast<int>(gridDim.x))
        for (int de = dlo; de <= dhi; ++de) {
            const double a = __longlong_as_double(f64_bits(ne, pattern64(pat), 0)), d = __longlong_as_double(f64_bits(de, pattern64(127 - pat), 0));
            bool         f1 = false, f2 = false;
            const double sd = __builtin_amdgcn_div_scale(a, d, false, &f1);  // scaled denominator
            const double sn = __builtin_amdgcn_div_scale(a, d, true, &f2);   // scaled numerator
            if (__double_as_longlong(sd) != __double_as_longlong(d) || __double_as_longlong(sn) != __double_as_longlong(a) || f1 || f2)
                record64(rep, __double_as_longlong(a), __double_as_longlong(d), sd, sn);
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

    // ---- fp64 (window: |n| in {+0} U 2^+-100, d = r2^2 in 2^[-200, 406], sqrt argument in 2^[-100, 203]) ----
    hipLaunchKernelGGL(check_div_scale_identity, dim3(201), dim3(128), 0, 0, rep, 100, -200, 406);
    show("fp64 v_div_scale is the identity in the window", rep);
    hipLaunchKernelGGL(check_f64_structured, dim3(128), dim3(128), 0, 0, rep, 100, -200, 406);
    show("fp64 div+sqrt structured mantissas", rep);
    for (int round = 0; round < 4; ++round) {
        hipLaunchKernelGGL(check_f64_random, dim3(4096), dim3(256), 0, 0, rep, 100, -200, 406, -100, 203, 2048u, 0xabcdefull + round);
        show("fp64 div+sqrt random, in window", rep);
    }
    hipLaunchKernelGGL(check_f64_random, dim3(4096), dim3(256), 0, 0, rep, 400, -900, 900, -1000, 1000, 1024u, 7ull);
    show("fp64 div+sqrt random, far outside the window", rep);
    return 0;
}
