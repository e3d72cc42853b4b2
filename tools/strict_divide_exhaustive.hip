// tools/strict_divide_exhaustive.hip -- is a SHORTER fp32 divide still the correctly rounded quotient?  Exhaustively, over every
// pair of significands: n = 1.f x 2^0 (2^23 values) against d = 1.g x 2^0 (2^23 values) = 2^46 quotients, each compared with `/`
// as hipcc emits it (correctly rounded).  Inside the STRICT kernel's operand window no instruction of these sequences produces a
// denormal or overflows, and every step (v_rcp_f32 included) scales exactly with the operands' exponents, so the significand
// pairs cover the whole window; signs are symmetric; n = +0 gives +0 in every form.
//
//   G7: r=rcp(d); e=fma(-d,r,1); r=fma(e,r,r); q=n*r; e=fma(-d,q,n); q=fma(e,r,q); e=fma(-d,q,n); q=fma(e,r,q)   (the kernel, = hipcc's)
//   G5: r=rcp(d); e=fma(-d,r,1); r=fma(e,r,r); q=n*r; e=fma(-d,q,n); q=fma(e,r,q)
//   H5: r=rcp(d);                              q=n*r; e=fma(-d,q,n); q=fma(e,r,q); e=fma(-d,q,n); q=fma(e,r,q)   (seed not refined)
//   H3: r=rcp(d);                              q=n*r; e=fma(-d,q,n); q=fma(e,r,q)
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/strict_divide_exhaustive.hip -o tools/strict_divide_exhaustive
// Run  : ./tools/strict_divide_exhaustive [first_slab [slabs]]      (1024 slabs of 2^13 numerator significands each = everything)
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

struct Report {
    unsigned long long bad[4];   // G7, G5, H5, H3
    uint32_t           n[4][4], d[4][4], got[4][4], want[4][4];
};

__device__ __forceinline__ void note(Report* rep, int which, float n, float d, float got, float want) {
    const unsigned long long k = atomicAdd(&rep->bad[which], 1ull);
    if (k < 4) rep->n[which][k] = __float_as_uint(n), rep->d[which][k] = __float_as_uint(d), rep->got[which][k] = __float_as_uint(got), rep->want[which][k] = __float_as_uint(want);
}

// one thread per denominator significand; numerator significands [n_first, n_first + n_count)
__global__ __launch_bounds__(256) void sweep(Report* rep, uint32_t n_first, uint32_t n_count) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^23-1
    const float    d = __uint_as_float(0x3f800000u | g);
    const float    r0 = __builtin_amdgcn_rcpf(d);
    const float    e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float    r1 = __builtin_fmaf(e0, r0, r0);
    unsigned       bad7 = 0, bad5 = 0, badh5 = 0, badh3 = 0;
    float          n7 = 0, n5 = 0, nh5 = 0, nh3 = 0;
    for (uint32_t k = 0; k < n_count; ++k) {
        const float n    = __uint_as_float(0x3f800000u | (n_first + k));
        const float want = n / d;
        // refined seed
        float q = n * r1;
        q       = __builtin_fmaf(__builtin_fmaf(-d, q, n), r1, q);
        const float g5 = q;
        q              = __builtin_fmaf(__builtin_fmaf(-d, q, n), r1, q);
        const float g7 = q;
        // raw seed
        float p = n * r0;
        p       = __builtin_fmaf(__builtin_fmaf(-d, p, n), r0, p);
        const float h3 = p;
        p              = __builtin_fmaf(__builtin_fmaf(-d, p, n), r0, p);
        const float h5 = p;
        if (__float_as_uint(g7) != __float_as_uint(want)) { if (!bad7) n7 = n; ++bad7; }
        if (__float_as_uint(g5) != __float_as_uint(want)) { if (!bad5) n5 = n; ++bad5; }
        if (__float_as_uint(h5) != __float_as_uint(want)) { if (!badh5) nh5 = n; ++badh5; }
        if (__float_as_uint(h3) != __float_as_uint(want)) { if (!badh3) nh3 = n; ++badh3; }
    }
    if (bad7) { note(rep, 0, n7, d, 0, n7 / d); atomicAdd(&rep->bad[0], static_cast<unsigned long long>(bad7 - 1)); }
    if (bad5) { note(rep, 1, n5, d, 0, n5 / d); atomicAdd(&rep->bad[1], static_cast<unsigned long long>(bad5 - 1)); }
    if (badh5) { note(rep, 2, nh5, d, 0, nh5 / d); atomicAdd(&rep->bad[2], static_cast<unsigned long long>(badh5 - 1)); }
    if (badh3) { note(rep, 3, nh3, d, 0, nh3 / d); atomicAdd(&rep->bad[3], static_cast<unsigned long long>(badh3 - 1)); }
}

// v_rcp_f32 scales exactly with its operand's exponent: rcp(g * 2^k) == rcp(g) * 2^-k for every significand g and the exponents
// of the window (so do mul and fma while nothing leaves the normal range) -- which is what lets significand pairs stand for all operands
__global__ __launch_bounds__(256) void rcp_scaling(unsigned long long* bad) {
    const uint32_t g    = blockIdx.x * blockDim.x + threadIdx.x;
    const float    base = __builtin_amdgcn_rcpf(__uint_as_float(0x3f800000u | g));
    unsigned       n    = 0;
    for (int k = -100; k <= 100; ++k) {
        const float d = __uint_as_float((static_cast<uint32_t>(127 + k) << 23) | g);
        const float r = __builtin_amdgcn_rcpf(d);
        const float want = __uint_as_float(__float_as_uint(base) - (static_cast<uint32_t>(k) << 23));  // base * 2^-k, exactly
        n += __float_as_uint(r) != __float_as_uint(want);
    }
    if (n) atomicAdd(bad, static_cast<unsigned long long>(n));
}

int main(int argc, char** argv) {
    const uint32_t slab = 1u << 13, all_slabs = (1u << 23) / slab;
    const uint32_t first = argc > 1 ? static_cast<uint32_t>(atoi(argv[1])) : 0, count = argc > 2 ? static_cast<uint32_t>(atoi(argv[2])) : all_slabs - first;
    Report* rep;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    {
        unsigned long long* bad = reinterpret_cast<unsigned long long*>(rep);
        hipLaunchKernelGGL(rcp_scaling, dim3((1u << 23) / 256), dim3(256), 0, 0, bad);
        CHECK(hipDeviceSynchronize());
        unsigned long long h = 0;
        CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
        printf("v_rcp_f32(g * 2^k) == v_rcp_f32(g) * 2^-k for 2^23 significands x k = -100..100: mismatches %llu\n", h);
        CHECK(hipMemset(rep, 0, sizeof(Report)));
    }
    for (uint32_t s = first; s < first + count && s < all_slabs; ++s) {
        hipLaunchKernelGGL(sweep, dim3((1u << 23) / 256), dim3(256), 0, 0, rep, s * slab, slab);
        if ((s - first) % 32 == 31 || s + 1 == first + count) {
            CHECK(hipDeviceSynchronize());
            Report r;
            CHECK(hipMemcpy(&r, rep, sizeof(r), hipMemcpyDeviceToHost));
            printf("numerator slabs %u..%u of %u done (%.3e quotients): mismatches G7 %llu  G5 %llu  H5 %llu  H3 %llu\n", first, s, all_slabs,
                   static_cast<double>(s - first + 1) * slab * (1u << 23), r.bad[0], r.bad[1], r.bad[2], r.bad[3]);
            fflush(stdout);
        }
    }
    Report r;
    CHECK(hipMemcpy(&r, rep, sizeof(r), hipMemcpyDeviceToHost));
    const char* names[4] = {"G7", "G5", "H5", "H3"};
    for (int w = 0; w < 4; ++w)
        for (unsigned k = 0; k < 4 && k < r.bad[w]; ++k) printf("  %s first mismatches: n=%08x d=%08x want=%08x\n", names[w], r.n[w][k], r.d[w][k], r.want[w][k]);
    return 0;
}
