// tools/mfma_overlap.hip -- does matrix-pipe work overlap with a packed-fp32 VALU stream on the same SIMD?
// Per loop iteration every wave issues 8 x (12 v_pk_fma_f32 + 2 v_rsq_f32) -- the N-body pair mix -- plus NM MFMAs:
//   KIND 0: none   KIND 1: 4 x v_mfma_f32_32x32x16_bf16   KIND 2: 4 x v_mfma_f32_32x32x2_f32   KIND 3: 4 x v_mfma_f32_16x16x32_bf16
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_overlap.hip -o tools/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITERS = 2048;

template <int KIND> __global__ __launch_bounds__(256) void k(float* out, float seed) {
    v2f p[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (v2f){seed + i + threadIdx.x * 1e-3f, seed * 0.5f + i}, q[i] = p[i] * 0.25f;
    v2f pb = {seed * 0.999f, seed * 0.999f}, pc = {seed * 1e-3f, seed * 1e-3f};
    v16f acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    v4f  s0 = {}, s1 = {}, s2 = {}, s3 = {};
    v8bf a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + i), b[i] = (__bf16)(0.001f * i);
    float af = seed, bf = seed * 0.5f;
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (KIND == 1) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
        } else if constexpr (KIND == 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc3, 0, 0, 0);
        } else if constexpr (KIND == 3) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, s3, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"
                "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"
                "v_rsq_f32 %1, %1\n\t"
                "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"
                "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"
                "v_rsq_f32 %1, %1"
                : "+v"(p[i]), "+v"(q[i].x)
                : "v"(pb), "v"(pc));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y + q[i].x;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    r += s0[0] + s1[1] + s2[2] + s3[3];
    if (r == 123.456f) out[0] = r;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    void (*kernels[])(float*, float) = {k<0>, k<1>, k<2>, k<3>};
    const char* names[] = {"VALU mix only", "+4 mfma_32x32x16_bf16", "+4 mfma_32x32x2_f32", "+4 mfma_16x16x32_bf16"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w : {1, 2, 4}) {
        for (int kind = 0; kind < 4; ++kind) {
            const int blocks = cus * w;
            hipLaunchKernelGGL(kernels[kind], dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kernels[kind], dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 3;
            // cycles per loop iteration per wave-slot on a SIMD (w waves share the SIMD)
            printf("waves/SIMD=%d  %-24s %.3f ms   %.1f cycles@2.3GHz per iteration per wave (VALU mix alone = 8 pairs)\n", w, names[kind], ms, ms * 1e-3 * 2.3e9 / ITERS / w);
        }
    }
    return 0;
}
