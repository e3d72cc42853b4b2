// tools/valu_microbench.hip -- measures gfx950 VALU issue rates that bound the N-body inner loop:
// v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_rsq_f32, and the scalar / packed interaction mixes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o tools/valu_microbench
// Output: per test, wave-instructions per cycle per SIMD (at the measured clock) and lane-ops/s chip-wide.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e = (x);                                                         \
        if (e != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr int ITERS = 4096;

// 16 independent accumulators, 16 instructions per asm block
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND> __global__ __launch_bounds__(256) void bench(float* out, float seed) {
    float  a[16];
    float2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + threadIdx.x * 1e-3f + i;
        p[i] = make_float2(a[i], a[i] + 0.5f);
    }
    double dd[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dd[i] = seed + i + threadIdx.x * 1e-3;
    double db = seed * 0.999, dc = seed * 1e-3;
    float  b = seed * 0.999f, c = seed * 1e-3f;
    float2 pb = make_float2(b, b), pc = make_float2(c, c);
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (KIND == 0) {  // v_fma_f32
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 1) {  // v_pk_fma_f32
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 2) {  // v_pk_mul_f32
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 3) {  // v_pk_add_f32
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 4) {  // v_rsq_f32
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 5) {  // v_mul_f32
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 6) {  // v_sub_f32 with SGPR operand
#define OP(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 7) {  // scalar interaction mix: 12 x (fma/mul/sub) + 1 rsq  (x16 lanes of ILP)
#define OP(i)                                                              \
    asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t" \
                 "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t" \
                 "v_rsq_f32 %0, %0\n\t"                                                                  \
                 "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t" \
                 "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"     \
                 : "+v"(a[i])                                                                            \
                 : "v"(b), "v"(c));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 8) {  // v_fma_f32 + v_rsq interleaved 1:1 on different regs
#define OP(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_rsq_f32 %1, %1" : "+v"(a[i]), "+v"(p[i].x) : "v"(b), "v"(c));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 9) {  // packed interaction mix for 2 bodies: 6 pk + 2 rsq + 5 pk... = 11 pk + 2 rsq
#define OP(i)                                                                                                       \
    asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"    \
                 "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"    \
                 "v_rsq_f32 %1, %1\n\tv_rsq_f32 %1, %1\n\t"                                                         \
                 "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\t"    \
                 "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3"        \
                 : "+v"(p[i]), "+v"(a[i])                                                                           \
                 : "v"(pb), "v"(pc));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 10) {  // v_fma_f32, single source register
#define OP(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 11) {  // v_pk_fma_f32, single source register pair
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 12) {  // v_pk_fma_f32 d, x, x, d  (two distinct sources, like d2 = fma(dx,dx,d2))
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(pb));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 13) {  // v_pk_mul_f32 d, x, x
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 14) {  // v_pk_add_f32 with broadcast + neg, as the kernel's dx = bj.x - px
#define OP(i) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pb));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 15) {  // v_fmac_f32 (VOP2)
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 16) {  // v_pk_fma_f32 acc += dx * s   (three distinct sources)
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 17) {  // v_add_f32 a, a, a
#define OP(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 18) {  // v_mov_b32
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 19) {  // v_fma_f64
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[i]) : "v"(db), "v"(dc));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 20) {  // v_mul_f64
#define OP(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(db));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 21) {  // v_add_f64
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dc));
            REP16(OP)
#undef OP
        } else if constexpr (KIND == 22) {  // v_rsq_f64
#define OP(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(dd[i]));
            REP16(OP)
#undef OP
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)dd[i];
    if (s == 123.456f) out[0] = s;
}

struct Test {
    const char* name;
    void (*kernel)(float*, float);
    double wave_instr_per_iter;   // wave-instructions per loop iteration per wave
    double lane_flops_per_instr;  // for info
};

int main(int argc, char** argv) {
    int waves_per_simd_list[] = {2, 8};
    float* out;
    CHECK(hipMalloc(&out, 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    Test tests[] = {
        {"v_fma_f32", bench<0>, 16, 2},        {"v_pk_fma_f32", bench<1>, 16, 4},      {"v_pk_mul_f32", bench<2>, 16, 2},
        {"v_pk_add_f32", bench<3>, 16, 2},     {"v_rsq_f32", bench<4>, 16, 1},         {"v_mul_f32", bench<5>, 16, 1},
        {"v_sub_f32(sgpr)", bench<6>, 16, 1},  {"mix 12fma+1rsq", bench<7>, 16 * 13, 0}, {"fma+rsq 1:1", bench<8>, 32, 0},
        {"pkmix 12pk+2rsq", bench<9>, 16 * 14, 0},
        {"v_fma_f32 1src", bench<10>, 16, 2},  {"v_pk_fma 1src", bench<11>, 16, 4},    {"v_pk_fma d,x,x,d", bench<12>, 16, 4},
        {"v_pk_mul d,x,x", bench<13>, 16, 2},  {"v_pk_add bcast+neg", bench<14>, 16, 2}, {"v_fmac_f32 vop2", bench<15>, 16, 2},
        {"v_pk_fma a+=x*s", bench<16>, 16, 4}, {"v_add_f32 1src", bench<17>, 16, 1},    {"v_mov_b32", bench<18>, 16, 0},
        {"v_fma_f64", bench<19>, 16, 2},       {"v_mul_f64", bench<20>, 16, 1},         {"v_add_f64", bench<21>, 16, 1},
        {"v_rsq_f64", bench<22>, 16, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto& t : tests) {
        for (int w : waves_per_simd_list) {
            const int blocks = cus * w;  // 256 threads = 4 waves = 1 wave per SIMD per block
            hipLaunchKernelGGL(t.kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(t.kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 3;
            const double wave_instr_per_simd = t.wave_instr_per_iter * ITERS * w;  // per SIMD
            const double cycles_24           = ms * 1e-3 * 2.4e9;
            printf("%-18s waves/SIMD=%d  %.3f ms  %.3f cycles/wave-instr/SIMD @2.4GHz  (%.2f T wave-lane-instr/s chip)\n", t.name, w, ms, cycles_24 / wave_instr_per_simd,
                   wave_instr_per_simd * 64 * cus * 4 / (ms * 1e-3) * 1e-12);
        }
    }
    return 0;
}
