// tools/mfma_proto.hip -- prototype: coordinate differences on the MFMA pipe.
// v_mfma_f32_32x32x2_f32 computes D[j][i] = A[j][0]*B[0][i] + A[j][1]*B[1][i]; with A[j] = (x_j, 1), B[.][i] = (1, -x_i)
// that is x_j - x_i, rounded exactly like v_sub_f32.  A wave owns 32 bodies i (lane l and l+32 share i = l%32 and
// split the 32 bodies j of a subtile: 16 each); the VALU keeps only d2 / rsq / coupling / accumulate.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_proto.hip -o tools/mfma_proto
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int BLOCK = 256;
constexpr int TILE  = 1024;
#ifndef WAVES_PER_SIMD
#define WAVES_PER_SIMD 3
#endif

template <int ACCS> __global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void accel_mfma(const float4* __restrict__ pos, float4* __restrict__ acc_out, unsigned n, float eps2s) {
    __shared__ float4 tile[2][TILE];
    const int tid   = threadIdx.x;
    const int wave  = tid >> 6;
    const int lane  = tid & 63;
    const bool upper = lane >= 32;
    const unsigned i = (blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const float4 pi  = pos[i < n ? i : n - 1];
    // B operands: k = 0 row (lanes 0-31) = 1, k = 1 row (lanes 32-63) = -x_i
    const float Bx = upper ? -pi.x : 1.0f, By = upper ? -pi.y : 1.0f, Bz = upper ? -pi.z : 1.0f, Bm = upper ? 0.0f : 1.0f;
    v2f eps2 = {eps2s, eps2s};
    asm volatile("" : "+v"(eps2));
    v2f ax[ACCS], ay[ACCS], az[ACCS];
#pragma unroll
    for (int a = 0; a < ACCS; ++a) ax[a] = ay[a] = az[a] = (v2f){0, 0};

    const unsigned n_tiles = (n + TILE - 1) / TILE;
    float4 regs[TILE / BLOCK];
    auto load_tile = [&](unsigned t) {
#pragma unroll
        for (int r = 0; r < TILE / BLOCK; ++r) {
            const unsigned j = t * TILE + r * BLOCK + tid;
            regs[r] = j < n ? pos[j] : make_float4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < TILE / BLOCK; ++r) tile[buf][r * BLOCK + tid] = regs[r];
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned t = 0; t < n_tiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < n_tiles) load_tile(t + 1);
        const float4* src = &tile[cur][lane & 31];
        // software pipeline: the MFMAs of subtile k+1 are in flight while the VALU consumes subtile k
        auto issue = [&](int sub, v16f& Dx, v16f& Dy, v16f& Dz, v16f& Dm) {
            const float4 e  = src[sub * 32];
            const float  Ax = upper ? 1.0f : e.x, Ay = upper ? 1.0f : e.y, Az = upper ? 1.0f : e.z, Am = e.w;
            Dx = __builtin_amdgcn_mfma_f32_32x32x2f32(Ax, Bx, zero, 0, 0, 0);
            Dy = __builtin_amdgcn_mfma_f32_32x32x2f32(Ay, By, zero, 0, 0, 0);
            Dz = __builtin_amdgcn_mfma_f32_32x32x2f32(Az, Bz, zero, 0, 0, 0);
            Dm = __builtin_amdgcn_mfma_f32_32x32x2f32(Am, Bm, zero, 0, 0, 0);
        };
        auto consume = [&](const v16f& Dx, const v16f& Dy, const v16f& Dz, const v16f& Dm) {
            v2f ux[8], uy[8], uz[8], d2[8], sc[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ux[p] = (v2f){Dx[2 * p], Dx[2 * p + 1]}, uy[p] = (v2f){Dy[2 * p], Dy[2 * p + 1]}, uz[p] = (v2f){Dz[2 * p], Dz[2 * p + 1]};
                d2[p] = __builtin_elementwise_fma(ux[p], ux[p], eps2);
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) d2[p] = __builtin_elementwise_fma(uy[p], uy[p], d2[p]);
#pragma unroll
            for (int p = 0; p < 8; ++p) d2[p] = __builtin_elementwise_fma(uz[p], uz[p], d2[p]);
#pragma unroll
            for (int p = 0; p < 8; ++p) d2[p] = (v2f){__builtin_amdgcn_rsqf(d2[p].x), __builtin_amdgcn_rsqf(d2[p].y)};
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const v2f m = {Dm[2 * p], Dm[2 * p + 1]};
                sc[p]       = m * d2[p];
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) d2[p] = d2[p] * d2[p];
#pragma unroll
            for (int p = 0; p < 8; ++p) sc[p] = sc[p] * d2[p];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ax[p % ACCS] = __builtin_elementwise_fma(ux[p], sc[p], ax[p % ACCS]);
                ay[p % ACCS] = __builtin_elementwise_fma(uy[p], sc[p], ay[p % ACCS]);
                az[p % ACCS] = __builtin_elementwise_fma(uz[p], sc[p], az[p % ACCS]);
            }
        };
        // no software pipelining inside a wave: the OTHER waves of the SIMD cover the MFMA latency
#pragma unroll 1
        for (int sub = 0; sub < TILE / 32; ++sub) {
            v16f Dx, Dy, Dz, Dm;
            issue(sub, Dx, Dy, Dz, Dm);
            consume(Dx, Dy, Dz, Dm);
        }
        if (t + 1 < n_tiles) store_tile(cur ^ 1);
        __syncthreads();
    }
    float fx = 0, fy = 0, fz = 0;
#pragma unroll
    for (int a = 0; a < ACCS; ++a) fx += ax[a].x + ax[a].y, fy += ay[a].x + ay[a].y, fz += az[a].x + az[a].y;
    // wavefront-64 fold: lanes l and l+32 hold the two halves of the j range for the same body i
    fx += __shfl_xor(fx, 32);
    fy += __shfl_xor(fy, 32);
    fz += __shfl_xor(fz, 32);
    if (!upper && i < n) acc_out[i] = make_float4(fx, fy, fz, 0);
}

int main(int argc, char** argv) {
    const unsigned n = argc > 1 ? atoi(argv[1]) : 65536;
    std::vector<float4> h(n);
    srand(1);
    for (auto& p : h) p = make_float4(6.f * rand() / RAND_MAX - 3.f, 6.f * rand() / RAND_MAX - 3.f, 6.f * rand() / RAND_MAX - 3.f, 0.5f + 1.f * rand() / RAND_MAX);
    float4 *d_pos, *d_acc;
    CHECK(hipMalloc(&d_pos, n * 16));
    CHECK(hipMalloc(&d_acc, n * 16));
    CHECK(hipMemcpy(d_pos, h.data(), n * 16, hipMemcpyHostToDevice));
    const float eps2 = 0.01f;
    const dim3 grid((n + 127) / 128);
    hipLaunchKernelGGL(accel_mfma<4>, grid, dim3(BLOCK), 0, 0, d_pos, d_acc, n, eps2);
    CHECK(hipDeviceSynchronize());
    std::vector<float4> a(n);
    CHECK(hipMemcpy(a.data(), d_acc, n * 16, hipMemcpyDeviceToHost));
    double worst = 0;
    for (unsigned k = 0; k < 64; ++k) {
        const unsigned i = (k * 2654435761u) % n;
        double fx = 0, fy = 0, fz = 0;
        for (unsigned j = 0; j < n; ++j) {
            const double dx = (double)h[j].x - h[i].x, dy = (double)h[j].y - h[i].y, dz = (double)h[j].z - h[i].z;
            const double r2 = dx * dx + dy * dy + dz * dz + eps2;
            const double s  = h[j].w / (r2 * sqrt(r2));
            fx += dx * s, fy += dy * s, fz += dz * s;
        }
        const double err = sqrt((a[i].x - fx) * (a[i].x - fx) + (a[i].y - fy) * (a[i].y - fy) + (a[i].z - fz) * (a[i].z - fz)) / sqrt(fx * fx + fy * fy + fz * fz);
        if (err > worst) worst = err;
    }
    printf("n=%u max rel force error vs fp64 (64 samples): %.3e\n", n, worst);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        const int K = 10;
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(accel_mfma<4>, grid, dim3(BLOCK), 0, 0, d_pos, d_acc, n, eps2);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= K;
        printf("mfma-dx kernel: %.3f ms/step, %.1f G interactions/s, %.3f of 157.3 TF\n", ms, (double)n * n / ms * 1e-6, 20.0 * n * n / (ms * 1e-3) / 157.3e12);
    }
    return 0;
}
