// tools/grid_barrier_probe.hip -- what would ONE persistent launch that advances K steps cost per step in synchronisation?
// Every step of the all-pairs integrator needs all new positions, i.e. a grid-wide barrier per step.  This measures a
// monotonic-counter grid barrier (one agent-scope atomic add per workgroup, relaxed polling with s_sleep, release fence before
// the arrive, acquire fence after; MI355X_MICROARCH.md "barrier-counter") for 32 ... 256 single-workgroup-per-CU grids,
// with a 16 KiB position array (1 024 bodies) re-read by every workgroup after each barrier, against the alternative the
// library uses today: K dependent kernel launches (captured in a hipGraph or not).
// Build: hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe
#include <hip/hip_runtime.h>

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

__global__ __launch_bounds__(256) void persistent(unsigned* counter, float4* pos_a, float4* pos_b, int steps, unsigned n) {
    float4* from = pos_a;
    float4* to   = pos_b;
    for (int s = 0; s < steps; ++s) {
        // stand-in for a step: every workgroup reads all positions, writes its own slice
        float acc = 0;
        for (unsigned j = threadIdx.x; j < n; j += blockDim.x) acc += __builtin_nontemporal_load(&from[j].x);
        const unsigned per = n / gridDim.x;
        if (threadIdx.x < per) to[blockIdx.x * per + threadIdx.x] = make_float4(acc, 1, 2, 1);
        // grid barrier
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = static_cast<unsigned>(s + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        float4* t = from;
        from = to, to = t;
    }
}

__global__ __launch_bounds__(256) void one_step(float4* from, float4* to, unsigned n) {
    float acc = 0;
    for (unsigned j = threadIdx.x; j < n; j += blockDim.x) acc += from[j].x;
    const unsigned per = n / gridDim.x;
    if (threadIdx.x < per) to[blockIdx.x * per + threadIdx.x] = make_float4(acc, 1, 2, 1);
}

int main() {
    const unsigned n = 1024;
    const int      K = 2000;
    unsigned*      counter;
    float4 *       a, *b;
    CHECK(hipMalloc(&counter, 4));
    CHECK(hipMalloc(&a, n * sizeof(float4)));
    CHECK(hipMalloc(&b, n * sizeof(float4)));
    CHECK(hipMemset(a, 0, n * sizeof(float4)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int grid : {32, 64, 128, 256}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(counter, 0, 4));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(persistent, dim3(grid), dim3(256), 0, 0, counter, a, b, K, n);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("persistent launch, %3d workgroups x 256 threads, grid barrier per step : %.2f us/step\n", grid, best * 1e3f / K);
        best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int s = 0; s < K; ++s) hipLaunchKernelGGL(one_step, dim3(grid), dim3(256), 0, 0, (s & 1) ? b : a, (s & 1) ? a : b, n);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("one launch per step,  %3d workgroups x 256 threads                      : %.2f us/step\n", grid, best * 1e3f / K);
    }
    return 0;
}
