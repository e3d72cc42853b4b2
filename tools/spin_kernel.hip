// tools/spin_kernel.hip -- a stand-in for RCCL's all-gather kernel in contention experiments: `blocks` workgroups of
// `threads` lanes that hold their CUs for `micros` microseconds (s_memrealtime runs at 100 MHz).
#include <hip/hip_runtime.h>
__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) *sink = 1;
}
extern "C" __attribute__((visibility("default"))) int spin_launch(int blocks, int threads, int micros, void* stream) {
    static int* sink = nullptr;
    if (!sink) (void)hipMalloc(&sink, 4);
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long>(micros) * 100ull, sink);
    return static_cast<int>(hipGetLastError());
}
