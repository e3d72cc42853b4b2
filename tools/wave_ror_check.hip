// does DPP wave_ror:1 / wave_rol:1 rotate all 64 lanes on gfx950?  prints the lane each lane reads from.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __global__ void k(int* out) {
    int v = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, false);
}
int main() {
    int* d; hipMalloc(&d, 64 * 4); int h[64];
    hipLaunchKernelGGL(k<0x13C>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("wave_ror:1:"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
    hipLaunchKernelGGL(k<0x134>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("wave_rol:1:"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
    hipLaunchKernelGGL(k<0x121>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("row_ror:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
    return 0;
}
