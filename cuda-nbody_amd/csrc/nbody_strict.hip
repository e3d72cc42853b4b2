// nbody_strict.hip -- the bit-reproducing kernels (NB_MODE_STRICT).  gfx950 only.
//
// MUST be compiled with -ffp-contract=off (csrc/Makefile does): every arithmetic op below has to stay
// the separate IEEE-754 mul/add/sub/div/sqrt the reference's CPU path executes
// (/root/reference/src/nbody/bodysystemcpu.cpp:140-303).  hipcc's defaults supply the rest:
// correctly rounded fp32 divide/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt is on by default),
// fp32 denormals kept, IEEE mode on.
//
// Mapping: one lane = one body i (bodysystemcuda.cu:151 uses the same mapping); the workgroup streams the
// j bodies through an LDS tile of `blockDim.x` bodies in ascending j, so lane i sees j = j_begin ..
// j_begin+j_count-1 in exactly the CPU path's order (bodysystemcpu.cpp:156 / :251).  No early return
// before a barrier (the reference kernel has one, bodysystemcuda.cu:153-155): out-of-range lanes clamp
// their load index and skip their stores.
#include "nbody_kernels.h"

namespace nb {
namespace {

template <typename T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

// sqrtf/sqrt lower to llvm.sqrt and `/` to fdiv, both expanded correctly rounded under hipcc's defaults.
// (NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that is the 1-ulp native v_sqrt_f32.)
__device__ __forceinline__ float  sqrt_T(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_T(double x) { return sqrt(x); }
__device__ __forceinline__ float  div_T(float a, float b) { return a / b; }
__device__ __forceinline__ double div_T(double a, double b) { return a / b; }

// r2 as the CPU path forms it:
//   fp32  bodysystemcpu.cpp:186-188   ((eps2 + dx2) + dy2) + dz2
//   fp64  bodysystemcpu.cpp:262-266   (dx2 + dy2) + (dz2 + eps2)
__device__ __forceinline__ float  r2_T(float dx2, float dy2, float dz2, float eps2) { return ((eps2 + dx2) + dy2) + dz2; }
__device__ __forceinline__ double r2_T(double dx2, double dy2, double dz2, double eps2) { return (dx2 + dy2) + (dz2 + eps2); }

template <typename T> __global__ void integrate_bodies_strict(Shard<T> s) {
    using vec4 = typename V4<T>::type;
    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
    vec4* tile = reinterpret_cast<vec4*>(smem_raw);

    const vec4* __restrict__ old_pos = reinterpret_cast<const vec4*>(s.old_pos);
    const unsigned p       = blockDim.x;
    const unsigned local   = blockIdx.x * p + threadIdx.x;
    const bool     active  = local < s.i_count;
    const unsigned i       = s.i_begin + (active ? local : s.i_count - 1);

    const vec4 pi = old_pos[i];
    T          ax = 0, ay = 0, az = 0;
    if (s.acc_in) {
        const vec4 a = reinterpret_cast<const vec4*>(s.acc)[i];
        ax = a.x, ay = a.y, az = a.z;
    }
    const T eps2 = s.eps2;

    for (unsigned base = 0; base < s.j_count; base += p) {
        const unsigned cnt = min(p, s.j_count - base);
        if (threadIdx.x < cnt) tile[threadIdx.x] = old_pos[s.j_begin + base + threadIdx.x];
        __syncthreads();
#pragma unroll 4
        for (unsigned k = 0; k < cnt; ++k) {
            const vec4 bj  = tile[k];
            const T    dx  = bj.x - pi.x;
            const T    dy  = bj.y - pi.y;
            const T    dz  = bj.z - pi.z;
            const T    dx2 = dx * dx;
            const T    dy2 = dy * dy;
            const T    dz2 = dz * dz;
            const T    r2  = r2_T(dx2, dy2, dz2, eps2);
            const T    r   = sqrt_T(r2);
            const T    mr4 = div_T(bj.w, r2 * r2);
            const T    mr3 = mr4 * r;
            ax             = ax + mr3 * dx;  // contraction is off: mul, then add (bodysystemcpu.cpp:200-210 / :278-280)
            ay             = ay + mr3 * dy;
            az             = az + mr3 * dz;
        }
        __syncthreads();
    }

    if (!active) return;  // after the last barrier

    if (s.finalize) {
        // bodysystemcpu.cpp:228-234 (fp32) / :283-298 (fp64): dv = acc*dt; v = (v + dv)*damping; p += v*dt
        vec4 v  = reinterpret_cast<const vec4*>(s.vel)[i];
        vec4 pn = pi;
        const T dvx = ax * s.dt, dvy = ay * s.dt, dvz = az * s.dt;
        v.x = (v.x + dvx) * s.damping;
        v.y = (v.y + dvy) * s.damping;
        v.z = (v.z + dvz) * s.damping;
        pn.x = pn.x + v.x * s.dt;
        pn.y = pn.y + v.y * s.dt;
        pn.z = pn.z + v.z * s.dt;
        reinterpret_cast<vec4*>(s.new_pos)[i] = pn;
        reinterpret_cast<vec4*>(s.vel)[i]     = v;
    } else {
        vec4 a;
        a.x = ax, a.y = ay, a.z = az, a.w = 0;
        reinterpret_cast<vec4*>(s.acc)[i] = a;
    }
}

}  // namespace

template <typename T> hipError_t launch_strict(const Shard<T>& s, int block_size, hipStream_t stream) {
    const unsigned p      = static_cast<unsigned>(block_size);
    const unsigned blocks = (s.i_count + p - 1) / p;
    const size_t   smem   = static_cast<size_t>(p) * 4 * sizeof(T);
    hipLaunchKernelGGL(integrate_bodies_strict<T>, dim3(blocks), dim3(p), smem, stream, s);
    return hipGetLastError();
}

template hipError_t launch_strict<float>(const Shard<float>&, int, hipStream_t);
template hipError_t launch_strict<double>(const Shard<double>&, int, hipStream_t);

}  // namespace nb
