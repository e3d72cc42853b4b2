// nbody_lane.h -- per-precision traits shared by the FAST kernels (nbody_fast.hip, nbody_pair.hip): the "vector" a lane
// computes with.  fp32 bodies travel in packed PAIRS (one pair per 64-bit VGPR pair -> v_pk_add/mul/fma_f32), fp64 has no
// packed form.  Included inside each translation unit's own anonymous namespace (device code, internal linkage).
#pragma once

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- per-precision traits: the "vector" a lane computes with -------------------------------------------
template <typename T> struct Lane;

template <> struct Lane<float> {
    using vec4                 = float4;
    typedef float raw4 __attribute__((ext_vector_type(4)));  // a body as the scalar unit loads it
    using bits                 = unsigned;
    using vec                  = v2f;  // two bodies i per vector -> v_pk_*_f32
    static constexpr int W     = 2;
    static __device__ __forceinline__ vec  splat(float a) { return vec{a, a}; }
    static __device__ __forceinline__ vec  fma(vec a, vec b, vec c) { return __builtin_elementwise_fma(a, b, c); }
    // s = m * d2^(-3/2): 2 x v_rsq_f32 (1 ulp, what the reference's rsqrtf is) + 3 v_pk_mul_f32   (bodysystemcuda.cu:110-115);
    // UNIT: the body's relative mass is 1 -> 2 v_pk_mul_f32
    // `zm` is the register pair {z, m} of the body j as loaded: the mass is broadcast from its HIGH half by op_sel (written
    // as one inline instruction: left to itself hipcc copies the mass into the low half of another pair first)
    struct Consts {};
    static __device__ __forceinline__ Consts make_consts() { return {}; }
    template <bool UNIT> static __device__ __forceinline__ vec coupling(vec zm, vec d2, const Consts&) {
        const vec inv  = vec{__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
        const vec inv2 = inv * inv;
        const vec inv3 = inv * inv2;
        if constexpr (UNIT) return inv3;
        // (the inline instruction must not read a v_rsq result directly: gfx950 needs a wait state between a transcendental
        // and a VALU op that uses its result, and hipcc's hazard pass does not look inside inline asm -- it multiplies inv^3)
        vec s;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(s) : "v"(zm), "v"(inv3));
        return s;
    }
    // the same with the body's relative mass already a vector (wave-stream kernel: body j arrives in scalar registers)
    template <bool UNIT> static __device__ __forceinline__ vec coupling_rel(vec mrel, vec d2, const Consts&) {
        const vec inv  = vec{__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
        const vec inv2 = inv * inv;
        const vec inv3 = inv * inv2;
        if constexpr (UNIT) return inv3; else return mrel * inv3;
    }
    static __device__ __forceinline__ vec z_and_mass(const vec4& b) { return vec{b.z, b.w}; }
    static __device__ __forceinline__ float get(vec a, int w) { return w == 0 ? a.x : a.y; }
    static __device__ __forceinline__ void  set(vec& a, int w, float v) {
        if (w == 0) a.x = v; else a.y = v;
    }
    static __device__ __forceinline__ void  keep_in_vgpr(vec& a) { asm volatile("" : "+v"(a)); }
};

template <> struct Lane<double> {
    using vec4                 = double4;
    typedef double raw4 __attribute__((ext_vector_type(4)));
    using bits                 = unsigned long long;
    using vec                  = double;
    static constexpr int W     = 1;
    static __device__ __forceinline__ vec splat(double a) { return a; }
    static __device__ __forceinline__ vec fma(vec a, vec b, vec c) { return __builtin_fma(a, b, c); }
    // s = m * d2^(-3/2) in full double precision from the v_rsq_f64 seed y0 (relative error <= 2^-23) WITHOUT
    // iterating on y: with r = 1 - d2*y0^2 (|r| <= 2^-22),  d2^(-3/2) = y0^3 (1-r)^(-3/2) = y0^3 (1 + 3/2 r + 15/8 r^2 + O(r^3)),
    // truncation 35/16 r^3 < 2^-64.  7 DP ops + the seed, against 10 for two Newton steps on y followed by the cube
    // (the reference calls CUDA's <= 1 ulp rsqrt(double) here, bodysystemcuda.cu:82-84,110-115).  UNIT: 6 DP ops.
    static __device__ __forceinline__ vec z_and_mass(const vec4& b) { return b.w; }
    // 15/8 and 3/2 are not inline constants of the ISA; left as literals hipcc rebuilds 1.5 in a register pair before
    // every v_fmac_f64 (2 v_mov_b32 per interaction).  Pinned in registers once per kernel they feed a 3-operand v_fma_f64.
    struct Consts {
        double c1875, c15;
    };
    static __device__ __forceinline__ Consts make_consts() {
        Consts c{1.875, 1.5};
        asm volatile("" : "+v"(c.c1875), "+v"(c.c15));
        return c;
    }
    template <bool UNIT> static __device__ __forceinline__ vec coupling(vec m, vec d2, const Consts& k) {
        const double c1875 = k.c1875, c15 = k.c15;
        const double y0 = __builtin_amdgcn_rsq(d2);
        const double t0 = y0 * y0;
        const double r  = __builtin_fma(-d2, t0, 1.0);
        double       mc = y0 * t0;
        if constexpr (!UNIT) mc = m * mc;
        const double w  = r * __builtin_fma(r, c1875, c15);
        return __builtin_fma(mc, w, mc);
    }
    template <bool UNIT> static __device__ __forceinline__ vec coupling_rel(vec mrel, vec d2, const Consts& k) { return coupling<UNIT>(mrel, d2, k); }
    static __device__ __forceinline__ double get(vec a, int) { return a; }
    static __device__ __forceinline__ void   set(vec& a, int, double v) { a = v; }
    static __device__ __forceinline__ void   keep_in_vgpr(vec& a) { asm volatile("" : "+v"(a)); }
};

