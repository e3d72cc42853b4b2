/*
 * nbody_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's CPU BodySystem path, used only as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under cuda-nbody_amd/ may include, link or call this file.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - oracle_randomise_*  : PINNED against the reference's own randomise_bodies.cpp,
 *                           compiled unmodified into oracle/_ref/ (oracle/Makefile, target ref)
 *                           and compared bit-for-bit in tests/test_oracle.py.
 *   - oracle_update_*     : PARITY UNPINNED.  The reference holds no tests, golden vectors or
 *                           data files for this path, and src/nbody/bodysystemcpu.cpp cannot be
 *                           compiled in this image without stand-in headers (<print> is absent)
 *                           and source edits (MSVC-only explicit-instantiation syntax), which the
 *                           build rules forbid.  The restatement below follows the reference
 *                           text line by line; two independently written forms (scalar and AVX)
 *                           are cross-checked bit-for-bit, plus a numpy restatement in tests/.
 *
 * Build: gcc -O2 -mavx -ffp-contract=off (never -mfma / -march=native: contraction changes bits).
 *
 * All reference citations are relative to /root/reference/.
 */
#include <immintrin.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * libc rand() access.  The reference never seeds (no srand anywhere), i.e. glibc seed 1.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_srand(unsigned seed) { srand(seed); }
ORACLE_API int  oracle_rand(void) { return rand(); }
ORACLE_API int  oracle_rand_max(void) { return RAND_MAX; }
ORACLE_API int  oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * BodySystemCPU<float>::update   src/nbody/bodysystemcpu.cpp:149-243
 *
 * Layout here is the interleaved {x,y,z,m} / {vx,vy,vz,_} of the device boundary
 * (src/nbody/bodysystemcuda.cu:148); the reference CPU class keeps SoA copies
 * (bodysystemcpu.hpp:69-72) -- a pure relayout, every arithmetic op below is the same
 * IEEE-754 binary32 op in the same order:
 *   :156      j outer, :168 i inner   => per-i accumulation order j = 0,1,...,N-1
 *   :174-176  d = p_j - p_i
 *   :182-188  r2 = ((eps2 + dx*dx) + dy*dy) + dz*dz
 *   :191      r  = sqrt(r2)
 *   :194      m_r4 = m_j / (r2*r2)
 *   :197      m_r3 = m_r4 * r
 *   :200-210  dv += m_r3 * d           (separate mul, then add)
 *   :228-234  dv *= dt; v = (v + dv) * damping; p += v * dt
 * Velocity .w and position .w (mass) are never written (set_velocity :130-137 ignores .w).
 * The reference requires N % 8 == 0 for this path; the scalar form accepts any N.
 * ---------------------------------------------------------------------------------------- */
static void update_f32_scalar(float* pos, float* vel, size_t n, float eps2, float damping, float dt, float* dv /* 3n scratch */) {
    float* dvx = dv;
    float* dvy = dv + n;
    float* dvz = dv + 2 * n;
    memset(dv, 0, 3 * n * sizeof(float));

    for (size_t j = 0; j < n; ++j) {
        const float xj = pos[4 * j], yj = pos[4 * j + 1], zj = pos[4 * j + 2], mj = pos[4 * j + 3];
        for (size_t i = 0; i < n; ++i) {
            const float dx  = xj - pos[4 * i];
            const float dy  = yj - pos[4 * i + 1];
            const float dz  = zj - pos[4 * i + 2];
            const float dx2 = dx * dx;
            const float dy2 = dy * dy;
            const float dz2 = dz * dz;
            float       r2  = eps2 + dx2;
            r2              = r2 + dy2;
            r2              = r2 + dz2;
            const float r    = sqrtf(r2);
            const float m_r4 = mj / (r2 * r2);
            const float m_r3 = m_r4 * r;
            const float fx   = m_r3 * dx;
            const float fy   = m_r3 * dy;
            const float fz   = m_r3 * dz;
            dvx[i]           = dvx[i] + fx;
            dvy[i]           = dvy[i] + fy;
            dvz[i]           = dvz[i] + fz;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        for (int d = 0; d < 3; ++d) {
            float a = dv[(size_t)d * n + i];
            a       = a * dt;
            float v = vel[4 * i + d] + a;
            v       = v * damping;
            const float dp = v * dt;
            vel[4 * i + d] = v;
            pos[4 * i + d] = pos[4 * i + d] + dp;
        }
    }
}

/* The same path written with the reference's own instruction selection (AVX-256, 8 bodies i per
 * op, SoA staging) -- used (a) to cross-check the scalar form bit-for-bit and (b) as the timed
 * CPU baseline, because this is how the reference's CPU path actually executes.
 * With -fopenmp the parallel-for sits INSIDE the j loop exactly as at bodysystemcpu.cpp:167. */
static void update_f32_avx(float* pos, float* vel, size_t n, float eps2, float damping, float dt, float* soa /* 7n scratch, 32B aligned */) {
    float *px = soa, *py = soa + n, *pz = soa + 2 * n, *pm = soa + 3 * n;
    float *ax = soa + 4 * n, *ay = soa + 5 * n, *az = soa + 6 * n;
    for (size_t i = 0; i < n; ++i) {
        px[i] = pos[4 * i];
        py[i] = pos[4 * i + 1];
        pz[i] = pos[4 * i + 2];
        pm[i] = pos[4 * i + 3];
        ax[i] = ay[i] = az[i] = 0.0f;
    }
    const __m256 soft8 = _mm256_set1_ps(eps2);
    for (size_t j = 0; j < n; ++j) {
        const __m256 xj = _mm256_set1_ps(px[j]);
        const __m256 yj = _mm256_set1_ps(py[j]);
        const __m256 zj = _mm256_set1_ps(pz[j]);
        const __m256 mj = _mm256_set1_ps(pm[j]);
#ifdef _OPENMP
#pragma omp parallel for
#endif
        for (size_t i = 0; i < n; i += 8) {
            const __m256 dx  = _mm256_sub_ps(xj, _mm256_loadu_ps(px + i));
            const __m256 dy  = _mm256_sub_ps(yj, _mm256_loadu_ps(py + i));
            const __m256 dz  = _mm256_sub_ps(zj, _mm256_loadu_ps(pz + i));
            const __m256 dx2 = _mm256_mul_ps(dx, dx);
            const __m256 dy2 = _mm256_mul_ps(dy, dy);
            const __m256 dz2 = _mm256_mul_ps(dz, dz);
            __m256       r2  = _mm256_add_ps(soft8, dx2);
            r2               = _mm256_add_ps(r2, dy2);
            r2               = _mm256_add_ps(r2, dz2);
            const __m256 r    = _mm256_sqrt_ps(r2);
            const __m256 m_r4 = _mm256_div_ps(mj, _mm256_mul_ps(r2, r2));
            const __m256 m_r3 = _mm256_mul_ps(m_r4, r);
            _mm256_storeu_ps(ax + i, _mm256_add_ps(_mm256_loadu_ps(ax + i), _mm256_mul_ps(m_r3, dx)));
            _mm256_storeu_ps(ay + i, _mm256_add_ps(_mm256_loadu_ps(ay + i), _mm256_mul_ps(m_r3, dy)));
            _mm256_storeu_ps(az + i, _mm256_add_ps(_mm256_loadu_ps(az + i), _mm256_mul_ps(m_r3, dz)));
        }
    }
    for (size_t i = 0; i < n; ++i) {
        const float* a[3] = {ax, ay, az};
        for (int d = 0; d < 3; ++d) {
            float dvd = a[d][i] * dt;
            float v   = vel[4 * i + d] + dvd;
            v         = v * damping;
            const float dp = v * dt;
            vel[4 * i + d] = v;
            pos[4 * i + d] = pos[4 * i + d] + dp;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * BodySystemCPU<double>::update   src/nbody/bodysystemcpu.cpp:245-299
 *   :246 i outer (omp parallel for), :251 j inner, scalar
 *   :253      d = p_j - p_i
 *   :258-266  r2 = (dx*dx + dy*dy) + (dz*dz + eps2)
 *   :269-275  r = sqrt(r2); m_r4 = m_j/(r2*r2); s = m_r4*r
 *   :278-280  acc += d * s
 *   :283-285  dv = acc * dt
 *   :291-298  v = (v + dv) * damping;  p += v * dt    (after ALL dv are known)
 * ---------------------------------------------------------------------------------------- */
static void update_f64(double* pos, double* vel, size_t n, double eps2, double damping, double dt, double* dv /* 3n scratch */) {
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (long i = 0; i < (long)n; ++i) {
        double       acc0 = 0, acc1 = 0, acc2 = 0;
        const double xi = pos[4 * i], yi = pos[4 * i + 1], zi = pos[4 * i + 2];
        for (size_t j = 0; j < n; ++j) {
            const double dx        = pos[4 * j] - xi;
            const double dy        = pos[4 * j + 1] - yi;
            const double dz        = pos[4 * j + 2] - zi;
            const double dx2       = dx * dx;
            const double dy2       = dy * dy;
            const double dz2       = dz * dz;
            const double dx2_dy2   = dx2 + dy2;
            const double dz2_soft2 = dz2 + eps2;
            const double r2        = dx2_dy2 + dz2_soft2;
            const double r         = sqrt(r2);
            const double m_r4      = pos[4 * j + 3] / (r2 * r2);
            const double s         = m_r4 * r;
            acc0                   = acc0 + dx * s;
            acc1                   = acc1 + dy * s;
            acc2                   = acc2 + dz * s;
        }
        dv[3 * i]     = acc0 * dt;
        dv[3 * i + 1] = acc1 * dt;
        dv[3 * i + 2] = acc2 * dt;
    }
    for (size_t i = 0; i < n; ++i) {
        for (int d = 0; d < 3; ++d) {
            const double v = (vel[4 * i + d] + dv[3 * i + d]) * damping;
            vel[4 * i + d] = v;
            pos[4 * i + d] = pos[4 * i + d] + v * dt;
        }
    }
}

/* steps x update(dt).  softening_sq is what BodySystemCPU's ctor computes (bodysystemcpu.cpp:100):
 * T(softening) * softening -- callers pass that product (see oracle_softening_sq_*). */
ORACLE_API int oracle_update_f32(float* pos, float* vel, size_t n, float softening_sq, float damping, float dt, int steps) {
    float* scratch = (float*)malloc(3 * n * sizeof(float) + 64);
    if (!scratch) return -1;
    for (int s = 0; s < steps; ++s) update_f32_scalar(pos, vel, n, softening_sq, damping, dt, scratch);
    free(scratch);
    return 0;
}

ORACLE_API int oracle_update_f32_avx(float* pos, float* vel, size_t n, float softening_sq, float damping, float dt, int steps) {
    if (n % 8) return -2; /* the reference's fp32 path has the same restriction (bodysystemcpu.cpp:168) */
    float* scratch = (float*)aligned_alloc(32, ((7 * n * sizeof(float) + 31) / 32) * 32);
    if (!scratch) return -1;
    for (int s = 0; s < steps; ++s) update_f32_avx(pos, vel, n, softening_sq, damping, dt, scratch);
    free(scratch);
    return 0;
}

ORACLE_API int oracle_update_f64(double* pos, double* vel, size_t n, double softening_sq, double damping, double dt, int steps) {
    double* scratch = (double*)malloc(3 * n * sizeof(double) + 64);
    if (!scratch) return -1;
    for (int s = 0; s < steps; ++s) update_f64(pos, vel, n, softening_sq, damping, dt, scratch);
    free(scratch);
    return 0;
}

/* softening^2 exactly as the constructors form it:
 * bodysystemcpu.cpp:100 `static_cast<T>(params.softening) * params.softening`,
 * bodysystemcuda.cpp:42-45 `softening * softening` with softening = T(params.softening). */
ORACLE_API float  oracle_softening_sq_f32(float softening) { return softening * softening; }
ORACLE_API double oracle_softening_sq_f64(float softening) { return (double)softening * softening; }

/* Timed run for the cpu_baseline leg: ComputeCPU::run_benchmark (compute_cpu.cpp:72-80):
 * steady clock, no warm-up, K x update(dt).  Returns milliseconds (<0 on error). */
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
ORACLE_API double oracle_benchmark_f32(float* pos, float* vel, size_t n, float softening_sq, float damping, float dt, int steps) {
    const double t0 = now_ms();
    if (oracle_update_f32_avx(pos, vel, n, softening_sq, damping, dt, steps)) return -1.0;
    return now_ms() - t0;
}
ORACLE_API double oracle_benchmark_f64(double* pos, double* vel, size_t n, double softening_sq, double damping, double dt, int steps) {
    const double t0 = now_ms();
    if (oracle_update_f64(pos, vel, n, softening_sq, damping, dt, steps)) return -1.0;
    return now_ms() - t0;
}

/* update() restricted to bodies i in [i0, i0+ni): their new position/velocity after ONE step, every body j of
 * the full system contributing in the order j = 0..n-1 -- the same arithmetic and order as update_f32_scalar /
 * update_f64 above, so the outputs are bit-identical to the corresponding rows of a full update.  Lets the
 * full-size tests (262 144 / 1 048 576 bodies) check the STRICT kernels bitwise on a sample in seconds.
 * out_pos / out_vel are 4*ni (xyz updated, .w copied). */
ORACLE_API void oracle_update_subset_f32(const float* pos, const float* vel, size_t n, size_t i0, size_t ni, float eps2, float damping, float dt, float* out_pos, float* out_vel) {
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (long k = 0; k < (long)ni; ++k) {
        const size_t i  = i0 + (size_t)k;
        const float  xi = pos[4 * i], yi = pos[4 * i + 1], zi = pos[4 * i + 2];
        float        dv[3] = {0.0f, 0.0f, 0.0f};
        for (size_t j = 0; j < n; ++j) {
            const float dx = pos[4 * j] - xi, dy = pos[4 * j + 1] - yi, dz = pos[4 * j + 2] - zi;
            const float dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
            float       r2  = eps2 + dx2;
            r2              = r2 + dy2;
            r2              = r2 + dz2;
            const float r    = sqrtf(r2);
            const float m_r4 = pos[4 * j + 3] / (r2 * r2);
            const float m_r3 = m_r4 * r;
            dv[0]            = dv[0] + m_r3 * dx;
            dv[1]            = dv[1] + m_r3 * dy;
            dv[2]            = dv[2] + m_r3 * dz;
        }
        for (int d = 0; d < 3; ++d) {
            const float a  = dv[d] * dt;
            float       v  = vel[4 * i + d] + a;
            v              = v * damping;
            const float dp = v * dt;
            out_vel[4 * k + d] = v;
            out_pos[4 * k + d] = pos[4 * i + d] + dp;
        }
        out_vel[4 * k + 3] = vel[4 * i + 3];
        out_pos[4 * k + 3] = pos[4 * i + 3];
    }
}

ORACLE_API void oracle_update_subset_f64(const double* pos, const double* vel, size_t n, size_t i0, size_t ni, double eps2, double damping, double dt, double* out_pos, double* out_vel) {
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (long k = 0; k < (long)ni; ++k) {
        const size_t i  = i0 + (size_t)k;
        const double xi = pos[4 * i], yi = pos[4 * i + 1], zi = pos[4 * i + 2];
        double       acc[3] = {0, 0, 0};
        for (size_t j = 0; j < n; ++j) {
            const double dx = pos[4 * j] - xi, dy = pos[4 * j + 1] - yi, dz = pos[4 * j + 2] - zi;
            const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
            const double r2  = (dx2 + dy2) + (dz2 + eps2);
            const double r   = sqrt(r2);
            const double s   = (pos[4 * j + 3] / (r2 * r2)) * r;
            acc[0] = acc[0] + dx * s, acc[1] = acc[1] + dy * s, acc[2] = acc[2] + dz * s;
        }
        for (int d = 0; d < 3; ++d) {
            const double dvd = acc[d] * dt;
            const double v   = (vel[4 * i + d] + dvd) * damping;
            out_vel[4 * k + d] = v;
            out_pos[4 * k + d] = pos[4 * i + d] + v * dt;
        }
        out_vel[4 * k + 3] = vel[4 * i + 3];
        out_pos[4 * k + 3] = pos[4 * i + 3];
    }
}

/* Bounded sample for bench.py's cpu_baseline leg: the O(N^2) force pass of update() restricted to bodies
 * i in [0, ni) against all n bodies j (same loop nests, same arithmetic as update_f32_avx / update_f64).
 * Returns milliseconds; *checksum keeps the work observable. */
ORACLE_API double oracle_benchmark_partial_f32(const float* pos, size_t n, size_t ni, float eps2, double* checksum) {
    if (ni % 8 || ni > n) return -2.0;
    float* soa = (float*)aligned_alloc(32, ((7 * n * sizeof(float) + 31) / 32) * 32);
    if (!soa) return -1.0;
    float *px = soa, *py = soa + n, *pz = soa + 2 * n, *pm = soa + 3 * n;
    float *ax = soa + 4 * n, *ay = soa + 5 * n, *az = soa + 6 * n;
    for (size_t i = 0; i < n; ++i) {
        px[i] = pos[4 * i], py[i] = pos[4 * i + 1], pz[i] = pos[4 * i + 2], pm[i] = pos[4 * i + 3];
        ax[i] = ay[i] = az[i] = 0.0f;
    }
    const double t0    = now_ms();
    const __m256 soft8 = _mm256_set1_ps(eps2);
    for (size_t j = 0; j < n; ++j) {
        const __m256 xj = _mm256_set1_ps(px[j]);
        const __m256 yj = _mm256_set1_ps(py[j]);
        const __m256 zj = _mm256_set1_ps(pz[j]);
        const __m256 mj = _mm256_set1_ps(pm[j]);
#ifdef _OPENMP
#pragma omp parallel for
#endif
        for (size_t i = 0; i < ni; i += 8) {
            const __m256 dx  = _mm256_sub_ps(xj, _mm256_loadu_ps(px + i));
            const __m256 dy  = _mm256_sub_ps(yj, _mm256_loadu_ps(py + i));
            const __m256 dz  = _mm256_sub_ps(zj, _mm256_loadu_ps(pz + i));
            __m256       r2  = _mm256_add_ps(soft8, _mm256_mul_ps(dx, dx));
            r2               = _mm256_add_ps(r2, _mm256_mul_ps(dy, dy));
            r2               = _mm256_add_ps(r2, _mm256_mul_ps(dz, dz));
            const __m256 r    = _mm256_sqrt_ps(r2);
            const __m256 m_r4 = _mm256_div_ps(mj, _mm256_mul_ps(r2, r2));
            const __m256 m_r3 = _mm256_mul_ps(m_r4, r);
            _mm256_storeu_ps(ax + i, _mm256_add_ps(_mm256_loadu_ps(ax + i), _mm256_mul_ps(m_r3, dx)));
            _mm256_storeu_ps(ay + i, _mm256_add_ps(_mm256_loadu_ps(ay + i), _mm256_mul_ps(m_r3, dy)));
            _mm256_storeu_ps(az + i, _mm256_add_ps(_mm256_loadu_ps(az + i), _mm256_mul_ps(m_r3, dz)));
        }
    }
    const double ms = now_ms() - t0;
    double       c  = 0;
    for (size_t i = 0; i < ni; ++i) c += (double)ax[i] + ay[i] + az[i];
    if (checksum) *checksum = c;
    free(soa);
    return ms;
}

ORACLE_API double oracle_benchmark_partial_f64(const double* pos, size_t n, size_t ni, double eps2, double* checksum) {
    if (ni > n) return -2.0;
    double* acc = (double*)malloc(3 * ni * sizeof(double) + 64);
    if (!acc) return -1.0;
    const double t0 = now_ms();
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (long i = 0; i < (long)ni; ++i) {
        double       acc0 = 0, acc1 = 0, acc2 = 0;
        const double xi = pos[4 * i], yi = pos[4 * i + 1], zi = pos[4 * i + 2];
        for (size_t j = 0; j < n; ++j) {
            const double dx = pos[4 * j] - xi, dy = pos[4 * j + 1] - yi, dz = pos[4 * j + 2] - zi;
            const double r2 = (dx * dx + dy * dy) + (dz * dz + eps2);
            const double r  = sqrt(r2);
            const double s  = (pos[4 * j + 3] / (r2 * r2)) * r;
            acc0 += dx * s, acc1 += dy * s, acc2 += dz * s;
        }
        acc[3 * i] = acc0, acc[3 * i + 1] = acc1, acc[3 * i + 2] = acc2;
    }
    const double ms = now_ms() - t0;
    double       c  = 0;
    for (size_t i = 0; i < 3 * ni; ++i) c += acc[i];
    if (checksum) *checksum = c;
    free(acc);
    return ms;
}

/* One fp64 evaluation of the accelerations of bodies [i0, i0+ni) from fp32 or fp64 positions, in the
 * plain textbook form a_i = sum_j m_j d / (|d|^2+eps2)^(3/2).  Not a reference function: it is the
 * high-precision yardstick for the fast kernels' per-step force error (tests/test_gpu_parity.py). */
ORACLE_API void oracle_accel_f64_from_f32(const float* pos, size_t n, size_t i0, size_t ni, double eps2, double* acc /* 3*ni */) {
#ifdef _OPENMP
#pragma omp parallel for
#endif
    for (long k = 0; k < (long)ni; ++k) {
        const size_t i  = i0 + (size_t)k;
        const double xi = pos[4 * i], yi = pos[4 * i + 1], zi = pos[4 * i + 2];
        double       a0 = 0, a1 = 0, a2 = 0;
        for (size_t j = 0; j < n; ++j) {
            const double dx = (double)pos[4 * j] - xi, dy = (double)pos[4 * j + 1] - yi, dz = (double)pos[4 * j + 2] - zi;
            const double r2 = dx * dx + dy * dy + dz * dz + eps2;
            const double s  = (double)pos[4 * j + 3] / (r2 * sqrt(r2));
            a0 += dx * s;
            a1 += dy * s;
            a2 += dz * s;
        }
        acc[3 * k] = a0, acc[3 * k + 1] = a1, acc[3 * k + 2] = a2;
    }
}

/* ------------------------------------------------------------------------------------------
 * randomise_bodies<T>(config, span pos, span vel, clusterScale, velocityScale)
 *   src/nbody/randomise_bodies.cpp:47-189   (the SoA overload :191-319 draws identically)
 *   rng   :37-39  rand() / T(RAND_MAX)
 *   rng_2 :41-43  rand() * (T(2)/T(RAND_MAX)) - T(1)
 *   normalize :14-24, cross :29-35, dot :25-27
 * Type notes kept from the reference: `scale`/`vscale` in SHELL are `float` even for T=double
 * (auto from float operands, :103-104) while inner/outer are T (:105-106).
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_RANDOMISE(T, SUFFIX, SQRT)                                                                                \
    static T rng_##SUFFIX(void) { return rand() / (T)RAND_MAX; }                                                         \
    static T rng2_##SUFFIX(void) { return rand() * ((T)2.0f / (T)RAND_MAX) - (T)1.0f; }                                 \
    static T normalize_##SUFFIX(T* v) {                                                                                  \
        const T dist = SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);                                                    \
        if (dist > 1e-6) {                                                                                               \
            v[0] /= dist;                                                                                                \
            v[1] /= dist;                                                                                                \
            v[2] /= dist;                                                                                                \
        }                                                                                                                \
        return dist;                                                                                                     \
    }                                                                                                                    \
    ORACLE_API void oracle_randomise_##SUFFIX(int config, T* pos, T* vel, size_t nb_bodies, float clusterScale, float velocityScale) { \
        size_t p = 0, v = 0, i = 0;                                                                                      \
        if (config == 1) { /* NBODY_CONFIG_SHELL :101-147 */                                                             \
            const float scale  = clusterScale;                                                                           \
            const float vscale = scale * velocityScale;                                                                  \
            const T     inner  = (T)2.5f * scale;                                                                        \
            const T     outer  = (T)4 * scale;                                                                           \
            while (i < nb_bodies) {                                                                                      \
                const T x = rng2_##SUFFIX();                                                                             \
                const T y = rng2_##SUFFIX();                                                                             \
                const T z = rng2_##SUFFIX();                                                                             \
                T       point[3] = {x, y, z};                                                                            \
                const T len      = normalize_##SUFFIX(point);                                                            \
                if (len > 1) continue;                                                                                   \
                pos[p++] = point[0] * (inner + (outer - inner) * rng_##SUFFIX());                                        \
                pos[p++] = point[1] * (inner + (outer - inner) * rng_##SUFFIX());                                        \
                pos[p++] = point[2] * (inner + (outer - inner) * rng_##SUFFIX());                                        \
                pos[p++] = 1.0f;                                                                                         \
                T axis[3] = {0, 0, 1};                                                                                   \
                if (1 - point[2] < 1e-6) {                                                                               \
                    axis[0] = point[1];                                                                                  \
                    axis[1] = point[0];                                                                                  \
                    normalize_##SUFFIX(axis);                                                                            \
                }                                                                                                        \
                const T a[3] = {pos[4 * i], pos[4 * i + 1], pos[4 * i + 2]};                                             \
                const T cx   = a[1] * axis[2] - a[2] * axis[1];                                                          \
                const T cy   = a[2] * axis[0] - a[0] * axis[2];                                                          \
                const T cz   = a[0] * axis[1] - a[1] * axis[0];                                                          \
                vel[v++]     = cx * vscale;                                                                              \
                vel[v++]     = cy * vscale;                                                                              \
                vel[v++]     = cz * vscale;                                                                              \
                vel[v++]     = 0.0f;                                                                                     \
                i++;                                                                                                     \
            }                                                                                                            \
        } else if (config == 2) { /* NBODY_CONFIG_EXPAND :149-187 */                                                     \
            T scale = clusterScale * nb_bodies / (T)1024;                                                                \
            if (scale < 1) scale = clusterScale;                                                                         \
            const T vscale = scale * velocityScale;                                                                      \
            while (i < nb_bodies) {                                                                                      \
                T point[3];                                                                                              \
                point[0]   = rng2_##SUFFIX();                                                                            \
                point[1]   = rng2_##SUFFIX();                                                                            \
                point[2]   = rng2_##SUFFIX();                                                                            \
                const T r2 = point[0] * point[0] + point[1] * point[1] + point[2] * point[2];                            \
                if (r2 > 1) continue;                                                                                    \
                pos[p++] = point[0] * scale;                                                                             \
                pos[p++] = point[1] * scale;                                                                             \
                pos[p++] = point[2] * scale;                                                                             \
                pos[p++] = 1.0f;                                                                                         \
                vel[v++] = point[0] * vscale;                                                                            \
                vel[v++] = point[1] * vscale;                                                                            \
                vel[v++] = point[2] * vscale;                                                                            \
                vel[v++] = 0.0f;                                                                                         \
                ++i;                                                                                                     \
            }                                                                                                            \
        } else { /* NBODY_CONFIG_RANDOM (and default) :56-99 */                                                          \
            const T n_1024 = nb_bodies / (T)1024;                                                                        \
            const T scale  = clusterScale * ((T)1 < n_1024 ? n_1024 : (T)1);                                             \
            const T vscale = velocityScale * scale;                                                                      \
            while (i < nb_bodies) {                                                                                      \
                T point[3], velocity[3];                                                                                 \
                point[0] = rng2_##SUFFIX();                                                                              \
                point[1] = rng2_##SUFFIX();                                                                              \
                point[2] = rng2_##SUFFIX();                                                                              \
                T r2     = point[0] * point[0] + point[1] * point[1] + point[2] * point[2];                              \
                if (r2 > 1) continue;                                                                                    \
                velocity[0] = rng2_##SUFFIX();                                                                           \
                velocity[1] = rng2_##SUFFIX();                                                                           \
                velocity[2] = rng2_##SUFFIX();                                                                           \
                r2          = velocity[0] * velocity[0] + velocity[1] * velocity[1] + velocity[2] * velocity[2];         \
                if (r2 > 1) continue;                                                                                    \
                pos[p++] = point[0] * scale;                                                                             \
                pos[p++] = point[1] * scale;                                                                             \
                pos[p++] = point[2] * scale;                                                                             \
                pos[p++] = 1.0f;                                                                                         \
                vel[v++] = velocity[0] * vscale;                                                                         \
                vel[v++] = velocity[1] * vscale;                                                                         \
                vel[v++] = velocity[2] * vscale;                                                                         \
                vel[v++] = 0.0f;                                                                                         \
                i++;                                                                                                     \
            }                                                                                                            \
        }                                                                                                                \
    }

DEFINE_RANDOMISE(float, f32, sqrtf)
DEFINE_RANDOMISE(double, f64, sqrt)
