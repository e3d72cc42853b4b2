// capi_host.cpp -- extern "C" doorway onto the C++ host mirror so pytest can drive it with ctypes
// (tests/test_host_cpp.py).  Not part of the drop-in boundary (that is include/nbody_hip.h).
#include "compute.hpp"
#include "integrate_nbody_hip.hpp"
#include "randomise_bodies.hpp"
#include "tipsy.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <span>

#define NBH_API extern "C" __attribute__((visibility("default")))

NBH_API void nbh_srand(unsigned seed) { std::srand(seed); }

NBH_API void nbh_randomise_f32(int config, float* pos, float* vel, std::size_t nb_bodies, float cluster_scale, float velocity_scale) {
    randomise_bodies<float>(static_cast<NBodyConfig>(config), std::span<float>(pos, 4 * nb_bodies), std::span<float>(vel, 4 * nb_bodies), cluster_scale, velocity_scale);
}
NBH_API void nbh_randomise_f64(int config, double* pos, double* vel, std::size_t nb_bodies, float cluster_scale, float velocity_scale) {
    randomise_bodies<double>(static_cast<NBodyConfig>(config), std::span<double>(pos, 4 * nb_bodies), std::span<double>(vel, 4 * nb_bodies), cluster_scale, velocity_scale);
}

NBH_API void nbh_scale_params_for(std::size_t nb_bodies, float* cluster_scale, float* velocity_scale) {
    auto params = Compute::demo_params[0];
    Compute::scale_params_for(nb_bodies, params);
    *cluster_scale  = params.cluster_scale;
    *velocity_scale = params.velocity_scale;
}

NBH_API int nbh_demo_params(std::size_t index, float* out5) {
    if (index >= Compute::demo_params.size()) return -1;
    const auto& p = Compute::demo_params[index];
    out5[0] = p.time_step, out5[1] = p.cluster_scale, out5[2] = p.velocity_scale, out5[3] = p.softening, out5[4] = p.damping;
    return 0;
}

// Drives the reference-shaped stack (Compute -> ComputeHIP -> BodySystemHIP*) for the GPU tests: N bodies, `steps32`
// fp32 steps, switch_precision() to fp64 (compute_cuda.cpp:152-181), `steps64` fp64 steps, switch back.
// out32_a: fp32 state after the fp32 steps; out64: fp64 state after the fp64 steps; out32_b: fp32 state after switching
// back (positions then velocities, 4N each).  Returns 0, or -1 on any exception.
NBH_API int nbh_precision_switch_roundtrip(std::size_t nb_bodies, int hostmem, int steps32, int steps64, float* out32_a, double* out64, float* out32_b) {
    try {
        auto compute = Compute(false, false, false, true, hostmem != 0, 256, nb_bodies, {});
        const auto n4 = 4 * compute.nb_bodies();
        for (int s = 0; s < steps32; ++s) compute.update_simulation();
        auto copy = [n4](auto span, auto* dst) { std::copy(span.begin(), span.begin() + static_cast<std::ptrdiff_t>(n4), dst); };
        copy(compute.positions_fp32(), out32_a);
        copy(compute.velocities_fp32(), out32_a + n4);
        compute.switch_precision();
        if (!compute.fp64_enabled()) return -2;
        for (int s = 0; s < steps64; ++s) compute.update_simulation();
        copy(compute.positions_fp64(), out64);
        copy(compute.velocities_fp64(), out64 + n4);
        compute.switch_precision();
        if (compute.fp64_enabled()) return -3;
        copy(compute.positions_fp32(), out32_b);
        copy(compute.velocities_fp32(), out32_b + n4);
        return 0;
    } catch (...) { return -1; }
}

// Compute::compare_results through the reference-shaped stack: 1 = check passed, 0 = check failed, -1 = exception.
// `injected_error` perturbs the FAST result (body 0, x) before the check -- a check that cannot fail checks nothing.
NBH_API int nbh_compare_results(std::size_t nb_bodies, int fp64, int hostmem, double injected_error) {
    try {
        auto compute = Compute(fp64 != 0, false, true, false, hostmem != 0, 256, nb_bodies, {});
        return compute.compare_results(injected_error) ? 1 : 0;
    } catch (...) { return -1; }
}

// Compute::select_demo + `steps` x update_simulation in the given mode; out = positions then velocities (4N each).
NBH_API int nbh_run_demo(std::size_t nb_bodies, std::size_t demo, int mode, int steps, float* out) {
    try {
        const auto saved              = nbody_hip::integration_mode();
        nbody_hip::integration_mode() = mode;
        auto compute                  = Compute(false, false, false, false, false, 256, nb_bodies, {});
        compute.select_demo(demo);
        for (int s = 0; s < steps; ++s) compute.update_simulation();
        const auto n4  = 4 * compute.nb_bodies();
        const auto pos = compute.positions_fp32();
        std::copy(pos.begin(), pos.begin() + static_cast<std::ptrdiff_t>(n4), out);
        const auto vel = compute.velocities_fp32();
        std::copy(vel.begin(), vel.begin() + static_cast<std::ptrdiff_t>(n4), out + n4);
        nbody_hip::integration_mode() = saved;
        return 0;
    } catch (...) { return -1; }
}

// tipsy round trip: returns the padded body count, or -1 on error; fills up to `capacity` bodies
NBH_API long nbh_read_tipsy(const char* path, double* pos, double* vel, std::size_t capacity) {
    try {
        auto [p, v] = read_tipsy_file(path);
        const auto n = p.size() / 4;
        if (n > capacity) return -2;
        std::memcpy(pos, p.data(), p.size() * sizeof(double));
        std::memcpy(vel, v.data(), v.size() * sizeof(double));
        return static_cast<long>(n);
    } catch (...) { return -1; }
}
NBH_API int nbh_write_tipsy(const char* path, const double* pos, const double* vel, std::size_t nb_bodies, int ndark) {
    try {
        write_tipsy_file(path, std::span<const double>(pos, 4 * nb_bodies), std::span<const double>(vel, 4 * nb_bodies), ndark);
        return 0;
    } catch (...) { return -1; }
}
