// capi_host.cpp -- extern "C" doorway onto the C++ host mirror so pytest can drive it with ctypes
// (tests/test_host_cpp.py).  Not part of the drop-in boundary (that is include/nbody_hip.h).
#include "compute.hpp"
#include "randomise_bodies.hpp"
#include "tipsy.hpp"

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
