// oracle/ref_randomise_harness.cpp -- TEST INFRASTRUCTURE.
// extern "C" doorway onto the reference's own randomise_bodies<T> (span overload,
// /root/reference/src/nbody/randomise_bodies.cpp:47), which oracle/Makefile compiles UNMODIFIED
// from the reference tree into oracle/_ref/librandomise_ref.so.  Nothing from the reference is
// copied here; this file only includes its header and forwards the call.
#include "randomise_bodies.hpp"

#include <cstddef>
#include <cstdlib>
#include <span>

extern "C" {
__attribute__((visibility("default"))) void ref_srand(unsigned seed) { std::srand(seed); }

__attribute__((visibility("default"))) void ref_randomise_f32(int config, float* pos, float* vel, std::size_t nb_bodies, float cluster_scale, float velocity_scale) {
    randomise_bodies<float>(static_cast<NBodyConfig>(config), std::span<float>(pos, 4 * nb_bodies), std::span<float>(vel, 4 * nb_bodies), cluster_scale, velocity_scale);
}

__attribute__((visibility("default"))) void ref_randomise_f64(int config, double* pos, double* vel, std::size_t nb_bodies, float cluster_scale, float velocity_scale) {
    randomise_bodies<double>(static_cast<NBodyConfig>(config), std::span<double>(pos, 4 * nb_bodies), std::span<double>(vel, 4 * nb_bodies), cluster_scale, velocity_scale);
}
}
