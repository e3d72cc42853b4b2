// integrate_nbody_hip.hpp -- the reference's device seam with its exact signatures, as thin inline wrappers over
// the C-ABI, so BodySystem-shaped host code compiles unchanged against the HIP library:
//   integrateNbodySystem<T>   /root/reference/src/nbody/integrate_nbody_cuda.hpp:5, def bodysystemcuda.cu:186-215
//   set_softening_squared     decl /root/reference/src/nbody/bodysystemcuda.cpp:37-38, def bodysystemcuda.cu:46-60
// Error behaviour is the reference's: a failed launch prints to stderr and exit(EXIT_FAILURE)
// (bodysystemcuda.cu:204-214); a failed softening upload throws std::runtime_error (:49-59).
#pragma once

#include "hip_check.hpp"

#include <concepts>
#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace nbody_hip {
// Arithmetic mode for subsequent integrateNbodySystem calls (NB_MODE_FAST by default).  Extension: the
// reference has one kernel; here NB_MODE_STRICT selects the kernels that bit-reproduce its CPU path.
inline int& integration_mode() {
    static int mode = NB_MODE_FAST;
    return mode;
}
// Extension: body systems that own device memory also own the scratch memory nb_workspace_bytes_* asks for, and step through
// nb_integrate_ws_* (FAST mode then evaluates every pair of bodies once, csrc/nbody_pair.hip).  `nbody --no-workspace`
// switches it off: every step is then exactly the reference-shaped integrateNbodySystem below.
inline bool& use_workspace() {
    static bool on = true;
    return on;
}
// ... and `nbody --workspace-mib=<n>` bounds what a body system spends on it (0 = whatever the library asks for: at most a third
// of the device's memory): the library then cuts the pair tournament into as many slices as the bound needs (nb_workspace_bytes_capped_*).
inline std::size_t& workspace_cap_bytes() {
    static std::size_t cap = 0;
    return cap;
}
}  // namespace nbody_hip

inline auto set_softening_squared(float softeningSq) -> void { hip_check(nb_set_softening_sq_f32(softeningSq), "set_softening_squared"); }
inline auto set_softening_squared(double softeningSq) -> void { hip_check(nb_set_softening_sq_f64(softeningSq), "set_softening_squared"); }

template <std::floating_point T>
void integrateNbodySystem(T* new_positions, const T* old_positions, T* velocities, [[maybe_unused]] unsigned int currentRead, T deltaTime, T damping, unsigned int numBodies, int blockSize) {
    int status;
    if constexpr (std::same_as<T, float>) {
        status = nb_integrate_f32(new_positions, old_positions, velocities, deltaTime, damping, numBodies, blockSize, nbody_hip::integration_mode(), nullptr);
    } else {
        static_assert(std::same_as<T, double>, "float or double");
        status = nb_integrate_f64(new_positions, old_positions, velocities, deltaTime, damping, numBodies, blockSize, nbody_hip::integration_mode(), nullptr);
    }
    if (status != 0) {
        std::fprintf(stderr, "%s(%i) : HIP error : Kernel execution failed : (%d) %s.\n", __FILE__, __LINE__, status, nb_error_string(status));
        std::exit(EXIT_FAILURE);
    }
}

// The same seam with a caller-owned workspace (include/nbody_hip.h, nb_integrate_ws_*): same arguments, same error behaviour.
template <std::floating_point T>
void integrateNbodySystemWs(T* new_positions, const T* old_positions, T* velocities, [[maybe_unused]] unsigned int currentRead, T deltaTime, T damping, unsigned int numBodies, int blockSize, void* workspace,
                            std::size_t workspace_bytes) {
    int status;
    if constexpr (std::same_as<T, float>) {
        status = nb_integrate_ws_f32(new_positions, old_positions, velocities, deltaTime, damping, numBodies, blockSize, nbody_hip::integration_mode(), workspace, workspace_bytes, nullptr);
    } else {
        static_assert(std::same_as<T, double>, "float or double");
        status = nb_integrate_ws_f64(new_positions, old_positions, velocities, deltaTime, damping, numBodies, blockSize, nbody_hip::integration_mode(), workspace, workspace_bytes, nullptr);
    }
    if (status != 0) {
        std::fprintf(stderr, "%s(%i) : HIP error : Kernel execution failed : (%d) %s.\n", __FILE__, __LINE__, status, nb_error_string(status));
        std::exit(EXIT_FAILURE);
    }
}
