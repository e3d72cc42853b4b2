// randomise_bodies.hpp -- initial conditions, interleaved {x,y,z,m}/{vx,vy,vz,0} layout.
// Same signature as the reference's span overload (/root/reference/src/nbody/randomise_bodies.hpp:38) and,
// by test (tests/test_host_cpp.py against the reference's own code in oracle/_ref), the same bits for the
// same libc rand() state.
#pragma once

#include "nbody_types.hpp"

#include <concepts>
#include <span>

template <std::floating_point T> auto randomise_bodies(NBodyConfig config, std::span<T> pos, std::span<T> vel, float clusterScale, float velocityScale) noexcept -> void;

extern template auto randomise_bodies<float>(NBodyConfig, std::span<float>, std::span<float>, float, float) noexcept -> void;
extern template auto randomise_bodies<double>(NBodyConfig, std::span<double>, std::span<double>, float, float) noexcept -> void;
