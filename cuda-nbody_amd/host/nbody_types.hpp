// nbody_types.hpp -- the two plain types every layer of the host mirror shares.
//
// Field names, order and enumerator values are the reference's, so code written against
// /root/reference/src/nbody/params.hpp:8-16 (NBodyParams) and nbody_config.hpp:3 (NBodyConfig) compiles unchanged.
#pragma once

#include "text.hpp"

#include <array>
#include <cstdio>

// Which start-up distribution randomise_bodies() draws.
enum class NBodyConfig {
    NBODY_CONFIG_RANDOM,  // uniform in a ball, random velocities
    NBODY_CONFIG_SHELL,   // spherical shell in rotation (the only one the reference uses at start-up)
    NBODY_CONFIG_EXPAND,  // uniform in a ball, velocity proportional to position
    NBODY_NUM_CONFIGS
};

// One row of the demo table (compute.hpp:90-97) / the values behind the reference's sliders.
struct NBodyParams {
    float                time_step;       // dt of one update()
    float                cluster_scale;   // start-up geometry only
    float                velocity_scale;  // start-up geometry only
    float                softening;       // eps; the kernels take eps^2
    float                damping;         // velocity multiplier per step
    std::array<float, 3> camera_origin;   // viewer only; carried so the table rows read like the reference's

    // "{ dt, cluster, velocity, softening, damping, cx, cy, cz }," -- what the reference prints (params.cpp:5-7)
    auto print() const -> void {
        using text::shortest;
        std::printf("{ %s, %s, %s, %s, %s, %s, %s, %s },\n", shortest(time_step).c_str(), shortest(cluster_scale).c_str(), shortest(velocity_scale).c_str(), shortest(softening).c_str(),
                    shortest(damping).c_str(), shortest(camera_origin[0]).c_str(), shortest(camera_origin[1]).c_str(), shortest(camera_origin[2]).c_str());
    }
};
