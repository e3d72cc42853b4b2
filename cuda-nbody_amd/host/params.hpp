// params.hpp -- simulation parameters, same fields and order as the reference's NBodyParams
// (/root/reference/src/nbody/params.hpp:8-16); camera_origin is carried for layout parity only (no display here).
#pragma once

#include <array>

struct NBodyParams {
    float                time_step;
    float                cluster_scale;
    float                velocity_scale;
    float                softening;
    float                damping;
    std::array<float, 3> camera_origin;

    auto print() const -> void;
};
