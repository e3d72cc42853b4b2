#include "params.hpp"

#include "text.hpp"

#include <cstdio>

// same rendering as /root/reference/src/nbody/params.cpp:5-7
auto NBodyParams::print() const -> void {
    using text::shortest;
    std::printf("{ %s, %s, %s, %s, %s, %s, %s, %s },\n", shortest(time_step).c_str(), shortest(cluster_scale).c_str(), shortest(velocity_scale).c_str(), shortest(softening).c_str(), shortest(damping).c_str(),
                shortest(camera_origin[0]).c_str(), shortest(camera_origin[1]).c_str(), shortest(camera_origin[2]).c_str());
}
