// hip_check.hpp -- turns the C-ABI's int status into the exceptions the reference throws
// (std::runtime_error(cudaGetErrorName(result)), /root/reference/src/nbody/bodysystemcuda.cu:49-59,
// unique_mapped_span.cpp:13-21).  Host code never includes hip_runtime.h: everything goes through nbody_hip.h.
#pragma once

#include "../../include/nbody_hip.h"

#include <stdexcept>
#include <string>

inline auto hip_check(int status, const char* what) -> void {
    if (status != 0) {
        throw std::runtime_error(std::string(what) + ": " + nb_error_string(status));
    }
}
