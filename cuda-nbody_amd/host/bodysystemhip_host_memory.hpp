// bodysystemhip_host_memory.hpp -- --hostmem variant: the three arrays live in mapped, pinned host memory and
// the kernel reads/writes them over PCIe.  Mirrors BodySystemCUDAHostMemory<T>,
// /root/reference/src/nbody/bodysystemcuda_host_memory.{hpp,cpp}.
#pragma once

#include "bodysystemhip.hpp"
#include "device_array.hpp"

#include <array>

template <std::floating_point T> class BodySystemHIPHostMemory : public BodySystemHIP<T> {
 public:
    BodySystemHIPHostMemory(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params);
    BodySystemHIPHostMemory(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities);

    auto get_position() const -> std::span<const T> override;
    auto get_velocity() const -> std::span<const T> override;
    auto update(T deltaTime) -> void override;
    auto set_position(std::span<const T> data) -> void override;
    auto set_velocity(std::span<const T> data) -> void override;

 private:
    std::array<MappedArray<T>, 2> positions_{MappedArray<T>(4 * this->nb_bodies_, T{0}), MappedArray<T>(4 * this->nb_bodies_, T{0})};
    MappedArray<T>                velocities_{4 * this->nb_bodies_, T{0}};
};

extern template class BodySystemHIPHostMemory<float>;
extern template class BodySystemHIPHostMemory<double>;
