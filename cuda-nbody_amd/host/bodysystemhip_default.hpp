// bodysystemhip_default.hpp -- device-memory storage variant (the benchmark path).
// Mirrors BodySystemCUDADefault<T>, /root/reference/src/nbody/bodysystemcuda_default.hpp:28-36.
#pragma once

#include "bodysystemhip.hpp"
#include "device_array.hpp"

#include <array>

template <std::floating_point T> class BodySystemHIPDefault : public BodySystemHIP<T> {
 public:
    BodySystemHIPDefault(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params);
    BodySystemHIPDefault(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities);

    auto get_position() const -> std::span<const T> override;
    auto get_velocity() const -> std::span<const T> override;
    auto update(T deltaTime) -> void override;
    auto prepare_many(T deltaTime, unsigned steps) -> void override;
    auto update_many(T deltaTime, unsigned steps) -> void override;
    ~BodySystemHIPDefault() override;
    auto set_position(std::span<const T> data) -> void override;
    auto set_velocity(std::span<const T> data) -> void override;

 private:
    // host mirrors the get_* spans alias (overwritten by the next get_*, as in the reference)
    mutable std::vector<T> host_pos_ = std::vector<T>(this->nb_bodies_ * 4, 0);
    mutable std::vector<T> host_vel_ = std::vector<T>(this->nb_bodies_ * 4, 0);

    std::array<DeviceArray<T>, 2> device_pos_{DeviceArray<T>(this->nb_bodies_ * 4), DeviceArray<T>(this->nb_bodies_ * 4)};
    DeviceArray<T>                device_vel_{this->nb_bodies_ * 4};

    // captured step loop (nb_graph_*): valid for one (dt, steps, read index) combination
    auto drop_graph() noexcept -> void;
    nb_graph_t   graph_       = nullptr;
    T            graph_dt_    = 0;
    unsigned     graph_steps_ = 0;
    unsigned int graph_read_  = 0;
};

extern template class BodySystemHIPDefault<float>;
extern template class BodySystemHIPDefault<double>;
