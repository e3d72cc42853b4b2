#include "bodysystemhip_host_memory.hpp"

#include "integrate_nbody_hip.hpp"

#include <algorithm>
#include <cassert>
#include <utility>

template <std::floating_point T> BodySystemHIPHostMemory<T>::BodySystemHIPHostMemory(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params) : BodySystemHIP<T>(nb_bodies, blockSize, params) {
    BodySystemHIPHostMemory<T>::reset(params, NBodyConfig::NBODY_CONFIG_SHELL);
}

template <std::floating_point T>
BodySystemHIPHostMemory<T>::BodySystemHIPHostMemory(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities)
    : BodySystemHIP<T>(nb_bodies, blockSize, params, std::move(positions), std::move(velocities)) {
    set_position(this->host_pos_vec_);
    set_velocity(this->host_vel_vec_);
}

template <std::floating_point T> auto BodySystemHIPHostMemory<T>::update(T deltaTime) -> void {
    this->apply_softening();
    integrateNbodySystem<T>(positions_[1 - this->current_read_].device_ptr(), positions_[this->current_read_].device_ptr(), velocities_.device_ptr(), this->current_read_, deltaTime, this->damping_, this->nb_bodies_,
                            static_cast<int>(this->block_size_));
    std::swap(this->current_read_, this->current_write_);
}

// The reference hands out the mapped host pointer with no synchronisation (its caller syncs on an event first,
// compute_cuda.cpp:284).  Reading results the GPU may still be writing is a foot-gun, so this variant waits for
// the device before exposing the span; a caller that already synchronised pays nothing measurable.
template <std::floating_point T> auto BodySystemHIPHostMemory<T>::get_position() const -> std::span<const T> {
    hip_check(nb_device_synchronize(), "nb_device_synchronize");
    return {positions_[this->current_read_].host_ptr(), static_cast<std::size_t>(this->nb_bodies_) * 4};
}
template <std::floating_point T> auto BodySystemHIPHostMemory<T>::get_velocity() const -> std::span<const T> {
    hip_check(nb_device_synchronize(), "nb_device_synchronize");
    return {velocities_.host_ptr(), static_cast<std::size_t>(this->nb_bodies_) * 4};
}

template <std::floating_point T> auto BodySystemHIPHostMemory<T>::set_position(std::span<const T> data) -> void {
    assert(data.size() == 4 * this->nb_bodies_);
    hip_check(nb_device_synchronize(), "nb_device_synchronize");
    this->current_read_  = 0;
    this->current_write_ = 1;
    std::ranges::copy(data, positions_[this->current_read_].host_ptr());
}
template <std::floating_point T> auto BodySystemHIPHostMemory<T>::set_velocity(std::span<const T> data) -> void {
    assert(data.size() == 4 * this->nb_bodies_);
    hip_check(nb_device_synchronize(), "nb_device_synchronize");
    this->current_read_  = 0;
    this->current_write_ = 1;
    std::ranges::copy(data, velocities_.host_ptr());
}

template class BodySystemHIPHostMemory<float>;
template class BodySystemHIPHostMemory<double>;
