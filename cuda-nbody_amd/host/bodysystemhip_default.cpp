#include "bodysystemhip_default.hpp"

#include "integrate_nbody_hip.hpp"

#include <cassert>
#include <utility>

// ctor: shell start-up configuration, /root/reference/src/nbody/bodysystemcuda_default.cu:8-17
template <std::floating_point T> BodySystemHIPDefault<T>::BodySystemHIPDefault(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params) : BodySystemHIP<T>(nb_bodies, blockSize, params) {
    BodySystemHIPDefault<T>::reset(params, NBodyConfig::NBODY_CONFIG_SHELL);
}

template <std::floating_point T>
BodySystemHIPDefault<T>::BodySystemHIPDefault(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities)
    : BodySystemHIP<T>(nb_bodies, blockSize, params, std::move(positions), std::move(velocities)) {
    set_position(this->host_pos_vec_);
    set_velocity(this->host_vel_vec_);
}

// write pos[1-read] from pos[read], then swap; asynchronous on the default stream   (:19-24)
template <std::floating_point T> auto BodySystemHIPDefault<T>::update(T deltaTime) -> void {
    this->apply_softening();
    integrateNbodySystem<T>(device_pos_[1 - this->current_read_].data(), device_pos_[this->current_read_].data(), device_vel_.data(), this->current_read_, deltaTime, this->damping_, this->nb_bodies_, static_cast<int>(this->block_size_));
    std::swap(this->current_read_, this->current_write_);
}

// blocking D2H into the host mirror   (:26-37)
template <std::floating_point T> auto BodySystemHIPDefault<T>::get_position() const -> std::span<const T> {
    device_pos_[this->current_read_].download(host_pos_);
    return host_pos_;
}
template <std::floating_point T> auto BodySystemHIPDefault<T>::get_velocity() const -> std::span<const T> {
    device_vel_.download(host_vel_);
    return host_vel_;
}

// blocking H2D, ping-pong indices back to 0/1   (:39-55)
template <std::floating_point T> auto BodySystemHIPDefault<T>::set_position(std::span<const T> data) -> void {
    assert(data.size() == 4 * this->nb_bodies_);
    this->current_read_  = 0;
    this->current_write_ = 1;
    device_pos_[this->current_read_].upload(data);
}
template <std::floating_point T> auto BodySystemHIPDefault<T>::set_velocity(std::span<const T> data) -> void {
    assert(data.size() == 4 * this->nb_bodies_);
    this->current_read_  = 0;
    this->current_write_ = 1;
    device_vel_.upload(data);
}

template class BodySystemHIPDefault<float>;
template class BodySystemHIPDefault<double>;
