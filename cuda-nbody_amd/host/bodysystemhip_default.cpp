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

template <std::floating_point T> BodySystemHIPDefault<T>::~BodySystemHIPDefault() { drop_graph(); }

template <std::floating_point T> auto BodySystemHIPDefault<T>::drop_graph() noexcept -> void {
    if (graph_ != nullptr) (void)nb_graph_destroy(graph_);
    graph_ = nullptr;
}

// Launch-bound small systems: capture `steps` ping-pong launches once, replay them with one host call.
template <std::floating_point T> auto BodySystemHIPDefault<T>::prepare_many(T deltaTime, unsigned steps) -> void {
    if (steps < 2 || (steps & 1u)) return;  // odd counts fall back to the loop in update_many
    if (graph_ != nullptr && graph_dt_ == deltaTime && graph_steps_ == steps && graph_read_ == this->current_read_) return;
    drop_graph();
    this->apply_softening();
    T* a = device_pos_[this->current_read_].data();
    T* b = device_pos_[1 - this->current_read_].data();
    int status;
    if constexpr (std::same_as<T, float>) {
        status = nb_graph_create_f32(&graph_, a, b, device_vel_.data(), deltaTime, this->damping_, this->nb_bodies_, static_cast<int>(this->block_size_), nbody_hip::integration_mode(), steps);
    } else {
        status = nb_graph_create_f64(&graph_, a, b, device_vel_.data(), deltaTime, this->damping_, this->nb_bodies_, static_cast<int>(this->block_size_), nbody_hip::integration_mode(), steps);
    }
    hip_check(status, "nb_graph_create");
    graph_dt_ = deltaTime, graph_steps_ = steps, graph_read_ = this->current_read_;
}

template <std::floating_point T> auto BodySystemHIPDefault<T>::update_many(T deltaTime, unsigned steps) -> void {
    if (steps < 2 || (steps & 1u)) {
        BodySystemHIP<T>::update_many(deltaTime, steps);
        return;
    }
    prepare_many(deltaTime, steps);
    hip_check(nb_graph_launch(graph_, nullptr), "nb_graph_launch");  // even step count: the read index is unchanged
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
