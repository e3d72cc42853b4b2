#include "bodysystemhip.hpp"

#include "integrate_nbody_hip.hpp"
#include "randomise_bodies.hpp"

// ctor chain and softening^2 = T(softening) * T(softening): /root/reference/src/nbody/bodysystemcuda.cpp:42-58
template <std::floating_point T>
BodySystemHIP<T>::BodySystemHIP(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params) : BodySystemHIP(nb_bodies, blockSize, params, std::vector<T>(nb_bodies * 4, T{0}), std::vector<T>(nb_bodies * 4, T{0})) {}

template <std::floating_point T>
BodySystemHIP<T>::BodySystemHIP(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities)
    : nb_bodies_(nb_bodies), host_pos_vec_(std::move(positions)), host_vel_vec_(std::move(velocities)), damping_(params.damping), block_size_(blockSize) {
    const auto softening = static_cast<T>(params.softening);
    softening_squared_   = softening * softening;
    apply_softening();
}

template <std::floating_point T> auto BodySystemHIP<T>::apply_softening() const -> void { set_softening_squared(softening_squared_); }

// :60-64
template <std::floating_point T> auto BodySystemHIP<T>::reset(const NBodyParams& params, NBodyConfig config) -> void {
    randomise_bodies<T>(config, host_pos_vec_, host_vel_vec_, params.cluster_scale, params.velocity_scale);
    set_position(host_pos_vec_);
    set_velocity(host_vel_vec_);
}

// :66-69
template <std::floating_point T> auto BodySystemHIP<T>::update_params(const NBodyParams& active_params) -> void {
    const auto softening = static_cast<T>(active_params.softening);
    softening_squared_   = softening * softening;
    apply_softening();
    damping_ = active_params.damping;
}

template class BodySystemHIP<float>;
template class BodySystemHIP<double>;
