#include "bodysystemhip_storage.hpp"

#include "integrate_nbody_hip.hpp"

#include <cassert>
#include <stdexcept>
#include <utility>

template <std::floating_point T, template <std::floating_point> class Storage>
BodySystemHIPStored<T, Storage>::BodySystemHIPStored(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params) : BodySystemHIP<T>(nb_bodies, blockSize, params) {
    this->reset(params, NBodyConfig::NBODY_CONFIG_SHELL);
}

template <std::floating_point T, template <std::floating_point> class Storage>
BodySystemHIPStored<T, Storage>::BodySystemHIPStored(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities)
    : BodySystemHIP<T>(nb_bodies, blockSize, params, std::move(positions), std::move(velocities)) {
    set_position(this->host_pos_vec_);
    set_velocity(this->host_vel_vec_);
}

template <std::floating_point T, template <std::floating_point> class Storage> BodySystemHIPStored<T, Storage>::~BodySystemHIPStored() { drop_graph(); }

template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::rewind() noexcept -> void {
    this->current_read_  = 0;
    this->current_write_ = 1;
}

template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::set_position(std::span<const T> data) -> void {
    assert(data.size() == 4 * static_cast<std::size_t>(this->nb_bodies_));
    rewind();
    storage_.write_position(this->current_read_, data);
}

template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::set_velocity(std::span<const T> data) -> void {
    assert(data.size() == 4 * static_cast<std::size_t>(this->nb_bodies_));
    rewind();
    storage_.write_velocity(data);
}

// One step: positions[write] <- integrate(positions[read]), velocities in place, then the two indices trade places
// (bodysystemcuda_default.cu:19-24).  Asynchronous on the default stream.
template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::update(T deltaTime) -> void {
    this->apply_softening();
    ensure_workspace();
    if (workspace_bytes_ != 0) {
        integrateNbodySystemWs<T>(storage_.position_ptr(this->current_write_), storage_.position_ptr(this->current_read_), storage_.velocity_ptr(), this->current_read_, deltaTime, this->damping_, this->nb_bodies_,
                                  static_cast<int>(this->block_size_), workspace_.data(), workspace_bytes_);
    } else {
        integrateNbodySystem<T>(storage_.position_ptr(this->current_write_), storage_.position_ptr(this->current_read_), storage_.velocity_ptr(), this->current_read_, deltaTime, this->damping_, this->nb_bodies_,
                                static_cast<int>(this->block_size_));
    }
    std::swap(this->current_read_, this->current_write_);
}

// The library allocates nothing (the reference's ownership rule): the body system asks how much scratch memory the current
// mode wants for this many bodies and owns it, like its three body arrays.
template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::ensure_workspace() -> void {
    const int mode = (Storage<T>::workspace_capable && nbody_hip::use_workspace()) ? nbody_hip::integration_mode() : -2;
    if (mode == workspace_mode_) return;
    workspace_mode_  = mode;
    // Ask, allocate; when the device has no room for it, ask again for a form of the layout that fits in half as much (the library
    // cuts the tournament into more slices) -- and only when nothing fits is the step the one-sided kernel.
    std::size_t cap  = nbody_hip::workspace_cap_bytes() != 0 ? nbody_hip::workspace_cap_bytes() : ~std::size_t{0};
    std::size_t need = 0;
    while (mode >= 0) {
        if constexpr (std::same_as<T, float>) {
            hip_check(nb_workspace_bytes_capped_f32(this->nb_bodies_, mode, cap, &need), "nb_workspace_bytes_capped_f32");
        } else {
            hip_check(nb_workspace_bytes_capped_f64(this->nb_bodies_, mode, cap, &need), "nb_workspace_bytes_capped_f64");
        }
        if (need <= workspace_.size()) break;
        try {
            workspace_ = DeviceArray<unsigned char>(need);
            break;
        } catch (const std::bad_alloc&) {  // (out of device memory -- DeviceBadAlloc; any other failure is not ours to paper over)
            cap  = need / 2;
            need = 0;
        }
    }
    workspace_bytes_ = need;
}

template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::drop_graph() noexcept -> void {
    if (graph_ != nullptr) (void)nb_graph_destroy(graph_);
    graph_ = nullptr;
}

// Capture `steps` (even) ping-pong launches once; update_many() then replays them with one host call.
template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::prepare_many(T deltaTime, unsigned steps) -> void {
    if (steps < 2 || (steps & 1u)) return;  // odd counts fall back to the loop in update_many
    const auto mode = nbody_hip::integration_mode();
    // everything nb_graph_create_* bakes into the captured launches is part of the key: dt, step count, which buffer is
    // read first, mode, and -- kernel arguments too -- this system's damping and softening^2 (update_params changes them)
    ensure_workspace();
    const void* const workspace = workspace_bytes_ != 0 ? workspace_.data() : nullptr;
    if (graph_ != nullptr && graph_dt_ == deltaTime && graph_steps_ == steps && graph_read_ == this->current_read_ && graph_mode_ == mode && graph_damping_ == this->damping_ &&
        graph_softening_squared_ == this->softening_squared_ && graph_workspace_ == workspace) {
        return;
    }
    drop_graph();
    this->apply_softening();
    T*  from = storage_.position_ptr(this->current_read_);
    T*  to   = storage_.position_ptr(this->current_write_);
    int status;
    if constexpr (std::same_as<T, float>) {
        status = nb_graph_create_ws_f32(&graph_, from, to, storage_.velocity_ptr(), deltaTime, this->damping_, this->nb_bodies_, static_cast<int>(this->block_size_), mode, steps, workspace_.data(), workspace_bytes_);
    } else {
        status = nb_graph_create_ws_f64(&graph_, from, to, storage_.velocity_ptr(), deltaTime, this->damping_, this->nb_bodies_, static_cast<int>(this->block_size_), mode, steps, workspace_.data(), workspace_bytes_);
    }
    hip_check(status, "nb_graph_create");
    graph_dt_ = deltaTime, graph_steps_ = steps, graph_read_ = this->current_read_, graph_mode_ = mode;
    graph_damping_ = this->damping_, graph_softening_squared_ = this->softening_squared_, graph_workspace_ = workspace;
}

template <std::floating_point T, template <std::floating_point> class Storage> auto BodySystemHIPStored<T, Storage>::update_many(T deltaTime, unsigned steps) -> void {
    if (steps < 2 || (steps & 1u)) {
        BodySystemHIP<T>::update_many(deltaTime, steps);
        return;
    }
    prepare_many(deltaTime, steps);
    hip_check(nb_graph_launch(graph_, nullptr), "nb_graph_launch");  // even step count: the read index is unchanged
}

template class BodySystemHIPStored<float, DeviceStorage>;
template class BodySystemHIPStored<double, DeviceStorage>;
template class BodySystemHIPStored<float, MappedStorage>;
template class BodySystemHIPStored<double, MappedStorage>;
