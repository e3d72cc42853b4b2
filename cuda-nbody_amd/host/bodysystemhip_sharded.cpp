#include "bodysystemhip_sharded.hpp"

#include "integrate_nbody_hip.hpp"

#include "../../include/nbody_hip_tuning.h"  // (nb_comm_last_enqueue_ms: what the host needed to enqueue a step)

#include <exception>

#include <cassert>
#include <stdexcept>
#include <string>
#include <utility>

namespace {
// hipMalloc / default-stream copies act on the CURRENT device: scope it for the calls of one shard
class CurrentDevice {
 public:
    explicit CurrentDevice(int device) {
        hip_check(nb_get_device(&saved_), "nb_get_device");
        if (saved_ != device) hip_check(nb_set_device(device), "nb_set_device");
    }
    ~CurrentDevice() { (void)nb_set_device(saved_); }
    CurrentDevice(const CurrentDevice&)                    = delete;
    auto operator=(const CurrentDevice&) -> CurrentDevice& = delete;

 private:
    int saved_ = 0;
};
}  // namespace

template <std::floating_point T>
BodySystemHIPSharded<T>::BodySystemHIPSharded(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::span<const int> devices) : BodySystemHIP<T>(nb_bodies, blockSize, params) {
    allocate(devices);
    this->reset(params, NBodyConfig::NBODY_CONFIG_SHELL);
}

template <std::floating_point T>
BodySystemHIPSharded<T>::BodySystemHIPSharded(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::span<const int> devices, std::vector<T> positions, std::vector<T> velocities)
    : BodySystemHIP<T>(nb_bodies, blockSize, params, std::move(positions), std::move(velocities)) {
    allocate(devices);
    set_position(this->host_pos_vec_);
    set_velocity(this->host_vel_vec_);
}

template <std::floating_point T> auto BodySystemHIPSharded<T>::allocate(std::span<const int> devices) -> void {
    if (devices.empty()) throw std::invalid_argument("--numdevices: at least one device");
    if (this->nb_bodies_ % devices.size() != 0) {
        throw std::invalid_argument("the number of bodies (" + std::to_string(this->nb_bodies_) + ") must be a multiple of the number of devices (" + std::to_string(devices.size()) + ")");
    }
    const auto values = static_cast<std::size_t>(this->nb_bodies_) * 4;
    host_pos_.assign(values, T{0});
    host_vel_.assign(values, T{0});
    comms_.assign(devices.size(), nullptr);
    hip_check(nb_comm_init_all(comms_.data(), static_cast<int>(devices.size()), devices.data()), "nb_comm_init_all");
    shards_.resize(devices.size());
    try {
        for (std::size_t g = 0; g < devices.size(); ++g) {
            auto&         shard = shards_[g];
            shard.device        = devices[g];
            CurrentDevice scope(shard.device);
            shard.pos[0] = DeviceArray<T>(values), shard.pos[1] = DeviceArray<T>(values);
            shard.vel = DeviceArray<T>(values), shard.acc = DeviceArray<T>(values);
            hip_check(nb_comm_stream_create(comms_[g], &shard.stream), "nb_comm_stream_create");
        }
    } catch (...) {  // (a constructor that throws runs no destructor: the communicators -- RCCL's included -- and streams made so far would stay)
        release();
        throw;
    }
}

template <std::floating_point T> BodySystemHIPSharded<T>::~BodySystemHIPSharded() { release(); }

template <std::floating_point T> auto BodySystemHIPSharded<T>::release() noexcept -> void {
    for (auto& shard : shards_) {
        if (shard.stream == nullptr) continue;
        (void)nb_set_device(shard.device);
        (void)nb_stream_synchronize(shard.stream);
    }
    for (auto& comm : comms_) {
        if (comm != nullptr) (void)nb_comm_destroy(comm);
        comm = nullptr;
    }
    for (auto& shard : shards_) {
        if (shard.stream == nullptr) continue;
        (void)nb_set_device(shard.device);
        (void)nb_stream_destroy(shard.stream);
        shard.stream = nullptr;
    }
    if (!shards_.empty()) (void)nb_set_device(shards_.front().device);
}

template <std::floating_point T> auto BodySystemHIPSharded<T>::set_position(std::span<const T> data) -> void {
    assert(data.size() == 4 * static_cast<std::size_t>(this->nb_bodies_));
    this->current_read_ = 0, this->current_write_ = 1;
    for (auto& shard : shards_) {
        CurrentDevice scope(shard.device);
        hip_check(nb_device_synchronize(), "nb_device_synchronize");  // nothing may still be exchanging into this array
        shard.pos[0].upload(data);
    }
}

template <std::floating_point T> auto BodySystemHIPSharded<T>::set_velocity(std::span<const T> data) -> void {
    assert(data.size() == 4 * static_cast<std::size_t>(this->nb_bodies_));
    this->current_read_ = 0, this->current_write_ = 1;
    for (auto& shard : shards_) {
        CurrentDevice scope(shard.device);
        hip_check(nb_stream_synchronize(shard.stream), "nb_stream_synchronize");  // (a step may still be updating it)
        shard.vel.upload(data);  // every device gets the whole array; it only ever touches its own slice
    }
}

// every device holds the complete current positions once the exchange has landed; read them from the first
template <std::floating_point T> auto BodySystemHIPSharded<T>::get_position() const -> std::span<const T> {
    const auto&   shard = shards_.front();
    CurrentDevice scope(shard.device);
    hip_check(nb_exchange_wait_all(comms_.front(), shard.stream), "nb_exchange_wait_all");
    hip_check(nb_stream_synchronize(shard.stream), "nb_stream_synchronize");  // (the copy below runs on the default stream, which orders nothing against this one)
    shard.pos[this->current_read_].download(host_pos_);
    return host_pos_;
}

// velocities live with their owners: collect the slices
template <std::floating_point T> auto BodySystemHIPSharded<T>::get_velocity() const -> std::span<const T> {
    const auto slice = static_cast<std::size_t>(this->nb_bodies_) / shards_.size() * 4;
    for (std::size_t g = 0; g < shards_.size(); ++g) {
        CurrentDevice scope(shards_[g].device);
        hip_check(nb_stream_synchronize(shards_[g].stream), "nb_stream_synchronize");
        hip_check(nb_d2h(host_vel_.data() + g * slice, shards_[g].vel.data() + g * slice, slice * sizeof(T), nullptr), "nb_d2h");
    }
    return host_vel_;
}

// The library allocates nothing: each shard owns what nb_comm_workspace_bytes_* asks for in the current mode (0 bytes: none; the
// library never asks for more than a third of a device's memory).  The layout of a step belongs to the communicator as a whole,
// so when the allocation fails on ANY shard, ALL shards lend nothing and the step is the one-sided tile schedule (as
// BodySystemHIPStored falls back to nb_integrate_* on one GPU).
template <std::floating_point T> auto BodySystemHIPSharded<T>::ensure_workspaces() -> void {
    const int mode = nbody_hip::use_workspace() ? nbody_hip::integration_mode() : -2;
    if (mode == workspace_mode_) return;
    workspace_mode_ = mode;
    std::vector<std::size_t> need(shards_.size(), 0);
    bool                     lend = mode >= 0;
    for (std::size_t g = 0; g < shards_.size() && lend; ++g) {
        if constexpr (std::same_as<T, float>) {
            hip_check(nb_comm_workspace_bytes_f32(comms_[g], this->nb_bodies_, mode, &need[g]), "nb_comm_workspace_bytes_f32");
        } else {
            hip_check(nb_comm_workspace_bytes_f64(comms_[g], this->nb_bodies_, mode, &need[g]), "nb_comm_workspace_bytes_f64");
        }
        CurrentDevice scope(shards_[g].device);
        if (need[g] > shards_[g].workspace.size()) {
            hip_check(nb_device_synchronize(), "nb_device_synchronize");  // nothing may still be using the old one
            try {
                shards_[g].workspace = DeviceArray<unsigned char>(need[g]);
            } catch (const std::bad_alloc&) {
                lend = false;  // no memory for it on this device: nobody steps pairwise
            }
        }
    }
    for (std::size_t g = 0; g < shards_.size(); ++g) {
        const bool have = lend && need[g] != 0;
        hip_check(nb_comm_set_workspace(comms_[g], have ? shards_[g].workspace.data() : nullptr, have ? need[g] : 0), "nb_comm_set_workspace");
    }
}

template <std::floating_point T> auto BodySystemHIPSharded<T>::update(T deltaTime) -> void {
    this->apply_softening();
    ensure_workspaces();
    const auto               n = shards_.size();
    std::vector<T*>          to(n), vel(n), acc(n);
    std::vector<const T*>    from(n);
    std::vector<nb_stream_t> streams(n);  // each shard's own stream (never the default stream: see the header)
    for (std::size_t g = 0; g < n; ++g) {
        to[g] = shards_[g].pos[this->current_write_].data(), from[g] = shards_[g].pos[this->current_read_].data();
        vel[g] = shards_[g].vel.data(), acc[g] = shards_[g].acc.data();
        streams[g] = shards_[g].stream;
    }
    int status;
    if constexpr (std::same_as<T, float>) {
        status = nb_sharded_step_all_f32(comms_.data(), static_cast<int>(n), to.data(), from.data(), vel.data(), acc.data(), this->nb_bodies_, deltaTime, this->damping_, static_cast<int>(this->block_size_),
                                         nbody_hip::integration_mode(), streams.data());
    } else {
        status = nb_sharded_step_all_f64(comms_.data(), static_cast<int>(n), to.data(), from.data(), vel.data(), acc.data(), this->nb_bodies_, deltaTime, this->damping_, static_cast<int>(this->block_size_),
                                         nbody_hip::integration_mode(), streams.data());
    }
    hip_check(status, "nb_sharded_step_all");
    if (double ms = 0; nb_comm_last_enqueue_ms(comms_.front(), &ms) == 0) enqueue_ms_ += ms, ++enqueue_steps_;
    std::swap(this->current_read_, this->current_write_);
}

template class BodySystemHIPSharded<float>;
template class BodySystemHIPSharded<double>;
