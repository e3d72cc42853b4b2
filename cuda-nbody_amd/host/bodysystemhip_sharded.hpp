// bodysystemhip_sharded.hpp -- the body system spread over several GPUs of one node (--numdevices / --devices).
//
// New: the reference is single-GPU (NVIDIA's original sample had a -numdevices mode; this fork removed it, SURVEY
// section 0).  Same interface as every other BodySystemHIP<T>, so ComputeHIP / Compute / the command line drive it
// unchanged.  One process, G devices (the library enqueues a step with one thread per device: csrc/nbody_comm.hip, StepCrew): device g owns bodies [g*N/G, (g+1)*N/G) -- its velocities, its
// slice of each new position array -- and holds full-size position arrays; every update() is one
// nb_sharded_step_all_* call (include/nbody_hip.h): per device the kernels of the own slice and of each position tile as
// it arrives over RCCL / xGMI, then the tile exchange of the new positions.  STRICT mode is bit-identical to one GPU.
// Each shard steps on a stream of its own, made by nb_comm_stream_create -- never on the default stream: with RCCL active a rank that
// computes on the null stream (or on a stream sharing its hardware queue) steps ~40 % slower, and so does one whose null stream
// merely waits for it after every step (measured, profiles/round5_hw_queue_collision.txt).  The host-side reads and writes
// synchronise the shard's stream themselves; ComputeHIP records its timing events on stream() = the first shard's.
// Like BodySystemHIPDefault, each shard owns the scratch memory the library asks for (nb_comm_workspace_bytes_*): FAST then
// evaluates every pair of bodies once, across the devices too (reaction sums travel to their owners); --no-workspace: off.
#pragma once

#include "bodysystemhip.hpp"
#include "device_array.hpp"

#include <span>
#include <vector>

template <std::floating_point T> class BodySystemHIPSharded final : public BodySystemHIP<T> {
 public:
    BodySystemHIPSharded(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::span<const int> devices);
    BodySystemHIPSharded(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::span<const int> devices, std::vector<T> positions, std::vector<T> velocities);
    ~BodySystemHIPSharded() override;

    auto get_position() const -> std::span<const T> override;
    auto get_velocity() const -> std::span<const T> override;
    auto set_position(std::span<const T> data) -> void override;
    auto set_velocity(std::span<const T> data) -> void override;
    auto update(T deltaTime) -> void override;
    auto stream() const noexcept -> nb_stream_t override { return shards_.empty() ? nullptr : shards_.front().stream; }

    auto nb_devices() const noexcept { return shards_.size(); }
    auto host_enqueue_ms_per_step() const noexcept -> double override { return enqueue_steps_ == 0 ? -1.0 : enqueue_ms_ / static_cast<double>(enqueue_steps_); }
    auto reset_host_enqueue() noexcept -> void override { enqueue_ms_ = 0, enqueue_steps_ = 0; }

 private:
    struct Shard {
        int            device = 0;
        DeviceArray<T> pos[2];
        DeviceArray<T> vel;
        DeviceArray<T> acc;
        DeviceArray<unsigned char> workspace;
        nb_stream_t    stream = nullptr;  // the shard steps here: nb_comm_stream_create (non-blocking, clear of the null stream's hardware queue)
    };
    auto allocate(std::span<const int> devices) -> void;
    auto release() noexcept -> void;  // communicators and streams (the destructor's work; also allocate()'s when a constructor throws half-way)
    auto ensure_workspaces() -> void;  // (re)lends every shard what the current mode asks for
    int  workspace_mode_ = -1;
    double        enqueue_ms_    = 0;  // host time of the nb_sharded_step_all_* calls since reset_host_enqueue()
    unsigned long enqueue_steps_ = 0;

    std::vector<Shard>     shards_;
    std::vector<nb_comm_t> comms_;
    mutable std::vector<T> host_pos_;
    mutable std::vector<T> host_vel_;
};

extern template class BodySystemHIPSharded<float>;
extern template class BodySystemHIPSharded<double>;
