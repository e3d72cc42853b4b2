// bodysystemhip.hpp -- abstract GPU body system; the interface of the reference's BodySystemCUDA<T>
// (/root/reference/src/nbody/bodysystemcuda.hpp:38-72) on HIP: same members, same virtuals, same ping-pong state.
#pragma once

#include "nbody_types.hpp"

#include "../../include/nbody_hip.h"  // nb_stream_t (an opaque handle: no HIP header on this side)

#include <concepts>
#include <span>
#include <vector>



template <std::floating_point T> class BodySystemHIP {
 public:
    using Type                    = T;
    constexpr static auto use_cpu = false;

    BodySystemHIP(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params);
    BodySystemHIP(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities);

    auto virtual get_position() const -> std::span<const T> = 0;
    auto virtual get_velocity() const -> std::span<const T> = 0;

    auto reset(const NBodyParams& params, NBodyConfig config) -> void;

    auto virtual update(T deltaTime) -> void = 0;

    // Extension: `steps` updates issued as one unit.  The default is a loop of update(); the device-memory
    // variant replays a captured hipGraph (steps even), built by prepare_many() outside any timed region.
    auto virtual prepare_many([[maybe_unused]] T deltaTime, [[maybe_unused]] unsigned steps) -> void {}
    auto virtual update_many(T deltaTime, unsigned steps) -> void {
        for (unsigned s = 0; s < steps; ++s) update(deltaTime);
    }

    auto update_params(const NBodyParams& active_params) -> void;

    auto virtual set_position(std::span<const T> data) -> void = 0;
    auto virtual set_velocity(std::span<const T> data) -> void = 0;

    auto nb_bodies() const noexcept { return nb_bodies_; }

    // The stream update() enqueues on (events that time the steps are recorded there).  The default stream for every variant of the
    // reference (everything there runs on stream 0); the sharded system steps on streams of its own -- see bodysystemhip_sharded.hpp.
    auto virtual stream() const noexcept -> nb_stream_t { return nullptr; }

    // Extension: what the HOST needed to enqueue a step, averaged over the update() calls since the last reset (milliseconds; < 0 = this
    // variant does not measure it).  The sharded system reports it (nb_comm_last_enqueue_ms): a step whose enqueue takes longer than
    // its kernels is bound by the host, and `--benchmark --numdevices N` says so.
    auto virtual host_enqueue_ms_per_step() const noexcept -> double { return -1.0; }
    auto virtual reset_host_enqueue() noexcept -> void {}

    virtual ~BodySystemHIP() = default;

 protected:
    // the softening constant is process-global per precision (as the reference's __constant__ is); every
    // update() re-asserts this system's value so two systems with different softening can coexist
    auto apply_softening() const -> void;

    unsigned int nb_bodies_;

    std::vector<T> host_pos_vec_;
    std::vector<T> host_vel_vec_;

    T damping_ = 0.995f;
    T softening_squared_{};

    unsigned int current_read_  = 0u;
    unsigned int current_write_ = 1u;

    unsigned int block_size_;
};

extern template class BodySystemHIP<float>;
extern template class BodySystemHIP<double>;
