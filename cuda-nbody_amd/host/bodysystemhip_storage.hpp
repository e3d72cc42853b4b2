// bodysystemhip_storage.hpp -- the concrete GPU body systems.
//
// The reference has one class per place the three body arrays can live (device memory:
// /root/reference/src/nbody/bodysystemcuda_default.{hpp,cu}; mapped host memory:
// bodysystemcuda_host_memory.{hpp,cpp}) with the ping-pong logic written out in each.  Here the ping-pong logic
// exists once, in BodySystemHIPStored<T, Storage>, and the place is a policy:
//
//   DeviceStorage<T>  three hipMalloc'd arrays + host mirrors; get_* are blocking D2H copies, set_* blocking H2D
//   MappedStorage<T>  three pinned, device-mapped host arrays (--hostmem): the kernel reads and writes them over
//                     PCIe, get_* hand out the host pointer after a device sync, set_* are host copies
//
// BodySystemHIPDefault<T> / BodySystemHIPHostMemory<T> are the reference's class names for the two instantiations.
#pragma once

#include "bodysystemhip.hpp"
#include "device_array.hpp"

#include <array>
#include <vector>

template <std::floating_point T> class DeviceStorage {
 public:
    explicit DeviceStorage(std::size_t values) : pos_{DeviceArray<T>(values), DeviceArray<T>(values)}, vel_(values), host_pos_(values, T{0}), host_vel_(values, T{0}) {}

    auto position_ptr(unsigned int which) const noexcept -> T* { return pos_[which].data(); }
    auto velocity_ptr() const noexcept -> T* { return vel_.data(); }

    auto write_position(unsigned int which, std::span<const T> data) -> void { pos_[which].upload(data); }
    auto write_velocity(std::span<const T> data) -> void { vel_.upload(data); }
    // the returned span aliases a host mirror that the next read overwrites (as in the reference)
    auto read_position(unsigned int which) const -> std::span<const T> {
        pos_[which].download(host_pos_);
        return host_pos_;
    }
    auto read_velocity() const -> std::span<const T> {
        vel_.download(host_vel_);
        return host_vel_;
    }
    constexpr static bool graph_capable = true;  // plain device pointers: the step loop can be captured in a hipGraph
    // the pairwise FAST layout re-reads the bodies j once per workgroup with vector loads: fine from HBM
    constexpr static bool workspace_capable = true;

 private:
    std::array<DeviceArray<T>, 2> pos_;
    DeviceArray<T>                vel_;
    mutable std::vector<T>        host_pos_;
    mutable std::vector<T>        host_vel_;
};

template <std::floating_point T> class MappedStorage {
 public:
    explicit MappedStorage(std::size_t values) : pos_{MappedArray<T>(values, T{0}), MappedArray<T>(values, T{0})}, vel_(values, T{0}), values_(values) {}

    auto position_ptr(unsigned int which) const noexcept -> T* { return pos_[which].device_ptr(); }
    auto velocity_ptr() const noexcept -> T* { return vel_.device_ptr(); }

    // The reference hands out / overwrites the mapped host memory with no synchronisation of its own (its caller
    // waits on an event first, compute_cuda.cpp:284).  Touching arrays the GPU may still be using is a foot-gun,
    // so every host-side access here waits for the device; a caller that already synchronised pays nothing.
    auto write_position(unsigned int which, std::span<const T> data) -> void {
        hip_check(nb_device_synchronize(), "nb_device_synchronize");
        std::copy(data.begin(), data.end(), pos_[which].host_ptr());
    }
    auto write_velocity(std::span<const T> data) -> void {
        hip_check(nb_device_synchronize(), "nb_device_synchronize");
        std::copy(data.begin(), data.end(), vel_.host_ptr());
    }
    auto read_position(unsigned int which) const -> std::span<const T> {
        hip_check(nb_device_synchronize(), "nb_device_synchronize");
        return {pos_[which].host_ptr(), values_};
    }
    auto read_velocity() const -> std::span<const T> {
        hip_check(nb_device_synchronize(), "nb_device_synchronize");
        return {vel_.host_ptr(), values_};
    }
    constexpr static bool graph_capable = true;
    // ... and not over PCIe: mapped host memory keeps the one-sided kernels (bodies j through the scalar cache)
    constexpr static bool workspace_capable = false;

 private:
    std::array<MappedArray<T>, 2> pos_;
    MappedArray<T>                vel_;
    std::size_t                   values_;
};

template <std::floating_point T, template <std::floating_point> class Storage> class BodySystemHIPStored final : public BodySystemHIP<T> {
 public:
    // shell start-up configuration drawn at construction (bodysystemcuda_default.cu:8-10)
    BodySystemHIPStored(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params);
    // caller-supplied bodies, e.g. from a tipsy file (:12-17)
    BodySystemHIPStored(unsigned int nb_bodies, unsigned int blockSize, const NBodyParams& params, std::vector<T> positions, std::vector<T> velocities);
    ~BodySystemHIPStored() override;

    auto get_position() const -> std::span<const T> override { return storage_.read_position(this->current_read_); }
    auto get_velocity() const -> std::span<const T> override { return storage_.read_velocity(); }
    auto set_position(std::span<const T> data) -> void override;
    auto set_velocity(std::span<const T> data) -> void override;

    auto update(T deltaTime) -> void override;
    auto prepare_many(T deltaTime, unsigned steps) -> void override;
    auto update_many(T deltaTime, unsigned steps) -> void override;

 private:
    auto rewind() noexcept -> void;  // set_* restart the ping-pong at read = 0 / write = 1
    auto drop_graph() noexcept -> void;
    auto ensure_workspace() -> void;  // (re)allocates what nb_workspace_bytes_* asks for in the current mode; 0 bytes = none

    Storage<T> storage_{static_cast<std::size_t>(this->nb_bodies_) * 4};

    // scratch memory of the pairwise FAST layout (nb_integrate_ws_*): owned here, as the three body arrays are
    DeviceArray<unsigned char> workspace_;
    std::size_t                workspace_bytes_ = 0;
    int                        workspace_mode_  = -1;

    // captured step loop (nb_graph_*): valid for one (dt, steps, read index, mode, damping, softening^2) combination
    nb_graph_t   graph_       = nullptr;
    T            graph_dt_    = 0;
    unsigned     graph_steps_ = 0;
    unsigned int graph_read_  = 0;
    int          graph_mode_  = 0;
    T            graph_damping_           = 0;
    T            graph_softening_squared_ = 0;
    const void*  graph_workspace_         = nullptr;
};

template <std::floating_point T> using BodySystemHIPDefault    = BodySystemHIPStored<T, DeviceStorage>;
template <std::floating_point T> using BodySystemHIPHostMemory = BodySystemHIPStored<T, MappedStorage>;

extern template class BodySystemHIPStored<float, DeviceStorage>;
extern template class BodySystemHIPStored<double, DeviceStorage>;
extern template class BodySystemHIPStored<float, MappedStorage>;
extern template class BodySystemHIPStored<double, MappedStorage>;
