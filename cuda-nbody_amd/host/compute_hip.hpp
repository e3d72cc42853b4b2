// compute_hip.hpp -- GPU facade: device checks, N rounding, storage-variant selection, benchmark timing,
// precision switch and the self-check.  Mirrors ComputeCUDA (/root/reference/src/nbody/compute_cuda.{hpp,cpp})
// minus the OpenGL interop variant (display only; an MI355X is a headless accelerator).
#pragma once

#include "device_array.hpp"
#include "nbody_config.hpp"

#include <chrono>
#include <concepts>
#include <memory>
#include <span>
#include <vector>

struct NBodyParams;
template <std::floating_point T> class BodySystemHIP;

class ComputeHIP {
 public:
    ComputeHIP(bool enable_host_mem, int block_size, bool fp64_enabled, std::size_t num_bodies, const NBodyParams& params);
    ComputeHIP(bool enable_host_mem, int block_size, bool fp64_enabled, std::size_t num_bodies, const NBodyParams& params, std::vector<float> positions_fp32, std::vector<float> velocities_fp32, std::vector<double> positions_fp64,
               std::vector<double> velocities_fp64);

    auto nb_bodies() const noexcept { return nb_bodies_; }
    auto use_host_mem() const noexcept { return use_host_mem_; }
    auto fp64_enabled() const noexcept { return fp64_enabled_; }

    auto get_position_fp32() const -> std::span<const float>;
    auto get_position_fp64() const -> std::span<const double>;
    auto get_velocity_fp32() const -> std::span<const float>;
    auto get_velocity_fp64() const -> std::span<const double>;

    auto switch_precision() -> void;
    auto reset(const NBodyParams& params, NBodyConfig config) -> void;
    auto set_values(std::span<const float> positions, std::span<const float> velocities) -> void;
    auto set_values(std::span<const double> positions, std::span<const double> velocities) -> void;
    auto update(float dt) -> void;
    auto update_params(const NBodyParams& params) -> void;

    // One dt = 0.001 step of the FAST kernels against the bit-reproducing STRICT kernels started from the SAME
    // pre-step state, |dp| <= 5e-4 per component (the reference's tolerance, compute_cuda.cpp:297-323).
    auto compare_results(const NBodyParams& params) -> bool;

    using Milliseconds = std::chrono::duration<float, std::milli>;
    auto get_milliseconds_passed() -> Milliseconds;
    auto run_benchmark(int nb_iterations, float dt) -> Milliseconds;
    // extension (--graph): issue the timed iterations as one captured hipGraph instead of K launches
    auto use_graph(bool enable) noexcept -> void { use_graph_ = enable; }

    ~ComputeHIP() noexcept;

 private:
    template <std::floating_point TNew, std::floating_point TOld> auto switch_precision(BodySystemHIP<TNew>& new_nbody, const BodySystemHIP<TOld>& old_nbody) -> void;
    template <std::floating_point T> auto run_benchmark(int nb_iterations, float dt, BodySystemHIP<T>& nbody) -> Milliseconds;
    template <std::floating_point T> auto compare_results(const NBodyParams& params, BodySystemHIP<T>& nbody) const -> bool;

    std::size_t nb_bodies_ = 0;
    int         block_size_;
    bool        fp64_enabled_;
    bool        use_host_mem_;
    bool        double_supported_ = true;
    bool        use_graph_        = false;

    std::unique_ptr<BodySystemHIP<float>>  nbody_fp32_;
    std::unique_ptr<BodySystemHIP<double>> nbody_fp64_;

    HipEvent host_mem_sync_event_;
    HipEvent start_event_;
    HipEvent stop_event_;
};
