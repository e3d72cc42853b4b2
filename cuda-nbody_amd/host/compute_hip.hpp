// compute_hip.hpp -- the GPU facade between Compute and the body systems.
//
// Same public surface as the reference's ComputeCUDA (/root/reference/src/nbody/compute_cuda.hpp:22-91) -- Compute and
// anything else written against it keeps compiling -- minus the OpenGL interop variant (display only; an MI355X is
// a headless accelerator).  It owns one fp32 and one fp64 body system side by side, as the reference does, and
// routes every call to the one that is active.
#pragma once

#include "device_array.hpp"
#include "nbody_types.hpp"

#include <chrono>
#include <concepts>
#include <memory>
#include <span>
#include <vector>

template <std::floating_point T> class BodySystemHIP;

class ComputeHIP {
 public:
    using Milliseconds = std::chrono::duration<float, std::milli>;

    // ---- construction: device checks, N rounding (or #CUs * 4 * blockSize when N == 0), storage variant ------------
    // `devices`: empty = the current device (the reference's behaviour); otherwise the bodies are sharded over these GPUs
    // (--numdevices / --devices: BodySystemHIPSharded, position tiles exchanged over RCCL)
    ComputeHIP(bool enable_host_mem, int block_size, bool fp64_enabled, std::size_t num_bodies, const NBodyParams& params, std::vector<int> devices = {});
    ComputeHIP(bool enable_host_mem, int block_size, bool fp64_enabled, std::size_t num_bodies, const NBodyParams& params, std::vector<float> positions_fp32, std::vector<float> velocities_fp32,
               std::vector<double> positions_fp64, std::vector<double> velocities_fp64, std::vector<int> devices = {});
    ~ComputeHIP() noexcept;

    // ---- state ---------------------------------------------------------------------------------------------------
    auto nb_bodies() const noexcept { return nb_bodies_; }
    auto use_host_mem() const noexcept { return use_host_mem_; }
    auto fp64_enabled() const noexcept { return fp64_enabled_; }
    auto get_position_fp32() const -> std::span<const float>;
    auto get_position_fp64() const -> std::span<const double>;
    auto get_velocity_fp32() const -> std::span<const float>;   // extension (the reference exposes positions only)
    auto get_velocity_fp64() const -> std::span<const double>;  // extension

    // ---- simulation ----------------------------------------------------------------------------------------------
    auto update(float dt) -> void;
    auto reset(const NBodyParams& params, NBodyConfig config) -> void;
    auto update_params(const NBodyParams& params) -> void;
    auto set_values(std::span<const float> positions, std::span<const float> velocities) -> void;
    auto set_values(std::span<const double> positions, std::span<const double> velocities) -> void;
    auto switch_precision() -> void;  // fp32 <-> fp64 through the host, compute_cuda.cpp:152-181,205-218

    // ---- measurement ---------------------------------------------------------------------------------------------
    auto get_milliseconds_passed() -> Milliseconds;                    // since the last call (event pair)
    auto run_benchmark(int nb_iterations, float dt) -> Milliseconds;   // 1 untimed step, then K between two events
    auto use_graph(bool enable) noexcept -> void { use_graph_ = enable; }  // extension (--graph): the K steps as one hipGraph
    // extension: host milliseconds per step the last run_benchmark needed to ENQUEUE its steps (< 0: the body system does not say -- only the sharded one does)
    auto host_enqueue_ms_per_step() const noexcept -> double { return host_enqueue_ms_per_step_; }

    // One dt = 0.001 step of the FAST kernels against the bit-reproducing STRICT kernels started from the SAME
    // pre-step state, |dp| <= 5e-4 per component (the reference's tolerance, compute_cuda.cpp:297-323).
    // `injected_error` (test hook, --inject-error) is added to the x coordinate of body 0 of the FAST result before
    // the check, so that a test can see the check fail (and the process exit with 1, nbody.cpp:375-379).
    auto compare_results(const NBodyParams& params, double injected_error = 0.0) -> bool;
    auto report_trajectory_error(const NBodyParams& params, std::size_t steps) -> void;  // --compare --steps=K (extension)

 private:
    // calls f(system) with the active precision's body system
    template <typename F> auto with_active(F&& f) -> decltype(auto);
    template <std::floating_point To, std::floating_point From> auto convert_state(BodySystemHIP<To>& to, const BodySystemHIP<From>& from) -> void;
    template <std::floating_point T> auto compare_results(const NBodyParams& params, BodySystemHIP<T>& nbody, double injected_error) const -> bool;
    template <std::floating_point T> auto report_trajectory_error(const NBodyParams& params, BodySystemHIP<T>& nbody, std::size_t steps) const -> void;

    std::size_t nb_bodies_ = 0;
    int         block_size_;
    bool        fp64_enabled_;
    bool        use_host_mem_;
    bool        use_graph_ = false;
    double      host_enqueue_ms_per_step_ = -1.0;

    std::unique_ptr<BodySystemHIP<float>>  nbody_fp32_;
    std::unique_ptr<BodySystemHIP<double>> nbody_fp64_;

    HipEvent host_mem_sync_event_;
    HipEvent start_event_;
    HipEvent stop_event_;
};
