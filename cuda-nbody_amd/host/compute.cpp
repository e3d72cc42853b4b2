#include "compute.hpp"

#include "compute_hip.hpp"
#include "text.hpp"
#include "tipsy.hpp"

#include <algorithm>
#include <cstdio>
#include <stdexcept>

namespace {
// compute.cpp:16-18
constexpr auto flops_per_interaction(bool fp64_enabled) { return fp64_enabled ? 30 : 20; }
}  // namespace

auto Compute::scale_params_for(std::size_t nb_bodies, NBodyParams& params) noexcept -> void {
    struct Row {
        std::size_t up_to;
        float       cluster, velocity;
    };
    constexpr Row table[] = {{1024, 1.52f, 2.f}, {2048, 1.56f, 2.64f}, {4096, 1.68f, 2.98f}, {8192, 1.98f, 2.9f}, {16384, 1.54f, 8.f}, {32768, 1.44f, 11.f}};
    for (const auto& row : table) {
        if (nb_bodies <= row.up_to) {
            params.cluster_scale  = row.cluster;
            params.velocity_scale = row.velocity;
            return;
        }
    }
    // larger systems keep the active demo's values
}

Compute::~Compute() noexcept = default;

// compute.cpp:27-103
Compute::Compute(bool enable_fp64, bool enable_cpu, [[maybe_unused]] bool enable_compare_to_cpu, [[maybe_unused]] bool enable_benchmark, bool enable_host_memory, int block_size, std::size_t nb_bodies,
                 const std::filesystem::path& tipsy_file, NBodyConfig initial_configuration, std::vector<int> devices)
    : fp64_enabled_(enable_fp64) {
    if (enable_cpu) {
        // The reference's --cpu path (BodySystemCPU) exists in this repository only as the test oracle
        // (oracle/nbody_oracle.c); the product has no CPU compute path and never falls back to one.
        throw std::invalid_argument("--cpu: this build has no CPU BodySystem path (it is test infrastructure under oracle/); run without --cpu");
    }
    if (!tipsy_file.empty()) {
        auto [positions, velocities] = read_tipsy_file(tipsy_file);
        tipsy_data_fp32_.positions.assign(positions.begin(), positions.end());
        tipsy_data_fp32_.velocities.assign(velocities.begin(), velocities.end());
        tipsy_data_fp64_.positions  = std::move(positions);
        tipsy_data_fp64_.velocities = std::move(velocities);
        // The reference takes N from --numbodies even with a file (compute.cpp:55-58) and then asserts the
        // sizes agree; here the file's (padded) body count wins unless --numbodies names the same number.
        const auto file_bodies = tipsy_data_fp64_.positions.size() / 4;
        if (nb_bodies != 0 && nb_bodies != file_bodies) {
            throw std::invalid_argument("--numbodies does not match the tipsy file (" + std::to_string(file_bodies) + " bodies after padding to a multiple of 256)");
        }
        compute_hip_ = std::make_unique<ComputeHIP>(enable_host_memory, block_size, enable_fp64, file_bodies, active_params_, tipsy_data_fp32_.positions, tipsy_data_fp32_.velocities, tipsy_data_fp64_.positions,
                                                    tipsy_data_fp64_.velocities, std::move(devices));
    } else {
        compute_hip_ = std::make_unique<ComputeHIP>(enable_host_memory, block_size, enable_fp64, nb_bodies, active_params_, std::move(devices));
    }
    num_bodies_ = compute_hip_->nb_bodies();

    scale_params_for(num_bodies_, active_params_);

    if (tipsy_file.empty()) compute_hip_->reset(active_params_, initial_configuration);
}

// compute.cpp:105-121 -- wording and number rendering of the three benchmark lines are the reference's
auto Compute::print_benchmark_results(int nb_iterations, float milliseconds) -> void {
    compute_perf_stats(nb_iterations * (1000.0f / milliseconds));
    std::printf("%zu bodies, total time for %d iterations: %s ms\n", num_bodies_, nb_iterations, text::width3(milliseconds).c_str());
    std::printf("= %s billion interactions per second\n", text::width3(interactions_per_second_).c_str());
    std::printf("= %s %s-precision GFLOP/s at %d flops per interaction\n", text::width3(g_flops_).c_str(), fp64_enabled_ ? "double" : "single", flops_per_interaction(fp64_enabled_));
}

auto Compute::compute_perf_stats(float frequency) -> void {
    interactions_per_second_ = (static_cast<float>(num_bodies_ * num_bodies_) * 1e-9f) * frequency;
    g_flops_                 = interactions_per_second_ * static_cast<float>(flops_per_interaction(fp64_enabled_));
}

auto Compute::switch_precision() -> void {
    compute_hip_->switch_precision();
    fp64_enabled_ = !fp64_enabled_;
}

// compute.cpp:156-187 without the camera.  One deliberate difference: the reference's select_demo leaves the body
// system's softening and damping at the PREVIOUS demo's values until a slider moves (only controls.cpp:51,64 call
// update_params); here the selected row's softening/damping take effect with the row, which is what its table means.
auto Compute::select_demo(std::size_t index) -> void {
    if (index >= demo_params.size()) throw std::invalid_argument("demo index out of range");
    active_demo_   = index;
    active_params_ = demo_params[index];
    compute_hip_->update_params(active_params_);
    reset(NBodyConfig::NBODY_CONFIG_SHELL);
}

auto Compute::update_simulation() -> void { compute_hip_->update(active_params_.time_step); }

auto Compute::reset(NBodyConfig initial_configuration) -> void {
    if (tipsy_data_fp32_.positions.empty()) {
        compute_hip_->reset(active_params_, initial_configuration);
    } else if (fp64_enabled_) {
        compute_hip_->set_values(tipsy_data_fp64_.positions, tipsy_data_fp64_.velocities);
    } else {
        compute_hip_->set_values(tipsy_data_fp32_.positions, tipsy_data_fp32_.velocities);
    }
}

auto Compute::update_params() -> void { compute_hip_->update_params(active_params_); }

auto Compute::run_benchmark(int nb_iterations) -> void {
    const auto milliseconds = compute_hip_->run_benchmark(nb_iterations, active_params_.time_step);
    print_benchmark_results(nb_iterations, milliseconds.count());
    // several devices driven by this one process: is the HOST what bounds a step?  (an extra line, after the reference's three)
    if (const double enqueue = compute_hip_->host_enqueue_ms_per_step(); enqueue >= 0) {
        std::printf("= %.3f ms of host time to enqueue a step (host_enqueue_ms_per_step; %.3f ms per step on the devices)\n", enqueue, milliseconds.count() / static_cast<float>(nb_iterations));
    }
}

auto Compute::use_graph(bool enable) -> void { compute_hip_->use_graph(enable); }

auto Compute::compare_results(double injected_error) -> bool { return compute_hip_->compare_results(active_params_, injected_error); }
auto Compute::report_trajectory_error(std::size_t steps) -> void { compute_hip_->report_trajectory_error(active_params_, steps); }

auto Compute::positions_fp32() const -> std::span<const float> { return compute_hip_->get_position_fp32(); }
auto Compute::positions_fp64() const -> std::span<const double> { return compute_hip_->get_position_fp64(); }
auto Compute::velocities_fp32() const -> std::span<const float> { return compute_hip_->get_velocity_fp32(); }
auto Compute::velocities_fp64() const -> std::span<const double> { return compute_hip_->get_velocity_fp64(); }
