// compute.hpp -- demo orchestration without the viewer: parameter table, N-dependent scales, perf statistics,
// benchmark and self-check dispatch.  Mirrors Compute (/root/reference/src/nbody/compute.{hpp,cpp}); the
// camera/slider/demo-cycling parts belong to the OpenGL viewer and are out of scope.
#pragma once

#include "nbody_config.hpp"
#include "params.hpp"

#include <array>
#include <chrono>
#include <filesystem>
#include <memory>
#include <span>
#include <vector>

class ComputeHIP;

class Compute {
 public:
    // same parameter order as the reference ctor (compute.hpp:19-27) minus enable_cycle_demo
    Compute(bool enable_fp64, bool enable_cpu, bool enable_compare_to_cpu, bool enable_benchmark, bool enable_host_memory, int block_size, std::size_t nb_bodies, const std::filesystem::path& tipsy_file,
            NBodyConfig initial_configuration = NBodyConfig::NBODY_CONFIG_SHELL);
    Compute(const Compute&)                    = delete;
    auto operator=(const Compute&) -> Compute& = delete;
    ~Compute() noexcept;

    auto nb_bodies() const noexcept { return num_bodies_; }
    auto& active_params() const noexcept { return active_params_; }
    auto interactions_per_second() const noexcept { return interactions_per_second_; }
    auto gflops() const noexcept { return g_flops_; }
    auto fp64_enabled() const noexcept { return fp64_enabled_; }

    auto run_benchmark(int nb_iterations) -> void;
    auto use_graph(bool enable) -> void;
    auto compare_results() -> bool;
    auto switch_precision() -> void;
    auto select_demo(std::size_t index) -> void;
    auto update_simulation() -> void;
    auto reset(NBodyConfig initial_configuration) -> void;
    auto update_params() -> void;

    auto positions_fp32() const -> std::span<const float>;
    auto positions_fp64() const -> std::span<const double>;
    auto velocities_fp32() const -> std::span<const float>;
    auto velocities_fp64() const -> std::span<const double>;

    // {dt, cluster scale, velocity scale, softening, damping, camera}   compute.hpp:90-97
    constexpr static auto demo_params = std::array{
        NBodyParams{0.016f, 1.54f, 8.0f, 0.1f, 1.0f, {0, -2, -100}},  NBodyParams{0.016f, 0.68f, 20.0f, 0.1f, 1.0f, {0, -2, -30}},     NBodyParams{0.0006f, 0.16f, 1000.0f, 1.0f, 1.0f, {0, 0, -15}},
        NBodyParams{0.0006f, 0.16f, 1000.0f, 1.0f, 1.0f, {0, 0, -15}}, NBodyParams{0.0019f, 0.32f, 276.0f, 1.0f, 1.0f, {0, 0, -50}},    NBodyParams{0.0016f, 0.32f, 272.0f, 0.145f, 1.0f, {0, 0, -50}},
        NBodyParams{0.016f, 6.04f, 0.0f, 1.0f, 1.0f, {0, 0, -50}}};

    // cluster / velocity scale as a function of N   compute.cpp:74-92
    static auto scale_params_for(std::size_t nb_bodies, NBodyParams& params) noexcept -> void;

 private:
    auto print_benchmark_results(int nb_iterations, float milliseconds) -> void;
    auto compute_perf_stats(float frequency) -> void;

    bool        fp64_enabled_;
    std::size_t active_demo_ = 0;
    std::size_t num_bodies_  = 16384;
    float       g_flops_                 = 0.f;
    float       interactions_per_second_ = 0.f;
    NBodyParams active_params_           = demo_params[0];

    std::unique_ptr<ComputeHIP> compute_hip_;

    template <typename T> struct TipsyData {
        std::vector<T> positions;
        std::vector<T> velocities;
    };
    TipsyData<float>  tipsy_data_fp32_;
    TipsyData<double> tipsy_data_fp64_;
};
