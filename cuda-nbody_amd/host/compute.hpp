// compute.hpp -- what sits between the command line and the GPU facade: the demo parameter table, the N-dependent
// start-up scales, performance accounting and the benchmark / self-check entry points.
//
// It keeps the public names of the reference's Compute (/root/reference/src/nbody/compute.{hpp,cpp}) for everything a
// headless run touches.  What belongs to the OpenGL viewer there -- camera, sliders, demo cycling on a timer, the
// display call -- has no counterpart here (out of scope: no display on an MI355X), and the CPU body system is
// not part of the product (it is the test oracle), so `enable_cpu` is rejected.
#pragma once

#include "nbody_types.hpp"

#include <array>
#include <filesystem>
#include <memory>
#include <span>
#include <vector>

class ComputeHIP;

class Compute {
 public:
    // Demo presets {dt, cluster scale, velocity scale, softening, damping, camera}; row 0 is what a benchmark runs.
    // Values: compute.hpp:90-97 of the reference.
    constexpr static auto demo_params = std::array{
        NBodyParams{0.016f, 1.54f, 8.0f, 0.1f, 1.0f, {0, -2, -100}},    //
        NBodyParams{0.016f, 0.68f, 20.0f, 0.1f, 1.0f, {0, -2, -30}},    //
        NBodyParams{0.0006f, 0.16f, 1000.0f, 1.0f, 1.0f, {0, 0, -15}},  //
        NBodyParams{0.0006f, 0.16f, 1000.0f, 1.0f, 1.0f, {0, 0, -15}},  //
        NBodyParams{0.0019f, 0.32f, 276.0f, 1.0f, 1.0f, {0, 0, -50}},   //
        NBodyParams{0.0016f, 0.32f, 272.0f, 0.145f, 1.0f, {0, 0, -50}}, //
        NBodyParams{0.016f, 6.04f, 0.0f, 1.0f, 1.0f, {0, 0, -50}}};

    // Start-up cluster / velocity scale as a function of the body count (compute.cpp:74-92); systems above 32 768
    // bodies keep the values already in `params`.
    static auto scale_params_for(std::size_t nb_bodies, NBodyParams& params) noexcept -> void;

    // Argument order of the reference's constructor (compute.hpp:19-27) without enable_cycle_demo; the trailing
    // parameters are extensions (the reference always starts from the shell configuration, on one GPU).
    Compute(bool enable_fp64, bool enable_cpu, bool enable_compare_to_cpu, bool enable_benchmark, bool enable_host_memory, int block_size, std::size_t nb_bodies, const std::filesystem::path& tipsy_file,
            NBodyConfig initial_configuration = NBodyConfig::NBODY_CONFIG_SHELL, std::vector<int> devices = {});
    Compute(const Compute&)                    = delete;
    auto operator=(const Compute&) -> Compute& = delete;
    ~Compute() noexcept;

    // ---- what a run does ---------------------------------------------------------------------------------------
    auto run_benchmark(int nb_iterations) -> void;  // prints the reference's three result lines
    auto compare_results(double injected_error = 0.0) -> bool;  // --compare / --qatest (injected_error: test hook)
    auto report_trajectory_error(std::size_t steps) -> void;    // --compare --steps=K: fast against strict after K steps
    auto update_simulation() -> void;               // one step of the active demo's time step
    auto reset(NBodyConfig initial_configuration) -> void;
    auto select_demo(std::size_t index) -> void;
    auto update_params() -> void;
    auto switch_precision() -> void;
    auto use_graph(bool enable) -> void;  // extension: --graph

    // ---- what it reports ---------------------------------------------------------------------------------------
    auto nb_bodies() const noexcept { return num_bodies_; }
    auto fp64_enabled() const noexcept { return fp64_enabled_; }
    auto& active_params() const noexcept { return active_params_; }
    auto interactions_per_second() const noexcept { return interactions_per_second_; }  // in units of 1e9, as printed
    auto gflops() const noexcept { return g_flops_; }
    auto positions_fp32() const -> std::span<const float>;
    auto positions_fp64() const -> std::span<const double>;
    auto velocities_fp32() const -> std::span<const float>;
    auto velocities_fp64() const -> std::span<const double>;

 private:
    auto print_benchmark_results(int nb_iterations, float milliseconds) -> void;
    auto compute_perf_stats(float frequency) -> void;

    template <typename T> struct Bodies {
        std::vector<T> positions;
        std::vector<T> velocities;
    };
    Bodies<float>  tipsy_data_fp32_;  // empty unless --tipsy
    Bodies<double> tipsy_data_fp64_;

    std::unique_ptr<ComputeHIP> compute_hip_;
    NBodyParams                 active_params_ = demo_params[0];
    std::size_t                 num_bodies_    = 16384;
    std::size_t                 active_demo_   = 0;
    bool                        fp64_enabled_;
    float                       interactions_per_second_ = 0.f;
    float                       g_flops_                 = 0.f;
};
