#include "compute_hip.hpp"

#include "bodysystemhip.hpp"
#include "bodysystemhip_sharded.hpp"
#include "bodysystemhip_storage.hpp"
#include "integrate_nbody_hip.hpp"
#include "text.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace {

// device discovery: /root/reference/src/nbody/compute_cuda.cpp:16-48
auto get_main_device() -> nb_device_info_t {
    int count = 0;
    hip_check(nb_device_count(&count), "nb_device_count");
    if (count == 0) throw std::runtime_error("gpuDeviceInit() HIP error: no devices supporting HIP.\n");
    int device = 0;
    hip_check(nb_get_device(&device), "nb_get_device");
    nb_device_info_t info{};
    hip_check(nb_device_info(device, &info), "nb_device_info");
    return info;
}

}  // namespace

ComputeHIP::ComputeHIP(bool enable_host_mem, int block_size, bool fp64_enabled, std::size_t num_bodies, const NBodyParams& params, std::vector<int> devices)
    : ComputeHIP(enable_host_mem, block_size, fp64_enabled, num_bodies, params, {}, {}, {}, {}, std::move(devices)) {}

// compute_cuda.cpp:55-150
ComputeHIP::ComputeHIP(bool enable_host_mem, int block_size, bool fp64_enabled, std::size_t num_bodies, const NBodyParams& params, std::vector<float> positions_fp32, std::vector<float> velocities_fp32,
                       std::vector<double> positions_fp64, std::vector<double> velocities_fp64, std::vector<int> devices)
    : block_size_(block_size), fp64_enabled_(fp64_enabled), use_host_mem_(enable_host_mem) {
    if (!devices.empty()) {
        int count = 0;
        hip_check(nb_device_count(&count), "nb_device_count");
        for (const auto d : devices) {
            if (d < 0 || d >= count) throw std::invalid_argument("--devices: device " + std::to_string(d) + " does not exist (" + std::to_string(count) + " visible)");
        }
        if (enable_host_mem) throw std::invalid_argument("--hostmem and --numdevices/--devices cannot be combined");
        hip_check(nb_set_device(devices.front()), "nb_set_device");
    }
    const auto device = get_main_device();

    std::printf("> %s HIP device: [%s], %d compute units\n", device.arch, device.name, device.compute_units);

    if (use_host_mem_ && !device.can_map_host_memory) {
        throw std::invalid_argument(std::string("Device ") + device.name + " cannot map host memory!");
    }
    // (the reference refuses --fp64 on CC <= 1.2 parts here, :89-96; every gfx9 GPU has native fp64)
    if (block_size_ <= 0 || block_size_ % 64 != 0 || block_size_ > 1024) {
        throw std::invalid_argument("--blockSize must be a multiple of the 64-lane wavefront, at most 1024");
    }

    if (num_bodies != 0u) {
        nb_bodies_ = num_bodies;
        // the reference kernel needs N % blockSize == 0 (bodysystemcuda.cu:153-155) so its facade rounds N up
        // (:103-110).  The HIP kernels take any N, but the rounding is part of the CLI's observable behaviour.
        if (nb_bodies_ % static_cast<std::size_t>(block_size_)) {
            const auto rounded = ((nb_bodies_ / block_size_) + 1) * block_size_;
            std::printf("Warning: \"number of bodies\" specified %zu is not a multiple of %d.\n", nb_bodies_, block_size_);
            std::printf("Rounding up to the nearest multiple: %zu.\n", rounded);
            nb_bodies_ = rounded;
        } else {
            std::printf("number of bodies = %zu\n", nb_bodies_);
        }
    } else {
        // default: #CUs * 4 * workgroup size (:111-114) = 262 144 on an MI355X with --blockSize 256
        nb_bodies_ = static_cast<std::size_t>(block_size_) * 4u * static_cast<std::size_t>(device.compute_units);
    }

    if (!devices.empty()) {
        if (nb_bodies_ % devices.size() != 0) {
            throw std::invalid_argument("the number of bodies (" + std::to_string(nb_bodies_) + ") must be a multiple of the number of devices (" + std::to_string(devices.size()) + ")");
        }
        std::printf("> %zu Devices used for simulation (bodies sharded, %zu per device)\n", devices.size(), nb_bodies_ / devices.size());
    }
    std::printf("> Simulation data stored in %s memory\n", use_host_mem_ ? "system" : "video");
    std::printf("> %s precision floating point simulation\n", fp64_enabled_ ? "Double" : "Single");

    const auto allocate = [&]<typename System32, typename System64>() {
        const auto n = static_cast<unsigned int>(nb_bodies_);
        const auto b = static_cast<unsigned int>(block_size_);
        if (!positions_fp32.empty()) {
            nbody_fp32_ = std::make_unique<System32>(n, b, params, std::move(positions_fp32), std::move(velocities_fp32));
            nbody_fp64_ = std::make_unique<System64>(n, b, params, std::move(positions_fp64), std::move(velocities_fp64));
        } else {
            nbody_fp32_ = std::make_unique<System32>(n, b, params);
            nbody_fp64_ = std::make_unique<System64>(n, b, params);
        }
    };
    if (!devices.empty()) {
        const auto n = static_cast<unsigned int>(nb_bodies_);
        const auto b = static_cast<unsigned int>(block_size_);
        if (!positions_fp32.empty()) {
            nbody_fp32_ = std::make_unique<BodySystemHIPSharded<float>>(n, b, params, devices, std::move(positions_fp32), std::move(velocities_fp32));
            nbody_fp64_ = std::make_unique<BodySystemHIPSharded<double>>(n, b, params, devices, std::move(positions_fp64), std::move(velocities_fp64));
        } else {
            nbody_fp32_ = std::make_unique<BodySystemHIPSharded<float>>(n, b, params, devices);
            nbody_fp64_ = std::make_unique<BodySystemHIPSharded<double>>(n, b, params, devices);
        }
    } else if (use_host_mem_) {
        allocate.template operator()<BodySystemHIPHostMemory<float>, BodySystemHIPHostMemory<double>>();
    } else {
        allocate.template operator()<BodySystemHIPDefault<float>, BodySystemHIPDefault<double>>();
    }

    start_event_.record(fp64_enabled_ ? nbody_fp64_->stream() : nbody_fp32_->stream());
}

template <typename F> auto ComputeHIP::with_active(F&& f) -> decltype(auto) {
    if (fp64_enabled_) return f(*nbody_fp64_);
    return f(*nbody_fp32_);
}

// The state crosses precisions through the host (compute_cuda.cpp:152-181): widening is exact, narrowing rounds once.
template <std::floating_point To, std::floating_point From> auto ComputeHIP::convert_state(BodySystemHIP<To>& to, const BodySystemHIP<From>& from) -> void {
    static_assert(!std::is_same_v<To, From>);
    hip_check(nb_device_synchronize(), "nb_device_synchronize");
    const auto pos = from.get_position();
    const auto converted_pos = std::vector<To>(pos.begin(), pos.end());
    const auto vel = from.get_velocity();
    const auto converted_vel = std::vector<To>(vel.begin(), vel.end());
    to.set_position(converted_pos);
    to.set_velocity(converted_vel);
    hip_check(nb_device_synchronize(), "nb_device_synchronize");
}

auto ComputeHIP::switch_precision() -> void {
    if (fp64_enabled_) {
        convert_state(*nbody_fp32_, *nbody_fp64_);
    } else {
        convert_state(*nbody_fp64_, *nbody_fp32_);
    }
    fp64_enabled_ = !fp64_enabled_;
    std::printf("> %s precision floating point simulation\n", fp64_enabled_ ? "Double" : "Single");
}

// one untimed step to prime the device, then K steps between two events   (:183-203)
auto ComputeHIP::run_benchmark(int nb_iterations, float dt) -> Milliseconds {
    return with_active([&](auto& nbody) {
        nbody.update(dt);
        if (use_graph_ && nb_iterations >= 2 && nb_iterations % 2 == 0) {
            nbody.prepare_many(dt, static_cast<unsigned>(nb_iterations));  // capture + instantiate outside the timed region
            start_event_.record(nbody.stream());
            nbody.update_many(dt, static_cast<unsigned>(nb_iterations));
            return get_milliseconds_passed();
        }
        start_event_.record(nbody.stream());
        nbody.reset_host_enqueue();
        for (int i = 0; i < nb_iterations; ++i) nbody.update(dt);
        host_enqueue_ms_per_step_ = nbody.host_enqueue_ms_per_step();
        return get_milliseconds_passed();
    });
}

auto ComputeHIP::reset(const NBodyParams& params, NBodyConfig config) -> void {
    with_active([&](auto& nbody) { nbody.reset(params, config); });
}

auto ComputeHIP::set_values(std::span<const float> positions, std::span<const float> velocities) -> void {
    nbody_fp32_->set_position(positions);
    nbody_fp32_->set_velocity(velocities);
}
auto ComputeHIP::set_values(std::span<const double> positions, std::span<const double> velocities) -> void {
    nbody_fp64_->set_position(positions);
    nbody_fp64_->set_velocity(velocities);
}

auto ComputeHIP::update(float dt) -> void {
    host_mem_sync_event_.record(with_active([](auto& nbody) { return nbody.stream(); }));  // what a renderer of mapped host memory would wait on (compute_cuda.cpp:237-246,284)
    with_active([&](auto& nbody) { nbody.update(dt); });
}

auto ComputeHIP::get_position_fp32() const -> std::span<const float> { return nbody_fp32_->get_position(); }
auto ComputeHIP::get_position_fp64() const -> std::span<const double> { return nbody_fp64_->get_position(); }
auto ComputeHIP::get_velocity_fp32() const -> std::span<const float> { return nbody_fp32_->get_velocity(); }
auto ComputeHIP::get_velocity_fp64() const -> std::span<const double> { return nbody_fp64_->get_velocity(); }

auto ComputeHIP::update_params(const NBodyParams& params) -> void {
    with_active([&](auto& nbody) { nbody.update_params(params); });
}

// record stop, wait, elapsed, restart   (:263-272)
auto ComputeHIP::get_milliseconds_passed() -> Milliseconds {
    const auto on = with_active([](auto& nbody) { return nbody.stream(); });  // (where the steps ran: the default stream unless the system is sharded)
    stop_event_.record(on);
    stop_event_.synchronize();
    const auto ms = HipEvent::elapsed_ms(start_event_, stop_event_);
    start_event_.record(on);
    return Milliseconds{ms};
}

// The reference's check (:294-329) steps the GPU, then seeds a CPU system from the GPU's POST-step state and
// steps that too, i.e. compares t2 against t1.  Here both systems start from the same PRE-step
// state.  The checker is the STRICT kernel pair, which tests/test_gpu_parity.py hold to 0 ulp against the CPU
// BodySystem path, so "strict" below reads as "what the reference's CPU path computes".
template <std::floating_point T> auto ComputeHIP::compare_results(const NBodyParams& params, BodySystemHIP<T>& nbody, double injected_error) const -> bool {
    auto passed = true;

    const auto pos_span = nbody.get_position();
    auto       pos0     = std::vector<T>(pos_span.begin(), pos_span.end());
    const auto vel_span = nbody.get_velocity();
    auto       vel0     = std::vector<T>(vel_span.begin(), vel_span.end());

    const auto saved_mode = nbody_hip::integration_mode();
    {
        nbody_hip::integration_mode() = NB_MODE_STRICT;
        auto checker = BodySystemHIPDefault<T>(static_cast<unsigned int>(nb_bodies_), static_cast<unsigned int>(block_size_), params, pos0, vel0);
        checker.update(0.001f);
        const auto want_span = checker.get_position();
        const auto want      = std::vector<T>(want_span.begin(), want_span.end());

        nbody_hip::integration_mode() = NB_MODE_FAST;
        nbody.update(0.001f);
        const auto got_span = nbody.get_position();
        auto       got      = std::vector<T>(got_span.begin(), got_span.end());
        if (injected_error != 0.0 && !got.empty()) got[0] += static_cast<T>(injected_error);

        constexpr auto tolerance = T{0.0005f};
        for (auto i = std::size_t{0}; i < nb_bodies_; ++i) {
            for (auto c = std::size_t{0}; c < 3; ++c) {
                const auto a = want[4 * i + c], b = got[4 * i + c];
                if (!(std::abs(a - b) <= tolerance)) {
                    passed = false;
                    std::printf("Error: (strict)%s != (fast)%s\n", text::shortest(a).c_str(), text::shortest(b).c_str());
                }
            }
        }
    }
    nbody_hip::integration_mode() = saved_mode;
    if (passed) std::printf("  OK\n");
    return passed;
}

// Extension (--compare --steps=K): how far the FAST trajectory has drifted from the STRICT one -- the CPU BodySystem path's
// bits -- after K steps of the simulation's own time step, per body and relative to the body's distance from the origin.
// The system is chaotic: both are roundings of it, and against an fp64 trajectory the two are equally far
// (tests/test_gpu_parity.py); this prints which half of "fast and within 1e-4 of the CPU path after 100 steps" a mode meets.
template <std::floating_point T> auto ComputeHIP::report_trajectory_error(const NBodyParams& params, BodySystemHIP<T>& nbody, std::size_t steps) const -> void {
    const auto pos_span = nbody.get_position();
    const auto pos0     = std::vector<T>(pos_span.begin(), pos_span.end());
    const auto vel_span = nbody.get_velocity();
    const auto vel0     = std::vector<T>(vel_span.begin(), vel_span.end());
    const auto dt       = static_cast<T>(params.time_step);
    const auto saved    = nbody_hip::integration_mode();
    auto       run      = [&](int mode) {
        nbody_hip::integration_mode() = mode;
        auto system = BodySystemHIPDefault<T>(static_cast<unsigned int>(nb_bodies_), static_cast<unsigned int>(block_size_), params, pos0, vel0);
        for (auto s = std::size_t{0}; s < steps; ++s) system.update(dt);
        const auto span = system.get_position();
        return std::vector<T>(span.begin(), span.end());
    };
    const auto strict = run(NB_MODE_STRICT);
    const auto fast   = run(NB_MODE_FAST);
    nbody_hip::integration_mode() = saved;
    auto errors = std::vector<double>(nb_bodies_);
    for (auto i = std::size_t{0}; i < nb_bodies_; ++i) {
        double d2 = 0, r2 = 0;
        for (auto c = std::size_t{0}; c < 3; ++c) {
            const auto a = static_cast<double>(strict[4 * i + c]), b = static_cast<double>(fast[4 * i + c]);
            d2 += (a - b) * (a - b), r2 += a * a;
        }
        errors[i] = std::sqrt(d2 / (r2 > 0 ? r2 : 1));
    }
    std::sort(errors.begin(), errors.end());
    const auto at = [&](double q) { return errors[std::min(errors.size() - 1, static_cast<std::size_t>(q * static_cast<double>(errors.size())))]; };
    std::printf("> after %zu steps of dt = %s: |fast - strict| / |strict| per body: max %.3g, 99th percentile %.3g, median %.3g (strict = the CPU BodySystem path, 0 ulp)\n", steps,
                text::shortest(params.time_step).c_str(), errors.back(), at(0.99), at(0.5));
}

auto ComputeHIP::report_trajectory_error(const NBodyParams& params, std::size_t steps) -> void {
    with_active([&](auto& nbody) { report_trajectory_error(params, nbody, steps); });
}

auto ComputeHIP::compare_results(const NBodyParams& params, double injected_error) -> bool {
    return with_active([&](auto& nbody) { return compare_results(params, nbody, injected_error); });
}

ComputeHIP::~ComputeHIP() noexcept = default;
