// nbody.cpp -- command line of the MI355X N-body hot path.  Flag names, defaults, banner/benchmark output and
// exit codes follow the reference's main (/root/reference/src/nbody.cpp:254-409); the OpenGL viewer is out of
// scope, so a run needs --benchmark, --compare/--qatest or --steps.
//
//   reference flags : --fullscreen --fp64 --hostmem --benchmark --numbodies=<n> --compare --qatest --cpu
//                     --tipsy=<file> -i,--iterations=<n> --blockSize=<n>      (single-dash spellings accepted too)
//   extensions      : --numdevices=<n> | --devices=<list> (the NVIDIA sample's -numdevices, which this fork of it dropped)
//                     --mode=fast|strict  --config=shell|random|expand  --demo=<0..6>  --steps=<n>  --dump=<file>
//                     --seed=<n>  --graph  --no-workspace  --workspace-mib=<n>  --inject-error=<x> (test hook for --compare)  --alloc-limit-mib=<n> (test hook)
#include "compute.hpp"
#include "integrate_nbody_hip.hpp"

#include <dlfcn.h>  // (the --alloc-limit-mib test hook lives in the lab library: looked up, never linked)

#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <limits>
#include <new>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace {

enum class Status { OK = 0, CleanShutDown, InvalidArguments };

struct Options {
    bool                  fullscreen = false;
    bool                  fp64       = false;
    bool                  hostmem    = false;
    bool                  benchmark  = false;
    std::size_t           numbodies  = 0;
    bool                  compare    = false;
    bool                  qatest     = false;
    bool                  cpu        = false;
    std::filesystem::path tipsy;
    std::size_t           iterations = 10;
    int                   block_size = 256;
    // extensions
    int                   mode   = NB_MODE_FAST;
    NBodyConfig           config = NBodyConfig::NBODY_CONFIG_SHELL;
    std::size_t           steps  = 0;
    std::filesystem::path dump;
    std::optional<unsigned> seed;
    bool                  graph = false;
    bool                  no_workspace = false;
    std::size_t           workspace_mib = 0;  // 0: no bound of our own
    std::size_t           alloc_limit_mib = 0;  // test hook: device allocations above this are refused (0: none)
    std::vector<int>      devices;  // --numdevices=<n> (devices 0..n-1) or --devices=<a,b,...>: bodies sharded over several GPUs
    std::optional<std::size_t> demo;   // row of Compute::demo_params (the reference reaches them from the viewer's keys only)
    double                inject_error = 0.0;
};

constexpr auto help_text = R"(The MI355X NBody hot path (drop-in for cuda-nbody's compute path).
Usage: nbody [OPTIONS]

Options:
  -h,--help                   Print this help message and exit
  --fullscreen                Accepted for compatibility; there is no viewer on a headless accelerator
  --fp64                      Use double precision floating point values for simulation
  --hostmem                   Stores simulation data in host memory
  --benchmark                 Run benchmark to measure performance
  --numbodies UINT            Number of bodies (>= 1) to run in simulation
  --compare                   Compares the fast kernels with the bit-reproducing strict kernels (the CPU path's arithmetic)
  --qatest                    Runs a QA test
  --cpu                       Rejected: the CPU BodySystem path is test infrastructure (oracle/), not part of this build
  --tipsy TEXT:FILE           Load a tipsy model file for simulation
  -i,--iterations UINT [10]   Number of iterations to run in the benchmark
  --blockSize INT [256]       Workgroup / LDS tile size of the strict kernels (multiple of 64); a hint for the fast ones
  --mode TEXT [fast]          fast | strict (strict bit-reproduces the reference's CPU BodySystem path)
  --config TEXT [shell]       shell | random | expand initial configuration
  --numdevices UINT           Shard the bodies over GPUs 0..n-1 of this node (position tiles exchanged over RCCL / xGMI)
  --devices LIST              ... or over the GPUs in this comma-separated list
  --demo UINT                 Select row 0..6 of the demo parameter table (dt, scales, softening, damping) and reset
  --steps UINT                Advance this many steps (untimed) before --dump; with --compare: also print how far the fast
                              trajectory is from the strict one after this many steps (max / 99th percentile / median)
  --dump TEXT                 Write final positions then velocities (raw little-endian T[4N] each) to this file
  --seed UINT                 srand() this value first (the reference never seeds: default stream = seed 1)
  --graph                     --benchmark issues its (even number of) iterations as one captured hipGraph
  --no-workspace              Fast mode without scratch memory: every directed interaction evaluated, as the reference kernel does
                              (default: the body system owns a workspace and every PAIR of bodies is evaluated once)
  --workspace-mib UINT        Spend at most this many MiB on that workspace (the pair tournament is then cut into slices that share
                              one region of reaction planes; default: what the library asks for, at most a third of the device's memory)
  --inject-error FLOAT        Test hook: added to body 0's x of the fast result before --compare checks it
  --alloc-limit-mib UINT      Test hook (needs LD_PRELOAD=libnbody_hip_lab.so): device allocations above this many MiB are refused
)";

template <typename I> auto parse_number(std::string_view text, I& out) -> bool {
    const auto* first = text.data();
    const auto* last  = text.data() + text.size();
    const auto [ptr, ec] = std::from_chars(first, last, out);
    return ec == std::errc{} && ptr == last;
}

auto parse_args(int argc, char** argv) -> std::pair<Status, Options> {
    auto options = Options{};

    auto error = [&](const std::string& message) {
        std::fprintf(stderr,
                     "-------------------------------------------\n"
                     "CRITICAL ERROR:\n"
                     "%s\n"
                     "-------------------------------------------\n\n",
                     message.c_str());
        std::fprintf(stderr, "%s\n", help_text);
        return std::pair(Status::InvalidArguments, options);
    };

    for (int a = 1; a < argc; ++a) {
        auto arg = std::string_view(argv[a]);
        if (arg == "-h" || arg == "--help" || arg == "-help") {
            std::printf("%s\n", help_text);
            return std::pair(Status::CleanShutDown, options);
        }
        if (arg.size() < 2 || arg[0] != '-') return error("The following argument was not expected: " + std::string(arg));
        // "-name" and "--name" are the same option (the NVIDIA sample used one dash, CLI11 in the reference two)
        auto name = arg.substr(arg[1] == '-' ? 2 : 1);
        std::optional<std::string_view> value;
        if (const auto eq = name.find('='); eq != std::string_view::npos) {
            value = name.substr(eq + 1);
            name  = name.substr(0, eq);
        }
        auto take_value = [&]() -> std::optional<std::string_view> {
            if (value) return value;
            if (a + 1 < argc) return std::string_view(argv[++a]);
            return std::nullopt;
        };
        auto flag = [&](bool& target) -> bool {
            if (value) return false;
            target = true;
            return true;
        };

        bool ok = true;
        if (name == "fullscreen") ok = flag(options.fullscreen);
        else if (name == "fp64") ok = flag(options.fp64);
        else if (name == "hostmem") ok = flag(options.hostmem);
        else if (name == "benchmark") ok = flag(options.benchmark);
        else if (name == "compare") ok = flag(options.compare);
        else if (name == "qatest") ok = flag(options.qatest);
        else if (name == "cpu") ok = flag(options.cpu);
        else if (name == "graph") ok = flag(options.graph);
        else if (name == "no-workspace" || name == "no_workspace") ok = flag(options.no_workspace);
        else if (name == "numbodies") {
            const auto v = take_value();
            ok           = v && parse_number(*v, options.numbodies) && options.numbodies >= 1;
            if (!ok) return error("--numbodies: Value not in range 1 to " + std::to_string(std::numeric_limits<std::size_t>::max()));
        } else if (name == "i" || name == "iterations") {
            const auto v = take_value();
            ok           = v && parse_number(*v, options.iterations);
        } else if (name == "blockSize") {
            const auto v = take_value();
            ok           = v && parse_number(*v, options.block_size);
        } else if (name == "steps") {
            const auto v = take_value();
            ok           = v && parse_number(*v, options.steps);
        } else if (name == "workspace-mib" || name == "workspace_mib") {
            const auto v = take_value();
            ok           = v && parse_number(*v, options.workspace_mib);
        } else if (name == "alloc-limit-mib") {
            const auto v = take_value();
            ok           = v && parse_number(*v, options.alloc_limit_mib);
        } else if (name == "seed") {
            const auto v = take_value();
            unsigned   s = 0;
            ok           = v && parse_number(*v, s);
            if (ok) options.seed = s;
        } else if (name == "tipsy") {
            const auto v = take_value();
            ok           = v.has_value();
            if (ok) {
                options.tipsy = std::filesystem::path(std::string(*v));
                if (!std::filesystem::is_regular_file(options.tipsy)) return error("--tipsy: File does not exist: " + options.tipsy.string());
            }
        } else if (name == "dump") {
            const auto v = take_value();
            ok           = v.has_value();
            if (ok) options.dump = std::filesystem::path(std::string(*v));
        } else if (name == "numdevices") {
            const auto v = take_value();
            int        n = 0;
            ok           = v && parse_number(*v, n) && n >= 1;
            if (!ok) return error("--numdevices: Value not in range 1 to " + std::to_string(std::numeric_limits<int>::max()));
            options.devices.clear();
            for (int d = 0; d < n; ++d) options.devices.push_back(d);
        } else if (name == "devices") {
            const auto v = take_value();
            ok           = v.has_value() && !v->empty();
            options.devices.clear();
            auto rest = ok ? *v : std::string_view{};
            while (ok && !rest.empty()) {
                const auto comma = rest.find(',');
                int        d     = -1;
                ok               = parse_number(rest.substr(0, comma), d) && d >= 0;
                options.devices.push_back(d);
                rest = comma == std::string_view::npos ? std::string_view{} : rest.substr(comma + 1);
            }
        } else if (name == "demo") {
            const auto  v = take_value();
            std::size_t d = 0;
            ok            = v && parse_number(*v, d) && d < Compute::demo_params.size();
            if (!ok) return error("--demo: Value not in range 0 to " + std::to_string(Compute::demo_params.size() - 1));
            options.demo = d;
        } else if (name == "inject-error") {
            const auto v = take_value();
            ok           = v.has_value();
            if (ok) {
                char* end            = nullptr;
                const auto text      = std::string(*v);
                options.inject_error = std::strtod(text.c_str(), &end);
                ok                   = end != nullptr && *end == '\0' && end != text.c_str();
            }
        } else if (name == "mode") {
            const auto v = take_value();
            ok           = v && (*v == "fast" || *v == "strict");
            if (ok) options.mode = (*v == "strict") ? NB_MODE_STRICT : NB_MODE_FAST;
        } else if (name == "config") {
            const auto v = take_value();
            ok           = v && (*v == "shell" || *v == "random" || *v == "expand");
            if (ok) options.config = *v == "shell" ? NBodyConfig::NBODY_CONFIG_SHELL : *v == "random" ? NBodyConfig::NBODY_CONFIG_RANDOM : NBodyConfig::NBODY_CONFIG_EXPAND;
        } else {
            return error("The following argument was not expected: " + std::string(arg));
        }
        if (!ok) return error("Could not parse argument: " + std::string(arg));
    }

    // the reference prints this hint and the full help on every successful parse (nbody.cpp:315-316)
    std::printf("Run \" nbody - benchmark[-numbodies = <numBodies>] \" to measure performance\n");
    std::printf("%s\n", help_text);
    return std::pair(Status::OK, options);
}

template <typename T> auto dump_state(const std::filesystem::path& file, std::span<const T> pos, std::span<const T> vel) -> void {
    auto out = std::ofstream(file, std::ios::binary | std::ios::trunc);
    if (!out) throw std::runtime_error("cannot open dump file " + file.string());
    out.write(reinterpret_cast<const char*>(pos.data()), static_cast<std::streamsize>(pos.size_bytes()));
    out.write(reinterpret_cast<const char*>(vel.data()), static_cast<std::streamsize>(vel.size_bytes()));
}

}  // namespace

auto main(int argc, char** argv) -> int {
    try {
        const auto [status, cmd_options] = parse_args(argc, argv);
        if (Status::InvalidArguments == status) return 1;
        if (Status::CleanShutDown == status) return 0;

        std::printf("NOTE: The HIP N-body hot path.  Results may vary with the GPU's power state.\n\n");
        std::printf("> %s mode\n", cmd_options.fullscreen ? "Fullscreen" : "Windowed");

        if (cmd_options.seed) std::srand(*cmd_options.seed);
        nbody_hip::integration_mode() = cmd_options.mode;
        nbody_hip::use_workspace()    = !cmd_options.no_workspace;
        nbody_hip::workspace_cap_bytes() = cmd_options.workspace_mib << 20;
        if (cmd_options.alloc_limit_mib != 0) {
            // nb_set_alloc_limit is exported by libnbody_hip_lab.so only (include/nbody_hip_lab.h): the tests preload that library
            using SetLimit   = int (*)(std::size_t);
            const auto limit = reinterpret_cast<SetLimit>(dlsym(RTLD_DEFAULT, "nb_set_alloc_limit"));
            if (limit == nullptr) throw std::invalid_argument("--alloc-limit-mib is a test hook of the lab library: run with LD_PRELOAD=libnbody_hip_lab.so");
            (void)limit(cmd_options.alloc_limit_mib << 20);
        }

        const auto compare_to_cpu = (cmd_options.compare || cmd_options.qatest) && (!cmd_options.cpu);
        const auto headless_run   = cmd_options.benchmark || compare_to_cpu || cmd_options.steps > 0 || !cmd_options.dump.empty();
        if (!headless_run && !cmd_options.cpu) {
            throw std::invalid_argument("the interactive OpenGL viewer is out of scope on a headless accelerator: pass --benchmark, --compare/--qatest or --steps/--dump");
        }

        auto compute = Compute(cmd_options.fp64, cmd_options.cpu, compare_to_cpu, cmd_options.benchmark, cmd_options.hostmem, cmd_options.block_size, cmd_options.numbodies, cmd_options.tipsy, cmd_options.config, cmd_options.devices);

        compute.use_graph(cmd_options.graph);
        if (cmd_options.demo) compute.select_demo(*cmd_options.demo);
        if (cmd_options.benchmark) {
            const auto nb_iterations = cmd_options.iterations == 0 ? 10 : static_cast<int>(cmd_options.iterations);
            compute.run_benchmark(nb_iterations);
            return 0;
        }
        if (compare_to_cpu) {
            if (cmd_options.steps > 0) compute.report_trajectory_error(cmd_options.steps);  // (before the check steps the system)
            const auto result = compute.compare_results(cmd_options.inject_error);
            return static_cast<int>(!result);
        }
        for (auto s = std::size_t{0}; s < cmd_options.steps; ++s) compute.update_simulation();
        if (!cmd_options.dump.empty()) {
            if (compute.fp64_enabled()) {
                dump_state<double>(cmd_options.dump, compute.positions_fp64(), compute.velocities_fp64());
            } else {
                dump_state<float>(cmd_options.dump, compute.positions_fp32(), compute.velocities_fp32());
            }
        }
        return 0;
    } catch (const std::invalid_argument& e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    } catch (const std::bad_alloc&) {
        std::fprintf(stderr, "ERROR: Unable to allocate memory!\n");
        return 3;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 2;
    } catch (...) {
        std::printf("ERROR: An unknown error occurred! Please inform your local developer!\n");
        return 4;
    }
}
