// randomise_bodies.cpp -- RANDOM / SHELL / EXPAND start-up configurations.
//
// Behavioural contract (reference: /root/reference/src/nbody/randomise_bodies.cpp:37-189): bodies are drawn from
// the process-global libc rand() stream (never seeded by the program), candidates outside the unit ball are
// rejected AFTER their draws are consumed, masses are 1 and velocity.w is 0.  The arithmetic below is arranged to
// round exactly like the reference's (operand types included: SHELL's scale/vscale stay `float` for T = double),
// which tests/test_host_cpp.py checks bit-for-bit against the reference's own translation unit.
#include "randomise_bodies.hpp"

#include <cmath>
#include <cstddef>
#include <cstdlib>

namespace {

template <typename T> struct Triple {
    T x, y, z;
};

// uniform in [0,1] and [-1,1] from one rand() draw each   (reference :37-43)
template <typename T> auto unit_draw() noexcept -> T { return std::rand() / static_cast<T>(RAND_MAX); }
template <typename T> auto signed_draw() noexcept -> T { return std::rand() * (T{2.0f} / static_cast<T>(RAND_MAX)) - T{1.0f}; }

template <typename T> auto signed_triple() noexcept -> Triple<T> {
    const T a = signed_draw<T>();
    const T b = signed_draw<T>();
    const T c = signed_draw<T>();
    return {a, b, c};
}

template <typename T> auto norm2(const Triple<T>& t) noexcept -> T { return t.x * t.x + t.y * t.y + t.z * t.z; }

// scales t to unit length unless it is (numerically) zero; returns the original length   (reference :14-24)
template <typename T> auto make_unit(Triple<T>& t) noexcept -> T {
    const T length = std::sqrt(norm2(t));
    if (length > 1e-6) {
        t.x /= length;
        t.y /= length;
        t.z /= length;
    }
    return length;
}

template <typename T> auto store(std::span<T> out, std::size_t body, T a, T b, T c, T w) noexcept -> void {
    out[4 * body + 0] = a;
    out[4 * body + 1] = b;
    out[4 * body + 2] = c;
    out[4 * body + 3] = w;
}

// uniform in the unit ball, velocities likewise   (reference :57-99)
template <typename T> auto fill_random(std::span<T> pos, std::span<T> vel, std::size_t count, float cluster, float velocity) noexcept -> void {
    const T per_1024 = count / T{1024};
    const T scale    = cluster * (T{1} < per_1024 ? per_1024 : T{1});
    const T vscale   = velocity * scale;
    for (std::size_t body = 0; body < count;) {
        const auto p = signed_triple<T>();
        if (norm2(p) > 1) continue;
        const auto v = signed_triple<T>();
        if (norm2(v) > 1) continue;
        store<T>(pos, body, p.x * scale, p.y * scale, p.z * scale, 1.0f);
        store<T>(vel, body, v.x * vscale, v.y * vscale, v.z * vscale, 0.0f);
        ++body;
    }
}

// directions on the sphere, each axis stretched by its own radius in [2.5, 4] x scale, rotating about z   (reference :101-147)
template <typename T> auto fill_shell(std::span<T> pos, std::span<T> vel, std::size_t count, float cluster, float velocity) noexcept -> void {
    const float scale  = cluster;
    const float vscale = scale * velocity;
    const T     inner  = T{2.5f} * scale;
    const T     outer  = T{4} * scale;
    for (std::size_t body = 0; body < count;) {
        auto dir = signed_triple<T>();
        if (make_unit(dir) > 1) continue;
        const T px = dir.x * (inner + (outer - inner) * unit_draw<T>());
        const T py = dir.y * (inner + (outer - inner) * unit_draw<T>());
        const T pz = dir.z * (inner + (outer - inner) * unit_draw<T>());
        store<T>(pos, body, px, py, pz, 1.0f);

        Triple<T> axis{0, 0, 1};
        if (1 - dir.z < 1e-6) {
            axis.x = dir.y;
            axis.y = dir.x;
            make_unit(axis);
        }
        // v = (p x axis) * vscale
        const T cx = py * axis.z - pz * axis.y;
        const T cy = pz * axis.x - px * axis.z;
        const T cz = px * axis.y - py * axis.x;
        store<T>(vel, body, cx * vscale, cy * vscale, cz * vscale, 0.0f);
        ++body;
    }
}

// uniform in the ball, velocity proportional to position (Hubble-like expansion)   (reference :149-187)
template <typename T> auto fill_expand(std::span<T> pos, std::span<T> vel, std::size_t count, float cluster, float velocity) noexcept -> void {
    T scale = cluster * count / T{1024};
    if (scale < 1) scale = cluster;
    const T vscale = scale * velocity;
    for (std::size_t body = 0; body < count;) {
        const auto p = signed_triple<T>();
        if (norm2(p) > 1) continue;
        store<T>(pos, body, p.x * scale, p.y * scale, p.z * scale, 1.0f);
        store<T>(vel, body, p.x * vscale, p.y * vscale, p.z * vscale, 0.0f);
        ++body;
    }
}

}  // namespace

template <std::floating_point T> auto randomise_bodies(NBodyConfig config, std::span<T> pos, std::span<T> vel, float clusterScale, float velocityScale) noexcept -> void {
    const auto count = pos.size() / 4;
    switch (config) {
        case NBodyConfig::NBODY_CONFIG_SHELL: fill_shell<T>(pos, vel, count, clusterScale, velocityScale); break;
        case NBodyConfig::NBODY_CONFIG_EXPAND: fill_expand<T>(pos, vel, count, clusterScale, velocityScale); break;
        case NBodyConfig::NBODY_CONFIG_RANDOM:
        default: fill_random<T>(pos, vel, count, clusterScale, velocityScale); break;
    }
}

template auto randomise_bodies<float>(NBodyConfig, std::span<float>, std::span<float>, float, float) noexcept -> void;
template auto randomise_bodies<double>(NBodyConfig, std::span<double>, std::span<double>, float, float) noexcept -> void;
