// tipsy.cpp -- on-disk layout, from the reference's reader (/root/reference/src/nbody/tipsy.cpp:25-50,52-127):
//   header  : double time; int nbodies, ndim, nsph, ndark, nstar;  -> 28 bytes of fields, sizeof == 32 with padding
//   dark    : float mass; float pos[3]; float vel[3]; float eps; int phi;                    -> 36 bytes
//   star    : float mass; float pos[3]; float vel[3]; float metals; float tform; float eps; int phi;  -> 44 bytes
// The first `ndark` records are dark particles, the remaining nbodies - ndark are stars (gas records are never
// read).  mass -> position.w, eps -> velocity.w, phi (a particle id in this variant) is dropped.
#include "tipsy.hpp"

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>

namespace {

struct Header {
    double       time;
    std::int32_t nbodies;
    std::int32_t ndim;
    std::int32_t nsph;
    std::int32_t ndark;
    std::int32_t nstar;
};
static_assert(sizeof(Header) == 32, "the reference reads sizeof(Dump) == 32 bytes including tail padding");

struct Dark {
    float        mass;
    float        pos[3];
    float        vel[3];
    float        eps;
    std::int32_t phi;
};
static_assert(sizeof(Dark) == 36);

struct Star {
    float        mass;
    float        pos[3];
    float        vel[3];
    float        metals;
    float        tform;
    float        eps;
    std::int32_t phi;
};
static_assert(sizeof(Star) == 44);

template <typename R> auto read_record(std::ifstream& in, R& r) -> void { in.read(reinterpret_cast<char*>(&r), sizeof(R)); }
template <typename R> auto write_record(std::ofstream& out, const R& r) -> void { out.write(reinterpret_cast<const char*>(&r), sizeof(R)); }

template <typename R> auto append(const R& r, std::vector<double>& pos, std::vector<double>& vel) -> void {
    pos.insert(pos.end(), {r.pos[0], r.pos[1], r.pos[2], r.mass});
    vel.insert(vel.end(), {r.vel[0], r.vel[1], r.vel[2], r.eps});
}

}  // namespace

auto read_tipsy_file(const std::filesystem::path& fileName) -> std::array<std::vector<double>, 2> {
    std::printf("Trying to read file: %s\n", fileName.string().c_str());
    auto in = std::ifstream(fileName, std::ios::in | std::ios::binary);
    if (!in.is_open()) throw std::runtime_error("Can't open input file");

    Header h{};
    read_record(in, h);
    const int total = h.nbodies;
    const int ndark = h.ndark;

    std::vector<double> positions, velocities;
    positions.reserve(4u * static_cast<std::size_t>(total > 0 ? total : 0));
    velocities.reserve(positions.capacity());
    Dark d{};
    Star s{};
    for (int i = 0; i < total; ++i) {
        if (i < ndark) {
            read_record(in, d);
            append(d, positions, velocities);
        } else {
            read_record(in, s);
            append(s, positions, velocities);
        }
    }
    // round up to a multiple of 256 bodies with zero-mass padding (tipsy.cpp:111-119)
    int padded = total;
    if (total % 256) padded = ((total / 256) + 1) * 256;
    positions.insert(positions.end(), 4u * static_cast<std::size_t>(padded - total), 0.0);
    velocities.insert(velocities.end(), 4u * static_cast<std::size_t>(padded - total), 0.0);

    std::printf("Read %d bodies\n", padded);
    return {std::move(positions), std::move(velocities)};
}

auto write_tipsy_file(const std::filesystem::path& fileName, std::span<const double> positions, std::span<const double> velocities, int ndark) -> void {
    if (positions.size() % 4 || positions.size() != velocities.size()) throw std::invalid_argument("write_tipsy_file: need interleaved 4-vectors");
    const int total = static_cast<int>(positions.size() / 4);
    if (ndark < 0 || ndark > total) throw std::invalid_argument("write_tipsy_file: bad ndark");
    auto out = std::ofstream(fileName, std::ios::out | std::ios::binary | std::ios::trunc);
    if (!out.is_open()) throw std::runtime_error("Can't open output file");
    Header h{};
    h.time = 0.0, h.nbodies = total, h.ndim = 3, h.nsph = 0, h.ndark = ndark, h.nstar = total - ndark;
    write_record(out, h);
    for (int i = 0; i < total; ++i) {
        const auto* p = &positions[4u * static_cast<std::size_t>(i)];
        const auto* v = &velocities[4u * static_cast<std::size_t>(i)];
        if (i < ndark) {
            Dark d{};
            d.mass = static_cast<float>(p[3]);
            for (int k = 0; k < 3; ++k) d.pos[k] = static_cast<float>(p[k]), d.vel[k] = static_cast<float>(v[k]);
            d.eps = static_cast<float>(v[3]);
            d.phi = i;
            write_record(out, d);
        } else {
            Star s{};
            s.mass = static_cast<float>(p[3]);
            for (int k = 0; k < 3; ++k) s.pos[k] = static_cast<float>(p[k]), s.vel[k] = static_cast<float>(v[k]);
            s.eps = static_cast<float>(v[3]);
            s.phi = i;
            write_record(out, s);
        }
    }
}
