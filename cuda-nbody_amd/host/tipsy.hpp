// tipsy.hpp -- reader (and, for fixtures, writer) of the tipsy variant the reference loads with --tipsy=<file>
// (/root/reference/src/nbody/tipsy.{hpp,cpp}).
#pragma once

#include <array>
#include <filesystem>
#include <span>
#include <vector>

// {positions[4N'], velocities[4N']}: pos = {x,y,z,mass}, vel = {vx,vy,vz,eps}; N' = N rounded up to a multiple of
// 256 with zero-mass, zero-everything bodies appended (tipsy.cpp:111-119)
auto read_tipsy_file(const std::filesystem::path& fileName) -> std::array<std::vector<double>, 2>;

// Writes `ndark` dark particles followed by stars in the same on-disk layout (the reference has no writer and
// ships no sample file; tests need one).  pos/vel are interleaved 4-vectors as returned by read_tipsy_file.
auto write_tipsy_file(const std::filesystem::path& fileName, std::span<const double> positions, std::span<const double> velocities, int ndark) -> void;
