// text.hpp -- the little formatting this host needs.  The reference prints with C++23 <print>/<format>, which
// libstdc++ 11 (this image) lacks; shortest() reproduces std::format's default "{}" rendering of a float
// (shortest round-trip digits, std::to_chars) so the benchmark lines read the same.
#pragma once

#include <charconv>
#include <string>
#include <system_error>

namespace text {

template <typename F> inline auto shortest(F value) -> std::string {
    char buf[64];
    const auto [end, ec] = std::to_chars(buf, buf + sizeof(buf), value);
    return ec == std::errc{} ? std::string(buf, end) : std::string("?");
}

// "{:3}" of the reference's benchmark lines: minimum width 3, right-aligned for arithmetic types
template <typename F> inline auto width3(F value) -> std::string {
    auto s = shortest(value);
    if (s.size() < 3) s.insert(0, 3 - s.size(), ' ');
    return s;
}

}  // namespace text
