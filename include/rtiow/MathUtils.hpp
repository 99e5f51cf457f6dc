// rtiow/MathUtils.hpp — host-side constants and the two sequential RNG streams of the reference's
// scene API (reference: MathUtils.hpp:8-25).  Used only for scene construction on the host; the
// GPU path draws from a counter-based generator instead (rtk_device.hpp).
#pragma once

#include <cmath>
#include <limits>
#include <random>

inline constexpr float infinity = std::numeric_limits<float>::infinity();
inline constexpr float pi = 3.14159265358979323846f;

inline float degrees_to_radians(float degrees) { return degrees * pi / 180.0f; }

namespace rtiow_detail {
// One default-seeded engine per overload, as in the reference (two independent, identically seeded
// mt19937 streams: G0 behind random_float(), G1 behind random_float(min, max)).
inline std::mt19937& stream0() {
    static std::mt19937 engine;
    return engine;
}
inline std::mt19937& stream1() {
    static std::mt19937 engine;
    return engine;
}
} // namespace rtiow_detail

// Uniform in [0, 1) from stream G0.
inline float random_float() { return std::uniform_real_distribution<float>(0.0f, 1.0f)(rtiow_detail::stream0()); }

// Uniform in [min, exclusive_max) from stream G1.
inline float random_float(float min, float exclusive_max) {
    return std::uniform_real_distribution<float>(min, exclusive_max)(rtiow_detail::stream1());
}
