// rtk_f16.hpp — IEEE binary16 <-> binary32 in portable host C++ (round to nearest even, subnormals kept: the mode
// the device's packed-f16 instructions run in under hipcc's defaults).  Used by the host-side scene preparation of
// the packed-f16 filter (RTK_FILTER_FORM == 2) and by the CPU build of the device code (tests/cpu_emul).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace rtk_f16 {

inline uint16_t from_float(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint16_t sign = static_cast<uint16_t>((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x > 0x7F800000u) return static_cast<uint16_t>(sign | 0x7E00u);          // NaN
    if (x >= 0x477FF000u) return static_cast<uint16_t>(sign | 0x7C00u);         // >= 65520 rounds to infinity
    if (x < 0x38800000u) {                                                      // below 2^-14: subnormal, spacing 2^-24
        float a;
        std::memcpy(&a, &x, 4);
        const float r = std::nearbyint(a * 16777216.0f);                        // exact scaling, then ties-to-even
        return static_cast<uint16_t>(sign | static_cast<uint16_t>(r));          // 1024 is the smallest normal: still right
    }
    const uint32_t mant = x & 0x7FFFFFu, exp = (x >> 23) - 127u + 15u;
    uint32_t h = (exp << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                     // a carry runs into the exponent correctly
    return static_cast<uint16_t>(sign | h);
}

inline float to_float(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu, mant = h & 0x3FFu;
    float f;
    if (exp == 0) {
        f = static_cast<float>(mant) * 0x1p-24f;
    } else if (exp == 31) {
        const uint32_t u = 0x7F800000u | (mant << 13);
        std::memcpy(&f, &u, 4);
    } else {
        const uint32_t u = ((exp - 15u + 127u) << 23) | (mant << 13);
        std::memcpy(&f, &u, 4);
    }
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u |= sign;
    std::memcpy(&f, &u, 4);
    return f;
}

inline float round(float f) { return to_float(from_float(f)); }

} // namespace rtk_f16
