// rtiow/Color.hpp — pixel quantiser and PPM pixel writers (reference: Color.hpp:9-48).
#pragma once

#include "Vector3.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iostream>

namespace rtiow_detail {
// sqrt(sum / spp) clamped to [0, 1]: gamma 2.
inline float gamma2(float sum, int samples_per_pixel) {
    const float scale = 1.0f / samples_per_pixel;
    return std::clamp(std::sqrt(scale * sum), 0.0f, 1.0f);
}
} // namespace rtiow_detail

// ASCII (P3) pixel.
inline void write_color(std::ostream& out, Color pixel_color, int samples_per_pixel) {
    for (int k = 0; k < 3; ++k)
        out << static_cast<int>(255 * rtiow_detail::gamma2(pixel_color[k], samples_per_pixel)) << (k == 2 ? '\n' : ' ');
}

// Binary (P6) pixel: three bytes R, G, B, truncating 255*x.
inline void write_color_binary(std::ostream& out, Color pixel_color, int samples_per_pixel) {
    char rgb[3];
    for (int k = 0; k < 3; ++k) rgb[k] = static_cast<char>(static_cast<uint8_t>(255 * rtiow_detail::gamma2(pixel_color[k], samples_per_pixel)));
    out.write(rgb, 3);
}
