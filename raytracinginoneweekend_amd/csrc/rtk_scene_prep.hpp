// rtk_scene_prep.hpp — host-side preparation of the flattened scene for the kernels (plain C++; shared by
// rtk_set_scene in rtk.hip and by the CPU emulation harness under tests/cpu_emul).
#pragma once

#include "rtk_device.hpp"

#include "../../include/rtk.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace rtk {

struct PreparedScene {
    std::vector<float4> filt, exact, m0, m1;
    FilterBounds fb;
    int n = 0, npad = 0;
};

// Precomputes, bit-identically to the reference's own fp32 operations, radius*radius (Sphere3.hpp:35) and
// 1/radius (Vector3.cpp:35-37 via Sphere3.hpp:54); classifies the spheres the filter may reject and bounds them.
inline void prepare_scene(const rtk_scene* s, PreparedScene& out) {
    const int n = s->n, npad = (n + 31) / 32 * 32;
    out.n = n;
    out.npad = npad;
    std::vector<float4>&filt = out.filt, &exact = out.exact, &m0 = out.m0, &m1 = out.m1;
    filt.assign(npad, make_float4(0.f, 0.f, 0.f, 0.f));
    exact.assign(npad, make_float4(0.f, 0.f, 0.f, 0.f));
    m0.assign(n, make_float4(0.f, 0.f, 0.f, 0.f));
    m1.assign(n, make_float4(0.f, 0.f, 0.f, 0.f));
    const float inf = std::numeric_limits<float>::infinity();
    // Which spheres the filter may reject ("filterable"): finite, moderate magnitudes, and not an outlier.
    // Outliers (the 1000-unit ground sphere of the stock scene; specks) would blow up the scene bounds the
    // projection filter scales its margin with, so up to 8 at either end are simply always candidates.
    std::vector<char> filterable(n, 0);
    std::vector<int> order;
    for (int i = 0; i < n; ++i) {
        const float big = 0x1p30f, ar = std::fabs(s->radius[i]);
        if (std::fabs(s->cx[i]) < big && std::fabs(s->cy[i]) < big && std::fabs(s->cz[i]) < big && ar > 0x1p-20f && ar < big) {
            filterable[i] = 1;
            order.push_back(i);
        }
    }
    auto extent = [&](int i) { return std::sqrt(s->cx[i] * s->cx[i] + s->cy[i] * s->cy[i] + s->cz[i] * s->cz[i]) + std::fabs(s->radius[i]); };
    for (int pass = 0; pass < 8 && order.size() > 1; ++pass) {     // peel far/large outliers while each halves the bound
        std::sort(order.begin(), order.end(), [&](int x, int y) { return extent(x) > extent(y); });
        if (extent(order[0]) > 2.0f * extent(order[1])) { filterable[order[0]] = 0; order.erase(order.begin()); } else break;
    }
    for (int pass = 0; pass < 8 && order.size() > 1; ++pass) {     // peel specks while each doubles r_min
        std::sort(order.begin(), order.end(), [&](int x, int y) { return std::fabs(s->radius[x]) < std::fabs(s->radius[y]); });
        if (std::fabs(s->radius[order[0]]) * 2.0f < std::fabs(s->radius[order[1]])) { filterable[order[0]] = 0; order.erase(order.begin()); } else break;
    }
    FilterBounds fb = {0.0f, 0.0f, 1.0f};
    if (!order.empty()) {
        fb.r_min = inf;
        for (int i : order) {
            fb.c_max = std::fmax(fb.c_max, extent(i) * (1.0f + 0x1p-20f));
            fb.r_max2 = std::fmax(fb.r_max2, s->radius[i] * s->radius[i] * (1.0f + 0x1p-20f));
            fb.r_min = std::fmin(fb.r_min, std::fabs(s->radius[i]) * (1.0f - 0x1p-20f));
        }
    }
    for (int i = 0; i < npad; ++i) {
        if (i >= n) {
            // pad (never read by the filter loops; kept inert): r^2 term of -inf can never be a candidate
            filt[i] = make_float4(0.f, 0.f, 0.f, -inf);
            exact[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const float cx = s->cx[i], cy = s->cy[i], cz = s->cz[i], r = s->radius[i];
        volatile float rr_v = r * r;                 // fl(radius*radius), Sphere3.hpp:35 (host compiler must not widen)
        const float rr = rr_v;
        exact[i] = make_float4(cx, cy, cz, rr);
        // spheres the filter must not reject are candidates for every ray: centre 0, r^2 term +inf
        filt[i] = filterable[i] ? make_float4(cx, cy, cz, rr * kFilterScaleSq) : make_float4(0.f, 0.f, 0.f, inf);
        m0[i] = make_float4(s->mat_r[i], s->mat_g[i], s->mat_b[i], s->roughness[i]);
        int32_t ty = s->mat_type[i];
        float tyf;
        std::memcpy(&tyf, &ty, 4);
        volatile float inv_v = 1.0f / r;             // fl(1/radius), Vector3.cpp:36
        m1[i] = make_float4(s->metallic[i], s->ior[i], tyf, inv_v);
    }
    out.fb = fb;
}

} // namespace rtk
