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

// The partition of a pixel's `spp` samples into runs (a lane sums one run in fp32, then adds it to the pixel's
// integer accumulator).  It depends on spp and the image size ONLY, never on the GPU count or the scheduling, so
// images stay bit-identical across ranks.  run_len > 0: equal runs of that length.  run_len == 0 (default):
// decreasing runs, L = ceil(remaining / k): units are large while there is plenty of work and shrink towards
// the end of the frame, so that no lane is still inside a long unit when the packet queue runs dry (the tail
// sets the kernel's end; hot pixels — glass interiors — cost up to 7x the mean per sample).  k grows as the
// frame shrinks (fewer pixels per lane -> the tail matters more): k = clamp(66.7e6 / (W*H), 8, 64), sized for
// the frame being split over up to 8 GPUs (tools/sim_schedule.py).
inline std::vector<uint2> run_schedule(int spp, int width, int height, int run_len) {
    std::vector<uint2> runs;
    if (spp <= 0) return runs;
    if (run_len > 0) {
        for (int s = 0; s < spp; s += run_len) runs.push_back(make_uint2(static_cast<uint32_t>(s), static_cast<uint32_t>(std::min(run_len, spp - s))));
        return runs;
    }
    const double px = static_cast<double>(width) * static_cast<double>(height);
    const int k = static_cast<int>(std::min(64.0, std::max(8.0, std::ceil(66.7e6 / std::max(1.0, px)))));
    int s = 0;
    while (s < spp) {
        const int rem = spp - s;
        const int len = (rem + k - 1) / k;          // >= 1
        runs.push_back(make_uint2(static_cast<uint32_t>(s), static_cast<uint32_t>(len)));
        s += len;
    }
    return runs;
}

} // namespace rtk
