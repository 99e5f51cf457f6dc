// rtk_scene_prep.hpp — host-side preparation of the flattened scene for the kernels (plain C++; shared by
// rtk_set_scene in rtk.hip and by the CPU emulation harness under tests/cpu_emul).
#pragma once

#include "rtk_device.hpp"
#include "rtk_intersect.hpp"

#include "../../include/rtk.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace rtk {

struct PreparedScene {
    std::vector<float4> filt, exact, m0, m1;
    std::vector<float4> wdesc;       // matrix-pipe layout only (SceneDev::wdesc); empty = streamed layout
    std::vector<float4> ltiles, lwdesc;   // streamed layout: word-local tiles and descriptors (SceneDev::ltiles, lwdesc)
    int local_words = 0;             // words of the local cut (class words and kind-0 words)
    int local_class_words = 0;       // those that have a local tile
    bool use_local = false;          // nearly all full words have one: the streamed walk filters on the matrix pipe
    int class_words = 0;
    int filt_lds = 0;                // SceneDev::filt_lds
    std::vector<char> filterable;    // per sphere: the filter may reject it (finite, moderate, not an outlier)
    FilterBounds fb;
    int n = 0, npad = 0;
};

// Precomputes, bit-identically to the reference's own fp32 operations, radius*radius (Sphere3.hpp:35) and
// 1/radius (Vector3.cpp:35-37 via Sphere3.hpp:54); classifies the spheres the filter may reject and bounds them.
// `compact`: the layout of LDS-resident scenes, whose projections run on the matrix pipe (SceneDev); without it the
// streamed layout of larger scenes.
inline void prepare_scene(const rtk_scene* s, PreparedScene& out, bool compact = false) {
    const int n = s->n, npad = (n + 31) / 32 * 32;
    out.n = n;
    out.npad = npad;
    std::vector<float4>&filt = out.filt, &exact = out.exact, &m0 = out.m0, &m1 = out.m1;
    filt.assign(npad + 8, make_float4(0.f, 0.f, 0.f, -std::numeric_limits<float>::infinity()));      // 8 inert entries past the end: kind-0 words of the local walk read whole batches of eight
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
    // The r^2 term of sphere i as the filter uses it, before scaling: fl(r*r) widened by (1+k).
    auto widened_r2 = [&](int i) {
        volatile float rr_v = s->radius[i] * s->radius[i];
        return static_cast<float>(rr_v) * kFilterScaleSq;
    };
    FilterBounds fb = {0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f, kQuadEpsClass, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
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
    out.filterable = filterable;
    out.wdesc.clear();
    out.class_words = 0;
    out.ltiles.clear();
    out.lwdesc.clear();
    out.local_class_words = 0;
    out.use_local = false;
    if (!compact) {
        // Word-local frames (rtk_device.hpp local_term).  The list is cut into words of AT MOST 32 consecutive entries: a class
        // word grows while its filterable spheres share cy and the widened r^2 bit for bit and stay within kLocalReach radii
        // of the middle of their x and z ranges (so that sigma_w r stays large enough for the margin: a word that wrapped from
        // the end of one row of random_scene() to the start of the next would be ten times wider than a row's word); what is
        // left between class words becomes kind-0 words, filtered on the vector ALUs.
        auto same_bits = [](float x, float y) { return std::memcmp(&x, &y, 4) == 0; };
        struct Span { int start, count; bool cls; float cy, r2, cxw, czw; double reach; };
        // the class word that would start at i: how many entries, and its frame
        auto class_run = [&](int i, Span& sp) {
            sp = {i, 0, false, 0.f, 0.f, 0.f, 0.f, 0.0};
            bool have = false;
            float lox = inf, hix = -inf, loz = inf, hiz = -inf;
            int members = 0;
            for (int j = i; j < n && j < i + 32; ++j) {
                if (filterable[j]) {
                    const float r2j = widened_r2(j);
                    if (have && !(same_bits(s->cy[j], sp.cy) && same_bits(r2j, sp.r2))) break;
                    const float nlox = std::fmin(lox, s->cx[j]), nhix = std::fmax(hix, s->cx[j]), nloz = std::fmin(loz, s->cz[j]), nhiz = std::fmax(hiz, s->cz[j]);
                    const double half = 0.5 * std::max(static_cast<double>(nhix) - nlox, static_cast<double>(nhiz) - nloz);
                    const double rj = std::sqrt(static_cast<double>(r2j));
                    if (!(half <= kLocalReach * rj)) break;          // also ends the word at a non-finite extent
                    if (!have) { sp.cy = s->cy[j]; sp.r2 = r2j; have = true; }
                    lox = nlox; hix = nhix; loz = nloz; hiz = nhiz;
                    ++members;
                }
                ++sp.count;
            }
            if (!have) { sp.count = 0; return; }
            sp.cxw = 0.5f * lox + 0.5f * hix;
            sp.czw = 0.5f * loz + 0.5f * hiz;
            double e = 0.0;
            for (int j = i; j < i + sp.count; ++j)
                if (filterable[j]) e = std::max({e, std::fabs(static_cast<double>(s->cx[j]) - sp.cxw), std::fabs(static_cast<double>(s->cz[j]) - sp.czw)});
            sp.reach = std::max(e, std::sqrt(static_cast<double>(sp.r2)));
            sp.cls = members >= 1 && std::isfinite(sp.cxw) && std::isfinite(sp.czw) && std::isfinite(sp.reach) && sp.reach > 0.0;
        };
        std::vector<Span> spans;
        int covered = 0;
        for (int i = 0; i < n;) {
            Span sp;
            class_run(i, sp);
            if (sp.cls && (sp.count > kLocalShortRun || i + sp.count == n)) {          // a class word costs ~500 SIMD cycles whatever it holds; a few spheres are cheaper on the vector ALUs
                spans.push_back(sp);
                covered += sp.count;
                i += sp.count;
                continue;
            }
            // no worthwhile class word starts here: this entry joins a kind-0 word
            if (spans.empty() || spans.back().cls || spans.back().count == 32) spans.push_back({i, 0, false, 0.f, 0.f, 0.f, 0.f, 0.0});
            ++spans.back().count;
            ++i;
        }
        const int words = static_cast<int>(spans.size());
        out.ltiles.assign(static_cast<size_t>(words) * 64, make_float4(0.f, 0.f, 0.f, 0.f));
        out.lwdesc.assign(static_cast<size_t>(words) * 3, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int w = 0; w < words; ++w) {
            const Span& sp = spans[w];
            const int base = sp.start;
            uint32_t always = 0, valid = 0;
            for (int i = base; i < base + sp.count; ++i) {
                valid |= 0x80000000u >> (i - base);
                if (!filterable[i]) always |= 0x80000000u >> (i - base);
            }
            int32_t kind = 0;
            float sigw = 1.f, k0 = 0.f;
            float4 scaled = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sp.cls) {
                int ex = static_cast<int>(std::floor(std::log2(kLocalB / sp.reach)));
                while (std::ldexp(sp.reach, ex) > kLocalB) --ex;
                while (std::ldexp(sp.reach, ex + 1) <= kLocalB) ++ex;
                if (ex >= -20 && ex <= 20) {
                    const double sg = std::ldexp(1.0, ex);
                    const double r2s = sg * sg * static_cast<double>(sp.r2);
                    sigw = static_cast<float>(sg);
                    out.fb.lsig_max = std::fmax(out.fb.lsig_max, sigw);
                    // The kernel works with the UNSCALED v = o - c_w and folds sigma_w (a power of two: exact) and the fixed-point
                    // scales into per-word constants:  kc = -2^48 2^27 K = (|p|^2 + ratio |v|^2) s1 + k0,
                    //   s1 = -2^75 (1 - A1) sigma^2,  k0 = +2^75 (r~'^2 + A0_w),  ratio = -A2 / (1 - A1)   (rtk_device.hpp kLocalRatio)
                    const double two75 = std::ldexp(1.0, 75);
                    k0 = std::nextafter(static_cast<float>(two75 * (r2s + local_margin_const(r2s)) * (1.0 + 1e-6)), inf);     // rounded up: more candidates
                    scaled = make_float4(static_cast<float>(-two75 * (1.0 - static_cast<double>(kLocalA1)) * sg * sg),      // s1
                                         static_cast<float>(-2.0 * sg * kQuadRayScale),                                      // linear coefficient per unit of p
                                         static_cast<float>(sg * sg), 0.f);
                    kind = 1;
                    uint16_t rows[32][16];
                    for (int rho = 0; rho < 32; ++rho) {
                        const int i = base + tile_row_sphere(rho);
                        const bool live = i < base + sp.count && filterable[i] != 0;
                        uint16_t parts[kQuadClassMonomials][2] = {};
                        if (live) {
                            const double x = sg * (static_cast<double>(s->cx[i]) - sp.cxw), z = sg * (static_cast<double>(s->cz[i]) - sp.czw);
                            const double mono_c[kQuadClassMonomials] = {x * x, z * z, x * z, x, z};
                            for (int m = 0; m < kQuadClassMonomials; ++m) {
                                const double v = mono_c[m] * kQuadSphereScale;
                                parts[m][0] = f16_rne(static_cast<float>(v));
                                parts[m][1] = f16_rne(static_cast<float>(v - static_cast<double>(f16_value(parts[m][0]))));
                            }
                        }
                        for (int k = 0; k < 16; ++k) {
                            int mono, spart, rp;
                            local_term(k, mono, spart, rp);
                            rows[rho][k] = (live && mono >= 0) ? parts[mono][spart] : static_cast<uint16_t>(0);
                        }
                    }
                    for (int l = 0; l < 64; ++l) {          // lane l: row l & 31, terms 8 (l >> 5) ... + 7
                        uint16_t frag[8];
                        for (int j = 0; j < 8; ++j) frag[j] = rows[l & 31][8 * (l >> 5) + j];
                        std::memcpy(&out.ltiles[static_cast<size_t>(w) * 64 + l], frag, 16);
                    }
                    ++out.local_class_words;
                } else {
                    covered -= sp.count;
                }
            }
            const int32_t start_kind = base * 2 + kind;          // the word's entries: the set bits of `valid` (a run from the top bit)
            float sk_f, always_f, valid_f;
            std::memcpy(&sk_f, &start_kind, 4);
            std::memcpy(&always_f, &always, 4);
            std::memcpy(&valid_f, &valid, 4);
            out.lwdesc[3 * static_cast<size_t>(w)] = make_float4(sp.cxw, sp.cy, sp.czw, 2.0f / sigw);          // |p| is clamped to 2 scaled units
            out.lwdesc[3 * static_cast<size_t>(w) + 1] = make_float4(k0, sk_f, always_f, valid_f);
            out.lwdesc[3 * static_cast<size_t>(w) + 2] = scaled;
        }
        out.local_words = words;
        // worth it when nearly every sphere sits in a class word (a kind-0 word costs a full vector-ALU word fed by scalar loads)
        // and the words are reasonably full (a word costs the same whether it holds 32 spheres or one)
        out.use_local = out.local_class_words > 0 && static_cast<double>(covered) >= 0.9 * n && words <= 2 * ((n + 31) / 32);
    }
    if (compact) {
        // Matrix-pipe layout (SceneDev): per word a tile of f16 monomial parts as lane fragments and a descriptor.
        auto same_bits = [](float x, float y) { return std::memcmp(&x, &y, 4) == 0; };
        const int words = npad / 32, rem = n & 31;
        const bool short_tail = rem != 0 && rem <= kShortTail;
        const int tile_words = short_tail ? words - 1 : words;
        // sigma: the power of two that brings every filterable centre (plus radius) into the unit ball
        int ec = 0;
        if (!order.empty()) ec = std::max(-60, std::min(60, static_cast<int>(std::ceil(std::log2(static_cast<double>(fb.c_max))))));
        while (std::ldexp(static_cast<double>(fb.c_max), -ec) > 1.0) ++ec;
        const double sigma = std::ldexp(1.0, -ec);
        fb.sigma = static_cast<float>(sigma);
        // The scene's class: the (cy, r~^2) pair shared by the filterable spheres of the most words.  A word qualifies
        // when it has no pads and all its filterable spheres (at least one) agree in both, bit for bit.
        struct Key { float cy, r2; int count; };
        std::vector<Key> keys;
        std::vector<int> word_key(words, -1);
        for (int w = 0; w < tile_words; ++w) {
            const int base = w * 32;
            if (base + 32 > n) continue;
            bool ok = true, have = false;
            float cy = 0.0f, r2 = 0.0f;
            for (int i = base; ok && i < base + 32; ++i)
                if (filterable[i]) {
                    if (!have) { cy = s->cy[i]; r2 = widened_r2(i); have = true; }
                    else ok = same_bits(s->cy[i], cy) && same_bits(widened_r2(i), r2);
                }
            if (!ok || !have) continue;
            int k = -1;
            for (size_t q = 0; q < keys.size(); ++q) if (same_bits(keys[q].cy, cy) && same_bits(keys[q].r2, r2)) k = static_cast<int>(q);
            if (k < 0) { keys.push_back({cy, r2, 0}); k = static_cast<int>(keys.size()) - 1; }
            ++keys[k].count;
            word_key[w] = k;
        }
        int cls_key = -1;
        for (size_t q = 0; q < keys.size(); ++q) if (cls_key < 0 || keys[q].count > keys[cls_key].count) cls_key = static_cast<int>(q);
        if (cls_key >= 0) {
            fb.cls_cy = static_cast<float>(sigma * keys[cls_key].cy);
            fb.cls_r2 = static_cast<float>(sigma * sigma * keys[cls_key].r2);
        }
        std::vector<float4> tiles;
        // one 16-term fragment set: rows[rho][k], k < 16 -> 64 lane fragments (lane l: row l & 31, terms 8 (l >> 5) ... + 7)
        auto push_fragments = [&](const uint16_t (*rows)[32], int k0) {
            for (int l = 0; l < 64; ++l) {
                uint16_t frag[8];
                for (int j = 0; j < 8; ++j) frag[j] = rows[l & 31][k0 + 8 * (l >> 5) + j];
                float4 v;
                std::memcpy(&v, frag, 16);
                tiles.push_back(v);
            }
        };
        bool any_general = false;
        for (int w = 0; w < words; ++w) {
            const int base = w * 32;
            uint32_t always = 0, valid = 0;
            for (int i = base; i < base + 32 && i < n; ++i) {
                valid |= 0x80000000u >> (i - base);
                if (!filterable[i]) always |= 0x80000000u >> (i - base);
            }
            const int32_t off = static_cast<int32_t>(tiles.size());
            int32_t kind;
            if (w >= tile_words) {           // short tail: plain records behind the tiles, tested on the vector ALUs
                kind = 2;
                for (int i = base; i < n; ++i) {             // three float4 per sphere: fp32 monomials of the scaled centre, r'^2
                    if (!filterable[i]) {                    // never rejected: r'^2 = +inf drives the sum to -inf
                        tiles.push_back(make_float4(0.f, 0.f, 0.f, 0.f)); tiles.push_back(make_float4(0.f, 0.f, 0.f, 0.f)); tiles.push_back(make_float4(0.f, inf, 0.f, 0.f));
                        continue;
                    }
                    const double x = sigma * s->cx[i], y = sigma * s->cy[i], z = sigma * s->cz[i];
                    const float r2 = static_cast<float>(sigma * sigma * widened_r2(i)) * (1.0f + 0x1p-20f);     // rounded up
                    tiles.push_back(make_float4(static_cast<float>(x * x), static_cast<float>(y * y), static_cast<float>(z * z), static_cast<float>(x * y)));
                    tiles.push_back(make_float4(static_cast<float>(x * z), static_cast<float>(y * z), static_cast<float>(x), static_cast<float>(y)));
                    tiles.push_back(make_float4(static_cast<float>(z), r2, 0.f, 0.f));
                }
            } else {
                const bool cls = word_key[w] >= 0 && word_key[w] == cls_key;
                kind = cls ? 1 : 0;
                const int n_mono = cls ? kQuadClassMonomials : kQuadGeneralMonomials, kk = cls ? 16 : 32;
                // the tile: row rho = sphere tile_row_sphere(rho); pads and always-candidates are zero rows
                uint16_t rows[32][32];
                for (int rho = 0; rho < 32; ++rho) {
                    const int i = base + tile_row_sphere(rho);
                    const bool live = i < n && filterable[i];
                    uint16_t parts[kQuadGeneralMonomials + 1][2] = {};
                    if (live) {
                        const double x = sigma * s->cx[i], y = sigma * s->cy[i], z = sigma * s->cz[i];
                        const double mono_c[kQuadClassMonomials] = {x * x, z * z, x * z, x, z};
                        const double mono_g[kQuadGeneralMonomials + 1] = {x * x, y * y, z * z, x * y, x * z, y * z, x, y, z, sigma * sigma * widened_r2(i)};
                        for (int m = 0; m < (cls ? n_mono : n_mono + 1); ++m) {
                            const double v = (cls ? mono_c[m] : mono_g[m]) * kQuadSphereScale;
                            parts[m][0] = f16_rne(static_cast<float>(v));
                            parts[m][1] = f16_rne(static_cast<float>(v - static_cast<double>(f16_value(parts[m][0]))));
                        }
                    }
                    for (int k = 0; k < 32; ++k) {
                        int mono, sp, rp;
                        quad_term(k, n_mono, mono, sp, rp);
                        rows[rho][k] = (k < kk && live && mono >= 0) ? parts[mono][sp] : static_cast<uint16_t>(0);
                    }
                }
                push_fragments(rows, 0);
                if (!cls) { push_fragments(rows, 16); any_general = true; } else ++out.class_words;
            }
            float off_f, kind_f, always_f, valid_f;
            std::memcpy(&off_f, &off, 4);
            std::memcpy(&kind_f, &kind, 4);
            std::memcpy(&always_f, &always, 4);
            std::memcpy(&valid_f, &valid, 4);
            out.wdesc.push_back(make_float4(off_f, kind_f, always_f, valid_f));
        }
        fb.kinds = (out.class_words ? 1 : 0) | (any_general ? 2 : 0);
        fb.acc_eps = any_general ? kQuadEpsGeneral : kQuadEpsClass;
        {
            const double cpd = sigma * fb.c_max, rp2d = sigma * sigma * fb.r_max2, up = 1.0 + 1e-6;       // rounded up: they widen the margin
            fb.cp = static_cast<float>(cpd * up);
            fb.cp_twice = static_cast<float>(2.0 * cpd * up);
            fb.rp2 = static_cast<float>(rp2d * up);
            fb.t_quad = static_cast<float>((1.42 * cpd * cpd + rp2d) * up);
            fb.cls_cy2 = fb.cls_cy * fb.cls_cy;
        }
        out.fb = fb;
        out.filt_lds = static_cast<int>(tiles.size());
        filt = tiles;
    }
}

// ---- uniform grid (GridDev in rtk_device.hpp) -----------------------------------------------------
struct GridHost {
    bool usable = false;
    const char* why = "";            // when not usable
    GridDev dev = {};                // pointers are filled in by whoever uploads the arrays
    std::vector<uint32_t> cell_start, items, always;
    double mean_items_per_cell = 0.0, mean_cells_per_sphere = 0.0;
};

constexpr int kGridMaxAlways = 64;          // spheres tested for every ray; more than this: the grid is refused
constexpr int kGridMaxCellsPerSphere = 4096;

// Registers every gridable sphere in the cells its inflated box overlaps, once per origin zone.
// Inflation: the reference accepts a sphere (and returns a root whose point lies) up to sqrt(r^2 + E) from the centre,
// E = 32u(|oc|^2 + r^2), u = 2^-24, |oc| the distance from the ray's ORIGIN to the centre: bounded per
// zone of origins; origins are admitted within the ball (centre, r_allow = 0.75 D), D the diagonal of the gridable
// spheres' bounding box.  On top of that, `slack` = 2^-14 D covers the float error of the device's cell walk
// (positions are accurate to ~2^-22 D).
inline void build_grid(const rtk_scene* s, const PreparedScene& ps, GridHost& g) {
    g = GridHost();
    const int n = s->n;
    std::vector<int> in;
    // The grid visits spheres in another order than the list; the reference's result is order-independent only while
    // every root is a finite number (a NaN root makes `closest` NaN, after which HittableList.hpp:34-38 accepts
    // whatever comes next in LIST order).  With all centres and radii finite and below 2^30, and rays as the device
    // code admits them (|o| < 2^30, |d|^2 in (2^-80, 2^60)), no intermediate of Sphere3::hit overflows.
    for (int i = 0; i < n; ++i) {
        const float big = 0x1p30f;
        if (!(std::fabs(s->cx[i]) < big && std::fabs(s->cy[i]) < big && std::fabs(s->cz[i]) < big && std::fabs(s->radius[i]) < big)) {
            g.why = "a sphere is non-finite or beyond 2^30: results would depend on the list order";
            return;
        }
    }
    // Gridable: everything except up to 8 spheres that would blow up the grid's box (the 1000-unit ground sphere of
    // the stock scene): largest reach from the median centre, peeled while each is more than twice the next.
    (void)ps;
    {
        std::vector<float> xs(s->cx, s->cx + n), ys(s->cy, s->cy + n), zs(s->cz, s->cz + n);
        auto median = [](std::vector<float>& v) { std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return static_cast<double>(v[v.size() / 2]); };
        const double mx = median(xs), my = median(ys), mz = median(zs);
        auto reach = [&](int i) {
            const double dx = s->cx[i] - mx, dy = s->cy[i] - my, dz = s->cz[i] - mz;
            return std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs(static_cast<double>(s->radius[i]));
        };
        in.resize(n);
        for (int i = 0; i < n; ++i) in[i] = i;
        for (int pass = 0; pass < 8 && in.size() > 1; ++pass) {
            std::partial_sort(in.begin(), in.begin() + 2, in.end(), [&](int x, int y) { return reach(x) > reach(y); });
            if (reach(in[0]) > 2.0 * reach(in[1])) { g.always.push_back(static_cast<uint32_t>(in[0])); in.erase(in.begin()); } else break;
        }
        std::sort(in.begin(), in.end());
    }
    if (in.empty()) { g.why = "no sphere can be put in a grid"; return; }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, r_max = 0.0;
    for (int i : in) {
        const double c[3] = {s->cx[i], s->cy[i], s->cz[i]}, r = std::fabs(static_cast<double>(s->radius[i]));
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], c[k] - r); hi[k] = std::max(hi[k], c[k] + r); }
        r_max = std::max(r_max, r);
    }
    const double D = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    const double centre[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    const double r_allow = 0.75 * D, oc_max = r_allow + 0.5 * D + r_max;
    const double u = std::ldexp(1.0, -24), slack = std::ldexp(D, -14);
    auto inflated = [&](int i, double oc) {          // half-size of sphere i's box for ray origins at most `oc` from its centre
        const double r = std::fabs(static_cast<double>(s->radius[i]));
        return (std::sqrt(r * r + 32.0 * u * (oc * oc + r * r)) + 2.0 * slack) * (1.0 + 1e-6);
    };
    // the grid box holds every sphere box at its largest inflation (the farthest admissible origin)
    double glo[3], ghi[3];
    for (int k = 0; k < 3; ++k) { glo[k] = 1e300; ghi[k] = -1e300; }
    std::vector<double> near_box;
    for (int i : in) {
        const double c[3] = {s->cx[i], s->cy[i], s->cz[i]}, rb = inflated(i, oc_max);
        for (int k = 0; k < 3; ++k) { glo[k] = std::min(glo[k], c[k] - rb); ghi[k] = std::max(ghi[k], c[k] + rb); }
        near_box.push_back(inflated(i, D / 6.0));
    }
    std::nth_element(near_box.begin(), near_box.begin() + near_box.size() / 2, near_box.end());
    double h0 = 2.0 * near_box[near_box.size() / 2];       // cells of about twice the median box seen from nearby
    // Origin zones: the inflation a sphere needs grows with the distance of the ray's ORIGIN (the reference's rounding
    // noise is ~|o - C|^2), so the space of origins is cut into zn[0] x zn[1] x zn[2] zones and every zone gets its own
    // registration of the spheres, each inflated for the farthest origin of that zone only.  A ray reads the cells of its
    // origin's zone: nearby spheres have tight boxes, only far ones are fat.
    int zn[3];
    for (int k = 0; k < 3; ++k) zn[k] = static_cast<int>(std::min(6.0, std::max(1.0, std::ceil((hi[k] - lo[k]) / (D / 5.0)))));
    int nc[3];
    float lof[3], hf[3], hif[3], ihf[3], izs[3];
    std::vector<int> gridded;
    std::vector<char> is_always(n, 0);
    for (uint32_t a : g.always) is_always[a] = 1;
    size_t n_cells = 0, n_zones = 0;
    // zone (zx, zy, zz): the box of origins it serves (outer zones reach to the admissible ball's box; a small overlap
    // covers the float rounding of the device's zone index)
    auto zone_box = [&](const int zi[3], double zl[3], double zh[3]) {
        for (int k = 0; k < 3; ++k) {
            const double zs = (static_cast<double>(hif[k]) - lof[k]) / zn[k], m = 1e-3 * zs;
            zl[k] = zi[k] == 0 ? centre[k] - r_allow : std::max(centre[k] - r_allow, lof[k] + zi[k] * zs - m);
            zh[k] = zi[k] == zn[k] - 1 ? centre[k] + r_allow : std::min(centre[k] + r_allow, lof[k] + (zi[k] + 1) * zs + m);
            if (zh[k] < zl[k]) zh[k] = zl[k];
        }
    };
    auto box_for_zone = [&](int i, const double zl[3], const double zh[3]) {
        const double c[3] = {s->cx[i], s->cy[i], s->cz[i]};
        double far2 = 0.0;
        for (int k = 0; k < 3; ++k) { const double f = std::max(std::fabs(c[k] - zl[k]), std::fabs(c[k] - zh[k])); far2 += f * f; }
        return inflated(i, std::sqrt(far2));
    };
    auto range = [&](int i, double rb, int k, int& a, int& b) {
        const double c = k == 0 ? s->cx[i] : k == 1 ? s->cy[i] : s->cz[i];
        a = static_cast<int>(std::floor((c - rb - lof[k]) / hf[k]));
        b = static_cast<int>(std::floor((c + rb - lof[k]) / hf[k]));
        a = std::max(0, std::min(nc[k] - 1, a));
        b = std::max(0, std::min(nc[k] - 1, b));
    };
    // Resolution and zone count: coarsened until the grid has at most 8 cells per sphere and all zones together hold
    // at most ~32M registrations.
    for (int attempt = 0;; ++attempt) {
        for (;;) {
            double cells = 1.0;
            for (int k = 0; k < 3; ++k) {
                nc[k] = static_cast<int>(std::min(1024.0, std::max(1.0, std::floor((ghi[k] - glo[k]) / h0))));
                cells *= nc[k];
            }
            if (cells <= 8.0 * in.size() + 4096.0) break;
            h0 *= 1.25;
        }
        // float parameters exactly as the device will use them; cell sizes rounded up so that lo + n h covers the box
        for (int k = 0; k < 3; ++k) {
            lof[k] = std::nextafter(static_cast<float>(glo[k]), -std::numeric_limits<float>::infinity());
            double h = (ghi[k] - static_cast<double>(lof[k])) / nc[k];
            if (!(h > 0.0)) h = 1.0;
            hf[k] = std::nextafter(static_cast<float>(h * (1.0 + 1e-6)), std::numeric_limits<float>::infinity());
            hif[k] = std::nextafter(static_cast<float>(static_cast<double>(lof[k]) + static_cast<double>(nc[k]) * hf[k]), std::numeric_limits<float>::infinity());
            ihf[k] = static_cast<float>(1.0 / static_cast<double>(hf[k]));
            izs[k] = static_cast<float>(zn[k] / (static_cast<double>(hif[k]) - lof[k]));
            if (!std::isfinite(lof[k]) || !std::isfinite(hf[k]) || !std::isfinite(hif[k]) || !std::isfinite(ihf[k]) || !(ihf[k] > 0.0f) || !std::isfinite(izs[k])) { g.why = "scene extent is not representable"; return; }
        }
        n_cells = static_cast<size_t>(nc[0]) * nc[1] * nc[2];
        n_zones = static_cast<size_t>(zn[0]) * zn[1] * zn[2];
        // spheres that span too many cells in ANY zone are tested for every ray instead
        std::vector<char> fat(n, 0);
        double registrations = 0.0;
        for (int z = 0; z < static_cast<int>(n_zones); ++z) {
            const int zi[3] = {z % zn[0], (z / zn[0]) % zn[1], z / (zn[0] * zn[1])};
            double zl[3], zh[3];
            zone_box(zi, zl, zh);
            for (int i : in) {
                const double rb = box_for_zone(i, zl, zh);
                double span = 1.0;
                for (int k = 0; k < 3; ++k) { int a, b; range(i, rb, k, a, b); span *= (b - a + 1); }
                if (span > kGridMaxCellsPerSphere) fat[i] = 1; else registrations += span;
            }
        }
        const double cell_words = static_cast<double>(n_zones) * (static_cast<double>(n_cells) + 1.0);
        if (registrations <= 33554432.0 && cell_words <= 33554432.0) {
            gridded.clear();
            for (int i : in) { if (fat[i]) is_always[i] = 1; else gridded.push_back(i); }
            break;
        }
        if (attempt == 24) { g.why = "grid too large"; return; }
        if (n_zones > 1) { for (int k = 0; k < 3; ++k) zn[k] = std::max(1, zn[k] - 1); }      // fewer zones first, then coarser cells
        else h0 *= 1.5;
    }
    g.always.clear();
    for (int i = 0; i < n; ++i) if (is_always[i]) g.always.push_back(static_cast<uint32_t>(i));
    if (static_cast<int>(g.always.size()) > kGridMaxAlways) { g.why = "too many spheres that a grid cannot hold (outliers, or spanning thousands of cells)"; return; }
    // CSR per zone: cell_start[zone * (n_cells + 1) + cell] indexes the shared items array
    g.cell_start.assign(n_zones * (n_cells + 1), 0u);
    std::vector<uint32_t> count(n_cells + 1);
    size_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) g.items.assign(std::max<size_t>(1, total), 0u);
        size_t base = 0;
        for (int z = 0; z < static_cast<int>(n_zones); ++z) {
            const int zi[3] = {z % zn[0], (z / zn[0]) % zn[1], z / (zn[0] * zn[1])};
            double zl[3], zh[3];
            zone_box(zi, zl, zh);
            uint32_t* cs = g.cell_start.data() + static_cast<size_t>(z) * (n_cells + 1);
            if (pass == 0) std::fill(count.begin(), count.end(), 0u);
            else for (size_t c = 0; c < n_cells; ++c) count[c] = cs[c];          // fill cursors
            for (int i : gridded) {                                             // ascending sphere index within every cell
                const double rb = box_for_zone(i, zl, zh);
                int a[3], b[3];
                for (int k = 0; k < 3; ++k) range(i, rb, k, a[k], b[k]);
                for (int zz = a[2]; zz <= b[2]; ++zz)
                    for (int y = a[1]; y <= b[1]; ++y)
                        for (int x = a[0]; x <= b[0]; ++x) {
                            const size_t cell = (static_cast<size_t>(zz) * nc[1] + y) * nc[0] + x;
                            if (pass == 0) ++count[cell + 1];
                            else g.items[count[cell]++] = static_cast<uint32_t>(i);
                        }
            }
            if (pass == 0) {
                uint64_t run = base;
                cs[0] = static_cast<uint32_t>(run);
                for (size_t c = 0; c < n_cells; ++c) { run += count[c + 1]; cs[c + 1] = static_cast<uint32_t>(run); }
                if (run > 0x7FFFFFFFull) { g.why = "grid too large"; return; }
                base = static_cast<size_t>(run);
                total = base;
            }
        }
    }
    if (g.always.empty()) g.always.push_back(0u);      // keep the array non-empty for the upload; n_always stays 0
    GridDev& d = g.dev;
    d.n_always = static_cast<int32_t>(n - static_cast<int>(gridded.size()));
    d.nx = nc[0]; d.ny = nc[1]; d.nz = nc[2];
    d.lox = lof[0]; d.loy = lof[1]; d.loz = lof[2];
    d.hix = hif[0]; d.hiy = hif[1]; d.hiz = hif[2];
    d.hx = hf[0]; d.hy = hf[1]; d.hz = hf[2];
    d.ihx = ihf[0]; d.ihy = ihf[1]; d.ihz = ihf[2];
    d.znx = zn[0]; d.zny = zn[1]; d.znz = zn[2];
    d.izx = izs[0]; d.izy = izs[1]; d.izz = izs[2];
    d.zone_stride = static_cast<int32_t>(n_cells + 1);
    d.cx = static_cast<float>(centre[0]); d.cy = static_cast<float>(centre[1]); d.cz = static_cast<float>(centre[2]);
    d.r_allow2 = static_cast<float>(r_allow * r_allow * (1.0 - 1e-5));
    d.slack = static_cast<float>(slack);
    g.mean_items_per_cell = static_cast<double>(total) / static_cast<double>(n_cells * n_zones);
    g.mean_cells_per_sphere = gridded.empty() ? 0.0 : static_cast<double>(total) / static_cast<double>(gridded.size() * n_zones);
    g.usable = true;
}

// The partition of a pixel's `spp` samples into runs (a lane traces one run of one pixel, then adds its integer colour
// sums to the pixel's accumulator).  Every SAMPLE is converted to fixed point before it is summed, so the image does not
// depend on the partition at all: it is chosen per rank for load balance alone.  run_len > 0: equal runs of that
// length.  run_len == 0 (default): decreasing runs, L = ceil(remaining / k): units are large while there is plenty of
// work and single samples at the end of the frame, so that no lane is still inside a long unit when the packet queue
// runs dry (the tail sets the kernel's end; hot pixels — glass interiors — cost up to 7x the mean per sample, and a
// wave that its SIMD's arbiter favours least runs at a third of the mean rate).  k grows as the rank's share shrinks
// (fewer pixels per lane -> the tail matters more): k = clamp(3e7 / pixels of this rank, 32, 256) — measured on the
// 1200x800x500 frame (tools/sched_sweep.py): k = 32 at 1 GPU (106 runs, 108.3 ms), k = 250 for an eighth of it
// (15.3 ms instead of 17.3 with k = 64).  run_len < 0: decreasing runs with k = -run_len (schedule studies).
// fill_runs: the table itself, usable from host and device code (the entry point that allocates nothing builds it with
// a one-thread kernel straight into its static device array).  Returns the number of runs; writes at most `cap`.
__host__ __device__ inline int fill_runs(int spp, long long n_pixels, int run_len, uint2* out, int cap) {
    int n = 0;
    if (spp <= 0) return 0;
    if (run_len > 0) {
        for (int s = 0; s < spp; s += run_len, ++n)
            if (n < cap) out[n] = make_uint2(static_cast<uint32_t>(s), static_cast<uint32_t>(run_len < spp - s ? run_len : spp - s));
        return n;
    }
    const double px = static_cast<double>(n_pixels);
    double kd = ceil(3.0e7 / (px > 1.0 ? px : 1.0));
    kd = kd < 32.0 ? 32.0 : (kd > 256.0 ? 256.0 : kd);
    const int k = run_len < 0 ? -run_len : static_cast<int>(kd);      // run_len < 0: decreasing runs with k = -run_len (schedule studies)
    for (int s = 0; s < spp; ++n) {
        const int rem = spp - s;
        const int len = (rem + k - 1) / k;          // >= 1
        if (n < cap) out[n] = make_uint2(static_cast<uint32_t>(s), static_cast<uint32_t>(len));
        s += len;
    }
    return n;
}

inline std::vector<uint2> run_schedule(int spp, long long n_pixels, int run_len) {
    std::vector<uint2> runs(static_cast<size_t>(fill_runs(spp, n_pixels, run_len, nullptr, 0)));
    fill_runs(spp, n_pixels, run_len, runs.data(), static_cast<int>(runs.size()));
    return runs;
}

} // namespace rtk
