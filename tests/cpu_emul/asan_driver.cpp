// asan_driver.cpp — the CPU build of the device code (emul.cpp) linked into one executable under AddressSanitizer and
// UBSan (host code only): every index the kernels form — sphere lists, candidate lists, grid cells and items, tile
// accumulators — is checked against the real extent of its buffer.  TEST INFRASTRUCTURE ONLY.
//   make -C tests/cpu_emul asan && tests/cpu_emul/asan_driver
#include "../../include/rtk.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

extern "C" {
void* emul_create(const rtk_scene* s);
void emul_destroy(void* h);
void emul_grid_info(void* h, int64_t* info8);
void emul_intersect(void* h, const float* rays, uint64_t n, float t_min, float t_max, uint8_t* hit, int32_t* index, float* t, uint8_t* flags,
                    uint64_t* n_exact_total, int ulp_noise, int accel);
int emul_render_tiles(void* h, const float* cam22, const rtk_params* p, float* tile_sums, uint64_t* stats4, int threads, uint8_t* seg_counts, int accel);
}

struct Scene {
    std::vector<float> cx, cy, cz, r, mr, mg, mb, rough, metal, ior;
    std::vector<int32_t> type;
    rtk_scene view() { return {static_cast<int32_t>(cx.size()), cx.data(), cy.data(), cz.data(), r.data(), mr.data(), mg.data(), mb.data(), rough.data(), metal.data(), ior.data(), type.data()}; }
};

static Scene make_scene(int k, std::mt19937& g, bool ground) {
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    Scene s;
    auto add = [&](float x, float y, float z, float rad, int ty) {
        s.cx.push_back(x); s.cy.push_back(y); s.cz.push_back(z); s.r.push_back(rad);
        s.mr.push_back(0.2f + 0.7f * u(g)); s.mg.push_back(0.2f + 0.7f * u(g)); s.mb.push_back(0.2f + 0.7f * u(g));
        s.rough.push_back(0.5f * u(g)); s.metal.push_back(ty == 2); s.ior.push_back(1.5f); s.type.push_back(ty);
    };
    if (ground) add(0, -1000, 0, 1000, 1);
    for (int a = -k; a < k; ++a)
        for (int b = -k; b < k; ++b) {
            const float m = u(g);
            add(a + 0.9f * u(g), 0.2f, b + 0.9f * u(g), 0.2f, m < 0.8f ? 1 : m < 0.95f ? 2 : 3);
        }
    add(0, 1, 0, 1, 3); add(-4, 1, 0, 1, 1); add(4, 1, 0, 1, 2);
    return s;
}

static int run(int k, bool ground, int accel, int n_rays, std::mt19937& g) {
    Scene s = make_scene(k, g, ground);
    rtk_scene v = s.view();
    void* h = emul_create(&v);
    int64_t info[9];
    emul_grid_info(h, info);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::vector<float> rays(static_cast<size_t>(n_rays) * 6);
    for (int i = 0; i < n_rays; ++i) {
        float* q = &rays[6 * i];
        const float ext = (i % 7 == 0) ? 40.0f * k : 1.0f * k;      // some origins outside the grid's ball
        q[0] = ext * u(g); q[1] = 0.01f + 2.0f * std::fabs(u(g)); q[2] = ext * u(g);
        q[3] = u(g); q[4] = 0.3f * u(g); q[5] = u(g);
        if (i % 11 == 0) { q[3] = 0; q[5] = 0; q[4] = u(g) < 0 ? -1.f : 1.f; }
        if (i % 13 == 0) q[3] = std::numeric_limits<float>::quiet_NaN();
        if (i % 17 == 0) q[0] = std::numeric_limits<float>::infinity();
        if (i % 19 == 0) { q[3] = q[4] = q[5] = 0; }
        if (i % 23 == 0) { q[3] *= 1e-30f; q[4] *= 1e-30f; q[5] *= 1e-30f; }
        if (i % 29 == 0) { q[3] *= 1e30f; q[4] *= 1e30f; q[5] *= 1e30f; }
    }
    std::vector<uint8_t> hit(n_rays);
    std::vector<int32_t> idx(n_rays);
    std::vector<float> t(n_rays);
    uint64_t ne = 0;
    const float ranges[4][2] = {{0.001f, INFINITY}, {-INFINITY, INFINITY}, {-5.0f, 3.0f}, {2.0f, 1.0f}};
    long hits = 0;
    for (const auto& tr : ranges) {
        emul_intersect(h, rays.data(), n_rays, tr[0], tr[1], hit.data(), idx.data(), t.data(), nullptr, &ne, 0, accel);
        for (int i = 0; i < n_rays; ++i) {
            hits += hit[i];
            if (idx[i] >= static_cast<int>(s.cx.size()) || idx[i] < -1) { std::printf("index out of range\n"); return 1; }
        }
    }
    // a small render, three ranks, odd image size
    float cam[22] = {13, 2, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.05f};
    // a plain pinhole-ish frame looking at the origin (exactness is not the point here, indexing is)
    cam[3] = 10.0f; cam[4] = 0.5f; cam[5] = 4.5f;      // lower-left corner
    cam[6] = 1.0f; cam[7] = 0.0f; cam[8] = -4.0f;      // horizontal
    cam[9] = -0.2f; cam[10] = 2.8f; cam[11] = -0.1f;   // vertical
    cam[12] = 0.3f; cam[13] = 0; cam[14] = -0.95f; cam[15] = 0; cam[16] = 1; cam[17] = 0; cam[18] = 0.95f; cam[19] = 0; cam[20] = 0.3f;
    for (int rank = 0; rank < 3; ++rank) {
        rtk_params p{};
        p.width = 37; p.height = 23; p.spp = 5; p.max_depth = 12; p.seed = 3; p.t_min = 0.001f; p.rank = rank; p.world_size = 3;
        const int tiles = ((37 + 7) / 8) * ((23 + 7) / 8), nlt = (tiles - rank + 2) / 3;
        std::vector<float> sums(static_cast<size_t>(nlt) * 64 * 3);
        uint64_t st[4];
        if (emul_render_tiles(h, cam, &p, sums.data(), st, 3, nullptr, accel) != 0) { std::printf("render failed\n"); return 1; }
    }
    std::printf("k=%d ground=%d accel=%d: %zu spheres, grid usable=%lld dims=%lldx%lldx%lld always=%lld, %ld hits, %.1f exact tests per ray\n", k, ground, accel,
                s.cx.size(), (long long)info[0], (long long)info[1], (long long)info[2], (long long)info[3], (long long)info[4], hits, double(ne) / n_rays);
    emul_destroy(h);
    return 0;
}

int main() {
    std::mt19937 g(7);
    int rc = 0;
    for (int accel = 0; accel < 2; ++accel) {
        rc |= run(3, true, accel, 4000, g);
        rc |= run(11, true, accel, 4000, g);
        rc |= run(11, false, accel, 2000, g);
        rc |= run(40, true, accel, accel ? 3000 : 300, g);
        rc |= run(130, true, accel, accel ? 1500 : 30, g);       // beyond 64k spheres: list re-basing; grid with one y layer
    }
    std::printf(rc ? "FAILED\n" : "asan driver: all clean\n");
    return rc;
}
