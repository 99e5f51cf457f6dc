// emul.cpp -> tests/cpu_emul/libemul.so: the DEVICE code of librtk (rtk_device.hpp, rtk_render.hpp) and its host
// scene preparation (rtk_scene_prep.hpp) compiled for the CPU with one lane per wave (rtk_host_shim.hpp).
// TEST INFRASTRUCTURE ONLY — lets `pytest -m "not gpu"` check the kernels' logic against the oracle.
// Build: g++ -std=c++17 -O2 -ffp-contract=off -mfma -DRTK_HOST_EMULATION -DRTK_LANES=1 -I. -I<csrc> ...
#include "rtk_device.hpp"
#include "rtk_intersect.hpp"
#include "rtk_render.hpp"
#include "rtk_scene_prep.hpp"

#include <atomic>
#include <thread>
#include <vector>

#include <mutex>

namespace rtk_shim {
thread_local int ulp_noise = 0;
thread_local uint32_t noise_state = 12345u;
thread_local WaveStats wave_stats = {};
}

// Statistics of all emulated waves since the last emul_reset_stats (every OS thread folds its own in).
static rtk_shim::WaveStats g_stats = {};
static std::mutex g_stats_mutex;
static bool g_simt_error = false;
static void fold_stats() {
    std::lock_guard<std::mutex> lock(g_stats_mutex);
    auto* dst = reinterpret_cast<unsigned long long*>(&g_stats);
    auto* src = reinterpret_cast<unsigned long long*>(&rtk_shim::wave_stats);
    for (size_t i = 0; i < sizeof(g_stats) / sizeof(unsigned long long); ++i) dst[i] += src[i];
    rtk_shim::wave_stats = {};
}

// Runs body(lane) for every lane of one wavefront on the calling thread: 64 fibers meeting at the votes
// (RTK_LANES == 64, rtk_simt.hpp) or a single lane (RTK_LANES == 1).
template <class F>
static void run_lanes(F&& body) {
#if RTK_LANES == 64
    rtk_simt::Wave* w = new rtk_simt::Wave();
    if (!rtk_simt::run_wave(*w, body)) g_simt_error = true;
    delete w;
#else
    body(0);
#endif
}

using namespace rtk;

struct Emul {
    PreparedScene ps;          // as the path the scene runs on gets it (compact layout on the LDS-resident path)
    SceneDev dev;
    GridHost grid;
    bool compact = false;
};

// Which instantiation a scene runs on: the GPU takes Intersector<true> (operands straight from the LDS copy) up to
// kMaxLdsSpheres and Intersector<false> (list streamed through a wave-private chunk) beyond.  The emulation follows
// that rule unless the staged path is forced for every scene (the 64-lane build and asan_driver_staged do, so that its
// cross-lane chunk logic is exercised on the small fixtures too).
#ifndef EMUL_FORCE_STAGED
#define EMUL_FORCE_STAGED (RTK_LANES == 64)
#endif
static bool g_force_staged = EMUL_FORCE_STAGED;
static bool g_local_frames = true;      // streamed scenes with local tiles use them (emul_local_frames(0): the staged vector-ALU walk)
static bool use_lds_path(const Emul& e) { return e.compact; }

template <bool LDS>
static Intersector<LDS> make_isect(const Emul& e, uint16_t* list, uint8_t* dbg, int lane = 0, float4* stage = nullptr, uint32_t* masks = nullptr) {
    static thread_local std::vector<uint2> entries_buf(static_cast<size_t>(kEntryCap) * kWave);          // one wave at a time per thread
    Intersector<LDS> it;
    it.filt = e.ps.filt.data();
    if (LDS) {
        it.wdesc = e.ps.wdesc.data();
        it.masks = masks;
    }
    it.exact = e.ps.exact.data();
    it.list = list;
    it.fb = e.ps.fb;
    it.n = e.ps.n;
    it.lane = lane;
    it.stage = stage;
    if (!LDS && e.ps.use_local && g_local_frames) {
        it.ltiles = e.ps.ltiles.data();
        it.lwdesc = e.ps.lwdesc.data();
        it.lwords = e.ps.local_words;
        it.entries = entries_buf.data();
    }
    it.dbg_flags = dbg;
    return it;
}

extern "C" {

void* emul_create(const rtk_scene* s) {
    Emul* e = new Emul();
    e->compact = !g_force_staged && s->n <= kMaxLdsSpheres;
    prepare_scene(s, e->ps, e->compact);
    e->dev = {e->ps.filt.data(), e->ps.exact.data(), e->ps.m0.data(), e->ps.m1.data(), e->ps.n, e->ps.npad, e->ps.fb};
    e->dev.filt_len = static_cast<int32_t>(e->ps.filt.size());
    e->dev.filt_lds = e->ps.filt_lds;
    if (e->compact) e->dev.wdesc = e->ps.wdesc.data();
    build_grid(s, e->ps, e->grid);
    e->grid.dev.cell_start = e->grid.cell_start.data();
    e->grid.dev.items = e->grid.items.data();
    e->grid.dev.always = e->grid.always.data();
    return e;
}

// uniform grid as build_grid made it: info = usable, nx, ny, nz, n_always, items (all zones), mean items per cell x1000,
// mean cells per sphere and zone x1000, origin zones
void emul_grid_info(void* h, int64_t* info8) {
    const Emul& e = *static_cast<Emul*>(h);
    const GridHost& g = e.grid;
    info8[0] = g.usable; info8[1] = g.dev.nx; info8[2] = g.dev.ny; info8[3] = g.dev.nz; info8[4] = g.dev.n_always;
    info8[5] = g.usable ? static_cast<int64_t>(g.cell_start.back()) : 0;
    info8[6] = static_cast<int64_t>(g.mean_items_per_cell * 1000.0); info8[7] = static_cast<int64_t>(g.mean_cells_per_sphere * 1000.0);
    info8[8] = static_cast<int64_t>(g.dev.znx) * g.dev.zny * g.dev.znz;
}
void emul_destroy(void* h) { delete static_cast<Emul*>(h); }
// words of the scene that got 8-byte class records (0 when the scene runs on the staged path)
int emul_class_words(void* h) { return static_cast<Emul*>(h)->ps.class_words; }

void emul_set_ulp_noise(int k) { rtk_shim::ulp_noise = k; }
// layout helpers of the matrix-pipe filter (rtk_device.hpp)
int emul_tile_row_sphere(int rho) { return tile_row_sphere(rho); }
uint16_t emul_f16_rne(float v) { return f16_rne(v); }
float emul_f16_value(uint16_t b) { return f16_value(b); }
void emul_quad_term(int k, int n_mono, int* out3) { quad_term(k, n_mono, out3[0], out3[1], out3[2]); }
void emul_force_staged(int on) { g_force_staged = on != 0; }
void emul_local_frames(int on) { g_local_frames = on != 0; }
// words of the scene with a tile in their own frame, and whether the streamed walk uses them
int emul_local_words(void* h) { const Emul& e = *static_cast<Emul*>(h); return (!e.compact && e.ps.use_local && g_local_frames) ? e.ps.local_class_words : 0; }

// filter bounds and per-sphere "filterable" flag as prepare_scene decided them
void emul_scene_info(void* h, float* bounds8, uint8_t* filterable) {
    const Emul& e = *static_cast<Emul*>(h);
    bounds8[0] = e.ps.fb.c_max; bounds8[1] = e.ps.fb.r_max2; bounds8[2] = e.ps.fb.r_min; bounds8[3] = e.ps.fb.sigma;
    bounds8[4] = e.ps.fb.cls_cy; bounds8[5] = e.ps.fb.cls_r2; bounds8[6] = e.ps.fb.acc_eps; bounds8[7] = static_cast<float>(e.ps.fb.kinds);
    for (int i = 0; i < e.ps.n; ++i) filterable[i] = e.ps.filterable[i] ? 1 : 0;
}

// Intersector::closest_hit per ray; `flags` (optional, [n][scene size]) receives the filter's candidate flags.
void emul_intersect(void* h, const float* rays, uint64_t n, float t_min, float t_max, uint8_t* hit, int32_t* index, float* t, uint8_t* flags,
                    uint64_t* n_exact_total, int ulp_noise, int accel) {
    const Emul& e = *static_cast<Emul*>(h);
    if (accel && !e.grid.usable) accel = 0;
    rtk_shim::ulp_noise = ulp_noise;
    std::vector<uint16_t> list(kListCap * kWave);
    std::vector<float4> stage(kStageSpheres);
    std::vector<uint32_t> masks(static_cast<size_t>(e.ps.npad / 32) * kWave);       // the wave's candidate mask words (LDS path)
    std::vector<uint32_t> ne_lane(kWave, 0);
    uint64_t total = 0;
    const bool lds = use_lds_path(e);
    for (uint64_t base = 0; base < n; base += kWave) {
        run_lanes([&](int lane) {
            const uint64_t i = base + static_cast<uint64_t>(lane);
            const bool active = i < n;
            uint8_t* dbg = (flags && active) ? flags + i * static_cast<uint64_t>(e.ps.n) : nullptr;
            Ray r = {};
            if (active) {
                const float* q = rays + 6 * i;
                r = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
            }
            float tt;
            int idx;
            uint32_t ne = 0;
            if (lds) {
                const Intersector<true> it = make_isect<true>(e, list.data(), dbg, lane, nullptr, masks.data());
                if (accel) it.closest_hit_grid(e.grid.dev, r, active, t_min, t_max, tt, idx, ne);
                else it.closest_hit(r, active, t_min, t_max, tt, idx, ne);
            } else {
                const Intersector<false> it = make_isect<false>(e, list.data(), dbg, lane, stage.data());
                if (accel) it.closest_hit_grid(e.grid.dev, r, active, t_min, t_max, tt, idx, ne);
                else it.closest_hit(r, active, t_min, t_max, tt, idx, ne);
            }
            ne_lane[lane] = ne;
            if (active) {
                hit[i] = idx >= 0;
                index[i] = idx;
                t[i] = idx >= 0 ? tt : 0.0f;
            }
        });
        total += ne_lane[0];          // closest_hit counts per wave: every lane holds the wave's total
        for (int l = 0; l < kWave; ++l) ne_lane[l] = 0;
    }
    fold_stats();
    if (n_exact_total) *n_exact_total = total;
    rtk_shim::ulp_noise = 0;
}

void emul_scatter(void* h, const float* rays, const int32_t* index, const float* t, const float* u4, uint64_t n, float* out7) {
    const Emul& e = *static_cast<Emul*>(h);
    for (uint64_t i = 0; i < n; ++i) {
        const float* q = rays + 6 * i;
        const Ray in = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
        const int idx = index[i];
        const float u[4] = {u4[4 * i], u4[4 * i + 1], u4[4 * i + 2], u4[4 * i + 3]};
        Ray out;
        const bool ok = hit_and_scatter(in, unit_vector(in.d), t[i], e.ps.exact[idx], e.ps.m0[idx], e.ps.m1[idx], u, out);
        float* o = out7 + 7 * i;
        o[0] = out.o.x; o[1] = out.o.y; o[2] = out.o.z; o[3] = out.d.x; o[4] = out.d.y; o[5] = out.d.z; o[6] = ok ? 1.0f : 0.0f;
    }
}

void emul_primary(const float* cam22, int W, int H, uint64_t seed, const int32_t* xys, uint64_t n, float* out) {
    CameraDev cam;
    std::memcpy(&cam, cam22, 88);
    for (uint64_t i = 0; i < n; ++i) {
        const int x = xys[3 * i], row = xys[3 * i + 1], s = xys[3 * i + 2];
        const uint32_t pixel = static_cast<uint32_t>(row) * static_cast<uint32_t>(W) + static_cast<uint32_t>(x);
        const uint4 rnd = philox4x32_10(make_uint4(pixel, static_cast<uint32_t>(s), 0u, 0u), make_uint2(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)));
        const float u[4] = {u01(rnd.x), u01(rnd.y), u01(rnd.z), u01(rnd.w)};
        const Ray r = primary_ray(cam, x, row, W, H, u);
        float* o = out + 6 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
    }
}

void emul_philox(uint64_t seed, const uint32_t* ctr4, uint64_t n, uint32_t* out4) {
    for (uint64_t i = 0; i < n; ++i) {
        const uint4 r = philox4x32_10(make_uint4(ctr4[4 * i], ctr4[4 * i + 1], ctr4[4 * i + 2], ctr4[4 * i + 3]),
                                      make_uint2(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)));
        out4[4 * i] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
    }
}

// render_wave on `threads` CPU threads (each one "wavefront" of one lane) for rank `p->rank` of `p->world_size`;
// writes this rank's tile sums [local tile][64][3] like rtk_render_tiles_device.  stats4 = rays, samples, exact tests, passes.
int emul_render_tiles(void* h, const float* cam22, const rtk_params* p, float* tile_sums, uint64_t* stats4, int threads, uint8_t* seg_counts, int accel) {
    const Emul& e = *static_cast<Emul*>(h);
    if (accel && !e.grid.usable) accel = 0;
    const int world = p->world_size > 0 ? p->world_size : 1;
    int tw = p->tile_w, th = p->tile_h;
    if (tw == 0 && th == 0) { tw = 8; th = 8; }
    if (tw * th != kTilePixels) return -1;
    const int tiles_x = (p->width + tw - 1) / tw;
    const int64_t tiles = static_cast<int64_t>(tiles_x) * ((p->height + th - 1) / th);
    const int64_t nlt = (tiles - p->rank + world - 1) / world;
    const std::vector<uint2> runs = run_schedule(p->spp, static_cast<long long>(nlt) * kTilePixels, p->run_len);
    const int n_runs = static_cast<int>(runs.size());
    std::vector<unsigned long long> accum(static_cast<size_t>(nlt) * kTilePixels * 3, 0ull);
    uint32_t queue = 0;
    unsigned long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RenderArgs A;
    A.scene = e.dev;
    std::memcpy(&A.cam, cam22, 88);
    A.W = p->width; A.H = p->height; A.spp = p->spp; A.max_depth = p->max_depth;
    A.sample_offset = static_cast<uint32_t>(p->sample_offset);
    A.seed_lo = static_cast<uint32_t>(p->seed); A.seed_hi = static_cast<uint32_t>(p->seed >> 32);
    A.t_min = p->t_min;
    A.tile_w = tw; A.tile_h = th; A.tiles_x = tiles_x;
    A.tile_w_log2 = 0;
    while ((1 << A.tile_w_log2) < tw) ++A.tile_w_log2;
    A.rank = p->rank; A.world = world;
    A.n_local_tiles = static_cast<int32_t>(nlt);
    A.runs = runs.data(); A.n_runs = n_runs;
    A.total_packets = static_cast<uint32_t>(nlt * n_runs);
    A.accum = accum.data();
    A.queue = &queue;
    A.stats = stats;
    A.dbg_seg = seg_counts;
    A.grid = e.grid.dev;
    const bool lds = use_lds_path(e);
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; ++k)
        pool.emplace_back([&] {
            std::vector<uint16_t> list(kListCap * kWave);
            std::vector<float4> stage(kStageSpheres);
            std::vector<uint32_t> masks(static_cast<size_t>(e.ps.npad / 32) * kWave);
            run_lanes([&](int lane) {
                if (lds) {
                    const Intersector<true> it = make_isect<true>(e, list.data(), nullptr, lane, nullptr, masks.data());
                    if (accel) render_wave<true, true>(A, it, &A.cam, &A.grid);
                    else render_wave<true, false>(A, it, &A.cam);
                } else {
                    const Intersector<false> it = make_isect<false>(e, list.data(), nullptr, lane, stage.data());
                    if (accel) render_wave<false, true>(A, it, &A.cam, &A.grid);
                    else render_wave<false, false>(A, it, &A.cam);
                }
            });
            fold_stats();
        });
    for (auto& t : pool) t.join();
    for (size_t i = 0; i < accum.size(); ++i) tile_sums[i] = accum_to_float(accum[i]);
    for (int i = 0; i < 4; ++i) stats4[i] = stats[i];
    return 0;
}

// wave-level statistics (rtk_shim::WaveStats as an array of u64) since the last reset; returns the lane count of the build
int emul_get_stats(unsigned long long* out, int cap) {
    std::lock_guard<std::mutex> lock(g_stats_mutex);
    const int n = static_cast<int>(sizeof(g_stats) / sizeof(unsigned long long));
    const auto* src = reinterpret_cast<const unsigned long long*>(&g_stats);
    for (int i = 0; i < n && i < cap; ++i) out[i] = src[i];
    return kWave;
}
void emul_reset_stats() {
    std::lock_guard<std::mutex> lock(g_stats_mutex);
    g_stats = {};
}
int emul_simt_error() { return g_simt_error ? 1 : 0; }

} // extern "C"
