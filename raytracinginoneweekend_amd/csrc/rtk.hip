// rtk.hip — kernels and C ABI of librtk.so (see include/rtk.h for the contract and the reference
// citations of every entry point).  gfx950 only; compiled with -ffp-contract=off (rtk_device.hpp).
#include "rtk_device.hpp"
#include "rtk_intersect.hpp"
#include "rtk_render.hpp"
#include "rtk_scene_prep.hpp"

#include "../../include/rtk.h"

#include <rccl/rccl.h>      // types and prototypes only: the library is opened on demand (rtk_multi_create), never linked
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <vector>

using namespace rtk;

// =================================================================================================
// Kernels
// =================================================================================================

// Dynamic LDS layout.  LDS-resident scene: [filt_lds] float4: A fragments of the words' f16 tiles, records of a short tail | [npad / 32] word
// descriptors | [n] exact float4 | [4 waves][npad / 32][64] u32 candidate mask words (40 KiB per workgroup is what four
// workgroups per CU allow: the stock scene needs 39.9).  Streamed scene, per wave: [kListCap][64] u16 candidate list | [kStageSpheres] float4 staging chunk, or in the same memory [kEntryCap][64] (mask, index) entries of the word-local walk.
constexpr size_t kStreamLdsPerWave = kEntryCap * kWave * sizeof(uint2) > kListCap * kWave * 2 + kStageSpheres * sizeof(float4) ? kEntryCap * kWave * sizeof(uint2) : kListCap * kWave * 2 + kStageSpheres * sizeof(float4);
template <bool LDS>
__device__ __forceinline__ Intersector<LDS> make_intersector(const SceneDev& sc, unsigned char* smem) {
    Intersector<LDS> it;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    unsigned char* p = smem;
    if (LDS) {
        const int nw = sc.npad / 32, nt = sc.filt_lds;
        float4* lf = reinterpret_cast<float4*>(p);
        float4* lw = lf + nt;
        float4* le = lw + nw;
        for (int i = threadIdx.x; i < nt; i += blockDim.x) lf[i] = sc.filt[i];
        for (int i = threadIdx.x; i < nw; i += blockDim.x) lw[i] = sc.wdesc[i];
        for (int i = threadIdx.x; i < sc.n; i += blockDim.x) le[i] = sc.exact[i];
        it.filt = lf;
        it.wdesc = lw;
        it.exact = le;
        p += static_cast<size_t>(nt + nw + sc.n) * sizeof(float4);
        it.masks = reinterpret_cast<uint32_t*>(p) + wave * (nw * kWave);
        it.list = nullptr;
    } else {
        it.filt = sc.filt;
        it.exact = sc.exact;
        // per wave kStreamLdsPerWave bytes: the staged walk's list and chunk, or (same memory) the local walk's entries
        unsigned char* mine = p + wave * kStreamLdsPerWave;
        it.list = reinterpret_cast<uint16_t*>(mine);
        it.stage = reinterpret_cast<float4*>(mine + kListCap * kWave * 2);
        it.entries = reinterpret_cast<uint2*>(mine);
        it.ltiles = sc.ltiles;
        it.lwdesc = sc.lwdesc;
        it.lwords = sc.lwords;
    }
    it.n = sc.n;
    it.fb = sc.fb;
    it.lane = lane;
    __syncthreads();
    return it;
}

static size_t lds_bytes(bool lds, const SceneDev& sc, int waves = kWavesPerBlock) {
    return lds ? static_cast<size_t>(sc.filt_lds + sc.npad / 32 + sc.n) * sizeof(float4) + static_cast<size_t>(waves) * (sc.npad / 32) * kWave * sizeof(uint32_t)
               : static_cast<size_t>(kWavesPerBlock) * kStreamLdsPerWave;
}

// GRID: closest hit through the uniform grid (rtk_context_set_accel) instead of the list walk.
// WAVES: wavefronts per workgroup.  4 (one per SIMD; up to four workgroups per CU) unless the scene's LDS footprint leaves room
// for only ONE such workgroup per CU (general words beyond ~600 spheres: tiles 2 KiB per word): then 8, so that the two
// waves of a SIMD share one copy of the tiles instead of a second workgroup not fitting at all (rtk_set_scene decides).
template <bool LDS, bool GRID = false, int WAVES = kWavesPerBlock>
__global__ __launch_bounds__(WAVES * kWave, WAVES == kWavesPerBlock ? 4 : 2) void render_kernel(const RenderArgs A) {      // 4 waves per SIMD: at most 128 VGPRs
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ CameraDev s_cam;                                   // 88 B of static LDS next to the dynamic part
    if (threadIdx.x < sizeof(CameraDev) / sizeof(float))
        reinterpret_cast<float*>(&s_cam)[threadIdx.x] = reinterpret_cast<const float*>(&A.cam)[threadIdx.x];
    // The grid's 35 parameters as well: as kernel arguments they are SGPRs that the grid walk keeps live through its
    // loops, and the register allocator spilled 51 of them (some to scratch memory).
    __shared__ GridDev s_grid;
    if (GRID && threadIdx.x < sizeof(GridDev) / 4)
        reinterpret_cast<uint32_t*>(&s_grid)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&A.grid)[threadIdx.x];
    const Intersector<LDS> isect = make_intersector<LDS>(A.scene, smem);      // ends with __syncthreads()
    render_wave<LDS, GRID>(A, isect, &s_cam, &s_grid);
}

// accumulators -> fp32 sums, and clear them for the next frame.
__global__ void finalize_kernel(unsigned long long* accum, float* out, size_t n) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i < n) {
        out[i] = accum_to_float(accum[i]);
        accum[i] = 0ull;
    }
}

// [world][tiles_per_rank_padded][64][3] -> [H][W][3]; T = float (colour sums) or uint8_t (quantised bytes)
template <typename T>
__global__ void deinterleave_kernel(const T* g, long long tprp, int W, int H, int tile_w, int tile_h, int tiles_x, int world, T* img) {
    const long long p = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (p >= static_cast<long long>(W) * H) return;
    const int x = static_cast<int>(p % W), row = static_cast<int>(p / W);
    const int tx = x / tile_w, ty = row / tile_h;
    const int gt = ty * tiles_x + tx;
    const int k = (row % tile_h) * tile_w + (x % tile_w);
    const long long src = ((static_cast<long long>(gt % world) * tprp + gt / world) * kTilePixels + k) * 3;
    img[p * 3 + 0] = g[src + 0];
    img[p * 3 + 1] = g[src + 1];
    img[p * 3 + 2] = g[src + 2];
}

// Color.hpp:28-48.
__global__ void quantise_kernel(const float* sums, long long n, int spp, uint8_t* out) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const float scale = 1.0f / static_cast<float>(spp);
    float v = __builtin_sqrtf(scale * sums[i]);
    v = (v < 0.0f) ? 0.0f : (1.0f < v) ? 1.0f : v;   // std::clamp semantics, NaN passes through
    out[i] = static_cast<uint8_t>(static_cast<int>(255.0f * v));
}

template <bool LDS, bool GRID = false>
__global__ __launch_bounds__(kBlockThreads) void intersect_kernel(const SceneDev sc, const float* rays, size_t n, float t_min, float t_max,
                                                                   uint8_t* out_hit, int32_t* out_index, float* out_t, const GridDev grid) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Intersector<LDS> isect = make_intersector<LDS>(sc, smem);
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    // whole wavefronts iterate together (closest_hit uses wave-level votes)
    const size_t n_round = (n + kWave - 1) / kWave * kWave;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n_round; i += stride) {
        const bool active = i < n;
        Ray r = {};
        if (active) {
            const float* q = rays + 6 * i;
            r = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
        }
        float t;
        int idx;
        uint32_t n_exact = 0;
        if (GRID) isect.closest_hit_grid(grid, r, active, t_min, t_max, t, idx, n_exact);
        else isect.closest_hit(r, active, t_min, t_max, t, idx, n_exact);
        if (active) {
            out_hit[i] = idx >= 0 ? 1 : 0;
            out_index[i] = idx;
            out_t[i] = idx >= 0 ? t : 0.0f;
        }
    }
}

__global__ void scatter_kernel(const SceneDev sc, const float* rays, const int32_t* index, const float* t, const float* u4, size_t n, float* out7) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const float* q = rays + 6 * i;
    const Ray in = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
    const int idx = index[i];
    const float u[4] = {u4[4 * i], u4[4 * i + 1], u4[4 * i + 2], u4[4 * i + 3]};
    Ray out;
    const bool ok = hit_and_scatter(in, unit_vector(in.d), t[i], sc.exact[idx], sc.mat0[idx], sc.mat1[idx], u, out);
    float* o = out7 + 7 * i;
    o[0] = out.o.x; o[1] = out.o.y; o[2] = out.o.z; o[3] = out.d.x; o[4] = out.d.y; o[5] = out.d.z;
    o[6] = ok ? 1.0f : 0.0f;
}

__global__ void primary_kernel(const CameraDev cam, int W, int H, uint32_t seed_lo, uint32_t seed_hi, const int32_t* xys, size_t n, float* out) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const int x = xys[3 * i], row = xys[3 * i + 1], s = xys[3 * i + 2];
    const uint32_t pixel = static_cast<uint32_t>(row) * static_cast<uint32_t>(W) + static_cast<uint32_t>(x);
    const uint4 rnd = philox4x32_10(make_uint4(pixel, static_cast<uint32_t>(s), 0u, 0u), make_uint2(seed_lo, seed_hi));
    const float u[4] = {u01(rnd.x), u01(rnd.y), u01(rnd.z), u01(rnd.w)};
    const Ray r = primary_ray(cam, x, row, W, H, u);
    float* o = out + 6 * i;
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

__global__ void sample_kernel(int kind, const float* u4, size_t n, float* out3) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const float* u = u4 + 4 * i;
    V3 v = {0.0f, 0.0f, 0.0f};
    if (kind == 0) v = sample_in_unit_sphere(u[0], u[1], u[2]);
    else if (kind == 1) v = sample_unit_vector(u[0], u[1]);
    else sample_disk(u[2], u[3], v.x, v.y);
    out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
}

__global__ void philox_kernel(uint32_t seed_lo, uint32_t seed_hi, const uint32_t* ctr4, size_t n, uint32_t* out4) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const uint4 r = philox4x32_10(make_uint4(ctr4[4 * i], ctr4[4 * i + 1], ctr4[4 * i + 2], ctr4[4 * i + 3]), make_uint2(seed_lo, seed_hi));
    out4[4 * i] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
}

// ---- rtk_render_tiles_device_noalloc: every byte of device memory the render kernels need beyond the caller's inputs
// and outputs, in ONE static device array (per GPU, held until the process ends):
//   [0, 256) packet queue | [256, 512) counters | [512, 512 + 8 kNoAllocMaxRuns) run table | pixel accumulators.
constexpr int kNoAllocMaxTiles = RTK_NOALLOC_MAX_TILES;          // 64-pixel tiles of one call (rtk.h)
constexpr int kNoAllocMaxRuns = 1024;                             // runs of a pixel's samples
constexpr size_t kScratchRunsOff = 512;
constexpr size_t kScratchAccumOff = kScratchRunsOff + sizeof(uint2) * kNoAllocMaxRuns;
constexpr size_t kScratchBytes = kScratchAccumOff + static_cast<size_t>(kNoAllocMaxTiles) * kTilePixels * 3 * sizeof(unsigned long long);
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// The run table of (spp, W, H, run_len), written by one thread into the static array (no host table, no copy).
__global__ void runs_kernel(int spp, long long n_pixels, int run_len, uint2* out, int cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) fill_runs(spp, n_pixels, run_len, out, cap);
}

// =================================================================================================
// Host side of the C ABI
// =================================================================================================

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        const hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(RTK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct rtk_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev3 = nullptr;
    hipEvent_t ev_switch = nullptr;                  // orders a new stream behind the old one (rtk_context_set_stream)
    static constexpr int kEventPairs = 256;          // render-kernel brackets of the frames since rtk_reset_stats
    hipEvent_t kev[2 * kEventPairs] = {};
    hipEvent_t kev_static[2] = {};                   // render-kernel bracket of the last rtk_render_tiles_device_noalloc call (its counters are per call)
    int n_frames = 0;                                // frames since the last reset
    int n_cu = 0;
    int blocks_per_cu[2] = {0, 0};   // [LDS]
    int lds_waves = kWavesPerBlock;  // wavefronts per workgroup of render_kernel<true> for the resident scene (4, or 8 for large general-word scenes)
    SceneDev scene = {};
    bool have_scene = false;
    DevBuf filt, wdesc, exact, mat0, mat1, ltiles, lwdesc;
    SceneDev scene_flat = {};   // the same scene in the streamed layout: what the kernels that do not keep it in LDS read (large scenes: == scene; small scenes under the grid)
    DevBuf filt_flat;
    int class_words = 0;     // words of 32 spheres with 8-byte class records (rtk_scene_layout)
    int local_words = 0;     // streamed scenes: words with a tile in their own frame (0: the vector-ALU walk)
    uint64_t last_waves = 0; // wavefronts of the last render launch
    DevBuf runs;         // sample partition table of the current (spp, W, H, run_len)
    std::vector<uint2> runs_host;
    int64_t runs_key[4] = {-1, -1, -1, -1};      // spp, local tiles, -, run_len of the resident table
    DevBuf accum;        // fixed-point accumulators
    size_t accum_zeroed = 0;   // number of accumulators known to be zero
    DevBuf control;      // queue counter (256 B) + stats
    DevBuf scratch_a, scratch_b, scratch_c, scratch_d, scratch_e;  // batch entry points
    DevBuf tile_out;     // for rtk_render (host output)
    DevBuf image_out;
    DevBuf image_u8;     // for rtk_render_u8
    // uniform grid (rtk_context_set_accel)
    int accel = RTK_ACCEL_LIST;
    bool grid_ok = false;            // a grid of the current scene is resident
    GridDev grid = {};
    DevBuf g_cells, g_items, g_always;
    int blocks_per_cu_grid = 1;
    std::vector<float> h_cx, h_cy, h_cz, h_radius;   // host copy of the geometry, for building the grid after rtk_set_scene
};

static bool use_lds(const rtk_context* c) { return c->scene.n <= kMaxLdsSpheres; }
static bool use_grid(const rtk_context* c) { return c->accel == RTK_ACCEL_GRID && c->grid_ok; }


// Builds the uniform grid of the resident scene and uploads it.  A scene a grid cannot hold (build_grid's reasons)
// leaves grid_ok false with the reason in rtk_last_error(): rendering then walks the list as usual.
static int upload_grid(rtk_context* c) {
    c->grid_ok = false;
    if (!c->have_scene) return RTK_OK;
    rtk_scene geo = {};
    geo.n = c->scene.n;
    geo.cx = c->h_cx.data(); geo.cy = c->h_cy.data(); geo.cz = c->h_cz.data(); geo.radius = c->h_radius.data();
    PreparedScene none;
    GridHost g;
    build_grid(&geo, none, g);
    if (!g.usable) {
        fail(RTK_OK, "uniform grid not used: %s", g.why);
        return RTK_OK;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(c->g_cells.reserve(g.cell_start.size() * sizeof(uint32_t)));
    HIP_TRY(c->g_items.reserve(g.items.size() * sizeof(uint32_t)));
    HIP_TRY(c->g_always.reserve(g.always.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(c->g_cells.p, g.cell_start.data(), g.cell_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->g_items.p, g.items.data(), g.items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->g_always.p, g.always.data(), g.always.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->grid = g.dev;
    c->grid.cell_start = static_cast<const uint32_t*>(c->g_cells.p);
    c->grid.items = static_cast<const uint32_t*>(c->g_items.p);
    c->grid.always = static_cast<const uint32_t*>(c->g_always.p);
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<false, true>, kBlockThreads, lds_bytes(false, c->scene)));
    c->blocks_per_cu_grid = nb > 0 ? nb : 1;
    c->grid_ok = true;
    return RTK_OK;
}

// Counters of the render kernels since they were last cleared (`d_counters`: 4 u64 in device memory) and the HIP-event
// times of the context's frames.  Synchronises the context's stream.
// `static_call`: the counters are those of ONE rtk_render_tiles_device_noalloc call (static workspace), timed by kev_static; the
// context's own frame counter and event ring (the allocating path's frames since rtk_reset_stats) are left alone.
static int read_stats(rtk_context* c, rtk_stats* stats, const unsigned char* d_counters, bool static_call = false) {
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[8];
    HIP_TRY(hipMemcpyAsync(h, d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    stats->rays = h[0];
    stats->samples = h[1];
    // list walk: every ray meets every sphere; grid: only the exact tests of the cells it crosses
    stats->sphere_tests = use_grid(c) ? h[2] : h[0] * static_cast<uint64_t>(c->scene.n);
    stats->exact_tests = h[2];
    stats->lane_slots = h[3] * kWave;
    stats->max_wave_passes = h[4];
    stats->waves = c->last_waves;
    {
        const unsigned long long t0 = ~h[5], t_first = ~h[6], t_last = h[7];
        stats->drain_frac = (h[5] != 0 && t_last > t0 && t_first >= t0) ? static_cast<float>(static_cast<double>(t_last - t_first) / static_cast<double>(t_last - t0)) : 0.0f;
        stats->reserved_ = 0.0f;
    }
    stats->kernel_ms = 0.0f;
    stats->total_ms = 0.0f;
    if (static_call) {
        stats->frames = 1;
        stats->frames_timed = 1;
        HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, c->kev_static[0], c->kev_static[1]));
        return RTK_OK;
    }
    stats->frames = static_cast<uint32_t>(c->n_frames);
    const int timed = c->n_frames < rtk_context::kEventPairs ? c->n_frames : rtk_context::kEventPairs;
    for (int i = 0; i < timed; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->kev[2 * i], c->kev[2 * i + 1]));
        stats->kernel_ms += ms;
    }
    stats->frames_timed = static_cast<uint32_t>(timed);
    return RTK_OK;
}

extern "C" {

int rtk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* rtk_last_error(void) { return g_err; }

void rtk_context_destroy(rtk_context* c);

int rtk_context_create(int device, rtk_context** out) {
    if (!out) return fail(RTK_ERR_BAD_ARG, "rtk_context_create: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RTK_ERR_NO_DEVICE, "no HIP device is visible; librtk has no CPU fallback");
    if (device < 0 || device >= n) return fail(RTK_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    rtk_context* c = new (std::nothrow) rtk_context();
    if (!c) return fail(RTK_ERR_HIP, "out of host memory");
    c->device = device;
    // everything below goes through init(): on any failure the half-built context is destroyed, nothing leaks
    const auto init = [c, device]() -> int {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(RTK_ERR_NO_DEVICE, "device %d is %s; librtk is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        c->n_cu = prop.multiProcessorCount;
        HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev3));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming));
        for (hipEvent_t& e : c->kev) HIP_TRY(hipEventCreate(&e));
        for (hipEvent_t& e : c->kev_static) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(c->control.reserve(512));
        HIP_TRY(hipMemset(c->control.p, 0, 512));
        return RTK_OK;
    };
    if (const int e = init()) {
        rtk_context_destroy(c);
        return e;
    }
    *out = c;
    return RTK_OK;
}

void rtk_context_destroy(rtk_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->filt, &c->filt_flat, &c->wdesc, &c->ltiles, &c->lwdesc, &c->exact, &c->mat0, &c->mat1, &c->runs, &c->accum, &c->control, &c->scratch_a, &c->scratch_b, &c->scratch_c,
                      &c->scratch_d, &c->scratch_e, &c->tile_out, &c->image_out, &c->image_u8, &c->g_cells, &c->g_items, &c->g_always})
        b->release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
    for (hipEvent_t e : c->kev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->kev_static)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int rtk_context_set_stream(rtk_context* c, void* hip_stream) {
    if (!c) return fail(RTK_ERR_BAD_ARG, "null context");
    hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    if (next != c->stream) {
        // the context's buffers (accumulators, queue, run table) may still be in use by work queued on the old stream:
        // whatever is submitted to the new stream from now on is ordered behind the last such work.  The OLD stream is not
        // touched here — it may be the caller's and already destroyed; ev_switch was recorded on it when that work was queued
        // (render_tiles_impl), and a wait on an event that was never recorded returns at once.
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamWaitEvent(next, c->ev_switch, 0));
        c->stream = next;
    }
    return RTK_OK;
}

int rtk_synchronize(rtk_context* c) {
    if (!c) return fail(RTK_ERR_BAD_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTK_OK;
}

int rtk_set_scene(rtk_context* c, const rtk_scene* s) {
    if (!c || !s) return fail(RTK_ERR_BAD_ARG, "rtk_set_scene: null argument");
    if (s->n <= 0) return fail(RTK_ERR_BAD_ARG, "rtk_set_scene: scene has %d spheres", s->n);
    if (s->n > (1 << 24)) return fail(RTK_ERR_TOO_LARGE, "rtk_set_scene: %d spheres exceed the limit of 2^24", s->n);
    if (!s->cx || !s->cy || !s->cz || !s->radius || !s->mat_r || !s->mat_g || !s->mat_b || !s->roughness || !s->metallic || !s->ior || !s->mat_type)
        return fail(RTK_ERR_BAD_ARG, "rtk_set_scene: null array");
    HIP_TRY(hipSetDevice(c->device));
    PreparedScene ps;
    const bool compact = s->n <= kMaxLdsSpheres;       // LDS-resident scenes get word descriptors and 8-byte class records
    prepare_scene(s, ps, compact);
    const int n = ps.n, npad = ps.npad;
    const std::vector<float4>&filt = ps.filt, &exact = ps.exact, &m0 = ps.m0, &m1 = ps.m1;
    const FilterBounds fb = ps.fb;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(c->filt.reserve(sizeof(float4) * filt.size()));
    HIP_TRY(c->exact.reserve(sizeof(float4) * npad));
    HIP_TRY(c->mat0.reserve(sizeof(float4) * n));
    HIP_TRY(c->mat1.reserve(sizeof(float4) * n));
    HIP_TRY(hipMemcpy(c->filt.p, filt.data(), sizeof(float4) * filt.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->exact.p, exact.data(), sizeof(float4) * npad, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->mat0.p, m0.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->mat1.p, m1.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    c->scene = {static_cast<const float4*>(c->filt.p), static_cast<const float4*>(c->exact.p), static_cast<const float4*>(c->mat0.p),
                static_cast<const float4*>(c->mat1.p), n, npad, fb};
    c->scene.filt_len = static_cast<int32_t>(filt.size());
    c->scene.filt_lds = ps.filt_lds;
    c->class_words = ps.class_words;
    if (compact) {
        if (ps.wdesc.size() != static_cast<size_t>(npad / 32)) return fail(RTK_ERR_HIP, "internal: word descriptor count");
        HIP_TRY(c->wdesc.reserve(sizeof(float4) * ps.wdesc.size()));
        HIP_TRY(hipMemcpy(c->wdesc.p, ps.wdesc.data(), sizeof(float4) * ps.wdesc.size(), hipMemcpyHostToDevice));
        c->scene.wdesc = static_cast<const float4*>(c->wdesc.p);
    }
    c->local_words = 0;
    // streamed scene of class words: the matrix-pipe filter in word-local frames (the variable: A/B measurements against the staged vector-ALU walk)
    if (ps.use_local && !std::getenv("RTK_NO_LOCAL_FRAMES")) {
        HIP_TRY(c->ltiles.reserve(sizeof(float4) * ps.ltiles.size()));
        HIP_TRY(c->lwdesc.reserve(sizeof(float4) * ps.lwdesc.size()));
        HIP_TRY(hipMemcpy(c->ltiles.p, ps.ltiles.data(), sizeof(float4) * ps.ltiles.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->lwdesc.p, ps.lwdesc.data(), sizeof(float4) * ps.lwdesc.size(), hipMemcpyHostToDevice));
        c->scene.ltiles = static_cast<const float4*>(c->ltiles.p);
        c->scene.lwdesc = static_cast<const float4*>(c->lwdesc.p);
        c->scene.lwords = ps.local_words;
        c->local_words = ps.local_class_words;
    }
    c->scene_flat = c->scene;
    if (compact) {
        // the grid kernels do not keep the scene in LDS; their fallback for rays the grid cannot serve streams the list
        PreparedScene flat;
        prepare_scene(s, flat, false);
        HIP_TRY(c->filt_flat.reserve(sizeof(float4) * flat.filt.size()));
        HIP_TRY(hipMemcpy(c->filt_flat.p, flat.filt.data(), sizeof(float4) * flat.filt.size(), hipMemcpyHostToDevice));
        c->scene_flat.filt = static_cast<const float4*>(c->filt_flat.p);
        c->scene_flat.filt_len = static_cast<int32_t>(flat.filt.size());
        c->scene_flat.wdesc = nullptr;
        c->scene_flat.fb = flat.fb;
    }
    c->have_scene = true;
    c->h_cx.assign(s->cx, s->cx + n); c->h_cy.assign(s->cy, s->cy + n); c->h_cz.assign(s->cz, s->cz + n); c->h_radius.assign(s->radius, s->radius + n);
    c->grid_ok = false;
    if (c->accel == RTK_ACCEL_GRID)
        if (int e = upload_grid(c)) return e;
    // persistent-grid sizing for both kernel flavours at this scene size
    for (int l = 0; l < 2; ++l) {
        int nb = 0;
        const size_t shm = lds_bytes(l == 1, c->scene);
        if (l == 1 && !use_lds(c)) { c->blocks_per_cu[l] = 0; continue; }
        if (l == 1) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm)));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&intersect_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm)));
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<true>, kBlockThreads, shm));
            c->lds_waves = kWavesPerBlock;
            // one 4-wave workgroup per CU would leave each SIMD a single wave: take 8-wave workgroups when they fit
            const size_t shm8 = lds_bytes(true, c->scene, 2 * kWavesPerBlock);
            if (nb <= 1 && shm8 + 1024 <= 160 * 1024) {
                int nb8 = 0;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<true, false, 2 * kWavesPerBlock>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm8)));
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb8, render_kernel<true, false, 2 * kWavesPerBlock>, 2 * kBlockThreads, shm8));
                if (nb8 >= 1) { c->lds_waves = 2 * kWavesPerBlock; nb = nb8; }
            }
        } else {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<false>, kBlockThreads, shm));
        }
        c->blocks_per_cu[l] = nb > 0 ? nb : 1;
        if (const char* e = std::getenv("RTK_EXP_BLOCKS_PER_CU"))          // occupancy studies only (tools/): fewer resident workgroups per CU
            if (std::atoi(e) > 0 && std::atoi(e) < c->blocks_per_cu[l]) c->blocks_per_cu[l] = std::atoi(e);
    }
    return RTK_OK;
}

int rtk_context_set_accel(rtk_context* c, int32_t mode) {
    if (!c) return fail(RTK_ERR_BAD_ARG, "null context");
    if (mode != RTK_ACCEL_LIST && mode != RTK_ACCEL_GRID) return fail(RTK_ERR_BAD_ARG, "rtk_context_set_accel: unknown mode %d", mode);
    HIP_TRY(hipSetDevice(c->device));
    c->accel = mode;
    g_err[0] = 0;
    if (mode == RTK_ACCEL_GRID && !c->grid_ok) return upload_grid(c);
    return RTK_OK;
}

int rtk_accel_in_use(rtk_context* c) { return (c && use_grid(c)) ? RTK_ACCEL_GRID : RTK_ACCEL_LIST; }

int rtk_scene_layout(rtk_context* c, int32_t* out4) {
    if (!c || !out4) return fail(RTK_ERR_BAD_ARG, "rtk_scene_layout: null argument");
    if (!c->have_scene) return fail(RTK_ERR_NO_SCENE, "rtk_scene_layout: call rtk_set_scene first");
    const bool lds = use_lds(c) && !use_grid(c);
    out4[0] = c->scene.n;
    out4[1] = use_lds(c) ? c->class_words : c->local_words;
    out4[2] = lds ? 1 : 0;
    out4[3] = static_cast<int32_t>(lds_bytes(lds, c->scene, lds ? c->lds_waves : kWavesPerBlock) + sizeof(CameraDev) + 8);     // dynamic + the kernels' static LDS (camera copy)
    return RTK_OK;
}

static int norm_tiles(const rtk_params* p, int& tw, int& th) {
    tw = p->tile_w;
    th = p->tile_h;
    if (tw == 0 && th == 0) { tw = 8; th = 8; }
    if (tw <= 0 || th <= 0 || tw * th != kTilePixels) return fail(RTK_ERR_BAD_ARG, "tile %d x %d: a tile must hold exactly 64 pixels", tw, th);
    return RTK_OK;
}

int64_t rtk_local_tile_count(int32_t W, int32_t H, int32_t tile_w, int32_t tile_h, int32_t rank, int32_t world) {
    if (tile_w == 0 && tile_h == 0) { tile_w = 8; tile_h = 8; }
    if (W <= 0 || H <= 0 || tile_w <= 0 || tile_h <= 0 || world <= 0 || rank < 0 || rank >= world) return -1;
    const int64_t tiles = static_cast<int64_t>((W + tile_w - 1) / tile_w) * ((H + tile_h - 1) / tile_h);
    return (tiles - rank + world - 1) / world;
}

int rtk_run_schedule(int32_t spp, int32_t width, int32_t height, int32_t run_len, uint32_t* out_pairs, int32_t cap) {
    if (spp <= 0 || width <= 0 || height <= 0 || run_len < -4096) return fail(RTK_ERR_BAD_ARG, "rtk_run_schedule: spp %d, image %d x %d, run_len %d", spp, width, height, run_len);
    const std::vector<uint2> runs = run_schedule(spp, static_cast<long long>(width) * height, run_len);
    const int n = static_cast<int>(runs.size());
    if (out_pairs) {
        if (cap < n) return fail(RTK_ERR_BAD_ARG, "rtk_run_schedule: %d runs do not fit in %d pairs", n, cap);
        for (int i = 0; i < n; ++i) { out_pairs[2 * i] = runs[i].x; out_pairs[2 * i + 1] = runs[i].y; }
    }
    return n;
}

// `static_ws`: the device address of this GPU's `scratch` array (rtk_render_tiles_device_noalloc) or null (the context's
// own buffers, grown with hipMalloc as needed).
static int render_tiles_impl(rtk_context* c, const rtk_camera* cam, const rtk_params* p, float* d_tile_sums, rtk_stats* stats, unsigned char* static_ws) {
    if (!c || !cam || !p || !d_tile_sums) return fail(RTK_ERR_BAD_ARG, "rtk_render_tiles_device: null argument");
    if (!c->have_scene) return fail(RTK_ERR_NO_SCENE, "rtk_render_tiles_device: call rtk_set_scene first");
    if (p->width < 2 || p->height < 2) return fail(RTK_ERR_BAD_ARG, "image %d x %d: the reference divides by (W-1) and (H-1), main.cpp:69-70", p->width, p->height);
    if (p->spp <= 0) return fail(RTK_ERR_BAD_ARG, "spp must be positive, got %d", p->spp);
    if (p->sample_offset < 0 || static_cast<int64_t>(p->sample_offset) + p->spp > 0x7fffffffll) return fail(RTK_ERR_BAD_ARG, "sample_offset %d with spp %d", p->sample_offset, p->spp);
    const int world = p->world_size > 0 ? p->world_size : 1;
    if (p->rank < 0 || p->rank >= world) return fail(RTK_ERR_BAD_ARG, "rank %d of %d", p->rank, world);
    int tw, th;
    if (int e = norm_tiles(p, tw, th)) return e;
    if (static_cast<int64_t>(p->width) * p->height >= (1ll << 31)) return fail(RTK_ERR_TOO_LARGE, "image has more than 2^31 pixels");
    if (p->width > 65535 || p->height > 65535) return fail(RTK_ERR_TOO_LARGE, "image %d x %d: width and height are limited to 65535 (the kernel packs a pixel's column and row into one register)", p->width, p->height);
    HIP_TRY(hipSetDevice(c->device));

    const int tiles_x = (p->width + tw - 1) / tw;
    const int64_t nlt = rtk_local_tile_count(p->width, p->height, tw, th, p->rank, world);
    if (p->run_len < -4096) return fail(RTK_ERR_BAD_ARG, "run_len must be >= -4096, got %d", p->run_len);
    int n_runs = 0;
    const long long n_local_px = static_cast<long long>(nlt) * kTilePixels;      // the partition is sized for this rank's share
    if (static_ws) {
        n_runs = fill_runs(p->spp, n_local_px, p->run_len, nullptr, 0);
        if (n_runs > kNoAllocMaxRuns) return fail(RTK_ERR_TOO_LARGE, "rtk_render_tiles_device_noalloc: %d sample runs exceed the static table of %d (use fewer spp per call or a larger run_len)", n_runs, kNoAllocMaxRuns);
    } else {
        if (c->runs_key[0] != p->spp || c->runs_key[1] != nlt || c->runs_key[3] != p->run_len) {
            // a new sample partition: upload its table (the stream may still be reading the old one)
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->runs_host = run_schedule(p->spp, n_local_px, p->run_len);
            HIP_TRY(c->runs.reserve(c->runs_host.size() * sizeof(uint2)));
            HIP_TRY(hipMemcpy(c->runs.p, c->runs_host.data(), c->runs_host.size() * sizeof(uint2), hipMemcpyHostToDevice));
            c->runs_key[0] = p->spp; c->runs_key[1] = nlt; c->runs_key[2] = 0; c->runs_key[3] = p->run_len;
        }
        n_runs = static_cast<int>(c->runs_host.size());
    }
    const int64_t packets = nlt * n_runs;
    if (packets >= (1ll << 32) - (1ll << 20)) return fail(RTK_ERR_TOO_LARGE, "%lld packets exceed the 32-bit queue", static_cast<long long>(packets));
    const size_t n_acc = static_cast<size_t>(nlt) * kTilePixels * 3;
    if (nlt == 0) return RTK_OK;

    unsigned long long* accum = nullptr;
    unsigned char* control = nullptr;
    const uint2* runs = nullptr;
    if (static_ws) {
        if (nlt > kNoAllocMaxTiles) return fail(RTK_ERR_TOO_LARGE, "rtk_render_tiles_device_noalloc: %lld tiles exceed the static limit of %d (RTK_NOALLOC_MAX_TILES, %d pixels)",
                                                static_cast<long long>(nlt), kNoAllocMaxTiles, kNoAllocMaxTiles * kTilePixels);
        // the array keeps the last call's contents (any context, any size): clear what must start at zero
        control = static_ws;
        accum = reinterpret_cast<unsigned long long*>(static_ws + kScratchAccumOff);
        uint2* rt = reinterpret_cast<uint2*>(static_ws + kScratchRunsOff);
        runs = rt;
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        HIP_TRY(hipMemsetAsync(accum, 0, n_acc * sizeof(unsigned long long), c->stream));
        HIP_TRY(hipMemsetAsync(control, 0, 512, c->stream));
        hipLaunchKernelGGL(runs_kernel, dim3(1), dim3(64), 0, c->stream, p->spp, n_local_px, p->run_len, rt, kNoAllocMaxRuns);
        HIP_TRY(hipGetLastError());
    } else {
        if (n_acc * 8 > c->accum.cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(c->accum.reserve(n_acc * 8));
            c->accum_zeroed = 0;
        }
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        if (c->accum_zeroed < n_acc) {
            HIP_TRY(hipMemsetAsync(c->accum.p, 0, c->accum.cap, c->stream));
            c->accum_zeroed = c->accum.cap / 8;
        }
        if (stats) {
            if (int e = rtk_reset_stats(c)) return e;
        }
        HIP_TRY(hipMemsetAsync(c->control.p, 0, 256, c->stream));   // the packet queue; the counters behind it accumulate
        accum = static_cast<unsigned long long*>(c->accum.p);
        control = static_cast<unsigned char*>(c->control.p);
        runs = static_cast<const uint2*>(c->runs.p);
    }

    RenderArgs A;
    A.scene = (use_lds(c) && !use_grid(c)) ? c->scene : c->scene_flat;
    std::memcpy(&A.cam, cam, sizeof(float) * 22);
    A.W = p->width; A.H = p->height; A.spp = p->spp; A.max_depth = p->max_depth;
    A.sample_offset = static_cast<uint32_t>(p->sample_offset);
    A.seed_lo = static_cast<uint32_t>(p->seed); A.seed_hi = static_cast<uint32_t>(p->seed >> 32);
    A.t_min = p->t_min;
    A.tile_w = tw; A.tile_h = th; A.tiles_x = tiles_x;
    A.tile_w_log2 = 0;
    while ((1 << A.tile_w_log2) < tw) ++A.tile_w_log2;
    A.rank = p->rank; A.world = world;
    A.n_local_tiles = static_cast<int32_t>(nlt);
    A.runs = runs; A.n_runs = n_runs;
    A.total_packets = static_cast<uint32_t>(packets);
    A.accum = accum;
    A.queue = reinterpret_cast<uint32_t*>(control);
    A.stats = reinterpret_cast<unsigned long long*>(control + 256);

    const bool accel = use_grid(c);
    const bool lds = use_lds(c) && !accel;
    const int waves = lds ? c->lds_waves : kWavesPerBlock;
    const size_t shm = lds_bytes(lds, c->scene, waves);
    A.grid = c->grid;
    int64_t grid = static_cast<int64_t>(c->n_cu) * (accel ? c->blocks_per_cu_grid : c->blocks_per_cu[lds ? 1 : 0]);
    const int64_t need_blocks = (packets + waves - 1) / waves;
    if (grid > need_blocks) grid = need_blocks;
    if (grid < 1) grid = 1;
    c->last_waves = static_cast<uint64_t>(grid) * waves;
    const int slot = c->n_frames % rtk_context::kEventPairs;
    hipEvent_t k_begin = static_ws ? c->kev_static[0] : c->kev[2 * slot], k_end = static_ws ? c->kev_static[1] : c->kev[2 * slot + 1];
    HIP_TRY(hipEventRecord(k_begin, c->stream));
    if (accel)
        hipLaunchKernelGGL((render_kernel<false, true>), dim3(static_cast<unsigned>(grid)), dim3(kBlockThreads), shm, c->stream, A);
    else if (lds && waves == 2 * kWavesPerBlock)
        hipLaunchKernelGGL((render_kernel<true, false, 2 * kWavesPerBlock>), dim3(static_cast<unsigned>(grid)), dim3(2 * kBlockThreads), shm, c->stream, A);
    else if (lds)
        hipLaunchKernelGGL(render_kernel<true>, dim3(static_cast<unsigned>(grid)), dim3(kBlockThreads), shm, c->stream, A);
    else
        hipLaunchKernelGGL(render_kernel<false>, dim3(static_cast<unsigned>(grid)), dim3(kBlockThreads), shm, c->stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(k_end, c->stream));
    if (!static_ws) ++c->n_frames;
    const unsigned fb = static_cast<unsigned>((n_acc + 255) / 256);
    hipLaunchKernelGGL(finalize_kernel, dim3(fb), dim3(256), 0, c->stream, accum, d_tile_sums, n_acc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev3, c->stream));
    HIP_TRY(hipEventRecord(c->ev_switch, c->stream));      // what a later stream switch has to wait for (rtk_context_set_stream)
    if (stats) {
        if (int e = read_stats(c, stats, control + 256, static_ws != nullptr)) return e;
        HIP_TRY(hipEventElapsedTime(&stats->total_ms, c->ev0, c->ev3));
    }
    return RTK_OK;
}

int rtk_render_tiles_device(rtk_context* c, const rtk_camera* cam, const rtk_params* p, float* d_tile_sums, rtk_stats* stats) {
    return render_tiles_impl(c, cam, p, d_tile_sums, stats, nullptr);
}

// Device address of this GPU's copy of `scratch`, looked up once per device with that device current.  In host code the
// array's NAME is only a stand-in for the symbol: the address the kernels can use comes from hipGetSymbolAddress.
static constexpr int kMaxDevices = 64;
static unsigned char* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

static int scratch_for_device(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(RTK_ERR_BAD_ARG, "device %d beyond the %d the static workspace table holds", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        HIP_TRY(hipSetDevice(device));
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RTK_ERR_HIP, "hipGetSymbolAddress returned a null address for the static workspace");
        g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return RTK_OK;
}

int rtk_render_tiles_device_noalloc(rtk_context* c, const rtk_camera* cam, const rtk_params* p, float* d_tile_sums, rtk_stats* stats) {
    if (!c) return fail(RTK_ERR_BAD_ARG, "rtk_render_tiles_device_noalloc: null context");
    unsigned char* ws = nullptr;
    if (int e = scratch_for_device(c->device, &ws)) return e;
    return render_tiles_impl(c, cam, p, d_tile_sums, stats, ws);
}

int rtk_reset_stats(rtk_context* c) {
    if (!c) return fail(RTK_ERR_BAD_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(static_cast<char*>(c->control.p) + 256, 0, 256, c->stream));
    c->n_frames = 0;
    return RTK_OK;
}

int rtk_get_stats(rtk_context* c, rtk_stats* stats) {
    if (!c || !stats) return fail(RTK_ERR_BAD_ARG, "rtk_get_stats: null argument");
    return read_stats(c, stats, static_cast<unsigned char*>(c->control.p) + 256);
}

extern "C++" template <typename T>
static int deinterleave_any(rtk_context* c, const T* d_gathered, int64_t tprp, const rtk_params* p, T* d_image, const char* who) {
    if (!c || !d_gathered || !p || !d_image) return fail(RTK_ERR_BAD_ARG, "%s: null argument", who);
    int tw, th;
    if (int e = norm_tiles(p, tw, th)) return e;
    const int world = p->world_size > 0 ? p->world_size : 1;
    if (p->width <= 0 || p->height <= 0) return fail(RTK_ERR_BAD_ARG, "image %d x %d", p->width, p->height);
    if (tprp < rtk_local_tile_count(p->width, p->height, tw, th, 0, world)) return fail(RTK_ERR_BAD_ARG, "tiles_per_rank_padded %lld is too small", static_cast<long long>(tprp));
    HIP_TRY(hipSetDevice(c->device));
    const long long np = static_cast<long long>(p->width) * p->height;
    hipLaunchKernelGGL(deinterleave_kernel<T>, dim3(static_cast<unsigned>((np + 255) / 256)), dim3(256), 0, c->stream, d_gathered, static_cast<long long>(tprp),
                       p->width, p->height, tw, th, (p->width + tw - 1) / tw, world, d_image);
    HIP_TRY(hipGetLastError());
    return RTK_OK;
}

int rtk_deinterleave_device(rtk_context* c, const float* d_gathered, int64_t tprp, const rtk_params* p, float* d_image) {
    return deinterleave_any<float>(c, d_gathered, tprp, p, d_image, "rtk_deinterleave_device");
}

int rtk_deinterleave_u8_device(rtk_context* c, const uint8_t* d_gathered, int64_t tprp, const rtk_params* p, uint8_t* d_image) {
    return deinterleave_any<uint8_t>(c, d_gathered, tprp, p, d_image, "rtk_deinterleave_u8_device");
}

int rtk_render(rtk_context* c, const rtk_camera* cam, const rtk_params* p, float* out, rtk_stats* stats) {
    if (!c || !cam || !p || !out) return fail(RTK_ERR_BAD_ARG, "rtk_render: null argument");
    if (p->world_size > 1 || p->rank != 0) return fail(RTK_ERR_BAD_ARG, "rtk_render renders the whole image: rank/world_size must be 0/1");
    int tw, th;
    if (int e = norm_tiles(p, tw, th)) return e;
    if (p->width < 2 || p->height < 2) return fail(RTK_ERR_BAD_ARG, "image %d x %d: the reference divides by (W-1) and (H-1), main.cpp:69-70", p->width, p->height);
    HIP_TRY(hipSetDevice(c->device));
    rtk_params q = *p;
    q.world_size = 1;
    q.rank = 0;
    const int64_t nlt = rtk_local_tile_count(p->width, p->height, tw, th, 0, 1);
    const size_t np = static_cast<size_t>(p->width) * p->height;
    HIP_TRY(c->tile_out.reserve(static_cast<size_t>(nlt) * kTilePixels * 3 * sizeof(float)));
    HIP_TRY(c->image_out.reserve(np * 3 * sizeof(float)));
    rtk_stats st;
    if (int e = rtk_render_tiles_device(c, cam, &q, static_cast<float*>(c->tile_out.p), &st)) return e;
    if (int e = rtk_deinterleave_device(c, static_cast<const float*>(c->tile_out.p), nlt, &q, static_cast<float*>(c->image_out.p))) return e;
    HIP_TRY(hipMemcpyAsync(out, c->image_out.p, np * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats) *stats = st;
    return RTK_OK;
}

int rtk_quantise_device(rtk_context* c, const float* d_sum, int64_t n_pixels, int32_t spp, uint8_t* d_u8);

int rtk_render_u8(rtk_context* c, const rtk_camera* cam, const rtk_params* p, uint8_t* out, rtk_stats* stats) {
    if (!c || !cam || !p || !out) return fail(RTK_ERR_BAD_ARG, "rtk_render_u8: null argument");
    if (p->world_size > 1 || p->rank != 0) return fail(RTK_ERR_BAD_ARG, "rtk_render_u8 renders the whole image: rank/world_size must be 0/1");
    if (p->sample_offset != 0) return fail(RTK_ERR_BAD_ARG, "rtk_render_u8 quantises a complete frame: sample_offset must be 0 (sum progressive batches with rtk_render)");
    int tw, th;
    if (int e = norm_tiles(p, tw, th)) return e;
    if (p->width < 2 || p->height < 2) return fail(RTK_ERR_BAD_ARG, "image %d x %d: the reference divides by (W-1) and (H-1), main.cpp:69-70", p->width, p->height);
    HIP_TRY(hipSetDevice(c->device));
    rtk_params q = *p;
    q.world_size = 1;
    q.rank = 0;
    const int64_t nlt = rtk_local_tile_count(p->width, p->height, tw, th, 0, 1);
    const size_t np = static_cast<size_t>(p->width) * p->height;
    HIP_TRY(c->tile_out.reserve(static_cast<size_t>(nlt) * kTilePixels * 3 * sizeof(float)));
    HIP_TRY(c->image_out.reserve(np * 3 * sizeof(float)));
    HIP_TRY(c->image_u8.reserve(np * 3));
    rtk_stats st;
    if (int e = rtk_render_tiles_device(c, cam, &q, static_cast<float*>(c->tile_out.p), &st)) return e;
    if (int e = rtk_deinterleave_device(c, static_cast<const float*>(c->tile_out.p), nlt, &q, static_cast<float*>(c->image_out.p))) return e;
    if (int e = rtk_quantise_device(c, static_cast<const float*>(c->image_out.p), static_cast<int64_t>(np), p->spp, static_cast<uint8_t*>(c->image_u8.p))) return e;
    HIP_TRY(hipMemcpyAsync(out, c->image_u8.p, np * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats) *stats = st;
    return RTK_OK;
}

int rtk_quantise_device(rtk_context* c, const float* d_sum, int64_t n_pixels, int32_t spp, uint8_t* d_u8) {
    if (!c || !d_sum || !d_u8) return fail(RTK_ERR_BAD_ARG, "rtk_quantise_device: null argument");
    if (n_pixels <= 0 || spp <= 0) return fail(RTK_ERR_BAD_ARG, "rtk_quantise_device: n_pixels %lld, spp %d", static_cast<long long>(n_pixels), spp);
    HIP_TRY(hipSetDevice(c->device));
    const long long n = n_pixels * 3;
    hipLaunchKernelGGL(quantise_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, c->stream, d_sum, n, spp, d_u8);
    HIP_TRY(hipGetLastError());
    return RTK_OK;
}

int rtk_intersect_batch_device(rtk_context* c, const float* d_rays, size_t n, float t_min, float t_max, uint8_t* d_hit, int32_t* d_index, float* d_t) {
    if (!c || !d_rays || !d_hit || !d_index || !d_t) return fail(RTK_ERR_BAD_ARG, "rtk_intersect_batch_device: null argument");
    if (!c->have_scene) return fail(RTK_ERR_NO_SCENE, "rtk_intersect_batch: call rtk_set_scene first");
    if (n == 0) return RTK_OK;
    HIP_TRY(hipSetDevice(c->device));
    const bool accel = use_grid(c);
    const bool lds = use_lds(c) && !accel;
    const size_t shm = lds_bytes(lds, c->scene);
    size_t grid = (n + kBlockThreads - 1) / kBlockThreads;
    const int bpc = accel ? c->blocks_per_cu_grid : c->blocks_per_cu[lds ? 1 : 0];
    const size_t cap = static_cast<size_t>(c->n_cu) * (bpc > 0 ? bpc : 1);
    if (grid > cap) grid = cap;
    if (accel)
        hipLaunchKernelGGL((intersect_kernel<false, true>), dim3(static_cast<unsigned>(grid)), dim3(kBlockThreads), shm, c->stream, c->scene_flat, d_rays, n, t_min, t_max, d_hit, d_index, d_t, c->grid);
    else if (lds)
        hipLaunchKernelGGL(intersect_kernel<true>, dim3(static_cast<unsigned>(grid)), dim3(kBlockThreads), shm, c->stream, c->scene, d_rays, n, t_min, t_max, d_hit, d_index, d_t, c->grid);
    else
        hipLaunchKernelGGL(intersect_kernel<false>, dim3(static_cast<unsigned>(grid)), dim3(kBlockThreads), shm, c->stream, c->scene_flat, d_rays, n, t_min, t_max, d_hit, d_index, d_t, c->grid);
    HIP_TRY(hipGetLastError());
    return RTK_OK;
}

int rtk_intersect_batch(rtk_context* c, const float* rays, size_t n, float t_min, float t_max, uint8_t* out_hit, int32_t* out_index, float* out_t) {
    if (!c || !out_hit || !out_index || !out_t || (!rays && n)) return fail(RTK_ERR_BAD_ARG, "rtk_intersect_batch: null argument");
    if (!c->have_scene) return fail(RTK_ERR_NO_SCENE, "rtk_intersect_batch: call rtk_set_scene first");
    if (n == 0) return RTK_OK;   // empty batch: nothing to do
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->scratch_a.reserve(n * 6 * sizeof(float)));
    HIP_TRY(c->scratch_b.reserve(n));
    HIP_TRY(c->scratch_c.reserve(n * sizeof(int32_t)));
    HIP_TRY(c->scratch_d.reserve(n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->scratch_a.p, rays, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (int e = rtk_intersect_batch_device(c, static_cast<const float*>(c->scratch_a.p), n, t_min, t_max, static_cast<uint8_t*>(c->scratch_b.p),
                                           static_cast<int32_t*>(c->scratch_c.p), static_cast<float*>(c->scratch_d.p)))
        return e;
    HIP_TRY(hipMemcpyAsync(out_hit, c->scratch_b.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_index, c->scratch_c.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_t, c->scratch_d.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTK_OK;
}

int rtk_scatter_batch(rtk_context* c, const float* rays, const int32_t* index, const float* t, const float* u4, size_t n, float* out7) {
    if (!c || !out7 || ((!rays || !index || !t || !u4) && n)) return fail(RTK_ERR_BAD_ARG, "rtk_scatter_batch: null argument");
    if (!c->have_scene) return fail(RTK_ERR_NO_SCENE, "rtk_scatter_batch: call rtk_set_scene first");
    if (n == 0) return RTK_OK;
    for (size_t i = 0; i < n; ++i)
        if (index[i] < 0 || index[i] >= c->scene.n) return fail(RTK_ERR_BAD_ARG, "rtk_scatter_batch: index[%zu] = %d outside the scene (%d spheres)", i, index[i], c->scene.n);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->scratch_a.reserve(n * 6 * sizeof(float)));
    HIP_TRY(c->scratch_b.reserve(n * sizeof(int32_t)));
    HIP_TRY(c->scratch_c.reserve(n * sizeof(float)));
    HIP_TRY(c->scratch_d.reserve(n * 4 * sizeof(float)));
    HIP_TRY(c->scratch_e.reserve(n * 7 * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->scratch_a.p, rays, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_b.p, index, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_c.p, t, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->scratch_d.p, u4, n * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(scatter_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, c->stream, c->scene, static_cast<const float*>(c->scratch_a.p),
                       static_cast<const int32_t*>(c->scratch_b.p), static_cast<const float*>(c->scratch_c.p), static_cast<const float*>(c->scratch_d.p), n,
                       static_cast<float*>(c->scratch_e.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out7, c->scratch_e.p, n * 7 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTK_OK;
}

int rtk_primary_rays(rtk_context* c, const rtk_camera* cam, const rtk_params* p, const int32_t* xys, size_t n, float* out_rays) {
    if (!c || !cam || !p || !out_rays || (!xys && n)) return fail(RTK_ERR_BAD_ARG, "rtk_primary_rays: null argument");
    if (p->width < 2 || p->height < 2) return fail(RTK_ERR_BAD_ARG, "image %d x %d", p->width, p->height);
    if (n == 0) return RTK_OK;
    for (size_t i = 0; i < n; ++i)
        if (xys[3 * i] < 0 || xys[3 * i] >= p->width || xys[3 * i + 1] < 0 || xys[3 * i + 1] >= p->height || xys[3 * i + 2] < 0)
            return fail(RTK_ERR_BAD_ARG, "rtk_primary_rays: triple %zu outside the image", i);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->scratch_a.reserve(n * 3 * sizeof(int32_t)));
    HIP_TRY(c->scratch_e.reserve(n * 6 * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->scratch_a.p, xys, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CameraDev cd;
    std::memcpy(&cd, cam, sizeof(float) * 22);
    hipLaunchKernelGGL(primary_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, c->stream, cd, p->width, p->height, static_cast<uint32_t>(p->seed),
                       static_cast<uint32_t>(p->seed >> 32), static_cast<const int32_t*>(c->scratch_a.p), n, static_cast<float*>(c->scratch_e.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rays, c->scratch_e.p, n * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTK_OK;
}

int rtk_sample_batch(rtk_context* c, int32_t kind, const float* u4, size_t n, float* out3) {
    if (!c || !out3 || (!u4 && n)) return fail(RTK_ERR_BAD_ARG, "rtk_sample_batch: null argument");
    if (kind < 0 || kind > 2) return fail(RTK_ERR_BAD_ARG, "rtk_sample_batch: kind %d (0 ball, 1 sphere, 2 disk)", kind);
    if (n == 0) return RTK_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->scratch_a.reserve(n * 16));
    HIP_TRY(c->scratch_e.reserve(n * 12));
    HIP_TRY(hipMemcpyAsync(c->scratch_a.p, u4, n * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(sample_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, c->stream, kind, static_cast<const float*>(c->scratch_a.p), n,
                       static_cast<float*>(c->scratch_e.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out3, c->scratch_e.p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTK_OK;
}

int rtk_philox_batch(rtk_context* c, uint64_t seed, const uint32_t* ctr4, size_t n, uint32_t* out4) {
    if (!c || !out4 || (!ctr4 && n)) return fail(RTK_ERR_BAD_ARG, "rtk_philox_batch: null argument");
    if (n == 0) return RTK_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->scratch_a.reserve(n * 16));
    HIP_TRY(c->scratch_e.reserve(n * 16));
    HIP_TRY(hipMemcpyAsync(c->scratch_a.p, ctr4, n * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(philox_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, c->stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32),
                       static_cast<const uint32_t*>(c->scratch_a.p), n, static_cast<uint32_t*>(c->scratch_e.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out4, c->scratch_e.p, n * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTK_OK;
}

} // extern "C"

// =================================================================================================
// One process, several GPUs (rtk_multi_*): the C/C++ drop-in's way to use a whole node.
// Rank r = devices[r] renders the tiles i % n == r with its own context and stream; the compact tile buffers are
// gathered on rank 0 — RCCL send/recv inside one group when the devices are distinct (ncclCommInitAll, one
// communicator per device, xGMI underneath), plain device copies ordered by events when a device is listed more than
// once (a test arrangement: several ranks on one GPU) — and de-interleaved there.  No collective during rendering.
// =================================================================================================
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;

static int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return RTK_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(RTK_ERR_HIP, "cannot open librccl.so.1 (needed to gather tiles from several GPUs): %s", dlerror());
    RcclApi a;
    a.handle = h;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.CommInitAll || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv || !a.GetErrorString)
        return fail(RTK_ERR_HIP, "librccl.so.1 lacks a symbol librtk needs");
    g_rccl = a;
    return RTK_OK;
}

#define RCCL_TRY(expr)                                                                                                   \
    do {                                                                                                                 \
        const ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(RTK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct rtk_multi {
    std::vector<int> devices;
    std::vector<rtk_context*> ctx;
    std::vector<ncclComm_t> comms;          // one per rank when the devices are distinct (RCCL gather), else empty
    std::vector<hipEvent_t> done;           // per rank: its tiles are rendered (the copy gather waits on these)
    std::vector<DevBuf> tiles;              // per rank >= 1: its compact tile sums, on its device
    DevBuf gathered, image, image_u8;       // on devices[0]
    bool use_rccl = false;
};

extern "C" {

void rtk_multi_destroy(rtk_multi* m) {
    if (!m) return;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        if (!m->ctx[r]) continue;
        (void)hipSetDevice(m->devices[r]);
        (void)hipStreamSynchronize(m->ctx[r]->stream);
        if (r < m->comms.size() && m->comms[r]) (void)g_rccl.CommDestroy(m->comms[r]);
        if (r < m->done.size() && m->done[r]) (void)hipEventDestroy(m->done[r]);
        if (r < m->tiles.size()) m->tiles[r].release();
    }
    if (!m->devices.empty()) {
        (void)hipSetDevice(m->devices[0]);
        m->gathered.release();
        m->image.release();
        m->image_u8.release();
    }
    for (rtk_context* c : m->ctx) rtk_context_destroy(c);
    delete m;
}

int rtk_multi_create(const int32_t* devices, int32_t n, rtk_multi** out) {
    if (!out) return fail(RTK_ERR_BAD_ARG, "rtk_multi_create: out is null");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(RTK_ERR_BAD_ARG, "rtk_multi_create: %d devices", n);
    rtk_multi* m = new (std::nothrow) rtk_multi();
    if (!m) return fail(RTK_ERR_HIP, "out of host memory");
    const auto init = [m, devices, n]() -> int {
        bool distinct = true;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
        m->devices.assign(devices, devices + n);
        m->ctx.assign(n, nullptr);
        m->done.assign(n, nullptr);
        m->tiles.resize(n);
        for (int r = 0; r < n; ++r) {
            if (int e = rtk_context_create(devices[r], &m->ctx[r])) return e;
            HIP_TRY(hipEventCreateWithFlags(&m->done[r], hipEventDisableTiming));
        }
        m->use_rccl = n > 1 && distinct;
        if (m->use_rccl) {
            if (int e = load_rccl()) return e;
            m->comms.assign(n, nullptr);
            RCCL_TRY(g_rccl.CommInitAll(m->comms.data(), n, m->devices.data()));
        }
        return RTK_OK;
    };
    if (const int e = init()) {
        rtk_multi_destroy(m);
        return e;
    }
    *out = m;
    return RTK_OK;
}

int rtk_multi_device_count(const rtk_multi* m) { return m ? static_cast<int>(m->devices.size()) : 0; }
int rtk_multi_uses_rccl(const rtk_multi* m) { return (m && m->use_rccl) ? 1 : 0; }

int rtk_multi_set_accel(rtk_multi* m, int32_t mode) {
    if (!m) return fail(RTK_ERR_BAD_ARG, "null rtk_multi");
    for (rtk_context* c : m->ctx)
        if (int e = rtk_context_set_accel(c, mode)) return e;
    return RTK_OK;
}

int rtk_multi_set_scene(rtk_multi* m, const rtk_scene* scene) {
    if (!m || !scene) return fail(RTK_ERR_BAD_ARG, "rtk_multi_set_scene: null argument");
    for (rtk_context* c : m->ctx)            // the scene is tiny and read-only: replicated
        if (int e = rtk_set_scene(c, scene)) return e;
    return RTK_OK;
}

int rtk_multi_render(rtk_multi* m, const rtk_camera* cam, const rtk_params* p, float* out_rgb_sum, uint8_t* out_u8, rtk_stats* per_device) {
    if (!m || !cam || !p || (!out_rgb_sum && !out_u8)) return fail(RTK_ERR_BAD_ARG, "rtk_multi_render: null argument");
    if (out_u8 && p->sample_offset != 0) return fail(RTK_ERR_BAD_ARG, "rtk_multi_render quantises a complete frame: sample_offset must be 0 for the 8-bit output");
    int tw, th;
    if (int e = norm_tiles(p, tw, th)) return e;
    if (p->width < 2 || p->height < 2) return fail(RTK_ERR_BAD_ARG, "image %d x %d: the reference divides by (W-1) and (H-1), main.cpp:69-70", p->width, p->height);
    const int n = static_cast<int>(m->devices.size());
    const int64_t tprp = rtk_local_tile_count(p->width, p->height, tw, th, 0, n);        // rank 0 owns the most tiles
    const size_t slice = static_cast<size_t>(tprp) * kTilePixels * 3;                     // floats per rank in the gather buffer
    const size_t np = static_cast<size_t>(p->width) * p->height;
    rtk_context* root = m->ctx[0];
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (slice * n * sizeof(float) > m->gathered.cap) {
        HIP_TRY(hipStreamSynchronize(root->stream));
        HIP_TRY(m->gathered.reserve(slice * n * sizeof(float)));
        HIP_TRY(hipMemset(m->gathered.p, 0, m->gathered.cap));      // tiles a rank does not own (padding of the last group) stay 0
    }
    HIP_TRY(m->image.reserve(np * 3 * sizeof(float)));
    if (out_u8) HIP_TRY(m->image_u8.reserve(np * 3));
    float* gathered = static_cast<float*>(m->gathered.p);
    // 1. every rank renders its tiles on its own device and stream (asynchronous: the kernels of all devices overlap)
    for (int r = 0; r < n; ++r) {
        HIP_TRY(hipSetDevice(m->devices[r]));
        float* dst = gathered;
        if (r > 0) {
            if (slice * sizeof(float) > m->tiles[r].cap) {
                HIP_TRY(hipStreamSynchronize(m->ctx[r]->stream));
                HIP_TRY(m->tiles[r].reserve(slice * sizeof(float)));
                HIP_TRY(hipMemset(m->tiles[r].p, 0, m->tiles[r].cap));
            }
            dst = static_cast<float*>(m->tiles[r].p);
        }
        rtk_params q = *p;
        q.rank = r;
        q.world_size = n;
        if (per_device)
            if (int e = rtk_reset_stats(m->ctx[r])) return e;
        if (int e = rtk_render_tiles_device(m->ctx[r], cam, &q, dst, nullptr)) return e;
        HIP_TRY(hipEventRecord(m->done[r], m->ctx[r]->stream));
    }
    // 2. gather on rank 0: one RCCL group of sends and receives, or device copies when ranks share a device
    if (n > 1) {
        if (m->use_rccl) {
            RCCL_TRY(g_rccl.GroupStart());
            for (int r = 1; r < n; ++r) {
                HIP_TRY(hipSetDevice(m->devices[r]));
                RCCL_TRY(g_rccl.Send(m->tiles[r].p, slice, ncclFloat, 0, m->comms[r], m->ctx[r]->stream));
                HIP_TRY(hipSetDevice(m->devices[0]));
                RCCL_TRY(g_rccl.Recv(gathered + static_cast<size_t>(r) * slice, slice, ncclFloat, r, m->comms[0], root->stream));
            }
            RCCL_TRY(g_rccl.GroupEnd());
        } else {
            HIP_TRY(hipSetDevice(m->devices[0]));
            for (int r = 1; r < n; ++r) {
                HIP_TRY(hipStreamWaitEvent(root->stream, m->done[r], 0));
                HIP_TRY(hipMemcpyAsync(gathered + static_cast<size_t>(r) * slice, m->tiles[r].p, slice * sizeof(float), hipMemcpyDeviceToDevice, root->stream));
            }
        }
    }
    // 3. scanline order (and the reference's bytes) on rank 0, then to the host
    HIP_TRY(hipSetDevice(m->devices[0]));
    rtk_params q = *p;
    q.rank = 0;
    q.world_size = n;
    if (int e = rtk_deinterleave_device(root, gathered, tprp, &q, static_cast<float*>(m->image.p))) return e;
    if (out_u8) {
        if (int e = rtk_quantise_device(root, static_cast<const float*>(m->image.p), static_cast<int64_t>(np), p->spp, static_cast<uint8_t*>(m->image_u8.p))) return e;
        HIP_TRY(hipMemcpyAsync(out_u8, m->image_u8.p, np * 3, hipMemcpyDeviceToHost, root->stream));
    }
    if (out_rgb_sum) HIP_TRY(hipMemcpyAsync(out_rgb_sum, m->image.p, np * 3 * sizeof(float), hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipStreamSynchronize(root->stream));
    for (int r = 0; r < n; ++r) {
        HIP_TRY(hipSetDevice(m->devices[r]));
        if (per_device) {
            if (int e = rtk_get_stats(m->ctx[r], &per_device[r])) return e;      // synchronises rank r's stream
        } else {
            HIP_TRY(hipStreamSynchronize(m->ctx[r]->stream));
        }
    }
    return RTK_OK;
}

} // extern "C"
