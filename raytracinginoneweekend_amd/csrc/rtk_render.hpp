// rtk_render.hpp — the render loop of one wavefront (device code; also built for the CPU by the
// emulation harness under tests/cpu_emul with one lane per wave).
#pragma once

#include "rtk_device.hpp"
#include "rtk_intersect.hpp"

namespace rtk {

struct RenderArgs {
    SceneDev scene;
    CameraDev cam;
    int32_t W, H, spp, max_depth;
    uint32_t sample_offset;      // first sample index of this batch (RNG counter), rtk_params::sample_offset
    uint32_t seed_lo, seed_hi;
    float t_min;
    int32_t tile_w, tile_h, tiles_x;
    int32_t tile_w_log2;         // tile_w * tile_h == 64, so tile_w is a power of two
    int32_t rank, world;
    int32_t n_local_tiles;
    const uint2* runs;           // [n_runs] (first sample, length) of every run, rtk_scene_prep.hpp run_schedule()
    int32_t n_runs;
    uint32_t total_packets;
    unsigned long long* accum;   // [n_local_tiles][64][3] signed 32.32 fixed point
    uint32_t* queue;             // packet counter
    unsigned long long* stats;   // [0] rays [1] samples [2] exact tests [3] wave passes through the intersector
    GridDev grid;                // only read by the GRID instantiation (rtk_context_set_accel)
#ifdef RTK_HOST_EMULATION
    uint8_t* dbg_seg = nullptr;  // emulation only: [pixel][spp] segments traced per sample (work-distribution studies)
#endif
};

// Persistent wavefronts.  A packet = one 64-pixel tile x one run of consecutive samples; a wavefront
// draws packets from a global counter and hands the packet's 64 (pixel, run) units to whichever of
// its lanes is free, so lanes whose path ended (sky, absorbed, depth exhausted) start the next sample
// at once instead of idling behind the long-tail paths (SURVEY.md §7 hard part 3).  A lane sums its
// run in fp32 and adds the run total once, as 32.32 fixed point, to the pixel's accumulator: integer
// adds commute, so the image is bit-identical for any scheduling, tile size and GPU count.
constexpr int kDequeue = 2;      // packets a wavefront draws from the queue per atomic

// fp32 run sum -> signed 32.32 fixed point (two's complement in a u64), entirely in fp32/integer instructions (f64
// conversions run at a fraction of the fp32 rate and this executes in nearly every pass).  Integer part = floor,
// fraction rounded to nearest-even 2^-32; NaN contributes 0, magnitudes are clamped to +-2^30.  Any deterministic map
// would do — integer accumulation is what makes the image independent of scheduling — this one is exact for every
// non-negative sum (f - floor(f) and its scaling by 2^32 are exact in fp32).
__device__ __forceinline__ unsigned long long to_fixed_32_32(float f) {
    const float lim = 0x1p30f;
    const float c = (f == f) ? fminf(fmaxf(f, -lim), lim) : 0.0f;
    float hi = floorf(c);
    float frac = c - hi;                       // [0, 1]; can round up to 1 only for tiny negative c
    if (frac >= 1.0f) { hi += 1.0f; frac = 0.0f; }
    const uint32_t lo = static_cast<uint32_t>(rintf(frac * 4294967296.0f));      // < 2^32
    const uint32_t top = static_cast<uint32_t>(static_cast<int32_t>(hi));
    return (static_cast<unsigned long long>(top) << 32) | lo;
}

// `cam`: where the camera is read from in the primary-ray block.  The kernel hands in an LDS copy: the 22 floats are
// needed once per pass, and as kernel arguments they sit in SGPRs that the register allocator spills to VGPR lanes
// and reads back with v_readlane (VALU slots) in every pass; from LDS they cost six reads on the LDS pipe instead.
template <bool LDS, bool GRID = false>
__device__ __forceinline__ void render_wave(const RenderArgs& A, const Intersector<LDS>& isect, const CameraDev* cam) {
    const int lane = isect.lane;
    const uint2 key = make_uint2(A.seed_lo, A.seed_hi);

    // lane state
    Ray ray = {};
    V3 thr = {0, 0, 0}, acc = {0, 0, 0};
    bool alive = false, has_unit = false;
    int depth_left = 0, unit_left = 0;
    uint32_t pixel = 0, slot = 0, sample = 0, bounce = 0;
    int px = 0, prow = 0;
    uint32_t n_seg = 0, n_samp = 0, n_exact = 0;
    // wave state (uniform)
    int pk_pos = kTilePixels, pk_lt = 0, pk_x0 = 0, pk_y0 = 0;
    uint32_t pk_first = 0, pk_len = 1, pk_next = 0, pk_end = 0;
    bool queue_empty = false;
    uint32_t wave_passes = 0;

    for (;;) {
        // ---- lanes without a live path: next sample of the unit, or retire the unit ----
        bool start = false, need = false;
        RTK_STAT_ANY(lane, pass_refill_block, !alive);
        RTK_STAT_ANY(lane, pass_unit_flush_block, !alive && has_unit && unit_left <= 0);
        if (!alive) {
            if (has_unit && unit_left > 0) {
                --unit_left;
                ++sample;
                start = true;
            } else {
                if (has_unit) {
                    const unsigned long long qr = to_fixed_32_32(acc.x), qg = to_fixed_32_32(acc.y), qb = to_fixed_32_32(acc.z);
                    unsigned long long* dst = A.accum + static_cast<size_t>(slot) * 3;
                    atomicAdd(dst + 0, qr);
                    atomicAdd(dst + 1, qg);
                    atomicAdd(dst + 2, qb);
                    has_unit = false;
                }
                need = true;
            }
        }
        // ---- hand out units of the wave's packet; draw a new packet when it runs dry ----
        for (;;) {
            const uint64_t nb = ballot(need);
            if (nb == 0ull) break;
            RTK_STAT(lane, pass_assign_iters);
            if (pk_pos >= kTilePixels) {
                if (pk_next >= pk_end) {
                    // draw kDequeue consecutive packets at once: towards the end of a frame the runs are single
                    // samples, and one packet per draw would push the shared counter towards its saturation rate
                    if (queue_empty) break;
                    uint32_t q0 = 0;
                    if (lane == 0) q0 = atomicAdd(A.queue, static_cast<uint32_t>(kDequeue));
                    q0 = __builtin_amdgcn_readfirstlane(q0);
                    if (q0 >= A.total_packets) {
                        queue_empty = true;
                        break;
                    }
                    pk_next = q0;
                    pk_end = (A.total_packets - q0 < static_cast<uint32_t>(kDequeue)) ? A.total_packets : q0 + kDequeue;
                }
                const uint32_t q = pk_next++;
                RTK_STAT(lane, packet_draws);
                // per-packet (wave-uniform) quantities: run of samples, tile origin
                const uint2 run = A.runs[q / static_cast<uint32_t>(A.n_local_tiles)];
                pk_first = run.x;
                pk_len = run.y;
                pk_lt = static_cast<int>(q % static_cast<uint32_t>(A.n_local_tiles));
                const int gt = pk_lt * A.world + A.rank;
                pk_x0 = (gt % A.tiles_x) * A.tile_w;
                pk_y0 = (gt / A.tiles_x) * A.tile_h;
                pk_pos = 0;
            }
            const int avail = kTilePixels - pk_pos;
            const int r = lane_prefix(nb);
            if (need && r < avail) {
                const int k = pk_pos + r;
                px = pk_x0 + (k & (A.tile_w - 1));
                prow = pk_y0 + (k >> A.tile_w_log2);
                if (px < A.W && prow < A.H) {
                    pixel = static_cast<uint32_t>(prow) * static_cast<uint32_t>(A.W) + static_cast<uint32_t>(px);
                    slot = static_cast<uint32_t>(pk_lt) * kTilePixels + static_cast<uint32_t>(k);
                    sample = pk_first + A.sample_offset;
                    unit_left = static_cast<int>(pk_len) - 1;
                    acc = {0, 0, 0};
                    has_unit = true;
                    start = true;
                    need = false;
                }
                // a unit that falls outside the image is dropped; the lane asks again
            }
            const int took = __popcll(nb);
            pk_pos += took < avail ? took : avail;
        }
        // ---- primary rays (main.cpp:69-71) ----
        RTK_STAT_ANY(lane, pass_start_block, start);
        if (start) {
            const uint4 rnd = philox4x32_10(make_uint4(pixel, sample, 0u, 0u), key);
            const float u[4] = {u01(rnd.x), u01(rnd.y), u01(rnd.z), u01(rnd.w)};
            asm volatile("" ::: "memory");      // keep the camera reads in this block (hoisted, they would occupy 22 VGPRs for the whole loop)
            ray = primary_ray(*cam, px, prow, A.W, A.H, u);
            thr = {1.0f, 1.0f, 1.0f};
            depth_left = A.max_depth;
            bounce = 0;
            alive = depth_left > 0;   // main.cpp:87-89: depth <= 0 gathers no light and traces nothing
            ++n_samp;
        }
        if (!wave_any(alive)) {
            if (!wave_any(has_unit)) break;
            continue;
        }
        // ---- one segment: HittableList::hit (main.cpp:90) ----
        float t;
        int idx;
        if (GRID) isect.closest_hit_grid(A.grid, ray, alive, A.t_min, __builtin_inff(), t, idx, n_exact);
        else isect.closest_hit(ray, alive, A.t_min, __builtin_inff(), t, idx, n_exact);
        ++wave_passes;
        RTK_STAT_ANY(lane, pass_sky_block, alive && idx < 0);
        RTK_STAT_ANY(lane, pass_hit_block, alive && idx >= 0);
#ifdef RTK_HOST_EMULATION
        {
            const int ty_ = (alive && idx >= 0) ? __float_as_int(A.scene.mat1[idx].z) : 0;
            RTK_STAT_ANY(lane, pass_lambert, ty_ == 1);
            RTK_STAT_ANY(lane, pass_metal, ty_ == 2);
            RTK_STAT_ANY(lane, pass_glass, ty_ == 3);
        }
#endif
        if (alive) {
            // unit_vector(r.direction()) of main.cpp:98 / Material.cpp:22,36 — one evaluation serves sky, metal and glass
            const V3 ud = unit_vector(ray.d);
            ++n_seg;
#ifdef RTK_HOST_EMULATION
            if (A.dbg_seg) ++A.dbg_seg[static_cast<size_t>(pixel) * A.spp + (sample - A.sample_offset)];
#endif
            if (idx < 0) {
                acc = acc + thr * sky_unit(ud);            // main.cpp:98-100
                alive = false;
            } else {
                ++bounce;
                const uint4 rnd = philox4x32_10(make_uint4(pixel, sample, bounce, 0u), key);
                const float u[4] = {u01(rnd.x), u01(rnd.y), u01(rnd.z), u01(rnd.w)};
                const float4 ex = A.scene.exact[idx], m0 = A.scene.mat0[idx], m1 = A.scene.mat1[idx];
                Ray out;
                const bool ok = hit_and_scatter(ray, ud, t, ex, m0, m1, u, out);
                thr = thr * V3{m0.x, m0.y, m0.z};        // main.cpp:93 (running product)
                ray = out;
                --depth_left;
                alive = ok && depth_left > 0;            // main.cpp:95 / 87-89: absorbed or out of depth -> black
            }
        }
    }
    atomicAdd(A.stats + 0, static_cast<unsigned long long>(n_seg));
    atomicAdd(A.stats + 1, static_cast<unsigned long long>(n_samp));
    atomicAdd(A.stats + 2, static_cast<unsigned long long>(n_exact));
    if (lane == 0) atomicAdd(A.stats + 3, static_cast<unsigned long long>(wave_passes));
}


// accumulator (signed 32.32 fixed point) -> fp32 colour sum
__device__ __forceinline__ float accum_to_float(unsigned long long a) {
    return static_cast<float>(static_cast<double>(static_cast<long long>(a)) * (1.0 / 4294967296.0));
}

} // namespace rtk
