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
    unsigned long long* stats;   // [0] rays [1] samples [2] exact tests [3] wave passes through the intersector [4] most passes of one wave [5] ~earliest start [6] ~earliest end [7] latest end (100 MHz ticks)
    GridDev grid;                // only read by the GRID instantiation (rtk_context_set_accel)
#ifdef RTK_HOST_EMULATION
    uint8_t* dbg_seg = nullptr;  // emulation only: [pixel][spp] segments traced per sample (work-distribution studies)
#endif
};

// Persistent wavefronts.  A packet = one 64-pixel tile x one run of consecutive samples; a wavefront
// draws packets from a global counter and hands the packet's 64 (pixel, run) units to whichever of
// its lanes is free, so lanes whose path ended (sky, absorbed, depth exhausted) start the next sample
// at once instead of idling behind the long-tail paths (SURVEY.md §7 hard part 3).  Every sample's
// colour becomes 32.32 fixed point the moment its path ends; a lane sums its run in integers and adds
// the run total to the pixel's accumulator: integer adds commute and associate, so the image is
// bit-identical for any scheduling, run partition, tile size and GPU count.
constexpr int kDequeue = 2;      // packets a wavefront draws from the queue per atomic

// fp32 sample colour -> signed 32.32 fixed point (two's complement in a u64), entirely in fp32/integer instructions (f64
// conversions run at a fraction of the fp32 rate and this executes in every pass).  Integer part = floor, fraction
// truncated to 2^-32 (exact whenever the fraction's ulp is at least 2^-32); NaN contributes 0, magnitudes are clamped
// to +-2^30.  Any deterministic map would do — integer accumulation is what makes the image independent of scheduling.
__device__ __forceinline__ unsigned long long to_fixed_32_32(float f) {
    const float lim = 0x1p30f;
    const float c = (f == f) ? fminf(fmaxf(f, -lim), lim) : 0.0f;
    const float hi = floorf(c);
    const float frac = fminf((c - hi) * 4294967296.0f, 4294967040.0f);      // [0, 2^32): c - hi can round up to 1 only for tiny negative c
    const uint32_t lo = static_cast<uint32_t>(frac);
    const uint32_t top = static_cast<uint32_t>(static_cast<int32_t>(hi));
    return (static_cast<unsigned long long>(top) << 32) | lo;
}

// `cam`: where the camera is read from in the primary-ray block.  The kernel hands in an LDS copy: the 22 floats are
// needed once per pass, and as kernel arguments they sit in SGPRs that the register allocator spills to VGPR lanes
// and reads back with v_readlane (VALU slots) in every pass; from LDS they cost six reads on the LDS pipe instead.
template <bool LDS, bool GRID = false>
__device__ __forceinline__ void render_wave(const RenderArgs& A, const Intersector<LDS>& isect, const CameraDev* cam, const GridDev* grid = nullptr) {
    const int lane = isect.lane;
    // The Philox key as the generator sees it: taken through an empty statement at every call, so that the ten round keys
    // are formed by scalar adds where they are used.  As loop invariants they were twenty scalar registers for the life
    // of the kernel, spilled to lanes of a vector register and read back with v_readlane in every pass.
    auto philox_key = [&]() {
        uint32_t k0 = A.seed_lo, k1 = A.seed_hi;
#ifndef RTK_HOST_EMULATION
        asm volatile("" : "+s"(k0), "+s"(k1));
#endif
        return make_uint2(k0, k1);
    };
    const unsigned long long t_begin = rtk_wall_clock();

    // lane state
    Ray ray = {};
    V3 thr = {0, 0, 0};
    unsigned long long acc_r = 0, acc_g = 0, acc_b = 0;      // this unit's colour sums, 32.32 fixed point
    bool alive = false, has_unit = false;
    int depth_left = 0, unit_left = 0;
    uint32_t slot = 0, sample = 0;
    uint32_t pxy = 0;            // pixel column | row << 16 (rtk_render_tiles checks W, H <= 65535); the pixel index and the bounce number
                                 // are re-derived where they are needed: every register that lives through the loop counts (128 per lane)
    // wave state (uniform)
    uint32_t n_seg = 0, n_samp = 0, n_exact = 0;      // totals of the wave (votes), not per lane
    int pk_pos = kTilePixels, pk_lt = 0, pk_x0 = 0, pk_y0 = 0;
    uint32_t pk_first = 0, pk_len = 1, pk_next = 0, pk_end = 0;
    bool queue_empty = false;
    uint32_t wave_passes = 0;

#ifdef RTK_PROFILE_SECTIONS
    const unsigned long long prof_begin = __builtin_readcyclecounter();
#endif
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
                    unsigned long long* dst = A.accum + static_cast<size_t>(slot) * 3;
                    atomicAdd(dst + 0, acc_r);
                    atomicAdd(dst + 1, acc_g);
                    atomicAdd(dst + 2, acc_b);
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
                const int px = pk_x0 + (k & (A.tile_w - 1));
                const int prow = pk_y0 + (k >> A.tile_w_log2);
                if (px < A.W && prow < A.H) {
                    pxy = static_cast<uint32_t>(px) | (static_cast<uint32_t>(prow) << 16);
                    slot = static_cast<uint32_t>(pk_lt) * kTilePixels + static_cast<uint32_t>(k);
                    sample = pk_first + A.sample_offset;
                    unit_left = static_cast<int>(pk_len) - 1;
                    acc_r = acc_g = acc_b = 0;
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
        n_samp += static_cast<uint32_t>(__popcll(ballot(start)));
        // the global pixel index, re-derived at each use (the empty statement keeps the compiler from carrying the first
        // result through the intersection in a register)
        auto pixel_index = [&]() {
            uint32_t q = pxy;
#ifndef RTK_HOST_EMULATION
            asm volatile("" : "+v"(q));
#endif
            return (q >> 16) * static_cast<uint32_t>(A.W) + (q & 0xFFFFu);
        };
        if (start) {
            const int px = static_cast<int>(pxy & 0xFFFFu), prow = static_cast<int>(pxy >> 16);
            const uint32_t pixel = pixel_index();
            const uint4 rnd = philox4x32_10(make_uint4(pixel, sample, 0u, 0u), philox_key());
            const float u[4] = {u01(rnd.x), u01(rnd.y), u01(rnd.z), u01(rnd.w)};
            asm volatile("" ::: "memory");      // keep the camera reads in this block (hoisted, they would occupy 22 VGPRs for the whole loop)
            ray = primary_ray(*cam, px, prow, A.W, A.H, u);
            thr = {1.0f, 1.0f, 1.0f};
            depth_left = A.max_depth;
            alive = depth_left > 0;   // main.cpp:87-89: depth <= 0 gathers no light and traces nothing
        }
        if (!wave_any(alive)) {
            if (!wave_any(has_unit)) break;
            continue;
        }
        // ---- one segment: HittableList::hit (main.cpp:90) ----
        float t;
        int idx;
        if (GRID) isect.closest_hit_grid(*grid, ray, alive, A.t_min, __builtin_inff(), t, idx, n_exact);
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
        n_seg += static_cast<uint32_t>(__popcll(ballot(alive)));
        if (alive) {
            // unit_vector(r.direction()) of main.cpp:98 / Material.cpp:22,36 — one evaluation serves sky, metal and glass
            const V3 ud = unit_vector(ray.d);
#ifdef RTK_HOST_EMULATION
            if (A.dbg_seg) ++A.dbg_seg[static_cast<size_t>(pixel_index()) * A.spp + (sample - A.sample_offset)];
#endif
            if (idx < 0) {
                const V3 c = thr * sky_unit(ud);           // main.cpp:98-100: the only non-zero outcome of a path
                acc_r += to_fixed_32_32(c.x);
                acc_g += to_fixed_32_32(c.y);
                acc_b += to_fixed_32_32(c.z);
                alive = false;
            } else {
                const uint32_t bounce = static_cast<uint32_t>(A.max_depth - depth_left) + 1u;      // 1 for the first hit of a path
                const uint4 rnd = philox4x32_10(make_uint4(pixel_index(), sample, bounce, 0u), philox_key());
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
#ifdef RTK_PROFILE_WORD
    if (lane == 0 && blockIdx.x < 2 && isect.wp[5])
        printf("wave %d.%d: class words %llu  ticks per word: head (descriptor wait, prefetches) %.0f  rebuild %.0f  mfma statement %.0f  64 FMAs + join %.0f  push/check %.0f\n", static_cast<int>(blockIdx.x),
               static_cast<int>(threadIdx.x / kWave), isect.wp[5], static_cast<double>(isect.wp[0]) / isect.wp[5], static_cast<double>(isect.wp[1]) / isect.wp[5], static_cast<double>(isect.wp[2]) / isect.wp[5],
               static_cast<double>(isect.wp[3]) / isect.wp[5], static_cast<double>(isect.wp[4]) / isect.wp[5]);
#endif
#ifdef RTK_PROFILE_SECTIONS
    // profile builds only: where this wave's shader cycles went (wall cycles of the wave: its SIMD interleaves it with the others)
    if (lane == 0 && blockIdx.x < 3) {
        const unsigned long long total = __builtin_readcyclecounter() - prof_begin;
        printf("wave %d.%d: passes %u  cycles/pass: total %.0f  filter %.0f  exact stage %.0f  (streamed local walk: compaction %.0f  final exact pass %.0f)  rest %.0f\n", static_cast<int>(blockIdx.x), static_cast<int>(threadIdx.x / kWave),
               wave_passes, static_cast<double>(total) / wave_passes, static_cast<double>(isect.prof[0]) / wave_passes, static_cast<double>(isect.prof[1]) / wave_passes,
               static_cast<double>(isect.prof[2]) / wave_passes, static_cast<double>(isect.prof[3]) / wave_passes,
               static_cast<double>(total - isect.prof[0] - isect.prof[1] - isect.prof[2] - isect.prof[3]) / wave_passes);
    }
#endif
    if (lane == 0) {
        atomicAdd(A.stats + 0, static_cast<unsigned long long>(n_seg));
        atomicAdd(A.stats + 1, static_cast<unsigned long long>(n_samp));
        atomicAdd(A.stats + 2, static_cast<unsigned long long>(n_exact));
        atomicAdd(A.stats + 3, static_cast<unsigned long long>(wave_passes));
        atomicMax(A.stats + 4, static_cast<unsigned long long>(wave_passes));
        // kernel start (earliest), first and last wave to run out of work: ~t so that the zeroed counter is the identity of max
        atomicMax(A.stats + 5, ~t_begin);
        const unsigned long long t_end = rtk_wall_clock();
        atomicMax(A.stats + 6, ~t_end);
        atomicMax(A.stats + 7, t_end);
    }
}


// accumulator (signed 32.32 fixed point) -> fp32 colour sum
__device__ __forceinline__ float accum_to_float(unsigned long long a) {
    return static_cast<float>(static_cast<double>(static_cast<long long>(a)) * (1.0 / 4294967296.0));
}

} // namespace rtk
