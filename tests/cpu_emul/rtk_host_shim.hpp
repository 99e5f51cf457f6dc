// rtk_host_shim.hpp — lets raytracinginoneweekend_amd/csrc/rtk_device.hpp and rtk_render.hpp compile with g++
// as a ONE-LANE-PER-WAVE CPU program (RTK_HOST_EMULATION, RTK_LANES=1).  TEST INFRASTRUCTURE ONLY: it exists so
// the device logic (filter conservativeness, compaction, exact stage, shading, work distribution) can be checked
// against the oracle in the CPU test-suite; the shipped library never includes it and has no CPU path.
//
// Lanes of a wavefront interact only through votes (ballot / any) and lane-private list slots, so with one lane
// per wave every vote degenerates to the lane's own predicate and the per-lane results are exactly those of the
// 64-lane kernel.  Hardware approximations (v_rsq_f32, v_rcp_f32, v_sqrt_f32, v_sin/cos, v_exp/log) are
// modelled by libm plus an optional adversarial +-ulp perturbation (emul_set_ulp_noise).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline

struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }

inline uint32_t __umulhi(uint32_t a, uint32_t b) { return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz(static_cast<unsigned>(x)); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline long long __double2ll_rn(double d) { return std::llrint(d); }
#if defined(RTK_LANES) && RTK_LANES == 64
#include "rtk_simt.hpp"
inline unsigned long long __ballot(bool p) { return rtk_simt::ballot(p); }
#else
inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
#endif
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long rtk_wall_clock() { return 0ull; }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

namespace rtk_shim {
extern thread_local int ulp_noise;      // 0: libm values; k > 0: pseudo-random perturbation within +-k ulp
extern thread_local uint32_t noise_state;
inline float nudge(float x) {
    if (ulp_noise == 0 || !(x == x) || std::isinf(x) || x == 0.0f) return x;
    noise_state = noise_state * 1664525u + 1013904223u;
    const int k = static_cast<int>((noise_state >> 16) % static_cast<uint32_t>(2 * ulp_noise + 1)) - ulp_noise;
    uint32_t u; std::memcpy(&u, &x, 4);
    u = static_cast<uint32_t>(static_cast<int32_t>(u) + k);
    float y; std::memcpy(&y, &u, 4);
    return y;
}
// v_frexp_exp_i32_f32: 0 for zero, infinities and NaN
inline int frexp_exp(float x) { int e = 0; if (x == 0.0f || !std::isfinite(x)) return 0; std::frexp(x, &e); return e; }
inline float rsqf(float x) { return nudge(static_cast<float>(1.0 / std::sqrt(static_cast<double>(x)))); }
inline float rcpf(float x) { return nudge(static_cast<float>(1.0 / static_cast<double>(x))); }
inline float sqrtf_(float x) { return nudge(std::sqrt(x)); }
inline float sin_rev(float x) { return nudge(static_cast<float>(std::sin(6.283185307179586 * static_cast<double>(x)))); }
inline float cos_rev(float x) { return nudge(static_cast<float>(std::cos(6.283185307179586 * static_cast<double>(x)))); }
inline float exp2f_(float x) { return nudge(std::exp2(x)); }
inline float log2f_(float x) { return nudge(std::log2(x)); }
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (sh & 31)); }
} // namespace rtk_shim

#define __builtin_amdgcn_rsqf(x) rtk_shim::rsqf(x)
#define __builtin_amdgcn_frexp_expf(x) rtk_shim::frexp_exp(x)
#define __builtin_amdgcn_rcpf(x) rtk_shim::rcpf(x)
#define __builtin_amdgcn_sqrtf(x) rtk_shim::sqrtf_(x)
#define __builtin_amdgcn_sinf(x) rtk_shim::sin_rev(x)
#define __builtin_amdgcn_cosf(x) rtk_shim::cos_rev(x)
#define __builtin_amdgcn_exp2f(x) rtk_shim::exp2f_(x)
#define __builtin_amdgcn_logf(x) rtk_shim::log2f_(x)
#define __builtin_amdgcn_alignbit(a, b, c) rtk_shim::alignbit(a, b, c)
#if defined(RTK_LANES) && RTK_LANES == 64
#define __builtin_amdgcn_mbcnt_lo(m, v) rtk_simt::mbcnt_lo(m, v)
#define __builtin_amdgcn_mbcnt_hi(m, v) rtk_simt::mbcnt_hi(m, v)
#define __builtin_amdgcn_readfirstlane(x) rtk_simt::readfirstlane(x)
#else
#define __builtin_amdgcn_mbcnt_lo(m, v) (v)
#define __builtin_amdgcn_mbcnt_hi(m, v) (v)
#define __builtin_amdgcn_readfirstlane(x) (x)
#endif

// Dynamic wave-level statistics of the emulated kernels (how often each block of the pass executes for a wave).
namespace rtk_shim {
struct WaveStats {
    unsigned long long closest_hit_calls, filter_words, compact_iters, early_flushes, flush_iters;
    unsigned long long pass_refill_block, pass_assign_iters, pass_start_block, pass_sky_block, pass_hit_block;
    unsigned long long pass_lambert, pass_metal, pass_glass, pass_unit_flush_block, packet_draws, class_words;
    unsigned long long tail_tests;           // passes that tested a short last word on the vector ALUs
    unsigned long long clamp_fractions;      // filter tests (every lane) whose clamped value fell strictly inside (0, 1)
};
extern thread_local WaveStats wave_stats;
}
// counted once per wave (by lane 0 of it)
#define RTK_STAT(lane, name) do { if ((lane) == 0) ++rtk_shim::wave_stats.name; } while (0)
// counted for every lane
#define RTK_STAT_EACH(name) do { ++rtk_shim::wave_stats.name; } while (0)
// counted once per wave when ANY lane satisfies `pred` (an extra vote that exists only in the emulation build)
#define RTK_STAT_ANY(lane, name, pred) do { const bool any_ = __ballot(pred) != 0ull; if (any_ && (lane) == 0) ++rtk_shim::wave_stats.name; } while (0)
