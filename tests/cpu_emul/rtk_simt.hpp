// rtk_simt.hpp — a 64-lane wavefront on the CPU for the emulation build (RTK_LANES=64): every lane is a user-space
// fiber (ucontext), the lanes of one wave run on one OS thread and meet at the wave-level primitives
// (__ballot, readfirstlane): a primitive parks the calling lane until every live lane of the wave has reached a
// primitive too, then all resume with the combined result.  This is exact for code whose votes sit in wave-uniform
// control flow, which is how the kernels are written; a lane that reaches a vote while another lane has finished
// the kernel, or a round in which lanes wait at DIFFERENT kinds of primitive, is reported as a divergence error.
// TEST INFRASTRUCTURE ONLY.
#pragma once

#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

namespace rtk_simt {

constexpr int kLanes = 64;

struct Wave {
    ucontext_t sched;
    ucontext_t ctx[kLanes];
    std::vector<char> stacks;
    std::function<void(int)> body;
    int cur = -1;
    bool finished[kLanes] = {};
    bool waiting[kLanes] = {};
    int kind[kLanes] = {};            // 1 ballot, 2 readfirstlane
    bool vote[kLanes] = {};
    uint32_t xchg[kLanes] = {};
    uint64_t result_mask = 0;
    uint32_t result_first = 0;
    // statistics
    uint64_t sync_points = 0;
    bool error = false;
};

extern thread_local Wave* g_wave;

inline int lane_id() { return g_wave->cur; }

inline void park(int kind) {
    Wave* w = g_wave;
    const int l = w->cur;
    w->waiting[l] = true;
    w->kind[l] = kind;
    swapcontext(&w->ctx[l], &w->sched);
}

inline uint64_t ballot(bool p) {
    Wave* w = g_wave;
    w->vote[w->cur] = p;
    park(1);
    return w->result_mask;
}

inline uint32_t readfirstlane(uint32_t x) {
    Wave* w = g_wave;
    w->xchg[w->cur] = x;
    park(2);
    return w->result_first;
}

inline uint32_t mbcnt_lo(uint32_t m, uint32_t v) {
    const int l = lane_id();
    const uint32_t below = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u);
    return v + static_cast<uint32_t>(__builtin_popcount(m & below));
}
inline uint32_t mbcnt_hi(uint32_t m, uint32_t v) {
    const int l = lane_id();
    const uint32_t below = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
    return v + static_cast<uint32_t>(__builtin_popcount(m & below));
}

void trampoline(int lane);

// Runs body(lane) for lanes 0..63 as one wavefront on the calling thread.  Returns false on a divergence error.
bool run_wave(Wave& w, std::function<void(int)> body);

} // namespace rtk_simt
