#include "rtk_simt.hpp"

namespace rtk_simt {

thread_local Wave* g_wave = nullptr;

void trampoline(int lane) {
    Wave* w = g_wave;
    w->body(lane);
    w->finished[lane] = true;
    swapcontext(&w->ctx[lane], &w->sched);
}

bool run_wave(Wave& w, std::function<void(int)> body) {
    constexpr size_t kStack = 256 * 1024;
    w.stacks.assign(kStack * kLanes, 0);
    w.body = std::move(body);
    g_wave = &w;
    for (int l = 0; l < kLanes; ++l) {
        w.finished[l] = w.waiting[l] = false;
        getcontext(&w.ctx[l]);
        w.ctx[l].uc_stack.ss_sp = w.stacks.data() + kStack * l;
        w.ctx[l].uc_stack.ss_size = kStack;
        w.ctx[l].uc_link = &w.sched;
        makecontext(&w.ctx[l], reinterpret_cast<void (*)()>(trampoline), 1, l);
    }
    for (;;) {
        // one round: every live lane runs to its next wave-level primitive (or to the end of the kernel)
        int live = 0, parked = 0, kind = 0;
        bool mixed = false;
        for (int l = 0; l < kLanes; ++l) {
            if (w.finished[l]) continue;
            w.cur = l;
            w.waiting[l] = false;
            swapcontext(&w.sched, &w.ctx[l]);
            if (w.finished[l]) continue;
            ++live;
            if (w.waiting[l]) {
                ++parked;
                if (kind == 0) kind = w.kind[l];
                else if (kind != w.kind[l]) mixed = true;
            }
        }
        if (live == 0) break;
        bool some_finished_now = false;
        for (int l = 0; l < kLanes; ++l) some_finished_now |= w.finished[l];
        if (mixed || parked != live || some_finished_now) {
            // a vote reached by only part of the wave: the kernel's control flow was not wave-uniform there
            std::fprintf(stderr, "rtk_simt: divergent wave-level primitive (live %d parked %d mixed %d finished %d)\n", live, parked, mixed, some_finished_now);
            w.error = true;
            g_wave = nullptr;
            return false;
        }
        ++w.sync_points;
        uint64_t mask = 0;
        int first = -1;
        for (int l = 0; l < kLanes; ++l) {
            if (w.finished[l]) continue;
            if (first < 0) first = l;
            if (w.vote[l]) mask |= 1ull << l;
        }
        w.result_mask = mask;
        w.result_first = w.xchg[first];
    }
    g_wave = nullptr;
    return true;
}

} // namespace rtk_simt
