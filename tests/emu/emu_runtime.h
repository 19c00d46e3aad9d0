// Fiber scheduler for the wave emulator (see hip/hip_runtime.h). TEST INFRASTRUCTURE ONLY.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <vector>

namespace emu {
int cur_lane = 0, n_threads = W;
dim3_ block_idx, grid_dim, block_dim;
ucontext_t lane_ctx[MAXT], sched_ctx;
bool lane_done[MAXT];
jmp_buf lane_jb[MAXT];
bool lane_has_jb[MAXT];
uint64_t xl_slots[2][MAXT];
int bar_count = 0, wbar_count[MAXW];
unsigned bar_gen = 0, wbar_gen[MAXW];
int active_count[MAXW], rconv_count[MAXW];
unsigned rconv_gen[MAXW];
bool lane_inactive[MAXT];
static std::function<void()>* cur_body = nullptr;
static std::vector<char> stacks;

static void lane_entry() {
    (*cur_body)();
    lane_done[cur_lane] = true;
    // hand control to any unfinished lane, else back to the scheduler
    for (int k = 1; k <= n_threads; ++k) {
        int c = (cur_lane + k) % n_threads;
        if (!lane_done[c]) { cur_lane = c; switch_to(c); }          // this fiber is finished: nothing to save
    }
    setcontext(&sched_ctx);
}

template <class F>
void launch(unsigned grid, F body, unsigned block) {
    std::function<void()> fn = body;
    cur_body = &fn;
    const size_t STK = 256 * 1024;
    if (block == 0 || block % W || block > (unsigned)MAXT) { fprintf(stderr, "emu: unsupported block size %u\n", block); abort(); }
    n_threads = (int)block;
    stacks.resize(STK * n_threads);
    grid_dim.x = grid; block_dim.x = block;
    for (unsigned b = 0; b < grid; ++b) {
        block_idx.x = b;
        bar_count = 0;
        for (int w = 0; w < MAXW; ++w) { wbar_count[w] = 0; active_count[w] = W; rconv_count[w] = 0; }
        for (int l = 0; l < MAXT; ++l) lane_inactive[l] = false;
        for (int l = 0; l < n_threads; ++l) {
            lane_done[l] = false;
            lane_has_jb[l] = false;
            getcontext(&lane_ctx[l]);
            lane_ctx[l].uc_stack.ss_sp = stacks.data() + STK * l;
            lane_ctx[l].uc_stack.ss_size = STK;
            lane_ctx[l].uc_link = &sched_ctx;
            makecontext(&lane_ctx[l], (void (*)())lane_entry, 0);
        }
        cur_lane = 0;
        volatile bool started = false;
        getcontext(&sched_ctx);
        if (!started) { started = true; setcontext(&lane_ctx[0]); }
        for (int l = 0; l < n_threads; ++l) if (!lane_done[l]) { fprintf(stderr, "emu: lane %d never finished\n", l); abort(); }
    }
}
}  // namespace emu
