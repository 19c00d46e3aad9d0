// Wave emulator -- TEST INFRASTRUCTURE ONLY (tests/emu/).
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile crispresso2_amd/csrc/c2_kernels.hip
// *unchanged* for the host and run one 64-lane workgroup as 64 cooperative fibers (ucontext),
// so the kernel's indexing / tie-breaking / traceback logic can be checked against the oracle
// in the GPU-less container.  Cross-lane builtins (DPP wave shift, ballot, readlane, shuffles)
// and __syncthreads() are rendezvous points: they must sit in wave-uniform control flow, which
// is also what the real hardware needs.  It says nothing about speed or about hardware-specific
// behaviour; the -m gpu tests are the parity tests proper.
#pragma once
#include <ucontext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define __restrict__

namespace emu {
constexpr int W = 64;
struct dim3_ { unsigned x = 1, y = 1, z = 1; };
extern int cur_lane;
extern dim3_ block_idx, grid_dim, block_dim;
extern ucontext_t lane_ctx[W], sched_ctx;
extern bool lane_done[W];
extern uint64_t xl_slots[2][W];
extern int xl_phase;
extern int bar_count;
extern unsigned bar_gen;

inline void yield_next() {
    int me = cur_lane;
    int nxt = me;
    for (int k = 1; k <= W; ++k) {
        int c = (me + k) % W;
        if (!lane_done[c]) { nxt = c; break; }
    }
    if (nxt == me) { fprintf(stderr, "emu: deadlock (non-uniform cross-lane op?)\n"); abort(); }
    cur_lane = nxt;
    swapcontext(&lane_ctx[me], &lane_ctx[nxt]);
}
inline void barrier() {
    unsigned gen = bar_gen;
    if (++bar_count == W) { bar_count = 0; bar_gen++; return; }
    while (bar_gen == gen) yield_next();
}
struct tid_t { unsigned x, y, z; };
inline tid_t tid() { return tid_t{(unsigned)cur_lane, 0, 0}; }
}  // namespace emu

#define threadIdx (emu::tid())
#define blockIdx (emu::block_idx)
#define gridDim (emu::grid_dim)
#define blockDim (emu::block_dim)

inline void __syncthreads() { emu::barrier(); }

// exchange helper with phase alternation handled per call site (two tables, flipped by lane 0 after barrier 2)
namespace emu {
inline uint64_t xl_get(uint64_t mine, int src_lane, bool* valid) {
    int ph = xl_phase;
    xl_slots[ph][cur_lane] = mine;
    barrier();
    uint64_t r = 0; bool ok = src_lane >= 0 && src_lane < W;
    if (ok) r = xl_slots[ph][src_lane];
    if (valid) *valid = ok;
    barrier();                      // everyone has read before anyone may overwrite
    return r;
}
}

// DPP: only the controls the kernels use
inline int __builtin_amdgcn_update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)row_mask; (void)bank_mask;
    int from;
    if (dpp_ctrl == 0x138) from = emu::cur_lane - 1;        // wave_shr:1
    else if (dpp_ctrl == 0x130) from = emu::cur_lane + 1;   // wave_shl:1
    else { fprintf(stderr, "emu: unsupported dpp_ctrl %x\n", dpp_ctrl); abort(); }
    bool ok; uint64_t v = emu::xl_get((uint32_t)src, from, &ok);
    if (!ok) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)v;
}
inline long long clock64() { return 0; }
inline int __builtin_amdgcn_sbfe(int x, int off, int width) { return (int)((unsigned)x << (32 - off - width)) >> (32 - width); }
inline unsigned long long __ballot(int pred) {
    emu::xl_slots[0][emu::cur_lane] = pred ? 1 : 0;
    emu::barrier();
    unsigned long long m = 0;
    for (int l = 0; l < emu::W; ++l) if (emu::xl_slots[0][l]) m |= 1ull << l;
    emu::barrier();
    return m;
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)emu::xl_get((uint32_t)v, lane, nullptr); }
inline int __builtin_amdgcn_writelane(int v, int lane, int old) { return emu::cur_lane == lane ? v : old; }
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)emu::xl_get((uint32_t)v, 0, nullptr); }
inline int __shfl(int v, int lane, int width = 64) { (void)width; return (int)(uint32_t)emu::xl_get((uint32_t)v, lane & 63, nullptr); }
inline int __shfl_xor(int v, int mask, int width = 64) { (void)width; return (int)(uint32_t)emu::xl_get((uint32_t)v, emu::cur_lane ^ mask, nullptr); }
inline int __shfl_down(int v, unsigned d, int width = 64) { (void)width; int s = emu::cur_lane + (int)d; bool ok; uint64_t r = emu::xl_get((uint32_t)v, s, &ok); return ok ? (int)(uint32_t)r : v; }
inline int __shfl_up(int v, unsigned d, int width = 64) { (void)width; int s = emu::cur_lane - (int)d; bool ok; uint64_t r = emu::xl_get((uint32_t)v, s, &ok); return ok ? (int)(uint32_t)r : v; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }

namespace emu {
// run `body()` as a grid of single-wave workgroups, sequentially
template <class F>
void launch(unsigned grid, F body);
}
