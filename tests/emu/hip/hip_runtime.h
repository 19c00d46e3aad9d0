// Wave emulator -- TEST INFRASTRUCTURE ONLY (tests/emu/).
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile crispresso2_amd/csrc/c2_kernels.hip
// *unchanged* for the host and run one workgroup (1..4 wavefronts of 64 lanes) as cooperative fibers (ucontext),
// so the kernel's indexing / tie-breaking / traceback logic can be checked against the oracle
// in the GPU-less container.  Cross-lane builtins (DPP wave shift, ballot, readlane, shuffles)
// and __syncthreads() are rendezvous points: they must sit in wave-uniform control flow, which
// is also what the real hardware needs.  It says nothing about speed or about hardware-specific
// behaviour; the -m gpu tests are the parity tests proper.
#pragma once
#include <ucontext.h>
#include <setjmp.h>
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
constexpr int W = 64;            // wavefront width
constexpr int MAXT = 512;        // threads per workgroup the emulator supports
constexpr int MAXW = MAXT / W;
struct dim3_ { unsigned x = 1, y = 1, z = 1; };
extern int cur_lane;             // thread index inside the workgroup (wave = cur_lane / 64, lane = cur_lane % 64)
extern int n_threads;
extern dim3_ block_idx, grid_dim, block_dim;
extern ucontext_t lane_ctx[MAXT], sched_ctx;
extern bool lane_done[MAXT];
// A fiber that has run is suspended / resumed with _setjmp / _longjmp (no signal-mask system call per switch, unlike
// swapcontext: a third of the emulator's run time); ucontext only starts fibers.  Built with _FORTIFY_SOURCE off (the
// fortified longjmp refuses to jump between stacks).
extern jmp_buf lane_jb[MAXT];
extern bool lane_has_jb[MAXT];
inline void switch_to(int nxt) {              // resume (or start) fiber nxt; does not return
    if (lane_has_jb[nxt]) _longjmp(lane_jb[nxt], 1);
    setcontext(&lane_ctx[nxt]);
}
extern uint64_t xl_slots[2][MAXT];
extern int bar_count, wbar_count[MAXW];
extern unsigned bar_gen, wbar_gen[MAXW];
extern int active_count[MAXW], rconv_count[MAXW];   // lanes of a wave inside a C2_LANES_ACTIVE region / arrived at its end
extern unsigned rconv_gen[MAXW];
extern bool lane_inactive[MAXT];

inline void yield_next() {
    int me = cur_lane;
    int nxt = me;
    for (int k = 1; k <= n_threads; ++k) {
        int c = (me + k) % n_threads;
        if (!lane_done[c]) { nxt = c; break; }
    }
    if (nxt == me) { fprintf(stderr, "emu: deadlock (non-uniform cross-lane op?)\n"); abort(); }
    cur_lane = nxt;
    lane_has_jb[me] = true;
    if (_setjmp(lane_jb[me]) == 0) switch_to(nxt);
}
inline void barrier() {                       // workgroup barrier (s_barrier)
    unsigned gen = bar_gen;
    if (++bar_count == n_threads) { bar_count = 0; bar_gen++; return; }
    while (bar_gen == gen) yield_next();
}
inline void wave_barrier() {                  // rendezvous of the 64 lanes of one wavefront (lock-step execution)
    const int w = cur_lane / W;
    unsigned gen = wbar_gen[w];
    if (++wbar_count[w] >= active_count[w]) { wbar_count[w] = 0; wbar_gen[w]++; return; }
    while (wbar_gen[w] == gen) yield_next();
}
// EXEC-masked region (C2_LANES_ACTIVE_BEGIN / _END in the kernels): lanes whose condition is false skip the region and
// wait at its end; inside, cross-lane operations rendezvous among the active lanes only, and a DPP move whose source lane
// is inactive leaves the destination untouched (gfx9 DPP semantics with bound_ctrl = 0).  Not nestable.
struct exec_scope {
    bool active;
    explicit exec_scope(bool a) : active(a) {
        if (a) return;
        const int w = cur_lane / W;
        lane_inactive[cur_lane] = true;
        xl_slots[1][cur_lane] = 0;
        --active_count[w];
        if (wbar_count[w] > 0 && wbar_count[w] >= active_count[w]) { wbar_count[w] = 0; wbar_gen[w]++; }
    }
    ~exec_scope() {
        const int w = cur_lane / W;
        const unsigned gen = rconv_gen[w];
        if (++rconv_count[w] == W) {
            rconv_count[w] = 0; active_count[w] = W;
            for (int l = 0; l < W; ++l) lane_inactive[w * W + l] = false;
            rconv_gen[w]++;
            return;
        }
        while (rconv_gen[w] == gen) yield_next();
    }
};
inline int lane_id() { return cur_lane % W; }
inline int wave_base() { return cur_lane - cur_lane % W; }
struct tid_t { unsigned x, y, z; };
inline tid_t tid() { return tid_t{(unsigned)cur_lane, 0, 0}; }
}  // namespace emu

#define C2_LANES_ACTIVE_BEGIN(cond) { emu::exec_scope c2_exec_scope_(cond); if (c2_exec_scope_.active) {
#define C2_LANES_ACTIVE_END() } }
struct uint4 { unsigned x, y, z, w; };
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))

#define threadIdx (emu::tid())
#define blockIdx (emu::block_idx)
#define gridDim (emu::grid_dim)
#define blockDim (emu::block_dim)

inline void __syncthreads() { emu::barrier(); }
inline void __threadfence_block() {}
inline void __builtin_amdgcn_wave_barrier() { emu::wave_barrier(); }

// exchange helper with phase alternation handled per call site (two tables, flipped by lane 0 after barrier 2)
namespace emu {
inline uint64_t xl_get(uint64_t mine, int src_lane, bool* valid) {     // src_lane: lane inside this thread's wavefront
    xl_slots[0][cur_lane] = mine;
    wave_barrier();
    uint64_t r = 0; bool ok = src_lane >= 0 && src_lane < W && !lane_inactive[wave_base() + src_lane];
    if (ok) r = xl_slots[0][wave_base() + src_lane];
    if (valid) *valid = ok;
    wave_barrier();                 // everyone has read before anyone may overwrite
    return r;
}
}

// DPP: only the controls the kernels use
inline int __builtin_amdgcn_update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)row_mask; (void)bank_mask;
    int from;
    bool in_row = true;
    if (dpp_ctrl == 0x138) from = emu::lane_id() - 1;        // wave_shr:1
    else if (dpp_ctrl == 0x130) from = emu::lane_id() + 1;   // wave_shl:1
    else if (dpp_ctrl == 0x111) { from = emu::lane_id() - 1; in_row = (emu::lane_id() & 15) != 0; }    // row_shr:1 (rows of 16 lanes)
    else if (dpp_ctrl == 0x101) { from = emu::lane_id() + 1; in_row = (emu::lane_id() & 15) != 15; }   // row_shl:1
    else { fprintf(stderr, "emu: unsupported dpp_ctrl %x\n", dpp_ctrl); abort(); }
    bool ok; uint64_t v = emu::xl_get((uint32_t)src, in_row ? from : -1, &ok);
    if (!ok || !in_row) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)v;
}
inline long long clock64() { return 0; }
// LDS-DMA: lane l's `size` bytes land at dst + off + l * size (dst wave-uniform); the waits are rendezvous of the wavefront's lanes
inline void __builtin_amdgcn_global_load_lds(const void* src, void* dst, int size, int off, int aux) {
    (void)aux; memcpy((char*)dst + off + emu::lane_id() * size, src, (size_t)size);
}
#define C2_WAIT_LDS_DMA() emu::wave_barrier()
#define C2_LDS_READS_DONE() emu::wave_barrier()
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned shift) { return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (8 * (shift & 3))); }
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned shift) { return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (shift & 31)); }
// v_perm_b32 for selectors 0..7: byte i of the result = byte sel[i] of the 64-bit {hi, lo}
inline uint32_t __builtin_amdgcn_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned k = (sel >> (8 * i)) & 0xffu;
        if (k > 7) { fprintf(stderr, "emu: unsupported v_perm selector %u\n", k); abort(); }
        r |= (uint32_t)((v >> (8 * k)) & 0xffu) << (8 * i);
    }
    return r;
}
inline int __builtin_amdgcn_sbfe(int x, int off, int width) { return (int)((unsigned)x << (32 - off - width)) >> (32 - width); }
inline unsigned long long __ballot(int pred) {
    emu::xl_slots[1][emu::cur_lane] = pred ? 1 : 0;
    emu::wave_barrier();
    unsigned long long m = 0;
    for (int l = 0; l < emu::W; ++l) if (emu::xl_slots[1][emu::wave_base() + l]) m |= 1ull << l;
    emu::wave_barrier();
    return m;
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)emu::xl_get((uint32_t)v, lane, nullptr); }
inline int __builtin_amdgcn_writelane(int v, int lane, int old) { return emu::lane_id() == lane ? v : old; }
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)emu::xl_get((uint32_t)v, 0, nullptr); }
inline int __shfl(int v, int lane, int width = 64) { (void)width; return (int)(uint32_t)emu::xl_get((uint32_t)v, lane & 63, nullptr); }
inline int __shfl_xor(int v, int mask, int width = 64) { (void)width; return (int)(uint32_t)emu::xl_get((uint32_t)v, emu::lane_id() ^ mask, nullptr); }
inline int __shfl_down(int v, unsigned d, int width = 64) { (void)width; int s = emu::lane_id() + (int)d; bool ok; uint64_t r = emu::xl_get((uint32_t)v, s, &ok); return ok ? (int)(uint32_t)r : v; }
inline int __shfl_up(int v, unsigned d, int width = 64) { (void)width; int s = emu::lane_id() - (int)d; bool ok; uint64_t r = emu::xl_get((uint32_t)v, s, &ok); return ok ? (int)(uint32_t)r : v; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <class T, class U> inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }

namespace emu {
// run `body()` as a grid of workgroups of `block` threads (a multiple of 64), sequentially
template <class F>
void launch(unsigned grid, F body, unsigned block = 64);
}
