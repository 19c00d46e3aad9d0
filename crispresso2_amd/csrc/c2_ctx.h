// c2_ctx.h -- the library context shared by the host-side translation units (c2_api_*.hip): device buffers it owns, the state
// c2_set_scoring / c2_set_refs leave behind, error text.  Internal: the C ABI (include/crispresso2_amd.h) only sees an opaque c2_ctx.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <memory>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the library is bound with dlopen (the copy PyTorch-ROCm already loaded, if any)

#include "crispresso2_amd.h"
#include "c2_device.h"
#include "c2_host_prep.h"

extern std::string g_create_error;     // what c2_last_error(NULL) returns (defined in c2_api_align.hip)
extern std::mutex g_mutex;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct TimedLaunch { hipEvent_t a, m0, m, b; };   // before the chain, before / after its dominant kernel (the first band tier's), after its last

struct c2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    std::string err;
    // scoring
    bool have_scoring = false;
    c2_scoring_tables sc;
    int gap_open = -1, gap_extend = -1;
    DevBuf d_tbl, d_code, d_pk;
    std::vector<int64_t> matrix_copy;   // to skip re-upload when c2_global_align is called with the same matrix
    // refs
    int n_refs = 0;
    int max_li = 0;
    std::vector<int> ref_len;
    DevBuf d_refblob, d_refdesc;
    // host copies needed to (re)build the diagonal-band kernel's row tables when refs or scoring change
    std::vector<std::string> ref_seq;
    std::vector<std::vector<int32_t>> ref_g32;
    std::vector<c2_dev_ref> ref_desc;
    DevBuf d_diagrows, d_diagrows_pk;   // row tables of the diagonal kernels: 32-bit records, and the packed (int16 pair) ones at the same indices
    bool diag_rows_dirty = true;
    std::vector<uint8_t> ref_pk_ok;     // per reference: admitted to the packed fill (c2_pk_eligible)
    bool any_pk_ok = false;
    int pk_bias = 0;                    // ... and this value bias (c2_pk_add32_bias_needed)
    int pk_beta = 0;                    // > 0: the admitted references run the packed kernels' 32-bit-add variant with this bias (c2_pk_add32_ok)
    bool pk_dirty = true;
    int occ_score_lds = -1, occ_score_blocks = 0;            // residency of c2_align_diags_kernel with its LDS plan
    int occ_p16_lds = -1, occ_p16_blocks = 0;                // ... of c2_align_diagp_kernel<16>
    int occ_pk_lds = -1, occ_pk_blocks = 0, occ_pk2_lds = -1, occ_pk2_blocks = 0, occ_pk3_lds = -1, occ_pk3_blocks = 0;
    int occ_pk6_lds = -1, occ_pk6_blocks = 0;                // ... of c2_align_diagp_kernel<6> (the 40-diagonal tier)
    // staging for the host batch path and the per-call path
    DevBuf d_reads, d_offsets, d_refids, d_strands, d_aln_read, d_aln_ref, d_records, d_misc;
    // timing
    bool timing = false;
    std::vector<TimedLaunch> timed;
    // LDS opt-in already requested for these kernels
    // optional per-phase cycle accounting (c2_phase_profile)
    bool phase_prof = false;
    DevBuf d_phase;
    // banded first launch: -1 auto, 0 off, >0 lanes each side; fallback list buffer
    int band_setting = -1;
    int band_target_wgs = 14;
    // 0 auto (diagonal-band tiers 4 -> 2 -> 1 alignments per wavefront when applicable), 1 banded row-strip, 2 full row-strip,
    // 3 single-alignment diagonal-band kernel only, 4 tiers 2 -> 1
    int kernel_mode = 0;
    int gmax = 0;          // largest gap incentive over the references
    DevBuf d_fb;
    int occ_diag_lds = -1, occ_diag_blocks = 0;
    int occ_x_lds[2] = {-1, -1}, occ_x_blocks[2] = {0, 0};   // [0] 4 alignments per wavefront, [1] 2
    DevBuf d_plane;        // pointer-word scratch of the multi-alignment diagonal kernels
    DevBuf d_plane16;      // ... of c2_align_diagp_kernel<16> (its residency is its own)
    DevBuf d_cnt_block;            // count route: the workgroups' accumulator blocks when they do not fit LDS
    DevBuf d_lists, d_lists_out;   // batched classifier: staging and flat output
    DevBuf d_order;        // count kernel: histogram + tasks grouped by reference
    DevBuf d_order2;       // ... several references with hints: per-reference counters and starts + the tasks the hinted kernel left, in every reference's own range
    int last_tiers = 0;    // banded launches in front of the full-plane launch in the last run_align
    bool last_score_stage = false, last_p16_stage = false; uint64_t last_n_tasks = 0;   // the last run_align: did the score-only stage run, over how many tasks in all
    int occ_lds[5][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
    int occ_blocks[5][3] = {};
    DevBuf d_cnt;          // count kernel: work counter + min_matches table
    DevBuf d_sel;          // selection kernel: per-reference score thresholds
    DevBuf d_seeds;        // strand-plan kernel: seed bytes and tables
    std::vector<uint8_t> seeds_host;   // ... and what they hold (the staging block of the last c2_strand_plan_device call)
    ncclComm_t comm = nullptr; // RCCL communicator of c2_comm_init (one rank per GPU)
    int comm_world = 0;
    std::vector<uint32_t> sel_table;
    std::vector<uint16_t> cnt_table;   // host copy of the table that is on the device (skip re-upload when unchanged)
    // host batch path, pipelined: pinned staging (two sets), copy streams and their events
    void* pin_in[2] = {nullptr, nullptr}; size_t pin_in_cap[2] = {0, 0};
    void* pin_out[2] = {nullptr, nullptr}; size_t pin_out_cap[2] = {0, 0};
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
};

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return C2_E_DEVICE;                                                             \
        }                                                                                   \
    } while (0)

inline int ensure(c2_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = std::max<size_t>(bytes, 256);
    want = (want + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return C2_E_NOMEM; }
    b.cap = want;
    return 0;
}

inline void release(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

inline int ensure_pinned(c2_ctx* ctx, void*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->err = std::string("hipHostMalloc: ") + hipGetErrorString(e); p = nullptr; return C2_E_NOMEM; }
    cap = want;
    return 0;
}

// memcpy on a few threads (pinned staging <-> the caller's pageable arrays: one thread moves ~10 GB/s, the link more)
inline void copy_parallel(void* dst, const void* src, size_t n, unsigned threads) {
    if (threads < 2 || n < ((size_t)4 << 20)) { memcpy(dst, src, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) {
        const size_t a = n * t / threads, z = n * (t + 1) / threads;
        pool.emplace_back([=] { memcpy((char*)dst + a, (const char*)src + a, z - a); });
    }
    for (auto& th : pool) th.join();
}

