// c2_k_common.h -- what the kernel units of this library share: DPP wave shift, the LDS plan of the row-strip kernel (sized on the host,
// carved in the kernel), the dynamic-LDS symbol, byte tables, the EXEC-masked region macro.  Kernel units: c2_k_align.hip, c2_k_classify.hip,
// c2_k_select.hip, c2_k_count.hip, c2_k_fastq.hip, c2_k_alleles.hip (each compiled in its own translation unit, see the Makefile).
#pragma once
#include <hip/hip_runtime.h>
#include "c2_device.h"

#define C2_DPP_WAVE_SHR1 0x138

// lane n receives `src` of lane n-1; lane 0 keeps `old`
__device__ __forceinline__ int c2_shr1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, C2_DPP_WAVE_SHR1, 0xf, 0xf, false);
}

__device__ __forceinline__ int c2_imax(int a, int b) { return a > b ? a : b; }
// sign-extended 4-bit field of x starting at bit `off`
__device__ __forceinline__ int c2_sbfe4(int x, int off) { return __builtin_amdgcn_sbfe(x, off, 4); }

struct c2_lds_plan {
    // byte offsets into dynamic LDS (all multiples of 16)
    uint32_t ptr, bnd, tbl, codeof, read, code, ref, incp, tmp_read, tmp_ref, total;
    uint32_t col_stride;  // halfwords per pointer column
};

__host__ __device__ inline uint32_t c2_align16(uint32_t x) { return (x + 15u) & ~15u; }

// The same function sizes the LDS on the host and carves it in the kernel.
// Banded pointer plane: at step t only the lanes lo(t) .. lo(t)+nslots-1 keep their pointer word, where
// lo(t) = floor((t - 1 - R*W) / (R+1)): lane l is at column j = t - l, the main-diagonal lane of that column is (j-1)/R,
// and |l - (j-1)/R| <= W  <=>  (R+1)*l within R*W of t-1.  The window is wave-uniform, so the in-band test and the
// LDS address of a step cost one subtract, one compare and one shift-add per lane.
__host__ __device__ inline int c2_band_slots(int R, int W) { return (2 * R * W) / (R + 1) + 2; }
__host__ __device__ inline int c2_band_lo(int R, int W, int t) { return (t - 1 - R * W + 64 * (R + 1)) / (R + 1) - 64; }

// band_lanes = 0: full plane (one row of 64 lanes per read column); > 0: banded plane (one row of nslots per step)
// plane_in_hbm: the pointer plane lives in a per-workgroup scratch area of HBM instead (c2_hbm_plane_halfwords: one row of
// 64 halfwords per STEP of the sweep, so that a step's 64 stores are one 128-byte line); LDS then only holds the O(Li + Lj) parts.
__host__ __device__ inline uint64_t c2_hbm_plane_halfwords(int max_lj, int max_passes) { return (uint64_t)max_passes * ((uint64_t)max_lj + 64u) * 64u; }
__host__ __device__ inline c2_lds_plan c2_make_plan(int R, int max_lj, int max_passes, int n_codes, int band_lanes, bool plane_in_hbm = false) {
    const int nslots = band_lanes > 0 ? c2_band_slots(R, band_lanes) : C2_LANES;
    const uint32_t plane_rows = band_lanes > 0 ? (uint32_t)max_lj + 64u : (uint32_t)max_lj;
    c2_lds_plan p;
    const uint32_t max_li = (uint32_t)max_passes * 64u * (uint32_t)R;
    p.col_stride = plane_in_hbm ? 64u : (uint32_t)nslots + C2_PTR_PAD;
    uint32_t off = 0;
    p.ptr = off;      off += plane_in_hbm ? 0u : c2_align16((uint32_t)max_passes * plane_rows * p.col_stride * 2u);
    p.bnd = off;      off += (max_passes > 1) ? c2_align16(3u * ((uint32_t)max_lj + 1u) * 4u) : 0u;
    p.tbl = off;      off += c2_align16((uint32_t)n_codes * (uint32_t)n_codes * 2u);
    p.codeof = off;   off += 256u;
    p.read = off;     off += c2_align16((uint32_t)max_lj);
    p.code = off;     off += c2_align16((uint32_t)max_lj);
    p.ref = off;      off += c2_align16(max_li);
    p.incp = off;     off += c2_align16((max_li + 2u) * 2u);
    p.tmp_read = off; off += c2_align16(max_li + (uint32_t)max_lj);
    p.tmp_ref = off;  off += c2_align16(max_li + (uint32_t)max_lj);
    p.total = off;
    return p;
}

extern __shared__ __attribute__((aligned(16))) unsigned char c2_smem[];

// (ch >> 1) & 7 is a perfect hash of A C T G - N (0 1 2 3 6 7; lower case lands on the same slots): byte tables as 64-bit constants
#define C2_BYTE_TABLE(a0, a1, a2, a3, a6, a7) ((unsigned long long)(a0) | ((unsigned long long)(a1) << 8) | ((unsigned long long)(a2) << 16) | \
                                               ((unsigned long long)(a3) << 24) | ((unsigned long long)(a6) << 48) | ((unsigned long long)(a7) << 56))
// complement of an upper-cased read character, 0 outside ACGTN_- (CRISPRessoShared.reverse_complement's dictionary).  A table look-up,
// not a switch: the compiler lowers a switch on a per-lane value to a tree of divergent branches (see c2_base_vector).  (c >> 1) & 7
// sends A/a C/c T/t G/g - N/n to 0 1 2 3 6 7; '_' shares N's slot and is tested by itself.
__device__ __forceinline__ unsigned c2_fq_complement(const unsigned c) {
    const unsigned h = (c >> 1) & 7u, sh = h * 8u;
    const unsigned is = (unsigned)(C2_BYTE_TABLE('A', 'C', 'T', 'G', '-', 'N') >> sh) & 0xffu;
    const unsigned to = (unsigned)(C2_BYTE_TABLE('T', 'G', 'A', 'C', '-', 'N') >> sh) & 0xffu;
    const bool letter = h != 6u && is != 0u && (c | 0x20u) == (is | 0x20u);
    return c == '_' ? (unsigned)'_' : (letter || c == '-') ? to : 0u;
}


// Region executed with some lanes switched off in EXEC for its whole length (one s_and_saveexec; no per-instruction
// cost).  The wave emulator (tests/emu) supplies its own definition, which parks the inactive fibers.
#ifndef C2_LANES_ACTIVE_BEGIN
#define C2_LANES_ACTIVE_BEGIN(cond) if (cond) {
#define C2_LANES_ACTIVE_END() }
#endif
// LDS-DMA (global_load_lds: the memory system writes the loaded bytes to LDS at a wave-uniform base + lane * size; no VGPR holds them).
// C2_WAIT_LDS_DMA: every copy this wavefront issued has landed; C2_LDS_READS_DONE: every LDS read it issued has returned (a slot may be
// overwritten).  The wave emulator (tests/emu) copies lane by lane and makes both a rendezvous of the wavefront's lanes.
#ifndef C2_WAIT_LDS_DMA
#define C2_WAIT_LDS_DMA() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#define C2_LDS_READS_DONE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

