// c2_k_count.hip -- aligned strings + records -> the per-amplicon count tensor.
#pragma once
#include "c2_k_common.h"

// =====================================================================================
// Per-amplicon count vectors: the device side of the reference's "Quantifying indels/substitutions"
// loop (CRISPRessoCORE.py:3964-4115, non-coding case) and of process_fastq's aln_stats (:1974-1979).
// Input: the aligned strings and records the align kernel left in HBM, plus per-task weights
// (read multiplicity; 0 = read not assigned to this reference).  One wavefront per alignment;
// every workgroup accumulates into a private int32 copy of one reference's block in LDS (ds_add),
// and flushes it to the int64 tensor in HBM with one atomic per non-zero entry.  The tensor is what
// the multi-GPU path reduces with one RCCL all-reduce.
// =====================================================================================
__device__ __forceinline__ int c2_wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}

// all_base_count vector of a read / reference character (CRISPRessoCORE.py:4075-4081), or -1.  Without a branch: the compiler turned
// the chain of comparisons this used to be into a tree of DIVERGENT branches -- ~50 scalar instructions of exec-mask bookkeeping per call,
// two calls per mismatching column -- and the count kernels are bound by the CU's one scalar unit (244 SALU per alignment,
// profiles/r03/README.md).  (ch >> 1) & 7 is a perfect hash of A C T G - N (0 1 2 3 6 7); the byte tables are 64-bit constants.
__device__ __forceinline__ int c2_base_vector(const unsigned char ch) {
    const unsigned sh = (((unsigned)ch >> 1) & 7u) * 8u;
    const unsigned is = (unsigned)(C2_BYTE_TABLE('A', 'C', 'T', 'G', '-', 'N') >> sh) & 0xffu;
    const unsigned v = (unsigned)(C2_BYTE_TABLE(C2_V_BASE_A, C2_V_BASE_C, C2_V_BASE_T, C2_V_BASE_G, C2_V_BASE_GAP, C2_V_BASE_N) >> sh) & 0xffu;
    return is == (unsigned)ch ? (int)v : -1;
}
// substitution_count_vectors of a read base (:4049-4054): A C G T only, else -1
__device__ __forceinline__ int c2_sub_base_vector(const unsigned char ch) {
    const unsigned sh = (((unsigned)ch >> 1) & 7u) * 8u;
    const unsigned is = (unsigned)(C2_BYTE_TABLE('A', 'C', 'T', 'G', 0, 0) >> sh) & 0xffu;
    const unsigned v = (unsigned)(C2_BYTE_TABLE(C2_V_ALL_SUB_BASE_A, C2_V_ALL_SUB_BASE_C, C2_V_ALL_SUB_BASE_T, C2_V_ALL_SUB_BASE_G, 0, 0) >> sh) & 0xffu;
    return (ch != 0 && is == (unsigned)ch) ? (int)v : -1;
}

// Tasks grouped by reference for the count kernel (a chunk of consecutive positions then holds one or two references
// instead of dozens: with 96 interleaved amplicons the LDS block would be flushed for almost every task).  Counting sort
// in three tiny launches: histogram of ref_id, exclusive scan (one wavefront), scatter.  The order inside a reference is
// whatever the atomics give -- the sums do not depend on it.
// One atomic per distinct reference per wavefront (wave-aggregated): lanes with the same ref_id are found with ballots,
// their leader adds the group's size, every lane gets base + its rank inside the group.  Returns the lane's slot (or -1).
__device__ __forceinline__ long long c2_grouped_add(uint32_t* counters, const bool active, const unsigned key, const int lane) {
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned long long todo = __ballot(active);
    long long slot = -1;
    for (int round = 0; round < 4 && todo; ++round) {       // the big groups; what is left after four rounds is scattered
        const int leader = __builtin_ctzll(todo);
        const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long grp = __ballot(active && key == k0) & todo;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(counters + k0, (unsigned)__popcll(grp));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
        if ((grp >> lane) & 1ull) slot = (long long)base + __popcll(grp & lt);
        todo &= ~grp;
    }
    if ((todo >> lane) & 1ull) slot = (long long)atomicAdd(counters + key, 1u);
    return slot;
}

__global__ __launch_bounds__(256) void c2_ref_histogram_kernel(const c2_aln_record* records, uint64_t n, uint32_t* hist)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool active = t < n;
    const unsigned key = active ? (unsigned)records[t].ref_id : 0u;
    (void)c2_grouped_add(hist, active, key, (int)(threadIdx.x & 63));
}

__global__ __launch_bounds__(64) void c2_ref_scan_kernel(uint32_t* hist, int n_refs)     // hist -> exclusive prefix, in place
{
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n_refs; base += 64) {
        const int k = base + lane;
        const int x = (k < n_refs) ? (int)hist[k] : 0;
        const int s = c2_wave_incl_scan(x, lane) + carry;
        if (k < n_refs) hist[k] = (uint32_t)(s - x);
        carry = __shfl(s, 63);
    }
}

__global__ __launch_bounds__(256) void c2_ref_scatter_kernel(const c2_aln_record* records, uint64_t n, uint32_t* cursor, uint32_t* order)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool active = t < n;
    const unsigned key = active ? (unsigned)records[t].ref_id : 0u;
    const long long slot = c2_grouped_add(cursor, active, key, (int)(threadIdx.x & 63));
    if (active) order[slot] = (uint32_t)t;
}

// Behind c2_count_hinted_kernel with several references: the tasks it left lie in every reference's own range of `ranged` (cnt[r] of them from the range's
// start, ends[r - 1]); one wavefront turns the counts into starts (pre) and their sum (*total), then the ranges are copied back to back into `dense` --
// still grouped by reference, which is what keeps c2_count_vectors_kernel's flushes few.
__global__ __launch_bounds__(64) void c2_rest_scan_kernel(const uint32_t* cnt, uint32_t* pre, uint32_t* total, int n_refs)
{
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n_refs; base += 64) {
        const int k = base + lane;
        const int x = (k < n_refs) ? (int)cnt[k] : 0;
        const int s = c2_wave_incl_scan(x, lane) + carry;
        if (k < n_refs) pre[k] = (uint32_t)(s - x);
        carry = __shfl(s, 63);
    }
    if (lane == 0) *total = (uint32_t)carry;
}
__global__ __launch_bounds__(256) void c2_rest_compact_kernel(const uint32_t* cnt, const uint32_t* pre, const uint32_t* ends, uint32_t per_ref, const uint32_t* ranged, uint32_t* dense, unsigned gx)
{
    const unsigned r = blockIdx.x / gx, j = blockIdx.x - r * gx;
    const uint32_t n = cnt[r], from = ends ? (r ? ends[r - 1] : 0u) : r * per_ref, to = pre[r];      // (ends == NULL: ranges of per_ref positions each)
    for (uint32_t k = j * 256u + threadIdx.x; k < n; k += gx * 256u) dense[to + k] = ranged[from + k];
}

// The same grouping through LDS (round 6; up to C2_REF_LDS_MAX references): a workgroup counts its 4,096 tasks per reference in LDS and asks the global
// counters once per reference it holds -- tasks sorted by amplicon in the input (the pooled shape) cost a workgroup one or two atomics instead of one per
// wavefront and reference (histogram 1.09 -> ms, scatter 1.77 -> ms for 12.5 M tasks of 96 references, profiles/r06).
#define C2_REF_LDS_MAX 4096
#define C2_REF_CHUNK 4096
__global__ __launch_bounds__(256) void c2_ref_histogram_lds_kernel(const c2_aln_record* records, uint64_t n, uint32_t* hist, int n_refs)
{
    unsigned* lh = (unsigned*)c2_smem;                              // [n_refs]
    const int tid = threadIdx.x;
    for (int r = tid; r < n_refs; r += 256) lh[r] = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * C2_REF_CHUNK;
    for (int k = 0; k < C2_REF_CHUNK / 256; ++k) {
        const uint64_t t = base + (uint64_t)(k * 256 + tid);
        if (t < n) atomicAdd(&lh[records[t].ref_id], 1u);
    }
    __syncthreads();
    for (int r = tid; r < n_refs; r += 256) { const unsigned c = lh[r]; if (c) atomicAdd(hist + r, c); }
}
__global__ __launch_bounds__(256) void c2_ref_scatter_lds_kernel(const c2_aln_record* records, uint64_t n, uint32_t* cursor, uint32_t* order, int n_refs)
{
    unsigned* lh = (unsigned*)c2_smem;                              // [n_refs] tasks of the chunk per reference, then: the next free place of the reference's run
    const int tid = threadIdx.x;
    for (int r = tid; r < n_refs; r += 256) lh[r] = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * C2_REF_CHUNK;
    unsigned key[C2_REF_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < C2_REF_CHUNK / 256; ++k) {
        const uint64_t t = base + (uint64_t)(k * 256 + tid);
        key[k] = t < n ? (unsigned)records[t].ref_id : 0xffffffffu;
        if (t < n) atomicAdd(&lh[key[k]], 1u);
    }
    __syncthreads();
    for (int r = tid; r < n_refs; r += 256) { const unsigned c = lh[r]; lh[r] = c ? atomicAdd(cursor + r, c) : 0u; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < C2_REF_CHUNK / 256; ++k)
        if (key[k] != 0xffffffffu) order[atomicAdd(&lh[key[k]], 1u)] = (uint32_t)(base + (uint64_t)(k * 256 + tid));
}

// (launch bounds: 5 workgroups per CU = 5 waves per SIMD = 96 VGPRs.  Left alone the compiler takes 101 -- 99 + 2 that hold 103 spilled
// SGPRs -- and the kernel runs at 4 waves per SIMD, 8 % slower; with the bound it is 8 % slower than a 96-VGPR build WITHOUT the bound
// would be (measured with round 1's source, which fits by itself: the occupancy target changes the schedule), but that is not on offer.)
// HBM: the int32 accumulator block of the workgroup lives in global memory (A.block_scratch) instead of LDS -- amplicons beyond
// ~1,650 bp, whose block does not fit 160 KB.  Its updates are the same atomics (they execute in L2); what reads the block with plain
// loads -- the flush -- first drops the CU's L1 lines (agent-scope fence), and the flush takes every entry with an exchange.
#define C2_HCNT_SMALL_W 1024                   // c2_count_hinted_kernel takes a gapped hint below this weight (its int32 LDS block)
template <bool HBM>
__device__ __forceinline__ void c2_count_vectors_body(const c2_count_args& A)
{
    // C2_CNT_WAVES wavefronts share one LDS block (the block is what limits residency, so sharing it multiplies the
    // waves per CU); each wavefront walks one alignment at a time.  The workgroup takes C2_CNT_WAVES * C2_CNT_TASKS_PER_WAVE
    // consecutive tasks per atomic; lane k of wave v holds the record of task base + k * C2_CNT_WAVES + v.
    constexpr int NT = 64 * C2_CNT_WAVES, K = C2_CNT_TASKS_PER_WAVE, CHUNK = C2_CNT_WAVES * K, NONE = 0x7fffffff;
    static_assert(K == 32 || K == 64, "a wavefront's tasks are one bit each of a 64-bit mask");
    typedef unsigned long long km_t;                            // one bit per task of the wavefront (lane k holds the record of task k)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* acc = HBM ? A.block_scratch + (size_t)blockIdx.x * (size_t)A.block_ints : (int*)c2_smem;
    const int VL = A.lmax + 1;                                  // vector length incl. the end slot of the difference arrays
    const int o_sc = C2_CNT_VECTORS * VL, o_h = o_sc + C2_CNT_SCALARS;
    const int per_ref = o_h + C2_CNT_HISTS * A.hl;
    auto block_barrier = [&]() {                                // a barrier after which plain loads see the block's latest values
        __syncthreads();
        if (HBM) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    };
    // cov: difference array over the reference positions of "weight of the alignments whose read base EQUALS the reference's here" --
    // runs of matching columns add +w at their first position and -w behind their last; flush() integrates it and adds every
    // position's total to the count vector of the reference's own base there (an LDS-only vector, not part of the tensor)
    // dcov: the same for "weight of the alignments that have a DELETION column here" (walk of eight columns per lane): a deletion adds +w at its
    // first reference position and -w behind its last; flush() adds the integrated totals to all_deletion and to the '-' base counts
    int* cov = HBM ? (int*)c2_smem : acc + per_ref;
    int* dcov = cov + VL;
    int* ctl = dcov + VL;                                        // [0..1] chunk base, [2..] chunk weight per wave, [16..] two sets of (ref, task) per wave
    uint16_t* incp = (uint16_t*)(ctl + C2_CNT_CTL_INTS);        // inc_prefix of the current reference (lmax + 2 entries)
    // staging slots of the wavefronts (c2_count_lds_tail_bytes: behind cov, the control words and inc_prefix, that part padded to 16 bytes)
    uint8_t* stage = (uint8_t*)cov + (((2 * (size_t)VL + C2_CNT_CTL_INTS) * sizeof(int) + (((size_t)A.lmax + 2 + 1) / 2) * 4 + 15) / 16 * 16);
    const uint64_t n_positions = A.n_tasks_dev ? (uint64_t)(*A.n_tasks_dev) : A.n_tasks;      // (behind c2_count_hinted_kernel: the list of the tasks it left)
    for (int k = tid; k < per_ref; k += NT) acc[k] = 0;
    for (int k = tid; k < 2 * VL; k += NT) cov[k] = 0;                 // (cov and dcov)
    block_barrier();
    const bool ign_sub = A.flags & C2_CNT_FLAG_IGNORE_SUBSTITUTIONS, ign_ins = A.flags & C2_CNT_FLAG_IGNORE_INSERTIONS;
    const bool ign_del = A.flags & C2_CNT_FLAG_IGNORE_DELETIONS, discard = A.flags & C2_CNT_FLAG_DISCARD_INDEL_READS;
    const bool rows_aligned = ((((uintptr_t)A.aln_read | (uintptr_t)A.aln_ref) & 3u) == 0) && ((A.aln_stride & 3u) == 0);   // string rows readable as dwords
    const bool rows_aligned16 = ((((uintptr_t)A.aln_read | (uintptr_t)A.aln_ref) & 15u) == 0) && ((A.aln_stride & 15u) == 0);   // ... as 16-byte words
    const bool legacy = A.flags & C2_CNT_FLAG_LEGACY;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int cur_ref = -1;                                           // workgroup-uniform, like everything that guards a barrier
    unsigned wsum = 0;                                          // load accumulated since the last flush (int32 safety, see C2_CNT_LOAD_BUDGET)
    int Li = 0, par = 0;

    // flush the LDS block of cur_ref into the int64 tensor (workgroup-wide)
    auto flush = [&]() {
        if (cur_ref < 0) return;
        block_barrier();
        // deletion vectors were accumulated as difference arrays (start += x, end -= x): integrate them first
        static_assert(C2_CNT_WAVES >= 4, "four difference arrays are integrated by four wavefronts");
        if (wave < 4) {
            int* d = wave == 3 ? dcov : wave == 2 ? cov : acc + (wave == 0 ? C2_V_DELETION : C2_V_DELETION_LENGTH) * VL;
            int carry = 0;
            for (int base = 0; base < VL; base += 64) {
                const int k = base + lane;
                const int x = (k < VL) ? d[k] : 0;
                const int s = c2_wave_incl_scan(x, lane) + carry;
                if (k < VL) d[k] = s;
                carry = __shfl(s, 63);
            }
        }
        block_barrier();
        {   // gap-free reads added only their deviations from the reference (see the column walk): every reference position
            // gets their total weight on the vector of its own base
            const int g = acc[o_sc + C2_S_RESERVED0];
            block_barrier();
            {   // ... plus, per position, the weight of the alignments with gaps whose read matches the reference there (cov, integrated above)
                const uint8_t* rs = A.refs[cur_ref].seq;
                for (int c = tid; c < VL; c += NT) {
                    const int x = g + cov[c], dx = dcov[c];
                    cov[c] = 0; dcov[c] = 0;
                    if (c < Li && x != 0) {
                        const int bv = c2_base_vector(rs[c]);
                        if (bv >= 0) acc[bv * VL + c] += x;
                    }
                    if (c < Li && dx != 0) {                                 // deletion columns: all_deletion (:4028) and the '-' base count (:4075-4081)
                        acc[C2_V_ALL_DELETION * VL + c] += dx;
                        acc[C2_V_BASE_GAP * VL + c] += dx;
                    }
                }
                if (tid == 0) acc[o_sc + C2_S_RESERVED0] = 0;
            }
        }
        block_barrier();
        long long* out = A.counts + (size_t)cur_ref * per_ref;
        for (int k = tid; k < per_ref; k += NT) {
            const int x = HBM ? atomicExch(acc + k, 0) : acc[k];
            if (x != 0) { atomicAdd((unsigned long long*)(out + k), (unsigned long long)(long long)x); if (!HBM) acc[k] = 0; }
        }
        wsum = 0;
        block_barrier();
    };

    for (;;) {
        if (tid == 0) {
            const unsigned long long b = atomicAdd(A.work_counter, (unsigned long long)CHUNK);
            ctl[0] = (int)(unsigned)(b & 0xffffffffu); ctl[1] = (int)(unsigned)(b >> 32);
        }
        __syncthreads();
        const uint64_t chunk_base = (uint64_t)(unsigned)ctl[0] | ((uint64_t)(unsigned)ctl[1] << 32);
        if (chunk_base >= n_positions) break;
        // ---- the records of this wave's K tasks, one per lane; selection test of CRISPRessoCORE.py:697 per lane
        const uint64_t my_pos = chunk_base + (uint64_t)((lane & (K - 1)) * C2_CNT_WAVES + wave);
        uint64_t my_task = my_pos;
        unsigned d0 = 0, d1 = 0, d2 = 0, d4 = 0, d5 = 0, d6 = 0; int v_w = 0;
        bool sel = false;
        if (lane < K && my_pos < n_positions) {
            if (A.order) my_task = (uint64_t)A.order[my_pos];            // tasks grouped by reference: few flushes per chunk
            else if (A.flags & C2_CNT_FLAG_ALL_REFS_LAYOUT) {            // all-references batch (task = read * n_refs + reference): the same grouping by arithmetic
                const uint64_t nr = A.n_tasks / (uint64_t)A.n_refs;
                const uint64_t r = my_pos / nr;
                my_task = (my_pos - r * nr) * (uint64_t)A.n_refs + r;
            }
            unsigned wq = A.weights ? A.weights[my_task] : 1u;
            if (A.hints) {                                               // (a hinted task is c2_count_hinted_kernel's: every main-diagonal hint, a gapped one below its weight limit)
                const unsigned h0 = A.hints[4u * my_task];
                if ((h0 & C2_HINT_VALID) || ((h0 & C2_HINT_GAPPED) && wq < (unsigned)C2_HCNT_SMALL_W)) wq = 0u;
            }
            v_w = (int)(wq > 0x7fffffffu ? 0x7fffffffu : wq);
            if (v_w > 0) {                                               // (an alignment that is not counted is not even read: most of an all-references batch)
                const unsigned* rp = (const unsigned*)(A.records + my_task);
                d0 = rp[0]; d1 = rp[1]; d2 = rp[2]; d4 = rp[4]; d5 = rp[5]; d6 = rp[6];
                if (legacy && (rp[3] >> 16) != 0u) d4 |= 0x80000000u;         // legacy: a deletion event can have NO positions (all_deletion_bases 0): mark "has a deletion column" in the top bit
                const int T = (int)(d0 & 0xffffu), matches = (int)(d0 >> 16), ref = (int)(d6 >> 16);
                sel = ((d5 >> 24) == 0) && (T > 0);
                if (sel && A.min_matches) sel = (T <= A.max_t) && (matches >= (int)A.min_matches[(size_t)ref * (A.max_t + 1) + T]);
            }
        }
        km_t pending = (km_t)__ballot(sel);
        // Everything an alignment adds to an int32 entry of the block is its weight times a count of its own columns -- at most
        // w * aln_len, its LOAD.  The loads since the last flush stay within C2_CNT_LOAD_BUDGET (2^30), so no entry can wrap;
        // saturating sums decide the flushes before anything is added.
        // (heavy chunks: v_w is what of the task's weight is still to be added; `counted`: lanes whose alignment has been counted once)
        km_t counted = 0;
        {   // (a task above ~2^26 makes the chunk heavy by itself; the others add up in 32 bits: 32 x 2^26 = 2^31.  The size test is
            // done in float -- 6.0e7 is safely below 2^26 for its rounding -- so that no 64-bit product has to be formed)
            const unsigned my_T = (d0 & 0xffffu) ? (d0 & 0xffffu) : 1u;
            const bool big = sel && (float)v_w * (float)my_T > 6.0e7f;
            unsigned wv = (sel && !big) ? (unsigned)v_w * my_T : 0u;
#pragma unroll
            for (int d = 1; d < K; d <<= 1) wv += (unsigned)__shfl_xor((int)wv, d);
            if (__ballot(big) != 0ull || wv > C2_CNT_LOAD_BUDGET) wv = C2_CNT_LOAD_BUDGET + 1u;
            if (lane == 0) ctl[2 + wave] = (int)wv;
        }
        __syncthreads();
        unsigned long long chunk_sum = 0;
#pragma unroll
        for (int v = 0; v < C2_CNT_WAVES; ++v) chunk_sum += (unsigned)ctl[2 + v];
        const unsigned chunk_w = chunk_sum > C2_CNT_LOAD_BUDGET ? C2_CNT_LOAD_BUDGET + 1u : (unsigned)chunk_sum;
        // a chunk heavier than the budget is processed one task at a time, its weight in pieces whose load fits, with a flush after each
        const bool heavy = chunk_w > C2_CNT_LOAD_BUDGET;
        if (!heavy && wsum + chunk_w > C2_CNT_LOAD_BUDGET) flush();
        wsum += heavy ? 0u : chunk_w;
        for (;;) {
            // lowest pending task of the workgroup -> the reference whose tasks are processed in this round
            const int first = pending ? __builtin_ctzll(pending) : -1;
            int fref = NONE, ftask = NONE;
            if (first >= 0) { fref = (int)((unsigned)__builtin_amdgcn_readlane((int)d6, first) >> 16); ftask = first * C2_CNT_WAVES + wave; }
            if (lane == 0) { ctl[16 + par * 2 * C2_CNT_WAVES + wave * 2] = fref; ctl[16 + par * 2 * C2_CNT_WAVES + wave * 2 + 1] = ftask; }
            __syncthreads();
            int tref = NONE, ttask = NONE;
#pragma unroll
            for (int v = 0; v < C2_CNT_WAVES; ++v) {
                const int t = ctl[16 + par * 2 * C2_CNT_WAVES + v * 2 + 1];
                if (t < ttask) { ttask = t; tref = ctl[16 + par * 2 * C2_CNT_WAVES + v * 2]; }
            }
            par ^= 1;
            if (ttask == NONE) break;
            if (tref != cur_ref) {
                flush();
                cur_ref = tref; Li = A.refs[tref].len;
                const uint16_t* g = A.refs[tref].inc_prefix;
                for (int k = tid; k < Li + 2; k += NT) incp[k] = g[k];
                __syncthreads();
            }
            km_t todo = heavy ? ((first >= 0 && ftask == ttask) ? ((km_t)1 << first) : (km_t)0) : pending;
            // ---- scalar counters and histograms of ALL tasks of this round at once: lane k holds the record of its own task, so
            //      every lane adds its task's contributions (LDS atomics; ~20 instructions per round instead of per task).
            //      aln_stats of process_fastq (CRISPRessoCORE.py:1974-1979), then the tallies of :3996-4072.
            const bool mine = lane < K && ((todo >> (lane & (K - 1))) & 1ull) && (int)(d6 >> 16) == tref;
            // the weight this round adds for the lane's task: all of it, or (heavy) a piece whose load fits the budget
            // heavy chunks: v_w becomes the piece of the weight this round adds, the remainder waits in LDS (no register of the
            // common path is spent on it: the kernel sits at 96 VGPRs = 5 waves per SIMD)
            if (heavy && mine) {
                const unsigned my_T = d0 & 0xffffu;
                const int piece = (int)(C2_CNT_LOAD_BUDGET / (my_T > 0 ? my_T : 1u));
                const int rest = v_w > piece ? v_w - piece : 0;
                ctl[C2_CNT_CTL_BASE_INTS + wave * K + (lane & (K - 1))] = rest;
                v_w -= rest;
            }
            {   // an alignment whose two strings are the reference itself (no gap column, every column a match) adds nothing but
                // its weight to the "spread over the reference's bases" scalar: done here, its strings are never read
                const int T_ = (int)(d0 & 0xffffu), matches_ = (int)(d0 >> 16);
                const bool perfect = mine && T_ == Li && matches_ == T_ && (d4 >> 16) == 0u && d1 == 0u;
                if (perfect) atomicAdd(acc + o_sc + C2_S_RESERVED0, v_w);
                const km_t pm = (km_t)__ballot(perfect);
                todo &= ~pm; pending &= ~pm;
            }
            if (mine) {
                const int w = v_w;
                const int insertion_n = (int)(d1 & 0xffffu), deletion_n = (int)(d1 >> 16), substitution_n = (int)(d2 & 0xffffu);
                const int all_ins = (int)(d2 >> 16), all_del_bases = (int)((d4 >> 16) & 0x7fffu), all_sub = (int)(d5 & 0xffffu);
                const bool irregular_ends = (d5 >> 16) & 0xffu;
                const int total_mods = all_ins + all_del_bases + all_sub;                               // :741
                const int in_win = substitution_n + deletion_n + insertion_n;                           // :742
                int* scal = acc + o_sc;
                atomicAdd(scal + C2_S_N_GLOBAL_SUBS, all_sub * w);
                atomicAdd(scal + C2_S_N_SUBS_OUTSIDE_WINDOW, (all_sub - substitution_n) * w);
                atomicAdd(scal + C2_S_N_MODS_IN_WINDOW, in_win * w);
                atomicAdd(scal + C2_S_N_MODS_OUTSIDE_WINDOW, (total_mods - in_win) * w);
                if (irregular_ends) atomicAdd(scal + C2_S_N_READS_IRREGULAR_ENDS, w);
                if (!((counted >> (lane & (K - 1))) & 1ull)) atomicAdd(scal + C2_S_ALIGNMENTS_COUNTED, 1);           // (once per alignment, not per piece of a heavy weight)
                if (discard && (deletion_n > 0 || insertion_n > 0)) atomicAdd(scal + C2_S_DISCARDED, w);                     // :3996-4000
                else {
                    const bool has_ins = !ign_ins && insertion_n > 0, has_del = !ign_del && deletion_n > 0, has_sub = !ign_sub && substitution_n > 0;
                    const bool modified = has_del || has_ins || has_sub;
                    atomicAdd(scal + C2_S_TOTAL, w);
                    atomicAdd(scal + (modified ? C2_S_MODIFIED : C2_S_UNMODIFIED), w);          // :746-760, :4003-4006
                    if (has_ins) atomicAdd(scal + C2_S_INSERTION, w);
                    if (has_del) atomicAdd(scal + C2_S_DELETION, w);
                    if (has_sub) atomicAdd(scal + C2_S_SUBSTITUTION, w);
                    int combo = -1;                                                             // :4058-4072
                    if (has_del) combo = has_ins ? (has_sub ? C2_S_INSERTION_AND_DELETION_AND_SUBSTITUTION : C2_S_INSERTION_AND_DELETION)
                                                 : (has_sub ? C2_S_DELETION_AND_SUBSTITUTION : C2_S_ONLY_DELETION);
                    else if (has_ins) combo = has_sub ? C2_S_INSERTION_AND_SUBSTITUTION : C2_S_ONLY_INSERTION;
                    else if (has_sub) combo = C2_S_ONLY_SUBSTITUTION;
                    if (combo >= 0) atomicAdd(scal + combo, w);
                    if (!ign_ins) atomicAdd(acc + o_h + C2_H_INSERTED_N * A.hl + insertion_n, w);   // :4020
                    if (!ign_del) atomicAdd(acc + o_h + C2_H_DELETED_N * A.hl + deletion_n, w);     // :4030
                    if (!ign_sub) atomicAdd(acc + o_h + C2_H_SUBSTITUTED_N * A.hl + substitution_n, w);   // :4043
                    const int eff = Li + (ign_ins ? 0 : insertion_n) - (ign_del ? 0 : deletion_n);   // :4010-4037
                    atomicAdd(acc + o_h + C2_H_EFFECTIVE_LEN * A.hl + eff, w);
                }
            }
            // ---- the alignments of this round that are walked: their strings are STAGED in LDS, C2_CNT_STAGE of them at a time, by the
            //      memory system itself (global_load_lds: no register waits for them), and only then walked one after the other -- a
            //      wavefront has that many alignments' loads in flight instead of one
            km_t walk = (km_t)__ballot(mine && ((todo >> (lane & (K - 1))) & 1ull) && !(discard && ((d1 & 0xffffu) != 0u || (d1 >> 16) != 0u)));   // (discarded reads: counted above; no vectors, :3996-4000)
            pending &= ~(km_t)__ballot(mine && ((todo >> (lane & (K - 1))) & 1ull));
            uint8_t* const stage_w = stage + (size_t)wave * (C2_CNT_STAGE * 2u * C2_CNT_STAGE_ROW);
            // columns [col0, col0 + ncols) of both strings of `task` -> slot (ncols <= C2_CNT_STAGE_ROW)
            auto stage_window = [&](uint8_t* slot, const uint64_t task, const int col0, const int ncols) {
                const uint8_t* R_ = A.aln_read + task * (uint64_t)A.aln_stride + col0;
                const uint8_t* F_ = A.aln_ref + task * (uint64_t)A.aln_stride + col0;
                if (rows_aligned) {
                    for (int b = 0; b < ncols; b += 256) {                                          // (one round unless the slot is longer than 256 columns)
                        const int p = b + 4 * lane;
                        if (p < ncols) {
                            __builtin_amdgcn_global_load_lds((const void*)(R_ + p), (void*)(slot + b), 4, 0, 0);
                            __builtin_amdgcn_global_load_lds((const void*)(F_ + p), (void*)(slot + C2_CNT_STAGE_ROW + b), 4, 0, 0);
                        }
                    }
                } else {
                    for (int b = 0; b < ncols; b += 64) {
                        const int p = b + lane;
                        if (p < ncols) {
                            __builtin_amdgcn_global_load_lds((const void*)(R_ + p), (void*)(slot + b), 1, 0, 0);
                            __builtin_amdgcn_global_load_lds((const void*)(F_ + p), (void*)(slot + C2_CNT_STAGE_ROW + b), 1, 0, 0);
                        }
                    }
                }
            };
            // ---- the gap-free alignments of the round that are no longer than 256 columns (most reads of an amplicon run): EIGHT of them at a
            //      time, eight lanes each -- a lane loads 32 columns of both strings (two 16-byte loads per string: eight lanes cover a row
            //      contiguously) and finds the differing columns as the non-zero bytes of four 64-bit XORs.  Nothing is staged, no record field
            //      travels through a scalar register, and eight alignments' loads are in flight per wavefront.  Like the one-at-a-time walk
            //      below it adds only the DEVIATIONS from the reference (+w on the read's base, -w on the reference's) and the weight to the
            //      scalar that flush() spreads over the reference's own bases.
            if (C2_CNT_GROUPED && rows_aligned16) {
                const km_t gfm = (km_t)__ballot(lane < K && ((walk >> (lane & (K - 1))) & 1ull) && (int)(d0 & 0xffffu) == Li && (d4 >> 16) == 0u &&
                                                        (d0 & 0xffffu) <= 256u);
                if (gfm) {
                    typedef unsigned long long u64;
                    walk &= ~gfm;
                    const bool gf_mine = lane < K && ((gfm >> (lane & (K - 1))) & 1ull);
                    if (gf_mine) atomicAdd(acc + o_sc + C2_S_RESERVED0, v_w);
                    // the q-th of them sits in lane tab[q] (the wavefront's staging area is free here)
                    C2_LDS_READS_DONE();
                    if (gf_mine) stage_w[__popcll(gfm & (((km_t)1 << (lane & (K - 1))) - 1ull))] = (uint8_t)lane;
                    C2_LDS_READS_DONE();
                    const int ng = __popcll(gfm), grp = lane >> 3, p0 = 32 * (lane & 7);
                    const u64 H = 0x8080808080808080ull, L7 = 0x7f7f7f7f7f7f7f7full;
                    for (int q0 = 0; q0 < ng; q0 += 8) {
                        const bool on = q0 + grp < ng;
                        const int src = on ? (int)stage_w[q0 + grp] : 0;
                        const uint64_t task = (uint64_t)(unsigned)__shfl((int)(unsigned)(my_task & 0xffffffffull), src) |
                                              ((uint64_t)(unsigned)__shfl((int)(unsigned)(my_task >> 32), src) << 32);
                        const int w = __shfl(v_w, src);
                        const int nbytes = on ? Li - p0 : 0;                                        // columns of this lane inside the alignment
                        u64 RD[4] = {0ull, 0ull, 0ull, 0ull}, RF[4] = {0ull, 0ull, 0ull, 0ull};
                        if (nbytes > 0) {
                            const uint8_t* R_ = A.aln_read + task * (uint64_t)A.aln_stride + p0;
                            const uint8_t* F_ = A.aln_ref + task * (uint64_t)A.aln_stride + p0;
                            const uint4 a = *(const uint4*)R_, b = *(const uint4*)F_;
                            RD[0] = (u64)a.x | ((u64)a.y << 32); RD[1] = (u64)a.z | ((u64)a.w << 32);
                            RF[0] = (u64)b.x | ((u64)b.y << 32); RF[1] = (u64)b.z | ((u64)b.w << 32);
                            if (nbytes > 16) {
                                const uint4 c = *(const uint4*)(R_ + 16), d = *(const uint4*)(F_ + 16);
                                RD[2] = (u64)c.x | ((u64)c.y << 32); RD[3] = (u64)c.z | ((u64)c.w << 32);
                                RF[2] = (u64)d.x | ((u64)d.y << 32); RF[3] = (u64)d.z | ((u64)d.w << 32);
                            }
                        }
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) {
                            const int nb = nbytes - 8 * k8;
                            const u64 vm = nb >= 8 ? H : (nb > 0 ? (H & ((1ull << (8 * nb)) - 1ull)) : 0ull);
                            const u64 x = RD[k8] ^ RF[k8];
                            u64 mm = (((x & L7) + L7) | x) & vm;
                            while (mm) {
                                const int bb = __builtin_ctzll(mm) >> 3;
                                mm &= mm - 1ull;
                                const int c = p0 + 8 * k8 + bb;
                                const unsigned char rd = (unsigned char)(RD[k8] >> (8 * bb)), rfc = (unsigned char)(RF[k8] >> (8 * bb));
                                const int bvr = c2_base_vector(rd), bvf = c2_base_vector(rfc);
                                if (bvr >= 0) atomicAdd(acc + bvr * VL + c, w);
                                if (bvf >= 0) atomicAdd(acc + bvf * VL + c, -w);
                                if (rd != 'N') {
                                    atomicAdd(acc + C2_V_ALL_SUBSTITUTION * VL + c, w);             // :4040
                                    if (!ign_sub) {
                                        if (incp[c + 1] != incp[c]) atomicAdd(acc + C2_V_SUBSTITUTION * VL + c, w);   // :4044
                                        const int sv = c2_sub_base_vector(rd);                      // :4049-4054
                                        if (sv >= 0) atomicAdd(acc + sv * VL + c, w);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            while (walk) {
                km_t batch = 0;
                C2_LDS_READS_DONE();                                                                // (the slots are free: every lane has read what it needed of them)
                for (int sl = 0; sl < C2_CNT_STAGE && walk; ++sl) {
                    const int kk = __builtin_ctzll(walk);
                    walk &= walk - 1; batch |= (km_t)1 << kk;
                    const uint64_t task = (uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_task & 0xffffffffull), kk) |
                                          ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_task >> 32), kk) << 32);
                    const int T = (int)((unsigned)__builtin_amdgcn_readlane((int)d0, kk) & 0xffffu);
                    stage_window(stage_w + sl * (2 * C2_CNT_STAGE_ROW), task, 0, T < C2_CNT_STAGE_ROW ? T : C2_CNT_STAGE_ROW);
                }
                C2_WAIT_LDS_DMA();
                for (int sl = 0; batch; ++sl) {
                const int kk = __builtin_ctzll(batch);
                batch &= batch - 1;
                uint8_t* const slot = stage_w + sl * (2 * C2_CNT_STAGE_ROW);
                const uint8_t* SR = slot;                                                           // the staged window of the aligned read ...
                const uint8_t* SF = slot + C2_CNT_STAGE_ROW;                                        // ... and of the aligned reference
                const uint64_t task = (uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_task & 0xffffffffull), kk) |
                                      ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_task >> 32), kk) << 32);
                const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)d0, kk), r1 = (unsigned)__builtin_amdgcn_readlane((int)d1, kk);
                const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)d2, kk), r4 = (unsigned)__builtin_amdgcn_readlane((int)d4, kk);
                const int w = __builtin_amdgcn_readlane(v_w, kk);
                const int T = (int)(r0 & 0xffffu);
                const int insertion_n = (int)(r1 & 0xffffu), deletion_n = (int)(r1 >> 16), substitution_n = (int)(r2 & 0xffffu);
                const bool any_del_column = (r4 >> 16) != 0u;                                      // (positions, or the legacy marker)
                // the next window of a long alignment (T > C2_CNT_STAGE_ROW) into this slot
                auto next_window = [&](const int win0) {
                    C2_LDS_READS_DONE();
                    stage_window(slot, task, win0, T - win0 < C2_CNT_STAGE_ROW ? T - win0 : C2_CNT_STAGE_ROW);
                    C2_WAIT_LDS_DMA();
                };
                if (T == Li && !any_del_column) {
                    // No gap column in either string (the usual read): the reference index of a column is the column, only substitutions can
                    // occur, and the read's base counts differ from "the reference's own base, once per read" only where the read differs
                    // from the reference.  So only the DEVIATIONS are added (+w on the read's base, -w on the reference's) and the read's
                    // weight goes to one scalar that flush() spreads over the reference's bases.  Four columns per lane: read and reference
                    // as dwords, the bytes in which they differ by the "has a zero byte" trick on their XOR.
                    for (int win0 = 0; win0 < T; win0 += C2_CNT_STAGE_ROW) {
                        if (win0) next_window(win0);
                        const int wl = T - win0 < C2_CNT_STAGE_ROW ? T - win0 : C2_CNT_STAGE_ROW;
                        for (int b = 0; b < wl; b += 256) {
                            const int p = b + 4 * lane, nb = wl - p;
                            unsigned rdw = 0, rfw = 0;
                            if (nb > 0) { rdw = *(const unsigned*)(SR + p); rfw = *(const unsigned*)(SF + p); }
                            const unsigned valid = nb >= 4 ? 0xffffffffu : (nb > 0 ? ((1u << (8 * nb)) - 1u) : 0u);
                            const unsigned x = (rdw ^ rfw) & valid;
                            unsigned mm = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
                            while (mm) {
                                const int bb = __builtin_ctz(mm) >> 3;
                                mm &= mm - 1u;
                                const int c = win0 + p + bb;
                                const unsigned char rd = (unsigned char)(rdw >> (8 * bb)), rfc = (unsigned char)(rfw >> (8 * bb));
                                const int bvr = c2_base_vector(rd), bvf = c2_base_vector(rfc);
                                if (bvr >= 0) atomicAdd(acc + bvr * VL + c, w);
                                if (bvf >= 0) atomicAdd(acc + bvf * VL + c, -w);
                                if (rd != 'N') {
                                    atomicAdd(acc + C2_V_ALL_SUBSTITUTION * VL + c, w);             // :4040
                                    if (!ign_sub) {
                                        if (incp[c + 1] != incp[c]) atomicAdd(acc + C2_V_SUBSTITUTION * VL + c, w);   // :4044
                                        const int sv = c2_sub_base_vector(rd);                      // :4049-4054
                                        if (sv >= 0) atomicAdd(acc + sv * VL + c, w);
                                    }
                                }
                            }
                        }
                    }
                    if (lane == 0) atomicAdd(acc + o_sc + C2_S_RESERVED0, w);
                    continue;
                }
                const bool has_ins = !ign_ins && insertion_n > 0, has_del = !ign_del && deletion_n > 0, has_sub = !ign_sub && substitution_n > 0;
                const bool modified = has_del || has_ins || has_sub;
                const bool len_block = modified;                                                // :4085 (no coding sequence)
                if (C2_CNT_SWAR && T <= C2_CNT_STAGE_ROW) {
                    // ---- An alignment with gaps that lies in its slot as a whole (the usual case): EIGHT columns per lane, classified as the bytes of
                    //      64-bit words (gap / same / different masks by the zero-byte trick), one pass instead of one 64-column chunk after the other.
                    //      Columns where nothing happens cost nothing beyond that; runs of matching columns touch `cov` at their two ends, a
                    //      mismatch adds its base, and everything a gap run adds is added by the lane that holds the column BEHIND the run -- it
                    //      finds the run's length by walking back over the staged string, so no state travels between lanes except the
                    //      reference-index prefix and one flag byte of the neighbours.
                    typedef unsigned long long u64;
                    const u64 H = 0x8080808080808080ull, L7 = 0x7f7f7f7f7f7f7f7full, DASH = 0x2d2d2d2d2d2d2d2dull;
                    const int p = 8 * lane, nb = T - p;                                             // this lane: columns p .. p + 7, nb of them inside the alignment
                    u64 RD = 0, RF = 0;
                    if (nb > 0) { RD = *(const u64*)(SR + p); RF = *(const u64*)(SF + p); }
                    const u64 vm = nb >= 8 ? H : (nb > 0 ? (H & ((1ull << (8 * nb)) - 1ull)) : 0ull);
                    auto nz = [&](const u64 x) { return (((x & L7) + L7) | x) & H; };             // bit 7 of every non-zero byte
                    const u64 g_rd = ~nz(RD ^ DASH) & vm, g_rf = ~nz(RF ^ DASH) & vm;              // gap columns of the read / of the reference
                    const u64 ng_rf = vm & ~g_rf;                                                   // columns that have a reference base
                    const u64 same = ng_rf & ~nz(RD ^ RF);                                          // ... and the read's base is that base
                    // reference index of the lane's first column: 8 * lane minus the reference's gap columns in the lanes below
                    int idx_lane = p;
                    {
                        const int gcnt = __popcll(g_rf);
                        if (__ballot(gcnt != 0) != 0ull) {
                            int below = 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) below += __popcll(__ballot((gcnt >> q) & 1) & lt) << q;
                            idx_lane -= below;
                        }
                    }
                    auto idx_of = [&](const int b) { return idx_lane + __popcll(ng_rf & ((1ull << (8 * b)) - 1ull)); };
                    // the neighbours: the lane below's last column (same / read gap / reference gap), the lane above's first (same)
                    int prev = __shfl_up((int)((same >> 63) | ((g_rd >> 63) << 1) | ((g_rf >> 63) << 2)), 1);
                    if (lane == 0) prev = 0;
                    int next_same = __shfl_down((int)((same >> 7) & 1ull), 1);
                    if (lane == 63) next_same = 0;
                    {   // all_base_count of the matching columns (:4075-4081): +w where a run of them starts, -w behind its end (see flush)
                        u64 starts = same & ~((same << 8) | ((prev & 1) ? 0x80ull : 0ull));
                        u64 ends = same & ~((same >> 8) | (next_same ? (0x80ull << 56) : 0ull));
                        while (starts) { const int b = __builtin_ctzll(starts) >> 3; starts &= starts - 1ull; atomicAdd(cov + idx_of(b), w); }
                        while (ends) { const int b = __builtin_ctzll(ends) >> 3; ends &= ends - 1ull; atomicAdd(cov + idx_of(b) + 1, -w); }
                    }
                    {   // columns where read and reference both have a base and differ
                        u64 mm = ng_rf & ~g_rd & ~same;
                        while (mm) {
                            const int b = __builtin_ctzll(mm) >> 3;
                            mm &= mm - 1ull;
                            const int ix = idx_of(b);
                            const unsigned char rd = (unsigned char)(RD >> (8 * b));
                            const int bv = c2_base_vector(rd);
                            if (bv >= 0) atomicAdd(acc + bv * VL + ix, w);
                            if (rd != 'N') {
                                atomicAdd(acc + C2_V_ALL_SUBSTITUTION * VL + ix, w);                // :4040
                                if (!ign_sub) {
                                    if (incp[ix + 1] != incp[ix]) atomicAdd(acc + C2_V_SUBSTITUTION * VL + ix, w);   // :4044
                                    const int sv = c2_sub_base_vector(rd);                          // :4049-4054
                                    if (sv >= 0) atomicAdd(acc + sv * VL + ix, w);
                                }
                            }
                        }
                    }
                    // the column behind a gap run: a reference base behind an insertion, a read base behind a deletion
                    u64 ic = ng_rf & ((g_rf << 8) | ((prev & 4) ? 0x80ull : 0ull));
                    u64 dc = (vm & ~g_rd) & ((g_rd << 8) | ((prev & 2) ? 0x80ull : 0ull));
                    const bool tail = nb >= 1 && nb <= 8 && ((g_rd >> (8 * (nb - 1) + 7)) & 1ull);  // the alignment ends in a deletion: this lane holds its last column
                    if (__ballot((ic | dc) != 0ull || tail) != 0ull) {
                        while (ic) {
                            // insertions: positions [idx-1, idx] of every event; numpy's fancy += counts a repeated position once (:4016, :4021)
                            const int b = __builtin_ctzll(ic) >> 3;
                            ic &= ic - 1ull;
                            const int ix = idx_of(b), c = p + b;
                            if (ix <= 0) continue;                                                  // (an insertion in front of the first reference base is none)
                            int k = c - 2;                                                          // (column c - 1 is a gap of the reference)
                            while (k >= 0 && SF[k] == '-') --k;                                     // the reference base in front of the insertion
                            // ... was it the end of an insertion itself?  Then position idx - 1 has been counted by that event
                            const bool prev_close = k >= 1 && SF[k - 1] == '-' && ix - 1 > 0;
                            const bool fl = incp[ix] != incp[ix - 1], fr = incp[ix + 1] != incp[ix];
                            const bool ins_win = legacy ? (fl || fr) : (fl && fr);                  // pyx:121 / legacy pyx:284
                            bool prev_wclose = false;
                            if (prev_close) {
                                const bool pl = incp[ix - 1] != incp[ix - 2];                       // (fl of that event; its fr is this one's fl)
                                prev_wclose = legacy ? (pl || fl) : (pl && fl);
                            }
                            atomicAdd(acc + C2_V_ALL_INSERTION_LEFT * VL + ix - 1, w);              // :4017
                            atomicAdd(acc + C2_V_ALL_INSERTION * VL + ix, w);
                            if (!prev_close) atomicAdd(acc + C2_V_ALL_INSERTION * VL + ix - 1, w);
                            if (ins_win) {
                                if (!ign_ins) {
                                    atomicAdd(acc + C2_V_INSERTION * VL + ix, w);
                                    if (!prev_wclose) atomicAdd(acc + C2_V_INSERTION * VL + ix - 1, w);
                                }
                                if (len_block) {                                                    // :4104-4106 (scalar index: repeats add twice)
                                    const int sz = (c - 1 - k) * w;
                                    atomicAdd(acc + C2_V_INSERTION_LENGTH * VL + ix - 1, sz);
                                    atomicAdd(acc + C2_V_INSERTION_LENGTH * VL + ix, sz);
                                }
                            }
                        }
                        while (dc) {
                            const int b = __builtin_ctzll(dc) >> 3;
                            dc &= dc - 1ull;
                            const int ix = idx_of(b), c = p + b;
                            int k = c - 2;                                                          // (column c - 1 is a gap of the read)
                            while (k >= 0 && SR[k] == '-') --k;                                     // the read base in front of the deletion
                            const int dlen = c - 1 - k;
                            atomicAdd(dcov + ix - dlen, w);                                         // its columns: all_deletion (:4028) and '-' base counts, as a range
                            atomicAdd(dcov + ix, -w);
                            // legacy (pyx:253-258): a run that starts in column 0 or 1 gets reference start 0 -- position 0 joins its
                            // positions although the read has a base there
                            const int dstart = (legacy && k <= 0) ? 0 : ix - dlen;
                            if (legacy && k == 0) atomicAdd(acc + C2_V_ALL_DELETION * VL + 0, w);
                            if (incp[ix] != incp[dstart]) {                                         // deletions that touch the window: range(start, end) as a difference array
                                if (!ign_del) { atomicAdd(acc + C2_V_DELETION * VL + dstart, w); atomicAdd(acc + C2_V_DELETION * VL + ix, -w); }   // :4031
                                if (len_block) { atomicAdd(acc + C2_V_DELETION_LENGTH * VL + dstart, dlen * w); atomicAdd(acc + C2_V_DELETION_LENGTH * VL + ix, -dlen * w); }   // :4114
                            }
                        }
                        if (tail) {                                                                 // trailing deletion, pyx:155-162
                            int k = T - 2;
                            while (k >= 0 && SR[k] == '-') --k;
                            const int dlen = T - 1 - k, idx_end = idx_lane + __popcll(ng_rf);       // (idx_end: the reference bases of the whole alignment)
                            atomicAdd(dcov + idx_end - dlen, w);
                            atomicAdd(dcov + idx_end, -w);
                            // legacy (pyx:259-261): the run ends at reference index idx - 1, exclusive -- the last base is not among its positions
                            const int dstart = (legacy && k <= 0) ? 0 : idx_end - dlen, dend = legacy ? idx_end - 1 : idx_end;
                            if (legacy) {
                                atomicAdd(acc + C2_V_ALL_DELETION * VL + idx_end - 1, -w);
                                if (k == 0) atomicAdd(acc + C2_V_ALL_DELETION * VL + 0, w);
                            }
                            if (dend > dstart && incp[dend] != incp[dstart]) {
                                if (!ign_del) { atomicAdd(acc + C2_V_DELETION * VL + dstart, w); atomicAdd(acc + C2_V_DELETION * VL + dend, -w); }
                                if (len_block) { atomicAdd(acc + C2_V_DELETION_LENGTH * VL + dstart, dlen * w); atomicAdd(acc + C2_V_DELETION_LENGTH * VL + dend, -dlen * w); }
                            }
                        }
                    }
                    continue;
                }
                // ---- column walk (same scan as the fused classifier), ds_add into the vectors
                int idx_base = 0, last_rf = -1, last_rd = -1;
                bool last_rf_close = false, last_rf_wclose = false;
                for (int base = 0; base < T; base += 64) {
                    if (base && (base % C2_CNT_STAGE_ROW) == 0) next_window(base);
                    const int c = base + lane;
                    const bool in = c < T;
                    const unsigned char rd = in ? SR[c % C2_CNT_STAGE_ROW] : (unsigned char)0, rfc = in ? SF[c % C2_CNT_STAGE_ROW] : (unsigned char)0;
                    const bool rf_ng = in && rfc != '-', rd_ng = in && rd != '-';
                    const unsigned long long m_rf = __ballot(rf_ng), m_rd = __ballot(rd_ng);
                    const int idx = idx_base + __popcll(m_rf & lt);
                    const unsigned long long below_rf = m_rf & lt, below_rd = m_rd & lt;
                    const int prev_rf = below_rf ? base + 63 - __clzll((long long)below_rf) : last_rf;
                    const int prev_rd = below_rd ? base + 63 - __clzll((long long)below_rd) : last_rd;
                    // all_base_count, :4075-4081.  Columns where the read's base IS the reference's (nearly all of them) are not added one
                    // by one: a run of them adds its weight to the difference array `cov` at its two ends (see flush)
                    const bool same = rf_ng && rd == rfc;
                    {
                        const unsigned long long m_same = __ballot(same);
                        if (same) {
                            if (lane == 0 || !((m_same >> (lane - 1)) & 1ull)) atomicAdd(cov + idx, w);
                            if (lane == 63 || !((m_same >> (lane + 1)) & 1ull)) atomicAdd(cov + idx + 1, -w);
                        }
                    }
                    if (rf_ng && !same) {
                        const int bv = c2_base_vector(rd);
                        if (bv >= 0) atomicAdd(acc + bv * VL + idx, w);
                        if (!rd_ng) atomicAdd(acc + C2_V_ALL_DELETION * VL + idx, w);                   // :4028
                    }
                    const bool sub = rf_ng && rd_ng && rd != rfc && rd != 'N';
                    if (sub) {
                        atomicAdd(acc + C2_V_ALL_SUBSTITUTION * VL + idx, w);                           // :4040
                        if (!ign_sub) {
                            if (incp[idx + 1] != incp[idx]) atomicAdd(acc + C2_V_SUBSTITUTION * VL + idx, w);   // :4044
                            const int sv = c2_sub_base_vector(rd);                                      // :4049-4054
                            if (sv >= 0) atomicAdd(acc + sv * VL + idx, w);
                        }
                    }
                    {   // 64 columns without a gap and no gap run open in front of them (most chunks of an alignment with gaps): no insertion
                        // or deletion can close here -- the rest of the body would find nothing
                        const unsigned long long m_in = __ballot(in);
                        if (m_rf == m_in && m_rd == m_in && last_rf == base - 1 && last_rd == base - 1) {
                            const int cols = __popcll(m_in);
                            idx_base += cols; last_rf = base + cols - 1; last_rd = last_rf;
                            last_rf_close = false; last_rf_wclose = false;
                            continue;
                        }
                    }
                    // insertions: positions [idx-1, idx] of every event; numpy's fancy += counts a repeated position once (:4016, :4021)
                    const bool ins_close = rf_ng && (prev_rf != c - 1) && idx > 0;
                    const bool fl = ins_close && (incp[idx] != incp[idx - 1]), fr = ins_close && (incp[idx + 1] != incp[idx]);
                    const bool ins_win = legacy ? (fl || fr) : (fl && fr);                              // pyx:121 / legacy pyx:284
                    const unsigned long long m_ic = __ballot(ins_close), m_iw = __ballot(ins_win);
                    if (ins_close) {
                        const bool prev_close = (prev_rf >= base) ? ((m_ic >> (prev_rf - base)) & 1ull) : last_rf_close;
                        const bool prev_wclose = (prev_rf >= base) ? ((m_iw >> (prev_rf - base)) & 1ull) : last_rf_wclose;
                        atomicAdd(acc + C2_V_ALL_INSERTION_LEFT * VL + idx - 1, w);                     // :4017
                        atomicAdd(acc + C2_V_ALL_INSERTION * VL + idx, w);
                        if (!prev_close) atomicAdd(acc + C2_V_ALL_INSERTION * VL + idx - 1, w);
                        if (ins_win) {
                            if (!ign_ins) {
                                atomicAdd(acc + C2_V_INSERTION * VL + idx, w);
                                if (!prev_wclose) atomicAdd(acc + C2_V_INSERTION * VL + idx - 1, w);
                            }
                            if (len_block) {                                                            // :4104-4106 (scalar index: repeats add twice)
                                const int sz = (c - 1 - prev_rf) * w;
                                atomicAdd(acc + C2_V_INSERTION_LENGTH * VL + idx - 1, sz);
                                atomicAdd(acc + C2_V_INSERTION_LENGTH * VL + idx, sz);
                            }
                        }
                    }
                    // deletions that touch the window: range(start, end) as a difference array (integrated in flush)
                    const bool del_close = rd_ng && (prev_rd != c - 1);
                    if (del_close) {
                        const int dlen = c - 1 - prev_rd;
                        // legacy (pyx:253-258): a run that starts in column 0 or 1 gets reference start 0 -- position 0 joins its
                        // positions although the read has a base there
                        const int dstart = (legacy && prev_rd <= 0) ? 0 : idx - dlen;
                        if (legacy && prev_rd == 0) atomicAdd(acc + C2_V_ALL_DELETION * VL + 0, w);
                        if (incp[idx] != incp[dstart]) {
                            if (!ign_del) { atomicAdd(acc + C2_V_DELETION * VL + dstart, w); atomicAdd(acc + C2_V_DELETION * VL + idx, -w); }   // :4031
                            if (len_block) { atomicAdd(acc + C2_V_DELETION_LENGTH * VL + dstart, dlen * w); atomicAdd(acc + C2_V_DELETION_LENGTH * VL + idx, -dlen * w); }   // :4114
                        }
                    }
                    idx_base += __popcll(m_rf);
                    if (m_rf) {
                        const int hi = 63 - __clzll((long long)m_rf);
                        last_rf = base + hi;
                        last_rf_close = (m_ic >> hi) & 1ull;
                        last_rf_wclose = (m_iw >> hi) & 1ull;
                    }
                    if (m_rd) last_rd = base + 63 - __clzll((long long)m_rd);
                }
                if (last_rd != T - 1 && lane == 0) {                                                    // trailing deletion, pyx:155-162
                    const int dlen = T - 1 - last_rd;
                    // legacy (pyx:259-261): the run ends at reference index idx - 1, exclusive -- the last base is not among its positions
                    const int dstart = (legacy && last_rd <= 0) ? 0 : idx_base - dlen, dend = legacy ? idx_base - 1 : idx_base;
                    if (legacy) {
                        atomicAdd(acc + C2_V_ALL_DELETION * VL + idx_base - 1, -w);
                        if (last_rd == 0) atomicAdd(acc + C2_V_ALL_DELETION * VL + 0, w);
                    }
                    if (dend > dstart && incp[dend] != incp[dstart]) {
                        if (!ign_del) { atomicAdd(acc + C2_V_DELETION * VL + dstart, w); atomicAdd(acc + C2_V_DELETION * VL + dend, -w); }
                        if (len_block) { atomicAdd(acc + C2_V_DELETION_LENGTH * VL + dstart, dlen * w); atomicAdd(acc + C2_V_DELETION_LENGTH * VL + dend, -dlen * w); }
                    }
                }
                }   // staged alignments of this batch
            }       // batches of this round
            if (heavy) {
                // a task whose weight was added only in part stays pending for another round
                counted |= (km_t)__ballot(mine);
                if (mine) v_w = ctl[C2_CNT_CTL_BASE_INTS + wave * K + (lane & (K - 1))];
                pending |= (km_t)__ballot(mine && v_w > 0);
                flush();
            }
        }       // rounds of this chunk
    }           // chunks
    flush();
}

__global__ __launch_bounds__(64 * C2_CNT_WAVES, C2_CNT_OCC) void c2_count_vectors_kernel(c2_count_args A) { c2_count_vectors_body<false>(A); }
__global__ __launch_bounds__(64 * C2_CNT_WAVES, C2_CNT_OCC) void c2_count_vectors_hbm_kernel(c2_count_args A) { c2_count_vectors_body<true>(A); }


// ---------------------------------------------------------------------------------------------------------------
// c2_count_hinted_kernel (round 6): the tasks whose alignment left a HINT (c2_batch.diag_hints, four words per task) counted from the hint, a lane per
// task -- the 2 x 250 bytes of aligned strings are not read back.  One reference (the host checks).
//   C2_HINT_VALID   (c2_align_partition_kernel: the read on its reference's main diagonal, at most two differing bases -- 58 % of the headline batch):
//                   from the word alone, not even the record is read.  Its weight w goes to the scalars and histogram bins of its class -- kept in
//                   64-bit registers per lane, reduced once per wavefront at the end --, w on the base vector of the reference's own base at every
//                   position (summed, spread at the end), and per differing base the deviations.
//   C2_HINT_GAPPED  (c2_group_epilogue: at most five runs, at most three differing columns, weight below C2_HCNT_SMALL_W): the record (32 bytes) for the
//                   scalar counters -- the statements of c2_count_vectors_body's scalar stage --, and from the runs what its column walk adds:
//                   runs of M as ranges of "the read's base is the reference's" (difference array `cov`), deletions as ranges (`dcov`, C2_V_DELETION,
//                   C2_V_DELETION_LENGTH), insertions at their two flanks, the differing columns one by one (CRISPRessoCORE.py:4010-4115).
// The position vectors accumulate in an int32 LDS block, flushed before anything can wrap (every C2_HCNT_FLUSH_ROUNDS rounds of 256 tasks: a task adds
// at most 512 * w to an entry, w < C2_HCNT_SMALL_W); a main-diagonal task of a larger weight goes to the tensor directly, a gapped one of a larger
// weight is left to c2_count_vectors_kernel (which skips exactly the tasks this kernel takes: c2_count_task_is_hinted).
// ---------------------------------------------------------------------------------------------------------------
#define C2_HCNT_FLUSH_ROUNDS 16                // 16 x 256 tasks x 512 x 1,023 < 2^31
__global__ __launch_bounds__(256) void c2_count_hinted_kernel(c2_count_args A)
{
    typedef unsigned long long u64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int VL = A.lmax + 1, NV = C2_CNT_VECTORS * VL, NH = C2_CNT_HISTS * A.hl;
    int* acc = (int*)c2_smem;                                       // [NV] position vectors, [NH] histograms, [2 * VL] cov / dcov
    int* hist = acc + NV;
    int* cov = hist + NH;
    int* dcov = cov + VL;
    u64* tot = (u64*)(c2_smem + (((size_t)(NV + NH + 2 * VL)) * sizeof(int) + 15) / 16 * 16);   // [16] the main-diagonal tasks' totals, [C2_CNT_SCALARS] the gapped tasks' scalars
    u64* scal = tot + 16;
    uint16_t* incp = (uint16_t*)(scal + C2_CNT_SCALARS);
    int* part = (int*)(incp + ((A.lmax + 2 + 7) / 8) * 8);          // [4] carries of the scans, [4] entries of lrest, [5] their place in the global list
    uint32_t* lrest = (uint32_t*)(part + 8);                        // [C2_HCNT_FLUSH_ROUNDS * 256] the tasks left to the column walk since the last flush
    uint8_t* sseq = (uint8_t*)(lrest + C2_HCNT_FLUSH_ROUNDS * 256);  // [lmax + 1] the reference's bases (a differing base's reference base: no load from HBM behind the hint)
    // one reference per workgroup: the batch's only one, or (tasks grouped by reference in A.order) reference blockIdx.x / hint_gx and its range of positions
    int ref = 0;
    unsigned bx = blockIdx.x, gx = gridDim.x;
    uint64_t p_lo = 0, p_hi = A.n_tasks;
    // (an all-references batch, task = read * n_refs + reference: position p of reference r's range is read p - p_lo -- the column walk's own arithmetic)
    const bool by_layout = !A.ref_ends && (A.flags & C2_CNT_FLAG_ALL_REFS_LAYOUT) && A.n_refs > 1;
    const bool per_ref = A.ref_ends || by_layout;
    if (A.ref_ends) { gx = A.hint_gx; ref = (int)(blockIdx.x / gx); bx = blockIdx.x - (unsigned)ref * gx; p_lo = ref ? A.ref_ends[ref - 1] : 0u; p_hi = A.ref_ends[ref]; }
    else if (by_layout) { gx = A.hint_gx; ref = (int)(blockIdx.x / gx); bx = blockIdx.x - (unsigned)ref * gx; const uint64_t nr = A.n_tasks / (uint64_t)A.n_refs; p_lo = (uint64_t)ref * nr; p_hi = p_lo + nr; }
    if (p_lo >= p_hi) return;
    const c2_dev_ref rf = A.refs[ref];
    const int Li = rf.len;
    const int o_sc = NV, o_h = o_sc + C2_CNT_SCALARS;
    for (int k = tid; k < NV + NH + 2 * VL; k += 256) acc[k] = 0;
    if (tid < 16 + C2_CNT_SCALARS) tot[tid] = 0ull;
    if (tid < 8) part[tid] = 0;
    for (int k = tid; k < Li + 2; k += 256) incp[k] = rf.inc_prefix[k];
    for (int k = tid; k < Li && k <= A.lmax; k += 256) sseq[k] = rf.seq[k];
    __syncthreads();
    const bool ign_sub = A.flags & C2_CNT_FLAG_IGNORE_SUBSTITUTIONS, ign_ins = A.flags & C2_CNT_FLAG_IGNORE_INSERTIONS, ign_del = A.flags & C2_CNT_FLAG_IGNORE_DELETIONS;
    const bool discard = A.flags & C2_CNT_FLAG_DISCARD_INDEL_READS;
    // the selection test of CRISPRessoCORE.py:697 for a main-diagonal alignment: Li columns, Li - k matches
    bool gate = Li > 0;
    int thresh = 0;
    const uint16_t* mm_row = A.min_matches ? A.min_matches + (size_t)ref * (size_t)(A.max_t + 1) : nullptr;
    if (mm_row) { if (Li > A.max_t) gate = false; else thresh = (int)mm_row[Li]; }
    long long* out = A.counts + (size_t)ref * (size_t)(o_h + NH);
    // the LDS block -> the tensor: the difference arrays integrated first (as c2_count_vectors_body's flush does)
    auto flush = [&]() {
        __syncthreads();
        {
            int* d = wave == 3 ? dcov : wave == 2 ? cov : acc + (wave == 0 ? C2_V_DELETION : C2_V_DELETION_LENGTH) * VL;
            int carry = 0;
            for (int base = 0; base < VL; base += 64) {
                const int k = base + lane;
                const int x = (k < VL) ? d[k] : 0;
                const int sc_ = c2_wave_incl_scan(x, lane) + carry;
                if (k < VL) d[k] = sc_;
                carry = __shfl(sc_, 63);
            }
        }
        __syncthreads();
        for (int c = tid; c < VL; c += 256) {
            const int x = cov[c], dx = dcov[c];
            cov[c] = 0; dcov[c] = 0;
            if (c < Li && x != 0) { const int bv = c2_base_vector(rf.seq[c]); if (bv >= 0) acc[bv * VL + c] += x; }
            if (c < Li && dx != 0) { acc[C2_V_ALL_DELETION * VL + c] += dx; acc[C2_V_BASE_GAP * VL + c] += dx; }     // :4028, :4075-4081
        }
        __syncthreads();
        for (int k = tid; k < NV; k += 256) { const int x = acc[k]; if (x != 0) { atomicAdd((u64*)(out + k), (u64)(long long)x); acc[k] = 0; } }
        for (int k = tid; k < NH; k += 256) { const int x = hist[k]; if (x != 0) { atomicAdd((u64*)(out + o_h + k), (u64)(long long)x); hist[k] = 0; } }
        __syncthreads();
    };
    // the LDS list of the tasks left to the column walk -> the global list (one atomic per call)
    auto flush_rest = [&]() {
        __syncthreads();
        const int nrest = part[4];
        // (several references: a counter per reference, the reference's tasks into its own range of the list -- c2_rest_compact_kernel closes the gaps)
        if (tid == 0 && nrest > 0) part[5] = (int)atomicAdd(A.rest_count + (per_ref ? ref : 0), (unsigned)nrest);
        __syncthreads();
        const uint64_t b0 = (per_ref ? p_lo : 0ull) + (uint64_t)(unsigned)part[5];
        for (int k = tid; k < nrest; k += 256) A.rest_list[b0 + (uint64_t)k] = lrest[k];
        __syncthreads();
        if (tid == 0) part[4] = 0;
        __syncthreads();
    };
    // without weights a task adds at most 512 to an entry: the block is emptied once, at the end (or after 2^13 rounds) -- every flush is some thousand
    // 64-bit atomics on the same addresses from every workgroup of the launch
    const unsigned acc_rounds = A.weights ? (unsigned)C2_HCNT_FLUSH_ROUNDS : 8192u;
    u64 sW = 0, sN = 0, sSubW = 0, sGsub = 0, sOut = 0, sIn = 0, sIrr = 0, sH0 = 0, sH1 = 0, sH2 = 0;
    // The gapped tasks' scalars, the histograms' commonest bins (no insertion / no deletion / no substitution in the window / the reference's own length) and
    // the two ends of `cov` are what EVERY gapped task of a wavefront adds to: as LDS atomics they are the same address from every lane -- serialised, and the
    // LDS pipe was busy for half of the kernel's time.  They are kept per lane (int32: drained every C2_HCNT_FLUSH_ROUNDS rounds, 16 x 1,023 x 512 < 2^31)
    // and reach LDS as one atomic per wavefront.  g_pat[has_del * 4 + has_ins * 2 + has_sub]: what the class counters of :746-760 / :4058-4072 are sums of.
    int g_gsub = 0, g_out = 0, g_in = 0, g_mout = 0, g_irr = 0, g_n = 0, g_disc = 0;
    int g_pat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int g_ins0 = 0, g_del0 = 0, g_sub0 = 0, g_eff0 = 0, g_cov0 = 0, g_covL = 0;
    auto wsum = [&](int x) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) x += __shfl_xor(x, d);
        return x;
    };
    auto drain = [&]() {
        auto sput = [&](const int k_, int& x) { const int t_ = wsum(x); x = 0; if (lane == 0 && t_) atomicAdd(scal + k_, (u64)(long long)t_); };
        auto iput = [&](int* dst, int& x) { const int t_ = wsum(x); x = 0; if (lane == 0 && t_) atomicAdd(dst, t_); };
        sput(C2_S_N_GLOBAL_SUBS, g_gsub); sput(C2_S_N_SUBS_OUTSIDE_WINDOW, g_out); sput(C2_S_N_MODS_IN_WINDOW, g_in); sput(C2_S_N_MODS_OUTSIDE_WINDOW, g_mout);
        sput(C2_S_N_READS_IRREGULAR_ENDS, g_irr); sput(C2_S_ALIGNMENTS_COUNTED, g_n); sput(C2_S_DISCARDED, g_disc);
        int pt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { pt[k] = wsum(g_pat[k]); g_pat[k] = 0; }
        if (lane == 0) {
            auto sa = [&](const int k_, const long long x) { if (x) atomicAdd(scal + k_, (u64)x); };
            const long long P0 = pt[0], P1 = pt[1], P2 = pt[2], P3 = pt[3], P4 = pt[4], P5 = pt[5], P6 = pt[6], P7 = pt[7];
            sa(C2_S_TOTAL, P0 + P1 + P2 + P3 + P4 + P5 + P6 + P7);
            sa(C2_S_UNMODIFIED, P0); sa(C2_S_MODIFIED, P1 + P2 + P3 + P4 + P5 + P6 + P7);                                   // :746-760, :4003-4006
            sa(C2_S_INSERTION, P2 + P3 + P6 + P7); sa(C2_S_DELETION, P4 + P5 + P6 + P7); sa(C2_S_SUBSTITUTION, P1 + P3 + P5 + P7);
            sa(C2_S_ONLY_SUBSTITUTION, P1); sa(C2_S_ONLY_INSERTION, P2); sa(C2_S_INSERTION_AND_SUBSTITUTION, P3);            // :4058-4072
            sa(C2_S_ONLY_DELETION, P4); sa(C2_S_DELETION_AND_SUBSTITUTION, P5); sa(C2_S_INSERTION_AND_DELETION, P6);
            sa(C2_S_INSERTION_AND_DELETION_AND_SUBSTITUTION, P7);
        }
        iput(hist + C2_H_INSERTED_N * A.hl + 0, g_ins0); iput(hist + C2_H_DELETED_N * A.hl + 0, g_del0); iput(hist + C2_H_SUBSTITUTED_N * A.hl + 0, g_sub0);
        iput(hist + C2_H_EFFECTIVE_LEN * A.hl + Li, g_eff0);
        iput(cov + 0, g_cov0); iput(cov + Li, g_covL);
    };
    const u64 chars = (u64)'A' | ((u64)'C' << 8) | ((u64)'T' << 16) | ((u64)'G' << 24) | ((u64)'N' << 56);      // indexed by (ch >> 1) & 7
    unsigned rounds = 0;
    for (uint64_t base = p_lo + (uint64_t)bx * 256u; base < p_hi; base += (uint64_t)gx * 256u) {
        const uint64_t pos = base + (uint64_t)tid;
        const bool in_range = pos < p_hi;
        const uint64_t t = in_range ? (A.ref_ends ? (uint64_t)A.order[pos] : by_layout ? (pos - p_lo) * (uint64_t)A.n_refs + (uint64_t)ref : pos) : 0ull;
        unsigned h = 0;
        if (in_range) h = A.hints[4u * t];
        // a gapped hint's record and other three words: asked for here, looked at behind the main-diagonal tasks' work
        unsigned d0 = 0, d1 = 0, d2 = 0, d4 = 0, d5 = 0;
        uint4 hw = uint4{h, 0u, 0u, 0u};
        if (!(h & C2_HINT_VALID) && (h & C2_HINT_GAPPED)) {
            const uint4* rq = (const uint4*)(A.records + t);
            const uint4 ra = rq[0], rb = rq[1];
            d0 = ra.x; d1 = ra.y; d2 = ra.z; d4 = rb.x; d5 = rb.y;
            hw = *(const uint4*)(A.hints + 4u * t);
        }
        if (A.rest_list) {
            // what this kernel does not take (the rule of c2_count_vectors_body's skip) and what has a weight at all goes to the column walk's list --
            // through an LDS list of the workgroup, emptied with the block (one global atomic per flush: an atomic per wavefront on one address
            // serialised in L2 and cost the kernel 1.2 ms)
            const unsigned wr = in_range ? (A.weights ? A.weights[t] : 1u) : 0u;
            const bool rest = in_range && wr > 0u && !((h & C2_HINT_VALID) || ((h & C2_HINT_GAPPED) && wr < (unsigned)C2_HCNT_SMALL_W));
            if (rest) lrest[atomicAdd(part + 4, 1)] = (uint32_t)t;
        }
        if (h & C2_HINT_VALID) {
            const unsigned wq = A.weights ? A.weights[t] : 1u;
            const int w = (int)(wq > 0x7fffffffu ? 0x7fffffffu : wq);
            const int k = (int)((h >> 24) & 3u);
            if (w > 0 && gate && Li - k >= thresh) {
                int all_sub = 0, sub_n = 0, irregular = 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (e >= k) continue;
                    const int c = (int)((h >> (12 * e)) & 0x1ffu);
                    const unsigned char rd = (unsigned char)(chars >> (8 * ((h >> (12 * e + 9)) & 7u)));
                    const unsigned char rfc = sseq[c];
                    const bool big = w >= C2_HCNT_SMALL_W;
                    auto add = [&](const int idx, const int x) {
                        if (!big) atomicAdd(acc + idx, x);
                        else atomicAdd((u64*)(out + idx), (u64)(long long)x);
                    };
                    const int bvr = c2_base_vector(rd), bvf = c2_base_vector(rfc);
                    if (bvr >= 0) add(bvr * VL + c, w);
                    if (bvf >= 0) add(bvf * VL + c, -w);
                    if (rd != 'N') {                                                       // COREResources.pyx:113-118
                        ++all_sub;
                        add(C2_V_ALL_SUBSTITUTION * VL + c, w);                            // :4040
                        const bool in_win = incp[c + 1] != incp[c];
                        if (in_win) ++sub_n;
                        if (!ign_sub) {
                            if (in_win) add(C2_V_SUBSTITUTION * VL + c, w);                // :4044
                            const int sv = c2_sub_base_vector(rd);                         // :4049-4054
                            if (sv >= 0) add(sv * VL + c, w);
                        }
                    }
                    if (c == 0 || c == Li - 1) irregular = 1;
                }
                const u64 W = (u64)(unsigned)w;
                sW += W; sN += 1ull;
                if (!ign_sub && sub_n > 0) sSubW += W;
                sGsub += W * (u64)all_sub; sOut += W * (u64)(all_sub - sub_n); sIn += W * (u64)sub_n;
                if (irregular) sIrr += W;
                if (sub_n == 0) sH0 += W; else if (sub_n == 1) sH1 += W; else sH2 += W;
            }
        } else if (h & C2_HINT_GAPPED) {
            const unsigned wq = A.weights ? A.weights[t] : 1u;
            if (wq > 0u && wq < (unsigned)C2_HCNT_SMALL_W) {
                const int w = (int)wq;
                const int T = (int)(d0 & 0xffffu), matches = (int)(d0 >> 16);
                bool sel = ((d5 >> 24) == 0) && (T > 0);
                if (sel && mm_row) sel = (T <= A.max_t) && (matches >= (int)mm_row[T]);
                if (sel) {
                    // ---- the scalar counters and histograms, as c2_count_vectors_body adds them from the record (aln_stats of process_fastq,
                    //      CRISPRessoCORE.py:1974-1979; the tallies of :3996-4072)
                    const int insertion_n = (int)(d1 & 0xffffu), deletion_n = (int)(d1 >> 16), substitution_n = (int)(d2 & 0xffffu);
                    const int all_ins = (int)(d2 >> 16), all_del_bases = (int)((d4 >> 16) & 0x7fffu), all_sub = (int)(d5 & 0xffffu);
                    const bool irregular_ends = (d5 >> 16) & 0xffu;
                    const int total_mods = all_ins + all_del_bases + all_sub, in_win = substitution_n + deletion_n + insertion_n;      // :741-742
                    g_gsub += all_sub * w; g_out += (all_sub - substitution_n) * w;
                    g_in += in_win * w; g_mout += (total_mods - in_win) * w;
                    if (irregular_ends) g_irr += w;
                    g_n += 1;
                    const bool has_ins = !ign_ins && insertion_n > 0, has_del = !ign_del && deletion_n > 0, has_sub = !ign_sub && substitution_n > 0;
                    const bool modified = has_del || has_ins || has_sub;
                    if (discard && (deletion_n > 0 || insertion_n > 0)) g_disc += w;                                 // :3996-4000: counted, no vectors
                    else {
                        const int pat = (has_del ? 4 : 0) | (has_ins ? 2 : 0) | (has_sub ? 1 : 0);                   // :746-760, :4003-4006, :4058-4072 (see drain)
#pragma unroll
                        for (int k_ = 0; k_ < 8; ++k_) g_pat[k_] += pat == k_ ? w : 0;
                        if (!ign_ins) { if (insertion_n == 0) g_ins0 += w; else atomicAdd(hist + C2_H_INSERTED_N * A.hl + insertion_n, w); }            // :4020
                        if (!ign_del) { if (deletion_n == 0) g_del0 += w; else atomicAdd(hist + C2_H_DELETED_N * A.hl + deletion_n, w); }              // :4030
                        if (!ign_sub) { if (substitution_n == 0) g_sub0 += w; else atomicAdd(hist + C2_H_SUBSTITUTED_N * A.hl + substitution_n, w); }  // :4043
                        {
                            const int eff = (ign_ins ? 0 : insertion_n) - (ign_del ? 0 : deletion_n);                                                   // :4010-4037
                            if (eff == 0) g_eff0 += w; else atomicAdd(hist + C2_H_EFFECTIVE_LEN * A.hl + Li + eff, w);
                        }
                        // ---- what the column walk adds, run by run (c2_count_vectors_body, "eight columns per lane")
                        const bool len_block = modified;                                                            // :4085 (no coding sequence)
                        const int nruns = (int)(hw.x & 7u), nmm = (int)((hw.x >> 3) & 3u);
                        int ix = 0;                                                                                 // reference bases in front of the run
#pragma unroll
                        for (int f = 0; f < 5; ++f) {
                            if (f >= nruns) continue;
                            const unsigned fld = (f == 0 ? hw.x >> 5 : f == 1 ? hw.x >> 16 : f == 2 ? hw.y : f == 3 ? hw.y >> 11 : hw.z) & 0x7ffu;
                            const int st = (int)(fld & 3u), len = (int)(fld >> 2);
                            if (st == C2_ST_M) {
                                // the read's base IS the reference's (all_base_count, :4075-4081; the differing columns are taken back below)
                                if (ix == 0) g_cov0 += w; else atomicAdd(cov + ix, w);
                                if (ix + len == Li) g_covL -= w; else atomicAdd(cov + ix + len, -w);
                                ix += len;
                            } else if (st == C2_ST_J) {
                                // a deletion: its columns (all_deletion :4028, the '-' base counts) as a range; the window counts as ranges too
                                atomicAdd(dcov + ix, w); atomicAdd(dcov + ix + len, -w);
                                if (incp[ix + len] != incp[ix]) {                                                   // range(start, end) touches the window
                                    if (!ign_del) { atomicAdd(acc + C2_V_DELETION * VL + ix, w); atomicAdd(acc + C2_V_DELETION * VL + ix + len, -w); }   // :4031
                                    if (len_block) { atomicAdd(acc + C2_V_DELETION_LENGTH * VL + ix, len * w); atomicAdd(acc + C2_V_DELETION_LENGTH * VL + ix + len, -len * w); }   // :4114
                                }
                                ix += len;
                            } else {
                                // an insertion closes at the next reference base: positions [ix - 1, ix] (:4016-4021); none in front of the first
                                // reference base, none behind the last run (pyx:119-136)
                                if (ix > 0 && f + 1 < nruns) {
                                    atomicAdd(acc + C2_V_ALL_INSERTION_LEFT * VL + ix - 1, w);                      // :4017
                                    atomicAdd(acc + C2_V_ALL_INSERTION * VL + ix, w);
                                    atomicAdd(acc + C2_V_ALL_INSERTION * VL + ix - 1, w);
                                    if ((incp[ix] != incp[ix - 1]) && (incp[ix + 1] != incp[ix])) {                 // both flanks in the window, pyx:121
                                        if (!ign_ins) { atomicAdd(acc + C2_V_INSERTION * VL + ix, w); atomicAdd(acc + C2_V_INSERTION * VL + ix - 1, w); }
                                        if (len_block) { atomicAdd(acc + C2_V_INSERTION_LENGTH * VL + ix - 1, len * w); atomicAdd(acc + C2_V_INSERTION_LENGTH * VL + ix, len * w); }   // :4104-4106
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            if (e >= nmm) continue;
                            const unsigned ent = (e == 0 ? hw.z >> 11 : e == 1 ? hw.w : hw.w >> 12) & 0xfffu;
                            const int c = (int)(ent & 0x1ffu);
                            const unsigned char rd = (unsigned char)(chars >> (8 * ((ent >> 9) & 7u)));
                            atomicAdd(cov + c, -w); atomicAdd(cov + c + 1, w);                                      // not "the reference's own base" here ...
                            const int bv = c2_base_vector(rd);
                            if (bv >= 0) atomicAdd(acc + bv * VL + c, w);                                           // ... but the read's
                            if (rd != 'N') {
                                atomicAdd(acc + C2_V_ALL_SUBSTITUTION * VL + c, w);                                 // :4040
                                if (!ign_sub) {
                                    if (incp[c + 1] != incp[c]) atomicAdd(acc + C2_V_SUBSTITUTION * VL + c, w);     // :4044
                                    const int sv = c2_sub_base_vector(rd);                                          // :4049-4054
                                    if (sv >= 0) atomicAdd(acc + sv * VL + c, w);
                                }
                            }
                        }
                    }
                }
            }
        }
        ++rounds;
        if ((rounds % C2_HCNT_FLUSH_ROUNDS) == 0u) { drain(); if (A.rest_list) flush_rest(); }
        if ((rounds % acc_rounds) == 0u) flush();
    }
    drain();
    if (A.rest_list) flush_rest();
    flush();
    // the lanes' totals of the main-diagonal tasks -> the wavefront's -> the workgroup's (LDS, 64-bit) -> the tensor
    u64 v[10] = {sW, sN, sSubW, sGsub, sOut, sIn, sIrr, sH0, sH1, sH2};
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        unsigned lo = (unsigned)v[q], hi = (unsigned)(v[q] >> 32);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned olo = (unsigned)__shfl_xor((int)lo, d), ohi = (unsigned)__shfl_xor((int)hi, d);
            const u64 sum = ((u64)hi << 32 | lo) + ((u64)ohi << 32 | olo);
            lo = (unsigned)sum; hi = (unsigned)(sum >> 32);
        }
        if (lane == 0) atomicAdd(tot + q, ((u64)hi << 32) | lo);
    }
    __syncthreads();
    auto gadd = [&](const int idx, const u64 x) { if (x) atomicAdd((u64*)(out + idx), x); };
    if (tid < C2_CNT_SCALARS) gadd(o_sc + tid, scal[tid]);           // the gapped tasks' scalars
    const u64 W = tot[0];
    if (W == 0ull && tot[1] == 0ull) return;
    if (tid == 0) {
        const u64 subw = tot[2];
        gadd(o_sc + C2_S_TOTAL, W); gadd(o_sc + C2_S_MODIFIED, subw); gadd(o_sc + C2_S_UNMODIFIED, W - subw);      // :746-760, :4003-4006
        gadd(o_sc + C2_S_SUBSTITUTION, subw); gadd(o_sc + C2_S_ONLY_SUBSTITUTION, subw);                             // :4058-4072
        gadd(o_sc + C2_S_N_GLOBAL_SUBS, tot[3]); gadd(o_sc + C2_S_N_SUBS_OUTSIDE_WINDOW, tot[4]);
        gadd(o_sc + C2_S_N_MODS_IN_WINDOW, tot[5]); gadd(o_sc + C2_S_N_MODS_OUTSIDE_WINDOW, tot[4]);                 // :741-742
        gadd(o_sc + C2_S_N_READS_IRREGULAR_ENDS, tot[6]); gadd(o_sc + C2_S_ALIGNMENTS_COUNTED, tot[1]);
        if (!ign_ins) gadd(o_h + C2_H_INSERTED_N * A.hl + 0, W);                                                     // :4020
        if (!ign_del) gadd(o_h + C2_H_DELETED_N * A.hl + 0, W);                                                      // :4030
        if (!ign_sub) { gadd(o_h + C2_H_SUBSTITUTED_N * A.hl + 0, tot[7]); gadd(o_h + C2_H_SUBSTITUTED_N * A.hl + 1, tot[8]); gadd(o_h + C2_H_SUBSTITUTED_N * A.hl + 2, tot[9]); }   // :4043
        gadd(o_h + C2_H_EFFECTIVE_LEN * A.hl + Li, W);                                                               // :4010-4037
    }
    // every reference position: the main-diagonal tasks' total weight on the vector of its own base (the deviations above took the differing ones back)
    for (int c = tid; c < Li; c += 256) {
        const int bv = c2_base_vector(rf.seq[c]);
        if (bv >= 0) gadd(bv * VL + c, W);
    }
}
