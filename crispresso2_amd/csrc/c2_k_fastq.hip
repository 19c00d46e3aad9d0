// c2_k_fastq.hip -- FASTQ framing, exact de-duplication, gather and reverse-complement partner look-up on the device.
#pragma once
#include "c2_k_common.h"

// =====================================================================================
// FASTQ framing and exact de-duplication on the device -- the step in front of the align kernels when the HOST is the bottleneck
// (the text reaches HBM at the link's rate, 16 host CPUs parse it five times slower).  Semantics: c2_fastq.cpp's, for
// text without carriage returns -- records are four consecutive '\n'-lines from the top whatever they contain, the sequence line
// str.strip()ped (ASCII whitespace incl. 0x0b 0x0c 0x1c-0x1f), equal sequences counted, first-seen order.
// =====================================================================================
__device__ __forceinline__ bool c2_py_space(const unsigned c) { return (c >= 0x09u && c <= 0x0du) || (c >= 0x1cu && c <= 0x20u); }

// exact per-byte flags (bit 7 of every byte) of "byte == c" in a 32-bit word
__device__ __forceinline__ unsigned c2_eq_bytes(const unsigned w, const unsigned c4) {
    const unsigned x = w ^ c4;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}

// the 64 bytes of a thread: as 16 words (zero beyond `hi`), and the byte in front of them ('\n' in front of the text)
__device__ __forceinline__ void c2_fq_load64(const c2_fq_frame_args& A, const uint64_t pos, unsigned (&w)[16], unsigned& prev) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t p = pos + 16u * q;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (p + 16 <= A.hi) v = *(const uint4*)(A.text + p);
        else if (p < A.hi) { unsigned char tmp[16]; for (int k = 0; k < 16; ++k) tmp[k] = p + k < A.hi ? A.text[p + k] : 0; __builtin_memcpy(&v, tmp, 16); }
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    prev = pos == 0 ? 0x0au : (pos <= A.hi ? (unsigned)A.text[pos - 1] : 0u);
}

__global__ __launch_bounds__(256) void c2_fq_count_kernel(c2_fq_frame_args A)
{
    unsigned* const s_acc = (unsigned*)c2_smem;                     // [3] (dynamic LDS: C2_FQ_LDS_BYTES)
    unsigned& s_nl = s_acc[0]; unsigned& s_em = s_acc[1]; unsigned& s_cr = s_acc[2];
    if (threadIdx.x == 0) { s_nl = 0; s_em = 0; s_cr = 0; }
    __syncthreads();
    const uint64_t pos = A.lo + (uint64_t)blockIdx.x * C2_FQ_TILE + (uint64_t)threadIdx.x * 64u;
    unsigned nl = 0, em = 0, cr = 0;
    if (pos < A.hi) {
        unsigned w[16], prev;
        c2_fq_load64(A, pos, w, prev);
        unsigned before = prev == 0x0au ? 0x80u : 0u;                 // "the byte in front is a newline", as bit 7 of a byte
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned m = c2_eq_bytes(w[q], 0x0a0a0a0au);
            cr |= c2_eq_bytes(w[q], 0x0d0d0d0du);
            nl += (unsigned)__builtin_popcount(m);
            em += (unsigned)__builtin_popcount(m & ((m << 8) | before));      // a newline whose predecessor is a newline
            before = m >> 24;
        }
        // (bytes beyond hi were loaded as 0: neither '\n' nor '\r')
    }
    if (nl) atomicAdd(&s_nl, nl);
    if (em) atomicAdd(&s_em, em);
    if (cr) atomicOr(&s_cr, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        A.tile_newlines[blockIdx.x] = s_nl; A.tile_empty[blockIdx.x] = s_em;
        if (s_cr) atomicOr(A.flags, 1u);
    }
}

__global__ __launch_bounds__(256) void c2_fq_lines_kernel(c2_fq_frame_args A)
{
    unsigned* const s_scan = (unsigned*)c2_smem;                    // [256]
    const uint64_t pos = A.lo + (uint64_t)blockIdx.x * C2_FQ_TILE + (uint64_t)threadIdx.x * 64u;
    unsigned w[16], prev, nl = 0;
    if (pos < A.hi) {
        c2_fq_load64(A, pos, w, prev);
#pragma unroll
        for (int q = 0; q < 16; ++q) nl += (unsigned)__builtin_popcount(c2_eq_bytes(w[q], 0x0a0a0a0au));
    }
    // newlines in front of this thread inside the tile (exclusive scan over the 256 threads)
    s_scan[threadIdx.x] = nl;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned v = threadIdx.x >= (unsigned)d ? s_scan[threadIdx.x - d] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    if (pos >= A.hi || nl == 0) return;
    uint64_t g = A.tile_base[blockIdx.x] + (uint64_t)(s_scan[threadIdx.x] - nl);
    for (int q = 0; q < 16; ++q) {
        unsigned m = c2_eq_bytes(w[q], 0x0a0a0a0au);
        while (m) {
            const int b = __builtin_ctz(m) >> 3;
            m &= m - 1u;
            const uint64_t p = pos + 4u * q + (uint64_t)b;
            const uint64_t r = g >> 2;
            if (r < A.n_records_cap) {
                if ((g & 3u) == 0u) A.seq_start[r] = p + 1;            // newline 4r ends the id line: the sequence line starts behind it
                else if ((g & 3u) == 1u) A.seq_end[r] = p;             // newline 4r + 1 ends the sequence line
                else if (A.qual_start) {                               // (paired input: the quality line too)
                    if ((g & 3u) == 2u) A.qual_start[r] = p + 1;
                    else A.qual_end[r] = p;
                }
            }
            ++g;
        }
    }
}

// ---- paired input: the reading loop of process_paired_fastq (CRISPRessoCORE.py:1309-1334) over two framed texts ----
__device__ __forceinline__ unsigned long long c2_fq_strip(const uint8_t* text, uint64_t s, uint64_t e, bool& too_long) {
    if (e < s) e = s;
    while (s < e && c2_py_space(text[s])) ++s;
    while (e > s && c2_py_space(text[e - 1])) --e;
    if (e - s >= (1ull << 24) || s >= (1ull << 40)) { too_long = true; return 0ull; }
    return ((unsigned long long)s << 24) | (unsigned long long)(e - s);
}

__global__ __launch_bounds__(256) void c2_fq_pair_lengths_kernel(c2_fq_pair_args A)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= A.n) return;
    bool bad = false;
    const unsigned long long s1 = c2_fq_strip(A.text1, A.seq_start1[r], A.seq_end1[r], bad), q1 = c2_fq_strip(A.text1, A.qual_start1[r], A.qual_end1[r], bad);
    const unsigned long long s2 = c2_fq_strip(A.text2, A.seq_start2[r], A.seq_end2[r], bad), q2 = c2_fq_strip(A.text2, A.qual_start2[r], A.qual_end2[r], bad);
    A.s1[r] = s1; A.q1[r] = q1; A.s2[r] = s2; A.q2[r] = q2;
    A.key_len[r] = (int64_t)((s1 & 0xffffffull) + 1ull + (s2 & 0xffffffull));
    A.qual_len[r] = (int64_t)((q1 & 0xffffffull) + 1ull + (q2 & 0xffffffull));
    if (bad) atomicOr(A.flags, 1u);
}

__global__ __launch_bounds__(256) void c2_fq_pair_write_kernel(c2_fq_pair_args A)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4u + (uint64_t)(threadIdx.x >> 6); r < A.n; r += (uint64_t)gridDim.x * 4u) {
        const unsigned long long s1 = A.s1[r], s2 = A.s2[r], q1 = A.q1[r], q2 = A.q2[r];
        const uint64_t l1 = s1 & 0xffffffull, l2 = s2 & 0xffffffull, lq1 = q1 & 0xffffffull, lq2 = q2 & 0xffffffull;
        uint8_t* const ko = A.key_out + A.key_off[r];
        uint8_t* const qo = A.qual_out + A.qual_off[r];
        const uint8_t* const p1 = A.text1 + (s1 >> 24);
        const uint8_t* const p2 = A.text2 + (s2 >> 24);
        for (uint64_t k = (uint64_t)lane; k < l1; k += 64) ko[k] = p1[k];
        if (lane == 0) { ko[l1] = (uint8_t)'+'; qo[lq1] = (uint8_t)' '; }
        bool bad = false;
        for (uint64_t k = (uint64_t)lane; k < l2; k += 64) {
            const unsigned c = c2_fq_complement(p2[l2 - 1 - k]);
            bad = bad || c == 0u;
            ko[l1 + 1 + k] = (uint8_t)c;
        }
        const uint8_t* const g1 = A.text1 + (q1 >> 24);
        const uint8_t* const g2 = A.text2 + (q2 >> 24);
        for (uint64_t k = (uint64_t)lane; k < lq1; k += 64) qo[k] = g1[k];
        for (uint64_t k = (uint64_t)lane; k < lq2; k += 64) qo[lq1 + 1 + k] = g2[lq2 - 1 - k];
        if (bad) atomicOr(A.flags, 2u);
    }
}

// weight of text position k in a sequence's hash: an odd 64-bit number from a mix of k (splitmix64's finaliser)
__device__ __forceinline__ unsigned long long c2_fq_weight(unsigned long long k) {
    k += 0x9e3779b97f4a7c15ull;
    k = (k ^ (k >> 30)) * 0xbf58476d1ce4e5b9ull;
    k = (k ^ (k >> 27)) * 0x94d049bb133111ebull;
    return (k ^ (k >> 31)) | 1ull;
}

// Same-address atomics serialise in L2 (measured: 80 k of them per launch cost 1 ms -- in real data more than half of the reads are
// one sequence): the table slot is READ before it is CAS-ed, `first` is read before it is lowered, and the occurrences are added up
// in a small LDS table per workgroup (C2_FQ_AGG entries: slot -> count; a collision goes to HBM directly) that is flushed at the end.
#define C2_FQ_AGG 256
__global__ __launch_bounds__(256) void c2_fq_dedup_kernel(c2_fq_dedup_args A)
{
    unsigned* const agg_key = (unsigned*)c2_smem;                   // [C2_FQ_AGG] slot + 1, 0 = free       (dynamic LDS: C2_FQ_DEDUP_LDS_BYTES)
    unsigned* const agg_cnt = agg_key + C2_FQ_AGG;                  // [C2_FQ_AGG]
    unsigned* const agg_stats = agg_cnt + C2_FQ_AGG;                // [3] keys created, longest, empty keys
    const int lane = threadIdx.x & 63;
    const uint64_t r0 = A.range[0], r1 = A.range[1];
    if (r1 > A.n_records_cap || r1 >= 0xffffffffull) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(A.flags, 4u); return; }
    for (unsigned e = threadIdx.x; e < 2u * C2_FQ_AGG + 3u; e += blockDim.x) agg_key[e] = 0u;
    __syncthreads();
    for (uint64_t r = r0 + (uint64_t)blockIdx.x * 4u + (uint64_t)(threadIdx.x >> 6); r < r1; r += (uint64_t)gridDim.x * 4u) {
        uint64_t s = A.seq_start[r], e = A.seq_end[r];
        if (e < s) e = s;
        // str.strip(): whitespace at either end goes (one probe at each end decides the usual read)
        while (s < e && c2_py_space(A.text[s])) ++s;
        while (e > s && c2_py_space(A.text[e - 1])) --e;
        const uint64_t len = e - s;
        if (len >= (1ull << 24) || s >= (1ull << 40)) {
            if (lane == 0) { atomicOr(A.flags, 2u); A.slot_of[r] = 0xffffffffu; A.rinfo[r] = 0ull; }
            continue;
        }
        const unsigned long long me = ((unsigned long long)s << 24) | (unsigned long long)len;
        // hash: sum over the bytes of (byte + 1) * weight(position in the sequence), 64-bit wrap-around; lane l takes bytes l, l + 64, ...
        unsigned long long h = 0;
        for (uint64_t k = (uint64_t)lane; k < len; k += 64) h += ((unsigned long long)A.text[s + k] + 1ull) * c2_fq_weight(k);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(h & 0xffffffffull), d), hi = (unsigned)__shfl_xor((int)(unsigned)(h >> 32), d);
            h += ((unsigned long long)hi << 32) | (unsigned long long)lo;
        }
        h ^= len * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
        if (lane == 0) A.rinfo[r] = me;
        uint64_t p = h & A.mask;
        for (;;) {
            unsigned long long cur = 0;
            if (lane == 0) {
                cur = __hip_atomic_load(A.slots + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == 0ull) cur = atomicCAS(A.slots + p, 0ull, me);          // empty: this record's own bytes become the key's representative
            }
            cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur >> 32)) << 32) |
                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur & 0xffffffffull));
            const bool created = cur == 0ull;
            bool same = false;
            if (!created) {
                const uint64_t os = cur >> 24, ol = cur & 0xffffffull;
                if (ol == len) {
                    bool eq = true;
                    for (uint64_t k = (uint64_t)lane; k < len; k += 64) eq = eq && A.text[os + k] == A.text[s + k];
                    same = __ballot(!eq) == 0ull;
                }
            }
            if (created || same) {
                if (lane == 0) {
                    A.slot_of[r] = (uint32_t)p;
                    if (__hip_atomic_load(A.first + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned)r) atomicMin(A.first + p, (unsigned)r);
                    const unsigned a = ((unsigned)p * 0x9e3779b1u) >> 24;         // (C2_FQ_AGG = 2^8)
                    const unsigned was = atomicCAS(agg_key + a, 0u, (unsigned)p + 1u);
                    if (was == 0u || was == (unsigned)p + 1u) atomicAdd(agg_cnt + a, 1u);
                    else atomicAdd(A.count + p, 1u);
                    if (created) {
                        atomicAdd(agg_stats, 1u);
                        atomicMax(agg_stats + 1, (unsigned)len);
                        if (len == 0) atomicAdd(agg_stats + 2, 1u);
                    }
                }
                break;
            }
            p = (p + 1) & A.mask;
        }
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < C2_FQ_AGG; e += blockDim.x) if (agg_key[e]) atomicAdd(A.count + (agg_key[e] - 1u), agg_cnt[e]);
    if (threadIdx.x == 0 && agg_stats[0]) {
        atomicAdd(A.stats, agg_stats[0]);
        atomicMax(A.stats + 1, agg_stats[1]);
        if (agg_stats[2]) atomicAdd(A.stats + 2, agg_stats[2]);
    }
}

__global__ __launch_bounds__(256) void c2_fq_gather_kernel(c2_fq_gather_args A)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t i = (uint64_t)blockIdx.x * 4u + (uint64_t)(threadIdx.x >> 6); i < A.n; i += (uint64_t)gridDim.x * 4u) {
        const unsigned long long info = A.info[A.records ? (uint64_t)A.records[i] : i];
        const uint64_t s = info >> 24, len = info & 0xffffffull;
        uint8_t* o = A.out + A.out_offsets[i];
        for (uint64_t k = (uint64_t)lane; k < len; k += 64) o[k] = A.text[s + k];
    }
}

__global__ __launch_bounds__(256) void c2_fq_rc_partner_kernel(c2_fq_rc_args A)
{
    const int lane = threadIdx.x & 63;
    for (uint64_t i = (uint64_t)blockIdx.x * 4u + (uint64_t)(threadIdx.x >> 6); i < A.n; i += (uint64_t)gridDim.x * 4u) {
        const unsigned long long info = A.info[A.records[i]];
        const uint64_t s = info >> 24, len = info & 0xffffffull;
        // the hash the de-duplication kernel would give the reverse complement: its byte k is the complement of this read's byte len - 1 - k
        unsigned long long h = 0;
        bool bad = false;
        for (uint64_t k = (uint64_t)lane; k < len; k += 64) {
            const unsigned c = c2_fq_complement(A.text[s + len - 1 - k]);
            bad = bad || c == 0u;
            h += ((unsigned long long)c + 1ull) * c2_fq_weight(k);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(h & 0xffffffffull), d), hi = (unsigned)__shfl_xor((int)(unsigned)(h >> 32), d);
            h += ((unsigned long long)hi << 32) | (unsigned long long)lo;
        }
        h ^= len * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
        int found = -1;
        if (__ballot(bad) == 0ull) {
            uint64_t p = h & A.mask;
            for (;;) {
                const unsigned long long cur = A.slots[p];             // (wave-uniform address)
                if (cur == 0ull) break;
                const uint64_t os = cur >> 24, ol = cur & 0xffffffull;
                if (ol == len) {
                    bool eq = true;
                    for (uint64_t k = (uint64_t)lane; k < len; k += 64) eq = eq && (unsigned)A.text[os + k] == c2_fq_complement(A.text[s + len - 1 - k]);
                    if (__ballot(!eq) == 0ull) { found = (int)p; break; }
                }
                p = (p + 1) & A.mask;
            }
        }
        if (lane == 0) A.partner_slot[i] = found;
    }
}
