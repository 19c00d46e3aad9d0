// c2_k_alleles.hip -- the allele frequency table on the device.
//
// Replaces the reference's per-variant Python loop that fills alleles_list, the DataFrame sort and the text of
// Alleles_frequency_table.txt (CRISPRessoCORE.py:3964-4010 rows, :4298-4303 sort and %Reads, :4498-4530 file) and the grouping of
// <ref>Alleles_frequency_table_around_<guide>.txt (CRISPRessoShared.py:1513-1531), from what the count route left in HBM: the aligned
// strings + 32-byte records of every (read, reference) alignment, the selection kernel's masks, the merged multiplicities.
//
//   c2_allele_jobs_kernel     one lane per unique read: which table rows it gives (pass 1: how many; pass 2, after a scan: the rows)
//   c2_allele_row_less        the table's order as a comparator over row indices (device merge sort; the strings are compared where they lie)
//   c2_allele_reads_kernel    #Reads in sorted order (the host finds the runs of equal values and prints their %Reads once)
//   c2_allele_probe_kernel    the two "contains dsODN" columns: one wavefront per row
//   c2_allele_text_kernel     pass 1 (one lane per row): bytes of its line; pass 2 (one wavefront per row): the line itself, into a chunk buffer
//   c2_allele_fetch_kernel    the sorted rows with zero-padded strings, for a caller that wants them in memory
//   c2_allele_window_kernel   around-cut windows as fixed-width keys whose byte order is the reference's group order
//   c2_allele_key_less        ... their comparator; c2_allele_group_kernel: group numbers and one key per group
// HBM-bound byte work: coalesced byte / dword traffic, no LDS beyond 128 bytes per wavefront for a line's numeric tail.
#pragma once
#include "c2_k_common.h"

__host__ __device__ inline const uint8_t* c2_allele_a(const c2_allele_strings& X, const uint32_t src) {
    return (src & 0x80000000u) ? X.a2 + (uint64_t)(src & 0x7fffffffu) * X.stride2 : X.a1 + (uint64_t)src * X.stride1;
}
__host__ __device__ inline const uint8_t* c2_allele_f(const c2_allele_strings& X, const uint32_t src) {
    return (src & 0x80000000u) ? X.f2 + (uint64_t)(src & 0x7fffffffu) * X.stride2 : X.f1 + (uint64_t)src * X.stride1;
}

// Python's str order of p[0:n] against q[0:m] (ASCII: byte order; a proper prefix sorts first): < 0, 0, > 0.  Both rows start on an
// 8-byte boundary and are readable to the next multiple of 8 (row strides are multiples of 16); bytes beyond n / m are not looked at.
__host__ __device__ inline int c2_bytes_cmp(const uint8_t* p, const uint32_t n, const uint8_t* q, const uint32_t m) {
    const uint32_t c = n < m ? n : m;
    const uint64_t* pw = (const uint64_t*)p;
    const uint64_t* qw = (const uint64_t*)q;
    const uint32_t full = c >> 3;
    for (uint32_t w = 0; w < full; ++w) {
        const uint64_t u = pw[w], v = qw[w];
        if (u != v) return __builtin_bswap64(u) < __builtin_bswap64(v) ? -1 : 1;
    }
    const uint32_t t = c & 7u;
    if (t) {
        const uint64_t u = __builtin_bswap64(pw[full]) >> (64u - 8u * t), v = __builtin_bswap64(qw[full]) >> (64u - 8u * t);
        if (u != v) return u < v ? -1 : 1;
    }
    return n < m ? -1 : n > m ? 1 : 0;
}

// df_alleles.sort_values(by=['#Reads', 'Aligned_Sequence', 'Reference_Sequence'], ascending=[False, True, True]) (CRISPRessoCORE.py:4303;
// pandas sorts several keys with a stable lexsort: rows that tie keep the order of alleles_list = unique-read order, then reference order)
struct c2_allele_row_less {
    c2_allele_strings X;
    const c2_allele_row* rows;
    __host__ __device__ bool operator()(const uint32_t& x, const uint32_t& y) const {
        const c2_allele_row rx = rows[x], ry = rows[y];
        if (rx.reads != ry.reads) return rx.reads > ry.reads;
        int c = c2_bytes_cmp(c2_allele_a(X, rx.src), rx.aln_len, c2_allele_a(X, ry.src), ry.aln_len);
        if (c) return c < 0;
        c = c2_bytes_cmp(c2_allele_f(X, rx.src), rx.aln_len, c2_allele_f(X, ry.src), ry.aln_len);
        if (c) return c < 0;
        return x < y;
    }
};

// fixed-width keys (key_bytes a multiple of 8) in byte order; ties keep subset order
struct c2_allele_key_less {
    const uint8_t* keys;
    uint32_t key_bytes;
    __host__ __device__ bool operator()(const uint32_t& x, const uint32_t& y) const {
        const uint64_t* p = (const uint64_t*)(keys + (uint64_t)x * key_bytes);
        const uint64_t* q = (const uint64_t*)(keys + (uint64_t)y * key_bytes);
        for (uint32_t w = 0; w < key_bytes / 8u; ++w) {
            const uint64_t u = p[w], v = q[w];
            if (u != v) return __builtin_bswap64(u) < __builtin_bswap64(v);
        }
        return x < y;
    }
};

// The rows a unique read gives (CRISPRessoCORE.py:3964-4010): none unless it aligned and still has copies after the reverse-complement
// transfer; a read the scaffold rule took: one row, its alignment against the Prime-edited amplicon, labelled 'Scaffold-incorporated';
// an ambiguous read: one 'AMBIGUOUS_<first best reference>' row (:3989-3993), or its first best reference only
// (--assign_ambiguous_alignments_to_first_reference), or one row per best reference (--expand_ambiguous_alignments);
// --discard_indel_reads relabels a counted row with an indel in the window 'DISCARDED_<first best reference>' (:3998-4002).
__global__ __launch_bounds__(256) void c2_allele_jobs_kernel(c2_allele_jobs_args A)
{
    const c2_allele_src& S = A.S;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= S.n_reads) return;
    const int k = S.n_refs, words = (k + 63) >> 6;
    const bool live = (S.d_flags[i] & C2_SEL_FLAG_ALIGNED) && S.d_counts[i] > 0u;
    const bool scaf = live && S.d_scaffold_hit != nullptr && S.d_scaffold_hit[i] != 0;
    int nb = 0, first = -1;
    if (live && !scaf)
        for (int w = 0; w < words; ++w) {
            const unsigned long long mw = S.d_member[i * (uint64_t)words + (uint64_t)w];
            if (mw && first < 0) first = w * 64 + __builtin_ctzll(mw);
            nb += __popcll(mw);
        }
    const int nj = !live ? 0 : scaf ? 1 : (nb > 1 && S.mode != C2_SEL_MODE_EXPAND) ? 1 : nb;
    if (A.offsets == nullptr) { A.njobs[i] = (uint32_t)nj; return; }
    if (nj == 0) return;
    uint64_t o = A.offsets[i];
    const uint32_t reads = S.d_counts[i];
    auto emit = [&](const int r, int label, const bool counted) {
        const bool in2 = S.d_use2 != nullptr && S.d_slot2 != nullptr && ((S.d_use2[i * (uint64_t)words + (uint64_t)(r >> 6)] >> (r & 63)) & 1ull);
        const uint64_t t1 = i * (uint64_t)k + (uint64_t)r;
        const uint32_t src = in2 ? (0x80000000u | (uint32_t)S.d_slot2[t1]) : (uint32_t)t1;
        const c2_aln_record rec = in2 ? S.d_records2[(uint32_t)S.d_slot2[t1]] : S.d_records1[t1];
        const bool modified = (!(S.flags & C2_CNT_FLAG_IGNORE_DELETIONS) && rec.deletion_n > 0) || (!(S.flags & C2_CNT_FLAG_IGNORE_INSERTIONS) && rec.insertion_n > 0) ||
                              (!(S.flags & C2_CNT_FLAG_IGNORE_SUBSTITUTIONS) && rec.substitution_n > 0);
        if (counted && (S.flags & C2_CNT_FLAG_DISCARD_INDEL_READS) && (rec.deletion_n > 0 || rec.insertion_n > 0))
            label = label == 3 * k ? 3 * k + 1 : 2 * k + first;
        c2_allele_row row;
        row.src = src; row.reads = reads; row.read = (uint32_t)i; row.aln_len = rec.aln_len; row.label = (uint16_t)label;
        row.n_deleted = rec.deletion_n; row.n_inserted = rec.insertion_n; row.n_mutated = rec.substitution_n;
        row.modified = modified ? 1 : 0; row.reserved = 0;
        A.rows[o++] = row;
    };
    if (scaf) emit(S.scaffold_ref, 3 * k, true);
    else if (nb > 1 && S.mode == C2_SEL_MODE_DROP_AMBIGUOUS) emit(first, k + first, false);
    else if (nb > 1 && S.mode == C2_SEL_MODE_FIRST) emit(first, first, true);
    else
        for (int w = 0; w < words; ++w) {
            unsigned long long mw = S.d_member[i * (uint64_t)words + (uint64_t)w];
            while (mw) {
                const int r = w * 64 + __builtin_ctzll(mw);
                mw &= mw - 1;
                emit(r, r, true);
            }
        }
}

__global__ __launch_bounds__(256) void c2_allele_reads_kernel(const c2_allele_row* rows, const uint32_t* order, uint64_t m, uint32_t* out)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q < m) out[q] = rows[order[q]].reads;
}

// `str.find(probe) > 0` of the aligned read against four probes (CRISPRessoCORE.py:4512-4522): the FIRST occurrence must lie behind column 0,
// so an occurrence at column 0 makes the answer False whatever follows; the empty probe is found at 0.  One wavefront per row.
__global__ __launch_bounds__(256) void c2_allele_probe_kernel(c2_allele_probe_args A)
{
    const int lane = threadIdx.x & 63;
    const uint64_t q = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= A.m) return;                                                    // (wave-uniform)
    const c2_allele_row row = A.rows[A.order[q]];
    const uint8_t* a = c2_allele_a(A.X, row.src);
    const int T = row.aln_len;
    unsigned res = 0;
    for (int p = 0; p < 4; ++p) {
        const uint8_t* pr = A.probe_blob + A.probe_off[p];
        const int pl = (int)(A.probe_off[p + 1] - A.probe_off[p]);
        bool found = false;
        if (pl > 0 && pl <= T) {
            // at column 0: the lanes compare pl bytes between them
            bool diff = false;
            for (int c = lane; c < pl; c += 64) diff |= a[c] != pr[c];
            const bool at0 = __ballot(diff) == 0ull;
            if (!at0) {
                for (int base = 1; base <= T - pl; base += 64) {             // (wave-uniform trip count)
                    const int s = base + lane;
                    bool hit = s <= T - pl;
                    for (int c = 0; hit && c < pl; ++c) hit = a[s + c] == pr[c];
                    if (__ballot(hit) != 0ull) { found = true; break; }
                }
            }
        }
        if (found) res |= 1u << p;
    }
    if (lane == 0) A.probe_bits[q] = (uint8_t)(((res & 3u) ? 1u : 0u) | ((res & 12u) ? 2u : 0u));
}

__host__ __device__ inline uint32_t c2_dec_digits(const uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u :
           v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
// the run of equal #Reads values sorted position q lies in: the last u with run_start[u] <= q
__device__ __forceinline__ uint32_t c2_allele_run_of(const c2_allele_text_args& A, const uint32_t q) {
    uint32_t lo = 0, hi = A.n_runs;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A.run_start[mid] <= q) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t c2_allele_tail_bytes(const c2_allele_text_args& A, const c2_allele_row& row, const uint32_t q) {
    uint32_t n = (row.modified ? 8u : 10u) + 1u + c2_dec_digits(row.n_deleted) + 1u + c2_dec_digits(row.n_inserted) + 1u + c2_dec_digits(row.n_mutated) + 1u +
                 c2_dec_digits(row.reads) + 1u + A.pct_len[c2_allele_run_of(A, q)];
    if (A.probe_bits) { const unsigned b = A.probe_bits[q]; n += 1u + ((b & 1u) ? 4u : 5u) + 1u + ((b & 2u) ? 4u : 5u); }
    return n + 1u;                                                            // '\n'
}

// pass 1: bytes of line q
__global__ __launch_bounds__(256) void c2_allele_lengths_kernel(c2_allele_text_args A)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= A.m) return;
    const c2_allele_row row = A.rows[A.order[q]];
    A.lengths[q] = 2u * row.aln_len + 2u + (A.label_off[row.label + 1] - A.label_off[row.label]) + 1u + c2_allele_tail_bytes(A, row, (uint32_t)q);
}

__device__ __forceinline__ uint32_t c2_put_dec(uint8_t* p, uint32_t v) {
    const uint32_t n = c2_dec_digits(v);
    for (uint32_t k = n; k-- > 0;) { p[k] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return n;
}

// pass 2: lines [q0, q1), one wavefront each (4 per workgroup; 128 bytes of LDS per wavefront hold the numeric tail lane 0 prints)
__global__ __launch_bounds__(256) void c2_allele_emit_kernel(c2_allele_text_args A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* tail = c2_smem + 128 * wave;
    const uint64_t q = A.q0 + (uint64_t)blockIdx.x * 4u + (uint64_t)wave;
    const bool live = q < A.q1;
    c2_allele_row row;
    uint32_t tl = 0;
    if (live) {
        row = A.rows[A.order[q]];
        if (lane == 0) {
            uint32_t n = 0;
            const char* st = row.modified ? "MODIFIED" : "UNMODIFIED";
            for (uint32_t k = 0; k < (row.modified ? 8u : 10u); ++k) tail[n++] = (uint8_t)st[k];
            tail[n++] = '\t'; n += c2_put_dec(tail + n, row.n_deleted);
            tail[n++] = '\t'; n += c2_put_dec(tail + n, row.n_inserted);
            tail[n++] = '\t'; n += c2_put_dec(tail + n, row.n_mutated);
            tail[n++] = '\t'; n += c2_put_dec(tail + n, row.reads);
            tail[n++] = '\t';
            const uint32_t u = c2_allele_run_of(A, (uint32_t)q);
            for (uint32_t k = 0; k < A.pct_len[u]; ++k) tail[n++] = A.pct_blob[A.pct_off[u] + k];
            if (A.probe_bits) {
                const unsigned b = A.probe_bits[q];
                for (int h = 0; h < 2; ++h) {
                    tail[n++] = '\t';
                    const char* w = ((b >> h) & 1u) ? "True" : "False";
                    for (uint32_t k = 0; k < (((b >> h) & 1u) ? 4u : 5u); ++k) tail[n++] = (uint8_t)w[k];
                }
            }
            tail[n++] = '\n';
        }
        tl = c2_allele_tail_bytes(A, row, (uint32_t)q);
    }
    __syncthreads();
    if (!live) return;
    const uint8_t* a = c2_allele_a(A.X, row.src);
    const uint8_t* f = c2_allele_f(A.X, row.src);
    const uint32_t T = row.aln_len;
    uint8_t* o = A.out + (A.offsets[q] - A.offsets[A.q0]);
    for (uint32_t c = lane; c < T; c += 64) o[c] = a[c];
    o += T;
    if (lane == 0) o[0] = '\t';
    o += 1;
    for (uint32_t c = lane; c < T; c += 64) o[c] = f[c];
    o += T;
    if (lane == 0) o[0] = '\t';
    o += 1;
    const uint32_t l0 = A.label_off[row.label], ll = A.label_off[row.label + 1] - l0;
    for (uint32_t c = lane; c < ll; c += 64) o[c] = A.label_blob[l0 + c];
    o += ll;
    if (lane == 0) o[0] = '\t';
    o += 1;
    for (uint32_t c = lane; c < tl; c += 64) o[c] = tail[c];
}

// the sorted rows themselves, strings zero-padded to `stride`: one wavefront per row
__global__ __launch_bounds__(256) void c2_allele_fetch_kernel(c2_allele_fetch_args A)
{
    const int lane = threadIdx.x & 63;
    const uint64_t q = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= A.m) return;
    const c2_allele_row row = A.rows[A.order[q]];
    if (lane == 0 && A.out_rows) A.out_rows[q] = row;
    const uint32_t T = row.aln_len;
    if (A.out_a) { const uint8_t* a = c2_allele_a(A.X, row.src); for (uint32_t c = lane; c < A.stride; c += 64) A.out_a[q * (uint64_t)A.stride + c] = c < T ? a[c] : (uint8_t)0; }
    if (A.out_f) { const uint8_t* f = c2_allele_f(A.X, row.src); for (uint32_t c = lane; c < A.stride; c += 64) A.out_f[q * (uint64_t)A.stride + c] = c < T ? f[c] : (uint8_t)0; }
}

// get_dataframe_around_cut_asymmetrical (CRISPRessoShared.py:1513-1531): the rows of ONE reference, each cut down to the alignment columns
// [cut_idx - left + 1, cut_idx + right + 1) where cut_idx = ref_positions.index(cut_point) = the column of reference base cut_point.
// Flag pass (sub_index == NULL): one lane per row, flag[q] = the row carries the label.  Key pass: one wavefront per flagged row.
__global__ __launch_bounds__(256) void c2_allele_window_kernel(c2_allele_window_args A)
{
    if (A.sub_index == nullptr) {
        const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
        if (q < A.m) A.flag[q] = A.rows[A.order[q]].label == (uint16_t)A.label ? 1u : 0u;
        return;
    }
    const int lane = threadIdx.x & 63;
    const uint64_t q = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= A.m) return;
    const c2_allele_row row = A.rows[A.order[q]];
    if (row.label != (uint16_t)A.label) return;                               // (wave-uniform)
    const uint64_t s = A.sub_index[q];
    const uint8_t* a = c2_allele_a(A.X, row.src);
    const uint8_t* f = c2_allele_f(A.X, row.src);
    const int T = row.aln_len;
    int seen = 0, col = -1;
    for (int base = 0; base < T; base += 64) {
        const int c = base + lane;
        const bool ng = c < T && f[c] != '-';
        const unsigned long long mask = __ballot(ng);
        const int cnt = __popcll(mask);
        if (seen + cnt > A.cut_point) {
            const int want = A.cut_point - seen;
            const int before = __popcll(mask & ((1ull << lane) - 1ull));
            const unsigned long long hm = __ballot(ng && before == want);
            col = base + __builtin_ctzll(hm);
            break;
        }
        seen += cnt;
    }
    if (col < 0) { if (lane == 0) atomicOr(A.error, 1u); col = 0; }
    const int start = col - A.left + 1 > 0 ? col - A.left + 1 : 0, stop = col + A.right + 1 < T ? col + A.right + 1 : T;
    uint8_t* key = A.keys + s * (uint64_t)A.key_bytes;
    const uint32_t W = A.W;
    for (uint32_t b = lane; b < A.key_bytes; b += 64) {
        uint8_t v = 0;
        if (b < W) { const int c = start + (int)b; v = c < stop ? a[c] : (uint8_t)0; }
        else if (b < 2u * W) { const int c = start + (int)(b - W); v = c < stop ? f[c] : (uint8_t)0; }
        else if (b == 2u * W) v = row.modified ? 0 : 1;                        // Unedited (False sorts first)
        else if (b == 2u * W + 1u) v = (uint8_t)(row.n_deleted >> 8);
        else if (b == 2u * W + 2u) v = (uint8_t)(row.n_deleted & 255u);
        else if (b == 2u * W + 3u) v = (uint8_t)(row.n_inserted >> 8);
        else if (b == 2u * W + 4u) v = (uint8_t)(row.n_inserted & 255u);
        else if (b == 2u * W + 5u) v = (uint8_t)(row.n_mutated >> 8);
        else if (b == 2u * W + 6u) v = (uint8_t)(row.n_mutated & 255u);
        key[b] = v;
    }
    if (lane == 0) A.sub_reads[s] = row.reads;
}

// sorted keys -> groups.  Pass 0 (head_scan == NULL): head[j] = key of position j differs from the one before.  Pass 1: the group of
// every subset row (gid, indexed by subset index = table order) and one key per group, in key order.
__global__ __launch_bounds__(256) void c2_allele_group_kernel(c2_allele_group_args A)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= A.ms) return;
    const uint64_t* kj = (const uint64_t*)(A.keys + (uint64_t)A.perm[j] * A.key_bytes);
    const uint32_t words = A.key_bytes / 8u;
    if (A.head_scan == nullptr) {
        bool differs = j == 0;
        if (!differs) {
            const uint64_t* kp = (const uint64_t*)(A.keys + (uint64_t)A.perm[j - 1] * A.key_bytes);
            for (uint32_t w = 0; w < words; ++w) differs |= kj[w] != kp[w];
        }
        A.head[j] = differs ? 1u : 0u;
        return;
    }
    const uint32_t g = (uint32_t)(A.head_scan[j] + A.head[j] - 1u);
    A.gid[A.perm[j]] = g;
    if (A.head[j]) {
        uint64_t* o = (uint64_t*)(A.gkeys + (uint64_t)g * A.key_bytes);
        for (uint32_t w = 0; w < words; ++w) o[w] = kj[w];
    }
}
