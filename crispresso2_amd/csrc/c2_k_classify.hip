// c2_k_classify.hip -- the per-call classifier with full position lists, the paired-read consensus walk, calculate_homology.
#pragma once
#include "c2_k_common.h"

// =====================================================================================
// Per-call classifier with full position lists: find_indels_substitutions
// (CRISPRessoCOREResources.pyx:68-187) and find_indels_substitutions_legacy (pyx:190-315).
// Accepts ANY pair of equal-length strings, including shapes the aligner never emits
// (double-gap columns, insertion next to deletion), so it follows the reference's column
// walk literally.  This is the drop-in for the reference's per-call API (O(10) calls per
// run); the throughput path is the fused classifier in c2_align_classify_kernel.
// One lane walks; launch <<<1, 64>>>.
// =====================================================================================
struct c2_list_writer {
    int32_t* base; int32_t cap; int32_t n;
    __device__ __forceinline__ void push(int32_t v) { if (n < cap) base[n] = v; n++; }
};

__device__ inline bool c2_inc_has(const int32_t* inc, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (inc[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && inc[lo] == x;
}
// include_set.intersection(range(a, b)) non-empty
__device__ inline bool c2_inc_hits(const int32_t* inc, int n, int a, int b) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (inc[mid] < a) lo = mid + 1; else hi = mid; }
    return lo < n && inc[lo] < b;
}

// The column walk of one alignment by one lane; the caller owns the list writers (L[C2_LIST_REF_POSITIONS] must have
// room for all n entries: the walk reads them back).
__device__ __forceinline__ void c2_classify_walk(const uint8_t* rd, const uint8_t* rf, const int n, const int32_t* inc, const int ni,
                                                 const int legacy, c2_list_writer (&L)[C2_LIST_COUNT], int64_t& ins_n, int64_t& del_n)
{
    int32_t* rp = L[C2_LIST_REF_POSITIONS].base;
    ins_n = 0; del_n = 0;
    int idx = 0;
    if (!legacy) {
        int start_deletion = -1, start_insertion = -1, cur_ins = 0;                  // pyx:94,101,109
        for (int c = 0; c < n; ++c) {                                                 // pyx:110
            if (rf[c] != '-') {
                L[C2_LIST_REF_POSITIONS].push(idx);
                if (rf[c] != rd[c] && rd[c] != '-' && rd[c] != 'N') {                 // pyx:113
                    L[C2_LIST_ALL_SUBSTITUTION_POSITIONS].push(idx); L[C2_LIST_ALL_SUBSTITUTION_VALUES].push(rd[c]);
                    if (c2_inc_has(inc, ni, idx)) { L[C2_LIST_SUBSTITUTION_POSITIONS].push(idx); L[C2_LIST_SUBSTITUTION_VALUES].push(rd[c]); }
                }
                if (start_insertion != -1) {                                          // pyx:119-128
                    L[C2_LIST_ALL_INSERTION_LEFT_POSITIONS].push(start_insertion);
                    L[C2_LIST_ALL_INSERTION_POSITIONS].push(start_insertion); L[C2_LIST_ALL_INSERTION_POSITIONS].push(idx);
                    if (c2_inc_has(inc, ni, start_insertion) && c2_inc_has(inc, ni, idx)) {
                        L[C2_LIST_INSERTION_COORDINATES].push(start_insertion); L[C2_LIST_INSERTION_COORDINATES].push(idx);
                        L[C2_LIST_INSERTION_POSITIONS].push(start_insertion); L[C2_LIST_INSERTION_POSITIONS].push(idx);
                        L[C2_LIST_INSERTION_SIZES].push(cur_ins); ins_n += cur_ins;
                    }
                    start_insertion = -1;
                }
                cur_ins = 0; idx++;
            } else {                                                                  // pyx:131-138
                L[C2_LIST_REF_POSITIONS].push(idx == 0 ? -1 : -idx);
                if (idx > 0 && start_insertion == -1) start_insertion = idx - 1;
                cur_ins++;
            }
            if (rd[c] == '-' && start_deletion == -1) {                               // pyx:140-144
                start_deletion = (c - 1 >= 0) ? rp[c] : 0;
            } else if (rd[c] != '-' && start_deletion != -1) {                        // pyx:145-153
                const int end_deletion = rp[c];
                for (int q = start_deletion; q < end_deletion; ++q) L[C2_LIST_ALL_DELETION_POSITIONS].push(q);
                L[C2_LIST_ALL_DELETION_COORDINATES].push(start_deletion); L[C2_LIST_ALL_DELETION_COORDINATES].push(end_deletion);
                if (c2_inc_hits(inc, ni, start_deletion, end_deletion)) {
                    for (int q = start_deletion; q < end_deletion; ++q) L[C2_LIST_DELETION_POSITIONS].push(q);
                    L[C2_LIST_DELETION_COORDINATES].push(start_deletion); L[C2_LIST_DELETION_COORDINATES].push(end_deletion);
                    L[C2_LIST_DELETION_SIZES].push(end_deletion - start_deletion); del_n += end_deletion - start_deletion;
                }
                start_deletion = -1;
            }
        }
        if (start_deletion != -1 && n > 0) {                                          // pyx:155-162
            const int end_deletion = rp[n - 1];
            for (int q = start_deletion; q < end_deletion + 1; ++q) L[C2_LIST_ALL_DELETION_POSITIONS].push(q);
            L[C2_LIST_ALL_DELETION_COORDINATES].push(start_deletion); L[C2_LIST_ALL_DELETION_COORDINATES].push(end_deletion + 1);
            if (c2_inc_hits(inc, ni, start_deletion, end_deletion + 1)) {
                for (int q = start_deletion; q < end_deletion + 1; ++q) L[C2_LIST_DELETION_POSITIONS].push(q);
                L[C2_LIST_DELETION_COORDINATES].push(start_deletion); L[C2_LIST_DELETION_COORDINATES].push(end_deletion + 1);
                L[C2_LIST_DELETION_SIZES].push(end_deletion + 1 - start_deletion); del_n += end_deletion + 1 - start_deletion;
            }
        }
    } else {
        // legacy, pyx:190-315
        for (int c = 0; c < n; ++c) {                                                 // pyx:218-233
            const uint8_t ch = rf[c];
            if (ch == 'A' || ch == 'T' || ch == 'C' || ch == 'G' || ch == 'N') {
                L[C2_LIST_REF_POSITIONS].push(idx);
                if (rf[c] != rd[c] && rd[c] != '-' && rd[c] != 'N') {
                    L[C2_LIST_ALL_SUBSTITUTION_POSITIONS].push(idx); L[C2_LIST_ALL_SUBSTITUTION_VALUES].push(rd[c]);
                    if (c2_inc_has(inc, ni, idx)) { L[C2_LIST_SUBSTITUTION_POSITIONS].push(idx); L[C2_LIST_SUBSTITUTION_VALUES].push(rd[c]); }
                }
                idx++;
            } else {
                L[C2_LIST_REF_POSITIONS].push(idx == 0 ? -1 : -idx);
            }
        }
        // deletions: runs of '-' in the read, pyx:253-267
        for (int st = 0; st < n;) {
            if (rd[st] != '-') { ++st; continue; }
            int en = st; while (en < n && rd[en] == '-') ++en;
            int ref_st = 0;
            if (st - 1 > 0) ref_st = rp[st];
            int ref_en = idx - 1;
            if (en < n) ref_en = rp[en];
            for (int q = ref_st; q < ref_en; ++q) L[C2_LIST_ALL_DELETION_POSITIONS].push(q);
            L[C2_LIST_ALL_DELETION_COORDINATES].push(ref_st); L[C2_LIST_ALL_DELETION_COORDINATES].push(ref_en);
            if (c2_inc_hits(inc, ni, ref_st, ref_en)) {
                for (int q = ref_st; q < ref_en; ++q) L[C2_LIST_DELETION_POSITIONS].push(q);
                L[C2_LIST_DELETION_COORDINATES].push(ref_st); L[C2_LIST_DELETION_COORDINATES].push(ref_en);
                L[C2_LIST_DELETION_SIZES].push(en - st); del_n += en - st;
            }
            st = en;
        }
        // insertions: runs of '-' in the reference, pyx:271-288 (either flank in the window counts, pyx:284)
        for (int st = 0; st < n;) {
            if (rf[st] != '-') { ++st; continue; }
            int en = st; while (en < n && rf[en] == '-') ++en;
            if (st != 0 && en != n) {
                const int ref_st = rp[st - 1], ref_en = rp[en];
                L[C2_LIST_ALL_INSERTION_LEFT_POSITIONS].push(ref_st);
                L[C2_LIST_ALL_INSERTION_POSITIONS].push(ref_st); L[C2_LIST_ALL_INSERTION_POSITIONS].push(ref_en);
                if (c2_inc_has(inc, ni, ref_st) || c2_inc_has(inc, ni, ref_en)) {
                    L[C2_LIST_INSERTION_COORDINATES].push(ref_st); L[C2_LIST_INSERTION_COORDINATES].push(ref_en);
                    L[C2_LIST_INSERTION_POSITIONS].push(ref_st); L[C2_LIST_INSERTION_POSITIONS].push(ref_en);
                    L[C2_LIST_INSERTION_SIZES].push(en - st); ins_n += en - st;
                }
            }
            st = en;
        }
    }
}

__global__ __launch_bounds__(64) void c2_classify_lists_kernel(c2_classify_args A)
{
    if (threadIdx.x != 0) return;
    c2_list_writer L[C2_LIST_COUNT];
    for (int k = 0; k < C2_LIST_COUNT; ++k) { L[k].base = A.lists + (size_t)k * A.cap; L[k].cap = A.cap; L[k].n = 0; }
    int64_t ins_n, del_n;
    c2_classify_walk(A.read_al, A.ref_al, A.n, A.include_sorted, A.n_include, A.legacy, L, ins_n, del_n);   // cap >= n is guaranteed by the host
    for (int k = 0; k < C2_LIST_COUNT; ++k) A.list_len[k] = L[k].n;
    A.counts[0] = ins_n; A.counts[1] = del_n; A.counts[2] = L[C2_LIST_SUBSTITUTION_POSITIONS].n;
}

// Batched form: one LANE per alignment (the walk is serial in the column index; alignments are independent), two passes
// over the same walk -- pass 0 counts the 15 list lengths (reference positions go to a scratch row), the host turns the
// lengths into offsets, pass 1 writes every list at its place in one flat int32 array.  Used by
// crispresso2_amd.variants.get_new_variant_objects (one launch pair per 32 k alignments instead of one launch each).
__global__ __launch_bounds__(64) void c2_classify_lists_batch_kernel(c2_classify_batch_args A)
{
    const uint64_t t = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (t >= A.n) return;
    const int set = A.set_ids ? (int)A.set_ids[t] : 0;
    const int32_t* inc = A.include_sorted + A.include_off[set];
    const int ni = (int)(A.include_off[set + 1] - A.include_off[set]);
    const int n = A.lens[t];
    c2_list_writer L[C2_LIST_COUNT];
#pragma unroll
    for (int k = 0; k < C2_LIST_COUNT; ++k) {
        if (A.pass == 0) { L[k].base = nullptr; L[k].cap = 0; }
        else { L[k].base = A.values + A.list_off[t * C2_LIST_COUNT + k]; L[k].cap = A.list_len[t * C2_LIST_COUNT + k]; }
        L[k].n = 0;
    }
    if (A.pass == 0) { L[C2_LIST_REF_POSITIONS].base = A.scratch_rp + t * (uint64_t)A.stride; L[C2_LIST_REF_POSITIONS].cap = (int32_t)A.stride; }
    int64_t ins_n, del_n;
    c2_classify_walk(A.aln_read + t * (uint64_t)A.stride, A.aln_ref + t * (uint64_t)A.stride, n, inc, ni, A.legacy, L, ins_n, del_n);
    if (A.pass == 0) {
#pragma unroll
        for (int k = 0; k < C2_LIST_COUNT; ++k) A.list_len[t * C2_LIST_COUNT + k] = L[k].n;
        A.counts[t * 3] = ins_n; A.counts[t * 3 + 1] = del_n; A.counts[t * 3 + 2] = L[C2_LIST_SUBSTITUTION_POSITIONS].n;
    }
}

// get_consensus_alignment_from_pairs (CRISPRessoCORE.py:829-984) with get_greater_qual_nuc (:800-826): the two-pointer walk
// over the alignments of read 1 and read 2 against the same reference, statement for statement -- including that the
// double-gap column adds no quality character, that a read alone past the other's end copies its gaps, and that the
// strings are trimmed where the consensus reference starts or ends with '-'.  A quality index past the end of its string
// is an IndexError in the reference: flag 2, and the host raises.  One lane per pair; every lane walks its own rows, so the
// six input strings are read a dword at a time (c2_row_bytes: the walk's indices only ever step by one, a dword serves four
// of them) instead of one byte per load instruction.
struct c2_row_bytes {
    const uint32_t* w; uint32_t cur; int at;
    __device__ __forceinline__ c2_row_bytes(const uint8_t* row) : w((const uint32_t*)row), cur(0), at(-1) {}
    __device__ __forceinline__ uint8_t operator[](const int i) {
        const int q = i >> 2;
        if (q != at) { cur = w[q]; at = q; }
        return (uint8_t)(cur >> ((i & 3) * 8));
    }
};

__global__ __launch_bounds__(64) void c2_consensus_pairs_kernel(c2_consensus_args A)
{
    const uint64_t t = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (t >= A.n) return;
    // (rows start at multiples of 4: the host checks the strides and allocates the arrays)
    c2_row_bytes s1(A.s1 + t * A.stride), f1(A.f1 + t * A.stride), s2(A.s2 + t * A.stride), f2(A.f2 + t * A.stride);
    c2_row_bytes q1(A.q1 + t * A.qstride), q2(A.q2 + t * A.qstride);
    const int n1 = A.n1[t], n2 = A.n2[t], lq1 = A.lq1[t], lq2 = A.lq2[t];
    const bool best1 = A.best1[t] != 0;
    uint8_t* oa = A.o_aln + t * A.ostride; uint8_t* orf = A.o_ref + t * A.ostride; uint8_t* oq = A.o_qual + t * A.ostride;
    int start1 = 0; while (start1 < n1 && s1[start1] == '-') ++start1;            // len(s) - len(s.lstrip('-'))
    int start2 = 0; while (start2 < n2 && s2[start2] == '-') ++start2;
    int stop1 = n1 - 1; while (stop1 >= 0 && s1[stop1] == '-') --stop1;            // len(s.rstrip('-')) - 1
    int stop2 = n2 - 1; while (stop2 >= 0 && s2[stop2] == '-') --stop2;
    int i1 = 0, i2 = 0, qi1 = 0, qi2 = 0, na = 0, nq = 0;
    bool caching = true, index_error = false;
    auto Q1 = [&]() -> uint8_t { if (qi1 >= lq1) { index_error = true; return (uint8_t)'!'; } return q1[qi1]; };
    auto Q2 = [&]() -> uint8_t { if (qi2 >= lq2) { index_error = true; return (uint8_t)'!'; } return q2[qi2]; };
    auto greater = [&](uint8_t c1, uint8_t a, uint8_t c2, uint8_t b, uint8_t& nuc, uint8_t& q) {     // :800-826
        if (c1 == c2) { nuc = c1; q = a >= b ? a : b; return; }
        caching = false;
        if (a == b) { nuc = best1 ? c1 : c2; q = b; }
        else if (a > b) { nuc = c1; q = a; }
        else { nuc = c2; q = b; }
    };
    while ((i1 < n1 || i2 < n2) && !index_error) {
        const bool in1 = i1 < n1, in2 = i2 < n2;
        if (in1 && f1[i1] == '-' && in2 && f2[i2] == '-') {
            uint8_t nuc, q; const uint8_t a = Q1(), b = Q2();
            greater(s1[i1], a, s2[i2], b, nuc, q);
            oa[na] = nuc; orf[na] = '-'; ++na; oq[nq++] = q;
            ++qi1; ++qi2; ++i1; ++i2;
            continue;
        } else if (in1 && f1[i1] == '-') {
            oa[na] = s1[i1]; orf[na] = '-'; ++na; oq[nq++] = Q1();
            ++qi1; ++i1;
            continue;
        } else if (in2 && f2[i2] == '-') {
            oa[na] = s2[i2]; orf[na] = '-'; ++na; oq[nq++] = Q2();
            ++qi2; ++i2;
            continue;
        }
        if (in1 && s1[i1] == '-' && in2 && s2[i2] == '-') {
            oa[na] = ((start1 <= i1 && i1 <= stop1) || (start2 <= i2 && i2 <= stop2)) ? '-' : 'N';
            orf[na] = f1[i1]; ++na;
        } else if (in1 && s1[i1] == '-' && in2 && s2[i2] != '-') {
            oa[na] = s2[i2]; orf[na] = f2[i2]; ++na; oq[nq++] = Q2(); ++qi2;
        } else if (in1 && s1[i1] != '-' && in2 && s2[i2] == '-') {
            oa[na] = s1[i1]; orf[na] = f1[i1]; ++na; oq[nq++] = Q1(); ++qi1;
        } else if (in1 && in2) {
            uint8_t nuc, q; const uint8_t a = Q1(), b = Q2();
            greater(s1[i1], a, s2[i2], b, nuc, q);
            oa[na] = nuc; orf[na] = f1[i1]; ++na; oq[nq++] = q; ++qi1; ++qi2;
        } else if (in1) {
            oa[na] = (s1[i1] == '-' && start1 <= i1 && i1 <= stop1) ? (uint8_t)'N' : s1[i1];
            oq[nq++] = Q1(); orf[na] = f1[i1]; ++na; ++qi1;
        } else if (in2) {
            oa[na] = (s2[i2] == '-' && start2 <= i2 && i2 <= stop2) ? (uint8_t)'N' : s2[i2];
            oq[nq++] = Q2(); orf[na] = f2[i2]; ++na; ++qi2;
        }
        ++i1; ++i2;
    }
    // trim where the consensus reference starts / ends with '-' (quality string sliced by the same counts, :968-975)
    int lead = 0; while (lead < na && orf[lead] == '-') ++lead;
    int trail = 0; while (trail < na - lead && orf[na - 1 - trail] == '-') ++trail;
    if (lead >= na) index_error = true;                                            // final_ref[0] on an empty string
    const int len = index_error ? 0 : na - lead - trail;
    int qlen = nq - (lead < nq ? lead : nq);
    qlen -= (trail < qlen ? trail : qlen);
    int hom = 0;
    for (int k = 0; k < len; ++k) {
        const uint8_t a = oa[lead + k], r = orf[lead + k];
        oa[k] = a; orf[k] = r;
        hom += (a == r);
    }
    for (int k = 0; k < qlen; ++k) oq[k] = oq[(lead < nq ? lead : nq) + k];
    int32_t* info = A.o_info + t * 4;
    info[0] = len; info[1] = qlen; info[2] = hom; info[3] = (caching ? 1 : 0) | (index_error ? 2 : 0);
}

// The 32-byte record of a GIVEN pair of aligned strings -- the counters the fused classifier of the align kernels derives for the
// alignments it emits (find_indels_substitutions, COREResources.pyx:68-187, + CRISPRessoCORE.py:726-760), here for strings that come from
// somewhere else: the consensus alignment of a read pair (get_consensus_alignment_from_pairs, CRISPRessoCORE.py:829-984).  One wavefront per
// item, 64 columns per step, the same scan as c2_emit_and_classify (forward order, strings in global memory).  That scan relies on what
// the aligner guarantees -- no column with a gap in both strings, no insertion column next to a deletion column -- and a consensus of two
// alignments need not keep to it: such an item gets C2_STATUS_SHAPE and the caller takes the list-based classifier for the run.
__global__ __launch_bounds__(256) void c2_classify_records_kernel(c2_records_args A)
{
    const int lane = threadIdx.x & 63;
    const uint64_t t = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= A.n) return;                                                          // (wave-uniform)
    const uint8_t* R = A.aln_read + t * (uint64_t)A.stride;
    const uint8_t* F = A.aln_ref + t * (uint64_t)A.stride;
    const int T = A.info[t * 4 + 0], matches = A.info[t * 4 + 2];
    const int ref_id = A.ref_ids ? (int)A.ref_ids[t] : (int)(t % (uint64_t)A.n_refs);
    const uint16_t* sIncP = A.refs[ref_id].inc_prefix;
    c2_aln_record rec;
    {
        uint32_t* z = (uint32_t*)&rec;
        for (int k = 0; k < 8; ++k) z[k] = 0u;
    }
    rec.ref_id = (uint16_t)ref_id;
    rec.strand = A.strands ? A.strands[t] : (uint8_t)0;
    if (T <= 0 || T > 65535 || (A.info[t * 4 + 3] & 2)) { rec.status = C2_STATUS_EMPTY; if (lane == 0) A.records[t] = rec; return; }
    int idx_base = 0, last_rf = -1, last_rd = -1;
    int n_all_sub = 0, n_win_sub = 0, n_all_ins = 0, n_win_ins = 0, n_all_del = 0, n_win_del = 0;
    int acc_ins_n = 0, acc_del_n = 0, acc_del_bases = 0;
    bool shape = false;
    bool prev_ins_col = false, prev_del_col = false;                                // the column in front of this chunk
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = 0; base < T; base += 64) {
        const int cidx = base + lane;
        const bool in = cidx < T;
        const unsigned char rd = in ? R[cidx] : 0, rfc = in ? F[cidx] : 0;
        const bool rf_ng = in && rfc != '-', rd_ng = in && rd != '-';
        const unsigned long long m_rf = __ballot(rf_ng), m_rd = __ballot(rd_ng), m_in = __ballot(in);
        {   // shapes outside the scan's assumptions
            const unsigned long long ins_col = m_in & ~m_rf, del_col = m_in & ~m_rd;
            if (ins_col & del_col) shape = true;
            if ((ins_col & (del_col << 1)) | (del_col & (ins_col << 1))) shape = true;
            if (((ins_col & 1ull) && prev_del_col) || ((del_col & 1ull) && prev_ins_col)) shape = true;
            prev_ins_col = (ins_col >> 63) & 1ull; prev_del_col = (del_col >> 63) & 1ull;
        }
        const int idx = idx_base + __popcll(m_rf & lt);
        const unsigned long long below_rf = m_rf & lt, below_rd = m_rd & lt;
        const int prev_rf = below_rf ? base + 63 - __clzll((long long)below_rf) : last_rf;
        const int prev_rd = below_rd ? base + 63 - __clzll((long long)below_rd) : last_rd;
        const bool sub = rf_ng && rd_ng && rd != rfc && rd != 'N';                   // pyx:113-118
        const bool sub_win = sub && (sIncP[idx + 1] != sIncP[idx]);
        n_all_sub += __popcll(__ballot(sub));
        n_win_sub += __popcll(__ballot(sub_win));
        const bool ins_close = rf_ng && (prev_rf != cidx - 1) && idx > 0;            // pyx:119-128, :136
        const bool fl = ins_close && (sIncP[idx] != sIncP[idx - 1]), fr = ins_close && (sIncP[idx + 1] != sIncP[idx]);
        const bool ins_win = A.legacy ? (fl || fr) : (fl && fr);                     // pyx:121 / legacy pyx:284
        n_all_ins += __popcll(__ballot(ins_close));
        n_win_ins += __popcll(__ballot(ins_win));
        if (ins_win) acc_ins_n += cidx - 1 - prev_rf;
        const bool del_close = rd_ng && (prev_rd != cidx - 1);                       // pyx:145-153
        const int dlen = cidx - 1 - prev_rd;
        const int dstart = (A.legacy && prev_rd <= 0) ? 0 : idx - dlen;              // legacy pyx:253-258
        const bool del_win = del_close && (sIncP[idx] != sIncP[dstart]);
        n_all_del += __popcll(__ballot(del_close));
        n_win_del += __popcll(__ballot(del_win));
        if (del_close) acc_del_bases += idx - dstart;
        if (del_win) acc_del_n += dlen;
        idx_base += __popcll(m_rf);
        if (m_rf) last_rf = base + 63 - __clzll((long long)m_rf);
        if (m_rd) last_rd = base + 63 - __clzll((long long)m_rd);
    }
    int tr_bases = 0, tr_win = 0;
    if (last_rd != T - 1) {                                                          // trailing deletion, pyx:155-162
        const int dlen = T - 1 - last_rd;
        n_all_del += 1;
        if (!A.legacy) {
            tr_bases = dlen;
            if (idx_base - dlen >= 0 && sIncP[idx_base] != sIncP[idx_base - dlen]) { tr_win = dlen; n_win_del += 1; }
        } else {
            const int dstart = last_rd <= 0 ? 0 : idx_base - dlen, dend = idx_base - 1;   // legacy pyx:259-261
            tr_bases = dend > dstart ? dend - dstart : 0;
            if (dend > dstart && sIncP[dend] != sIncP[dstart]) { tr_win = dlen; n_win_del += 1; }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        acc_ins_n += __shfl_xor(acc_ins_n, m);
        acc_del_n += __shfl_xor(acc_del_n, m);
        acc_del_bases += __shfl_xor(acc_del_bases, m);
    }
    const unsigned char r0 = R[0], f0 = F[0], rL = R[T - 1], fL = F[T - 1];
    rec.irregular_ends = (r0 == '-' || f0 == '-' || r0 != f0 || rL == '-' || fL == '-' || rL != fL) ? 1 : 0;   // CRISPRessoCORE.py:1106-1110
    rec.aln_len = (uint16_t)T;
    rec.matches = (uint16_t)matches;
    rec.insertion_n = (uint16_t)acc_ins_n;
    rec.deletion_n = (uint16_t)(acc_del_n + tr_win);
    rec.substitution_n = (uint16_t)n_win_sub;
    rec.all_insertion_events = (uint16_t)n_all_ins;
    rec.win_insertion_events = (uint16_t)n_win_ins;
    rec.all_deletion_events = (uint16_t)n_all_del;
    rec.win_deletion_events = (uint16_t)n_win_del;
    rec.all_deletion_bases = (uint16_t)(acc_del_bases + tr_bases);
    rec.all_substitutions = (uint16_t)n_all_sub;
    rec.status = shape ? C2_STATUS_SHAPE : 0;
    if (lane == 0) A.records[t] = rec;
}

// calculate_homology, COREResources.pyx:318-327 (float32 accumulator; result = score / strlen(a))
__global__ __launch_bounds__(64) void c2_homology_kernel(const uint8_t* a, const uint8_t* b, int n, float* out)
{
    if (threadIdx.x != 0) return;
    float score = 0.0f;
    for (int k = 0; k < n; ++k) if (a[k] == b[k]) score += 1;
    *out = score / (float)n;
}
