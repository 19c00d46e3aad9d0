// c2_api_classify.hip -- host side of the C ABI declared in include/crispresso2_amd.h: the classifier with full position lists (per call and batched), the paired-read consensus, calculate_homology.
// Marshals the caller's inputs into the kernels' tables, owns the device buffers of a context, picks launch geometry and
// launches.  Nothing here computes an alignment or a classification on the CPU.
#include "c2_ctx.h"
#include "c2_k_classify.hip"

extern "C" {

int c2_find_indels_substitutions(c2_ctx* ctx, const char* read_aln, const char* ref_aln, int32_t n,
                                 const int32_t* include_idx, int32_t n_include, int32_t legacy,
                                 int32_t* out, int32_t out_cap, int32_t* out_index, int64_t* out_counts, int32_t* out_needed) {
    if (!ctx || !read_aln || !ref_aln || n < 0 || !out || !out_index || !out_counts) { if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> inc(include_idx, include_idx + (n_include > 0 ? n_include : 0));
    std::sort(inc.begin(), inc.end());
    inc.erase(std::unique(inc.begin(), inc.end()), inc.end());
    hipStream_t s = ctx->stream;
    int cap = std::max(2 * n + 8, 64);
    std::vector<int32_t> lens(C2_LIST_COUNT);
    std::vector<int32_t> lists;
    for (int attempt = 0; attempt < 3; ++attempt) {
        // d_misc layout: [read n][ref n][pad][include][list_len 15][counts 3 x int64][lists 15 x cap]
        size_t o_read = 0, o_ref = (size_t)n, o_inc = ((size_t)2 * n + 15) / 16 * 16;
        size_t o_len = o_inc + ((inc.size() * 4 + 15) / 16 * 16);
        size_t o_cnt = o_len + 64, o_lists = o_cnt + 32;
        size_t total = o_lists + (size_t)C2_LIST_COUNT * cap * 4;
        int rc;
        if ((rc = ensure(ctx, ctx->d_misc, total))) return rc;
        uint8_t* base = (uint8_t*)ctx->d_misc.p;
        if (n) {
            HIPCHK(ctx, hipMemcpyAsync(base + o_read, read_aln, n, hipMemcpyHostToDevice, s));
            HIPCHK(ctx, hipMemcpyAsync(base + o_ref, ref_aln, n, hipMemcpyHostToDevice, s));
        }
        if (!inc.empty()) HIPCHK(ctx, hipMemcpyAsync(base + o_inc, inc.data(), inc.size() * 4, hipMemcpyHostToDevice, s));
        c2_classify_args A;
        A.read_al = base + o_read; A.ref_al = base + o_ref; A.include_sorted = (const int32_t*)(base + o_inc);
        A.n = n; A.n_include = (int32_t)inc.size(); A.legacy = legacy ? 1 : 0; A.cap = cap;
        A.lists = (int32_t*)(base + o_lists); A.list_len = (int32_t*)(base + o_len); A.counts = (int64_t*)(base + o_cnt);
        hipLaunchKernelGGL(c2_classify_lists_kernel, dim3(1), dim3(64), 0, s, A);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(lens.data(), base + o_len, C2_LIST_COUNT * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(out_counts, base + o_cnt, 24, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        const int need = *std::max_element(lens.begin(), lens.end());
        if (need <= cap) {
            lists.resize((size_t)C2_LIST_COUNT * cap);
            HIPCHK(ctx, hipMemcpy(lists.data(), base + o_lists, lists.size() * 4, hipMemcpyDeviceToHost));
            break;
        }
        cap = need + 8;   // only the negative-coordinate quirk of the reference can get here
        if (attempt == 2) { ctx->err = "classification lists did not converge"; return C2_E_DEVICE; }
    }
    int64_t total = 0;
    for (int k = 0; k < C2_LIST_COUNT; ++k) total += lens[k];
    if (out_needed) *out_needed = (int32_t)total;
    if (total > out_cap) { ctx->err = "output buffer too small"; return C2_E_OVERFLOW; }
    int32_t pos = 0;
    for (int k = 0; k < C2_LIST_COUNT; ++k) {
        out_index[2 * k] = pos; out_index[2 * k + 1] = lens[k];
        if (lens[k]) memcpy(out + pos, lists.data() + (size_t)k * cap, (size_t)lens[k] * 4);
        pos += lens[k];
    }
    return 0;
}

struct c2_lists {
    std::vector<int64_t> index;    // n * C2_LIST_COUNT + 1 offsets into values
    std::vector<int32_t> values;
    std::vector<int64_t> counts;   // n x 3
};

int c2_classify_lists_batch(c2_ctx* ctx, uint64_t n, const uint8_t* aln_read, const uint8_t* aln_ref, uint32_t stride,
                            const int32_t* lens, const uint16_t* set_ids, const int32_t* include_idx, const int64_t* include_off,
                            int32_t n_sets, int32_t legacy, c2_lists** out) {
    if (!ctx || !out || (n && (!aln_read || !aln_ref || !lens)) || n_sets < 1 || !include_off || stride == 0) {
        if (ctx) ctx->err = "bad argument"; return C2_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the include sets, each sorted and unique (Python's `in` / set.intersection semantics; any integers)
    std::vector<int32_t> inc;
    std::vector<int64_t> inc_off(1, 0);
    for (int k = 0; k < n_sets; ++k) {
        std::vector<int32_t> one(include_idx + include_off[k], include_idx + include_off[k + 1]);
        std::sort(one.begin(), one.end());
        one.erase(std::unique(one.begin(), one.end()), one.end());
        inc.insert(inc.end(), one.begin(), one.end());
        inc_off.push_back((int64_t)inc.size());
    }
    for (uint64_t t = 0; t < n; ++t) {
        if (lens[t] < 0 || (uint32_t)lens[t] > stride) { ctx->err = "alignment longer than stride"; return C2_E_INVALID; }
        if (set_ids && set_ids[t] >= n_sets) { ctx->err = "include set id out of range"; return C2_E_INVALID; }
    }
    std::unique_ptr<c2_lists> R(new c2_lists);
    R->index.assign((size_t)n * C2_LIST_COUNT + 1, 0);
    R->counts.assign((size_t)n * 3, 0);
    hipStream_t s = ctx->stream;
    const uint64_t CH = 32768;
    std::vector<int32_t> llen;
    std::vector<int64_t> loff;
    int rc;
    for (uint64_t c0 = 0; c0 < n; c0 += CH) {
        const uint64_t m = std::min<uint64_t>(CH, n - c0);
        // d_lists layout: [read m*stride][ref m*stride][lens m][set ids m][include][include_off][scratch m*stride*4][len m*15][off m*15*8][counts m*3*8]
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t o = 0;
        const size_t o_rd = o; o += al(m * stride);
        const size_t o_rf = o; o += al(m * stride);
        const size_t o_ln = o; o += al(m * 4);
        const size_t o_id = o; o += al(m * 2);
        const size_t o_inc = o; o += al(inc.size() * 4 + 4);
        const size_t o_ioff = o; o += al(inc_off.size() * 8);
        const size_t o_rp = o; o += al(m * (size_t)stride * 4);
        const size_t o_len = o; o += al(m * C2_LIST_COUNT * 4);
        const size_t o_off = o; o += al(m * C2_LIST_COUNT * 8);
        const size_t o_cnt = o; o += al(m * 3 * 8);
        if ((rc = ensure(ctx, ctx->d_lists, o))) return rc;
        uint8_t* base = (uint8_t*)ctx->d_lists.p;
        HIPCHK(ctx, hipMemcpyAsync(base + o_rd, aln_read + c0 * stride, m * stride, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(base + o_rf, aln_ref + c0 * stride, m * stride, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(base + o_ln, lens + c0, m * 4, hipMemcpyHostToDevice, s));
        if (set_ids) HIPCHK(ctx, hipMemcpyAsync(base + o_id, set_ids + c0, m * 2, hipMemcpyHostToDevice, s));
        if (!inc.empty()) HIPCHK(ctx, hipMemcpyAsync(base + o_inc, inc.data(), inc.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(base + o_ioff, inc_off.data(), inc_off.size() * 8, hipMemcpyHostToDevice, s));
        c2_classify_batch_args A;
        A.aln_read = base + o_rd; A.aln_ref = base + o_rf; A.lens = (const int32_t*)(base + o_ln);
        A.set_ids = set_ids ? (const uint16_t*)(base + o_id) : nullptr;
        A.include_sorted = (const int32_t*)(base + o_inc); A.include_off = (const int64_t*)(base + o_ioff);
        A.n = m; A.stride = stride; A.legacy = legacy ? 1 : 0; A.pass = 0; A.reserved = 0;
        A.scratch_rp = (int32_t*)(base + o_rp); A.list_len = (int32_t*)(base + o_len); A.list_off = (const int64_t*)(base + o_off);
        A.values = nullptr; A.counts = (int64_t*)(base + o_cnt);
        const unsigned grid = (unsigned)((m + 63) / 64);
        hipLaunchKernelGGL(c2_classify_lists_batch_kernel, dim3(grid), dim3(64), 0, s, A);
        HIPCHK(ctx, hipGetLastError());
        llen.resize(m * C2_LIST_COUNT);
        HIPCHK(ctx, hipMemcpyAsync(llen.data(), base + o_len, llen.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(R->counts.data() + c0 * 3, base + o_cnt, m * 3 * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        loff.resize(llen.size());
        int64_t tot = 0;
        for (size_t k = 0; k < llen.size(); ++k) { loff[k] = tot; tot += llen[k]; }
        const int64_t g0 = (int64_t)R->values.size();
        for (size_t k = 0; k < llen.size(); ++k) R->index[c0 * C2_LIST_COUNT + k] = g0 + loff[k];
        R->values.resize((size_t)(g0 + tot));
        R->index[(c0 + m) * C2_LIST_COUNT] = g0 + tot;
        if (tot > 0) {
            if ((rc = ensure(ctx, ctx->d_lists_out, (size_t)tot * 4))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(base + o_off, loff.data(), loff.size() * 8, hipMemcpyHostToDevice, s));
            A.pass = 1; A.values = (int32_t*)ctx->d_lists_out.p;
            hipLaunchKernelGGL(c2_classify_lists_batch_kernel, dim3(grid), dim3(64), 0, s, A);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(R->values.data() + g0, ctx->d_lists_out.p, (size_t)tot * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(ctx, hipStreamSynchronize(s));
        }
    }
    *out = R.release();
    return 0;
}

uint64_t c2_lists_total(const c2_lists* r) { return r ? (uint64_t)r->values.size() : 0; }
const int64_t* c2_lists_index(const c2_lists* r) { return r ? r->index.data() : nullptr; }
const int32_t* c2_lists_values(const c2_lists* r) { return r ? r->values.data() : nullptr; }
const int64_t* c2_lists_counts(const c2_lists* r) { return r ? r->counts.data() : nullptr; }
void c2_lists_free(c2_lists* r) { delete r; }

// Launch only: every pointer is a device address.  The paired route keeps both reads' alignments on the device (BatchAligner.align_device)
// and hands their rows straight to this; only the qualities come from the host.  65,536 pairs per launch: every lane streams its own
// rows, and more lanes in flight thrash L2 (25.7 against 19 ns per pair at 262,144).
int c2_consensus_pairs_device(c2_ctx* ctx, uint64_t n, const uint8_t* d_s1, const uint8_t* d_f1, const uint8_t* d_s2, const uint8_t* d_f2,
                              uint32_t stride, const int32_t* d_n1, const int32_t* d_n2, const uint8_t* d_q1, const uint8_t* d_q2,
                              uint32_t qstride, const int32_t* d_lq1, const int32_t* d_lq2, const uint8_t* d_best1,
                              uint8_t* d_out_aln, uint8_t* d_out_ref, uint8_t* d_out_qual, uint32_t ostride, int32_t* d_out_info, void* hip_stream) {
    if (!ctx || (n && (!d_s1 || !d_f1 || !d_s2 || !d_f2 || !d_n1 || !d_n2 || !d_q1 || !d_q2 || !d_lq1 || !d_lq2 || !d_best1 || !d_out_aln || !d_out_ref ||
                       !d_out_qual || !d_out_info))) { if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID; }
    if (n == 0) return 0;
    if (stride == 0 || qstride == 0 || ostride < 2 * stride) { ctx->err = "ostride must be at least 2 * stride"; return C2_E_INVALID; }
    if ((stride & 3u) || (qstride & 3u)) { ctx->err = "stride and qstride must be multiples of 4"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const uint64_t CH = 65536;
    for (uint64_t c0 = 0; c0 < n; c0 += CH) {
        const uint64_t m = std::min<uint64_t>(CH, n - c0);
        c2_consensus_args A;
        A.s1 = d_s1 + c0 * stride; A.f1 = d_f1 + c0 * stride; A.s2 = d_s2 + c0 * stride; A.f2 = d_f2 + c0 * stride;
        A.q1 = d_q1 + c0 * qstride; A.q2 = d_q2 + c0 * qstride;
        A.n1 = d_n1 + c0; A.n2 = d_n2 + c0; A.lq1 = d_lq1 + c0; A.lq2 = d_lq2 + c0; A.best1 = d_best1 + c0;
        A.n = m; A.stride = stride; A.qstride = qstride; A.ostride = ostride; A.reserved = 0;
        A.o_aln = d_out_aln + c0 * ostride; A.o_ref = d_out_ref + c0 * ostride; A.o_qual = d_out_qual + c0 * ostride; A.o_info = d_out_info + c0 * 4;
        hipLaunchKernelGGL(c2_consensus_pairs_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, A);
        HIPCHK(ctx, hipGetLastError());
    }
    return 0;
}

// Host arrays.  Chunks of 65,536 pairs through pinned staging, three streams: while chunk c runs, chunk c + 1's six input streams are
// packed into a pinned block by a few host threads and copied in, and chunk c - 1's results come back -- only the bytes that were
// written: the lengths first (16 bytes per pair), then the three output arrays as 2-D copies of the widest row (rows are 2 * stride
// wide; a consensus is about as long as the longer of the two alignments).
int c2_consensus_pairs_batch(c2_ctx* ctx, uint64_t n, const uint8_t* s1, const uint8_t* f1, const uint8_t* s2, const uint8_t* f2,
                             uint32_t stride, const int32_t* n1, const int32_t* n2, const uint8_t* q1, const uint8_t* q2,
                             uint32_t qstride, const int32_t* lq1, const int32_t* lq2, const uint8_t* best1,
                             uint8_t* out_aln, uint8_t* out_ref, uint8_t* out_qual, uint32_t ostride, int32_t* out_info) {
    if (!ctx || (n && (!s1 || !f1 || !s2 || !f2 || !n1 || !n2 || !q1 || !q2 || !lq1 || !lq2 || !best1 || !out_aln || !out_ref || !out_qual || !out_info))) {
        if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID;
    }
    if (n == 0) return 0;
    if (stride == 0 || qstride == 0 || ostride < 2 * stride) { ctx->err = "ostride must be at least 2 * stride"; return C2_E_INVALID; }
    if ((stride & 3u) || (qstride & 3u)) { ctx->err = "stride and qstride must be multiples of 4"; return C2_E_INVALID; }
    for (uint64_t t = 0; t < n; ++t)
        if (n1[t] < 0 || n2[t] < 0 || (uint32_t)n1[t] > stride || (uint32_t)n2[t] > stride || lq1[t] < 0 || lq2[t] < 0 ||
            (uint32_t)lq1[t] > qstride || (uint32_t)lq2[t] > qstride) { ctx->err = "length exceeds stride"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->s_in) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_in[k], hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_done[k], hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_out[k], hipEventDisableTiming));
        }
    }
    hipStream_t s = ctx->stream;
    unsigned threads = std::thread::hardware_concurrency();
    if (const char* e = getenv("C2_HOST_THREADS")) threads = (unsigned)atoi(e);
    threads = std::max(1u, std::min(threads, 16u));
    const uint64_t CH = 65536;
    const uint64_t n_chunks = (n + CH - 1) / CH, mmax = std::min<uint64_t>(CH, n);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    // one chunk's input block (the same layout in pinned and device memory) and output arrays
    const size_t o_s1 = 0, o_f1 = o_s1 + al(mmax * stride), o_s2 = o_f1 + al(mmax * stride), o_f2 = o_s2 + al(mmax * stride);
    const size_t o_q1 = o_f2 + al(mmax * stride), o_q2 = o_q1 + al(mmax * qstride), o_n = o_q2 + al(mmax * qstride);
    const size_t o_b = o_n + al(mmax * 16), in_bytes = o_b + al(mmax);
    const size_t o_oa = 0, o_or = al(mmax * ostride), o_oq = 2 * al(mmax * ostride), o_info = 3 * al(mmax * ostride), out_bytes = o_info + al(mmax * 16);
    int rc;
    if ((rc = ensure(ctx, ctx->d_lists, 2 * in_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->d_lists_out, 2 * out_bytes))) return rc;
    for (int k = 0; k < 2; ++k) {
        if ((rc = ensure_pinned(ctx, ctx->pin_in[k], ctx->pin_in_cap[k], in_bytes))) return rc;
        if ((rc = ensure_pinned(ctx, ctx->pin_out[k], ctx->pin_out_cap[k], out_bytes))) return rc;
    }
    auto m_of = [&](uint64_t c) { return std::min<uint64_t>(CH, n - c * CH); };
    auto stage_in = [&](uint64_t c) -> int {
        const int k = (int)(c & 1);
        const uint64_t c0 = c * CH, m = m_of(c);
        if (c >= 2) HIPCHK(ctx, hipEventSynchronize(ctx->ev_in[k]));
        uint8_t* pi = (uint8_t*)ctx->pin_in[k];
        copy_parallel(pi + o_s1, s1 + c0 * stride, m * stride, threads);
        copy_parallel(pi + o_f1, f1 + c0 * stride, m * stride, threads);
        copy_parallel(pi + o_s2, s2 + c0 * stride, m * stride, threads);
        copy_parallel(pi + o_f2, f2 + c0 * stride, m * stride, threads);
        copy_parallel(pi + o_q1, q1 + c0 * qstride, m * qstride, threads);
        copy_parallel(pi + o_q2, q2 + c0 * qstride, m * qstride, threads);
        memcpy(pi + o_n, n1 + c0, m * 4); memcpy(pi + o_n + mmax * 4, n2 + c0, m * 4);
        memcpy(pi + o_n + mmax * 8, lq1 + c0, m * 4); memcpy(pi + o_n + mmax * 12, lq2 + c0, m * 4);
        memcpy(pi + o_b, best1 + c0, m);
        HIPCHK(ctx, hipMemcpyAsync((uint8_t*)ctx->d_lists.p + k * in_bytes, pi, in_bytes, hipMemcpyHostToDevice, ctx->s_in));
        HIPCHK(ctx, hipEventRecord(ctx->ev_in[k], ctx->s_in));
        return 0;
    };
    // results of chunk c: lengths, then the written part of the rows, then to the caller's arrays
    auto fetch_out = [&](uint64_t c) -> int {
        const int k = (int)(c & 1);
        const uint64_t c0 = c * CH, m = m_of(c);
        uint8_t* dout = (uint8_t*)ctx->d_lists_out.p + k * out_bytes;
        uint8_t* po = (uint8_t*)ctx->pin_out[k];
        HIPCHK(ctx, hipStreamWaitEvent(ctx->s_out, ctx->ev_done[k], 0));
        HIPCHK(ctx, hipMemcpyAsync(po + o_info, dout + o_info, m * 16, hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipStreamSynchronize(ctx->s_out));
        const int32_t* info = (const int32_t*)(po + o_info);
        uint32_t w = 4;
        for (uint64_t t = 0; t < m; ++t) w = std::max<uint32_t>(w, (uint32_t)std::max(info[4 * t], info[4 * t + 1]));
        w = std::min<uint32_t>((w + 15u) & ~15u, ostride);
        HIPCHK(ctx, hipMemcpy2DAsync(po + o_oa, w, dout + o_oa, ostride, w, m, hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipMemcpy2DAsync(po + o_or, w, dout + o_or, ostride, w, m, hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipMemcpy2DAsync(po + o_oq, w, dout + o_oq, ostride, w, m, hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipStreamSynchronize(ctx->s_out));
        memcpy(out_info + c0 * 4, info, m * 16);
        auto rows = [&](uint8_t* dst, const uint8_t* src) {
            auto part = [=](uint64_t a, uint64_t z) { for (uint64_t t = a; t < z; ++t) memcpy(dst + (c0 + t) * ostride, src + t * w, w); };
            if (threads < 2 || m < 4096) { part(0, m); return; }
            std::vector<std::thread> pool;
            for (unsigned q = 0; q < threads; ++q) pool.emplace_back(part, m * q / threads, m * (q + 1) / threads);
            for (auto& th : pool) th.join();
        };
        rows(out_aln, po + o_oa); rows(out_ref, po + o_or); rows(out_qual, po + o_oq);
        return 0;
    };
    if ((rc = stage_in(0))) return rc;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const int k = (int)(c & 1);
        const uint64_t m = m_of(c);
        uint8_t* din = (uint8_t*)ctx->d_lists.p + k * in_bytes;
        uint8_t* dout = (uint8_t*)ctx->d_lists_out.p + k * out_bytes;
        HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_in[k], 0));
        const int32_t* dn = (const int32_t*)(din + o_n);
        if ((rc = c2_consensus_pairs_device(ctx, m, din + o_s1, din + o_f1, din + o_s2, din + o_f2, stride, dn, dn + mmax, din + o_q1, din + o_q2, qstride,
                                            dn + 2 * mmax, dn + 3 * mmax, din + o_b, dout + o_oa, dout + o_or, dout + o_oq, ostride,
                                            (int32_t*)(dout + o_info), (void*)s))) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_done[k], s));
        if (c + 1 < n_chunks && (rc = stage_in(c + 1))) return rc;          // the next chunk travels while this one runs
        if (c >= 1 && (rc = fetch_out(c - 1))) return rc;                   // ... and the previous one's results come back
    }
    if ((rc = fetch_out(n_chunks - 1))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(s));
    return 0;
}

int c2_classify_records_device(c2_ctx* ctx, uint64_t n, const uint8_t* d_aln_read, const uint8_t* d_aln_ref, uint32_t stride, const int32_t* d_info,
                               const uint16_t* d_ref_ids, const uint8_t* d_strands, int32_t legacy, c2_aln_record* d_records, void* hip_stream) {
    if (!ctx || (n && (!d_aln_read || !d_aln_ref || !d_info || !d_records))) { if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID; }
    if (n == 0) return 0;
    if (ctx->n_refs <= 0) { ctx->err = "references must be set first"; return C2_E_STATE; }
    if (!d_ref_ids && (n % (uint64_t)ctx->n_refs)) { ctx->err = "all-references layout: n must be a multiple of the number of references"; return C2_E_INVALID; }
    if (n > 0x7fffffffull * 4ull) { ctx->err = "too many items for one launch"; return C2_E_TOO_LARGE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_records_args A;
    A.aln_read = d_aln_read; A.aln_ref = d_aln_ref; A.info = d_info; A.ref_ids = d_ref_ids; A.strands = d_strands; A.refs = (const c2_dev_ref*)ctx->d_refdesc.p;
    A.records = d_records; A.n = n; A.stride = stride; A.n_refs = ctx->n_refs; A.legacy = legacy ? 1 : 0; A.reserved = 0;
    hipLaunchKernelGGL(c2_classify_records_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_calculate_homology(c2_ctx* ctx, const char* a, const char* b, int32_t n, double* out) {
    if (!ctx || !a || !b || !out || n < 0) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->d_misc, (size_t)2 * n + 64))) return rc;
    uint8_t* base = (uint8_t*)ctx->d_misc.p;
    const size_t o_out = ((size_t)2 * n + 15) / 16 * 16;
    if ((rc = ensure(ctx, ctx->d_misc, o_out + 16))) return rc;
    base = (uint8_t*)ctx->d_misc.p;
    hipStream_t s = ctx->stream;
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(base, a, n, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(base + n, b, n, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(c2_homology_kernel, dim3(1), dim3(64), 0, s, base, base + n, n, (float*)(base + o_out));
    HIPCHK(ctx, hipGetLastError());
    float f = 0;
    HIPCHK(ctx, hipMemcpyAsync(&f, base + o_out, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    *out = (double)f;
    return 0;
}

}  // extern "C"
