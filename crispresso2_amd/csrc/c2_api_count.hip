// c2_api_count.hip -- host side of the C ABI declared in include/crispresso2_amd.h: the seed test, strand / best-reference selection, the count tensor and its RCCL all-reduce.
// Marshals the caller's inputs into the kernels' tables, owns the device buffers of a context, picks launch geometry and
// launches.  Nothing here computes an alignment or a classification on the CPU.
#include "c2_ctx.h"
#include "c2_k_select.hip"
#include "c2_k_count.hip"

extern "C" {

int c2_count_vectors_device(c2_ctx* ctx, uint64_t n_tasks, const uint8_t* d_aln_read, const uint8_t* d_aln_ref,
                            uint32_t aln_stride, const c2_aln_record* d_records, const uint32_t* d_weights,
                            const uint16_t* h_min_matches, int32_t max_t, int32_t flags, int32_t hl,
                            int64_t* d_counts, void* hip_stream) {
    return c2_count_vectors_hinted_device(ctx, n_tasks, d_aln_read, d_aln_ref, aln_stride, d_records, d_weights, nullptr, h_min_matches, max_t, flags, hl, d_counts, hip_stream);
}

int c2_count_vectors_hinted_device(c2_ctx* ctx, uint64_t n_tasks, const uint8_t* d_aln_read, const uint8_t* d_aln_ref,
                                   uint32_t aln_stride, const c2_aln_record* d_records, const uint32_t* d_weights, const uint32_t* d_hints,
                                   const uint16_t* h_min_matches, int32_t max_t, int32_t flags, int32_t hl,
                                   int64_t* d_counts, void* hip_stream) {
    if (!ctx || !d_aln_read || !d_aln_ref || !d_records || !d_counts) { if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID; }
    if (ctx->n_refs <= 0) { ctx->err = "references must be set first"; return C2_E_STATE; }
    static_assert(C2_CNT_VECTORS == C2_COUNT_VECTORS && C2_CNT_SCALARS == C2_COUNT_SCALARS && C2_CNT_HISTS == C2_COUNT_HISTS, "count layout");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (n_tasks == 0) return 0;
    const int lmax = ctx->max_li;
    if (hl < lmax + 2) { ctx->err = "hl too small"; return C2_E_INVALID; }
    const size_t per_ref = (size_t)C2_CNT_VECTORS * (lmax + 1) + C2_CNT_SCALARS + (size_t)C2_CNT_HISTS * hl;
    size_t lds = c2_count_lds_bytes(per_ref, lmax);
    // the int32 accumulator block of a workgroup normally lives in LDS; amplicons beyond ~1,650 bp (with 250-bp reads) take the variant
    // that keeps it in HBM scratch (LDS then only holds the O(lmax) parts)
    const bool hbm_block = lds > 163840 || getenv("C2_COUNT_HBM_BLOCK");
    if (hbm_block) {
        lds = c2_count_lds_bytes_hbm(lmax);
        if (lds > 163840) { ctx->err = "count route: reference of " + std::to_string(lmax) + " bp needs " + std::to_string(lds) + " bytes of LDS"; return C2_E_TOO_LARGE; }
    }
    int rc;
    c2_count_args A;
    A.min_matches = nullptr;
    size_t o_tbl = 64;
    const size_t tbl_bytes = h_min_matches ? (size_t)ctx->n_refs * (size_t)(max_t + 1) * sizeof(uint16_t) : 0;
    if (o_tbl + tbl_bytes > ctx->d_cnt.cap) ctx->cnt_table.clear();
    if ((rc = ensure(ctx, ctx->d_cnt, o_tbl + tbl_bytes))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_cnt.p, 0, 64, s));
    if (h_min_matches) {
        const size_t nel = tbl_bytes / sizeof(uint16_t);
        if (ctx->cnt_table.size() != nel || memcmp(ctx->cnt_table.data(), h_min_matches, tbl_bytes) != 0) {
            HIPCHK(ctx, hipStreamSynchronize(s));      // no earlier launch may still read the old table
            HIPCHK(ctx, hipMemcpy((uint8_t*)ctx->d_cnt.p + o_tbl, h_min_matches, tbl_bytes, hipMemcpyHostToDevice));
            ctx->cnt_table.assign(h_min_matches, h_min_matches + nel);
        }
        A.min_matches = (const uint16_t*)((uint8_t*)ctx->d_cnt.p + o_tbl);
    }
    A.aln_read = d_aln_read; A.aln_ref = d_aln_ref; A.records = d_records; A.weights = d_weights;
    A.refs = (const c2_dev_ref*)ctx->d_refdesc.p; A.counts = (long long*)d_counts;
    A.work_counter = (unsigned long long*)ctx->d_cnt.p;
    A.n_tasks = n_tasks; A.aln_stride = aln_stride; A.n_refs = ctx->n_refs; A.lmax = lmax; A.hl = hl; A.max_t = max_t; A.flags = flags;
    A.order = nullptr;
    // the batch's hint words (c2_batch.diag_hints): one reference only, and its position vectors must fit the hinted kernel's LDS block
    A.hints = nullptr; A.rest_list = nullptr; A.rest_count = nullptr; A.n_tasks_dev = nullptr; A.ref_ends = nullptr; A.hint_gx = 0;
    const bool hints_usable = d_hints && c2_count_hinted_lds_bytes(lmax, hl) <= 65536 && !getenv("C2_NO_COUNT_HINTS");
    if (hints_usable && ctx->n_refs == 1) {
        A.hints = d_hints;
        // the tasks the hinted kernel leaves go to a list (in the buffer the grouping by reference uses: one reference here), the column walk runs over it
        if (n_tasks < 0xFFFFFFFFull && !getenv("C2_NO_COUNT_REST_LIST")) {
            if ((rc = ensure(ctx, ctx->d_order, 256 + n_tasks * sizeof(uint32_t)))) return rc;
            A.rest_count = (uint32_t*)ctx->d_cnt.p + 4;              // (bytes 16 .. 19 of the header zeroed above)
            A.rest_list = (uint32_t*)((uint8_t*)ctx->d_order.p + 256);
        }
        const size_t hl_lds = c2_count_hinted_lds_bytes(lmax, hl);
        const unsigned hgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_tasks + 255) / 256, (uint64_t)ctx->prop.multiProcessorCount * 4u));
        hipLaunchKernelGGL(c2_count_hinted_kernel, dim3(hgrid), dim3(256), hl_lds, s, A);
        HIPCHK(ctx, hipGetLastError());
        if (A.rest_list) { A.order = A.rest_list; A.n_tasks_dev = A.rest_count; A.hints = nullptr; }
    }
    if ((flags & C2_CNT_FLAG_ALL_REFS_LAYOUT) && (ctx->n_refs <= 1 || n_tasks % (uint64_t)ctx->n_refs != 0)) A.flags = flags & ~C2_CNT_FLAG_ALL_REFS_LAYOUT;
    if (ctx->n_refs > 1 && n_tasks < 0xFFFFFFFFull && !(A.flags & C2_CNT_FLAG_ALL_REFS_LAYOUT)) {
        // group the tasks by reference on the device (see c2_ref_histogram_kernel)
        const size_t hist_bytes = ((size_t)ctx->n_refs * 4 + 255) / 256 * 256;
        if ((rc = ensure(ctx, ctx->d_order, hist_bytes + n_tasks * sizeof(uint32_t)))) return rc;
        uint32_t* hist = (uint32_t*)ctx->d_order.p;
        uint32_t* order = (uint32_t*)((uint8_t*)ctx->d_order.p + hist_bytes);
        HIPCHK(ctx, hipMemsetAsync(hist, 0, hist_bytes, s));
        const unsigned gb = (unsigned)((n_tasks + 255) / 256);
        if (ctx->n_refs <= C2_REF_LDS_MAX && !getenv("C2_NO_REF_LDS_GROUPING")) {
            const unsigned gc = (unsigned)((n_tasks + C2_REF_CHUNK - 1) / C2_REF_CHUNK);
            const size_t lb = (size_t)ctx->n_refs * sizeof(uint32_t);
            hipLaunchKernelGGL(c2_ref_histogram_lds_kernel, dim3(gc), dim3(256), lb, s, d_records, n_tasks, hist, ctx->n_refs);
            hipLaunchKernelGGL(c2_ref_scan_kernel, dim3(1), dim3(64), 0, s, hist, ctx->n_refs);
            hipLaunchKernelGGL(c2_ref_scatter_lds_kernel, dim3(gc), dim3(256), lb, s, d_records, n_tasks, hist, order, ctx->n_refs);
        } else {
        hipLaunchKernelGGL(c2_ref_histogram_kernel, dim3(gb), dim3(256), 0, s, d_records, n_tasks, hist);
        hipLaunchKernelGGL(c2_ref_scan_kernel, dim3(1), dim3(64), 0, s, hist, ctx->n_refs);
        hipLaunchKernelGGL(c2_ref_scatter_kernel, dim3(gb), dim3(256), 0, s, d_records, n_tasks, hist, order);
        }
        HIPCHK(ctx, hipGetLastError());
        A.order = order;
        if (hints_usable) {
            // several references, every task against its own (CRISPRessoPooled's shape): the hinted kernel runs per reference over that reference's range
            // of the grouped order (after the scatter, hist[r] is the position behind reference r's last task); the column walk skips what it took
            A.hints = d_hints; A.ref_ends = hist;
            const uint64_t per = n_tasks / (uint64_t)ctx->n_refs + 1;
            A.hint_gx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((per + 4095) / 4096, 64));
            // ... and walks only what the hinted kernel left: a list per reference (in the reference's own range of a second buffer), closed up into `order`
            uint32_t* cnt = nullptr; uint32_t* pre = nullptr; uint32_t* total = nullptr; uint32_t* ranged = nullptr;
            if (!getenv("C2_NO_COUNT_REST_LIST")) {
                if ((rc = ensure(ctx, ctx->d_order2, 2 * hist_bytes + 256 + n_tasks * sizeof(uint32_t)))) return rc;
                cnt = (uint32_t*)ctx->d_order2.p; pre = (uint32_t*)((uint8_t*)ctx->d_order2.p + hist_bytes);
                total = (uint32_t*)((uint8_t*)ctx->d_order2.p + 2 * hist_bytes); ranged = (uint32_t*)((uint8_t*)ctx->d_order2.p + 2 * hist_bytes + 256);
                HIPCHK(ctx, hipMemsetAsync(cnt, 0, 2 * hist_bytes + 256, s));
                A.rest_list = ranged; A.rest_count = cnt;
            }
            hipLaunchKernelGGL(c2_count_hinted_kernel, dim3(A.hint_gx * (unsigned)ctx->n_refs), dim3(256), c2_count_hinted_lds_bytes(lmax, hl), s, A);
            HIPCHK(ctx, hipGetLastError());
            if (A.rest_list) {
                hipLaunchKernelGGL(c2_rest_scan_kernel, dim3(1), dim3(64), 0, s, cnt, pre, total, ctx->n_refs);
                const unsigned cgx = 8;
                hipLaunchKernelGGL(c2_rest_compact_kernel, dim3(cgx * (unsigned)ctx->n_refs), dim3(256), 0, s, cnt, pre, hist, 0u, ranged, order, cgx);
                HIPCHK(ctx, hipGetLastError());
                A.n_tasks_dev = total; A.hints = nullptr; A.ref_ends = nullptr;      // (A.order = order: now the tasks left, still grouped by reference)
            }
        }
    }
    if (hints_usable && ctx->n_refs > 1 && (A.flags & C2_CNT_FLAG_ALL_REFS_LAYOUT) && n_tasks < 0xFFFFFFFFull && !getenv("C2_NO_COUNT_REST_LIST")) {
        // every read against every reference (task = read * n_refs + reference; the weights are the selection's: most tasks have none): the hinted kernel per
        // reference over that reference's tasks by arithmetic, what it leaves in a list per reference, closed up into the list the column walk runs over
        const size_t hist_bytes = ((size_t)ctx->n_refs * 4 + 255) / 256 * 256;
        if ((rc = ensure(ctx, ctx->d_order, hist_bytes + n_tasks * sizeof(uint32_t)))) return rc;
        if ((rc = ensure(ctx, ctx->d_order2, 2 * hist_bytes + 256 + n_tasks * sizeof(uint32_t)))) return rc;
        uint32_t* order = (uint32_t*)((uint8_t*)ctx->d_order.p + hist_bytes);
        uint32_t* cnt = (uint32_t*)ctx->d_order2.p; uint32_t* pre = (uint32_t*)((uint8_t*)ctx->d_order2.p + hist_bytes);
        uint32_t* total = (uint32_t*)((uint8_t*)ctx->d_order2.p + 2 * hist_bytes); uint32_t* ranged = (uint32_t*)((uint8_t*)ctx->d_order2.p + 2 * hist_bytes + 256);
        HIPCHK(ctx, hipMemsetAsync(cnt, 0, 2 * hist_bytes + 256, s));
        const uint64_t per = n_tasks / (uint64_t)ctx->n_refs;
        A.hints = d_hints; A.rest_list = ranged; A.rest_count = cnt;
        A.hint_gx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((per + 4095) / 4096, (uint64_t)ctx->prop.multiProcessorCount * 4u / (uint64_t)ctx->n_refs + 1));
        hipLaunchKernelGGL(c2_count_hinted_kernel, dim3(A.hint_gx * (unsigned)ctx->n_refs), dim3(256), c2_count_hinted_lds_bytes(lmax, hl), s, A);
        hipLaunchKernelGGL(c2_rest_scan_kernel, dim3(1), dim3(64), 0, s, cnt, pre, total, ctx->n_refs);
        const unsigned cgx = 64;
        hipLaunchKernelGGL(c2_rest_compact_kernel, dim3(cgx * (unsigned)ctx->n_refs), dim3(256), 0, s, cnt, pre, (const uint32_t*)nullptr, (uint32_t)per, ranged, order, cgx);
        HIPCHK(ctx, hipGetLastError());
        A.order = order; A.n_tasks_dev = total; A.hints = nullptr; A.rest_list = nullptr; A.rest_count = nullptr;
    }
    const void* fn = hbm_block ? (const void*)c2_count_vectors_hbm_kernel : (const void*)c2_count_vectors_kernel;
    HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    int nb = 1;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * C2_CNT_WAVES, lds));
    if (nb < 1) nb = 1;
    uint64_t resident = (uint64_t)ctx->prop.multiProcessorCount * (uint64_t)nb;
    A.block_scratch = nullptr; A.block_ints = 0;
    if (hbm_block) {
        A.block_ints = (per_ref + 63) / 64 * 64;
        resident = std::max<uint64_t>(1, std::min<uint64_t>(resident, ((uint64_t)2 << 30) / (A.block_ints * sizeof(int))));     // at most 2 GiB of blocks
        if ((rc = ensure(ctx, ctx->d_cnt_block, (size_t)(resident * A.block_ints * sizeof(int))))) return rc;
        A.block_scratch = (int32_t*)ctx->d_cnt_block.p;
    }
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_tasks + 31) / 32, resident));
    if (hbm_block) hipLaunchKernelGGL(c2_count_vectors_hbm_kernel, dim3(grid), dim3(64 * C2_CNT_WAVES), lds, s, A);
    else           hipLaunchKernelGGL(c2_count_vectors_kernel, dim3(grid), dim3(64 * C2_CNT_WAVES), lds, s, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// ---- multi-GPU: the only exchange step of the sharded path, SURVEY 8(e): all-reduce of the per-amplicon count tensor over RCCL ----
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy that is already in the process (torch.distributed's), else the ROCm installation's
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.h) { api.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) api.err = "librccl.so lacks an expected symbol";
    });
    return &api;
}
#define RCCLCHK(ctx, api, call)                                                                             \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) {                                                                            \
            (ctx)->err = std::string(#call) + ": " + ((api)->GetErrorString ? (api)->GetErrorString(r_) : "rccl error"); \
            return C2_E_DEVICE;                                                                             \
        }                                                                                                   \
    } while (0)
}  // namespace

int c2_comm_unique_id(uint8_t* out_id) {
    RcclApi* R = rccl();
    if (!out_id || !R->err.empty()) { g_create_error = R->err.empty() ? "out_id is NULL" : R->err; return C2_E_DEVICE; }
    static_assert(sizeof(ncclUniqueId) == C2_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return C2_E_DEVICE; }
    memcpy(out_id, &id, sizeof id);
    return 0;
}

int c2_comm_init(c2_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { if (ctx) ctx->err = "bad communicator arguments"; return C2_E_INVALID; }
    RcclApi* R = rccl();
    if (!R->err.empty()) { ctx->err = R->err; return C2_E_DEVICE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { (void)R->CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    RCCLCHK(ctx, R, R->CommInitRank(&ctx->comm, world, uid, rank));
    ctx->comm_world = world;
    return 0;
}

int c2_reduce_counts(c2_ctx* ctx, int64_t* d_counts, uint64_t n_elements, void* hip_stream) {
    if (!ctx || (!d_counts && n_elements)) { if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID; }
    if (!ctx->comm) { ctx->err = "c2_comm_init has not been called"; return C2_E_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_elements == 0) return 0;
    RcclApi* R = rccl();
    RCCLCHK(ctx, R, R->AllReduce(d_counts, d_counts, (size_t)n_elements, ncclInt64, ncclSum, ctx->comm, (hipStream_t)hip_stream));
    return 0;
}

int c2_comm_destroy(c2_ctx* ctx) {
    if (!ctx) return C2_E_INVALID;
    if (ctx->comm) { RcclApi* R = rccl(); if (R->CommDestroy) (void)R->CommDestroy(ctx->comm); ctx->comm = nullptr; ctx->comm_world = 0; }
    return 0;
}

int c2_select_best_device(c2_ctx* ctx, uint64_t n_reads, int32_t n_refs, const c2_aln_record* d_records,
                          const c2_aln_record* d_records2, const int32_t* d_slot2, const uint32_t* h_min_mscore,
                          const uint32_t* d_raw_counts, const uint32_t* d_counts, int32_t mode, int32_t max_aln_len,
                          uint64_t* d_member, uint64_t* d_use2, uint8_t* d_flags, uint32_t* d_weights, uint32_t* d_weights2,
                          uint64_t* d_stats, void* hip_stream) {
    if (!ctx || !d_records || !h_min_mscore || n_refs <= 0 || n_refs > 32767 || mode < 0 || mode > 2) { if (ctx) ctx->err = "bad selection arguments"; return C2_E_INVALID; }
    if ((d_records2 != nullptr) != (d_slot2 != nullptr)) { ctx->err = "d_records2 and d_slot2 go together"; return C2_E_INVALID; }
    static_assert(C2_SEL_STATS == C2_SELECT_STATS, "selection statistics");
    // the integer form of round(100*matches/len, 3) is exact below 8000 columns (c2_mscore)
    if (max_aln_len >= 8000) { ctx->err = "c2_select_best_device: alignments of 8000 columns or more"; return C2_E_TOO_LARGE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_reads == 0) return 0;
    hipStream_t s = (hipStream_t)hip_stream;
    int rc;
    // thresholds: a small device table, re-uploaded when it changes
    if ((size_t)n_refs * sizeof(uint32_t) > ctx->d_sel.cap) { HIPCHK(ctx, hipDeviceSynchronize()); ctx->sel_table.clear(); }      // (the table moves: nothing may still read the old one)
    if ((rc = ensure(ctx, ctx->d_sel, (size_t)std::max(n_refs, 64) * sizeof(uint32_t)))) return rc;
    if (ctx->sel_table.size() != (size_t)n_refs || memcmp(ctx->sel_table.data(), h_min_mscore, (size_t)n_refs * 4) != 0) {
        HIPCHK(ctx, hipStreamSynchronize(s));          // no earlier launch may still read the old table
        HIPCHK(ctx, hipMemcpy(ctx->d_sel.p, h_min_mscore, (size_t)n_refs * 4, hipMemcpyHostToDevice));
        ctx->sel_table.assign(h_min_mscore, h_min_mscore + n_refs);
    }
    c2_select_args A;
    A.records = d_records; A.records2 = d_records2; A.slot2 = d_slot2; A.min_mscore = (const uint32_t*)ctx->d_sel.p;
    A.raw_counts = d_raw_counts; A.counts = d_counts;
    A.member = (unsigned long long*)d_member; A.use2 = (unsigned long long*)d_use2; A.flags = d_flags;
    A.weights = d_weights; A.weights2 = d_weights2; A.stats = (unsigned long long*)d_stats;
    A.n_reads = n_reads; A.n_refs = n_refs; A.mode = mode;
    hipLaunchKernelGGL(c2_select_best_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), C2_SEL_STATS * sizeof(unsigned long long), s, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_strand_plan_device(c2_ctx* ctx, uint64_t n_reads, const uint8_t* d_reads, const uint64_t* d_offsets, int32_t max_read_len,
                          int32_t n_refs, int32_t max_seeds, const int32_t* h_n_seeds, const uint8_t* h_seed_blob, int32_t blob_bytes,
                          const int32_t* h_seed_off, const int32_t* h_seed_len, int32_t seed_min, uint8_t* d_plan, void* hip_stream) {
    if (!ctx || !d_reads || !d_offsets || !d_plan || n_refs <= 0 || max_seeds < 0 || blob_bytes < 0 || max_read_len < 0 ||
        (max_seeds && (!h_n_seeds || !h_seed_off || !h_seed_len)) || (blob_bytes && !h_seed_blob)) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    if (n_reads == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t tbl = (size_t)n_refs * 2 * (size_t)std::max(max_seeds, 1);
    for (int r = 0; r < n_refs; ++r) if (h_n_seeds && (h_n_seeds[r] < 0 || h_n_seeds[r] > max_seeds)) { ctx->err = "n_seeds out of range"; return C2_E_INVALID; }
    for (size_t q = 0; q < (max_seeds ? tbl : 0); ++q)
        if (h_seed_len[q] < 0 || h_seed_off[q] < 0 || (int64_t)h_seed_off[q] + h_seed_len[q] > blob_bytes) { ctx->err = "seed outside the blob"; return C2_E_INVALID; }
    uint32_t lds = 4u * c2_strand_row_bytes(max_read_len);
    if (lds > 163840u) { ctx->err = "read longer than the strand-plan kernel's LDS row"; return C2_E_TOO_LARGE; }
    // seeds of at most C2_SEED_SLOT bytes whose table fits behind the read rows are compared from LDS, four bytes at a time
    bool seed_table = max_seeds > 0;
    for (size_t q = 0; q < (max_seeds ? tbl : 0) && seed_table; ++q) if ((uint32_t)h_seed_len[q] > C2_SEED_SLOT) seed_table = false;
    if (seed_table && (uint64_t)lds + (uint64_t)tbl * C2_SEED_SLOT > 65536u) seed_table = false;
    if (getenv("C2_STRAND_PLAN_BYTEWISE")) seed_table = false;
    if (seed_table) lds += (uint32_t)tbl * C2_SEED_SLOT;
    // one staging block: blob | seed_off | seed_len | n_seeds  (host copy first: a single small upload)
    const size_t o_off = ((size_t)blob_bytes + 15) & ~(size_t)15, o_len = o_off + tbl * 4, o_n = o_len + tbl * 4, total = o_n + (size_t)n_refs * 4;
    std::vector<uint8_t> host(total, 0);
    if (blob_bytes) memcpy(host.data(), h_seed_blob, (size_t)blob_bytes);
    if (max_seeds) { memcpy(host.data() + o_off, h_seed_off, tbl * 4); memcpy(host.data() + o_len, h_seed_len, tbl * 4); memcpy(host.data() + o_n, h_n_seeds, (size_t)n_refs * 4); }
    int rc;
    if (!(ctx->seeds_host.size() == total && memcmp(ctx->seeds_host.data(), host.data(), total) == 0)) {
        // (the same seeds as last time -- every batch of a streamed run -- are on the device already: no upload, and no wait on the stream)
        HIPCHK(ctx, hipDeviceSynchronize());                                      // an earlier launch may still read the old block
        ctx->seeds_host.clear();
        if ((rc = ensure(ctx, ctx->d_seeds, total))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_seeds.p, host.data(), total, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipStreamSynchronize(s));                                     // `host` is pageable memory owned by this call
        ctx->seeds_host = host;
    }
    c2_strand_args A;
    const uint8_t* base = (const uint8_t*)ctx->d_seeds.p;
    A.reads = d_reads; A.offsets = d_offsets; A.n_reads = n_reads; A.seed_blob = base;
    A.seed_off = (const int32_t*)(base + o_off); A.seed_len = (const int32_t*)(base + o_len); A.n_seeds = (const int32_t*)(base + o_n);
    A.n_refs = n_refs; A.max_seeds = std::max(max_seeds, 1); A.seed_min = seed_min; A.max_read_len = max_read_len; A.plan = d_plan;
    A.seed_table = seed_table ? 1 : 0; A.reserved = 0;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)c2_strand_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const uint64_t wgs = std::min<uint64_t>((n_reads + 3) / 4, (uint64_t)ctx->prop.multiProcessorCount * 8u);
    hipLaunchKernelGGL(c2_strand_plan_kernel, dim3((unsigned)std::max<uint64_t>(1, wgs)), dim3(256), lds, s, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
