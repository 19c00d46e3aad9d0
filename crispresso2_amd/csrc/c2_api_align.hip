// c2_api_align.hip -- host side of the C ABI declared in include/crispresso2_amd.h: contexts, scoring, references, the align + classify launch chain (device and host batches), the per-call global_align, self-tests.
// Marshals the caller's inputs into the kernels' tables, owns the device buffers of a context, picks launch geometry and
// launches.  Nothing here computes an alignment or a classification on the CPU.
#include "c2_ctx.h"
#include "c2_k_align.hip"

std::string g_create_error;
std::mutex g_mutex;

namespace {

struct Geometry {
    int R, passes, max_lj;
    uint32_t lds_full; int blocks_full;          // full pointer plane
    bool full_hbm; uint64_t full_plane_words;    // ... in per-workgroup HBM scratch (it does not fit LDS, or a batch prefers it): 32-bit words per workgroup
    bool full_both; uint32_t lds_full_l; int blocks_full_l;    // the plane fits LDS too: a batch's last launch comes in both forms, the list's length decides (list_gate)
    int band_lanes; uint32_t lds_band; int blocks_band;   // banded first launch (band_lanes == 0: not used)
    bool diag; uint32_t lds_diag; int blocks_diag;         // diagonal-band launches
    bool x[2]; uint32_t lds_x[2]; int blocks_x[2]; uint32_t plane_words;   // multi-alignment tiers in front of it: 4, 2 per wavefront
    bool pk; uint32_t lds_pk; int blocks_pk; uint32_t plane_words_pk;      // packed first tier (8 per wavefront, int16) in place of the 4-per-wavefront one
    bool pk2; uint32_t lds_pk2; int blocks_pk2; uint32_t plane_words_pk2;  // packed second tier (4 per wavefront, 62 diagonals) in place of the 2-per-wavefront one
    bool pk3; uint32_t lds_pk3; int blocks_pk3; uint32_t plane_words_pk3;  // packed third tier (2 per wavefront, 128 diagonals) in front of c2_align_diag_kernel
    bool pk6; uint32_t lds_pk6; int blocks_pk6; uint32_t plane_words_pk6;  // round 5: a packed tier of 40 diagonals (6 per wavefront: three lane groups of 21 lanes) between the first two
};

template <int R, int BAND>          // (the kernel's MODE: 0 full plane in LDS, 1 banded, 2 full plane in HBM)
int occupancy(c2_ctx* ctx, uint32_t lds, int& blocks) {
    // cached per (kernel instance, LDS size): the per-call API path comes through here for every alignment
    int& cached_lds = ctx->occ_lds[R][BAND];
    int& cached_blocks = ctx->occ_blocks[R][BAND];
    if (cached_lds != (int)lds) {
        int nb = 0;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)c2_align_classify_kernel<R, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)c2_align_classify_kernel<R, BAND>, 64, lds));
        cached_blocks = nb < 1 ? 1 : nb;
        cached_lds = (int)lds;
    }
    blocks = cached_blocks;
    return 0;
}

template <int BAND>
int occupancy_r(c2_ctx* ctx, int R, uint32_t lds, int& blocks) {
    switch (R) {
        case 1: return occupancy<1, BAND>(ctx, lds, blocks);
        case 2: return occupancy<2, BAND>(ctx, lds, blocks);
        case 3: return occupancy<3, BAND>(ctx, lds, blocks);
        default: return occupancy<4, BAND>(ctx, lds, blocks);
    }
}

// which references the packed (int16) fill may take: host-side arithmetic on the copies kept for the row tables
void update_pk_eligibility(c2_ctx* ctx) {
    if (!ctx->pk_dirty) return;
    ctx->ref_pk_ok.assign((size_t)ctx->n_refs, 0);
    ctx->any_pk_ok = false;
    if (ctx->have_scoring && !getenv("C2_NO_PACKED_FILL"))
        for (int r = 0; r < ctx->n_refs; ++r) {
            ctx->ref_pk_ok[r] = c2_pk_eligible(ctx->ref_seq[r].data(), ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend, 126) ? 1 : 0;   // (the widest band a packed kernel sweeps)
            if (ctx->ref_pk_ok[r]) ctx->any_pk_ok = true;
        }
    // the 32-bit-add variant of the packed fill: one bias for the whole context (the score-pair tables in LDS carry it), admitted only
    // if every reference the packed fill admits stays in range with it
    ctx->pk_beta = 0;
    if (ctx->any_pk_ok && !getenv("C2_NO_PK_ADD32")) {
        int beta = 0;
        for (int r = 0; r < ctx->n_refs; ++r)
            if (ctx->ref_pk_ok[r]) beta = std::max(beta, c2_pk_beta_needed(ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend));
        int bias = 0;
        for (int r = 0; r < ctx->n_refs; ++r)
            if (ctx->ref_pk_ok[r]) bias = std::max(bias, c2_pk_add32_bias_needed(ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend, 126));
        bool ok = beta > 0;
        for (int r = 0; r < ctx->n_refs && ok; ++r)
            if (ctx->ref_pk_ok[r]) ok = c2_pk_add32_ok(ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend, 126, beta, bias);
        if (ok) { ctx->pk_beta = beta; ctx->pk_bias = bias; }
    }
    ctx->pk_dirty = false;
}

// the packed kernel of a tier: NA alignments per wavefront, with packed (v_pk_add_i16) or plain 32-bit adds
const void* pk_kernel(int na, bool add32) {
    switch (na) {
        case 16: return add32 ? (const void*)c2_align_diagp_kernel<16, true> : (const void*)c2_align_diagp_kernel<16, false>;
        case 8: return add32 ? (const void*)c2_align_diagp_kernel<8, true> : (const void*)c2_align_diagp_kernel<8, false>;
        case 6: return add32 ? (const void*)c2_align_diagp_kernel<6, true> : (const void*)c2_align_diagp_kernel<6, false>;
        case 4: return add32 ? (const void*)c2_align_diagp_kernel<4, true> : (const void*)c2_align_diagp_kernel<4, false>;
        default: return add32 ? (const void*)c2_align_diagp_kernel<2, true> : (const void*)c2_align_diagp_kernel<2, false>;
    }
}

// n_tasks: the batch's size if the caller has one (0: a per-call alignment, or an information request)
int geometry(c2_ctx* ctx, int max_lj, Geometry& g, const uint64_t n_tasks = 0) {
    if (!ctx->have_scoring || ctx->n_refs <= 0) { ctx->err = "scoring and references must be set first"; return C2_E_STATE; }
    g.R = c2_choose_rows_per_lane(ctx->max_li);
    g.passes = (ctx->max_li + 64 * g.R - 1) / (64 * g.R);
    g.max_lj = std::max(max_lj, 1);
    const size_t lds_cu = 163840;
    g.lds_full = c2_make_plan(g.R, g.max_lj, g.passes, ctx->sc.n_codes, 0).total;
    g.full_hbm = false; g.full_plane_words = 0;
    g.full_both = false; g.lds_full_l = 0; g.blocks_full_l = 0;
    bool hbm_by_choice = false;                                    // (the plane would fit LDS: the banded row-strip launch in front is still possible)
    int blocks_full_lds = 0;                                       // workgroups per CU of the full launch with its plane in LDS
    int rc;
    if (g.lds_full > lds_cu || getenv("C2_FORCE_HBM_PLANE")) {
        // the full pointer plane does not fit LDS: it goes to per-workgroup scratch in HBM, LDS keeps the O(Li + Lj) parts
        g.full_hbm = true;
        g.lds_full = c2_make_plan(g.R, g.max_lj, g.passes, ctx->sc.n_codes, 0, true).total;
        g.full_plane_words = (c2_hbm_plane_halfwords(g.max_lj, g.passes) + 1) / 2;
        if (g.lds_full > lds_cu || g.full_plane_words > 0xFFFFFFFFull) {
            ctx->err = "alignment of " + std::to_string(ctx->max_li) + " x " + std::to_string(g.max_lj) +
                       " needs " + std::to_string(g.lds_full) + " bytes of LDS for its strings and boundary rows; limit is " + std::to_string(lds_cu);
            return C2_E_TOO_LARGE;
        }
        if ((rc = occupancy_r<2>(ctx, g.R, g.lds_full, g.blocks_full))) return rc;
    } else {
        if ((rc = occupancy_r<0>(ctx, g.R, g.lds_full, g.blocks_full))) return rc;
        blocks_full_lds = g.blocks_full;
        // Round 5: the plane of a 250 x 250 alignment fits LDS (33 KB) -- but then four workgroups share a CU, one wavefront per SIMD, and the sweep
        // waits for its own dependent chain: 71 ns per alignment.  With the plane in HBM scratch (one 128-byte line per step) LDS holds the O(Li + Lj)
        // parts only, the registers allow four wavefronts per SIMD, and the same launch takes 35 ns per alignment (measured: the reads that match their
        // amplicon nowhere, which every band tier hands on to this launch, are 5 % of a real run).  A batch takes that plan whenever it puts more
        // workgroups on a CU; a single alignment (the per-call API) keeps the plane in LDS.  C2_FULL_PLANE_IN_LDS=1: as before.
        if (n_tasks >= 4096 && !getenv("C2_FULL_PLANE_IN_LDS")) {
            const uint32_t lds_h = c2_make_plan(g.R, g.max_lj, g.passes, ctx->sc.n_codes, 0, true).total;
            const uint64_t words_h = (c2_hbm_plane_halfwords(g.max_lj, g.passes) + 1) / 2;
            int blocks_h = 0;
            if (lds_h <= lds_cu && words_h <= 0xFFFFFFFFull && !(rc = occupancy_r<2>(ctx, g.R, lds_h, blocks_h)) && blocks_h > g.blocks_full) {
                g.full_both = true; g.lds_full_l = g.lds_full; g.blocks_full_l = g.blocks_full;
                g.full_hbm = true; g.lds_full = lds_h; g.full_plane_words = words_h; g.blocks_full = blocks_h;
                hbm_by_choice = true;
            }
        }
    }
    // Diagonal-band first launch (c2_align_diag_kernel): needs the packed score rows and a negative per-gap-base bound
    g.diag = false; g.lds_diag = 0; g.blocks_diag = 0;
    for (int t = 0; t < 2; ++t) { g.x[t] = false; g.lds_x[t] = 0; g.blocks_x[t] = 0; }
    g.plane_words = 0;
    g.pk = false; g.lds_pk = 0; g.blocks_pk = 0; g.plane_words_pk = 0;
    g.pk2 = false; g.lds_pk2 = 0; g.blocks_pk2 = 0; g.plane_words_pk2 = 0;
    g.pk3 = false; g.lds_pk3 = 0; g.blocks_pk3 = 0; g.plane_words_pk3 = 0;
    g.pk6 = false; g.lds_pk6 = 0; g.blocks_pk6 = 0; g.plane_words_pk6 = 0;
    const int km = ctx->kernel_mode;
    update_pk_eligibility(ctx);
    if ((km == 0 || km == 3 || km == 4 || km == 5) && !ctx->sc.pk.empty() && std::max(ctx->gap_open, ctx->gap_extend) + ctx->gmax < 0) {
        g.lds_diag = c2_make_diag_plan(ctx->max_li, g.max_lj).total;
        if (const char* pad = getenv("C2_DEBUG_DIAG_LDS_PAD")) g.lds_diag += (uint32_t)atoi(pad);   // occupancy experiments
        if (g.lds_diag <= lds_cu) {
            if (ctx->occ_diag_lds != (int)g.lds_diag) {
                int nb = 0;
                HIPCHK(ctx, hipFuncSetAttribute((const void*)c2_align_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)c2_align_diag_kernel, 64, g.lds_diag));
                ctx->occ_diag_blocks = nb < 1 ? 1 : nb; ctx->occ_diag_lds = (int)g.lds_diag;
            }
            g.blocks_diag = ctx->occ_diag_blocks;
            g.diag = true;
        }
        if (g.diag && km == 0 && ctx->any_pk_ok) {
            // first tier: eight alignments per wavefront, two per lane group in int16 (c2_align_diagp_kernel)
            c2_diagx_plan PP = c2_make_diagx_plan(8, ctx->max_li, g.max_lj, true);
            if (const char* pad = getenv("C2_DEBUG_PK_LDS_PAD")) PP.total += (uint32_t)atoi(pad);   // occupancy experiments
            if (PP.total <= lds_cu) {
                if (ctx->occ_pk_lds != (int)PP.total * (ctx->pk_beta > 0 ? -1 : 1)) {
                    int nb = 0;
                    HIPCHK(ctx, hipFuncSetAttribute(pk_kernel(8, ctx->pk_beta > 0), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pk_kernel(8, ctx->pk_beta > 0), 64, PP.total));
                    ctx->occ_pk_blocks = nb < 1 ? 1 : nb; ctx->occ_pk_lds = (int)PP.total * (ctx->pk_beta > 0 ? -1 : 1);
                }
                g.pk = true; g.lds_pk = PP.total; g.blocks_pk = ctx->occ_pk_blocks; g.plane_words_pk = PP.n_words * 128u;   // 8 slots x 16 lanes
            }
            // second tier: four per wavefront, two lane groups of 32 lanes (62 diagonals) in int16
            const c2_diagx_plan P2 = c2_make_diagx_plan(4, ctx->max_li, g.max_lj, true);
            if (g.pk && P2.total <= lds_cu && !getenv("C2_NO_PACKED_TIER2")) {
                if (ctx->occ_pk2_lds != (int)P2.total * (ctx->pk_beta > 0 ? -1 : 1)) {
                    int nb = 0;
                    HIPCHK(ctx, hipFuncSetAttribute(pk_kernel(4, ctx->pk_beta > 0), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pk_kernel(4, ctx->pk_beta > 0), 64, P2.total));
                    ctx->occ_pk2_blocks = nb < 1 ? 1 : nb; ctx->occ_pk2_lds = (int)P2.total * (ctx->pk_beta > 0 ? -1 : 1);
                }
                g.pk2 = true; g.lds_pk2 = P2.total; g.blocks_pk2 = ctx->occ_pk2_blocks; g.plane_words_pk2 = P2.n_words * 128u;   // 4 slots x 32 lanes
            }
            // between the first two (round 5): six per wavefront, three lane groups of 21 lanes, 40 diagonals -- the band of a read that overhangs its
            // amplicon at both ends (4 bases in front, 23 behind in the reference's own test data: 28 diagonals + margins), which the 32-diagonal tier
            // cannot certify and the 62-diagonal tier fills at twice the cost.  No 32-bit twin: what it cannot pair goes on to the next tier.
            // C2_NO_TIER40=1 leaves it out.
            const c2_diagx_plan P6 = c2_make_diagx_plan(6, ctx->max_li, g.max_lj, true);
            if (g.pk2 && P6.total <= lds_cu && !getenv("C2_NO_TIER40")) {
                if (ctx->occ_pk6_lds != (int)P6.total * (ctx->pk_beta > 0 ? -1 : 1)) {
                    int nb = 0;
                    HIPCHK(ctx, hipFuncSetAttribute(pk_kernel(6, ctx->pk_beta > 0), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pk_kernel(6, ctx->pk_beta > 0), 64, P6.total));
                    ctx->occ_pk6_blocks = nb < 1 ? 1 : nb; ctx->occ_pk6_lds = (int)P6.total * (ctx->pk_beta > 0 ? -1 : 1);
                }
                g.pk6 = true; g.lds_pk6 = P6.total; g.blocks_pk6 = ctx->occ_pk6_blocks; g.plane_words_pk6 = P6.n_words * 144u;   // 6 slots x 24 words (21 lanes rounded up)
            }
            // third tier: two per wavefront, one lane group of 64 lanes (126 diagonals) in int16
            const c2_diagx_plan P3 = c2_make_diagx_plan(2, ctx->max_li, g.max_lj, true);
            if (g.pk2 && P3.total <= lds_cu && !getenv("C2_NO_PACKED_TIER3")) {
                if (ctx->occ_pk3_lds != (int)P3.total * (ctx->pk_beta > 0 ? -1 : 1)) {
                    int nb = 0;
                    HIPCHK(ctx, hipFuncSetAttribute(pk_kernel(2, ctx->pk_beta > 0), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pk_kernel(2, ctx->pk_beta > 0), 64, P3.total));
                    ctx->occ_pk3_blocks = nb < 1 ? 1 : nb; ctx->occ_pk3_lds = (int)P3.total * (ctx->pk_beta > 0 ? -1 : 1);
                }
                g.pk3 = true; g.lds_pk3 = P3.total; g.blocks_pk3 = ctx->occ_pk3_blocks; g.plane_words_pk3 = P3.n_words * 128u;   // 2 slots x 64 lanes
            }
        }
        for (int t = 0; t < 2 && g.diag && km != 3; ++t) {
            const int na = t == 0 ? C2_TIER0_NA : 2;
            if (t == 0 && km == 4) continue;                                   // mode 4: 2 -> 1  (with a packed kernel in the tier, the 32-bit kernel runs what it could not pair)
            c2_diagx_plan PX = c2_make_diagx_plan(na, ctx->max_li, g.max_lj);
            if (const char* pad = getenv("C2_DEBUG_X_LDS_PAD")) PX.total += (uint32_t)atoi(pad);   // occupancy experiments
            if (PX.total > lds_cu) continue;
            const void* fn = t == 0 ? (const void*)c2_align_diagx_kernel<C2_TIER0_NA> : (const void*)c2_align_diagx_kernel<2>;
            if (ctx->occ_x_lds[t] != (int)PX.total) {
                int nb = 0;
                HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64, PX.total));
                ctx->occ_x_blocks[t] = nb < 1 ? 1 : nb; ctx->occ_x_lds[t] = (int)PX.total;
            }
            g.lds_x[t] = PX.total; g.blocks_x[t] = ctx->occ_x_blocks[t]; g.plane_words = PX.n_words * 64u;                      // [slot][group of 8 anti-diagonals][lane of the slot]
            g.x[t] = true;
        }
        // A packed kernel hands the tasks it could not pair to the 32-bit kernel of the SAME band (un_list).  That kernel's LDS plan is
        // the larger one (references of ~3.3-4 kb fit the packed plan only): without it nothing would ever run those tasks, so such a
        // tier gets no packed kernel either and its tasks fall through to the next tier.  (The third packed tier's unpaired list is
        // run by c2_align_diag_kernel, which every diagonal chain has.)
        if (g.pk && !g.x[0]) g.pk = false;
        if (g.pk2 && !g.x[1]) g.pk2 = false;
        if (!g.pk2) g.pk6 = false;
    }
    // Banded first launch: keep only the pointer words of the lanes near the main diagonal so that more workgroups fit
    // a CU (the DP is latency-bound at one wave per SIMD).  band: -1 auto, 0 off, >0 lanes on each side.
    g.band_lanes = 0; g.lds_band = 0; g.blocks_band = 0;
    int want = ctx->kernel_mode == 2 ? 0 : ctx->band_setting;
    if (!g.diag && g.passes == 1 && want != 0 && (!g.full_hbm || hbm_by_choice)) {
        if (want < 0) {
            // auto: the widest band whose plan still lets `band_target_wgs` workgroups share a CU's LDS
            want = 0;
            for (int w = 1; w < 24; ++w)
                if (c2_make_plan(g.R, g.max_lj, 1, ctx->sc.n_codes, w).total <= lds_cu / (size_t)ctx->band_target_wgs) want = w;
        }
        if (want >= 2 && c2_band_slots(g.R, want) < C2_LANES - 8) {
            g.band_lanes = want;
            g.lds_band = c2_make_plan(g.R, g.max_lj, 1, ctx->sc.n_codes, want).total;
            if ((rc = occupancy_r<1>(ctx, g.R, g.lds_band, g.blocks_band))) return rc;
            if (g.blocks_band <= (hbm_by_choice ? blocks_full_lds : g.blocks_full)) g.band_lanes = 0;      // no occupancy to gain
        }
    }
    return 0;
}

template <int R, int BAND>
int launch_one(c2_ctx* ctx, const c2_align_args& A, uint32_t lds, int blocks_per_cu, uint64_t work_items, hipStream_t s) {
    const uint64_t resident = (uint64_t)ctx->prop.multiProcessorCount * (uint64_t)blocks_per_cu;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(work_items, resident));
    hipLaunchKernelGGL((c2_align_classify_kernel<R, BAND>), dim3(grid), dim3(64), lds, s, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// workgroups of the HBM-plane launch: what is resident, but no more than fit C2_HBM_PLANE_BUDGET bytes of scratch
constexpr uint64_t C2_HBM_PLANE_BUDGET = 8ull << 30;
uint64_t hbm_plane_wgs(const c2_ctx* ctx, const Geometry& g, uint64_t work_items) {
    const uint64_t resident = (uint64_t)ctx->prop.multiProcessorCount * (uint64_t)g.blocks_full;
    const uint64_t by_budget = std::max<uint64_t>(1, C2_HBM_PLANE_BUDGET / (g.full_plane_words * 4));
    return std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(work_items, resident), by_budget));
}

// the last launch of every chain: full pointer plane, in LDS or (if it does not fit) in HBM scratch
template <int R>
int launch_full(c2_ctx* ctx, c2_align_args& A, const Geometry& g, hipStream_t s, unsigned long long* second_counter = nullptr) {
    if (!g.full_hbm) return launch_one<R, 0>(ctx, A, g.lds_full, g.blocks_full, A.n_tasks, s);
    int rc;
    constexpr int GATE = 8192;                                     // (0.66 ms = what one alignment takes with the plane in HBM = 9 k alignments with it in LDS)
    if (g.full_both && A.task_list && second_counter) {
        // a list of unknown length: the LDS form works if it is short, the HBM form if it is long (the other one returns at once)
        c2_align_args L = A;
        L.list_gate = GATE;
        if ((rc = launch_one<R, 0>(ctx, L, g.lds_full_l, g.blocks_full_l, std::min<uint64_t>(A.n_tasks, (uint64_t)GATE), s))) return rc;
        A.list_gate = -GATE;
        A.work_counter = second_counter;
    }
    const uint64_t wgs = hbm_plane_wgs(ctx, g, A.n_tasks);
    if ((rc = ensure(ctx, ctx->d_plane, (size_t)(wgs * g.full_plane_words * sizeof(uint32_t))))) return rc;
    A.plane = (uint32_t*)ctx->d_plane.p; A.plane_words_per_wg = (uint32_t)g.full_plane_words;
    hipLaunchKernelGGL((c2_align_classify_kernel<R, 2>), dim3((unsigned)wgs), dim3(64), g.lds_full, s, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// Can a band tier of `bandw` diagonals serve ANY task of a batch whose reads are min_lj .. max_lj long?  The kernels' own test (c2_diagx_body:
// the band d0 .. d0 + bandw - 1 placed symmetrically about D / 2 must hold diagonal 0 and diagonal D = len(ref) - len(read)), over every
// reference and read length.  min_lj <= 0: not known -> yes.
bool tier_can_serve(const c2_ctx* ctx, const int bandw, const int min_lj, const int max_lj) {
    if (min_lj <= 0 || min_lj > max_lj) return true;
    for (const int li : ctx->ref_len)
        for (int lj = min_lj; lj <= max_lj; ++lj) {
            const int D = li - lj, d0 = ((D - bandw + 3) >> 1) & ~1;
            if (d0 <= 0 && d0 + bandw - 1 >= 0 && D >= d0 && D <= d0 + bandw - 1) return true;
        }
    return false;
}

template <int R>
int launch_align(c2_ctx* ctx, c2_align_args A, const Geometry& g, hipStream_t s, const int min_lj = 0) {
    TimedLaunch tl{};
    if (ctx->timing) {
        HIPCHK(ctx, hipEventCreate(&tl.a)); HIPCHK(ctx, hipEventCreate(&tl.m0)); HIPCHK(ctx, hipEventCreate(&tl.m)); HIPCHK(ctx, hipEventCreate(&tl.b));
        HIPCHK(ctx, hipEventRecord(tl.a, s));
    }
    ctx->last_score_stage = false; ctx->last_n_tasks = A.n_tasks;
    bool first_marked = false, first_started = false;
    // the "first kernel" of the timing split: the kernel of the first band tier (behind the score-only stage, if that runs)
    auto start_first = [&]() { if (ctx->timing && !first_started) { (void)hipEventRecord(tl.m0, s); first_started = true; } };
    auto mark_first = [&]() { start_first(); if (ctx->timing && !first_marked) { (void)hipEventRecord(tl.m, s); first_marked = true; } };
    int rc;
    A.band_lanes = 0; A.reserved = (getenv("C2_DEBUG_SKIP_STRINGS") ? 1 : 0) | (getenv("C2_DEBUG_SKIP_EPILOGUE") ? 2 : 0) | (getenv("C2_DEBUG_HALF_FILL") ? 4 : 0) | (getenv("C2_DEBUG_SKIP_EMIT") ? 16 : 0) | (getenv("C2_DEBUG_SKIP_TRACE") ? 32 : 0) | (getenv("C2_NO_PAIR_SORT") ? 64 : 0);   // (measurement knobs)
    if (A.n_refs > 1 && !getenv("C2_NO_BLOCK_GRABS")) A.reserved |= 8;      // several references: the work counter hands out blocks of groups (c2_diagx_body)
    A.fb_count = nullptr; A.fb_list = nullptr; A.task_list = nullptr; A.task_count = nullptr;
    A.un_list = nullptr; A.un_count = nullptr; A.pair_order = 0;
    if (g.diag || g.band_lanes > 0) {
        if (A.n_tasks > 0xFFFFFFFFull) { ctx->err = "more than 2^32 tasks in one launch"; return C2_E_INVALID; }
        // d_fb: 64 header words -- [0..7] length of the list each BAND tier leaves for the next one, [8..15] length of the list of
        // tasks a packed kernel could not pair (run by the 32-bit kernel of the same band), [16 + 2l ..] work counter of launch l --
        // then eight task lists: one per band tier (c2_align_partition_kernel writes into the lists of LATER tiers, so they cannot share
        // buffers), the unpaired tasks', the score-only launch's, the first tier's when the partition ran, the 14-diagonal launch's
        // (header words 48..54: tasks per class of the partition, 56 / 57: length of the first tier's list after the score-only launch and after
        //  the 14-diagonal launch, 60 / 62 / 61: lengths of the score-only launch's list, the 14-diagonal launch's, the first tier's)
        const size_t list_words = (size_t)A.n_tasks;
        if ((rc = ensure(ctx, ctx->d_fb, 256 + 8 * list_words * sizeof(uint32_t)))) return rc;
        uint32_t* hdr = (uint32_t*)ctx->d_fb.p;
        uint32_t* lists[4] = {hdr + 64, hdr + 64 + list_words, hdr + 64 + 2 * list_words, hdr + 64 + 3 * list_words};
        uint32_t* ulist = hdr + 64 + 4 * list_words;
        uint32_t* elist = hdr + 64 + 5 * list_words;                 // the tasks of the score-only launch
        uint32_t* nlist = hdr + 64 + 6 * list_words;                 // the first band tier's, when the partition ran
        uint32_t* plist = hdr + 64 + 7 * list_words;                 // the 14-diagonal launch's
        HIPCHK(ctx, hipMemsetAsync(hdr, 0, 256, s));
        const uint64_t cus = (uint64_t)ctx->prop.multiProcessorCount;
        int tier = 0;                                            // band tiers so far; the next one reads lists[(tier - 1) & 1]
        int launch = 0;                                          // launches so far (each has its own work counter)
        // wire a launch into the chain.  from_unpaired: it runs the tasks the packed kernel of this band tier could not pair and
        // appends what IT cannot finish to the same list as that kernel
        auto chain = [&](c2_align_args& T, const bool from_unpaired, const bool packed_kernel) {
            if (from_unpaired) { T.task_list = ulist; T.task_count = hdr + 8 + tier; }
            else { T.task_list = tier ? lists[tier - 1] : nullptr; T.task_count = tier ? hdr + (tier - 1) : nullptr; }
            T.fb_list = lists[tier]; T.fb_count = hdr + tier;
            T.un_list = packed_kernel ? ulist : nullptr; T.un_count = packed_kernel ? hdr + 8 + tier : nullptr;
            T.pair_order = packed_kernel && !from_unpaired && tier == 0 && T.all_refs && T.n_refs > 1;
            T.work_counter = (unsigned long long*)(hdr + 16 + 2 * launch);
            ++launch;
        };
        if (g.diag) {
            {   // one scratch plane, sized for the tier that needs the most (no reallocation between launches)
                uint64_t most = 0;
                for (int t = 0; t < 2; ++t) if (g.x[t]) most = std::max<uint64_t>(most, cus * (uint64_t)g.blocks_x[t] * g.plane_words);
                if (g.pk) most = std::max<uint64_t>(most, cus * (uint64_t)g.blocks_pk * g.plane_words_pk);
                if (g.pk2) most = std::max<uint64_t>(most, cus * (uint64_t)g.blocks_pk2 * g.plane_words_pk2);
                if (g.pk3) most = std::max<uint64_t>(most, cus * (uint64_t)g.blocks_pk3 * g.plane_words_pk3);
                if (g.pk6) most = std::max<uint64_t>(most, cus * (uint64_t)g.blocks_pk6 * g.plane_words_pk6);
                if (g.full_hbm) most = std::max<uint64_t>(most, hbm_plane_wgs(ctx, g, A.n_tasks) * g.full_plane_words);
                if (most && (rc = ensure(ctx, ctx->d_plane, (size_t)most * sizeof(uint32_t)))) return rc;
            }
            // band tier t (0: 30 diagonals, 1: 62): the packed kernel if the tier has one, then the 32-bit kernel of the same band --
            // over everything if there is no packed kernel, else over the tasks the packed kernel could not pair
            // In front of the first band tier, when that tier has its packed kernel: c2_align_partition_kernel looks at every task and says which
            // launch should see it first --
            //   the tasks whose read is as long as its reference and differs from it in few columns go through the SCORE-ONLY packed fill
            //   (c2_align_diags_kernel: no pointer bits, no pointer words), which finishes the ones whose alignment is the main diagonal;
            //   the tasks whose path needs more diagonals than the first tier's band has go straight to the list of the tier that has them;
            //   (C2_P16_TIER=1 only: the tasks whose path needs a few diagonals through c2_align_diagp_kernel<16>, 14 diagonals, sixteen per wavefront.)
            // What a launch cannot finish joins the list of the next wider one, as ever.  (Not for an all-references batch of several references:
            // its pairs are formed by the order of the tasks.  C2_NO_SCORE_TIER=1 switches the whole stage off, C2_NO_ROUTE=1 the routing to
            // later tiers.)
            // (round 5: also for an all-references batch of several references -- the partition walks its chunks reference-major, so the lists'
            //  neighbours share a reference and pair; C2_NO_ALLREFS_PARTITION=1 gives such a batch round 4's chain: the first packed launch in pair order)
            const bool many_refs = A.all_refs && A.n_refs > 1;
            bool score_stage = g.pk && !(many_refs && (A.n_refs > 64 || getenv("C2_NO_ALLREFS_PARTITION"))) && !getenv("C2_NO_SCORE_TIER") &&
                               ctx->kernel_mode == 0 && tier_can_serve(ctx, 32, min_lj, A.max_lj);
            bool p16_stage = false;
            bool tier40_runs = false;                                    // (only behind the partition: without its routing every task the first tier leaves would be filled there too)
            if (score_stage) {
                const bool a32s = ctx->pk_beta > 0;
                const bool tier1_runs = (g.pk2 || g.x[1]) && tier_can_serve(ctx, 62, min_lj, A.max_lj);
                tier40_runs = g.pk6 && tier1_runs && tier_can_serve(ctx, 40, min_lj, A.max_lj);
                const bool route = !getenv("C2_NO_ROUTE");
                c2_diagx_plan PP = c2_make_diagx_plan(16, ctx->max_li, g.max_lj, true, false);
                p16_stage = getenv("C2_P16_TIER") && PP.total <= 163840u && tier_can_serve(ctx, 14, min_lj, A.max_lj);   // (measured: slower than leaving its tasks to the first tier -- opt-in)
                // the list every band tier READS is lists[its index - 1]; the tiers in launch order: 32 | 40 (if it runs) | 62 (if it runs) | 128
                const int i40 = 1, i62 = 1 + (tier40_runs ? 1 : 0), i128 = i62 + (tier1_runs ? 1 : 0), ifull = i128 + 1;
                c2_partition_args PA;
                PA.A = A;
                PA.list[0] = elist; PA.count[0] = hdr + 60;
                PA.list[1] = plist; PA.count[1] = hdr + 62;
                PA.list[2] = nlist; PA.count[2] = hdr + 61;
                PA.list[3] = lists[i40 - 1]; PA.count[3] = hdr + (i40 - 1);      // (class 3 exists only when the 40-diagonal tier runs: bandw[2] below)
                PA.list[4] = lists[i62 - 1]; PA.count[4] = hdr + (i62 - 1);      // the 62-diagonal tier's list -- or, without that tier, the list the 128-diagonal tier reads
                PA.list[5] = lists[i128 - 1]; PA.count[5] = hdr + (i128 - 1);
                // class 6 (a read that matches its reference nowhere): the list the LAST launch reads
                PA.list[6] = lists[ifull - 1]; PA.count[6] = hdr + (ifull - 1);
                PA.class_count = hdr + 48;
                PA.bandw[0] = p16_stage ? 14 : 0; PA.bandw[1] = 32; PA.bandw[2] = (route && tier40_runs) ? 40 : 0;
                PA.bandw[3] = (route && tier1_runs) ? 62 : 0; PA.bandw[4] = route ? 128 : 0;
                PA.max_mismatch = 6;                                     // (of the last 32 columns)
                PA.probe_max_mismatch = 4; PA.margin = 3; PA.max_shift = (p16_stage || route) ? 64 : 0;
                PA.direct_full = (route && !getenv("C2_NO_DIRECT_FULL")) ? 1 : 0;
                PA.sort_by_length = getenv("C2_NO_LENGTH_ORDER") ? 0 : 1;
                // a read on its reference's main diagonal (at most two differing bases) is finished by the partition itself where c2_main_diagonal_certificate allows
                PA.exact_copies = ((((uintptr_t)A.aln_read | (uintptr_t)A.aln_ref) & 3u) == 0 && (A.aln_stride & 3u) == 0) ? 1 : 0;
                PA.route_cert = getenv("C2_NO_ROUTE_CERT") ? 0 : 1;
                PA.check_cut = many_refs ? 1 : 0;                        // (one amplicon: a read that differs from it around the cut fails the score-only certificate anyway, and the look costs 0.3 ms per 10 M)
                if (const char* e = getenv("C2_SCORE_TIER_MAX_MISMATCH")) PA.max_mismatch = atoi(e);
                if (const char* e = getenv("C2_ROUTE_PROBE_MISMATCH")) PA.probe_max_mismatch = atoi(e);
                if (const char* e = getenv("C2_ROUTE_MARGIN")) PA.margin = atoi(e);
                const uint64_t chunk_tasks = c2_part_chunk_tasks(A.all_refs, A.n_refs);
                const unsigned pgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + chunk_tasks - 1) / chunk_tasks, cus * 16));
                hipLaunchKernelGGL(c2_align_partition_kernel, dim3(pgrid), dim3(256), C2_PART_LDS, s, PA);
                HIPCHK(ctx, hipGetLastError());
                // (its own LDS plan -- no staging area for pointer words -- and its own residency; SIXTEEN alignments per wavefront: lane groups of
                //  8 lanes, 14 diagonals -- the alignments it can finish run along the main diagonal, and a band that narrow still certifies them;
                //  C2_SCORE_TIER_NA=8: eight per wavefront, the first tier's geometry)
                int sna = 16;
                if (const char* e = getenv("C2_SCORE_TIER_NA")) sna = atoi(e) == 8 ? 8 : 16;
                c2_diagx_plan PS = c2_make_diagx_plan(sna, ctx->max_li, g.max_lj, true, true);
                if (sna == 16 && PS.total > 163840u) { sna = 8; PS = c2_make_diagx_plan(8, ctx->max_li, g.max_lj, true, true); }
                const void* fn = sna == 16 ? (a32s ? (const void*)c2_align_diags_kernel<16, true> : (const void*)c2_align_diags_kernel<16, false>)
                                           : (a32s ? (const void*)c2_align_diags_kernel<8, true> : (const void*)c2_align_diags_kernel<8, false>);
                const int key = (int)PS.total * (a32s ? -1 : 1) * (sna == 16 ? 2 : 1);
                if (ctx->occ_score_lds != key) {
                    int nb = 0;
                    HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64, PS.total));
                    ctx->occ_score_blocks = nb < 1 ? 1 : nb; ctx->occ_score_lds = key;
                }
                {
                    const uint64_t resident = cus * (uint64_t)ctx->occ_score_blocks;
                    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + sna - 1) / sna, resident));
                    c2_align_args T = A;
                    T.task_list = elist; T.task_count = hdr + 60;
                    T.fb_list = nlist; T.fb_count = hdr + 61;               // what it cannot finish joins the first tier's tasks
                    T.un_list = nullptr; T.un_count = nullptr; T.pair_order = 0;
                    T.work_counter = (unsigned long long*)(hdr + 16 + 2 * launch);
                    ++launch;
                    T.plane = nullptr; T.plane_words_per_wg = 0;
                    if (sna == 16) { if (a32s) hipLaunchKernelGGL((c2_align_diags_kernel<16, true>), dim3(grid), dim3(64), PS.total, s, T);
                                     else      hipLaunchKernelGGL((c2_align_diags_kernel<16, false>), dim3(grid), dim3(64), PS.total, s, T); }
                    else           { if (a32s) hipLaunchKernelGGL((c2_align_diags_kernel<8, true>), dim3(grid), dim3(64), PS.total, s, T);
                                     else      hipLaunchKernelGGL((c2_align_diags_kernel<8, false>), dim3(grid), dim3(64), PS.total, s, T); }
                    HIPCHK(ctx, hipGetLastError());
                    HIPCHK(ctx, hipMemcpyAsync(hdr + 56, hdr + 61, 4, hipMemcpyDeviceToDevice, s));      // (statistics: the list's length now)
                }
                if (p16_stage) {
                    // the pointer-keeping fill at sixteen per wavefront over the tasks predicted to need a few diagonals only; a task it cannot pair
                    // or certify joins the first tier's list like the score-only launch's
                    const void* fp = a32s ? (const void*)c2_align_diagp_kernel<16, true> : (const void*)c2_align_diagp_kernel<16, false>;
                    const int keyp = (int)PP.total * (a32s ? -1 : 1);
                    if (ctx->occ_p16_lds != keyp) {
                        int nb = 0;
                        HIPCHK(ctx, hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                        HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fp, 64, PP.total));
                        ctx->occ_p16_blocks = nb < 1 ? 1 : nb; ctx->occ_p16_lds = keyp;
                    }
                    const uint64_t resident = cus * (uint64_t)ctx->occ_p16_blocks;
                    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + 15) / 16, resident));
                    const uint64_t plane_words = (uint64_t)grid * PP.n_words * 128u;                      // 16 slots x 8 lanes
                    if ((rc = ensure(ctx, ctx->d_plane16, (size_t)plane_words * sizeof(uint32_t)))) return rc;
                    c2_align_args T = A;
                    T.task_list = plist; T.task_count = hdr + 62;
                    T.fb_list = nlist; T.fb_count = hdr + 61;
                    T.un_list = nullptr; T.un_count = nullptr; T.pair_order = 0;
                    T.work_counter = (unsigned long long*)(hdr + 16 + 2 * launch);
                    ++launch;
                    T.plane = (uint32_t*)ctx->d_plane16.p; T.plane_words_per_wg = PP.n_words * 128u;
                    if (a32s) hipLaunchKernelGGL((c2_align_diagp_kernel<16, true>), dim3(grid), dim3(64), PP.total, s, T);
                    else      hipLaunchKernelGGL((c2_align_diagp_kernel<16, false>), dim3(grid), dim3(64), PP.total, s, T);
                    HIPCHK(ctx, hipGetLastError());
                }
                HIPCHK(ctx, hipMemcpyAsync(hdr + 57, hdr + 61, 4, hipMemcpyDeviceToDevice, s));
                start_first();                                          // (the timing split's "first kernel" is the one that follows)
            }
            else start_first();
            ctx->last_p16_stage = p16_stage;
            ctx->last_score_stage = score_stage; ctx->last_n_tasks = A.n_tasks;
            for (int t = 0; t < 2; ++t) {
                const bool packed = t == 0 ? g.pk : g.pk2;
                if (!packed && !g.x[t]) continue;
                if (!tier_can_serve(ctx, t == 0 ? 32 : 62, min_lj, A.max_lj)) continue;     // (c2_batch.min_read_len: no task could use this band -- the next tier takes them all)
                if (packed) {
                    const int na = t == 0 ? 8 : 4;
                    const uint64_t resident = cus * (uint64_t)(t == 0 ? g.blocks_pk : g.blocks_pk2);
                    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + na - 1) / na, resident));
                    c2_align_args T = A;
                    chain(T, false, true);
                    if (t == 0 && score_stage) { T.task_list = nlist; T.task_count = hdr + 61; }       // (the tasks the partition gave this tier and those the two launches in front could not finish)
                    T.plane = (uint32_t*)ctx->d_plane.p; T.plane_words_per_wg = t == 0 ? g.plane_words_pk : g.plane_words_pk2;
                    const bool a32 = ctx->pk_beta > 0;
                    if (t == 0) { if (a32) hipLaunchKernelGGL((c2_align_diagp_kernel<8, true>), dim3(grid), dim3(64), g.lds_pk, s, T);
                                  else     hipLaunchKernelGGL((c2_align_diagp_kernel<8, false>), dim3(grid), dim3(64), g.lds_pk, s, T); }
                    else        { if (a32) hipLaunchKernelGGL((c2_align_diagp_kernel<4, true>), dim3(grid), dim3(64), g.lds_pk2, s, T);
                                  else     hipLaunchKernelGGL((c2_align_diagp_kernel<4, false>), dim3(grid), dim3(64), g.lds_pk2, s, T); }
                    HIPCHK(ctx, hipGetLastError());
                    mark_first();
                }
                if (g.x[t]) {
                    const int na = t == 0 ? C2_TIER0_NA : 2;
                    const uint64_t resident = cus * (uint64_t)g.blocks_x[t];
                    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + na - 1) / na, resident));
                    c2_align_args T = A;
                    chain(T, packed, false);
                    T.plane = (uint32_t*)ctx->d_plane.p; T.plane_words_per_wg = g.plane_words;
                    if (t == 0) hipLaunchKernelGGL(c2_align_diagx_kernel<C2_TIER0_NA>, dim3(grid), dim3(64), g.lds_x[t], s, T);
                    else         hipLaunchKernelGGL(c2_align_diagx_kernel<2>, dim3(grid), dim3(64), g.lds_x[t], s, T);
                    HIPCHK(ctx, hipGetLastError());
                    mark_first();
                }
                ++tier;
                if (t == 0 && tier40_runs) {
                    // the 40-diagonal tier: six alignments per wavefront (three lane groups of 21 lanes), over what the first tier left and what the
                    // partition sent here directly; a task it cannot pair or certify goes on to the 62-diagonal tier's list
                    const uint64_t resident6 = cus * (uint64_t)g.blocks_pk6;
                    const unsigned grid6 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + 5) / 6, resident6));
                    c2_align_args T6 = A;
                    chain(T6, false, true);
                    T6.un_list = nullptr; T6.un_count = nullptr;             // (no 32-bit twin of this band: unpaired tasks join the next tier's list)
                    T6.plane = (uint32_t*)ctx->d_plane.p; T6.plane_words_per_wg = g.plane_words_pk6;
                    if (ctx->pk_beta > 0) hipLaunchKernelGGL((c2_align_diagp_kernel<6, true>), dim3(grid6), dim3(64), g.lds_pk6, s, T6);
                    else                  hipLaunchKernelGGL((c2_align_diagp_kernel<6, false>), dim3(grid6), dim3(64), g.lds_pk6, s, T6);
                    HIPCHK(ctx, hipGetLastError());
                    ++tier;
                }
            }
            if (g.pk3) {                                               // third band tier, packed: two alignments per wavefront
                const uint64_t resident3 = cus * (uint64_t)g.blocks_pk3;
                const unsigned grid3 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((A.n_tasks + 1) / 2, resident3));
                c2_align_args T3 = A;
                chain(T3, false, true);
                T3.plane = (uint32_t*)ctx->d_plane.p; T3.plane_words_per_wg = g.plane_words_pk3;
                if (ctx->pk_beta > 0) hipLaunchKernelGGL((c2_align_diagp_kernel<2, true>), dim3(grid3), dim3(64), g.lds_pk3, s, T3);
                else                  hipLaunchKernelGGL((c2_align_diagp_kernel<2, false>), dim3(grid3), dim3(64), g.lds_pk3, s, T3);
                HIPCHK(ctx, hipGetLastError());
                mark_first();
            }
            const uint64_t resident = cus * (uint64_t)g.blocks_diag;
            const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(A.n_tasks, resident));
            c2_align_args T = A;
            chain(T, g.pk3, false);                                     // (after a packed kernel: the tasks it could not pair)
            hipLaunchKernelGGL(c2_align_diag_kernel, dim3(grid), dim3(64), g.lds_diag, s, T);
            HIPCHK(ctx, hipGetLastError());
            mark_first();
            ++tier;
        } else {
            A.band_lanes = g.band_lanes;
            chain(A, false, false);
            if ((rc = launch_one<R, 1>(ctx, A, g.lds_band, g.blocks_band, A.n_tasks, s))) return rc;
            mark_first();
            ++tier;
        }
        // the tasks no banded tier could finish, redone with the full pointer plane (usually a handful; an empty list costs one tiny launch)
        A.band_lanes = 0;
        chain(A, false, false);
        A.fb_list = nullptr; A.fb_count = nullptr;
        ctx->last_tiers = tier;
        if ((rc = launch_full<R>(ctx, A, g, s, (unsigned long long*)(hdr + 16 + 2 * launch)))) return rc;      // (the second form's own work counter)
    } else {
        ctx->last_tiers = 0;
        if ((rc = ensure(ctx, ctx->d_fb, 32))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_fb.p, 0, 32, s));
        A.work_counter = (unsigned long long*)((uint32_t*)ctx->d_fb.p + 2);
        if ((rc = launch_full<R>(ctx, A, g, s))) return rc;
    }
    mark_first();
    if (ctx->timing) { HIPCHK(ctx, hipEventRecord(tl.b, s)); ctx->timed.push_back(tl); }
    return 0;
}

// (Re)build the per-reference row tables of the diagonal-band kernel after the references or the scoring changed.
int refresh_diag_rows(c2_ctx* ctx, hipStream_t s) {
    if (!ctx->diag_rows_dirty) return 0;
    std::vector<c2_diag_row> all, one, allpk;
    std::vector<size_t> off(ctx->n_refs, 0);
    update_pk_eligibility(ctx);
    for (int r = 0; r < ctx->n_refs; ++r) {
        c2_build_diag_rows(ctx->ref_seq[r].data(), ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend, one);
        off[r] = all.size();
        all.insert(all.end(), one.begin(), one.end());
        if (ctx->any_pk_ok) {                                      // the packed tables mirror the indexing; a reference that is not admitted gets padding
            const size_t want = one.size();
            if (ctx->ref_pk_ok[r]) c2_build_diag_rows_pk(ctx->ref_seq[r].data(), ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend, one, ctx->pk_beta);
            else one.assign(want, c2_diag_row{0, 0, 0, 5u * C2_PK_LUT_STRIDE});
            allpk.insert(allpk.end(), one.begin(), one.end());
        }
    }
    HIPCHK(ctx, hipDeviceSynchronize());      // (any stream may still run kernels that read the old row tables)
    int rc;
    if (!all.empty()) {
        if ((rc = ensure(ctx, ctx->d_diagrows, all.size() * sizeof(c2_diag_row)))) return rc;
        HIPCHK(ctx, hipMemcpy(ctx->d_diagrows.p, all.data(), all.size() * sizeof(c2_diag_row), hipMemcpyHostToDevice));
    }
    if (!allpk.empty()) {
        if ((rc = ensure(ctx, ctx->d_diagrows_pk, allpk.size() * sizeof(c2_diag_row)))) return rc;
        HIPCHK(ctx, hipMemcpy(ctx->d_diagrows_pk.p, allpk.data(), allpk.size() * sizeof(c2_diag_row), hipMemcpyHostToDevice));
    }
    for (int r = 0; r < ctx->n_refs; ++r) {
        ctx->ref_desc[r].diag_rows = all.empty() ? nullptr : (const c2_diag_row*)ctx->d_diagrows.p + off[r] + C2_DIAG_ROW_PAD;
        ctx->ref_desc[r].pk_ok = (!allpk.empty() && ctx->ref_pk_ok[r]) ? 1 : 0;
        // (on top of the packed fill's admission: its conditions -- no gap step adds score, the sentinel out of reach -- are part of the proof)
        c2_diag_cert dc;
        dc.kmax = -1; dc.mmax[0] = dc.mmax[1] = dc.mmax[2] = dc.mmax[3] = -1;
        if (ctx->ref_desc[r].pk_ok && !getenv("C2_NO_EXACT_COPIES"))
            dc = c2_main_diagonal_certificate(ctx->ref_seq[r].data(), ctx->ref_len[r], ctx->ref_g32[r].data(), ctx->sc, ctx->gap_open, ctx->gap_extend);
        if (const char* e = getenv("C2_DIAG_CERT_KMAX")) dc.kmax = std::min(dc.kmax, atoi(e));       // (0: byte-for-byte copies only)
        ctx->ref_desc[r].diag_kmax = dc.kmax;
        for (int k = 0; k < 4; ++k) ctx->ref_desc[r].diag_mmax[k] = dc.mmax[k];
        ctx->ref_desc[r].reserved_pad = 0;
    }
    HIPCHK(ctx, hipMemcpy(ctx->d_refdesc.p, ctx->ref_desc.data(), sizeof(c2_dev_ref) * (size_t)ctx->n_refs, hipMemcpyHostToDevice));
    ctx->diag_rows_dirty = false;
    return 0;
}

int run_align(c2_ctx* ctx, const c2_batch* b, int max_lj, hipStream_t s) {
    Geometry g;
    int rc = geometry(ctx, max_lj, g, b->n_reads * (uint64_t)(b->all_refs ? std::max(ctx->n_refs, 1) : 1));
    if (rc) return rc;
    if (b->n_reads == 0) return 0;
    if ((rc = refresh_diag_rows(ctx, s))) return rc;
    const uint64_t n_tasks = b->n_reads * (uint64_t)(b->all_refs ? ctx->n_refs : 1);
    if (b->aln_stride < (uint32_t)(ctx->max_li + g.max_lj)) { ctx->err = "aln_stride smaller than longest read + longest reference"; return C2_E_INVALID; }
    c2_align_args A;
    A.reads = b->reads; A.offsets = b->offsets; A.ref_ids = b->all_refs ? nullptr : b->ref_ids; A.strands = b->strands;
    A.refs = (const c2_dev_ref*)ctx->d_refdesc.p;
    A.score_tbl = (const int16_t*)ctx->d_tbl.p; A.code_of_char = (const uint8_t*)ctx->d_code.p;
    A.score_pk = ctx->sc.pk.empty() ? nullptr : (const uint32_t*)ctx->d_pk.p;
    A.aln_read = b->aln_read; A.aln_ref = b->aln_ref; A.records = b->records;
    A.n_tasks = n_tasks; A.aln_stride = b->aln_stride; A.n_refs = ctx->n_refs; A.all_refs = b->all_refs ? 1 : 0;
    A.n_codes = ctx->sc.n_codes; A.gap_open = ctx->gap_open; A.gap_extend = ctx->gap_extend;
    A.max_lj = g.max_lj; A.max_passes = g.passes;
    A.max_li = ctx->max_li;
    A.legacy = (b->flags & C2_BATCH_LEGACY_CLASSIFIER) ? 1 : 0;
    A.plane = nullptr; A.plane_words_per_wg = 0; A.pk_beta = (uint32_t)ctx->pk_beta; A.pk_bias = (uint32_t)ctx->pk_bias; A.list_gate = 0; A.diag_base = (const c2_diag_row*)ctx->d_diagrows.p;
    A.diagpk_base = ctx->any_pk_ok ? (const c2_diag_row*)ctx->d_diagrows_pk.p : nullptr;
    A.mat_dim = ctx->sc.mat_dim; A.first_ext_code = ctx->sc.first_ext_code;
    c2_build_base_luts(ctx->sc, A.lut_code_lo, A.lut_code_hi, A.lut_chr_lo, A.lut_chr_hi);
    {
        int mx = 0;
        for (int16_t v : ctx->sc.tbl) mx = std::max(mx, (int)v);
        A.max_score = mx;
    }
    A.phase_cycles = ctx->phase_prof ? (unsigned long long*)ctx->d_phase.p : nullptr;
    // the batch's hint words: zero, then c2_align_partition_kernel (if the chain has it) writes the ones it has something to say about
    A.diag_hints = b->diag_hints;
    if (b->diag_hints) HIPCHK(ctx, hipMemsetAsync(b->diag_hints, 0, (size_t)n_tasks * 4u * sizeof(uint32_t), s));
    switch (g.R) {
        case 1: return launch_align<1>(ctx, A, g, s, b->min_read_len);
        case 2: return launch_align<2>(ctx, A, g, s, b->min_read_len);
        case 3: return launch_align<3>(ctx, A, g, s, b->min_read_len);
        default: return launch_align<4>(ctx, A, g, s, b->min_read_len);
    }
}

}  // namespace

extern "C" {

int c2_abi_version(void) { return C2_ABI_VERSION; }

int c2_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* c2_last_error(const c2_ctx* ctx) {
    if (!ctx) return g_create_error.c_str();
    return ctx->err.c_str();
}

int c2_create(int device, c2_ctx** out) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (!out) { g_create_error = "out is NULL"; return C2_E_INVALID; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count 0"); return C2_E_DEVICE; }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return C2_E_INVALID; }
    c2_ctx* ctx = new c2_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("device init: ") + hipGetErrorString(e);
        delete ctx;
        return C2_E_DEVICE;
    }
    *out = ctx;
    return 0;
}

void c2_destroy(c2_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& t : ctx->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    DevBuf* all[] = {&ctx->d_tbl, &ctx->d_code, &ctx->d_pk, &ctx->d_refblob, &ctx->d_refdesc, &ctx->d_reads, &ctx->d_offsets, &ctx->d_refids,
                     &ctx->d_strands, &ctx->d_aln_read, &ctx->d_aln_ref, &ctx->d_records, &ctx->d_misc, &ctx->d_phase, &ctx->d_fb, &ctx->d_cnt, &ctx->d_diagrows, &ctx->d_diagrows_pk, &ctx->d_plane, &ctx->d_plane16, &ctx->d_lists, &ctx->d_lists_out, &ctx->d_order, &ctx->d_sel, &ctx->d_seeds, &ctx->d_cnt_block};
    for (DevBuf* b : all) release(*b);
    (void)c2_comm_destroy(ctx);
    for (int k = 0; k < 2; ++k) {
        if (ctx->pin_in[k]) (void)hipHostFree(ctx->pin_in[k]);
        if (ctx->pin_out[k]) (void)hipHostFree(ctx->pin_out[k]);
        if (ctx->ev_in[k]) (void)hipEventDestroy(ctx->ev_in[k]);
        if (ctx->ev_done[k]) (void)hipEventDestroy(ctx->ev_done[k]);
        if (ctx->ev_out[k]) (void)hipEventDestroy(ctx->ev_out[k]);
    }
    if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
    if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int c2_set_scoring(c2_ctx* ctx, const int64_t* matrix, int32_t mat_dim, int32_t gap_open, int32_t gap_extend) {
    if (!ctx) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nel = (size_t)mat_dim * (size_t)mat_dim;
    const bool same = ctx->have_scoring && ctx->matrix_copy.size() == nel && matrix &&
                      memcmp(ctx->matrix_copy.data(), matrix, nel * sizeof(int64_t)) == 0;
    if (ctx->gap_open != gap_open || ctx->gap_extend != gap_extend) { ctx->diag_rows_dirty = true; ctx->pk_dirty = true; }
    ctx->gap_open = gap_open; ctx->gap_extend = gap_extend;
    if (same) return 0;
    c2_scoring_tables sc;
    if (!c2_build_scoring(matrix, mat_dim, sc, ctx->err)) return C2_E_INVALID;
    int rc;
    if ((rc = ensure(ctx, ctx->d_tbl, sc.tbl.size() * sizeof(int16_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_code, 256))) return rc;
    if ((rc = ensure(ctx, ctx->d_pk, C2_MAX_CODES * sizeof(uint32_t)))) return rc;
    // make sure no launch still reads the old tables -- launches may be on any stream of the caller's, so wait for the device
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(ctx->d_tbl.p, sc.tbl.data(), sc.tbl.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_code.p, sc.code_of_char, 256, hipMemcpyHostToDevice));
    if (!sc.pk.empty()) HIPCHK(ctx, hipMemcpy(ctx->d_pk.p, sc.pk.data(), sc.pk.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->sc = sc;
    ctx->matrix_copy.assign(matrix, matrix + nel);
    ctx->have_scoring = true;
    ctx->diag_rows_dirty = true; ctx->pk_dirty = true;
    return 0;
}

int c2_set_refs(c2_ctx* ctx, int32_t n_refs, const char* const* seqs, const int32_t* lens,
                const int64_t* const* gap_incentives, const int32_t* const* include_idx, const int32_t* n_include) {
    if (!ctx || n_refs <= 0 || n_refs > 65535 || !seqs || !lens || !gap_incentives) { if (ctx) ctx->err = "bad reference arguments"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // one blob: per ref [seq | pad][gap_incentive int32 x (L+1)][inc_prefix uint16 x (L+2)][seq2: 2-bit codes, 16 per word, padded]
    std::vector<uint8_t> blob;
    std::vector<size_t> off_seq(n_refs), off_g(n_refs), off_p(n_refs), off_s2(n_refs);
    int max_li = 0;
    for (int r = 0; r < n_refs; ++r) {
        const int L = lens[r];
        if (L < 0 || L > 60000) { ctx->err = "reference length out of range"; return C2_E_INVALID; }
        max_li = std::max(max_li, L);
        auto align = [&](size_t a) { blob.resize((blob.size() + a - 1) / a * a); };
        align(16); off_seq[r] = blob.size(); blob.insert(blob.end(), (const uint8_t*)seqs[r], (const uint8_t*)seqs[r] + L);
        align(16); off_g[r] = blob.size(); blob.resize(blob.size() + (size_t)(L + 1) * 4);
        int32_t* g32 = (int32_t*)(blob.data() + off_g[r]);
        for (int k = 0; k <= L; ++k) g32[k] = (int32_t)gap_incentives[r][k];   // the reference adds it into C ints
        std::vector<uint16_t> ip;
        c2_build_inc_prefix(include_idx ? include_idx[r] : nullptr, (include_idx && n_include) ? n_include[r] : 0, L, ip);
        align(16); off_p[r] = blob.size();
        blob.insert(blob.end(), (const uint8_t*)ip.data(), (const uint8_t*)(ip.data() + ip.size()));
        std::vector<uint32_t> s2;
        c2_build_seq2(seqs[r], L, s2);
        align(16); off_s2[r] = blob.size();
        blob.insert(blob.end(), (const uint8_t*)s2.data(), (const uint8_t*)(s2.data() + s2.size()));
    }
    int rc;
    HIPCHK(ctx, hipDeviceSynchronize());      // earlier launches (on whatever stream the caller used) may still read the old tables
    if ((rc = ensure(ctx, ctx->d_refblob, blob.size()))) return rc;
    if ((rc = ensure(ctx, ctx->d_refdesc, sizeof(c2_dev_ref) * (size_t)n_refs))) return rc;
    std::vector<c2_dev_ref> desc(n_refs);
    ctx->ref_len.resize(n_refs);
    ctx->gmax = 0;
    for (int r = 0; r < n_refs; ++r) {
        uint8_t* base = (uint8_t*)ctx->d_refblob.p;
        desc[r].seq = base + off_seq[r];
        desc[r].gap_incentive = (const int32_t*)(base + off_g[r]);
        desc[r].inc_prefix = (const uint16_t*)(base + off_p[r]);
        desc[r].seq2 = (const uint32_t*)(base + off_s2[r]) + 2;
        desc[r].len = lens[r];
        desc[r].diag_rows = nullptr; desc[r].pk_ok = 0; desc[r].first_incentive_pos = -1; desc[r].diag_kmax = -1; desc[r].reserved_pad = 0;
        for (int k = 0; k < 4; ++k) desc[r].diag_mmax[k] = -1;
        for (int i = 0; i <= lens[r]; ++i) if (gap_incentives[r][i] > 0) { desc[r].first_incentive_pos = i; break; }
        int64_t gm = 0;                                          // over the values the kernels add: the reference's C ints
        for (int k = 0; k <= lens[r]; ++k) gm = std::max<int64_t>(gm, (int64_t)(int32_t)gap_incentives[r][k]);
        desc[r].gap_incentive_max = (int32_t)std::min<int64_t>(gm, 1 << 20);
        desc[r].gap_incentive_last_pos = (int32_t)gap_incentives[r][lens[r]] > 0 ? 1 : 0;
        { int mc = 0; for (int k = 0; k < lens[r]; ++k) mc = std::max(mc, (int)(unsigned char)seqs[r][k]); desc[r].max_char = mc; }
        ctx->gmax = std::max(ctx->gmax, desc[r].gap_incentive_max);
        ctx->ref_len[r] = lens[r];
    }
    HIPCHK(ctx, hipMemcpy(ctx->d_refblob.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_refdesc.p, desc.data(), sizeof(c2_dev_ref) * (size_t)n_refs, hipMemcpyHostToDevice));
    ctx->n_refs = n_refs;
    ctx->max_li = std::max(max_li, 1);
    ctx->ref_seq.assign(n_refs, std::string());
    ctx->ref_g32.assign(n_refs, std::vector<int32_t>());
    for (int r = 0; r < n_refs; ++r) {
        ctx->ref_seq[r].assign(seqs[r], seqs[r] + lens[r]);
        ctx->ref_g32[r].resize(lens[r] + 1);
        for (int k = 0; k <= lens[r]; ++k) ctx->ref_g32[r][k] = (int32_t)gap_incentives[r][k];
    }
    ctx->ref_desc = desc;
    ctx->diag_rows_dirty = true; ctx->pk_dirty = true;
    return 0;
}

int c2_launch_info(c2_ctx* ctx, int32_t max_read_len, int32_t* rows_per_lane, int32_t* passes, int32_t* lds_bytes,
                   int32_t* workgroups_per_cu, int32_t* compute_units) {
    if (!ctx) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Geometry g;
    int rc = geometry(ctx, max_read_len, g);
    if (rc) return rc;
    if (rows_per_lane) *rows_per_lane = g.R;
    if (passes) *passes = g.passes;
    if (lds_bytes) *lds_bytes = (int32_t)(g.diag ? g.lds_diag : g.band_lanes ? g.lds_band : g.lds_full);
    if (workgroups_per_cu) *workgroups_per_cu = g.diag ? g.blocks_diag : g.band_lanes ? g.blocks_band : g.blocks_full;
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    return 0;
}

int c2_set_band(c2_ctx* ctx, int32_t band_lanes, int32_t target_workgroups_per_cu) {
    if (!ctx) return C2_E_INVALID;
    ctx->band_setting = band_lanes;
    if (target_workgroups_per_cu > 0) ctx->band_target_wgs = target_workgroups_per_cu;
    return 0;
}

int c2_set_kernel_mode(c2_ctx* ctx, int32_t mode) {
    if (!ctx || mode < 0 || mode > 5) return C2_E_INVALID;
    ctx->kernel_mode = mode;
    return 0;
}

int c2_band_info(c2_ctx* ctx, int32_t max_read_len, int32_t* band_lanes, int32_t* fallback_tasks_last_launch) {
    if (!ctx) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Geometry g;
    int rc = geometry(ctx, max_read_len, g);
    if (rc) return rc;
    if (band_lanes) *band_lanes = g.diag ? -1 : g.band_lanes;
    if (fallback_tasks_last_launch) {
        *fallback_tasks_last_launch = 0;
        if (ctx->d_fb.p && ctx->last_tiers > 0) {
            HIPCHK(ctx, hipDeviceSynchronize());
            uint32_t c = 0;
            HIPCHK(ctx, hipMemcpy(&c, (uint32_t*)ctx->d_fb.p + (ctx->last_tiers - 1), 4, hipMemcpyDeviceToHost));
            *fallback_tasks_last_launch = (int32_t)c;
        }
    }
    return 0;
}

// The score-only stage of the most recent batch: did it run, how many tasks did the partition give it, how many did it finish.
int c2_score_stage_info(c2_ctx* ctx, int32_t* ran, int64_t* tasks, int64_t* finished) {
    if (!ctx || !ran || !tasks || !finished) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *ran = ctx->last_score_stage ? 1 : 0; *tasks = 0; *finished = 0;
    if (ctx->last_score_stage && ctx->d_fb.p) {
        HIPCHK(ctx, hipDeviceSynchronize());
        uint32_t c[64];
        HIPCHK(ctx, hipMemcpy(c, (const uint32_t*)ctx->d_fb.p, 256, hipMemcpyDeviceToHost));
        // the first tier's list held the partition's class 2 before the launch; what the launch could not finish was appended to it
        *tasks = (int64_t)c[60]; *finished = (int64_t)c[60] - ((int64_t)c[56] - (int64_t)c[50]);
    }
    return 0;
}

// The partition of the most recent batch (c2_align_partition_kernel): did it run; tasks per class (0: score-only launch, 1: 14-diagonal launch,
// 2: first band tier, 3: second, 4: third); how many of their tasks the score-only launch and the 14-diagonal launch finished.
int c2_partition_info(c2_ctx* ctx, int32_t* ran, int64_t* class_tasks7, int64_t* finished2) {
    if (!ctx || !ran || !class_tasks7 || !finished2) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *ran = ctx->last_score_stage ? (ctx->last_p16_stage ? 3 : 1) : 0;
    for (int k = 0; k < 7; ++k) class_tasks7[k] = 0;
    finished2[0] = finished2[1] = 0;
    if (ctx->last_score_stage && ctx->d_fb.p) {
        HIPCHK(ctx, hipDeviceSynchronize());
        uint32_t c[64];
        HIPCHK(ctx, hipMemcpy(c, (const uint32_t*)ctx->d_fb.p, 256, hipMemcpyDeviceToHost));
        for (int k = 0; k < 7; ++k) class_tasks7[k] = (int64_t)c[48 + k];
        finished2[0] = (int64_t)c[60] - ((int64_t)c[56] - (int64_t)c[50]);
        finished2[1] = ctx->last_p16_stage ? (int64_t)c[62] - ((int64_t)c[57] - (int64_t)c[56]) : 0;
    }
    return 0;
}

// ... and how many class-0 tasks the partition finished itself: reads on the main diagonal with at most two differing bases (c2_main_diagonal_certificate)
int c2_partition_finished(c2_ctx* ctx, int64_t* n) {
    if (!ctx || !n) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *n = 0;
    if (ctx->last_score_stage && ctx->d_fb.p) {
        HIPCHK(ctx, hipDeviceSynchronize());
        uint32_t c = 0;
        HIPCHK(ctx, hipMemcpy(&c, (const uint32_t*)ctx->d_fb.p + 55, 4, hipMemcpyDeviceToHost));
        *n = (int64_t)c;
    }
    return 0;
}

int c2_tier_info(c2_ctx* ctx, int32_t* n_tiers, int32_t* left_over4) {
    if (!ctx || !n_tiers || !left_over4) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *n_tiers = ctx->last_tiers;
    for (int k = 0; k < 4; ++k) left_over4[k] = 0;
    if (ctx->d_fb.p && ctx->last_tiers > 0) {
        HIPCHK(ctx, hipDeviceSynchronize());
        uint32_t c[4] = {0, 0, 0, 0};
        HIPCHK(ctx, hipMemcpy(c, ctx->d_fb.p, 16, hipMemcpyDeviceToHost));
        for (int k = 0; k < ctx->last_tiers && k < 4; ++k) left_over4[k] = (int32_t)c[k];
    }
    return 0;
}

// Which kernels the launch chain of a batch with reads up to max_read_len consists of, and which references the packed fill admits.
int c2_chain_info(c2_ctx* ctx, int32_t max_read_len, uint32_t* kernels, uint8_t* ref_packed_ok) {
    if (!ctx || !kernels) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Geometry g;
    int rc = geometry(ctx, max_read_len, g);
    if (rc) return rc;
    *kernels = (g.pk ? 1u : 0u) | (g.x[0] ? 2u : 0u) | (g.pk2 ? 4u : 0u) | (g.x[1] ? 8u : 0u) | (g.pk3 ? 16u : 0u) | (g.diag ? 32u : 0u) |
               (g.band_lanes > 0 ? 64u : 0u) | (g.full_hbm ? 128u : 0u) | ((g.pk && ctx->pk_beta > 0) ? 256u : 0u) |
               ((g.pk && ctx->kernel_mode == 0 && !getenv("C2_NO_SCORE_TIER")) ? 512u : 0u);
    if (ref_packed_ok) for (int r = 0; r < ctx->n_refs; ++r) ref_packed_ok[r] = ctx->ref_pk_ok[r];
    return 0;
}

// c2_tier_info plus, per band tier, the tasks its packed kernel could not pair (handed to the 32-bit kernel of the same band).  A tier
// with a packed kernel therefore finished at least  tasks_in - unpaired - left_over  of its tasks in int16 arithmetic.
int c2_tier_info_ex(c2_ctx* ctx, int32_t* n_tiers, int32_t* left_over8, int32_t* unpaired8) {
    if (!ctx || !n_tiers || !left_over8 || !unpaired8) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *n_tiers = ctx->last_tiers;
    for (int k = 0; k < 8; ++k) left_over8[k] = unpaired8[k] = 0;
    if (ctx->d_fb.p && ctx->last_tiers > 0) {
        HIPCHK(ctx, hipDeviceSynchronize());
        uint32_t c[16];
        HIPCHK(ctx, hipMemcpy(c, ctx->d_fb.p, 64, hipMemcpyDeviceToHost));
        for (int k = 0; k < ctx->last_tiers && k < 8; ++k) { left_over8[k] = (int32_t)c[k]; unpaired8[k] = (int32_t)c[8 + k]; }
    }
    return 0;
}

int c2_timing_enable(c2_ctx* ctx, int on) { if (!ctx) return C2_E_INVALID; ctx->timing = on != 0; return 0; }

int c2_timing_read_split(c2_ctx* ctx, double* total_ms, double* first_kernel_ms, int64_t* launches, int reset) {
    if (!ctx) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double tot = 0, first = 0;
    for (auto& t : ctx->timed) {
        HIPCHK(ctx, hipEventSynchronize(t.b));
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, t.a, t.b));
        tot += ms;
        HIPCHK(ctx, hipEventElapsedTime(&ms, t.m0, t.m));
        first += ms;
    }
    if (total_ms) *total_ms = tot;
    if (first_kernel_ms) *first_kernel_ms = first;
    if (launches) *launches = (int64_t)ctx->timed.size();
    if (reset) {
        for (auto& t : ctx->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.m0); (void)hipEventDestroy(t.m); (void)hipEventDestroy(t.b); }
        ctx->timed.clear();
    }
    return 0;
}

int c2_timing_read(c2_ctx* ctx, double* total_ms, int64_t* launches, int reset) {
    return c2_timing_read_split(ctx, total_ms, nullptr, launches, reset);
}

int c2_synchronize(c2_ctx* ctx, void* hip_stream) {
    if (!ctx) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)hip_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int c2_align_classify_batch_device(c2_ctx* ctx, const c2_batch* b, void* hip_stream) {
    if (!ctx || !b) return C2_E_INVALID;
    if (b->n_reads && (!b->reads || !b->offsets || !b->aln_read || !b->aln_ref || !b->records)) { ctx->err = "NULL batch pointer"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // device-resident offsets: the host cannot see the longest read, the caller states it (max_read_len);
    // reads longer than that are reported per record as C2_STATUS_TOO_LONG, never overrun the LDS plan.
    const int max_lj = b->max_read_len > 0 ? b->max_read_len : (int)b->aln_stride - ctx->max_li;
    if (max_lj < 1) { ctx->err = "aln_stride must be at least longest reference + longest read"; return C2_E_INVALID; }
    return run_align(ctx, b, max_lj, (hipStream_t)hip_stream);
}

namespace {

// Host batch, pipelined.  The batch is cut into chunks of reads; for chunk c, concurrently:
//   host threads   reads of chunk c+1 -> pinned input set;   pinned output set of chunk c-1 -> the caller's arrays
//   stream s_in    pinned input of chunk c+1 -> device
//   ctx->stream    launch chain of chunk c (after its input arrived)
//   stream s_out   outputs of chunk c-1 -> pinned output set
// Offsets / reference ids / strands are small and go up front.  Two pinned sets each way; events order their reuse.
int align_host_pipelined(c2_ctx* ctx, const c2_batch* b, int max_lj, uint64_t chunk_reads, int32_t min_lj) {
    const uint64_t n = b->n_reads;
    const uint64_t tpr = (uint64_t)(b->all_refs ? ctx->n_refs : 1);          // tasks per read
    const uint64_t n_tasks = n * tpr;
    const uint64_t stride = b->aln_stride;
    const uint64_t base0 = b->offsets[0];
    const uint64_t nbytes = b->offsets[n] - base0;
    int rc;
    if ((rc = ensure(ctx, ctx->d_reads, nbytes + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_offsets, (n + 1) * 8))) return rc;
    if (b->ref_ids && !b->all_refs) if ((rc = ensure(ctx, ctx->d_refids, n * 2))) return rc;
    if (b->strands) if ((rc = ensure(ctx, ctx->d_strands, n_tasks))) return rc;
    if ((rc = ensure(ctx, ctx->d_aln_read, n_tasks * stride))) return rc;
    if ((rc = ensure(ctx, ctx->d_aln_ref, n_tasks * stride))) return rc;
    if ((rc = ensure(ctx, ctx->d_records, n_tasks * sizeof(c2_aln_record)))) return rc;
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
    // chunk boundaries and the largest chunk, in input bytes and in tasks
    const uint64_t n_chunks = (n + chunk_reads - 1) / chunk_reads;
    auto c_lo = [&](uint64_t c) { return std::min<uint64_t>(n, c * chunk_reads); };
    uint64_t max_in = 0;
    for (uint64_t c = 0; c < n_chunks; ++c) max_in = std::max<uint64_t>(max_in, b->offsets[c_lo(c + 1)] - b->offsets[c_lo(c)]);
    const uint64_t max_tasks = std::min<uint64_t>(n, chunk_reads) * tpr;
    const size_t out_bytes = (size_t)(max_tasks * (2 * stride + sizeof(c2_aln_record)));
    for (int k = 0; k < 2; ++k) {
        if ((rc = ensure_pinned(ctx, ctx->pin_in[k], ctx->pin_in_cap[k], (size_t)max_in + 16))) return rc;
        if ((rc = ensure_pinned(ctx, ctx->pin_out[k], ctx->pin_out_cap[k], out_bytes))) return rc;
    }
    // small arrays up front (pageable copies on the compute stream; `rel` must outlive them)
    std::vector<uint64_t> rel(n + 1);
    for (uint64_t k = 0; k <= n; ++k) rel[k] = b->offsets[k] - base0;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_offsets.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
    if (b->ref_ids && !b->all_refs) HIPCHK(ctx, hipMemcpyAsync(ctx->d_refids.p, b->ref_ids, n * 2, hipMemcpyHostToDevice, s));
    if (b->strands) HIPCHK(ctx, hipMemcpyAsync(ctx->d_strands.p, b->strands, n_tasks, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipStreamSynchronize(s));

    auto stage_in = [&](uint64_t c) -> int {                      // host: reads of chunk c -> pinned set; s_in: -> device
        const int k = (int)(c & 1);
        const uint64_t a = rel[c_lo(c)], z = rel[c_lo(c + 1)];
        if (c >= 2) HIPCHK(ctx, hipEventSynchronize(ctx->ev_in[k]));          // the set's previous copy (chunk c-2) has left it
        copy_parallel(ctx->pin_in[k], b->reads + base0 + a, (size_t)(z - a), threads);
        if (z > a) HIPCHK(ctx, hipMemcpyAsync((uint8_t*)ctx->d_reads.p + a, ctx->pin_in[k], (size_t)(z - a), hipMemcpyHostToDevice, ctx->s_in));
        HIPCHK(ctx, hipEventRecord(ctx->ev_in[k], ctx->s_in));
        return 0;
    };
    auto drain_out = [&](uint64_t c) -> int {                      // host: pinned outputs of chunk c -> the caller's arrays
        const int k = (int)(c & 1);
        const uint64_t t0 = c_lo(c) * tpr, nt = (c_lo(c + 1) - c_lo(c)) * tpr;
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_out[k]));
        const uint8_t* po = (const uint8_t*)ctx->pin_out[k];
        copy_parallel(b->aln_read + t0 * stride, po, (size_t)(nt * stride), threads);
        copy_parallel(b->aln_ref + t0 * stride, po + max_tasks * stride, (size_t)(nt * stride), threads);
        memcpy(b->records + t0, po + 2 * max_tasks * stride, (size_t)(nt * sizeof(c2_aln_record)));
        return 0;
    };
    if ((rc = stage_in(0))) return rc;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const int k = (int)(c & 1);
        const uint64_t r0 = c_lo(c), nr = c_lo(c + 1) - r0, t0 = r0 * tpr, nt = nr * tpr;
        // launch chain of chunk c, after its reads arrived
        HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_in[k], 0));
        c2_batch d = *b;
        d.diag_hints = nullptr;                                          // (a device-route output)
        d.n_reads = nr; d.min_read_len = min_lj;
        d.reads = (const uint8_t*)ctx->d_reads.p; d.offsets = (const uint64_t*)ctx->d_offsets.p + r0;
        d.ref_ids = (b->ref_ids && !b->all_refs) ? (const uint16_t*)ctx->d_refids.p + r0 : nullptr;
        d.strands = b->strands ? (const uint8_t*)ctx->d_strands.p + t0 : nullptr;
        d.aln_read = (uint8_t*)ctx->d_aln_read.p + t0 * stride; d.aln_ref = (uint8_t*)ctx->d_aln_ref.p + t0 * stride;
        d.records = (c2_aln_record*)ctx->d_records.p + t0;
        if ((rc = run_align(ctx, &d, max_lj, s))) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_done[k], s));
        // the next chunk's reads travel while this one computes
        if (c + 1 < n_chunks && (rc = stage_in(c + 1))) return rc;
        // outputs of chunk c -> pinned set k (free once chunk c-2 was drained, which happened below in iteration c-1)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->s_out, ctx->ev_done[k], 0));
        uint8_t* po = (uint8_t*)ctx->pin_out[k];
        HIPCHK(ctx, hipMemcpyAsync(po, d.aln_read, (size_t)(nt * stride), hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipMemcpyAsync(po + max_tasks * stride, d.aln_ref, (size_t)(nt * stride), hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipMemcpyAsync(po + 2 * max_tasks * stride, d.records, (size_t)(nt * sizeof(c2_aln_record)), hipMemcpyDeviceToHost, ctx->s_out));
        HIPCHK(ctx, hipEventRecord(ctx->ev_out[k], ctx->s_out));
        // meanwhile: the previous chunk's outputs go to the caller
        if (c >= 1 && (rc = drain_out(c - 1))) return rc;
    }
    if ((rc = drain_out(n_chunks - 1))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(s));
    return 0;
}

}  // namespace

int c2_align_classify_batch_host(c2_ctx* ctx, const c2_batch* b) {
    if (!ctx || !b) return C2_E_INVALID;
    if (b->n_reads == 0) return 0;
    if (!b->reads || !b->offsets || !b->aln_read || !b->aln_ref || !b->records) { ctx->err = "NULL batch pointer"; return C2_E_INVALID; }
    if (!ctx->have_scoring || ctx->n_refs <= 0) { ctx->err = "scoring and references must be set first"; return C2_E_STATE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = b->n_reads;
    const uint64_t n_tasks = n * (uint64_t)(b->all_refs ? ctx->n_refs : 1);
    int max_lj = 1;
    uint64_t min_len = ~0ull;                                            // (the host sees the lengths: the hint c2_batch.min_read_len is computed here, whatever the caller put there)
    for (uint64_t k = 0; k < n; ++k) {
        if (b->offsets[k + 1] < b->offsets[k]) { ctx->err = "offsets must be non-decreasing"; return C2_E_INVALID; }
        max_lj = (int)std::max<uint64_t>(max_lj, b->offsets[k + 1] - b->offsets[k]);
        min_len = std::min<uint64_t>(min_len, b->offsets[k + 1] - b->offsets[k]);
    }
    const int32_t min_lj = (n && min_len <= 0x7fffffffull) ? (int32_t)min_len : 0;
    if (!b->all_refs && b->ref_ids)
        for (uint64_t k = 0; k < n; ++k) if (b->ref_ids[k] >= ctx->n_refs) { ctx->err = "ref_id out of range"; return C2_E_INVALID; }
    const uint32_t need_stride = (uint32_t)(ctx->max_li + max_lj);
    if (b->aln_stride < need_stride) { ctx->err = "aln_stride smaller than longest read + longest reference"; return C2_E_INVALID; }
    const uint64_t nbytes = b->offsets[n] - b->offsets[0];
    int rc;
    {   // large batches: chunks through pinned staging, copies both ways overlapped with the launch chains
        uint64_t pipe_min = 131072, chunk_tasks = 65536;
        if (const char* e = getenv("C2_HOST_PIPE_MIN_TASKS")) pipe_min = strtoull(e, nullptr, 10);
        if (const char* e = getenv("C2_HOST_PIPE_CHUNK_TASKS")) chunk_tasks = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        const uint64_t tpr = (uint64_t)(b->all_refs ? ctx->n_refs : 1);
        if (n_tasks >= pipe_min && n >= 2) return align_host_pipelined(ctx, b, max_lj, std::max<uint64_t>(1, chunk_tasks / tpr), min_lj);
    }
    if ((rc = ensure(ctx, ctx->d_reads, nbytes + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_offsets, (n + 1) * 8))) return rc;
    if (b->ref_ids && !b->all_refs) if ((rc = ensure(ctx, ctx->d_refids, n * 2))) return rc;
    if (b->strands) if ((rc = ensure(ctx, ctx->d_strands, n_tasks))) return rc;
    if ((rc = ensure(ctx, ctx->d_aln_read, n_tasks * (uint64_t)b->aln_stride))) return rc;
    if ((rc = ensure(ctx, ctx->d_aln_ref, n_tasks * (uint64_t)b->aln_stride))) return rc;
    if ((rc = ensure(ctx, ctx->d_records, n_tasks * sizeof(c2_aln_record)))) return rc;
    hipStream_t s = ctx->stream;
    std::vector<uint64_t> rel(n + 1);
    for (uint64_t k = 0; k <= n; ++k) rel[k] = b->offsets[k] - b->offsets[0];
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_reads.p, b->reads + b->offsets[0], nbytes, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_offsets.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
    if (b->ref_ids && !b->all_refs) HIPCHK(ctx, hipMemcpyAsync(ctx->d_refids.p, b->ref_ids, n * 2, hipMemcpyHostToDevice, s));
    if (b->strands) HIPCHK(ctx, hipMemcpyAsync(ctx->d_strands.p, b->strands, n_tasks, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipStreamSynchronize(s));      // `rel` is pageable host memory owned by this call
    c2_batch d = *b;
    d.min_read_len = min_lj;
    d.reads = (const uint8_t*)ctx->d_reads.p; d.offsets = (const uint64_t*)ctx->d_offsets.p;
    d.ref_ids = (b->ref_ids && !b->all_refs) ? (const uint16_t*)ctx->d_refids.p : nullptr;
    d.strands = b->strands ? (const uint8_t*)ctx->d_strands.p : nullptr;
    d.aln_read = (uint8_t*)ctx->d_aln_read.p; d.aln_ref = (uint8_t*)ctx->d_aln_ref.p; d.records = (c2_aln_record*)ctx->d_records.p;
    if ((rc = run_align(ctx, &d, max_lj, s))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(b->aln_read, d.aln_read, n_tasks * (uint64_t)b->aln_stride, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(b->aln_ref, d.aln_ref, n_tasks * (uint64_t)b->aln_stride, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(b->records, d.records, n_tasks * sizeof(c2_aln_record), hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return 0;
}

int c2_global_align(c2_ctx* ctx, const char* read, int32_t Lj, const char* ref, int32_t Li,
                    const int64_t* matrix, int32_t mat_dim, const int64_t* gap_incentive, int32_t n_gap_incentive,
                    int32_t gap_open, int32_t gap_extend, char* out_read_aln, char* out_ref_aln,
                    int32_t* out_len, int32_t* out_matches, int32_t* out_status) {
    if (!ctx || !read || !ref || !matrix || !gap_incentive || !out_read_aln || !out_ref_aln || !out_len || !out_matches || !out_status) {
        if (ctx) ctx->err = "NULL argument";
        return C2_E_INVALID;
    }
    *out_len = 0; *out_matches = 0; *out_status = 0;
    if (n_gap_incentive != Li + 1) { *out_status = -1; return 0; }          // CRISPResso2Align.pyx:124-126
    if (Li <= 0 || Lj <= 0) { *out_status = C2_STATUS_EMPTY; return 0; }      // undefined in the reference
    int rc;
    if ((rc = c2_set_scoring(ctx, matrix, mat_dim, gap_open, gap_extend))) return rc;
    const char* seqs[1] = {ref};
    const int32_t lens[1] = {Li};
    const int64_t* gis[1] = {gap_incentive};
    const int32_t* incs[1] = {nullptr};
    const int32_t ninc[1] = {0};
    if ((rc = c2_set_refs(ctx, 1, seqs, lens, gis, incs, ninc))) return rc;
    const uint32_t stride = (uint32_t)((Li + Lj + 15) / 16 * 16);
    std::vector<uint8_t> o1(stride), o2(stride);
    c2_aln_record rec;
    memset(&rec, 0, sizeof rec);
    const uint64_t offs[2] = {0, (uint64_t)Lj};
    c2_batch b;
    memset(&b, 0, sizeof b);
    b.n_reads = 1; b.reads = (const uint8_t*)read; b.offsets = offs;
    b.aln_read = o1.data(); b.aln_ref = o2.data(); b.aln_stride = stride; b.records = &rec;
    if ((rc = c2_align_classify_batch_host(ctx, &b))) return rc;
    *out_status = rec.status;
    if (rec.status == 0) {
        memcpy(out_read_aln, o1.data(), rec.aln_len);
        memcpy(out_ref_aln, o2.data(), rec.aln_len);
        *out_len = rec.aln_len; *out_matches = rec.matches;
    }
    return 0;
}

int c2_phase_profile(c2_ctx* ctx, int enable, uint64_t* out4) {
    if (!ctx) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->d_phase, 4 * sizeof(uint64_t)))) return rc;
    HIPCHK(ctx, hipDeviceSynchronize());
    if (out4) HIPCHK(ctx, hipMemcpy(out4, ctx->d_phase.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemset(ctx->d_phase.p, 0, 4 * sizeof(uint64_t)));
    ctx->phase_prof = enable != 0;
    return 0;
}

int c2_selftest(c2_ctx* ctx, int32_t* out448) {
    if (!ctx || !out448) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->d_misc, 448 * 4))) return rc;
    hipLaunchKernelGGL(c2_selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, (int*)ctx->d_misc.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out448, ctx->d_misc.p, 448 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int c2_selftest_rows(c2_ctx* ctx, int32_t* out128) {
    if (!ctx || !out128) return C2_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->d_misc, 128 * 4))) return rc;
    hipLaunchKernelGGL(c2_selftest_rows_kernel, dim3(1), dim3(64), 0, ctx->stream, (int*)ctx->d_misc.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out128, ctx->d_misc.p, 128 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
