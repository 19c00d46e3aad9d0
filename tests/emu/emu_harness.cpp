// Wave-emulator harness -- TEST INFRASTRUCTURE ONLY.
// Compiles crispresso2_amd/csrc/c2_kernels.hip unchanged with g++ (hip/hip_runtime.h is the shim in
// this directory) and runs it on 64 cooperative fibers.  tests/test_kernel_emulated.py drives it
// through ctypes and compares with the oracle.  Not part of the product; never used as a fallback.
#include "emu_runtime.h"
#include <vector>
#include <string>
#include <algorithm>
#include <string.h>
#include "../../crispresso2_amd/csrc/c2_host_prep.h"

alignas(16) unsigned char c2_smem[163840];
#include "../../crispresso2_amd/csrc/c2_kernels.hip"

static int g_last_pk_beta = 0, g_last_pk_bias = 0;   // the last emu_align_batch's 32-bit-add parameters (0: packed adds)
static unsigned g_last_unpaired = 0;     // tasks the first packed tier of the last emu_align_batch could not pair

extern "C" {

unsigned emu_last_unpaired() { return g_last_unpaired; }
static const uint32_t* g_last_classes = nullptr;   // tasks per class of the last emu_align_batch's partition (all zero: it did not run)
static unsigned g_last_p16_finished = 0;           // tasks its 14-diagonal launch finished
void emu_last_partition(uint32_t* classes7, uint32_t* p16_finished) {
    for (int k = 0; k < 7; ++k) classes7[k] = g_last_classes ? g_last_classes[k] : 0u;
    *p16_finished = g_last_p16_finished;
}
unsigned emu_last_exact_copies() { return g_last_classes ? g_last_classes[7] : 0u; }   // class-0 reads equal to their reference, finished by the partition
int emu_last_pk_beta() { return g_last_pk_beta; }
int emu_last_pk_bias() { return g_last_pk_bias; }

// c2_batch.diag_hints of the NEXT emu_align_batch (n_tasks words, or NULL); c2_count_args.hints of the NEXT emu_count_vectors
static uint32_t* g_next_hints_out = nullptr;
static const uint32_t* g_next_count_hints = nullptr;
void emu_set_hints_out(uint32_t* p) { g_next_hints_out = p; }
void emu_set_count_hints(const uint32_t* p) { g_next_count_hints = p; }

int emu_align_batch(uint64_t n_reads, const uint8_t* reads, const uint64_t* offsets, const uint16_t* ref_ids,
                    const uint8_t* strands, int all_refs,
                    int n_refs, const char* const* seqs, const int32_t* lens, const int64_t* const* gap_inc,
                    const int32_t* const* include_idx, const int32_t* n_include,
                    const int64_t* matrix, int dim, int go, int ge,
                    uint8_t* aln_read, uint8_t* aln_ref, uint32_t aln_stride, c2_aln_record* records,
                    int force_R, unsigned grid, int no_packed, int band_lanes, int* n_fallback)
{
    c2_scoring_tables sc; std::string err;
    if (!c2_build_scoring(matrix, dim, sc, err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return -1; }
    std::vector<c2_dev_ref> refs(n_refs);
    std::vector<std::vector<int32_t>> g32(n_refs);
    std::vector<std::vector<uint16_t>> incp(n_refs);
    std::vector<std::vector<uint32_t>> seq2(n_refs);
    std::vector<std::vector<c2_diag_row>> drows(n_refs), drows_pk(n_refs);
    bool any_pk = false;
    int max_li = 1;
    for (int r = 0; r < n_refs; ++r) {
        g32[r].resize(lens[r] + 1);
        for (int k = 0; k <= lens[r]; ++k) g32[r][k] = (int32_t)gap_inc[r][k];
        c2_build_inc_prefix(include_idx[r], n_include[r], lens[r], incp[r]);
        refs[r].seq = (const uint8_t*)seqs[r]; refs[r].gap_incentive = g32[r].data(); refs[r].inc_prefix = incp[r].data();
        c2_build_seq2(seqs[r], lens[r], seq2[r]);
        refs[r].seq2 = getenv("C2_EMU_NO_SEQ2") ? nullptr : seq2[r].data() + 2;
        c2_build_diag_rows(seqs[r], lens[r], g32[r].data(), sc, go, ge, drows[r]);
        refs[r].diag_rows = (drows[r].empty() || no_packed) ? nullptr : drows[r].data() + C2_DIAG_ROW_PAD;
        refs[r].pk_ok = (!no_packed && c2_pk_eligible(seqs[r], lens[r], g32[r].data(), sc, go, ge, 126)) ? 1 : 0; refs[r].first_incentive_pos = -1;
        for (int i = 0; i <= lens[r]; ++i) if (g32[r][i] > 0) { refs[r].first_incentive_pos = i; break; }
        {
            c2_diag_cert dc;
            dc.kmax = -1; dc.mmax[0] = dc.mmax[1] = dc.mmax[2] = dc.mmax[3] = -1;
            if (refs[r].pk_ok && !getenv("C2_NO_EXACT_COPIES")) dc = c2_main_diagonal_certificate(seqs[r], lens[r], g32[r].data(), sc, go, ge);
            if (const char* e = getenv("C2_DIAG_CERT_KMAX")) dc.kmax = std::min(dc.kmax, atoi(e));
            if (getenv("C2_EMU_TRACE")) fprintf(stderr, "ref %d: main-diagonal certificate kmax %d, mmax %d %d %d %d\n", r, dc.kmax, dc.mmax[0], dc.mmax[1], dc.mmax[2], dc.mmax[3]);
            refs[r].diag_kmax = dc.kmax;
            for (int k = 0; k < 4; ++k) refs[r].diag_mmax[k] = dc.mmax[k];
            refs[r].reserved_pad = 0;
        }
        if (refs[r].pk_ok) any_pk = true;
        refs[r].len = lens[r];
        int64_t gm = 0;
        for (int k = 0; k <= lens[r]; ++k) gm = std::max<int64_t>(gm, (int64_t)g32[r][k]);
        refs[r].gap_incentive_max = (int32_t)gm;
        refs[r].gap_incentive_last_pos = g32[r][lens[r]] > 0 ? 1 : 0;
        { int mc = 0; for (int k = 0; k < lens[r]; ++k) mc = std::max(mc, (int)(unsigned char)seqs[r][k]); refs[r].max_char = mc; }
        max_li = std::max(max_li, lens[r]);
    }
    // the library's choice (c2_api_align.hip, update_pk_eligibility): one bias for the batch, the 32-bit-add variant only if every admitted
    // reference stays in range with it (C2_EMU_NO_ADD32: the packed-add variant, as for references beyond that range)
    int pk_beta = 0, pk_bias = 0;
    if (any_pk && !getenv("C2_EMU_NO_ADD32")) {
        int beta = 0;
        for (int r = 0; r < n_refs; ++r) if (refs[r].pk_ok) beta = std::max(beta, c2_pk_beta_needed(lens[r], g32[r].data(), sc, go, ge));
        int bias = 0;
        for (int r = 0; r < n_refs; ++r) if (refs[r].pk_ok) bias = std::max(bias, c2_pk_add32_bias_needed(lens[r], g32[r].data(), sc, go, ge, 126));
        bool ok = beta > 0;
        for (int r = 0; r < n_refs && ok; ++r) if (refs[r].pk_ok) ok = c2_pk_add32_ok(lens[r], g32[r].data(), sc, go, ge, 126, beta, bias);
        if (ok) { pk_beta = beta; pk_bias = bias; }
    }
    g_last_pk_beta = pk_beta; g_last_pk_bias = pk_bias;
    for (int r = 0; r < n_refs; ++r) {
        if (refs[r].pk_ok) c2_build_diag_rows_pk(seqs[r], lens[r], g32[r].data(), sc, go, ge, drows_pk[r], pk_beta);
        else drows_pk[r].assign(drows[r].size(), c2_diag_row{0, 0, 0, 5u * C2_PK_LUT_STRIDE});
    }
    int max_lj = 1;
    for (uint64_t k = 0; k < n_reads; ++k) max_lj = std::max<int>(max_lj, (int)(offsets[k + 1] - offsets[k]));
    const int R = force_R ? force_R : c2_choose_rows_per_lane(max_li);
    uint32_t* hints = g_next_hints_out;
    g_next_hints_out = nullptr;
    if (hints) memset(hints, 0, (size_t)(n_reads * (uint64_t)(all_refs ? n_refs : 1)) * 4u * sizeof(uint32_t));     // (the library's hipMemsetAsync)
    c2_align_args A;
    A.reads = reads; A.offsets = offsets; A.ref_ids = ref_ids; A.strands = strands; A.refs = refs.data();
    A.score_tbl = sc.tbl.data(); A.code_of_char = sc.code_of_char;
    A.score_pk = (sc.pk.empty() || no_packed) ? nullptr : sc.pk.data();
    A.aln_read = aln_read; A.aln_ref = aln_ref; A.records = records;
    A.n_tasks = n_reads * (uint64_t)(all_refs ? n_refs : 1); A.aln_stride = aln_stride; A.n_refs = n_refs; A.all_refs = all_refs;
    A.n_codes = sc.n_codes; A.gap_open = go; A.gap_extend = ge; A.max_lj = max_lj;
    A.max_passes = (max_li + 64 * R - 1) / (64 * R);
    A.phase_cycles = nullptr;
    // banded first launch (if asked for and the reference fits one pass), then the full-plane launch over the fallback list
    std::vector<uint32_t> fb_list(A.n_tasks ? A.n_tasks : 1);
    uint32_t fb_count = 0;
    unsigned long long work_counter = 0;
    A.work_counter = &work_counter;
    A.band_lanes = 0; A.reserved = 0; A.fb_count = &fb_count; A.fb_list = fb_list.data(); A.task_list = nullptr; A.task_count = nullptr;
    if (grid == 0) grid = (unsigned)std::min<uint64_t>(A.n_tasks, 3);
    A.max_li = max_li; A.reserved = ((n_refs > 1 && !getenv("C2_NO_BLOCK_GRABS")) ? 8 : 0) | (getenv("C2_NO_PAIR_SORT") ? 64 : 0);      // (several references: blocks of groups from the work counter, as the host library sets it)
    {
        int mx = 0;
        for (int16_t v : sc.tbl) mx = std::max(mx, (int)v);
        A.max_score = mx;
    }
    // band_lanes: -1 single-alignment diagonal-band kernel, -2 / -4 the 2- / 4-alignments-per-wavefront kernel, -7 the whole
    // chain 4 -> 2 -> 1; every chain ends with the full-plane kernel over what is left (the host library's launch order)
    // -8 / -84: the packed kernels (8 / 4 per wavefront, int16 pairs) alone; -87: the library's default chain 8 (packed) -> 4 (packed) -> 1
    // (references the packed fill does not admit: 2 per wavefront in 32 bits instead)
    const bool diag = band_lanes == -1 || band_lanes == -2 || band_lanes == -4 || band_lanes == -5 || band_lanes == -7 || band_lanes == -75 || band_lanes == -8 || band_lanes == -84 || band_lanes == -87 || band_lanes == -82 || band_lanes == -86;
    const bool band = band_lanes > 0 && band_lanes < 32 && A.max_passes == 1;
    std::vector<uint32_t> fb_list2(A.n_tasks ? A.n_tasks : 1), fb_list3(A.n_tasks ? A.n_tasks : 1), fb_list4(A.n_tasks ? A.n_tasks : 1);   // (the full-plane launch below reads the last tier's)
    uint32_t fb_counts[5] = {0, 0, 0, 0, 0};
    std::vector<uint32_t> plane;
    A.plane = nullptr; A.plane_words_per_wg = 0; A.pk_beta = (uint32_t)pk_beta; A.pk_bias = (uint32_t)pk_bias; A.list_gate = 0; A.diag_hints = hints;
    A.un_list = nullptr; A.un_count = nullptr; A.pair_order = 0; A.legacy = getenv("C2_EMU_LEGACY") ? 1 : 0;
    A.mat_dim = sc.mat_dim; A.first_ext_code = sc.first_ext_code;
    c2_build_base_luts(sc, A.lut_code_lo, A.lut_code_hi, A.lut_chr_lo, A.lut_chr_hi);
    if (no_packed) A.lut_chr_lo = A.lut_chr_hi = 0xffffffffu;
    A.diag_base = nullptr; A.diagpk_base = nullptr;
    if (!no_packed && n_refs > 0 && !drows[0].empty()) {
        // the references' row tables must sit in one buffer (the kernels index it relative to diag_base)
        static std::vector<c2_diag_row> all_rows, all_rows_pk;
        all_rows.clear(); all_rows_pk.clear();
        std::vector<size_t> off(n_refs);
        for (int r = 0; r < n_refs; ++r) {
            off[r] = all_rows.size();
            all_rows.insert(all_rows.end(), drows[r].begin(), drows[r].end());
            all_rows_pk.insert(all_rows_pk.end(), drows_pk[r].begin(), drows_pk[r].end());
        }
        for (int r = 0; r < n_refs; ++r) refs[r].diag_rows = all_rows.data() + off[r] + C2_DIAG_ROW_PAD;
        A.diag_base = all_rows.data();
        A.diagpk_base = all_rows_pk.data();
    }
    if (diag) {
        uint32_t* lists[4] = {fb_list.data(), fb_list2.data(), fb_list3.data(), fb_list4.data()};       // (one per band tier: the partition writes into later tiers' lists)
        std::vector<uint32_t> un_list(A.n_tasks ? A.n_tasks : 1);
        uint32_t un_counts[5] = {0, 0, 0, 0, 0};
        int tier = 0;
        // the host library's wiring (c2_api_align.hip, launch_align): a band tier = the packed kernel (if any), then the 32-bit kernel of the
        // same band -- over everything, or over the tasks the packed kernel could not pair
        auto chain = [&](c2_align_args& T, const bool from_unpaired, const bool packed_kernel) {
            if (from_unpaired) { T.task_list = un_list.data(); T.task_count = &un_counts[tier]; }
            else { T.task_list = tier ? lists[tier - 1] : nullptr; T.task_count = tier ? &fb_counts[tier - 1] : nullptr; }
            T.fb_list = lists[tier]; T.fb_count = &fb_counts[tier];
            T.un_list = packed_kernel ? un_list.data() : nullptr; T.un_count = packed_kernel ? &un_counts[tier] : nullptr;
            T.pair_order = packed_kernel && !from_unpaired && tier == 0 && T.all_refs && T.n_refs > 1;
            work_counter = 0;
        };
        // the partition and the two launches in front of the first band tier, as the host library wires them (c2_api_align.hip: c2_align_partition_kernel,
        // then c2_align_diags_kernel over class 0 and c2_align_diagp_kernel<16> over class 1; what they cannot finish joins the first tier's list
        // (class 2); classes 3 and 4 go straight to the lists of the second and third tier).  C2_EMU_NO_SCORE_TIER=1 switches all of it off,
        // C2_NO_ROUTE=1 the routing to later tiers; the 14-diagonal launch runs with C2_P16_TIER=1 only (as in the library).
        bool score_stage = false, tier40_runs = false;
        std::vector<uint32_t> elist(A.n_tasks ? A.n_tasks : 1), nlist(A.n_tasks ? A.n_tasks : 1), plist(A.n_tasks ? A.n_tasks : 1);
        uint32_t e_count = 0, ne_count = 0, p_count = 0;
        static uint32_t class_counts[8];                                   // [7]: class-0 reads equal to their reference, finished by the partition itself
        for (int k = 0; k < 8; ++k) class_counts[k] = 0;
        g_last_classes = class_counts; g_last_p16_finished = 0;
        if (any_pk && (band_lanes == -87 || band_lanes == -8 || band_lanes == -80) && !(A.all_refs && A.n_refs > 1 && (A.n_refs > 64 || getenv("C2_NO_ALLREFS_PARTITION"))) && !getenv("C2_EMU_NO_SCORE_TIER")) {
            const int sna = (getenv("C2_SCORE_TIER_NA") && atoi(getenv("C2_SCORE_TIER_NA")) == 8) ? 8 : 16;
            const c2_diagx_plan PP = c2_make_diagx_plan(sna, A.max_li, A.max_lj, true, true);
            if (PP.total > sizeof(c2_smem)) return -5;
            score_stage = true;
            const c2_diagx_plan P16 = c2_make_diagx_plan(16, A.max_li, A.max_lj, true, false);
            const bool p16_stage = getenv("C2_P16_TIER") && P16.total <= sizeof(c2_smem);
            const bool tier1_runs = band_lanes == -87, tier2_runs = band_lanes == -87;       // (-8 / -80: the first tier's kernels alone, then the full plane)
            const c2_diagx_plan P6 = c2_make_diagx_plan(6, A.max_li, A.max_lj, true);
            tier40_runs = tier1_runs && !getenv("C2_NO_TIER40") && P6.total <= sizeof(c2_smem);
            const bool route = !getenv("C2_NO_ROUTE");
            // the list a band tier reads is lists[its index - 1]; launch order: 32 | 40 | 62 | 128 (as the host library wires it, c2_api_align.hip)
            const int i40 = 1, i62 = 1 + (tier40_runs ? 1 : 0), i128 = i62 + (tier1_runs ? 1 : 0), ifull = i128 + (tier2_runs ? 1 : 0);
            c2_partition_args PA;
            PA.A = A;
            PA.list[0] = elist.data(); PA.count[0] = &e_count;
            PA.list[1] = plist.data(); PA.count[1] = &p_count;
            PA.list[2] = nlist.data(); PA.count[2] = &ne_count;
            PA.list[3] = lists[i40 - 1]; PA.count[3] = &fb_counts[i40 - 1];
            PA.list[4] = lists[i62 - 1]; PA.count[4] = &fb_counts[i62 - 1];
            PA.list[5] = lists[i128 - 1]; PA.count[5] = &fb_counts[i128 - 1];
            PA.list[6] = lists[ifull - 1]; PA.count[6] = &fb_counts[ifull - 1];      // the list the last launch (the full plane) reads
            PA.direct_full = (route && !getenv("C2_NO_DIRECT_FULL")) ? 1 : 0;
            PA.sort_by_length = getenv("C2_NO_LENGTH_ORDER") ? 0 : 1;
            PA.check_cut = (A.all_refs && A.n_refs > 1) ? 1 : 0; PA.route_cert = getenv("C2_NO_ROUTE_CERT") ? 0 : 1;
            PA.exact_copies = ((((uintptr_t)A.aln_read | (uintptr_t)A.aln_ref) & 3u) == 0 && (A.aln_stride & 3u) == 0) ? 1 : 0;
            PA.class_count = class_counts;
            PA.bandw[0] = p16_stage ? 14 : 0; PA.bandw[1] = 32; PA.bandw[2] = (route && tier40_runs) ? 40 : 0;
            PA.bandw[3] = (route && tier1_runs) ? 62 : 0; PA.bandw[4] = (route && tier2_runs) ? 128 : 0;
            PA.max_mismatch = getenv("C2_SCORE_TIER_MAX_MISMATCH") ? atoi(getenv("C2_SCORE_TIER_MAX_MISMATCH")) : 6;
            PA.probe_max_mismatch = getenv("C2_ROUTE_PROBE_MISMATCH") ? atoi(getenv("C2_ROUTE_PROBE_MISMATCH")) : 4;
            PA.margin = getenv("C2_ROUTE_MARGIN") ? atoi(getenv("C2_ROUTE_MARGIN")) : 3;
            PA.max_shift = (p16_stage || route) ? 64 : 0;
            emu::launch(2, [&] { c2_align_partition_kernel(PA); }, 256);      // (c2_smem: C2_PART_LDS bytes)
            {
                c2_align_args T = A;
                T.task_list = elist.data(); T.task_count = &e_count; T.fb_list = nlist.data(); T.fb_count = &ne_count;
                T.un_list = nullptr; T.un_count = nullptr; T.pair_order = 0;
                work_counter = 0;
                T.plane = nullptr; T.plane_words_per_wg = 0;
                if (getenv("C2_EMU_TRACE")) fprintf(stderr, "launch score-only stage over %u tasks (classes %u %u %u %u %u)\n", e_count, class_counts[0], class_counts[1], class_counts[2], class_counts[3], class_counts[4]);
                if (sna == 16) { if (pk_beta > 0) emu::launch(grid, [&] { c2_align_diags_kernel<16, true>(T); });
                                 else             emu::launch(grid, [&] { c2_align_diags_kernel<16, false>(T); }); }
                else           { if (pk_beta > 0) emu::launch(grid, [&] { c2_align_diags_kernel<8, true>(T); });
                                 else             emu::launch(grid, [&] { c2_align_diags_kernel<8, false>(T); }); }
            }
            if (p16_stage) {
                plane.assign((size_t)grid * P16.n_words * 128u, 0xdeadbeefu);
                c2_align_args T = A;
                T.task_list = plist.data(); T.task_count = &p_count; T.fb_list = nlist.data(); T.fb_count = &ne_count;
                T.un_list = nullptr; T.un_count = nullptr; T.pair_order = 0;
                work_counter = 0;
                T.plane = plane.data(); T.plane_words_per_wg = P16.n_words * 128u;
                const uint32_t before = ne_count;
                if (pk_beta > 0) emu::launch(grid, [&] { c2_align_diagp_kernel<16, true>(T); });
                else             emu::launch(grid, [&] { c2_align_diagp_kernel<16, false>(T); });
                g_last_p16_finished = p_count - (ne_count - before);
                if (getenv("C2_EMU_TRACE")) fprintf(stderr, "launch packed 16 over %u tasks, %u handed on\n", p_count, ne_count - before);
            }
        }
        if (band_lanes == -86 && any_pk) {
            // -86: c2_align_diagp_kernel<6> alone over every task (40 diagonals, three lane groups of 21 lanes), then the full plane
            const c2_diagx_plan PP = c2_make_diagx_plan(6, A.max_li, A.max_lj, true);
            if (PP.total > sizeof(c2_smem)) return -5;
            plane.assign((size_t)grid * PP.n_words * 144u, 0xdeadbeefu);
            c2_align_args T = A;
            chain(T, false, true);
            T.un_list = nullptr; T.un_count = nullptr; T.pair_order = 0;
            T.plane = plane.data(); T.plane_words_per_wg = PP.n_words * 144u;
            if (pk_beta > 0) emu::launch(grid, [&] { c2_align_diagp_kernel<6, true>(T); });
            else             emu::launch(grid, [&] { c2_align_diagp_kernel<6, false>(T); });
            ++tier;
        }
        // -5: five alignments per wavefront (lane groups of 12, lanes 60..63 idle); -75: the chain 5 -> 2 -> 1 -> full plane
        for (int t = 0; t < 2; ++t) {
            const int pna = t == 0 ? 8 : 4, xna = t == 0 ? ((band_lanes == -5 || band_lanes == -75) ? 5 : 4) : 2;
            const bool packed = any_pk && (band_lanes == -87 || band_lanes == -(80 + (t == 0 ? 0 : 4)) || (band_lanes == -8 && t == 0));
            const bool xk = band_lanes == -7 || band_lanes == -75 || band_lanes == -xna || band_lanes == -87 || (packed && (band_lanes == -8 || band_lanes == -84));
            if (!packed && !xk) continue;
            if (packed) {
                const c2_diagx_plan PP = c2_make_diagx_plan(pna, A.max_li, A.max_lj, true);
                if (PP.total > sizeof(c2_smem)) return -5;
                plane.assign((size_t)grid * PP.n_words * 128u, 0xdeadbeefu);
                c2_align_args T = A;
                chain(T, false, true);
                if (t == 0 && score_stage) { T.task_list = nlist.data(); T.task_count = &ne_count; }
                T.plane = plane.data(); T.plane_words_per_wg = PP.n_words * 128u;
                if (getenv("C2_EMU_TRACE")) fprintf(stderr, "launch packed %d tier %d\n", pna, tier);
                if (pk_beta > 0) { if (pna == 8) emu::launch(grid, [&] { c2_align_diagp_kernel<8, true>(T); });
                                   else          emu::launch(grid, [&] { c2_align_diagp_kernel<4, true>(T); }); }
                else             { if (pna == 8) emu::launch(grid, [&] { c2_align_diagp_kernel<8, false>(T); });
                                   else          emu::launch(grid, [&] { c2_align_diagp_kernel<4, false>(T); }); }
            }
            if (xk) {
                const c2_diagx_plan PX = c2_make_diagx_plan(xna, A.max_li, A.max_lj);
                if (PX.total > sizeof(c2_smem)) return -5;
                plane.assign((size_t)grid * PX.n_words * 64u, 0xdeadbeefu);
                c2_align_args T = A;
                chain(T, packed, false);
                T.plane = plane.data(); T.plane_words_per_wg = PX.n_words * 64u;
                if (getenv("C2_EMU_TRACE")) fprintf(stderr, "launch x %d tier %d from_unpaired %d count %u\n", xna, tier, (int)packed, packed ? un_counts[tier] : 0u);
                if (xna == 5) emu::launch(grid, [&] { c2_align_diagx_kernel<5>(T); });
                else if (xna == 4) emu::launch(grid, [&] { c2_align_diagx_kernel<4>(T); });
                else          emu::launch(grid, [&] { c2_align_diagx_kernel<2>(T); });
            }
            ++tier;
            if (t == 0 && score_stage && tier40_runs) {
                // the 40-diagonal tier (round 5): six alignments per wavefront, three lane groups of 21 lanes; no 32-bit twin -- what it cannot pair
                // goes on to the next list
                const c2_diagx_plan PP = c2_make_diagx_plan(6, A.max_li, A.max_lj, true);
                if (PP.total > sizeof(c2_smem)) return -5;
                plane.assign((size_t)grid * PP.n_words * 144u, 0xdeadbeefu);
                c2_align_args T = A;
                chain(T, false, true);
                T.un_list = nullptr; T.un_count = nullptr;
                T.plane = plane.data(); T.plane_words_per_wg = PP.n_words * 144u;
                if (getenv("C2_EMU_TRACE")) fprintf(stderr, "launch packed 6 tier %d\n", tier);
                if (pk_beta > 0) emu::launch(grid, [&] { c2_align_diagp_kernel<6, true>(T); });
                else             emu::launch(grid, [&] { c2_align_diagp_kernel<6, false>(T); });
                ++tier;
            }
        }
        if (band_lanes == -7 || band_lanes == -75 || band_lanes == -1 || band_lanes == -87 || band_lanes == -82) {
            // third band tier (128 diagonals): two alignments per wavefront in int16 halves (c2_align_diagp_kernel<2>, one lane group of
            // 64 lanes) where the references admit it, then the single-alignment kernel -- over everything, or over what could not be paired
            const bool packed = any_pk && (band_lanes == -87 || band_lanes == -82);
            if (packed) {
                const c2_diagx_plan PP = c2_make_diagx_plan(2, A.max_li, A.max_lj, true);
                if (PP.total > sizeof(c2_smem)) return -5;
                plane.assign((size_t)grid * PP.n_words * 128u, 0xdeadbeefu);
                c2_align_args T = A;
                chain(T, false, true);
                T.plane = plane.data(); T.plane_words_per_wg = PP.n_words * 128u;
                if (getenv("C2_EMU_TRACE")) fprintf(stderr, "launch packed 2 tier %d\n", tier);
                if (pk_beta > 0) emu::launch(grid, [&] { c2_align_diagp_kernel<2, true>(T); });
                else             emu::launch(grid, [&] { c2_align_diagp_kernel<2, false>(T); });
            }
            const c2_diag_plan PD = c2_make_diag_plan(A.max_li, A.max_lj);
            if (PD.total > sizeof(c2_smem)) return -5;
            c2_align_args T = A;
            chain(T, packed, false);
            emu::launch(grid, [&] { c2_align_diag_kernel(T); });
            ++tier;
        }
        g_last_unpaired = un_counts[0];
        if (tier == 0) {                                           // (no banded tier ran, e.g. -8 without an admitted reference: everything to the full plane)
            A.task_list = nullptr; A.task_count = nullptr; fb_count = 1;
        } else {
            A.task_list = lists[tier - 1]; A.task_count = &fb_counts[tier - 1];
            fb_count = fb_counts[tier - 1];
        }
        A.fb_list = nullptr; A.fb_count = nullptr;
        work_counter = 0;
        if (n_fallback) *n_fallback = tier ? (int)fb_counts[0] : -1;
    } else if (band) {
        A.band_lanes = band_lanes;
        const c2_lds_plan PB = c2_make_plan(R, A.max_lj, A.max_passes, A.n_codes, band_lanes);
        if (PB.total > sizeof(c2_smem)) return -5;
        switch (R) {
            case 1: emu::launch(grid, [&] { c2_align_classify_kernel<1, 1>(A); }); break;
            case 2: emu::launch(grid, [&] { c2_align_classify_kernel<2, 1>(A); }); break;
            case 3: emu::launch(grid, [&] { c2_align_classify_kernel<3, 1>(A); }); break;
            default: emu::launch(grid, [&] { c2_align_classify_kernel<4, 1>(A); }); break;
        }
        A.task_list = fb_list.data(); A.task_count = &fb_count;
        work_counter = 0;
        if (n_fallback) *n_fallback = (int)fb_count;
    } else if (n_fallback) *n_fallback = -1;
    const c2_lds_plan P = c2_make_plan(R, A.max_lj, A.max_passes, A.n_codes, 0);
    A.band_lanes = 0;
    if (P.total > sizeof(c2_smem) || getenv("C2_EMU_HBM_PLANE")) {
        // the host library's rule: a pointer plane that does not fit LDS lives in per-workgroup HBM scratch
        const c2_lds_plan PH = c2_make_plan(R, A.max_lj, A.max_passes, A.n_codes, 0, true);
        if (PH.total > sizeof(c2_smem)) { fprintf(stderr, "emu: LDS plan %u too large\n", PH.total); return -5; }
        const uint64_t words = (c2_hbm_plane_halfwords(A.max_lj, A.max_passes) + 1) / 2;
        std::vector<uint32_t> scratch((size_t)words * (size_t)grid, 0xdeadbeefu);
        A.plane = scratch.data(); A.plane_words_per_wg = (uint32_t)words;
        if (!(band || diag) || fb_count > 0) {
            switch (R) {
                case 1: emu::launch(grid, [&] { c2_align_classify_kernel<1, 2>(A); }); break;
                case 2: emu::launch(grid, [&] { c2_align_classify_kernel<2, 2>(A); }); break;
                case 3: emu::launch(grid, [&] { c2_align_classify_kernel<3, 2>(A); }); break;
                default: emu::launch(grid, [&] { c2_align_classify_kernel<4, 2>(A); }); break;
            }
        }
        return 0;
    }
    if (!(band || diag) || fb_count > 0) {
        switch (R) {
            case 1: emu::launch(grid, [&] { c2_align_classify_kernel<1, 0>(A); }); break;
            case 2: emu::launch(grid, [&] { c2_align_classify_kernel<2, 0>(A); }); break;
            case 3: emu::launch(grid, [&] { c2_align_classify_kernel<3, 0>(A); }); break;
            default: emu::launch(grid, [&] { c2_align_classify_kernel<4, 0>(A); }); break;
        }
    }
    return 0;
}

int emu_classify_lists(const uint8_t* read_al, const uint8_t* ref_al, int n, const int32_t* include_sorted, int n_include,
                       int legacy, int cap, int32_t* lists, int32_t* list_len, int64_t* counts)
{
    c2_classify_args A;
    A.read_al = read_al; A.ref_al = ref_al; A.include_sorted = include_sorted; A.n = n; A.n_include = n_include;
    A.legacy = legacy; A.cap = cap; A.lists = lists; A.list_len = list_len; A.counts = counts;
    emu::launch(1, [&] { c2_classify_lists_kernel(A); });
    return 0;
}

// batched classifier: pass 0 (lengths), host prefix sum, pass 1 (values) -- the host library's sequence
int emu_classify_lists_batch(uint64_t n, const uint8_t* aln_read, const uint8_t* aln_ref, uint32_t stride, const int32_t* lens,
                             const uint16_t* set_ids, const int32_t* include_sorted, const int64_t* include_off, int legacy,
                             int64_t* index /* n*15+1 */, int32_t* values, int64_t values_cap, int64_t* counts /* n*3 */)
{
    std::vector<int32_t> scratch((size_t)n * stride), llen((size_t)n * C2_LIST_COUNT);
    std::vector<int64_t> loff((size_t)n * C2_LIST_COUNT);
    c2_classify_batch_args A;
    A.aln_read = aln_read; A.aln_ref = aln_ref; A.lens = lens; A.set_ids = set_ids; A.include_sorted = include_sorted;
    A.include_off = include_off; A.n = n; A.stride = stride; A.legacy = legacy; A.pass = 0; A.reserved = 0;
    A.scratch_rp = scratch.data(); A.list_len = llen.data(); A.list_off = loff.data(); A.values = nullptr; A.counts = counts;
    const unsigned grid = (unsigned)((n + 63) / 64);
    emu::launch(grid, [&] { c2_classify_lists_batch_kernel(A); });
    int64_t tot = 0;
    for (size_t k = 0; k < llen.size(); ++k) { loff[k] = tot; index[k] = tot; tot += llen[k]; }
    index[llen.size()] = tot;
    if (tot > values_cap) return -6;
    A.pass = 1; A.values = values;
    emu::launch(grid, [&] { c2_classify_lists_batch_kernel(A); });
    return 0;
}

int emu_consensus_pairs(uint64_t n, const uint8_t* s1, const uint8_t* f1, const uint8_t* s2, const uint8_t* f2, uint32_t stride,
                        const int32_t* n1, const int32_t* n2, const uint8_t* q1, const uint8_t* q2, uint32_t qstride,
                        const int32_t* lq1, const int32_t* lq2, const uint8_t* best1,
                        uint8_t* o_aln, uint8_t* o_ref, uint8_t* o_qual, uint32_t ostride, int32_t* o_info)
{
    c2_consensus_args A;
    A.s1 = s1; A.f1 = f1; A.s2 = s2; A.f2 = f2; A.q1 = q1; A.q2 = q2; A.n1 = n1; A.n2 = n2; A.lq1 = lq1; A.lq2 = lq2; A.best1 = best1;
    A.n = n; A.stride = stride; A.qstride = qstride; A.ostride = ostride; A.reserved = 0;
    A.o_aln = o_aln; A.o_ref = o_ref; A.o_qual = o_qual; A.o_info = o_info;
    emu::launch((unsigned)((n + 63) / 64), [&] { c2_consensus_pairs_kernel(A); });
    return 0;
}

int emu_count_vectors(uint64_t n_tasks, const uint8_t* aln_read, const uint8_t* aln_ref, uint32_t aln_stride,
                      const c2_aln_record* records, const uint32_t* weights, const uint16_t* min_matches, int max_t,
                      int n_refs, const int32_t* lens, const int32_t* const* include_idx, const int32_t* n_include,
                      int flags, int hl, long long* counts, unsigned grid, const char* const* seqs)
{
    std::vector<c2_dev_ref> refs(n_refs);
    std::vector<std::vector<uint16_t>> incp(n_refs);
    int lmax = 1;
    for (int r = 0; r < n_refs; ++r) {
        c2_build_inc_prefix(include_idx[r], n_include[r], lens[r], incp[r]);
        refs[r].seq = (const uint8_t*)seqs[r]; refs[r].gap_incentive = nullptr; refs[r].inc_prefix = incp[r].data(); refs[r].diag_rows = nullptr; refs[r].len = lens[r]; refs[r].gap_incentive_max = 0; refs[r].gap_incentive_last_pos = 0; refs[r].max_char = 0;
        lmax = std::max(lmax, lens[r]);
    }
    unsigned long long wc = 0;
    c2_count_args A;
    A.aln_read = aln_read; A.aln_ref = aln_ref; A.records = records; A.weights = weights; A.min_matches = min_matches;
    A.refs = refs.data(); A.counts = counts; A.work_counter = &wc; A.n_tasks = n_tasks; A.aln_stride = aln_stride;
    A.n_refs = n_refs; A.lmax = lmax; A.hl = hl; A.max_t = max_t; A.flags = flags;
    A.hints = nullptr; A.order = nullptr; A.block_scratch = nullptr; A.block_ints = 0;
    A.rest_list = nullptr; A.rest_count = nullptr; A.n_tasks_dev = nullptr; A.ref_ends = nullptr; A.hint_gx = 0;
    const uint32_t* hints_multi = (g_next_count_hints && n_refs > 1 && !(flags & C2_CNT_FLAG_ALL_REFS_LAYOUT)) ? g_next_count_hints : nullptr;
    std::vector<uint32_t> rest(n_tasks ? n_tasks : 1);
    uint32_t n_rest = 0;
    bool rest_order = false;
    if (g_next_count_hints && n_refs == 1) {                         // (c2_count_vectors_hinted_device: the hinted tasks first, by their own kernel; what it leaves as a list)
        A.hints = g_next_count_hints;
        if (!getenv("C2_NO_COUNT_REST_LIST")) { A.rest_list = rest.data(); A.rest_count = &n_rest; }
        emu::launch(grid ? grid : 2, [&] { c2_count_hinted_kernel(A); }, 256);
        if (A.rest_list) { rest_order = true; A.n_tasks_dev = &n_rest; A.hints = nullptr; }
    }
    g_next_count_hints = nullptr;
    // tasks grouped by reference, as the host library does for more than one reference
    std::vector<uint32_t> hist(n_refs + 1, 0), order(n_tasks ? n_tasks : 1);
    A.order = nullptr;
    if (n_refs > 1) {
        if (!getenv("C2_NO_REF_LDS_GROUPING")) {                     // (the library's choice up to C2_REF_LDS_MAX references)
            const unsigned gc = (unsigned)((n_tasks + C2_REF_CHUNK - 1) / C2_REF_CHUNK);
            emu::launch(gc, [&] { c2_ref_histogram_lds_kernel(records, n_tasks, hist.data(), n_refs); }, 256);
            emu::launch(1, [&] { c2_ref_scan_kernel(hist.data(), n_refs); });
            emu::launch(gc, [&] { c2_ref_scatter_lds_kernel(records, n_tasks, hist.data(), order.data(), n_refs); }, 256);
        } else {
        emu::launch((unsigned)((n_tasks + 255) / 256), [&] { c2_ref_histogram_kernel(records, n_tasks, hist.data()); }, 256);
        emu::launch(1, [&] { c2_ref_scan_kernel(hist.data(), n_refs); });
        emu::launch((unsigned)((n_tasks + 255) / 256), [&] { c2_ref_scatter_kernel(records, n_tasks, hist.data(), order.data()); }, 256);
        }
        A.order = order.data();
        if (hints_multi) {                                           // (c2_count_vectors_hinted_device with several references: per reference over the grouped order)
            A.hints = hints_multi; A.ref_ends = hist.data(); A.hint_gx = 2;
            emu::launch(2u * (unsigned)n_refs, [&] { c2_count_hinted_kernel(A); }, 256);
        }
    }
    if (rest_order) A.order = rest.data();
    A.block_scratch = nullptr; A.block_ints = 0;
    if (getenv("C2_EMU_COUNT_HBM")) {
        // the variant whose accumulator blocks live in "HBM" (one per workgroup), as the library takes for amplicons beyond the LDS block
        const size_t per_ref = (size_t)C2_CNT_VECTORS * (lmax + 1) + C2_CNT_SCALARS + (size_t)C2_CNT_HISTS * hl;
        A.block_ints = (per_ref + 63) / 64 * 64;
        std::vector<int32_t> blocks((size_t)(grid ? grid : 2) * A.block_ints, 0x5a5a5a5a);       // (garbage: the kernel must zero its block)
        A.block_scratch = blocks.data();
        emu::launch(grid ? grid : 2, [&] { c2_count_vectors_hbm_kernel(A); }, 64 * C2_CNT_WAVES);
        return 0;
    }
    emu::launch(grid ? grid : 2, [&] { c2_count_vectors_kernel(A); }, 64 * C2_CNT_WAVES);
    return 0;
}

// c2_strand_plan_device on the emulator: same arguments, minus the context and the stream
int emu_strand_plan(uint64_t n_reads, const uint8_t* reads, const uint64_t* offsets, int32_t max_read_len, int32_t n_refs, int32_t max_seeds,
                    const int32_t* n_seeds, const uint8_t* seed_blob, int32_t blob_bytes, const int32_t* seed_off, const int32_t* seed_len,
                    int32_t seed_min, uint8_t* plan)
{
    (void)blob_bytes;
    c2_strand_args A;
    A.reads = reads; A.offsets = offsets; A.n_reads = n_reads; A.seed_blob = seed_blob; A.seed_off = seed_off; A.seed_len = seed_len;
    A.n_seeds = n_seeds; A.n_refs = n_refs; A.max_seeds = max_seeds; A.seed_min = seed_min; A.max_read_len = max_read_len; A.plan = plan;
    // the seed table in LDS, as c2_strand_plan_device decides it (C2_EMU_STRAND_BYTEWISE: the byte-by-byte path instead)
    const size_t tbl = (size_t)n_refs * 2 * (size_t)max_seeds;
    bool table = tbl > 0 && !getenv("C2_EMU_STRAND_BYTEWISE");
    for (size_t q = 0; q < tbl && table; ++q) if ((uint32_t)seed_len[q] > C2_SEED_SLOT) table = false;
    size_t lds = (size_t)4 * c2_strand_row_bytes(max_read_len);
    if (table && lds + tbl * C2_SEED_SLOT > 65536) table = false;
    if (table) lds += tbl * C2_SEED_SLOT;
    A.seed_table = table ? 1 : 0; A.reserved = 0;
    if (lds > sizeof(c2_smem)) return -5;
    emu::launch(3, [&] { c2_strand_plan_kernel(A); }, 256);
    return 0;
}

// c2_fq_*_device on the emulator: same arguments, minus the context and the stream; `grid` of the grid-stride kernels is small
int emu_fq_count(const uint8_t* text, uint64_t lo, uint64_t hi, uint32_t* tile_newlines, uint32_t* tile_empty, uint32_t* flags)
{
    if (hi <= lo) return 0;
    c2_fq_frame_args A{};
    A.text = text; A.lo = lo; A.hi = hi; A.tile_newlines = tile_newlines; A.tile_empty = tile_empty; A.flags = flags;
    emu::launch((unsigned)((hi - lo + C2_FQ_TILE - 1) / C2_FQ_TILE), [&] { c2_fq_count_kernel(A); }, 256);
    return 0;
}
int emu_fq_lines(const uint8_t* text, uint64_t lo, uint64_t hi, const uint64_t* tile_base, uint64_t* seq_start, uint64_t* seq_end, uint64_t cap)
{
    if (hi <= lo) return 0;
    c2_fq_frame_args A{};
    A.text = text; A.lo = lo; A.hi = hi; A.tile_base = tile_base; A.seq_start = seq_start; A.seq_end = seq_end; A.n_records_cap = cap;
    emu::launch((unsigned)((hi - lo + C2_FQ_TILE - 1) / C2_FQ_TILE), [&] { c2_fq_lines_kernel(A); }, 256);
    return 0;
}
int emu_fq_lines4(const uint8_t* text, uint64_t lo, uint64_t hi, const uint64_t* tile_base, uint64_t* seq_start, uint64_t* seq_end, uint64_t* qual_start,
                  uint64_t* qual_end, uint64_t cap)
{
    if (hi <= lo) return 0;
    c2_fq_frame_args A{};
    A.text = text; A.lo = lo; A.hi = hi; A.tile_base = tile_base; A.seq_start = seq_start; A.seq_end = seq_end; A.n_records_cap = cap;
    A.qual_start = qual_start; A.qual_end = qual_end;
    emu::launch((unsigned)((hi - lo + C2_FQ_TILE - 1) / C2_FQ_TILE), [&] { c2_fq_lines_kernel(A); }, 256);
    return 0;
}
// c2_fq_pair_lengths_device / c2_fq_pair_write_device
int emu_fq_pair_lengths(const uint8_t* text1, const uint8_t* text2, const uint64_t* const* lines1, const uint64_t* const* lines2, uint64_t n, uint64_t* s1,
                        uint64_t* q1, uint64_t* s2, uint64_t* q2, int64_t* key_len, int64_t* qual_len, uint32_t* flags)
{
    if (!n) return 0;
    c2_fq_pair_args A{};
    A.text1 = text1; A.text2 = text2;
    A.seq_start1 = lines1[0]; A.seq_end1 = lines1[1]; A.qual_start1 = lines1[2]; A.qual_end1 = lines1[3];
    A.seq_start2 = lines2[0]; A.seq_end2 = lines2[1]; A.qual_start2 = lines2[2]; A.qual_end2 = lines2[3];
    A.n = n; A.s1 = (unsigned long long*)s1; A.q1 = (unsigned long long*)q1; A.s2 = (unsigned long long*)s2; A.q2 = (unsigned long long*)q2;
    A.key_len = key_len; A.qual_len = qual_len; A.flags = flags;
    emu::launch((unsigned)((n + 255) / 256), [&] { c2_fq_pair_lengths_kernel(A); }, 256);
    return 0;
}
int emu_fq_pair_write(const uint8_t* text1, const uint8_t* text2, uint64_t n, const uint64_t* s1, const uint64_t* q1, const uint64_t* s2, const uint64_t* q2,
                      const int64_t* key_off, const int64_t* qual_off, uint8_t* key_out, uint8_t* qual_out, uint32_t* flags)
{
    if (!n) return 0;
    c2_fq_pair_args A{};
    A.text1 = text1; A.text2 = text2; A.n = n;
    A.s1 = (unsigned long long*)s1; A.q1 = (unsigned long long*)q1; A.s2 = (unsigned long long*)s2; A.q2 = (unsigned long long*)q2;
    A.key_off = key_off; A.qual_off = qual_off; A.key_out = key_out; A.qual_out = qual_out; A.flags = flags;
    emu::launch(3, [&] { c2_fq_pair_write_kernel(A); }, 256);
    return 0;
}
int emu_fq_dedup(const uint8_t* text, const uint64_t* seq_start, const uint64_t* seq_end, const uint64_t* range, uint64_t cap, uint64_t* slots,
                 uint64_t n_slots, uint32_t* count, uint32_t* first, uint32_t* slot_of, uint64_t* rinfo, uint32_t* flags, uint32_t* stats)
{
    c2_fq_dedup_args A{};
    A.text = text; A.seq_start = seq_start; A.seq_end = seq_end; A.range = range; A.n_records_cap = cap; A.slots = (unsigned long long*)slots;
    A.mask = n_slots - 1; A.count = count; A.first = first; A.slot_of = slot_of; A.rinfo = (unsigned long long*)rinfo; A.flags = flags; A.stats = stats;
    emu::launch(3, [&] { c2_fq_dedup_kernel(A); }, 256);
    return 0;
}
int emu_fq_gather(const uint8_t* text, const uint64_t* info, const int64_t* records, const int64_t* out_offsets, uint8_t* out, uint64_t n)
{
    if (!n) return 0;
    c2_fq_gather_args A{};
    A.text = text; A.info = (const unsigned long long*)info; A.records = records; A.out_offsets = out_offsets; A.out = out; A.n = n;
    emu::launch(3, [&] { c2_fq_gather_kernel(A); }, 256);
    return 0;
}

int emu_fq_rc_partner(const uint8_t* text, const uint64_t* info, const int64_t* records, uint64_t n, const uint64_t* slots, uint64_t n_slots, int32_t* partner_slot)
{
    if (!n) return 0;
    c2_fq_rc_args A{};
    A.text = text; A.info = (const unsigned long long*)info; A.records = records; A.n = n; A.slots = (const unsigned long long*)slots; A.mask = n_slots - 1;
    A.partner_slot = partner_slot;
    emu::launch(3, [&] { c2_fq_rc_partner_kernel(A); }, 256);
    return 0;
}

// The per-call C ABI on the emulator, argument for argument (the context handle is ignored): what
// crispresso2_amd.CRISPResso2Align.global_align / CRISPRessoCOREResources.find_indels_substitutions[_legacy] call.
// Host-side marshalling follows c2_api_align.hip / c2_api_classify.hip (c2_global_align, c2_find_indels_substitutions).
int emu_c2_global_align(void*, const char* read, int32_t Lj, const char* ref, int32_t Li, const int64_t* matrix, int32_t mat_dim,
                        const int64_t* gap_incentive, int32_t n_gap_incentive, int32_t gap_open, int32_t gap_extend,
                        char* out_read_aln, char* out_ref_aln, int32_t* out_len, int32_t* out_matches, int32_t* out_status)
{
    *out_len = 0; *out_matches = 0; *out_status = 0;
    if (n_gap_incentive != Li + 1) { *out_status = -1; return 0; }
    if (Li <= 0 || Lj <= 0) { *out_status = C2_STATUS_EMPTY; return 0; }
    const uint64_t offs[2] = {0, (uint64_t)Lj};
    const char* seqs[1] = {ref};
    const int32_t lens[1] = {Li};
    const int64_t* gis[1] = {gap_incentive};
    const int32_t* incs[1] = {nullptr};
    const int32_t ninc[1] = {0};
    const uint32_t stride = (uint32_t)((Li + Lj + 15) / 16 * 16);
    std::vector<uint8_t> o1(stride), o2(stride);
    c2_aln_record rec;
    memset(&rec, 0, sizeof rec);
    int nfb = 0;
    const int rc = emu_align_batch(1, (const uint8_t*)read, offs, nullptr, nullptr, 0, 1, seqs, lens, gis, incs, ninc, matrix, mat_dim,
                                   gap_open, gap_extend, o1.data(), o2.data(), stride, &rec, 0, 1, 0, -7, &nfb);
    if (rc == -5) return C2_E_TOO_LARGE;
    if (rc) return C2_E_INVALID;
    *out_status = rec.status;
    if (rec.status == 0) {
        memcpy(out_read_aln, o1.data(), rec.aln_len);
        memcpy(out_ref_aln, o2.data(), rec.aln_len);
        *out_len = rec.aln_len; *out_matches = rec.matches;
    }
    return 0;
}

int emu_c2_find_indels_substitutions(void*, const char* read_aln, const char* ref_aln, int32_t n, const int32_t* include_idx,
                                     int32_t n_include, int32_t legacy, int32_t* out, int32_t out_cap, int32_t* out_index,
                                     int64_t* out_counts, int32_t* out_needed)
{
    std::vector<int32_t> inc(include_idx, include_idx + (n_include > 0 ? n_include : 0));
    std::sort(inc.begin(), inc.end());
    inc.erase(std::unique(inc.begin(), inc.end()), inc.end());
    int cap = std::max(2 * n + 8, 64);
    std::vector<int32_t> lens(C2_LIST_COUNT), lists;
    for (int attempt = 0; attempt < 3; ++attempt) {
        lists.assign((size_t)C2_LIST_COUNT * cap, 0);
        emu_classify_lists((const uint8_t*)read_aln, (const uint8_t*)ref_aln, n, inc.data(), (int)inc.size(), legacy, cap, lists.data(), lens.data(), out_counts);
        const int need = *std::max_element(lens.begin(), lens.end());
        if (need <= cap) break;
        cap = need + 8;
        if (attempt == 2) return C2_E_DEVICE;
    }
    int64_t total = 0;
    for (int k = 0; k < C2_LIST_COUNT; ++k) total += lens[k];
    if (out_needed) *out_needed = (int32_t)total;
    if (total > out_cap) return C2_E_OVERFLOW;
    int32_t pos = 0;
    for (int k = 0; k < C2_LIST_COUNT; ++k) {
        out_index[2 * k] = pos; out_index[2 * k + 1] = lens[k];
        if (lens[k]) memcpy(out + pos, lists.data() + (size_t)k * cap, (size_t)lens[k] * 4);
        pos += lens[k];
    }
    return 0;
}

int emu_c2_calculate_homology(void*, const char* a, const char* b, int32_t n, double* out)
{
    float f = 0;
    emu::launch(1, [&] { c2_homology_kernel((const uint8_t*)a, (const uint8_t*)b, n, &f); });
    *out = (double)f;
    return 0;
}

int emu_select_best(uint64_t n_reads, int n_refs, const c2_aln_record* records, const c2_aln_record* records2, const int32_t* slot2,
                    const uint32_t* min_mscore, const uint32_t* raw_counts, const uint32_t* counts, int mode,
                    unsigned long long* member, unsigned long long* use2, uint8_t* flags, uint32_t* weights, uint32_t* weights2,
                    unsigned long long* stats)
{
    c2_select_args A;
    A.records = records; A.records2 = records2; A.slot2 = slot2; A.min_mscore = min_mscore; A.raw_counts = raw_counts; A.counts = counts;
    A.member = member; A.use2 = use2; A.flags = flags; A.weights = weights; A.weights2 = weights2; A.stats = stats;
    A.n_reads = n_reads; A.n_refs = n_refs; A.mode = mode;
    emu::launch((unsigned)((n_reads + 255) / 256), [&] { c2_select_best_kernel(A); }, 256);
    return 0;
}

int emu_classify_records(uint64_t n, const uint8_t* aln_read, const uint8_t* aln_ref, uint32_t stride, const int32_t* info, const uint16_t* ref_ids,
                         const uint8_t* strands, int legacy, c2_aln_record* records, int n_refs, const int32_t* lens,
                         const int32_t* const* include_idx, const int32_t* n_include)
{
    std::vector<c2_dev_ref> refs(n_refs);
    std::vector<std::vector<uint16_t>> incp(n_refs);
    for (int r = 0; r < n_refs; ++r) {
        c2_build_inc_prefix(include_idx[r], n_include[r], lens[r], incp[r]);
        memset(&refs[r], 0, sizeof refs[r]);
        refs[r].inc_prefix = incp[r].data(); refs[r].len = lens[r];
    }
    c2_records_args A;
    A.aln_read = aln_read; A.aln_ref = aln_ref; A.info = info; A.ref_ids = ref_ids; A.strands = strands; A.refs = refs.data(); A.records = records;
    A.n = n; A.stride = stride; A.n_refs = n_refs; A.legacy = legacy; A.reserved = 0;
    emu::launch((unsigned)((n + 3) / 4), [&] { c2_classify_records_kernel(A); }, 256);
    return 0;
}

uint32_t emu_mscore(uint32_t matches, uint32_t T) { return c2_mscore(matches, T); }

int emu_selftest(int* out) { emu::launch(1, [&] { c2_selftest_kernel(out); }); return 0; }

}  // extern "C"

// ---- the allele table: c2_alleles_host.h (the product's own orchestration) over an emulator backend -- host memory, the fiber launcher,
// std::stable_sort with the kernels' comparators.  Same entry points as the C ABI's c2_allele_table_*, without the context.
#include "../../crispresso2_amd/csrc/c2_alleles_host.h"
namespace {
struct EmuBackend {
    std::string error;
    void* dalloc(size_t n) { return malloc(n ? n : 1); }
    void dfree(void* p) { free(p); }
    void* halloc(size_t n) { return malloc(n ? n : 1); }
    void hfree(void* p) { free(p); }
    bool h2d(void* d, const void* h, size_t n) { memcpy(d, h, n); return true; }
    bool d2h(void* h, const void* d, size_t n) { memcpy(h, d, n); return true; }
    bool zero(void* d, size_t n) { memset(d, 0, n); return true; }
    bool sync() { return true; }
    static unsigned blocks(uint64_t n, unsigned per) { return (unsigned)std::max<uint64_t>(1, (n + per - 1) / per); }
    bool jobs(const c2_allele_jobs_args& A) { emu::launch(blocks(A.S.n_reads, 256), [&] { c2_allele_jobs_kernel(A); }, 256); return true; }
    bool iota(uint32_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = (uint32_t)i; return true; }
    bool reads(const c2_allele_row* rows, const uint32_t* order, uint64_t m, uint32_t* out) { emu::launch(blocks(m, 256), [&] { c2_allele_reads_kernel(rows, order, m, out); }, 256); return true; }
    bool probe(const c2_allele_probe_args& A) { emu::launch(blocks(A.m, 4), [&] { c2_allele_probe_kernel(A); }, 256); return true; }
    bool lengths(const c2_allele_text_args& A) { emu::launch(blocks(A.m, 256), [&] { c2_allele_lengths_kernel(A); }, 256); return true; }
    bool emit(const c2_allele_text_args& A) { emu::launch(blocks(A.q1 - A.q0, 4), [&] { c2_allele_emit_kernel(A); }, 256); return true; }
    bool fetch(const c2_allele_fetch_args& A) { emu::launch(blocks(A.m, 4), [&] { c2_allele_fetch_kernel(A); }, 256); return true; }
    bool window(const c2_allele_window_args& A) { emu::launch(blocks(A.m, A.sub_index ? 4 : 256), [&] { c2_allele_window_kernel(A); }, 256); return true; }
    bool group(const c2_allele_group_args& A) { emu::launch(blocks(A.ms, 256), [&] { c2_allele_group_kernel(A); }, 256); return true; }
    bool scan(const uint32_t* in, uint64_t* out, uint64_t n) { uint64_t acc = 0; for (uint64_t i = 0; i < n; ++i) { const uint64_t v = in[i]; out[i] = acc; acc += v; } return true; }
    template <class Less> bool sort_with(Less less, uint32_t* in, uint32_t* out, uint64_t n) { std::copy(in, in + n, out); std::stable_sort(out, out + n, less); return true; }
    bool sort_rows(c2_allele_row_less less, uint32_t* in, uint32_t* out, uint64_t n) { return sort_with(less, in, out, n); }
    bool sort_keys(c2_allele_key_less less, uint32_t* in, uint32_t* out, uint64_t n) { return sort_with(less, in, out, n); }
};
std::string g_allele_err;
}  // namespace

extern "C" {
const char* emu_allele_last_error() { return g_allele_err.c_str(); }
int emu_allele_table_build(const c2_allele_src* src, void** out) {
    c2a_table<EmuBackend>* t = nullptr;
    const int rc = c2a_build(EmuBackend(), *src, &t, g_allele_err);
    if (rc == 0) *out = t;
    return rc;
}
uint64_t emu_allele_table_rows(void* t) { return t ? ((c2a_table<EmuBackend>*)t)->m : 0; }
int emu_allele_table_write(void* t, const char* path, const char* const* labels, int64_t n_total, const char* const* probes, int32_t threads, uint64_t* bytes_written) {
    auto* T = (c2a_table<EmuBackend>*)t;
    const int rc = c2a_write(T, path, labels, n_total, probes, threads, bytes_written);
    if (rc) g_allele_err = T->err;
    return rc;
}
int emu_allele_table_write_zip(void* t, const char* zip_path, const char* member, const char* const* labels, int64_t n_total, const char* const* probes,
                               int32_t threads, int32_t level, uint64_t* text_bytes, uint64_t* zip_bytes) {
    auto* T = (c2a_table<EmuBackend>*)t;
    const int rc = c2a_write(T, zip_path, labels, n_total, probes, threads, text_bytes, member, level, zip_bytes);
    if (rc) g_allele_err = T->err;
    return rc;
}
int emu_allele_table_fetch(void* t, c2_allele_row* rows, uint8_t* aligned, uint8_t* reference, uint32_t stride) {
    auto* T = (c2a_table<EmuBackend>*)t;
    const int rc = c2a_fetch(T, rows, aligned, reference, stride);
    if (rc) g_allele_err = T->err;
    return rc;
}
int emu_allele_table_around_cut_write(void* t, int32_t label, int32_t cut_point, int32_t ref_len, int32_t plot_window_size, int64_t n_total, const char* path,
                                      int32_t threads, uint64_t* n_groups) {
    auto* T = (c2a_table<EmuBackend>*)t;
    const int rc = c2a_around_cut_write(T, label, cut_point, ref_len, plot_window_size, n_total, path, threads, n_groups);
    if (rc) g_allele_err = T->err;
    return rc;
}
void emu_allele_table_free(void* t) { delete (c2a_table<EmuBackend>*)t; }
int emu_format_float_repr(double v, char* out) { return c2_py_float_repr(v, out); }
}
