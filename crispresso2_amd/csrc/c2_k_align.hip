// c2_k_align.hip -- CDNA4 (gfx950) kernels for CRISPResso2's align + classify hot path.
//
// Replaces, bit-identically on the reference's defined domain:
//   CRISPResso2Align.global_align               (reference CRISPResso2/CRISPResso2Align.pyx:101-434)
//   CRISPRessoCOREResources.find_indels_substitutions (CRISPResso2/CRISPRessoCOREResources.pyx:68-187)
//
// Design (DESIGN.md has the long form):
//   * every batch runs through a chain of launches; each kernel proves its own results and hands the tasks it cannot prove
//     to the next one (device-side task lists):
//       c2_align_diagx_kernel<4>, <2>  lanes own DIAGONALS, the sweep runs over anti-diagonals; 4 or 2 alignments share a
//                                      wavefront in lane groups isolated by an EXEC-disabled lane; optimality certificate
//       c2_align_diag_kernel           the same fill with 128 diagonals for one alignment
//       c2_align_classify_kernel<R,B>  lanes own R reference ROWS (systolic sweep over the whole matrix): any path
//   * cross-lane traffic is DPP only (wave_shr:1 / wave_shl:1); scores are int32 exactly as the reference's C ints; the
//     three tie rules are kept as compare results, 4 pointer bits per cell -- the 6 x (Li+1) x (Lj+1) int32 matrices of
//     the reference (1.5 MB per 250x250 alignment) never exist.
//   * traceback is wave-parallel: the 64 lanes probe 64 consecutive cells along the current direction (diagonal / row /
//     column), a ballot finds the length of the run, and the run's columns are emitted by the lanes in one shot.
//   * the indel / substitution / quantification-window classification is computed from the aligned strings while they are
//     still in LDS (ballot + popcount prefix scans); c2_count_vectors_kernel turns strings + records into the per-amplicon
//     count tensor; c2_classify_lists[_batch]_kernel produce the reference's full position lists.
//   * HBM traffic: the read comes in once (coalesced byte loads), the two aligned strings and one 32-byte record go out
//     once; the multi-alignment kernels additionally park their pointer words in a scratch plane (written once, read once).
#pragma once
#include "c2_k_common.h"

// h-state (argmax with the reference's tie rule, pyx:216-228) of a cell on row 0 or column 0,
// from the closed-form boundary values (pyx:153-176).
__device__ __forceinline__ int c2_boundary_hstate(int i, int j, int min_score, int ge, int g0) {
    if (i == 0 && j == 0) return C2_ST_M;                 // M[0,0]=0 beats I=J=min_score
    if (i == 0) return (ge * j + g0 >= min_score) ? C2_ST_I : C2_ST_J;   // M=J=min_score, I=ge*j+g0
    return (ge * i + g0 <= min_score) ? C2_ST_I : C2_ST_J;              // M=I=min_score, J=ge*i+g0
}

// Append the four pointer bits of one cell to `bits` (newest in the low bits):
//   bit3 = a0 > b0 (I opened), bit2 = a1 > b1 (J opened), bit1 = a2 == b2 (H is I), bit0 = a3 >= b3 (J beats M).
// On the device: four v_cmp into SGPR pairs, then four v_addc_co_u32 (bits = 2*bits + carry) -- 8 VALU issues, and each
// compare result is read 3+ issues after it was written (gfx950 needs 2 wait states there, which hipcc cannot see in asm).
__device__ __forceinline__ void c2_push4(unsigned& bits, int a0, int b0, int a1, int b1, int a2, int b2, int a3, int b3) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long m1, m2, m3;
    asm("v_cmp_gt_i32 %1, %4, %5\n\t"
        "v_cmp_gt_i32 %2, %6, %7\n\t"
        "v_cmp_eq_u32 %3, %8, %9\n\t"
        "v_cmp_ge_i32 vcc, %10, %11\n\t"
        "v_addc_co_u32 %0, %1, %0, %0, %1\n\t"
        "v_addc_co_u32 %0, %2, %0, %0, %2\n\t"
        "v_addc_co_u32 %0, %3, %0, %0, %3\n\t"
        "v_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(bits), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
        : "vcc");
#else
    bits = (bits << 4) | ((unsigned)(a0 > b0) << 3) | ((unsigned)(a1 > b1) << 2) | ((unsigned)(a2 == b2) << 1) | (unsigned)(a3 >= b3);
#endif
}

// Optional per-phase cycle accounting (c2_align_args.phase_cycles != NULL): every workgroup sums the s_memtime delta of
// each phase of each of its tasks in registers and adds the four sums to the global counters once, when it runs out of
// work.  Used by tools/phase_profile.py; costs one uniform branch per phase when disabled.
struct c2_phase_acc { unsigned long long t_last, sum[4]; };
__device__ __forceinline__ void c2_phase_begin(const unsigned long long* acc, c2_phase_acc& P) { if (acc) P.t_last = (unsigned long long)clock64(); }
template <int PHASE>
__device__ __forceinline__ void c2_phase_mark(const unsigned long long* acc, c2_phase_acc& P) {
    if (acc) {
        const unsigned long long now = (unsigned long long)clock64();
        P.sum[PHASE] += now - P.t_last;
        P.t_last = now;
    }
}
__device__ __forceinline__ void c2_phase_flush(unsigned long long* acc, const c2_phase_acc& P, const int lane) {
#ifndef C2_PART_PHASES                              // (A/B build: the four counters belong to c2_align_partition_kernel's phases alone)
    if (acc && lane == 0) { for (int k = 0; k < 4; ++k) atomicAdd(acc + k, P.sum[k]); }
#endif
}
#ifdef C2_PART_PHASES
#define C2_PART_MARK(k) do { if (A.phase_cycles && tid == 0) { const unsigned long long now_ = (unsigned long long)clock64(); atomicAdd(A.phase_cycles + (k), now_ - pt_last); pt_last = now_; } } while (0)
#else
#define C2_PART_MARK(k) do { } while (0)
#endif

// floor(x / R) for the rows-per-lane values in use (x < 32768)
template <int R>
__device__ __forceinline__ int c2_div_rows(int x) {
    if (R == 1) return x;
    if (R == 2) return x >> 1;
    if (R == 4) return x >> 2;
    return (x * 21846) >> 16;                           // R == 3
}

// Halfword index of the pointer word of (row-lane `rl`, column j) inside one pass's pointer plane.
// Full plane: row j-1, slot rl.  BAND: row t-1 (t = j + rl is the step at which that lane computed the column), slot
// rl - lo(t); *inband tells whether the word was stored.
// MODE 0: full plane in LDS, 1: banded plane in LDS (BAND), 2: full plane in HBM scratch, row t-1, slot rl.
template <int R, int MODE>
__device__ __forceinline__ int c2_ptr_index(const int rl, const int j, const int colStride, const int band_lanes, bool& inband) {
    if (MODE == 2) { inband = true; return (j + rl - 1) * 64 + rl; }
    if (MODE == 0) { inband = true; return (j - 1) * colStride + rl; }
    const int t = j + rl;
    const int slot = rl - c2_band_lo(R, band_lanes, t);
    inband = (unsigned)slot < (unsigned)c2_band_slots(R, band_lanes);
    return (t - 1) * colStride + slot;
}

// Per-lane DP state of one systolic pass: R consecutive reference rows.
template <int R>
struct c2_strip {
    int a[R], b[R], c[R], delta[R];   // gap constants of the rows (see c2_dp_pass)
    int sel[R];                       // PACKED: 8 signed score nibbles of the row's base; else: LDS row offset into the score table
    int Ml[R], Il[R], Hl[R];          // M, I, H=max(M,I,J) of the rows at the column computed last
    int Mb, Jb, Hb;                   // bottom row at that column: what the lane below receives
    int upM, upJ, upH;                // hand-off registers: lanes 1..63 receive the lane above (DPP), lane 0 keeps the boundary row
    int dgsave;                       // H(row above the strip, previous column)
    int cj;                           // read symbol of the current column (PACKED: 4*code, else code)
    unsigned bits;                    // pointer nibbles, newest in the low bits
};

// One column of the strip (the lane is active: 1 <= j <= Lj).  TAIL: the lane may be on the last column, where
// gap_open is replaced by gap_extend (pyx:234-273) -- delta[] carries ge-go for every row but the last one.
template <int R, bool PACKED, bool TAIL>
__device__ __forceinline__ void c2_dp_column(c2_strip<R>& S, const int upM0, const int upJ0, const int ge,
                                             const bool lastcol, const int16_t* sTbl)
{
    int upM = upM0, upJ = upJ0, dg = S.dgsave;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int s;
        if (PACKED) s = c2_sbfe4(S.sel[r], S.cj);                             // signed nibble number cj/4: one v_bfe_i32
        else        s = (int)sTbl[S.sel[r] + S.cj];
        int iFromM = S.Ml[r] + S.a[r];
        int jFromM = upM + S.c[r];
        if (TAIL) { const int corr = lastcol ? S.delta[r] : 0; iFromM += corr; jFromM += corr; }
        const int iExt = S.Il[r] + S.b[r];
        const int In = c2_imax(iFromM, iExt);             // pyx:191-196: tie -> extend (pointer bit: iFromM > iExt)
        const int jExt = upJ + ge;
        const int Jn = c2_imax(jFromM, jExt);             // pyx:199-211: tie -> extend (pointer bit: jFromM > jExt)
        const int Mn = dg + s;                            // H(i-1,j-1) + matrix[ci,cj], pyx:213-228
        const int Hn = c2_imax(c2_imax(Mn, Jn), In);      // v_max3_i32
        // H is I iff In >= max(Mn, Jn) iff In == Hn (I wins all ties); else J iff Jn >= Mn (J beats M on a tie)
        c2_push4(S.bits, iFromM, iExt, jFromM, jExt, In, Hn, Jn, Mn);
        dg = S.Hl[r];
        S.Ml[r] = Mn; S.Il[r] = In; S.Hl[r] = Hn;
        upM = Mn; upJ = Jn;
    }
    S.Mb = upM; S.Jb = upJ; S.Hb = S.Hl[R - 1];
}

// Symbols of read positions base+l, base+l+64, base+l+128, base+l+192 packed into one register per lane l
// (PACKED: 4*code, the bit offset of the score nibble).
template <bool PACKED>
__device__ __forceinline__ int c2_load_rsym(const unsigned char* sCode, const int base, const int Lj, const int lane) {
    int w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int q = base + 64 * b + lane;
        const int c = q < Lj ? (int)sCode[q] : 0;
        w |= (PACKED ? (c << 2) : c) << (8 * b);
    }
    return w;
}

// One step of the systolic sweep: hand-off from the lane above (DPP, full EXEC), then this lane's column j = t - lane.
// PHASE 0: ramp-up (t < 64: lanes with j < 1 wait), 1: steady state (every lane is inside 1..Lj-1: no mask at all),
// 2: tail (lanes may be on the last column, or past it).
template <int R, bool PACKED, int PHASE, int MODE>
__device__ __forceinline__ void c2_dp_step(c2_strip<R>& S, const int t, const int lane, const int Lj, const int ge, const int g0,
                                           const int min_score, const bool first, const bool feeds_next,
                                           const int rsym, int& nM, int& nJ, int& nH,
                                           const unsigned char* sCode, const int16_t* sTbl, int* sBnd,
                                           uint16_t* planePtr, const int colStride, const int band_lanes)
{
    // lane 0's inputs: the row above the pass at column t.  First pass: closed-form row 0 (pyx:153-176): M = J = min_score
    // (they sit in lane 0 of S.upM / S.upJ since the pass started; the DPP never writes lane 0), H(0,t) = iScore[0,t].
    if (first) {
        const int bH = c2_imax(min_score, ge * t + g0);
        if (lane == 0) S.upH = bH;
    } else {
        if (lane == 0) { S.upM = nM; S.upJ = nJ; S.upH = nH; }
        const int tn = (t + 1 <= Lj) ? t + 1 : Lj;
        nM = sBnd[3 * tn]; nJ = sBnd[3 * tn + 1]; nH = sBnd[3 * tn + 2];
    }
    // read symbol of column t: lane (t-1)&63 of rsym holds the symbols of read positions l, l+64, l+128, l+192 of the
    // current 256-column chunk (v_readlane + scalar byte extract: no LDS access, nothing to wait for)
    const int pos = t - 1;
    const int bC = (__builtin_amdgcn_readlane(rsym, pos & 63) >> ((pos >> 3) & 24)) & 0xff;
    S.upM = c2_shr1(S.upM, S.Mb);
    S.upJ = c2_shr1(S.upJ, S.Jb);
    S.upH = c2_shr1(S.upH, S.Hb);
    S.cj = c2_shr1(bC, S.cj);
    const int j = t - lane;
    bool active = true;
    if (PHASE == 0) active = (j >= 1);
    if (PHASE == 2) active = (j >= 1 && j <= Lj);
    if (active) {
        c2_dp_column<R, PACKED, PHASE == 2>(S, S.upM, S.upJ, ge, PHASE == 2 && (j == Lj), sTbl);
        if (MODE == 1) {
            const int slot = lane - c2_band_lo(R, band_lanes, t);
            if ((unsigned)slot < (unsigned)c2_band_slots(R, band_lanes)) planePtr[(t - 1) * colStride + slot] = (uint16_t)S.bits;
        } else if (MODE == 2) {
            planePtr[(t - 1) * 64 + lane] = (uint16_t)S.bits;             // (HBM: one 128-byte line per step)
        } else {
            planePtr[(j - 1) * colStride + lane] = (uint16_t)S.bits;
        }
        if (feeds_next && lane == 63) { sBnd[3 * j] = S.Mb; sBnd[3 * j + 1] = S.Jb; sBnd[3 * j + 2] = S.Hb; }
    }
    S.dgsave = S.upH;
}

// One systolic pass over reference rows p*64R+1 .. p*64R+64R.  SINGLE: the reference fits one pass, so the row above
// lane 0 is the closed-form row 0 and nothing is handed to a next pass.
template <int R, bool PACKED, bool SINGLE, int MODE>
__device__ __forceinline__ void c2_dp_pass(const c2_align_args& A, const c2_dev_ref& rf, const unsigned char* sRef,
                                           const unsigned char* sCode, const int16_t* sTbl, int* sBnd, uint16_t* planePtr,
                                           const int colStride, const int lane, const int p, const int passes,
                                           const int Li, const int Lj, const int g0, const int min_score)
{
    const int ROWS_PER_PASS = 64 * R;
    const int ge = A.gap_extend, go = A.gap_open;
    const int row0 = p * ROWS_PER_PASS + lane * R;      // 0-based row above this lane's strip
    c2_strip<R> S;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = row0 + r + 1;                      // 1-based reference row
        const int ic = i <= Li ? i : Li;                 // padding rows below Li compute garbage nobody reads
        const int gi = rf.gap_incentive[ic], gim1 = rf.gap_incentive[ic - 1];
        const bool last_row = (i == Li);
        // last row: gap_open is replaced by gap_extend (pyx:277-317)
        S.a[r] = (last_row ? ge : go) + gi;              // I opened from M
        S.b[r] = ge + gi;                                // I extended (incentive on every extension, pyx:197)
        S.c[r] = (last_row ? ge : go) + gim1;            // J opened from M (incentive only on open, pyx:205-207)
        S.delta[r] = last_row ? 0 : (ge - go);
        const int rcode = (int)A.code_of_char[sRef[ic - 1]];
        S.sel[r] = PACKED ? (int)A.score_pk[rcode] : rcode * A.n_codes;
        const int J0 = ge * i + g0;                      // jScore[i,0], pyx:170-171
        S.Ml[r] = min_score; S.Il[r] = min_score;        // mScore[i,0], iScore[i,0]
        S.Hl[r] = c2_imax(min_score, J0);
    }
    S.Mb = min_score; S.Jb = ge * (row0 + R) + g0; S.Hb = S.Hl[R - 1];
    // diagonal input of the strip's first row at its first column: H(row0, 0)
    S.dgsave = (row0 == 0) ? 0 : c2_imax(min_score, ge * row0 + g0);
    S.cj = 0;
    S.bits = 0;
    S.upM = min_score; S.upJ = min_score; S.upH = 0;
    const int nrows = (Li - p * ROWS_PER_PASS) < ROWS_PER_PASS ? (Li - p * ROWS_PER_PASS) : ROWS_PER_PASS;
    const int nl = (nrows + R - 1) / R;
    const int steps = Lj + nl - 1;
    const bool first = SINGLE || (p == 0);
    const bool feeds_next = !SINGLE && (p + 1 < passes);

    // values entering lane 0 at step t (column t of the row above the pass), fetched one step ahead
    int rsym = 0;
    int nM = min_score, nJ = min_score, nH = 0;
    if (!first) { nM = sBnd[3]; nJ = sBnd[4]; nH = sBnd[5]; }
    // ramp-up: steps 1 .. min(64, Lj)-1;  steady state: 64 .. Lj-1 (all 64 lanes inside the matrix, not on its last column);
    // tail: Lj .. steps
    const int t1 = Lj < 64 ? Lj : 64;
    const int t2 = Lj < steps + 1 ? Lj : steps + 1;
    int t = 1;
    while (t <= steps) {
        // one 256-column chunk of read symbols per register (c2_load_rsym), then the steps that consume it
        const int seg_end = (((t - 1) | 255) + 1) < steps ? (((t - 1) | 255) + 1) : steps;
        rsym = c2_load_rsym<PACKED>(sCode, (t - 1) & ~255, Lj, lane);
        for (; t <= seg_end && t < t1; ++t)
            c2_dp_step<R, PACKED, 0, MODE>(S, t, lane, Lj, ge, g0, min_score, first, feeds_next, rsym, nM, nJ, nH, sCode, sTbl, sBnd, planePtr, colStride, A.band_lanes);
        for (; t <= seg_end && t < t2; ++t)
            c2_dp_step<R, PACKED, 1, MODE>(S, t, lane, Lj, ge, g0, min_score, first, feeds_next, rsym, nM, nJ, nH, sCode, sTbl, sBnd, planePtr, colStride, A.band_lanes);
        for (; t <= seg_end; ++t)
            c2_dp_step<R, PACKED, 2, MODE>(S, t, lane, Lj, ge, g0, min_score, first, feeds_next, rsym, nM, nJ, nH, sCode, sTbl, sBnd, planePtr, colStride, A.band_lanes);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pieces shared by the row-strip kernel (c2_align_classify_kernel) and the diagonal-band kernel (c2_align_diag_kernel)
// ---------------------------------------------------------------------------------------------------------------
struct c2_wg {                       // one workgroup's LDS views
    unsigned char* sRead;            // read characters (reverse-complemented if the task asks for it)
    unsigned char* sCode;            // their score-table codes
    unsigned char* sRef;             // reference characters
    uint16_t* sIncP;                 // window prefix counts of the reference
    unsigned char* sTmpRead;         // aligned strings, reversed (as the traceback emits them)
    unsigned char* sTmpRef;
};

// Pull the next task index from the device counter (chunks of C2_TASK_CHUNK per atomic), so a launch never waits for the
// slowest statically assigned share and does not depend on how many workgroups are resident.  Returns false when done.
__device__ __forceinline__ bool c2_next_task(const c2_align_args& A, const int lane, uint64_t& chunk_base, int& chunk_left, uint64_t& task) {
    const uint64_t n_iter = A.task_list ? (uint64_t)(*A.task_count) : A.n_tasks;
    if (chunk_left == 0) {
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(A.work_counter, (unsigned long long)C2_TASK_CHUNK);
        chunk_base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                     (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
        chunk_left = C2_TASK_CHUNK;
    }
    const uint64_t it = chunk_base;
    if (it >= n_iter) return false;
    ++chunk_base; --chunk_left;
    task = A.task_list ? (uint64_t)A.task_list[it] : it;
    return true;
}

// The next task's descriptor and the first 256 bytes of its read, requested from HBM while the current task is still in
// its DP (nothing waits for these loads until c2_commit_task uses them): one task of software pipelining per workgroup.
struct c2_prefetch {
    uint64_t task, off;
    int valid, Lj, ref_id, rc;
    unsigned b4;                     // read bytes lane, lane+64, lane+128, lane+192 in bytes 0..3 (already reversed for rc tasks)
};

__device__ __forceinline__ void c2_prefetch_issue(const c2_align_args& A, const int lane, uint64_t& chunk_base, int& chunk_left,
                                                  c2_prefetch& pf)
{
    // (locals, one unconditional struct assignment at the end: keeps the struct in registers)
    uint64_t task = 0, off = 0;
    int Lj = 0, ref_id = 0, rc = 0;
    unsigned b4 = 0;
    const bool valid = c2_next_task(A, lane, chunk_base, chunk_left, task);
    if (valid) {
        uint64_t read_id;
        if (A.all_refs) { read_id = task / (uint64_t)A.n_refs; ref_id = (int)(task % (uint64_t)A.n_refs); }
        else            { read_id = task; ref_id = A.ref_ids ? (int)A.ref_ids[task] : 0; }
        rc = A.strands ? (int)A.strands[task] : 0;
        off = A.offsets[read_id];
        Lj = (int)(A.offsets[read_id + 1] - off);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 64 * q + lane;
            const unsigned byte = (k < Lj) ? (unsigned)A.reads[off + (uint64_t)(rc ? Lj - 1 - k : k)] : 0u;
            b4 |= byte << (8 * q);
        }
    }
    pf.task = task; pf.off = off; pf.valid = valid ? 1 : 0; pf.Lj = Lj; pf.ref_id = ref_id; pf.rc = rc; pf.b4 = b4;
}

// Stage the prefetched task in LDS: reference (only when the amplicon changes), read characters (reverse complement on
// request, CRISPRessoShared.py:399-403) and their codes.  Returns the wave-uniform status bits.  max_li / A.max_lj bound
// what may be written.  sCodeOf: the 256-entry character -> code table, in LDS.
// Stage a reference in its LDS slot (characters, window prefix counts; `bad`: a character outside the score matrix) -- done
// only when the amplicon of a slot changes.  sWin (optional): per dword of four reference positions, bit 7 of byte b set iff
// position 4k + b lies in the quantification window (what c2_emit_gapless4 tests substitutions against).
__device__ __forceinline__ void c2_stage_ref(const c2_align_args& A, const c2_wg& W, const unsigned char* sCodeOf, const int ref_id, const int lane,
                                             const int max_li, int& Li, int& g0, int& ref_bad, uint32_t* sWin = nullptr)
{
    const c2_dev_ref rf = A.refs[ref_id];
    Li = rf.len;
    g0 = rf.gap_incentive[0];
    const int LiLoad = Li < max_li ? Li : max_li;
    int bad = 0;                                            // a reference character outside the score matrix: checked when the
    for (int k = lane; k < LiLoad; k += 64) {               // reference is staged, remembered with it (ref_bad)
        const unsigned char ch = rf.seq[k];
        W.sRef[k] = ch;
        if ((int)sCodeOf[ch] >= A.first_ext_code) bad = 1;      // ord >= matrix dimension (codes >= first_ext_code are read-only symbols)
    }
    ref_bad = __ballot(bad) ? 1 : 0;
    for (int k = lane; k < LiLoad + 2; k += 64) W.sIncP[k] = rf.inc_prefix[k];
    if (sWin) {
        for (int k = lane; 4 * k < LiLoad; k += 64) {
            uint32_t m = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = 4 * k + b;
                if (c < LiLoad && rf.inc_prefix[c + 1] != rf.inc_prefix[c]) m |= 0x80u << (8 * b);
            }
            sWin[k] = m;
        }
    }
}

// code_shift / code_or (packed kernel): the column table holds pair symbols, code A << 5 | code B << 2 -- the first alignment of
// a lane group writes its codes shifted by 5 (and the zeros around them), the second ORs its codes in shifted by 2.
template <bool HAVE_B4 = true>
__device__ __forceinline__ int c2_commit_task(const c2_align_args& A, const c2_wg& W, const unsigned char* sCodeOf, const c2_prefetch& pf,
                                              const int lane, const int max_li, int& cur_ref, int& Li, int& g0, int& ref_bad, bool& packed,
                                              unsigned char* sCodes4 = nullptr, uint32_t* sWin = nullptr, const int code_shift = 2, const bool code_or = false,
                                              const bool stage_ref = true)
{
    // sCodes4 (multi-alignment kernel): the zero-padded table of 4 * code per column, written in the same pass -- columns
    // 1 .. Lj at sCodes4[C2_DIAG_CODE_PAD + 1 ..]; the zeros in front are written once per kernel, the nine behind per task
    const int Lj = pf.Lj, rc = pf.rc;
    int status = 0;
    int read_code_max = 0;
    const int LjLoad = Lj < A.max_lj ? Lj : A.max_lj;       // never write past the LDS plan
    if (stage_ref && pf.ref_id != cur_ref) {
        cur_ref = pf.ref_id;
        c2_stage_ref(A, W, sCodeOf, pf.ref_id, lane, max_li, Li, g0, ref_bad, sWin);
    }
    for (int k = lane; k < LjLoad; k += 64) {
        unsigned char ch;
        if (HAVE_B4 && k < 256) ch = (unsigned char)((pf.b4 >> ((k >> 6) * 8)) & 0xffu);
        else ch = A.reads[pf.off + (uint64_t)(rc ? Lj - 1 - k : k)];
        if (rc) {
            unsigned char cc = (unsigned char)c2_fq_complement(ch);     // seq.upper(), then the dictionary; 0: a character outside it
            if (cc == 0) { status |= C2_STATUS_RC_CHAR; cc = 'N'; }
            ch = cc;
        }
        const unsigned char code = sCodeOf[ch];
        if (code == C2_INVALID_CODE) status |= C2_STATUS_OOB_CHAR;
        W.sRead[k] = ch;
        if (sCodes4) {                                                               // (the multi-alignment kernels read only this table)
            const unsigned char cv = (unsigned char)((code & 7u) << code_shift);
            if (code_or) sCodes4[C2_DIAG_CODE_PAD + 1 + k] |= cv; else sCodes4[C2_DIAG_CODE_PAD + 1 + k] = cv;
        }
        else W.sCode[k] = code;
        read_code_max = read_code_max > (int)code ? read_code_max : (int)code;
    }
    if (sCodes4 && !code_or && lane < 16) sCodes4[C2_DIAG_CODE_PAD + 1 + LjLoad + lane] = 0;   // (16: the packed kernel's OR pass touches whole dwords)
    if (ref_bad) status |= C2_STATUS_OOB_CHAR;
    if (Li <= 0 || Lj <= 0) status |= C2_STATUS_EMPTY;
    if (Lj > A.max_lj || Li > max_li) status |= C2_STATUS_TOO_LONG;
    if (__ballot(status != 0))                                  // (rare: one ballot decides for the usual task)
        status = (__ballot(status & C2_STATUS_EMPTY) ? C2_STATUS_EMPTY : 0) |
                 (__ballot(status & C2_STATUS_OOB_CHAR) ? C2_STATUS_OOB_CHAR : 0) |
                 (__ballot(status & C2_STATUS_RC_CHAR) ? C2_STATUS_RC_CHAR : 0) |
                 (__ballot(status & C2_STATUS_TOO_LONG) ? C2_STATUS_TOO_LONG : 0);
    // packed = every read symbol has a code < 8 and every score fits a signed nibble: the score row of a reference
    // base is then one register and a lookup is one v_bfe_i32 (no LDS in the inner loop).
    packed = (A.score_pk != nullptr) && (__ballot(read_code_max >= 8) == 0ull);
    if (__ballot(read_code_max >= A.first_ext_code && read_code_max != (int)C2_INVALID_CODE)) {
        // a read character beyond the matrix dimension: the reference reads the flat element ci * dim + cj (pyx:212, bounds
        // checking off), which exists iff the LARGEST reference character keeps it inside the buffer (rare path: one more pass)
        const int lim = A.mat_dim * A.mat_dim - A.refs[pf.ref_id].max_char * A.mat_dim;
        bool oob = false;
        for (int k = lane; k < LjLoad; k += 64) if ((int)W.sRead[k] >= lim) oob = true;
        if (__ballot(oob)) status |= C2_STATUS_OOB_CHAR;
    }
    return status;
}

__device__ __forceinline__ void c2_clear_record(c2_aln_record& rec, const int rc, const int ref_id) {
    rec.aln_len = 0; rec.matches = 0; rec.insertion_n = 0; rec.deletion_n = 0; rec.substitution_n = 0;
    rec.all_insertion_events = 0; rec.win_insertion_events = 0; rec.all_deletion_events = 0;
    rec.win_deletion_events = 0; rec.all_deletion_bases = 0; rec.all_substitutions = 0;
    rec.irregular_ends = 0; rec.status = 0; rec.strand = (uint8_t)rc; rec.reserved0 = 0; rec.ref_id = (uint16_t)ref_id;
    rec.reserved2 = 0;
}

// Pointer plane of the row-strip kernel: nibble of cell (pi, pj), pi, pj >= 1.
template <int R, int MODE>
struct c2_row_plane {
    static constexpr bool kWordRuns = false;
    const uint16_t* sPtr; int pass_halfwords, colStride, band_lanes;
    __device__ __forceinline__ bool fetch(const int pi, const int pj, unsigned& nib) const {
        const int pp = (pi - 1) / (64 * R), rem = (pi - 1) % (64 * R);
        bool inb;
        const int pidx = c2_ptr_index<R, MODE>(rem / R, pj, colStride, band_lanes, inb);
        if (!inb) return false;
        const unsigned hw = sPtr[(size_t)pp * (size_t)pass_halfwords + pidx];
        nib = (hw >> (4 * (R - 1 - rem % R))) & 0xF;
        return true;
    }
};

// Traceback (pyx:338-421), wave-parallel: the wave-uniform state (i, j, s) advances one RUN at a time -- lane k probes the
// k-th cell ahead along the current direction (diagonal for M, row for I, column for J), a ballot gives the length of the
// run that stays in s, its columns are emitted by the lanes in one shot.  Emits the aligned strings reversed into
// W.sTmpRead / W.sTmpRef.  need_full: a pointer word that decides the path is not stored in this plane.
template <class PLANE>
__device__ __forceinline__ void c2_traceback(const PLANE& P, const c2_wg& W, const int Li, const int Lj, const int min_score,
                                             const int ge, const int g0, const int lane,
                                             int& cnt, int& matches, int& status, bool& need_full)
{
    int i = Li, j = Lj;
    int s = C2_ST_M;
    cnt = 0; matches = 0; need_full = false;
    {
        unsigned nib = 0;
        if (P.fetch(i, j, nib)) s = (nib & 2) ? C2_ST_I : ((nib & 1) ? C2_ST_J : C2_ST_M);   // start state, pyx:349-358
        else need_full = true;
    }
    while (!need_full && (i > 0 || j > 0)) {
        if (i == 0 || j == 0) {
            const int need = (i == 0) ? C2_ST_I : C2_ST_J;           // initialised chains: iPointer[0,1:], jPointer[1:,0]
            if (s != need) { status |= (s == C2_ST_M) ? C2_STATUS_SENTINEL_PATH : C2_STATUS_UNINIT_PTR; break; }
            const int len = (i == 0) ? j : i;
            for (int k = lane; k < len; k += 64) {
                W.sTmpRead[cnt + k] = (i == 0) ? W.sRead[j - 1 - k] : (unsigned char)'-';
                W.sTmpRef[cnt + k] = (i == 0) ? (unsigned char)'-' : W.sRef[i - 1 - k];
            }
            cnt += len; i = 0; j = 0;
            break;
        }
        if constexpr (PLANE::kWordRuns) {
            // A run of state M in the packed kernels' words, FOUR cells per lane: the cells that decide it, (i-1-k, j-1-k), lie on one diagonal,
            // every other anti-diagonal -- four of them in each word of the diagonal's lane slot.  "The path stays in M behind the cell" is two
            // bits of its word (NOT "H is I" and NOT "J beats M", c2_pk_push4), so a lane tests its word's four cells with a mask, and the
            // lanes together see 250 cells at once where the probe below sees 64.  Interior cells only (k < min(i, j) - 1): the cell on a
            // matrix edge, and every other state, go through the probe below.
            const int K = (i < j ? i : j) - 1;
            const int slw = (i - j - P.d0) >> 1;
            if (s == C2_ST_M && P.pk && K >= 1 && (unsigned)slw < (unsigned)P.nl) {
                const int a0 = i + j - 2, par = a0 & 1, ctop = a0 & 7, W0 = a0 >> 3;
                const int n0 = (ctop >> 1) + 1;                           // path cells in word W0: c = ctop, ctop - 2, ..
                const int wi = W0 - lane;                                 // this lane's word
                const int kfirst = lane == 0 ? 0 : n0 + 4 * (lane - 1);   // k of its topmost cell
                int cells = lane == 0 ? n0 : 4;
                if (wi < 0 || kfirst >= K) cells = 0; else if (kfirst + cells > K) cells = K - kfirst;
                unsigned w = 0;
                if (cells > 0) w = P.words[wi * P.lpa + slw];
                const unsigned u = ((w & (w >> 8)) >> (2 * par)) & 0x00110011u;      // cell 6+par -> bit 20, 4+par -> 16, 2+par -> 4, par -> 0
                unsigned m4 = (((u >> 20) & 1u) << 3) | (((u >> 16) & 1u) << 2) | (((u >> 4) & 1u) << 1) | (u & 1u);   // topmost cell first
                const int ctop_l = lane == 0 ? ctop : 6 + par;            // the lane's topmost cell
                if (lane == 0) m4 = (m4 << (4 - n0)) & 0xfu;
                int n_lead = __builtin_clz((((~m4) & 0xfu) << 28) | 0x08000000u);    // cells from the top that keep the path in M (4: all of them)
                if (n_lead > cells) n_lead = cells;
                const unsigned long long stop = __ballot(n_lead < cells);
                const int c_dec = ctop_l - 2 * n_lead;                    // the cell that ends the run, if it is this lane's
                const int ns_dec = ((w >> (16 * ((c_dec & 7) >> 2) + 2 * (c_dec & 3))) & 1u) ? C2_ST_J : C2_ST_I;     // "H is I" first (pyx:349-358 order)
                int covered = n0 + 4 * 63; if (covered > K) covered = K;  // cells the 64 words hold
                int E, s_next;
                if (stop == 0ull) { E = covered; s_next = C2_ST_M; }
                else {
                    const int wl = __builtin_ctzll(stop);
                    E = (wl == 0 ? 0 : n0 + 4 * (wl - 1)) + __builtin_amdgcn_readlane(n_lead, wl) + 1;
                    s_next = __builtin_amdgcn_readlane(ns_dec, wl);
                }
                for (int base = 0; base < E; base += 64) {
                    const int k = base + lane;
                    unsigned char rch = 0, fch = 1;
                    if (k < E) {
                        rch = W.sRead[j - 1 - k]; fch = W.sRef[i - 1 - k];
                        W.sTmpRead[cnt + k] = rch; W.sTmpRef[cnt + k] = fch;
                    }
                    matches += __popcll(__ballot(k < E && rch == fch));               // pyx:375-376
                }
                cnt += E; i -= E; j -= E; s = s_next;
                continue;
            }
        }
        const int di = (s != C2_ST_I) ? 1 : 0, dj = (s != C2_ST_J) ? 1 : 0;
        const int ik = i - lane * di, jk = j - lane * dj;
        const bool valid = (ik >= 1) && (jk >= 1);
        int ns = 0;
        bool oob = false;
        if (valid) {
            // cell whose pointer nibble decides the next state
            const int pi = (s == C2_ST_M) ? ik - 1 : ik, pj = (s == C2_ST_M) ? jk - 1 : jk;
            if (pi == 0 || pj == 0) {
                ns = c2_boundary_hstate(pi, pj, min_score, ge, g0);   // only reachable for s == M
            } else {
                unsigned nib = 0;
                if (P.fetch(pi, pj, nib)) {
                    if (s == C2_ST_M) ns = (nib & 2) ? C2_ST_I : ((nib & 1) ? C2_ST_J : C2_ST_M);
                    else if (s == C2_ST_I) ns = (nib & 8) ? C2_ST_M : C2_ST_I;
                    else ns = (nib & 4) ? C2_ST_M : C2_ST_J;
                } else oob = true;                                    // not stored: ns stays 0 (never == s)
            }
        }
        const unsigned long long vmask = __ballot(valid);
        const unsigned long long cmask = __ballot(valid && ns == s);
        const unsigned long long omask = __ballot(oob);
        const int nv = (~vmask == 0ull) ? 64 : __builtin_ctzll(~vmask);
        const int nc = (~cmask == 0ull) ? 64 : __builtin_ctzll(~cmask);
        int E, s_next;
        if (nc < nv) {
            // lane nc decides the next state; if its pointer word is not stored, another kernel must redo this task
            if ((omask >> nc) & 1ull) { need_full = true; break; }
            E = nc + 1; s_next = __builtin_amdgcn_readlane(ns, nc);
        } else { E = nv; s_next = s; }
        unsigned char rch = '-', fch = '-';
        if (lane < E) {
            if (s != C2_ST_J) rch = W.sRead[jk - 1];
            if (s != C2_ST_I) fch = W.sRef[ik - 1];
            W.sTmpRead[cnt + lane] = rch;
            W.sTmpRef[cnt + lane] = fch;
        }
        if (s == C2_ST_M) matches += __popcll(__ballot(lane < E && rch == fch));   // pyx:375-376
        cnt += E; i -= E * di; j -= E * dj; s = s_next;
    }
}

// Aligned strings out (reversed copy, pyx:434) + fused classification (COREResources.pyx:68-187 and the derived counters
// of CRISPRessoCORE.py:726-760) from the strings still in LDS.  Column c (forward order) = tmp[T-1-c].  The aligner never
// emits a double-gap column and never puts an insertion column next to a deletion column (I and J only hand over to M),
// so every gap run is pure and idx advances by one on every non-insertion column.
__device__ __forceinline__ void c2_emit_and_classify(const c2_align_args& A, const c2_wg& W, const uint64_t task, const int T,
                                                     const int matches, const int lane, c2_aln_record& rec, const int Li, const int Lj)
{
    const unsigned char* sTmpRead = W.sTmpRead; const unsigned char* sTmpRef = W.sTmpRef; const uint16_t* sIncP = W.sIncP;
    uint8_t* outR = A.aln_read + task * (uint64_t)A.aln_stride;
    uint8_t* outF = A.aln_ref + task * (uint64_t)A.aln_stride;
    if (!(A.reserved & 1)) {                                   // (debug knob: C2_DEBUG_SKIP_STRINGS measures the cost of these stores)
        for (int cidx = lane; cidx < T; cidx += 64) {
            outR[cidx] = sTmpRead[T - 1 - cidx];
            outF[cidx] = sTmpRef[T - 1 - cidx];
        }
    }
    int idx_base = 0, last_rf = -1, last_rd = -1;
    int n_all_sub = 0, n_win_sub = 0, n_all_ins = 0, n_win_ins = 0, n_all_del = 0, n_win_del = 0;
    int acc_ins_n = 0, acc_del_n = 0, acc_del_bases = 0;   // sums over the events (wave-uniform)
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (T == Li && T == Lj) {
        // no gap column in either string (T = Li + insertion columns = Lj + deletion columns): the reference index of a
        // column is the column, and only substitutions can occur -- most reads of an amplicon run take this path
        for (int base = 0; base < T; base += 64) {
            const int cidx = base + lane;
            const bool in = cidx < T;
            const unsigned char rd = in ? sTmpRead[T - 1 - cidx] : 0, rfc = in ? sTmpRef[T - 1 - cidx] : 0;
            const bool sub = in && rd != rfc && rd != 'N';                                  // pyx:113-118
            const bool sub_win = sub && (sIncP[cidx + 1] != sIncP[cidx]);
            n_all_sub += __popcll(__ballot(sub));
            n_win_sub += __popcll(__ballot(sub_win));
        }
        last_rd = T - 1;
    } else
    for (int base = 0; base < T; base += 64) {
        const int cidx = base + lane;
        const bool in = cidx < T;
        const unsigned char rd = in ? sTmpRead[T - 1 - cidx] : 0, rfc = in ? sTmpRef[T - 1 - cidx] : 0;
        const bool rf_ng = in && rfc != '-', rd_ng = in && rd != '-';
        const unsigned long long m_rf = __ballot(rf_ng), m_rd = __ballot(rd_ng);
        {   // 64 columns without a gap, and no gap run open in front of them (most chunks of a traced alignment: its indels sit in one
            // or two places): only substitutions can occur here, and the reference index of a column is idx_base + lane
            const unsigned long long m_in = __ballot(in);
            if (m_rf == m_in && m_rd == m_in && last_rf == base - 1 && last_rd == base - 1) {
                const int idx0 = idx_base + lane;
                const bool sub0 = in && rd != rfc && rd != 'N';
                n_all_sub += __popcll(__ballot(sub0));
                n_win_sub += __popcll(__ballot(sub0 && (sIncP[idx0 + 1] != sIncP[idx0])));
                const int cols = __popcll(m_in);
                idx_base += cols; last_rf = base + cols - 1; last_rd = last_rf;
                continue;
            }
        }
        const int idx = idx_base + __popcll(m_rf & lt);             // ref bases left of this column
        const unsigned long long below_rf = m_rf & lt, below_rd = m_rd & lt;
        const int prev_rf = below_rf ? base + 63 - __clzll((long long)below_rf) : last_rf;
        const int prev_rd = below_rd ? base + 63 - __clzll((long long)below_rd) : last_rd;
        // substitution, pyx:113-118
        const bool sub = rf_ng && rd_ng && rd != rfc && rd != 'N';
        const bool sub_win = sub && (sIncP[idx + 1] != sIncP[idx]);
        n_all_sub += __popcll(__ballot(sub));
        n_win_sub += __popcll(__ballot(sub_win));
        // insertion closes at this column, pyx:119-128; leading insertions (idx==0) are never opened, pyx:136.  (Skipped when the chunk has no
        // gap in the reference string and no such run is open in front of it: no column of it can close one -- a read with one deletion.)
        const unsigned long long m_in2 = __ballot(in);
        if (m_rf != m_in2 || last_rf != base - 1) {
            const bool ins_close = rf_ng && (prev_rf != cidx - 1) && idx > 0;
            // in the window: both flanks (pyx:121) -- the legacy classifier: either flank (pyx:284)
            const bool fl = ins_close && (sIncP[idx] != sIncP[idx - 1]), fr = ins_close && (sIncP[idx + 1] != sIncP[idx]);
            const bool ins_win = A.legacy ? (fl || fr) : (fl && fr);
            n_all_ins += __popcll(__ballot(ins_close));
            // (the sums over events are wave-uniform: an alignment has one or two events, so each is fetched from its lane -- no per-lane partial
            //  sums, and no three six-step reductions at the end for every traced alignment)
            unsigned long long ev = __ballot(ins_win);
            n_win_ins += __popcll(ev);
            const int isz = cidx - 1 - prev_rf;
            while (ev) { const int l = __builtin_ctzll(ev); ev &= ev - 1ull; acc_ins_n += __builtin_amdgcn_readlane(isz, l); }
        }
        // deletion closes at this column, pyx:145-153 (likewise skipped when the read string has no gap here and no run is open)
        if (m_rd != m_in2 || last_rd != base - 1) {
            const bool del_close = rd_ng && (prev_rd != cidx - 1);
            const int dlen = cidx - 1 - prev_rd;
            // legacy (pyx:253-258): a run that starts in column 0 or 1 is given reference start 0 (`if st-1 > 0`)
            const int dstart = (A.legacy && prev_rd <= 0) ? 0 : idx - dlen;
            const bool del_win = del_close && (sIncP[idx] != sIncP[dstart]);       // include set hits range(start,end)
            unsigned long long ev = __ballot(del_close);
            const unsigned long long evw = __ballot(del_win);
            n_all_del += __popcll(ev);
            n_win_del += __popcll(evw);
            const int dbases = idx - dstart;
            while (ev) {
                const int l = __builtin_ctzll(ev);
                ev &= ev - 1ull;
                acc_del_bases += __builtin_amdgcn_readlane(dbases, l);
                if ((evw >> l) & 1ull) acc_del_n += __builtin_amdgcn_readlane(dlen, l);
            }
        }
        idx_base += __popcll(m_rf);
        if (m_rf) last_rf = base + 63 - __clzll((long long)m_rf);
        if (m_rd) last_rd = base + 63 - __clzll((long long)m_rd);
    }
    // trailing deletion, pyx:155-162
    int tr_bases = 0, tr_win = 0;
    if (last_rd != T - 1) {
        const int dlen = T - 1 - last_rd;
        n_all_del += 1;
        if (!A.legacy) {
            tr_bases = dlen;
            if (sIncP[idx_base] != sIncP[idx_base - dlen]) { tr_win = dlen; n_win_del += 1; }
        } else {
            // legacy (pyx:259-261): a run that reaches the end of the alignment ends at reference index idx - 1 (exclusive), and
            // starts at 0 if it begins in column 0 or 1
            const int dstart = last_rd <= 0 ? 0 : idx_base - dlen, dend = idx_base - 1;
            tr_bases = dend > dstart ? dend - dstart : 0;
            if (dend > dstart && sIncP[dend] != sIncP[dstart]) { tr_win = dlen; n_win_del += 1; }
        }
    }
    const unsigned char r0 = sTmpRead[T - 1], f0 = sTmpRef[T - 1], rL = sTmpRead[0], fL = sTmpRef[0];
    rec.irregular_ends = (r0 == '-' || f0 == '-' || r0 != f0 || rL == '-' || fL == '-' || rL != fL) ? 1 : 0;
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
}

// four bytes from LDS at any byte offset `base` of the 4-byte-aligned buffer `buf` (bytes base .. base + 3; `last_word`: the last dword that holds
// a byte of the buffer -- one dword in front of the buffer and one behind that are read, never used: the LDS plan has them)
__device__ __forceinline__ unsigned c2_lds_load4(const unsigned char* buf, const int base, const int last_word) {
    int w4 = base >> 2;
    w4 = w4 < -1 ? -1 : (w4 > last_word ? last_word : w4);
    const uint32_t* p = (const uint32_t*)buf + w4;
    return __builtin_amdgcn_alignbyte(p[1], p[0], (unsigned)base & 3u);
}
__device__ __forceinline__ unsigned c2_nonzero_bytes(const unsigned x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }   // bit 7 of every byte of x that is not 0

// Shortcut for the commonest alignment of an amplicon run: equal lengths and no gap at all.  The reference's traceback stays
// in state M from (L, L) to (0, 0) iff the H-state of every cell (i, i) is M, i.e. the two low pointer bits of all L main-
// diagonal cells are clear (start-state rule pyx:349-358 for (L, L); Mptr(i+1, i+1) = H-state of (i, i) for the rest; the
// boundary cell (0, 0) is M).  64 lanes read those nibbles in ceil(L/64) probes and one ballot decides; the aligned
// strings are then the read and the reference themselves, and only substitutions can occur.  Returns false (nothing
// written) if the path leaves the diagonal or a nibble is not stored in this plane.
__device__ __forceinline__ void c2_emit_gapless(const c2_align_args& A, const c2_wg& W, const uint64_t task, const int L, const int lane,
                                                c2_aln_record& rec);
template <class PLANE>
__device__ __forceinline__ bool c2_try_gapless(const PLANE& P, const c2_align_args& A, const c2_wg& W, const uint64_t task, const int L,
                                               const int lane, c2_aln_record& rec)
{
    bool off = false;
    for (int base = 0; base < L; base += 64) {
        const int i = base + lane + 1;
        if (i <= L) { unsigned nib = 0; if (!P.fetch(i, i, nib) || (nib & 3u)) off = true; }
    }
    if (__ballot(off)) return false;
    c2_emit_gapless(A, W, task, L, lane, rec);
    return true;
}

// c2_emit_gapless for the multi-alignment kernels, four columns per lane: the read and the reference leave LDS as dwords and
// go out as dwords (rows are 16-byte aligned: the caller checks the base pointers); mismatching bytes are found with the
// "has a zero byte" bit trick on read ^ reference, the few lanes that hold one are visited with scalar code.
// sWin: c2_stage_ref's window masks.  L <= 256 (the caller checks).
__device__ __forceinline__ void c2_emit_gapless4(const c2_align_args& A, const c2_wg& W, const uint32_t* sWin, const uint64_t task, const int L,
                                                 const int lane, c2_aln_record& rec)
{
    const int p = 4 * lane;
    const int nb = L - p;                                             // valid bytes of this lane's dword
    const uint32_t valid = nb >= 4 ? 0xffffffffu : (nb > 0 ? ((1u << (8 * nb)) - 1u) : 0u);
    const uint32_t rd = ((const uint32_t*)W.sRead)[lane] & valid, rf = ((const uint32_t*)W.sRef)[lane] & valid;
    if (!(A.reserved & 1) && nb > 0) {
        ((uint32_t*)(A.aln_read + task * (uint64_t)A.aln_stride))[lane] = rd;      // (a partial last dword is padded with zeros: the row has room, aln_stride is a multiple of 16)
        ((uint32_t*)(A.aln_ref + task * (uint64_t)A.aln_stride))[lane] = rf;
    }
    const uint32_t x = rd ^ rf;
    const uint32_t mm = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;      // bit 7 of every byte in which read and reference differ
    int mism = 0, n_all_sub = 0, n_win_sub = 0;
    unsigned long long todo = __ballot(mm != 0);
    // (c2_batch.diag_hints: a gap-free alignment with at most three differing columns leaves a hint too -- one run of M and the columns; see c2_group_epilogue)
    const bool want_hint = A.diag_hints != nullptr && L <= 511;
    unsigned ent[3] = {0u, 0u, 0u};
    int n_ent = 0;
    if (todo) {
        const uint32_t y = rd ^ 0x4e4e4e4eu;                                         // COREResources.pyx:113-118: a read 'N' is no substitution
        const uint32_t sub = mm & ((((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u);
        const uint32_t win = sWin ? (sub & sWin[lane]) : 0u;
        while (todo) {                                                               // (a read of an amplicon run differs in a lane or two)
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (want_hint) {
                const unsigned mml = (unsigned)__builtin_amdgcn_readlane((int)mm, l), rdl = (unsigned)__builtin_amdgcn_readlane((int)rd, l);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((mml >> (8 * b + 7)) & 1u) {
                        const unsigned ch = (rdl >> (8 * b)) & 0xffu, code = (ch >> 1) & 7u;
                        const unsigned long long chars = (unsigned long long)'A' | ((unsigned long long)'C' << 8) | ((unsigned long long)'T' << 16) | ((unsigned long long)'G' << 24) | ((unsigned long long)'N' << 56);
                        if (((unsigned)(chars >> (8 * code)) & 0xffu) != ch) n_ent = 64;          // (no base: no hint)
                        else { if (n_ent < 3) ent[n_ent] = (unsigned)(4 * l + b) | (code << 9); ++n_ent; }
                    }
            }
            mism += __builtin_popcount((unsigned)__builtin_amdgcn_readlane((int)mm, l));
            const unsigned subl = (unsigned)__builtin_amdgcn_readlane((int)sub, l);
            n_all_sub += __builtin_popcount(subl);
            if (sWin) n_win_sub += __builtin_popcount((unsigned)__builtin_amdgcn_readlane((int)win, l));
            else if (subl) {                                                         // no mask table (packed kernel): the window prefix counts of the lane's four positions
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((subl >> (8 * b + 7)) & 1u) n_win_sub += (W.sIncP[4 * l + b + 1] != W.sIncP[4 * l + b]) ? 1 : 0;
            }
        }
    }
    const unsigned char r0 = W.sRead[0], f0 = W.sRef[0], rL = W.sRead[L - 1], fL = W.sRef[L - 1];
    rec.irregular_ends = (r0 == '-' || f0 == '-' || r0 != f0 || rL == '-' || fL == '-' || rL != fL) ? 1 : 0;
    rec.aln_len = (uint16_t)L;
    rec.matches = (uint16_t)(L - mism);                                              // pyx:375-376
    rec.substitution_n = (uint16_t)n_win_sub;
    rec.all_substitutions = (uint16_t)n_all_sub;
    if (want_hint && n_ent <= 3 && lane == 0) {
        uint32_t* hp = A.diag_hints + 4u * task;
        hp[0] = C2_HINT_GAPPED | 1u | ((unsigned)n_ent << 3) | (((unsigned)C2_ST_M | ((unsigned)L << 2)) << 5);
        hp[1] = 0u;
        hp[2] = ent[0] << 11;
        hp[3] = ent[1] | (ent[2] << 12);
    }
}

// A gap-free alignment of two sequences of equal length: the aligned strings are the read and the reference themselves, and
// only substitutions can occur.  Strings out, counts into `rec`.
__device__ __forceinline__ void c2_emit_gapless(const c2_align_args& A, const c2_wg& W, const uint64_t task, const int L, const int lane,
                                                c2_aln_record& rec)
{
    uint8_t* outR = A.aln_read + task * (uint64_t)A.aln_stride;
    uint8_t* outF = A.aln_ref + task * (uint64_t)A.aln_stride;
    const bool strings = !(A.reserved & 1);
    int matches = 0, n_all_sub = 0, n_win_sub = 0;
    for (int base = 0; base < L; base += 64) {
        const int c = base + lane;
        const bool in = c < L;
        const unsigned char rd = in ? W.sRead[c] : 0, rf = in ? W.sRef[c] : 0;
        if (in && strings) { outR[c] = rd; outF[c] = rf; }
        matches += __popcll(__ballot(in && rd == rf));                                   // pyx:375-376
        const bool sub = in && rd != rf && rd != 'N';                                    // COREResources.pyx:113-118 (no '-' here)
        n_all_sub += __popcll(__ballot(sub));
        n_win_sub += __popcll(__ballot(sub && (W.sIncP[c + 1] != W.sIncP[c])));
    }
    const unsigned char r0 = W.sRead[0], f0 = W.sRef[0], rL = W.sRead[L - 1], fL = W.sRef[L - 1];
    rec.irregular_ends = (r0 == '-' || f0 == '-' || r0 != f0 || rL == '-' || fL == '-' || rL != fL) ? 1 : 0;
    rec.aln_len = (uint16_t)L;
    rec.matches = (uint16_t)matches;
    rec.substitution_n = (uint16_t)n_win_sub;
    rec.all_substitutions = (uint16_t)n_all_sub;
}

// ---------------------------------------------------------------------------------------------------------------
// Row-strip kernel.  BAND = false: full pointer plane (any path).  BAND = true: only the lanes within A.band_lanes of the
// main diagonal keep their pointer words (single-pass references only); a traceback that needs a word outside the band
// appends the task to A.fb_list, and the host re-runs exactly those tasks with the full-plane kernel (A.task_list mode).
// The band limits what is STORED, never what is computed, so results do not depend on it.
// ---------------------------------------------------------------------------------------------------------------
#ifndef C2_FULL_WAVES
#define C2_FULL_WAVES 4                             // wavefronts per SIMD the full-matrix kernel is compiled for (its registers; measured: see profiles/r05/README.md)
#endif
template <int R, int MODE>
__global__ __launch_bounds__(64, C2_FULL_WAVES) void c2_align_classify_kernel(c2_align_args A)
{
    constexpr bool BAND = MODE == 1;
    constexpr int MULTI = MODE == 2 ? 2 : 0;                     // plane mode of the multi-pass sweeps (never banded)
    const int lane = threadIdx.x;
    // the last launch of a batch's chain comes in two forms -- plane in LDS (a short list is done sooner: one alignment takes 0.3 ms there, 0.6 ms
    // with the plane in HBM) and plane in HBM scratch (a long list: four wavefronts per SIMD, twice the rate) -- and the host cannot know the
    // list's length: both are launched, and the list decides which of them works
    if (A.task_list && A.list_gate) {
        const uint32_t n = *A.task_count;
        if (A.list_gate > 0 ? n > (uint32_t)A.list_gate : n <= (uint32_t)(-A.list_gate)) return;
    }
    const c2_lds_plan P = c2_make_plan(R, A.max_lj, A.max_passes, A.n_codes, BAND ? A.band_lanes : 0, MODE == 2);
    const int pass_halfwords = MODE == 2 ? (A.max_lj + 64) * 64 : A.max_lj * (int)P.col_stride;
    uint16_t* sPtr = MODE == 2 ? (uint16_t*)(A.plane + (size_t)blockIdx.x * A.plane_words_per_wg) : (uint16_t*)(c2_smem + P.ptr);
    int* sBnd = (int*)(c2_smem + P.bnd);
    int16_t* sTbl = (int16_t*)(c2_smem + P.tbl);
    c2_wg W;
    W.sRead = c2_smem + P.read; W.sCode = c2_smem + P.code; W.sRef = c2_smem + P.ref;
    W.sIncP = (uint16_t*)(c2_smem + P.incp); W.sTmpRead = c2_smem + P.tmp_read; W.sTmpRef = c2_smem + P.tmp_ref;
    unsigned char* sCodeOf = c2_smem + P.codeof;
    const int colStride = (int)P.col_stride;
    const int ROWS_PER_PASS = 64 * R;
    const int ge = A.gap_extend, go = A.gap_open;

    // score table -> LDS, once per workgroup
    for (int k = lane; k < A.n_codes * A.n_codes; k += 64) sTbl[k] = A.score_tbl[k];

    for (int k = lane; k < 256; k += 64) sCodeOf[k] = A.code_of_char[k];
    int cur_ref = -1;
    int Li = 0, g0 = 0, ref_bad = 0;
    uint64_t chunk_base = 0;
    int chunk_left = 0;
    c2_phase_acc PH; PH.t_last = 0; PH.sum[0] = PH.sum[1] = PH.sum[2] = PH.sum[3] = 0;
    c2_prefetch pf;
    c2_prefetch_issue(A, lane, chunk_base, chunk_left, pf);
    while (pf.valid) {
        __syncthreads();   // previous task's LDS readers are done
        c2_phase_begin(A.phase_cycles, PH);
        const uint64_t task = pf.task;
        const int Lj = pf.Lj, ref_id = pf.ref_id, rc = pf.rc;
        bool packed;
        int status = c2_commit_task(A, W, sCodeOf, pf, lane, A.max_passes * ROWS_PER_PASS, cur_ref, Li, g0, ref_bad, packed);
        c2_prefetch_issue(A, lane, chunk_base, chunk_left, pf);   // next task's loads fly during this task's DP
        const c2_dev_ref rf = A.refs[ref_id];
        const int passes = (Li + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
        if (BAND && passes != 1) status |= C2_STATUS_TOO_LONG;
        __syncthreads();

        c2_aln_record rec;
        c2_clear_record(rec, rc, ref_id);
        if (status == 0) {
            // pyx:150  int min_score = gap_open * max_j * max_i   (wraps like the reference's C int)
            const int min_score = (int)(uint32_t)((uint64_t)(int64_t)go * (uint64_t)Lj * (uint64_t)Li);
            c2_phase_mark<0>(A.phase_cycles, PH);   // phase 0: task fetch (offsets, read, reference rows)
            // =========================== DP: systolic sweep, pass by pass ===========================
            for (int p = 0; p < passes; ++p) {
                const bool single = (passes == 1);
                uint16_t* planePtr = sPtr + (size_t)p * (size_t)pass_halfwords;
                if (packed) {
                    if (single) c2_dp_pass<R, true, true, MODE>(A, rf, W.sRef, W.sCode, sTbl, sBnd, planePtr, colStride, lane, p, passes, Li, Lj, g0, min_score);
                    else        c2_dp_pass<R, true, false, MULTI>(A, rf, W.sRef, W.sCode, sTbl, sBnd, planePtr, colStride, lane, p, passes, Li, Lj, g0, min_score);
                } else {
                    if (single) c2_dp_pass<R, false, true, MODE>(A, rf, W.sRef, W.sCode, sTbl, sBnd, planePtr, colStride, lane, p, passes, Li, Lj, g0, min_score);
                    else        c2_dp_pass<R, false, false, MULTI>(A, rf, W.sRef, W.sCode, sTbl, sBnd, planePtr, colStride, lane, p, passes, Li, Lj, g0, min_score);
                }
                __syncthreads();
            }
            if (MODE == 2) __threadfence_block();                 // the traceback reads other lanes' pointer words back from HBM
            c2_phase_mark<1>(A.phase_cycles, PH);   // phase 1: DP fill
            // =========================== traceback ===========================
            c2_row_plane<R, MODE> plane;
            plane.sPtr = sPtr; plane.pass_halfwords = pass_halfwords; plane.colStride = colStride; plane.band_lanes = A.band_lanes;
            if (!(Li == Lj && c2_try_gapless(plane, A, W, task, Li, lane, rec))) {
                int cnt, matches;
                bool need_full;
                c2_traceback(plane, W, Li, Lj, min_score, ge, g0, lane, cnt, matches, status, need_full);
                __syncthreads();
                c2_phase_mark<2>(A.phase_cycles, PH);   // phase 2: traceback
                if (need_full) {                                   // only possible with BAND
                    status |= C2_STATUS_NEED_FULL;
                    if (lane == 0) { const unsigned k = atomicAdd(A.fb_count, 1u); A.fb_list[k] = (uint32_t)task; }
                }
                if (status == 0) c2_emit_and_classify(A, W, task, cnt, matches, lane, rec, Li, Lj);
            }
        }
        rec.status = (uint8_t)status;
        if (lane == 0) A.records[task] = rec;
        c2_phase_mark<3>(A.phase_cycles, PH);       // phase 3: strings out, classification, record
    }
    c2_phase_flush(A.phase_cycles, PH, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Diagonal-band kernel.  Lanes own DIAGONALS instead of rows: lane l owns d = d0 + 2l ("E") and d0 + 2l + 1 ("O"), 128
// diagonals around the one that joins (0,0) and (Li,Lj); the sweep runs over anti-diagonals a = i + j, one cell per lane
// per step (E cells at even a, O cells at odd a), 499 steps x 1 cell instead of 313 steps x 4 cells for 250 x 250.
// Cells outside the band are never computed (they read as -2^30: the number 0 under C2_DIAG_BIAS), which is exact iff no optimal path leaves the
// band.  That is PROVEN per alignment after the fill: c2_outside_band_bound gives the most any path that touches a diagonal
// beyond either band edge can score (its steps down, steps right and diagonal steps priced by the reference's own cost
// rules).  If the banded score H(Li,Lj) exceeds it, every optimal path -- and every path that ties with one at any cell the
// reference's traceback visits -- lies inside the band, where banded and full DP values coincide, so the pointers the
// traceback reads are the full DP's.  Otherwise (and for reads the packed score rows cannot encode) the task goes to the
// fallback list and the next launch of the chain redoes it.  Neighbour traffic: two DPP reads per step (wave_shr at even
// steps, wave_shl at odd steps), folded into the adds that consume them; row constants {a_i, b_i, c_i, score row} come from a zero-padded per-reference table in
// global memory (L2-resident), the column symbols from a zero-padded LDS table, both fetched one group of eight
// anti-diagonals ahead (c2_diagx_fetch).
// ---------------------------------------------------------------------------------------------------------------
#define C2_DPP_WAVE_SHL1 0x130
// lane n receives `src` of lane n+1; lane 63 keeps `old`
__device__ __forceinline__ int c2_shl1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, C2_DPP_WAVE_SHL1, 0xf, 0xf, false);
}
// The diagonal kernels keep every DP value with C2_DIAG_BIAS added, so that "outside the band" is the number 0 -- which is
// what a DPP read with bound_ctrl set returns for a lane without a source (wavefront end, or a source lane switched off in
// EXEC).  A bound_ctrl move with old = 0 folds into the VALU instruction that consumes it (v_add_u32_dpp): the hand-off
// between neighbouring diagonals then costs no instruction of its own.  All recurrences add constants to DP values and
// compare sums of that form, so the bias changes no comparison (values stay within [-2^20, 2^30 + 2^20]).
#if defined(__HIP_DEVICE_COMPILE__)
#define C2_KEEP_IN_VGPR(x) asm volatile("" : "+v"(x))
#else
#define C2_KEEP_IN_VGPR(x) (void)(x)
#endif
__device__ __forceinline__ int c2_shr1z(int src) { return __builtin_amdgcn_update_dpp(0, src, C2_DPP_WAVE_SHR1, 0xf, 0xf, true); }
__device__ __forceinline__ int c2_shl1z(int src) { return __builtin_amdgcn_update_dpp(0, src, C2_DPP_WAVE_SHL1, 0xf, 0xf, true); }
// The same hand-off inside a ROW of 16 lanes (row_shr:1 / row_shl:1): the first / last lane of a row has no source and reads 0.
// A lane group of 16 lanes (eight alignments per wavefront, packed) is exactly a row, so its two ends see "outside the band"
// without a lane being switched off: all 16 lanes hold diagonals (32 per band instead of 30).
#define C2_DPP_ROW_SHR1 0x111
#define C2_DPP_ROW_SHL1 0x101
__device__ __forceinline__ int c2_rshr1z(int src) { return __builtin_amdgcn_update_dpp(0, src, C2_DPP_ROW_SHR1, 0xf, 0xf, true); }
__device__ __forceinline__ int c2_rshl1z(int src) { return __builtin_amdgcn_update_dpp(0, src, C2_DPP_ROW_SHL1, 0xf, 0xf, true); }

struct c2_diag_plan { uint32_t plane, codes, codeof, read, code, ref, incp, tmp_read, tmp_ref, total; uint32_t n_words; };

__host__ __device__ inline c2_diag_plan c2_make_diag_plan(int max_li, int max_lj) {
    c2_diag_plan p;
    p.n_words = (uint32_t)(max_li + max_lj) / 8u + 1u;              // one 32-bit word per lane per 8 anti-diagonals
    uint32_t off = 0;
    p.plane = off;    off += p.n_words * (uint32_t)C2_DIAG_STORE_N * 4u;
    p.codes = off;    off += c2_align16((uint32_t)C2_DIAG_CODE_PAD + (uint32_t)max_lj + 2u + 8u);   // zeros | columns 0 .. Lj+1 | zeros
    p.codeof = off;   off += 256u;
    p.read = off;     off += c2_align16((uint32_t)max_lj);
    p.code = off;     off += c2_align16((uint32_t)max_lj);
    p.ref = off;      off += c2_align16((uint32_t)max_li);
    p.incp = off;     off += c2_align16(((uint32_t)max_li + 2u) * 2u);
    p.tmp_read = off; off += c2_align16((uint32_t)max_li + (uint32_t)max_lj);
    p.tmp_ref = off;  off += c2_align16((uint32_t)max_li + (uint32_t)max_lj);
    p.total = off;
    return p;
}

// Upper bound of the score of any path from (0,0) to (Li,Lj) that touches a diagonal outside the band [dlo1 + 1, dhi1 - 1]
// (D = Li - Lj lies inside it).  Such a path takes nv >= dhi1 steps down (or nh >= -dlo1 steps right), nh = nv - D, and
// exactly Li - nv diagonal steps of at most maxS each.  With gm = max(0, max g) and cb = max(go, ge) + gm:
//  * a step DOWN never collects the incentive when it extends (jExt = ge + J, pyx:201); opening costs go + g[i-1] -- or
//    ge + g[i-1] where the reference waives the open: the run down column 0 from (0,0) (pyx:170), a step that lands on the
//    last row, the run in the last column (pyx:234-317), at most three such places on a path.  If go + gm <= ge, the nv
//    steps down therefore cost at most ge * nv + 3 gm; otherwise cb each.
//  * a step RIGHT in row i costs ge + g[i] (extension) or go + g[i] (open; ge + g[i] on the last row or when it lands on the
//    last column -- one step, once per path); the run along row 0 costs ge per step + g[0] once (pyx:160).  In a row without
//    incentive every step costs at most ge (go <= ge); a run inside an interior incentive row pays its open, n (ge + gm) +
//    (go - ge) for n steps.  So nh steps right cost at most max(ge * nh, cb * nh + (go - ge)) + 2 gm -- unless the LAST row
//    carries an incentive (`last_pos`) or go > ge: then cb each.
// Every term falls as nv grows, so the bound is taken at the smallest nv.  -> C2_DIAG_NEG if no such path exists.
__device__ __forceinline__ int c2_outside_band_bound(const int maxS, const int Li, const int Lj, const int D, const int dhi1, const int dlo1,
                                                     const int cb, const int go, const int ge, const int last_pos)
{
    if (maxS < 0) return 0x7fffffff;                            // (the bound grows with nv then: no certificate)
    const int gm = cb - (go > ge ? go : ge);                    // max(0, max gap incentive)
    const bool waived = go + gm <= ge;
    const int down = waived ? ge : cb, extra = waived ? 3 * gm : 0;
    const bool runs = go <= ge && !last_pos;
    auto right = [&](const int nh) { return runs ? c2_imax(ge * nh, cb * nh + (go - ge)) + 2 * gm : cb * nh; };
    int U = C2_DIAG_NEG;
    if (dhi1 <= Li) U = c2_imax(U, maxS * (Li - dhi1) + down * dhi1 + extra + right(dhi1 - D));
    if (-dlo1 <= Lj) U = c2_imax(U, maxS * (Lj + dlo1) + down * (D - dlo1) + extra + right(-dlo1));
    return U;
}

struct c2_diag_plane {
    static constexpr bool kWordRuns = false;
    const unsigned* words; int d0;
    __device__ __forceinline__ bool fetch(const int pi, const int pj, unsigned& nib) const {
        const int sl = ((pi - pj - d0) >> 1) - C2_DIAG_STORE_LO;         // stored lane slot of the cell's diagonal
        if ((unsigned)sl >= (unsigned)C2_DIAG_STORE_N) return false;
        const int a = pi + pj;
        nib = (words[(a >> 3) * C2_DIAG_STORE_N + sl] >> (4 * (7 - (a & 7)))) & 0xF;
        return true;
    }
};

struct c2_diag_state {
    int ME, IE, JE, HE;              // latest cell of the even diagonal
    int MO, IO, JO, HO;              // latest cell of the odd diagonal
    unsigned bits;
};                                   // (all values carry C2_DIAG_BIAS; a neighbour outside the band reads as 0)

// One pair of steps: the E cell on anti-diagonal a = 2k, then the O cell on a + 1.  rowE: constants of the E cell's row,
// rowO: of the O cell's row (= E row + 1); cj4: 4 * code of their common column.
// MASK: lanes whose diagonal has not reached its first interior cell yet keep their boundary-cell values.
// LASTCOL: the column may be the last one, where gap_open is replaced by gap_extend (pyx:234-273): a_i -> b_i, c_i += b_i - a_i.
template <bool MASK, bool LASTCOL>
__device__ __forceinline__ void c2_diag_pair(c2_diag_state& S, const int a, const c2_diag_row rowE, const c2_diag_row rowO,
                                             const int cj4, const int ge, const int startE, const int startO, const bool lastcol)
{
    // ---- even step: E cell.  left (i, j-1) is this lane's O cell, up (i-1, j) is the O cell of the lane below (wave_shr).
    //      `ge` is a VGPR here: a DPP instruction cannot take an SGPR as its second source.
    const int upM = c2_shr1z(S.MO);
    const int upJ = c2_shr1z(S.JO);
    if (!MASK || a >= startE) {
        const int corr = (LASTCOL && lastcol) ? rowE.b - rowE.a : 0;
        const int s = c2_sbfe4((int)rowE.prof, cj4);
        const int iFromM = S.MO + rowE.a + corr;
        const int iExt = S.IO + rowE.b;
        const int jFromM = upM + rowE.c + corr;
        const int jExt = upJ + ge;
        const int In = c2_imax(iFromM, iExt);
        const int Jn = c2_imax(jFromM, jExt);
        const int Mn = S.HE + s;                             // H(i-1, j-1): this diagonal, two steps ago
        const int Hn = c2_imax(c2_imax(Mn, Jn), In);
        c2_push4(S.bits, iFromM, iExt, jFromM, jExt, In, Hn, Jn, Mn);
        S.ME = Mn; S.IE = In; S.JE = Jn; S.HE = Hn;
    } else {
        S.bits <<= 4;
    }
    // ---- odd step: O cell.  left (i, j-1) is the E cell of the lane above (wave_shl), up (i-1, j) is this lane's E cell
    const int lfM = c2_shl1z(S.ME);
    const int lfI = c2_shl1z(S.IE);
    if (!MASK || a + 1 >= startO) {
        const int corr = (LASTCOL && lastcol) ? rowO.b - rowO.a : 0;
        const int s = c2_sbfe4((int)rowO.prof, cj4);
        const int iFromM = lfM + rowO.a + corr;
        const int iExt = lfI + rowO.b;
        const int jFromM = S.ME + rowO.c + corr;
        const int jExt = S.JE + ge;
        const int In = c2_imax(iFromM, iExt);
        const int Jn = c2_imax(jFromM, jExt);
        const int Mn = S.HO + s;
        const int Hn = c2_imax(c2_imax(Mn, Jn), In);
        c2_push4(S.bits, iFromM, iExt, jFromM, jExt, In, Hn, Jn, Mn);
        S.MO = Mn; S.IO = In; S.JO = Jn; S.HO = Hn;
    } else {
        S.bits <<= 4;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Packed fill (c2_align_diagp_kernel): TWO alignments per lane, 16 bits each.  A lane group sweeps two reads of the same
// length against the same reference; every DP value is an int16 with C2_PK_BIAS added (0 = "outside the band" = what a DPP read
// with bound_ctrl returns, in both halves), every recurrence one v_pk_*_i16 instruction for both.  The host admits a
// reference to this kernel only if its DP values provably stay inside int16 around the bias (c2_pk_eligible): the reference's
// finite sentinel min_score = gap_open * Li * Lj is replaced by -C2_PK_BIAS (the number 0), which changes no comparison a
// traceback can see: sentinel-derived values keep their order among themselves (same offsets) and stay below every real value.
// Pointer bits: the signs of four packed differences per cell, gathered by two v_perm_b32 and pushed into four byte-wide shift
// registers (c2_pk_push4) -- ~24.7 VALU instructions per cell for the two alignments together, against 2 x 17.2 in the 32-bit kernels.
// ---------------------------------------------------------------------------------------------------------------
#define C2_PK_BIAS 16384
#define C2_PK_LUT_CODES 6                     // reference symbols with codes 0..4 (A C G T N), plus an all-zero table (index 5) for the padding rows
#define C2_PK_PAD_TABLE 5
#define C2_PK_LUT_LDS_OFFSET 1024u            // = character codes (256) + per-slot table (8 x 24 ints) in c2_make_diagx_plan
#if defined(__HIP_DEVICE_COMPILE__)
typedef short c2_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short c2_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned c2_pk_add(const unsigned a, const unsigned b) { return __builtin_bit_cast(unsigned, (c2_s16x2)(__builtin_bit_cast(c2_s16x2, a) + __builtin_bit_cast(c2_s16x2, b))); }
__device__ __forceinline__ unsigned c2_pk_sub(const unsigned a, const unsigned b) { return __builtin_bit_cast(unsigned, (c2_s16x2)(__builtin_bit_cast(c2_s16x2, a) - __builtin_bit_cast(c2_s16x2, b))); }
__device__ __forceinline__ unsigned c2_pk_max(const unsigned a, const unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(c2_s16x2, a), __builtin_bit_cast(c2_s16x2, b))); }
__device__ __forceinline__ unsigned c2_pk_lshr(const unsigned a, const int n) { return __builtin_bit_cast(unsigned, (c2_u16x2)(__builtin_bit_cast(c2_u16x2, a) >> (c2_u16x2)(unsigned short)n)); }
#else
__device__ __forceinline__ unsigned c2_pk_add(const unsigned a, const unsigned b) { return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16); }
__device__ __forceinline__ unsigned c2_pk_sub(const unsigned a, const unsigned b) { return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16); }
__device__ __forceinline__ unsigned c2_pk_max(const unsigned a, const unsigned b) {
    const int al = (int16_t)(a & 0xffffu), bl = (int16_t)(b & 0xffffu), ah = (int16_t)(a >> 16), bh = (int16_t)(b >> 16);
    return ((unsigned)(al > bl ? al : bl) & 0xffffu) | ((unsigned)(ah > bh ? ah : bh) << 16);
}
__device__ __forceinline__ unsigned c2_pk_lshr(const unsigned a, const int n) { return ((a & 0xffffu) >> n) | (((a >> 16) >> n) << 16); }
#endif
__host__ __device__ inline unsigned c2_pk_dup(const int x) { return ((unsigned)x & 0xffffu) | ((unsigned)x << 16); }   // the same int16 in both halves

struct c2_pk_state {
    unsigned ME, IE, JE, HE;         // latest cell of the even diagonal, two alignments packed
    unsigned MO, IO, JO, HO;         // latest cell of the odd diagonal
    unsigned acc;                    // pointer bits of the word in the making: four byte-wide shift registers (see c2_pk_push4), newest cell on top
    unsigned gf;                     // AND of the finished words (gap-free predicate: "H is not I" and "M beats J" of the E cells all set)
};

// The four pointer bits of one cell, for both alignments, from the SIGNS of four packed differences (bits 15 and 31 of each).
// Two v_perm_b32 gather the eight sign-carrying bytes, one shift + one bit-select interleave them, one shift + one bit-select
// push them into `acc` -- 6 instructions per cell instead of 8 (a 16-bit shift + and-or per bit).  `acc` is four byte-wide
// shift registers, two bits per cell, newest cell in bits 7..6, four cells per byte:
//     byte 0: alignment A  { I opened (iFromM > iExt), NOT "H is I" (In < Hn) }      byte 1: A  { J opened, NOT "J beats M" (Jn < Mn) }
//     byte 2: alignment B  { I opened, NOT "H is I" }                                 byte 3: B  { J opened, NOT "J beats M" }
#define C2_PK_HI_BYTES 0x03070105u            // v_perm selector: [hi byte of x.lo16, hi byte of y.lo16, hi byte of x.hi16, hi byte of y.hi16] of (x, y)
__device__ __forceinline__ void c2_pk_push4(unsigned& acc, const unsigned dI, const unsigned dJ, const unsigned dH, const unsigned dM) {
    const unsigned p_open = __builtin_amdgcn_perm(dI, dJ, C2_PK_HI_BYTES);             // sign of dI / dJ in bit 7 of bytes 0,2 / 1,3
    const unsigned p_state = __builtin_amdgcn_perm(dH, dM, C2_PK_HI_BYTES);            // sign of dH / dM likewise
    const unsigned r = (p_open & 0x80808080u) | ((p_state >> 1) & 0x7f7f7f7fu);         // bits 7 and 6 of every byte are the cell's; the rest is noise
    acc = (r & 0xC0C0C0C0u) | ((acc >> 2) & 0x3F3F3F3Fu);                                // (v_bfi_b32 through inline asm was measured: not faster than what hipcc makes of this)
}
// a cell that is not computed: "nothing opened, H is not I, M beats J" (neutral for the gap-free predicate)
__device__ __forceinline__ void c2_pk_push_none(unsigned& acc) { acc = 0x40404040u | ((acc >> 2) & 0x3F3F3F3Fu); }

// One pair of steps (E cell on anti-diagonal a = 2k, O cell on a + 1) for both alignments of the lane.  rowE / rowO: packed row
// constants {a, b, c} (both halves equal: the two reads share the reference); sE / sO: the score pairs of the two cells.
// ROW: the lane group is a DPP row of 16 lanes (row_shr / row_shl hand-off, no lane switched off).
// ADD32: the sums are plain 32-bit adds (v_add_u32, full rate) instead of v_pk_add_i16 (half rate): every operand half is
// non-negative and the sums stay below 2^15 (c2_pk_add32_ok), so no carry crosses the halves.  The differences and maxima stay packed.
template <bool ADD32>
__device__ __forceinline__ unsigned c2_pk_sum(const unsigned a, const unsigned b) { return ADD32 ? a + b : c2_pk_add(a, b); }

// SCORE: no pointer bits are formed -- only the gap-free predicate of the E cells ("H is not I" and "M beats J": the signs of two packed
// differences, ANDed into S.gf bits 15 / 31).  What such a fill can finish is the alignment that IS the main diagonal (c2_align_diags_kernel).
template <bool MASK, bool LASTCOL, bool ROW, bool ADD32, bool SCORE = false>
__device__ __forceinline__ void c2_pk_pair(c2_pk_state& S, const int a, const c2_diag_row rowE, const c2_diag_row rowO, const unsigned sE, const unsigned sO,
                                           const unsigned ge2, const int startE, const int startO, const bool lastcol)
{
    const unsigned upM = (unsigned)(ROW ? c2_rshr1z((int)S.MO) : c2_shr1z((int)S.MO));
    const unsigned upJ = (unsigned)(ROW ? c2_rshr1z((int)S.JO) : c2_shr1z((int)S.JO));
    if (!MASK || a >= startE) {
        const unsigned corr = (LASTCOL && lastcol) ? c2_pk_sub((unsigned)rowE.b, (unsigned)rowE.a) : 0u;
        const unsigned iFromM = c2_pk_sum<ADD32>(c2_pk_sum<ADD32>(S.MO, (unsigned)rowE.a), corr);
        const unsigned iExt = c2_pk_sum<ADD32>(S.IO, (unsigned)rowE.b);
        const unsigned jFromM = c2_pk_sum<ADD32>(c2_pk_sum<ADD32>(upM, (unsigned)rowE.c), corr);
        const unsigned jExt = c2_pk_sum<ADD32>(upJ, ge2);
        const unsigned In = c2_pk_max(iFromM, iExt);
        const unsigned Jn = c2_pk_max(jFromM, jExt);
        const unsigned Mn = c2_pk_sum<ADD32>(S.HE, sE);
        const unsigned Hn = c2_pk_max(c2_pk_max(Mn, Jn), In);
        // I opened (iFromM > iExt), J opened (jFromM > jExt), NOT H is I (In < Hn; In <= Hn always), NOT J beats M (Jn < Mn)
        if (SCORE) S.gf &= c2_pk_sub(In, Hn) & c2_pk_sub(Jn, Mn);
        else c2_pk_push4(S.acc, c2_pk_sub(iExt, iFromM), c2_pk_sub(jExt, jFromM), c2_pk_sub(In, Hn), c2_pk_sub(Jn, Mn));
        S.ME = Mn; S.IE = In; S.JE = Jn; S.HE = Hn;
    } else if (!SCORE) {
        c2_pk_push_none(S.acc);
    }
    const unsigned lfM = (unsigned)(ROW ? c2_rshl1z((int)S.ME) : c2_shl1z((int)S.ME));
    const unsigned lfI = (unsigned)(ROW ? c2_rshl1z((int)S.IE) : c2_shl1z((int)S.IE));
    if (!MASK || a + 1 >= startO) {
        const unsigned corr = (LASTCOL && lastcol) ? c2_pk_sub((unsigned)rowO.b, (unsigned)rowO.a) : 0u;
        const unsigned iFromM = c2_pk_sum<ADD32>(c2_pk_sum<ADD32>(lfM, (unsigned)rowO.a), corr);
        const unsigned iExt = c2_pk_sum<ADD32>(lfI, (unsigned)rowO.b);
        const unsigned jFromM = c2_pk_sum<ADD32>(c2_pk_sum<ADD32>(S.ME, (unsigned)rowO.c), corr);
        const unsigned jExt = c2_pk_sum<ADD32>(S.JE, ge2);
        const unsigned In = c2_pk_max(iFromM, iExt);
        const unsigned Jn = c2_pk_max(jFromM, jExt);
        const unsigned Mn = c2_pk_sum<ADD32>(S.HO, sO);
        const unsigned Hn = c2_pk_max(c2_pk_max(Mn, Jn), In);
        if (!SCORE) c2_pk_push4(S.acc, c2_pk_sub(iExt, iFromM), c2_pk_sub(jExt, jFromM), c2_pk_sub(In, Hn), c2_pk_sub(Jn, Mn));
        S.MO = Mn; S.IO = In; S.JO = Jn; S.HO = Hn;
    } else if (!SCORE) {
        c2_pk_push_none(S.acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-alignment diagonal-band kernel: NA (2 or 4) alignments share one wavefront.  One anti-diagonal step costs the
// same ~19 VALU issues whether 64, 31 or 15 of the lanes hold diagonals that matter, and an amplicon read rarely needs
// more than a few diagonals either side of the corner-to-corner one -- so the wavefront is cut into NA lane groups of
// LPA = 64 / NA lanes, each sweeping its own alignment with a band of 2 * (LPA - 1) diagonals, all with the same
// instruction stream (per-lane table bases and clamps instead of wave-uniform ones).  The last lane of every group is
// switched off in EXEC for the whole fill: a DPP read (bound_ctrl set) whose source lane is disabled returns 0, so the
// first lane of the next group (wave_shr) and the last live lane of this group (wave_shl) see "outside the band", exactly
// what the lanes at the two ends of the wavefront see.  Isolation costs no instruction.
// A band this narrow fails the optimality certificate more often; those tasks go to the fallback list and the host
// chains the launches NA = 4 -> NA = 2 -> c2_align_diag_kernel (128 diagonals) -> row-strip kernel (any path), each over
// the previous list.  The pointer words go to a per-workgroup scratch plane in HBM/L2 instead of LDS (16 KB per group of
// alignments would halve the resident waves; the words are written once, coalesced, and the traceback reads a handful
// of them), read back with agent-scope loads that bypass the CU's L1.
// ---------------------------------------------------------------------------------------------------------------
#define C2_RUNS_MAX 16                              // runs of one alignment the lane-group epilogue keeps (more: the next launch takes the task)
#define C2_GRP_SLOT_WORDS (C2_RUNS_MAX * 3 + 12)    // LDS words per alignment: run table (3 words per run) + 8 accumulator words + the differing columns' list (count, 3 entries)
struct c2_diagx_plan {
    uint32_t codeof, table, tmp_read, tmp_ref, stage, slot0, slot_bytes, total, n_words;
    uint32_t pairlut, pcodes0, pcodes_bytes, group0, group_bytes, gref, gincp;   // packed kernels only
    uint32_t codes, read, code, ref, incp, win;                     // offsets inside one alignment's slot
    uint32_t slot_read_bytes;                                       // bytes of a slot's read buffer
};

// score_only (c2_align_diags_kernel): nothing is traced, so the staging area of an alignment's pointer words is not part of the plan
__host__ __device__ inline c2_diagx_plan c2_make_diagx_plan(int na, int max_li, int max_lj, bool pk = false, bool score_only = false) {
    c2_diagx_plan p;
    const uint32_t lpa = ((64u / (uint32_t)(pk ? na / 2 : na)) + 3u) & ~3u;   // words per group of 8 anti-diagonals in a slot's plane: the lanes of one lane group (pk: two
                                                                    // alignments share a group, 16 bits each), rounded up to four (three groups of 21 lanes: 24) -- the slice is staged as 16-byte words
    p.n_words = (uint32_t)(max_li + max_lj) / 8u + 1u;              // per lane: one 32-bit word per 8 anti-diagonals
    p.slot_read_bytes = c2_align16((uint32_t)max_lj);
    uint32_t off = 0;
    p.codeof = off;   off += 256u;                                  // character -> code
    p.table = off;    off += 8u * 24u * 4u;                         // up to 8 slots x C2X_INTS (more slots: the table moves behind everything else, below)
    if (pk) off += c2_align16((uint32_t)C2_PK_LUT_CODES * C2_PK_LUT_STRIDE);                // pair-score tables at the FIXED offset C2_PK_LUT_LDS_OFFSET: it folds into the look-ups' immediate offset
    p.tmp_read = off; off += c2_align16((uint32_t)max_li + (uint32_t)max_lj);   // aligned strings of the alignment being traced
    p.tmp_ref = off;  off += c2_align16((uint32_t)max_li + (uint32_t)max_lj);
    p.stage = off;    off += (uint32_t)na * C2_GRP_SLOT_WORDS * 4u;  // per alignment: its run table + the epilogue's accumulators (c2_group_epilogue)
    p.pairlut = off;  p.pcodes0 = off; p.pcodes_bytes = 0; p.group0 = off; p.group_bytes = 0; p.gref = 0; p.gincp = 0;
    if (pk) {
        // LDS is what limits the resident waves of this kernel (8 alignments per wavefront), so its layout is lean: the lane
        // groups' column tables share the bytes of the two traceback strings (the fill is over when a traceback starts; the
        // staging rewrites the tables, zeros in front included, every time); reference + window prefix once per lane group (its
        // two alignments share the reference); per alignment only the read
        p.pcodes_bytes = c2_align16((uint32_t)C2_DIAG_CODE_PAD + (uint32_t)max_lj + 2u + 16u);   // per lane group: (code A << 5 | code B << 2) per column
        p.pcodes0 = p.tmp_read;
        const uint32_t need = (uint32_t)(na / 2) * p.pcodes_bytes, have = 2u * c2_align16((uint32_t)max_li + (uint32_t)max_lj);
        if (need > have) { off += need - have; }                    // (tmp_read, tmp_ref, stage are consecutive: the tables may run into `stage`, which is rewritten before use too)
        p.stage = p.tmp_ref + c2_align16((uint32_t)max_li + (uint32_t)max_lj) + (need > have ? need - have : 0u);
        off = p.stage + (score_only ? 0u : (uint32_t)na * C2_GRP_SLOT_WORDS * 4u);
        p.pairlut = C2_PK_LUT_LDS_OFFSET;                           // per reference symbol: the score pair of every (symbol of read A, symbol of read B)
        p.group0 = off;
        p.gref = 0; p.gincp = c2_align16((uint32_t)max_li);
        p.group_bytes = p.gincp + c2_align16(((uint32_t)max_li + 2u) * 2u);
        off += (uint32_t)(na / 2) * p.group_bytes;
        p.slot0 = off;
        p.codes = 0; p.read = 0; p.code = 0; p.ref = 0; p.incp = 0; p.win = 0;
        p.slot_bytes = c2_align16((uint32_t)max_lj);
        p.total = p.slot0 + (uint32_t)na * p.slot_bytes;
        if (na > 8) { p.table = p.total; p.total += (uint32_t)na * 24u * 4u; }     // (the pair-score tables keep their fixed offset)
        return p;
    }
    p.slot0 = off;
    uint32_t so = 0;
    p.codes = so;    so += c2_align16((uint32_t)C2_DIAG_CODE_PAD + (uint32_t)max_lj + 2u + 16u);  // zeros | columns 0 .. Lj+1 | zeros (the staging writes them as dwords: up to 15 behind column Lj)
    p.read = so;     so += c2_align16((uint32_t)max_lj);
    p.code = so;     so += c2_align16((uint32_t)max_lj);
    p.ref = so;      so += c2_align16((uint32_t)max_li);
    p.incp = so;     so += c2_align16(((uint32_t)max_li + 2u) * 2u);
    p.win = so;      so += c2_align16((uint32_t)max_li + 4u);            // one byte per reference position, read as dwords: 0x80 = inside the quantification window
    p.slot_bytes = so;
    p.total = p.slot0 + (uint32_t)na * so;
    return p;
}

// pointer words of ONE alignment, staged in LDS: [group of 8 anti-diagonals][lane of the alignment's lane group]
struct c2_diagx_plane {
    static constexpr bool kWordRuns = true;                         // (c2_traceback: runs of state M are read off whole pointer words where pk is set)
    const unsigned* words; int d0, lpa, nl; bool pk;               // nl: lanes of a group that hold diagonals (lpa - 1, or lpa for a row-DPP group)
    __device__ __forceinline__ bool fetch(const int pi, const int pj, unsigned& nib) const {
        const int sl = (pi - pj - d0) >> 1;                              // lane of the cell's diagonal inside its group
        if ((unsigned)sl >= (unsigned)nl) return false;
        const int a = pi + pj;
        const unsigned w = words[(a >> 3) * lpa + sl];
        if (!pk) { nib = (w >> (4 * (7 - (a & 7)))) & 0xF; return true; }
        // packed kernels (c2_pk_push4): the word's low half holds anti-diagonals 8g .. 8g+3, its high half 8g+4 .. 8g+7; in a half,
        // byte 0 = { I opened, NOT "H is I" } and byte 1 = { J opened, NOT "J beats M" }, cell c in bits 2c+1 .. 2c of both
        const int c = a & 7;
        const unsigned h = w >> (16 * (c >> 2));
        const unsigned ih = (h >> (2 * (c & 3))) & 3u, jm = (h >> (8 + 2 * (c & 3))) & 3u;
        nib = ((ih >> 1) << 3) | ((jm >> 1) << 2) | (((ih & 1u) ^ 1u) << 1) | ((jm & 1u) ^ 1u);
        return true;
    }
};

// per-lane view of the tables and of the matrix edges.  The row table (global memory) and the column-symbol table (LDS) are
// padded with zeros, so a lane that is before / past the matrix needs no clamp at the low end and one v_min at the high end
// per group; the records of a group are consecutive, so one address serves all of its loads (immediate offsets).
struct c2_diagx_lane {
    unsigned rowOff, rowMax;         // byte offset from A.diag_base of row (hE + 0) of pair 0's E cell; largest offset a group may start at
    unsigned colOff, colMax;         // LDS byte address of column (0 - hE); largest address a group may start at
    int kLast;                       // pair whose cells are on the last column
    int kCap; bool capOdd;           // pair (and cell of it) that holds H(Li, Lj), if this lane owns that diagonal
    int startE, startO;              // first interior anti-diagonal of the two diagonals
};

// rows and column symbols of group g (pairs 4g .. 4g+3): row records 0..4 (the O cell of the last pair needs row + 1) and four symbols.
// FIRST = false: record 0 is the caller's business -- it is record 4 of group g - 1 (also when the v_min below clamps either
// group's address: every record from the clamp on is zero padding), so a group costs four 16-byte loads, not five.
template <bool FIRST>
__device__ __forceinline__ void c2_diagx_fetch(const int g, const c2_diagx_lane& L, const c2_diag_row* rows, const unsigned char* lds,
                                               c2_diag_row (&R)[5], int (&C)[4])
{
    const unsigned ro = min(L.rowOff + (unsigned)(g * 4 * (int)sizeof(c2_diag_row)), L.rowMax);
    const c2_diag_row* rp = (const c2_diag_row*)((const unsigned char*)rows + ro);
#pragma unroll
    for (int q = FIRST ? 0 : 1; q < 5; ++q) R[q] = rp[q];
    const unsigned co = min(L.colOff + (unsigned)(g * 4), L.colMax);
#pragma unroll
    for (int q = 0; q < 4; ++q) C[q] = (int)lds[co + q];
}

// One group = four pairs = eight anti-diagonals = one pointer word per lane; the next group's tables are requested first.
// c2_gapfree: the "is the alignment gap-free" predicate of c2_try_gapless, kept in registers while the pointer bits are made.
// acc collects the two low pointer bits ("H is I", "J beats M") of every E cell a lane has finished -- for the lane that owns
// diagonal 0 of a square alignment those are the main-diagonal cells (i, i) -- one v_and_or per group of eight anti-
// diagonals; cap is acc at the moment the lane passes the cell (Li, Lj) (the cells a lane computes beyond that are zero-
// padding garbage, like H).  cap == 0 in that lane <=> the traceback never leaves state M: no pointer word has to be read back.
struct c2_gapfree { unsigned acc, cap; };
#define C2_GAPFREE_E_LOW2 0x30303030u      // E cells sit in the odd nibbles of a word (anti-diagonals 8g, 8g+2, ...): their bits 1..0

template <bool MASK, bool LASTCOL>
__device__ __forceinline__ void c2_diagx_group(c2_diag_state& S, const int g, const c2_diagx_lane& L, const int ge, int& Hcap, c2_gapfree& GF,
                                               const c2_diag_row (&R)[5], const int (&C)[4], c2_diag_row (&RN)[5], int (&CN)[4],
                                               const c2_diag_row* rows, const unsigned char* lds, unsigned* myWords, const int wordStride,
                                               const bool stores = true)
{
    RN[0] = R[4];
    c2_diagx_fetch<false>(g + 1, L, rows, lds, RN, CN);
    GF.acc |= S.bits & C2_GAPFREE_E_LOW2;                            // the previous group's word (0 before the first group)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = 4 * g + q;
        c2_diag_pair<MASK, LASTCOL>(S, 2 * k, R[q], R[q + 1], C[q], ge, L.startE, L.startO, LASTCOL && (k == L.kLast));
        if (LASTCOL && k == L.kCap) {
            Hcap = (L.capOdd ? S.HO : S.HE) - C2_DIAG_BIAS;
            // nibbles of this group so far: the low 8 (q + 1) bits of the word in the making (the O cell of this pair lies beyond the matrix)
            GF.cap = GF.acc | (S.bits & (C2_GAPFREE_E_LOW2 & (q == 3 ? 0xffffffffu : ((1u << (8 * (q + 1))) - 1u))));
        }
    }
    if (stores) myWords[g * wordStride] = S.bits;                    // anti-diagonals 8g .. 8g+7
}

// Groups g .. g_stop; the tables alternate between two register sets (no copies).  `cur` tells which set holds group g's.
template <bool MASK, bool LASTCOL>
__device__ __forceinline__ void c2_diagx_groups(c2_diag_state& S, int& g, const int g_stop, const c2_diagx_lane& L, const int ge,
                                                int& Hcap, c2_gapfree& GF, c2_diag_row (&RA)[5], int (&CA)[4], c2_diag_row (&RB)[5], int (&CB)[4],
                                                const c2_diag_row* rows, const unsigned char* lds, unsigned* myWords, const int wordStride,
                                                const bool stores = true)
{
    for (; g + 1 <= g_stop; g += 2) {
        c2_diagx_group<MASK, LASTCOL>(S, g, L, ge, Hcap, GF, RA, CA, RB, CB, rows, lds, myWords, wordStride, stores);
        c2_diagx_group<MASK, LASTCOL>(S, g + 1, L, ge, Hcap, GF, RB, CB, RA, CA, rows, lds, myWords, wordStride, stores);
    }
    if (g <= g_stop) {                                               // odd count: one more group, then move its successor's tables to set A
        c2_diagx_group<MASK, LASTCOL>(S, g, L, ge, Hcap, GF, RA, CA, RB, CB, rows, lds, myWords, wordStride, stores);
        ++g;
#pragma unroll
        for (int q = 0; q < 5; ++q) RA[q] = RB[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) CA[q] = CB[q];
    }
}

__global__ __launch_bounds__(64, 2) void c2_align_diag_kernel(c2_align_args A)
{
    const int lane = threadIdx.x;
    const c2_diag_plan P = c2_make_diag_plan(A.max_li, A.max_lj);
    unsigned* sWords = (unsigned*)(c2_smem + P.plane);
    unsigned char* sCodes = c2_smem + P.codes;
    c2_wg W;
    W.sRead = c2_smem + P.read; W.sCode = c2_smem + P.code; W.sRef = c2_smem + P.ref;
    W.sIncP = (uint16_t*)(c2_smem + P.incp); W.sTmpRead = c2_smem + P.tmp_read; W.sTmpRef = c2_smem + P.tmp_ref;
    const int ge = A.gap_extend, go = A.gap_open;
    unsigned char* sCodeOf = c2_smem + P.codeof;
    for (int k = lane; k < 256; k += 64) sCodeOf[k] = A.code_of_char[k];

    int cur_ref = -1;
    int Li = 0, g0 = 0, ref_bad = 0;
    uint64_t chunk_base = 0;
    int chunk_left = 0;
    c2_phase_acc PH; PH.t_last = 0; PH.sum[0] = PH.sum[1] = PH.sum[2] = PH.sum[3] = 0;
    c2_prefetch pf;
    c2_prefetch_issue(A, lane, chunk_base, chunk_left, pf);
    while (pf.valid) {
        __syncthreads();
        c2_phase_begin(A.phase_cycles, PH);
        const uint64_t task = pf.task;
        const int Lj = pf.Lj, ref_id = pf.ref_id, rc = pf.rc;
        bool packed;
        int status = c2_commit_task(A, W, sCodeOf, pf, lane, A.max_li, cur_ref, Li, g0, ref_bad, packed);
        c2_prefetch_issue(A, lane, chunk_base, chunk_left, pf);   // next task's loads fly during this task's DP
        const c2_dev_ref rf = A.refs[ref_id];
        __syncthreads();
        c2_aln_record rec;
        c2_clear_record(rec, rc, ref_id);
        bool need_full = false;
        const int D = Li - Lj;
        const int d0 = ((D >> 1) - 64) & ~1;                  // even; band = d0 .. d0+127 around the corner-to-corner diagonal
        int cb = 0;
        if (status == 0) {
            cb = (go > ge ? go : ge) + rf.gap_incentive_max;      // the most one gap base can add to a score
            if (!packed || rf.diag_rows == nullptr || cb >= 0 || d0 > 0 || d0 + 127 < 0 || D < d0 || D > d0 + 127) need_full = true;
        }
        if (status == 0 && !need_full) {
            // ---- tables: row constants (per reference) and 4*code per column (per read), both padded so that the lanes that
            //      are still before / already past the matrix read zeros instead of running off the arrays
            // (same zero-padded tables and alternating register sets as the multi-alignment kernel)
            for (int j = lane; j < C2_DIAG_CODE_PAD + Lj + 2 + 8; j += 64) {
                const int col = j - C2_DIAG_CODE_PAD;
                sCodes[j] = (col >= 1 && col <= Lj) ? (unsigned char)(W.sCode[col - 1] << 2) : (unsigned char)0;
            }
            __syncthreads();
            const int min_score = (int)(uint32_t)((uint64_t)(int64_t)go * (uint64_t)Lj * (uint64_t)Li);
            c2_phase_mark<0>(A.phase_cycles, PH);

            // ---- per-lane diagonals and their boundary cells (pyx:153-176)
            const int hE = (d0 >> 1) + lane;                  // dE = 2*hE, dO = 2*hE + 1
            const int dE = 2 * hE, dO = dE + 1;
            c2_diag_state S;
            S.bits = 0;
            // diagonal d >= 1 starts at cell (d, 0): M = I = min_score, J = ge*d + g0;  d <= -1 at (0, -d): M = J = min_score,
            // I = ge*(-d) + g0;  d == 0 at (0, 0): M = 0, I = J = min_score.  H = max of the three.  (+ C2_DIAG_BIAS)
            {
                const int ms = min_score + C2_DIAG_BIAS;
                const int bE = ((dE == 0) ? 0 : ge * (dE > 0 ? dE : -dE) + g0) + C2_DIAG_BIAS;
                S.ME = (dE == 0) ? C2_DIAG_BIAS : ms;
                S.IE = (dE < 0) ? bE : ms;
                S.JE = (dE > 0) ? bE : ms;
                S.HE = c2_imax(c2_imax(S.ME, S.IE), S.JE);
                const int bO = ge * (dO > 0 ? dO : -dO) + g0 + C2_DIAG_BIAS;   // dO is odd, never 0
                S.MO = ms;
                S.IO = (dO < 0) ? bO : ms;
                S.JO = (dO > 0) ? bO : ms;
                S.HO = c2_imax(c2_imax(S.MO, S.IO), S.JO);
            }
            const int startE = (dE > 0 ? dE : -dE) + 2, startO = (dO > 0 ? dO : -dO) + 2;   // first interior anti-diagonal
            const int a_end = Li + Lj;
            const int k_end = a_end >> 1;                      // pair that holds the cell (Li, Lj)
            const int max_start = (d0 + 127 > -d0 ? d0 + 127 : -d0) + 2;
            const int gA = ((max_start + 1) >> 1) >> 2;        // groups 0..gA contain lanes that have not started
            const int gC = ((2 * Lj + d0) >> 1) >> 2;          // first group in which some lane is on the last column
            const int g_end = k_end >> 2;
            unsigned* myWords = sWords + (lane - C2_DIAG_STORE_LO);
            const bool stores = (unsigned)(lane - C2_DIAG_STORE_LO) < (unsigned)C2_DIAG_STORE_N;
            c2_diagx_lane L;
            const int vrow = (int)(rf.diag_rows - A.diag_base), vcode = (int)P.codes + C2_DIAG_CODE_PAD;
            L.rowOff = (unsigned)((vrow + hE) * (int)sizeof(c2_diag_row));
            L.rowMax = (unsigned)((vrow + Li + 1 + C2_DIAG_ROW_PAD - 5) * (int)sizeof(c2_diag_row));
            L.colOff = (unsigned)(vcode - hE);
            L.colMax = (unsigned)(vcode + Lj + 2);
            L.kLast = Lj + hE;
            L.kCap = k_end; L.capOdd = (a_end & 1) != 0;
            L.startE = startE; L.startO = startO;
            const c2_diag_row* rows = A.diag_base;
            c2_diag_row RA[5], RB[5];
            int CA[4], CB[4];
            c2_diagx_fetch<true>(0, L, rows, c2_smem, RA, CA);
            int Hcap = C2_DIAG_NEG;
            c2_gapfree GF; GF.acc = 0; GF.cap = 0xffffffffu;       // (this kernel reads its LDS plane instead: c2_try_gapless)
            int g = 0;
            const int gA_stop = gA < g_end ? gA : g_end;
            int geV = ge;                                          // gap_extend in a VGPR (second source of a DPP add)
            C2_KEEP_IN_VGPR(geV);
            if (gC <= gA_stop) {
                c2_diagx_groups<true, true>(S, g, gA_stop, L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, C2_DIAG_STORE_N, stores);
            } else {
                c2_diagx_groups<true, false>(S, g, gA_stop, L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, C2_DIAG_STORE_N, stores);
                c2_diagx_groups<false, false>(S, g, (gC - 1 < g_end ? gC - 1 : g_end), L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, C2_DIAG_STORE_N, stores);
            }
            c2_diagx_groups<false, true>(S, g, g_end, L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, C2_DIAG_STORE_N, stores);
            __syncthreads();
            c2_phase_mark<1>(A.phase_cycles, PH);

            // ---- optimality certificate
            const int lane_end = (D - d0) >> 1;
            const int Hend = __builtin_amdgcn_readlane(Hcap, lane_end);
            const int maxS = A.max_score;
            const int dhi1 = d0 + 128, dlo1 = d0 - 1;         // first diagonals outside the band
            const int U = c2_outside_band_bound(maxS, Li, Lj, D, dhi1, dlo1, cb, go, ge, rf.gap_incentive_last_pos);
            if (!(Hend > U)) need_full = true;

            if (!need_full) {
                c2_diag_plane plane;
                plane.words = sWords; plane.d0 = d0;
                if (!(Li == Lj && c2_try_gapless(plane, A, W, task, Li, lane, rec))) {
                    int cnt, matches;
                    bool nf2;
                    c2_traceback(plane, W, Li, Lj, min_score, ge, g0, lane, cnt, matches, status, nf2);
                    __syncthreads();
                    c2_phase_mark<2>(A.phase_cycles, PH);
                    if (nf2) need_full = true;                     // cannot happen when the certificate holds; kept as a guard
                    else if (status == 0) c2_emit_and_classify(A, W, task, cnt, matches, lane, rec, Li, Lj);
                }
            }
        }
        if (need_full) {
            status |= C2_STATUS_NEED_FULL;
            if (lane == 0) { const unsigned q = atomicAdd(A.fb_count, 1u); A.fb_list[q] = (uint32_t)task; }
        }
        rec.status = (uint8_t)status;
        if (lane == 0) A.records[task] = rec;
        c2_phase_mark<3>(A.phase_cycles, PH);
    }
    c2_phase_flush(A.phase_cycles, PH, lane);
}


// per-alignment ("slot") table in LDS: wave-uniform values written by lane 0 and read back through readfirstlane, so the
// staging / traceback / output code exists once (a loop over the slots) instead of once per slot
enum { C2X_VALID = 0, C2X_TASK_LO, C2X_TASK_HI, C2X_LJ, C2X_REF, C2X_RC, C2X_STATUS, C2X_PACKED, C2X_CURREF, C2X_LI, C2X_G0,
       C2X_OK, C2X_D, C2X_D0, C2X_CB, C2X_MINSC, C2X_ROWBASE, C2X_BAND_LI, C2X_BAND_LJ, C2X_LASTPOS, C2X_REFBAD, C2X_UNPAIRED, C2X_INTS = 24 };
__device__ __forceinline__ int c2_uni(const int* p) { return __builtin_amdgcn_readfirstlane(*p); }
// the whole table of one slot with ONE LDS read (lane k gets entry k); C2_TF picks an entry: a v_readlane instead of an
// LDS round trip per entry
__device__ __forceinline__ int c2_tab_load(const int* T, const int lane) { return T[lane < C2X_INTS ? lane : 0]; }
#define C2_TF(v, k) __builtin_amdgcn_readlane((v), (k))

// ---- packed fill: one group = four pairs = eight anti-diagonals = one pointer word per lane AND PER ALIGNMENT.
// Rows come from the packed row table (A.diagpk_base: {a, b, c} duplicated into both halves, and the LDS offset of the reference
// symbol's pair-score table), columns from the lane group's pair-symbol table; the eight score pairs of the group are LDS
// look-ups (row table offset + pair symbol), requested at the top of the group.
struct c2_pk_cap { unsigned H, gf; };                              // the two alignments' H(Li, Lj) and gap-free words, captured at the cell (Li, Lj)

template <bool MASK, bool LASTCOL, bool ROW, bool ADD32, bool SCORE = false>
__device__ __forceinline__ void c2_pk_group(c2_pk_state& S, const int g, const c2_diagx_lane& L, const unsigned ge2, c2_pk_cap& CAP,
                                            const c2_diag_row (&R)[5], const int (&C)[4], c2_diag_row (&RN)[5], int (&CN)[4],
                                            const c2_diag_row* rows, const unsigned char* lds, const unsigned lutBase,
                                            unsigned* wordsA, unsigned* wordsB, const int wordStride)
{
    RN[0] = R[4];
    c2_diagx_fetch<false>(g + 1, L, rows, lds, RN, CN);
    unsigned sc[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[2 * q] = *(const unsigned*)(lds + C2_PK_LUT_LDS_OFFSET + R[q].prof + (unsigned)C[q]);
        sc[2 * q + 1] = *(const unsigned*)(lds + C2_PK_LUT_LDS_OFFSET + R[q + 1].prof + (unsigned)C[q]);
    }
    unsigned w0 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = 4 * g + q;
        c2_pk_pair<MASK, LASTCOL, ROW, ADD32, SCORE>(S, 2 * k, R[q], R[q + 1], sc[2 * q], sc[2 * q + 1], ge2, L.startE, L.startO, LASTCOL && (k == L.kLast));
        if (!SCORE && q == 1) { w0 = S.acc; S.gf &= w0; }          // anti-diagonals 8g .. 8g+3 of both alignments
        if (!SCORE && q == 3) S.gf &= S.acc;                        // ... 8g+4 .. 8g+7
        if (LASTCOL && k == L.kCap) {
            CAP.H = L.capOdd ? S.HO : S.HE;
            // q even: the word in the making holds two cells so far, the E cell in bits 5..4 of every byte ("H is not I" / "M beats J" = bit 4)
            // (SCORE: S.gf is the predicate itself, the cell (Li, Lj) included)
            CAP.gf = SCORE ? S.gf : ((q & 1) ? S.gf : (S.gf & (S.acc | 0xEFEFEFEFu)));
        }
    }
    if (SCORE) return;                                              // (no pointer word: nothing is traced from this fill)
    // alignment A's word: bytes 0, 1 of the two accumulators, alignment B's: bytes 2, 3 (layout: c2_diagx_plane::fetch)
    wordsA[g * wordStride] = __builtin_amdgcn_perm(S.acc, w0, 0x05040100u);
    wordsB[g * wordStride] = __builtin_amdgcn_perm(S.acc, w0, 0x07060302u);
}

template <bool MASK, bool LASTCOL, bool ROW, bool ADD32, bool SCORE = false>
__device__ __forceinline__ void c2_pk_groups(c2_pk_state& S, int& g, const int g_stop, const c2_diagx_lane& L, const unsigned ge2, c2_pk_cap& CAP,
                                             c2_diag_row (&RA)[5], int (&CA)[4], c2_diag_row (&RB)[5], int (&CB)[4],
                                             const c2_diag_row* rows, const unsigned char* lds, const unsigned lutBase,
                                             unsigned* wordsA, unsigned* wordsB, const int wordStride)
{
    for (; g + 1 <= g_stop; g += 2) {
        c2_pk_group<MASK, LASTCOL, ROW, ADD32, SCORE>(S, g, L, ge2, CAP, RA, CA, RB, CB, rows, lds, lutBase, wordsA, wordsB, wordStride);
        c2_pk_group<MASK, LASTCOL, ROW, ADD32, SCORE>(S, g + 1, L, ge2, CAP, RB, CB, RA, CA, rows, lds, lutBase, wordsA, wordsB, wordStride);
    }
    if (g <= g_stop) {
        c2_pk_group<MASK, LASTCOL, ROW, ADD32, SCORE>(S, g, L, ge2, CAP, RA, CA, RB, CB, rows, lds, lutBase, wordsA, wordsB, wordStride);
        ++g;
#pragma unroll
        for (int q = 0; q < 5; ++q) RA[q] = RB[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) CA[q] = CB[q];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The traced epilogue of the multi-alignment kernels, lane groups in parallel (round 6).
// Until round 5 the alignments of an iteration were traced and written out ONE AFTER THE OTHER, the whole wavefront on each: wave-uniform state
// in SGPRs, a few hundred scalar instructions and as many vector ones per alignment whatever its shape (measured on the first band tier:
// 281 VALU + 478 SALU for the walk, 288 + 274 for the strings and the classification, of ~2,300 + ~960 per task).  Here EL = 8 / 16 / 32 lanes
// take one alignment each (NA = 8 / 4 / 2 alignments per wavefront), all traced alignments of the iteration at once, state per lane:
//   * walk (pyx:338-421): a run of state M is read off the pointer words of its diagonal -- 32 / EL words per lane and step, 128 cells per step
//     (the leaving cell found with the word masks of c2_traceback); a gap run 8 / 16 / 32 cells per step; the words come straight from the scratch
//     plane (L2), no staging copy.  The walk leaves RUNS -- (state, length, i, j, columns before it) -- in a small LDS table per alignment.
//   * indel events (COREResources.pyx:119-162; legacy pyx:253-261, 284) in closed form from the runs, a lane per run.
//   * strings FORWARDS, a dword of four columns per lane and step: a dword inside one run is two unaligned LDS dwords (ds_read2 + v_alignbyte);
//     the few dwords that straddle a run boundary are put aside and done byte by byte afterwards; substitutions and matches by the zero-byte
//     trick on read ^ reference over the M columns (pyx:113-118, CRISPResso2Align.pyx:375-376); exactly T bytes of a row are written.
//   * sums through LDS atomics into the alignment's accumulator words; lane s writes the record of slot s.
// More than C2_RUNS_MAX runs, a pointer word outside the band, a literal '-' in a column that takes its character from a sequence: the task goes
// to the next launch (the last launch of every chain, c2_align_classify_kernel, walks column by column and takes anything).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned c2_word_m4(const unsigned w, const int par, const bool pk) {
    // the four cells of one diagonal in a pointer word (anti-diagonals 6+par, 4+par, 2+par, par of the word's eight; topmost first in bit 3):
    // bit set = the path stays in state M behind the cell
    if (pk) {
        const unsigned u = ((w & (w >> 8)) >> (2 * par)) & 0x00110011u;      // NOT "H is I" and NOT "J beats M" (c2_pk_push4): cell 6+par -> bit 20, 4+par -> 16, 2+par -> 4, par -> 0
        return (((u >> 20) & 1u) << 3) | (((u >> 16) & 1u) << 2) | (((u >> 4) & 1u) << 1) | (u & 1u);
    }
    // 32-bit kernels: nibble of anti-diagonal c at bits 4 * (7 - c); its two low bits ("H is I", "J beats M") clear = stays in M
    const unsigned z = ~(w | (w >> 1)) & 0x11111111u;                       // bit 4 * (7 - c) set iff both are clear
    const unsigned zz = z >> (4 * (1 - par));                                // cells 6+par, 4+par, 2+par, par -> bits 0, 8, 16, 24
    return ((zz & 1u) << 3) | (((zz >> 8) & 1u) << 2) | (((zz >> 16) & 1u) << 1) | ((zz >> 24) & 1u);
}
__device__ __forceinline__ unsigned c2_word_nib(const unsigned w, const int a, const bool pk) {
    // the pointer nibble of anti-diagonal a in its lane's word, as c2_diagx_plane::fetch decodes it
    if (!pk) return (w >> (4 * (7 - (a & 7)))) & 0xFu;
    const int c = a & 7;
    const unsigned h = w >> (16 * (c >> 2));
    const unsigned ih = (h >> (2 * (c & 3))) & 3u, jm = (h >> (8 + 2 * (c & 3))) & 3u;
    return ((ih >> 1) << 3) | ((jm >> 1) << 2) | (((ih & 1u) ^ 1u) << 1) | ((jm & 1u) ^ 1u);
}

// a differing column joins the alignment's hint (list[0]: how many so far, list[1 .. 3]: reference index | read base code << 9).  The hint names a read base
// by (ch >> 1) & 7, which tells A C G T N apart and nothing else: any other character (an IUPAC code) spoils the count, and with it the hint
__device__ __forceinline__ void c2_hint_note(int* list, const int idx, const unsigned ch) {
    const unsigned long long chars = (unsigned long long)'A' | ((unsigned long long)'C' << 8) | ((unsigned long long)'T' << 16) | ((unsigned long long)'G' << 24) | ((unsigned long long)'N' << 56);
    const unsigned code = (ch >> 1) & 7u;
    if (((unsigned)(chars >> (8 * code)) & 0xffu) != ch) { atomicAdd(list, 64); return; }
    const int k = atomicAdd(list, 1);
    if (k < 3) list[1 + k] = idx | (int)(code << 9);
}

template <int NA, bool PK, int LPW, int NL>
__device__ __forceinline__ void c2_group_epilogue(const c2_align_args& A, const c2_diagx_plan& P, const int lane, const unsigned m_trace_all, const int sbase,
                                                  int* sTab, const unsigned* gWords, const int slotWords, const bool rows_aligned)
{
    // (sbase: the first slot of this call -- sixteen alignments per wavefront, the opt-in 14-diagonal launch, take two calls of eight)
    constexpr int NAH = NA > 8 ? 8 : NA;                            // alignments per call
    constexpr int EL = NAH > 4 ? 8 : (NAH > 2 ? 16 : 32);           // lanes per alignment
    constexpr int NW = 32 / EL;                                     // pointer words per lane and step of a run of state M
    const unsigned m_trace = m_trace_all >> sbase;
    const int e = lane / EL, q = lane - e * EL;
    const int es = sbase + (e < NAH ? e : 0);
    const bool act = e < NAH && ((m_trace >> e) & 1u);
    const unsigned long long gmask = (EL == 32 ? 0xffffffffull : ((1ull << EL) - 1ull));
    auto sub = [&](const unsigned long long b) { return (unsigned)((b >> (e * EL)) & gmask); };     // this lane group's bits of a ballot
    const int ge = A.gap_extend;
    const int* T = sTab + es * C2X_INTS;
    const int Li = T[C2X_LI], Lj = T[C2X_LJ], d0 = T[C2X_D0], minsc = T[C2X_MINSC], g0 = T[C2X_G0];
    const unsigned* gW = gWords + es * slotWords;
    unsigned* runs = (unsigned*)(c2_smem + P.stage) + es * C2_GRP_SLOT_WORDS;
    int* acc = (int*)(runs + C2_RUNS_MAX * 3);
    const unsigned char* sRead; const unsigned char* sRef; const uint16_t* sIncP;
    if (PK) {
        const unsigned char* gb = c2_smem + P.group0 + (uint32_t)(es >> 1) * P.group_bytes;
        sRead = c2_smem + P.slot0 + (uint32_t)es * P.slot_bytes; sRef = gb + P.gref; sIncP = (const uint16_t*)(gb + P.gincp);
    } else {
        const unsigned char* base = c2_smem + P.slot0 + (uint32_t)es * P.slot_bytes;
        sRead = base + P.read; sRef = base + P.ref; sIncP = (const uint16_t*)(base + P.incp);
    }
    auto nib_at = [&](const int pi, const int pj, unsigned& nib) -> bool {
        const int sl = (pi - pj - d0) >> 1;
        if ((unsigned)sl >= (unsigned)NL) return false;
        const int a = pi + pj;
        nib = c2_word_nib(gW[(a >> 3) * LPW + sl], a, PK);
        return true;
    };

    // ================= the walk =================
    int i = Li, j = Lj, s = C2_ST_M, cnt = 0, nr = 0, status = 0;
    int cur_s = 0, cur_len = 0, cur_i = 0, cur_j = 0, cur_cnt = 0;    // the run being collected (a run the steps below see in pieces is one run)
    bool done = !act, nf = false;
    auto flush = [&]() {
        if (cur_s) {
            if (q == 0 && nr < C2_RUNS_MAX) {
                runs[3 * nr] = (unsigned)cur_cnt | ((unsigned)cur_len << 16);
                runs[3 * nr + 1] = (unsigned)cur_i | ((unsigned)cur_j << 16);
                runs[3 * nr + 2] = (unsigned)cur_s;
            }
            ++nr;
        }
    };
    if (act) {
        unsigned nib = 0;
        if (nib_at(i, j, nib)) s = (nib & 2) ? C2_ST_I : ((nib & 1) ? C2_ST_J : C2_ST_M);   // start state, pyx:349-358
        else { nf = true; done = true; }
    }
    while (true) {
        if (!__ballot(!done)) break;
        const bool live = !done;
        const bool bnd = live && (i == 0 || j == 0);
        const int K = (i < j ? i : j) - 1;
        const int slw = (i - j - d0) >> 1;
        bool wordp = live && !bnd && s == C2_ST_M && K >= 1 && (unsigned)slw < (unsigned)NL;
        bool probe = live && !bnd && !wordp;
        {   // One kind of step per round: the alignments of an iteration have the same shape (M, a gap run, M, ...) but runs of different lengths, so
            // most rounds would find some of them inside a run of M and some inside a gap run and pay for both kinds of look.  The kind more lanes
            // ask for is taken; the others wait a round (they ask again, and a round always advances at least half of the live alignments).
            const int nw = __popcll(__ballot(wordp)), np = __popcll(__ballot(probe));
            if (nw > 0 && np > 0) { if (nw >= np) probe = false; else wordp = false; }
        }
        const bool waits = live && !bnd && !wordp && !probe;
        // ---- a run of state M off the words of its diagonal: word n of the run (n = 0: the word of cell (i-1, j-1)) holds its cells
        //      k = kfirst(n) .. kfirst(n) + 3; lane q takes words q * NW .. q * NW + NW - 1.  Interior cells only (k < K): the cell on a matrix
        //      edge goes through the probe below
        const int a0 = i + j - 2, par = a0 & 1, ctop = a0 & 7, W0 = a0 >> 3, n0 = (ctop >> 1) + 1;
        unsigned stopm = 0u;
        int wl = 0, ncont_w = 0, ns_w = 0;
        if (__ballot(wordp)) {                                      // (a step in which every live alignment is inside a gap run skips this)
            int ncont = 0, ns_dec = 0;
            bool stopped = false;
            if (wordp) {
                // the lane's NW words, all requested at once (a word that holds no cell of the run -- behind word 0 of the plane, or behind cell K - 1 --
                // is read from a clamped address and not used); a word whose four cells all keep the path in M costs a mask and a compare: only the
                // first word that does not is decoded
                unsigned wv[NW];
#pragma unroll
                for (int t = 0; t < NW; ++t) { const int wi = W0 - (q * NW + t); wv[t] = gW[(wi < 0 ? 0 : wi) * LPW + slw]; }
                unsigned wbad = 0u, m4bad = 0u; int clbad = 0, nbad = 0;
                bool open = true;                                    // no word so far ended the lane's look
#pragma unroll
                for (int t = 0; t < NW; ++t) {
                    const int n = q * NW + t;
                    const int kfirst = n == 0 ? 0 : n0 + 4 * (n - 1);
                    int cells = n == 0 ? n0 : 4;
                    if (W0 - n < 0 || kfirst >= K) cells = 0; else if (kfirst + cells > K) cells = K - kfirst;
                    unsigned m4 = c2_word_m4(wv[t], par, PK);
                    if (n == 0) m4 = (m4 << (4 - n0)) & 0xfu;                          // (word 0 holds n0 cells of the run: topmost first from bit 3 on)
                    const bool whole = cells > 0 && (m4 >> (4 - cells)) == ((1u << cells) - 1u);      // every cell of the run in this word keeps the path in M
                    if (open) {
                        if (whole) ncont += cells;
                        else { open = false; wbad = wv[t]; m4bad = m4; clbad = cells; nbad = n; }
                    }
                }
                if (!open && clbad > 0) {
                    const unsigned m4 = m4bad;
                    int n_lead = __builtin_clz((((~m4) & 0xfu) << 28) | 0x08000000u);      // cells from the top that keep the path in M (4: all of them)
                    if (n_lead > clbad) n_lead = clbad;
                    ncont += n_lead;
                    if (n_lead < clbad) {
                        stopped = true;
                        const int c_dec = (nbad == 0 ? ctop : 6 + par) - 2 * n_lead;        // the cell that ends the run
                        ns_dec = (c2_word_nib(wbad, c_dec, PK) & 2u) ? C2_ST_I : C2_ST_J;   // "H is I" first (pyx:349-358 order); one of the two is set
                    }
                }
            }
            stopm = sub(__ballot(stopped));
            wl = stopm ? __builtin_ctz(stopm) : 0;
            ncont_w = __shfl(ncont, e * EL + wl); ns_w = __shfl(ns_dec, e * EL + wl);
        }
        int covered = n0 + 4 * (EL * NW - 1); if (covered > K) covered = K;
        // ---- any other state, and the cells on the matrix edges: lane q probes the q-th cell ahead along the current direction
        const int di = (s != C2_ST_I) ? 1 : 0, dj = (s != C2_ST_J) ? 1 : 0;
        const int ik = i - q * di, jk = j - q * dj;
        const bool valid = probe && ik >= 1 && jk >= 1;
        int ns = 0;
        bool oob = false;
        const bool any_probe = __ballot(probe) != 0ull;
        if (valid) {
            const int pi = (s == C2_ST_M) ? ik - 1 : ik, pj = (s == C2_ST_M) ? jk - 1 : jk;
            if (pi == 0 || pj == 0) ns = c2_boundary_hstate(pi, pj, minsc, ge, g0);      // only reachable for s == M
            else {
                unsigned nib = 0;
                if (nib_at(pi, pj, nib)) {
                    if (s == C2_ST_M) ns = (nib & 2) ? C2_ST_I : ((nib & 1) ? C2_ST_J : C2_ST_M);
                    else if (s == C2_ST_I) ns = (nib & 8) ? C2_ST_M : C2_ST_I;
                    else ns = (nib & 4) ? C2_ST_M : C2_ST_J;
                } else oob = true;
            }
        }
        int nv = 0, nc = 0, ns_c = 0;
        unsigned om = 0u;
        if (any_probe) {
            const unsigned vm = sub(__ballot(valid)), cm = sub(__ballot(valid && ns == s));
            om = sub(__ballot(oob));
            nv = (~vm & (unsigned)gmask) ? __builtin_ctz(~vm & (unsigned)gmask) : EL;
            nc = (~cm & (unsigned)gmask) ? __builtin_ctz(~cm & (unsigned)gmask) : EL;
            ns_c = __shfl(ns, e * EL + (nc < EL ? nc : EL - 1));
        }
        // ---- the step
        if (live && !waits) {
            int E = 0, s_next = s;
            bool fin = false;
            if (bnd) {
                const int need = (i == 0) ? C2_ST_I : C2_ST_J;       // initialised chains: iPointer[0,1:], jPointer[1:,0]
                if (s != need) status |= (s == C2_ST_M) ? C2_STATUS_SENTINEL_PATH : C2_STATUS_UNINIT_PTR;
                else E = (i == 0) ? j : i;
                fin = true;
            } else if (wordp) {
                if (stopm == 0u) { E = covered; s_next = C2_ST_M; }
                else { const int nfirst = wl * NW; E = (nfirst == 0 ? 0 : n0 + 4 * (nfirst - 1)) + ncont_w + 1; s_next = ns_w; }
            } else {
                if (nc < nv) {
                    if ((om >> nc) & 1u) { nf = true; fin = true; }              // the deciding pointer word is not in this plane
                    else { E = nc + 1; s_next = ns_c; }
                } else { E = nv; s_next = s; }
            }
            if (E > 0) {
                if (s == cur_s) cur_len += E;
                else { flush(); cur_s = s; cur_len = E; cur_i = i; cur_j = j; cur_cnt = cnt; }
                cnt += E;
                if (bnd) { i = 0; j = 0; } else { i -= E * ((s != C2_ST_I) ? 1 : 0); j -= E * ((s != C2_ST_J) ? 1 : 0); }
            }
            s = s_next;
            if (fin || (i == 0 && j == 0)) done = true;
        }
    }
    flush();
    if (nr > C2_RUNS_MAX) nf = true;
    const bool ok = act && !nf && status == 0 && !(A.reserved & 16);     // (16: debug knob C2_DEBUG_SKIP_EMIT -- the walk alone)
    const int TT = cnt;                                              // columns of the alignment
    if (e < NAH && q < 8) { acc[q] = 0; if (q < 4) acc[8 + q] = 0; }
    __builtin_amdgcn_wave_barrier();                                 // (the run tables and the zeroed accumulators: LDS operations of one wavefront complete in order)

    // ================= indel events, a lane per run =================
    int ev_counts = 0, ev_ins_n = 0, ev_del_n = 0, ev_del_bases = 0, flags = 0;
#pragma unroll
    for (int rb = 0; rb < C2_RUNS_MAX; rb += EL) {
        const int r = rb + q;
        if (ok && r < nr) {
            const unsigned ra = runs[3 * r], rbw = runs[3 * r + 1];
            const int st = (int)runs[3 * r + 2];
            const int len_r = (int)(ra >> 16), cs_r = TT - (int)(ra & 0xffffu) - len_r, i_r = (int)(rbw & 0xffffu);
            if ((r == 0 || r == nr - 1) && st != C2_ST_M) flags |= 2;                     // a gap in the first or the last column (CRISPRessoCORE.py:729-733)
            if (st == C2_ST_I) {
                // a gap in the reference string closes at the next column that has a reference base (pyx:119-128): never for a trailing run (run 0),
                // and a leading one (no reference base in front) was never opened (pyx:136)
                if (i_r > 0 && r != 0) {
                    const bool fl = sIncP[i_r] != sIncP[i_r - 1], fr = sIncP[i_r + 1] != sIncP[i_r];
                    const bool win = A.legacy ? (fl || fr) : (fl && fr);                // both flanks in the window (pyx:121); legacy: either (pyx:284)
                    ev_counts += 1 + (win ? 0x100 : 0);
                    if (win) ev_ins_n += len_r;
                }
            } else if (st == C2_ST_J) {
                bool win; int bases;
                if (r != 0) {                                                           // closes at the next read base (pyx:145-153); i_r = reference bases left of that column
                    const int dstart = (A.legacy && cs_r - 1 <= 0) ? 0 : i_r - len_r;   // legacy (pyx:253-258): a run that starts in column 0 or 1 is given start 0
                    win = sIncP[i_r] != sIncP[dstart];
                    bases = i_r - dstart;
                } else if (!A.legacy) {                                                 // trailing deletion (pyx:155-162)
                    bases = len_r;
                    win = sIncP[Li] != sIncP[Li - len_r];
                } else {                                                                // legacy (pyx:259-261): ends at reference index Li - 1 (exclusive)
                    const int dstart = (cs_r - 1 <= 0) ? 0 : Li - len_r, dend = Li - 1;
                    bases = dend > dstart ? dend - dstart : 0;
                    win = dend > dstart && sIncP[dend] != sIncP[dstart];
                }
                ev_counts += 0x10000 + (win ? 0x1000000 : 0);
                ev_del_bases += bases;
                if (win) ev_del_n += len_r;
            }
        }
    }

    // ================= the strings, forwards =================
    uint8_t* outR = A.aln_read + (uint64_t)(unsigned)T[C2X_TASK_LO] * (uint64_t)A.aln_stride;
    uint8_t* outF = A.aln_ref + (uint64_t)(unsigned)T[C2X_TASK_LO] * (uint64_t)A.aln_stride;
    const bool strings = !(A.reserved & 1);
    const int lastR = (Lj - 1) >> 2, lastF = (Li - 1) >> 2;
    int n_mism = 0, n_sub = 0, n_win = 0;
    bool dash = false;
    // (c2_batch.diag_hints: an alignment of at most five runs and three differing columns leaves as four words -- its runs, and where and what its differing
    //  columns are --, which is all the count pass needs of it: c2_count_hinted_kernel.  Not under the legacy classifier, whose positions differ.)
    const bool want_hint = A.diag_hints != nullptr && !A.legacy && nr <= 5 && Li <= 511 && Lj <= 511;
    struct run_d { int cs, ce, dR, dF, st, len; };                 // a run as the strings need it: columns [cs, ce); read / reference index of its column c: c + dR / c + dF
    auto load_run = [&](const int r, run_d& d) {
        const unsigned ra = runs[3 * r], rbw = runs[3 * r + 1];
        d.st = (int)runs[3 * r + 2];
        d.len = (int)(ra >> 16); d.cs = TT - (int)(ra & 0xffffu) - d.len; d.ce = d.cs + d.len;
        d.dR = (int)(rbw >> 16) - d.len - d.cs; d.dF = (int)(rbw & 0xffffu) - d.len - d.cs;
    };
    auto store4 = [&](const int c0, const unsigned r4, const unsigned f4, const int nb) {     // nb > 0 bytes of the dword at column c0 are columns
        if (!strings) return;
        if (rows_aligned && nb >= 4) { ((uint32_t*)outR)[c0 >> 2] = r4; ((uint32_t*)outF)[c0 >> 2] = f4; }
        else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb) { outR[c0 + b] = (uint8_t)(r4 >> (8 * b)); outF[c0 + b] = (uint8_t)(f4 >> (8 * b)); }
        }
    };
    // a dword that straddles run boundaries, byte by byte from the first run on (rare: an alignment has a boundary or two)
    auto do_bytes = [&](const int c0) {
        int r2 = nr - 1, il = 0;
        run_d D;
        load_run(r2, D);
        unsigned r4 = 0u, f4 = 0u;
        int nb = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = c0 + b;
            if (c < TT) {
                while (c >= D.ce && r2 > 0) { if (D.st == C2_ST_I) il += D.len; --r2; load_run(r2, D); }
                const int st = D.st;
                const unsigned char rch = (st != C2_ST_J) ? sRead[c + D.dR] : (unsigned char)'-';
                const unsigned char fch = (st != C2_ST_I) ? sRef[c + D.dF] : (unsigned char)'-';
                r4 |= (unsigned)rch << (8 * b); f4 |= (unsigned)fch << (8 * b);
                nb = b + 1;
                if (st == C2_ST_M) {
                    if (rch != fch) {
                        ++n_mism;
                        const int idx = c - il;
                        if (rch != 'N') { ++n_sub; n_win += (sIncP[idx + 1] != sIncP[idx]) ? 1 : 0; }
                        if (want_hint) c2_hint_note(acc + 8, idx, (unsigned)rch);
                    }
                    if (rch == '-' || fch == '-') dash = true;
                } else if ((st == C2_ST_I && rch == '-') || (st == C2_ST_J && fch == '-')) dash = true;
            }
        }
        store4(c0, r4, f4, nb);
    };
    unsigned defer = 0u;
    int ins_left = 0;                                               // insertion columns of the runs in front of the current one
    int rp = 0;
    run_d C; C.cs = 0; C.ce = 0; C.dR = 0; C.dF = 0; C.st = 0; C.len = 0;          // the run this lane's columns are in
    if (ok) { rp = nr - 1; load_run(rp, C); }
    for (int t = 0;; ++t) {
        const int c0 = 4 * (q + EL * t);
        const bool on = ok && c0 < TT;
        if (!__ballot(on)) break;
        if (on) {
            while (c0 >= C.ce && rp > 0) { if (C.st == C2_ST_I) ins_left += C.len; --rp; load_run(rp, C); }
            const int st = C.st, dR = C.dR, dF = C.dF;
            if (c0 + 4 <= C.ce || rp == 0) {
                // the whole dword lies in the run (or the run is the last one: the bytes behind column T - 1 are cut off)
                const int nb = TT - c0;
                const unsigned valid = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
                const unsigned r4 = ((st != C2_ST_J) ? c2_lds_load4(sRead, c0 + dR, lastR) : 0x2d2d2d2du) & valid;
                const unsigned f4 = ((st != C2_ST_I) ? c2_lds_load4(sRef, c0 + dF, lastF) : 0x2d2d2d2du) & valid;
                store4(c0, r4, f4, nb);
                const unsigned vb = valid & 0x80808080u;
                if (st == C2_ST_M) {
                    const unsigned mm = c2_nonzero_bytes(r4 ^ f4) & vb;                               // columns whose characters differ
                    const unsigned sb = mm & c2_nonzero_bytes(r4 ^ 0x4e4e4e4eu);                     // ... and the read's is not 'N' (COREResources.pyx:113-118)
                    if (((~c2_nonzero_bytes(r4 ^ 0x2d2d2d2du) | ~c2_nonzero_bytes(f4 ^ 0x2d2d2d2du)) & vb) != 0u) dash = true;
                    if (mm != 0u) {
                        n_mism += __builtin_popcount(mm);
                        n_sub += __builtin_popcount(sb);
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            if ((mm >> (8 * b + 7)) & 1u) {
                                const int idx = c0 + b - ins_left;                                      // reference index of the column
                                if ((sb >> (8 * b + 7)) & 1u) n_win += (sIncP[idx + 1] != sIncP[idx]) ? 1 : 0;
                                if (want_hint) c2_hint_note(acc + 8, idx, (r4 >> (8 * b)) & 0xffu);
                            }
                    }
                } else {
                    const unsigned x = (st == C2_ST_I) ? r4 : f4;                                    // a literal '-' in the sequence that fills the gap run
                    if ((~c2_nonzero_bytes(x ^ 0x2d2d2d2du) & vb) != 0u) dash = true;
                }
            } else if (t < 32) defer |= 1u << t;
            else do_bytes(c0);
        }
    }
    while (true) {
        if (!__ballot(defer != 0u)) break;
        if (defer != 0u) { const int t = __builtin_ctz(defer); defer &= defer - 1u; do_bytes(4 * (q + EL * t)); }
    }
    // ================= sums, records =================
    if (ok) {
        if (q == 0) {
            const unsigned char r0 = sRead[0], f0 = sRef[0], rL = sRead[Lj - 1], fL = sRef[Li - 1];
            if (r0 != f0 || rL != fL) flags |= 2;
        }
        if (dash) flags |= 1;
        if (n_mism) atomicAdd(&acc[0], n_mism);
        if (n_sub) atomicAdd(&acc[1], n_sub | (n_win << 16));
        if (ev_counts) atomicAdd(&acc[2], ev_counts);
        if (ev_ins_n | ev_del_n) atomicAdd(&acc[3], ev_ins_n | (ev_del_n << 16));
        if (ev_del_bases) atomicAdd(&acc[4], ev_del_bases);
        if (flags) atomicOr(&acc[5], flags);
    }
    if (e < NAH && q == 0) { acc[6] = TT; acc[7] = (act ? 1 : 0) | (nf ? 2 : 0) | (status << 8) | ((nr < 255 ? nr : 255) << 16); }
    __builtin_amdgcn_wave_barrier();
    if (lane < NAH && ((m_trace >> lane) & 1u)) {
        const int* Ts = sTab + (sbase + lane) * C2X_INTS;
        const int* ac = (const int*)((unsigned*)(c2_smem + P.stage) + (sbase + lane) * C2_GRP_SLOT_WORDS + C2_RUNS_MAX * 3);
        const uint64_t task = (uint64_t)(unsigned)Ts[C2X_TASK_LO] | ((uint64_t)(unsigned)Ts[C2X_TASK_HI] << 32);
        int st8 = (ac[7] >> 8) & 0xff;
        const bool need_full = (ac[7] & 2) != 0 || (st8 == 0 && (ac[5] & 1) != 0);
        c2_aln_record rec;
        c2_clear_record(rec, Ts[C2X_RC], Ts[C2X_REF]);
        if (!need_full && st8 == 0) {
            const int Tn = ac[6], sLi = Ts[C2X_LI], sLj = Ts[C2X_LJ];
            rec.aln_len = (uint16_t)Tn;
            rec.matches = (uint16_t)(sLi + sLj - Tn - ac[0]);        // M columns = Li + Lj - T (pyx:375-376 counts the equal ones)
            rec.insertion_n = (uint16_t)(ac[3] & 0xffff);
            rec.deletion_n = (uint16_t)((unsigned)ac[3] >> 16);
            rec.substitution_n = (uint16_t)((unsigned)ac[1] >> 16);
            rec.all_insertion_events = (uint16_t)(ac[2] & 0xff);
            rec.win_insertion_events = (uint16_t)((ac[2] >> 8) & 0xff);
            rec.all_deletion_events = (uint16_t)((ac[2] >> 16) & 0xff);
            rec.win_deletion_events = (uint16_t)((ac[2] >> 24) & 0xff);
            rec.all_deletion_bases = (uint16_t)ac[4];
            rec.all_substitutions = (uint16_t)(ac[1] & 0xffff);
            rec.irregular_ends = (ac[5] & 2) ? 1 : 0;
            if (A.diag_hints != nullptr && !A.legacy && sLi <= 511 && sLj <= 511 && ac[0] <= 3 && ac[8] == ac[0]) {
                // the hint (include/crispresso2_amd.h): runs in forward order as state | length << 2, eleven bits each; a run of M of a single column is left to
                // the count pass's column walk (two insertions one reference base apart share a position: numpy's repeated index, CRISPRessoCORE.py:4016-4021)
                const unsigned* rt = (const unsigned*)(c2_smem + P.stage) + (sbase + lane) * C2_GRP_SLOT_WORDS;
                const int nrs = (ac[7] >> 16) & 0xff;
                unsigned hw[4] = {0u, 0u, 0u, 0u};
                bool fits = nrs >= 1 && nrs <= 5;
                for (int f = 0; f < 5 && fits; ++f) {
                    if (f >= nrs) break;
                    const int r = nrs - 1 - f;                       // trace order -> forward order
                    const unsigned len = rt[3 * r] >> 16, stt = rt[3 * r + 2] & 3u;
                    if (len > 511u || (stt == (unsigned)C2_ST_M && len < 2u && nrs > 1)) { fits = false; break; }
                    const unsigned fld = stt | (len << 2);
                    if (f == 0) hw[0] |= fld << 5; else if (f == 1) hw[0] |= fld << 16;
                    else if (f == 2) hw[1] |= fld; else if (f == 3) hw[1] |= fld << 11; else hw[2] |= fld;
                }
                if (fits) {
                    hw[0] |= C2_HINT_GAPPED | (unsigned)nrs | ((unsigned)ac[0] << 3);
                    if (ac[0] > 0) hw[2] |= ((unsigned)ac[9] & 0xfffu) << 11;
                    if (ac[0] > 1) hw[3] |= (unsigned)ac[10] & 0xfffu;
                    if (ac[0] > 2) hw[3] |= ((unsigned)ac[11] & 0xfffu) << 12;
                    uint32_t* hp = A.diag_hints + 4u * task;
                    hp[0] = hw[0]; hp[1] = hw[1]; hp[2] = hw[2]; hp[3] = hw[3];
                }
            }
        }
        if (need_full) {
            st8 |= C2_STATUS_NEED_FULL;
            const unsigned k = atomicAdd(A.fb_count, 1u);
            A.fb_list[k] = (uint32_t)task;
        }
        rec.status = (uint8_t)st8;
        A.records[task] = rec;
    }
}

#ifndef C2X_GRAB
#define C2X_GRAB 8                                  // groups of NA positions per grab of the work counter, at most
#endif
template <int NA, bool PK, bool ADD32 = false, bool SCORE = false>
__device__ __forceinline__ void c2_diagx_body(const c2_align_args& A)
{
    const int beta = (PK && ADD32) ? (int)A.pk_beta : 0;               // per-anti-diagonal bias of the 32-bit-add variant (else 0)
    const int PKB = (PK && ADD32) ? (int)A.pk_bias : C2_PK_BIAS;        // the value bias: the smallest one that keeps c2_pk_eligible's margin there
    // PK (c2_align_diagp_kernel): NA alignments in NA / 2 lane groups, two per group (slots 2g and 2g+1 in the two halves of the lanes' registers)
    constexpr int NG = PK ? NA / 2 : NA;                             // lane groups
    // lanes per group, live lanes, diagonals per band.  A packed group of 16 lanes is a DPP row: its hand-off is row_shr / row_shl,
    // which already reads 0 at the row's ends, so no lane is switched off and all 16 hold diagonals (c2_rshr1z)
    constexpr int LPA = 64 / NG;
    constexpr int LPW = (LPA + 3) & ~3;                               // words per group of 8 anti-diagonals in the plane (c2_make_diagx_plan)
    constexpr bool ROWDPP = PK && LPA == 16;
    constexpr int NL = (ROWDPP || NG == 1) ? LPA : LPA - 1, BANDW = 2 * NL;   // (one group = the whole wavefront: its ends read 0 anyway)
    const int lane = threadIdx.x, grp = lane / LPA, sl = lane - grp * LPA;
    const int slot = PK ? 2 * grp : grp;                             // (PK: the group's first slot)
    const c2_diagx_plan P = c2_make_diagx_plan(NA, A.max_li, A.max_lj, PK, SCORE);
    unsigned char* sCodeOf = c2_smem + P.codeof;
    int* sTab = (int*)(c2_smem + P.table);
    auto wg_of = [&](const int s) {
        unsigned char* base = c2_smem + P.slot0 + (uint32_t)s * P.slot_bytes;
        c2_wg W;
        if (PK) {                                                  // the read per alignment, reference and window prefix per lane group
            unsigned char* gb = c2_smem + P.group0 + (uint32_t)(s >> 1) * P.group_bytes;
            W.sRead = base; W.sCode = nullptr; W.sRef = gb + P.gref; W.sIncP = (uint16_t*)(gb + P.gincp);
        } else {
            W.sRead = base + P.read; W.sCode = base + P.code; W.sRef = base + P.ref; W.sIncP = (uint16_t*)(base + P.incp);
        }
        W.sTmpRead = c2_smem + P.tmp_read; W.sTmpRef = c2_smem + P.tmp_ref;
        return W;
    };
    auto win_of = [&](const int s) { return PK ? (uint32_t*)nullptr : (uint32_t*)(c2_smem + P.slot0 + (uint32_t)s * P.slot_bytes + P.win); };
    // the column table the fill reads: per alignment (4 * code), or per lane group (PK: the pair symbols)
    auto coltab_of = [&](const int s) {
        return PK ? c2_smem + P.pcodes0 + (uint32_t)(s >> 1) * P.pcodes_bytes : c2_smem + P.slot0 + (uint32_t)s * P.slot_bytes + P.codes;
    };
    // rows of the output arrays can be written as dwords (c2_emit_gapless4) when their addresses are multiples of 4
    const bool rows_aligned = ((((uintptr_t)A.aln_read | (uintptr_t)A.aln_ref) & 3u) == 0) && ((A.aln_stride & 3u) == 0);
    unsigned* sStage = (unsigned*)(c2_smem + P.stage);
    unsigned* gWords = A.plane + (size_t)blockIdx.x * A.plane_words_per_wg;   // [slot][group][lane of the slot]
    const int slotWords = (int)P.n_words * LPW;
    const int ge = A.gap_extend, go = A.gap_open;
    for (int k = lane; k < 256; k += 64) sCodeOf[k] = A.code_of_char[k];
    if (lane < NA) { sTab[lane * C2X_INTS + C2X_CURREF] = -1; sTab[lane * C2X_INTS + C2X_LI] = 0; sTab[lane * C2X_INTS + C2X_G0] = 0; sTab[lane * C2X_INTS + C2X_REFBAD] = 0; }
    if (!PK)
        for (int s = 0; s < NA; ++s)                               // zeros in front of column 1 of every slot's symbol table (written once)
            if (lane <= C2_DIAG_CODE_PAD) c2_smem[P.slot0 + (uint32_t)s * P.slot_bytes + P.codes + lane] = 0;

    if (PK) {
        // pair-score tables: for reference symbol rc (codes 0..4; table 5 = zeros, for the padding rows) and read symbols (cA, cB) the two
        // int16 scores side by side, at byte offset rc * C2_PK_LUT_STRIDE + (cA << 5 | cB << 2) -- the column table holds that pair symbol
        unsigned* lut = (unsigned*)(c2_smem + P.pairlut);
        for (int e = lane; e < C2_PK_LUT_CODES * 64; e += 64) {
            const int rc = e >> 6, cA = (e >> 3) & 7, cB = e & 7;
            unsigned v = 0;
            if (rc < C2_PK_PAD_TABLE && rc < A.n_codes)
                v = ((unsigned)(c2_sbfe4((int)A.score_pk[rc], 4 * cA) + 2 * beta) & 0xffffu) | ((unsigned)(c2_sbfe4((int)A.score_pk[rc], 4 * cB) + 2 * beta) << 16);
            lut[rc * (int)(C2_PK_LUT_STRIDE / 4u) + (e & 63)] = v;
        }
    }
    c2_phase_acc PH; PH.t_last = 0; PH.sum[0] = PH.sum[1] = PH.sum[2] = PH.sum[3] = 0;
    // Task fetch as a four-stage software pipeline, one stage per group of NA alignments, so that no stage ever waits for
    // the memory access it depends on (in list mode every one of them misses the caches):
    //   A0  atomic on the work counter for the group four iterations ahead
    //   A   lane s < NA: task index of slot s (from the task list, if any)              -- three ahead
    //   B   lane s < NA: read offsets, reference id and strand of that task             -- two ahead
    //   C   the first 256 read bytes of every slot, lanes = bytes (c2_prefetch)         -- one ahead
    //   D   c2_commit_task into LDS, then the fill                                      -- this iteration
    // Each stage consumes what the previous iteration's earlier stage requested; the barrier at the top of the loop has
    // waited for all of it.  Task indices fit 32 bits in these launches (the host checks).
    const bool pair_order = PK && A.pair_order && !A.task_list && A.all_refs;
    const uint64_t n_reads_po = pair_order ? A.n_tasks / (uint64_t)A.n_refs : 0;
    const uint64_t n_iter = A.task_list ? (uint64_t)(*A.task_count) : (pair_order ? ((n_reads_po + 1) >> 1) * 2u * (uint64_t)A.n_refs : A.n_tasks);
    unsigned long long pend = 0;
    bool pend_valid = false, exhausted = false;
    // The work counter hands out BLOCKS of C2X_GRAB groups of NA consecutive positions (round 5): a list's neighbours share their reference (the
    // partition writes a chunk's tasks reference-major, class by class), and a workgroup that walks 8 consecutive groups re-stages the reference
    // of its lane groups once where it did for every group -- with three candidate amplicons the score-only launch spent two thirds of its time
    // there (5.8 ns per task against 1.6 with one amplicon).  One atomic per block instead of one per group.
    // (a block is C2X_GRAB groups when the launch has work for at least four blocks per workgroup, fewer for a short list: a million-task batch
    //  over 2,300 workgroups in blocks of 128 tasks left some of them with three blocks and some with two)
    //  -- and only for a batch of several references (A.reserved bit 3): with one reference nothing is re-staged, and single groups balance the
    //  workgroups better (measured on the headline batch: 34.7 ms with single groups, 35.5 with blocks of 8, 36.1 with 16)
    int grab = (A.reserved & 8) ? (int)(n_iter / ((uint64_t)gridDim.x * (uint64_t)(NA * 4))) : 1;
    grab = grab < 1 ? 1 : (grab > C2X_GRAB ? C2X_GRAB : grab);
    unsigned long long blk_next = 0;                                // next position of the current block (wave-uniform)
    int blk_left = 0;                                               // groups of it not handed out yet
    bool pend_atomic = false;
    unsigned mA_task = 0; int mA_valid = 0;
    unsigned mB_task = 0; int mB_valid = 0, mB_ref = 0, mB_rc = 0;
    unsigned long long mB_off = 0, mB_off1 = 0;
    // stage C keeps its per-slot descriptors in lane s of a few VGPRs too (mC_*: no SGPR is live across the fill); only the
    // read bytes need a register per slot
    unsigned mC_task = 0; int mC_valid = 0, mC_lj = 0, mC_ref = 0, mC_rc = 0;
    unsigned long long mC_off = 0;
    // (sixteen scalars, not an array: as an array they become ONE 16-register tuple, and copies of that tuple were spilled around the fill)
    unsigned b4_0 = 0, b4_1 = 0, b4_2 = 0, b4_3 = 0, b4_4 = 0, b4_5 = 0, b4_6 = 0, b4_7 = 0, b4_8 = 0, b4_9 = 0, b4_10 = 0, b4_11 = 0, b4_12 = 0, b4_13 = 0, b4_14 = 0, b4_15 = 0;
    auto b4s = [&](const int s) -> unsigned& {                       // (s is a constant wherever this is called: unrolled loops)
        switch (s) {
            case 0: return b4_0; case 1: return b4_1; case 2: return b4_2; case 3: return b4_3; case 4: return b4_4; case 5: return b4_5;
            case 6: return b4_6; case 7: return b4_7; case 8: return b4_8; case 9: return b4_9; case 10: return b4_10; case 11: return b4_11;
            case 12: return b4_12; case 13: return b4_13; case 14: return b4_14; default: return b4_15;
        }
    };
    auto lane64 = [&](const unsigned long long v, const int s) {
        return (uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), s) |
               ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), s) << 32);
    };
    for (int iter = 0;; ++iter) {
        __syncthreads();
        const bool have_group = __builtin_amdgcn_readlane(mC_valid, 0) != 0;
        if (iter >= 4 && !have_group) break;
        c2_phase_begin(A.phase_cycles, PH);
        // ---- D: stage the NA prefetched tasks in their LDS slots
        if (have_group) {
            // what needs no decision goes into LDS for all NA slots at once: the prefetched dwords into the slots' read buffers (the staging below
            // reads its slot's back: one ds_read instead of a select over NA registers), the descriptors lane s holds into the table
            {
                if (4 * lane < (int)P.slot_read_bytes) {
                    uint32_t* rd0 = (uint32_t*)(c2_smem + P.slot0 + P.read) + lane;     // (slot s's read buffer: wg_of(s).sRead)
                    const uint32_t stride = P.slot_bytes / 4u;
#pragma unroll
                    for (int s = 0; s < NA; ++s) rd0[(uint32_t)s * stride] = b4s(s);
                }
                if (lane < NA) {
                    int* T = sTab + lane * C2X_INTS;
                    T[C2X_VALID] = mC_valid; T[C2X_TASK_LO] = (int)mC_task; T[C2X_TASK_HI] = 0;
                    T[C2X_LJ] = mC_lj; T[C2X_REF] = mC_ref; T[C2X_RC] = mC_rc;
                }
            }
#pragma nounroll                                                     // (one copy of the staging code)
            for (int s = 0; s < NA; ++s) {
                int* T = sTab + s * C2X_INTS;
                const int tvd = c2_tab_load(T, lane);
                int cref = C2_TF(tvd, C2X_CURREF), li = C2_TF(tvd, C2X_LI), g0 = C2_TF(tvd, C2X_G0), rbad = C2_TF(tvd, C2X_REFBAD);
                int st = 0;
                bool packed = false, unpaired = false;
                c2_prefetch cur;
                cur.valid = __builtin_amdgcn_readlane(mC_valid, s);
                cur.task = (uint64_t)(unsigned)__builtin_amdgcn_readlane((int)mC_task, s);
                cur.off = lane64(mC_off, s);
                cur.Lj = __builtin_amdgcn_readlane(mC_lj, s); cur.ref_id = __builtin_amdgcn_readlane(mC_ref, s);
                cur.rc = __builtin_amdgcn_readlane(mC_rc, s); cur.b4 = 0;
                if (cur.valid) {
                    unsigned char* sCodes4 = coltab_of(s);
                    bool done = false;
                    // PK: the second alignment of a lane group (odd slot) shares the group's reference and column table with the first: it
                    // joins only if the first one is staged and runs (valid, no status, packed), against the same reference, with a read
                    // of the same length -- otherwise it is handed to the next launch (packed = false, no status)
                    bool joins = true;
                    const bool second = PK && (s & 1);
                    if (second) {
                        const int tva = c2_tab_load(T - C2X_INTS, lane);
                        joins = C2_TF(tva, C2X_VALID) && C2_TF(tva, C2X_STATUS) == 0 && C2_TF(tva, C2X_PACKED) &&
                                C2_TF(tva, C2X_REF) == cur.ref_id && C2_TF(tva, C2X_LJ) == cur.Lj;
                        cref = C2_TF(tva, C2X_CURREF); li = C2_TF(tva, C2X_LI); g0 = C2_TF(tva, C2X_G0); rbad = C2_TF(tva, C2X_REFBAD);   // the group's reference
                    }
                    if (joins && !cur.rc && cur.Lj >= 4 && cur.Lj <= 256 && cur.Lj <= A.max_lj) {
                        // the usual read: forward strand, at most 256 bases, nothing but A C G T N.  Four bases per lane: codes through two
                        // byte permutes ((ch >> 1) & 7 is a perfect hash of the five letters), checked by permuting the letters back.
                        const c2_wg W = wg_of(s);
                        if (!second && cur.ref_id != cref) { cref = cur.ref_id; c2_stage_ref(A, W, sCodeOf, cur.ref_id, lane, A.max_li, li, g0, rbad, win_of(s)); }
                        const uint32_t w = ((const uint32_t*)W.sRead)[4 * lane < (int)P.slot_read_bytes ? lane : 0];
                        const uint32_t idx = (w >> 1) & 0x07070707u;
                        const uint32_t codes = __builtin_amdgcn_perm(A.lut_code_hi, A.lut_code_lo, idx);
                        const uint32_t chk = __builtin_amdgcn_perm(A.lut_chr_hi, A.lut_chr_lo, idx);
                        const int nb = cur.Lj - 4 * lane;
                        const uint32_t valid = nb >= 4 ? 0xffffffffu : (nb > 0 ? ((1u << (8 * nb)) - 1u) : 0u);
                        if (__ballot(((chk ^ w) & valid) != 0) == 0ull) {
                            uint32_t* col = (uint32_t*)(sCodes4 + C2_DIAG_CODE_PAD + 1);
                            if (nb > 0) {                                                    // (the read itself is in its buffer already)
                                if (!PK) col[lane] = (codes & valid) << 2;                    // 4 * code per column, zeros behind the last one
                                else if (!second) col[lane] = (codes & valid) << 5;          // pair symbol: code A << 5 ...
                                else col[lane] |= (codes & valid) << 2;                      // ... | code B << 2
                            }
                            if (!second && lane < 3) col[((cur.Lj + 3) >> 2) + lane] = 0u;    // ... nine zeros at least
                            if (PK && !second && lane < (C2_DIAG_CODE_PAD + 1) / 4) ((uint32_t*)sCodes4)[lane] = 0u;   // (the table shares its bytes with the traceback strings: zeros in front every time)
                            st = (rbad ? C2_STATUS_OOB_CHAR : 0) | (li <= 0 ? C2_STATUS_EMPTY : 0) | (li > A.max_li ? C2_STATUS_TOO_LONG : 0);
                            packed = true;
                            done = true;
                        }
                    }
                    if (!done && joins) {
                        if (PK && !second && lane < (C2_DIAG_CODE_PAD + 1) / 4) ((uint32_t*)sCodes4)[lane] = 0u;
                        st = c2_commit_task<false>(A, wg_of(s), sCodeOf, cur, lane, A.max_li, cref, li, g0, rbad, packed, sCodes4, win_of(s),
                                                   PK && !second ? 5 : 2, second, !second);
                    }
                    if (!joins) { st = 0; packed = false; }
                    unpaired = !joins;
                }
                if (lane == 0) {
                    T[C2X_STATUS] = st; T[C2X_PACKED] = packed ? 1 : 0;
                    T[C2X_CURREF] = cref; T[C2X_LI] = li; T[C2X_G0] = g0; T[C2X_REFBAD] = rbad; T[C2X_UNPAIRED] = unpaired ? 1 : 0;
                }
            }
        }
        // ---- C: read bytes of the next group
        mC_valid = mB_valid; mC_task = mB_task; mC_off = mB_off; mC_ref = mB_ref; mC_rc = mB_rc;
        mC_lj = mB_valid ? (int)(mB_off1 - mB_off) : 0;
        if (PK && !pair_order && !(A.reserved & 64)) {              // (64: C2_NO_PAIR_SORT, an A/B knob)
            // Round 6: two alignments share a lane group only if they share reference and read length (the staging's `joins`), and the lists behind the first
            // launch are in the order of their atomics: neighbours there are whatever failed next to each other (three amplicons, or reads of 248 .. 250
            // bases: two thirds of a list's tasks went to the 32-bit twin at four times the cost).  The NA tasks a wavefront holds are put in the order
            // of their keys first -- whole pairs in front, then the tasks left over, then the empty slots.  Nothing depends on a task's slot.
            const bool v = lane < NA && mB_valid;
            const unsigned none = 0xffffffffu;
            const unsigned key = v ? (((unsigned)mB_ref << 12) | (unsigned)(mC_lj < 4095 ? mC_lj : 4095)) : none;
            const unsigned k0 = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
            if (__ballot(lane < NA && key != k0) != 0ull) {
                int r = 0, c = 0;                                   // tasks of my key in front of me / in all
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, j);
                    c += kj == key ? 1 : 0; r += (kj == key && j < lane) ? 1 : 0;
                }
                int pb = 0, lb = 0, P2 = 0, nv = 0;                 // pairs / left-over tasks of smaller keys, pairs in all, tasks in all
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, j);
                    const int cj = __builtin_amdgcn_readlane(c, j), rj = __builtin_amdgcn_readlane(r, j);
                    if (kj != none) {
                        ++nv;
                        if (rj == 0) { P2 += cj >> 1; if (kj < key) { pb += cj >> 1; lb += cj & 1; } }
                    }
                }
                const int slot = key == none ? nv + r : (r < 2 * (c >> 1) ? 2 * pb + r : 2 * P2 + lb);
                int src = lane;
#pragma unroll
                for (int j = 0; j < NA; ++j) if (__builtin_amdgcn_readlane(slot, j) == lane) src = j;
                mC_valid = __shfl(mB_valid, src); mC_task = (unsigned)__shfl((int)mB_task, src); mC_ref = __shfl(mB_ref, src); mC_rc = __shfl(mB_rc, src);
                mC_lj = __shfl(mC_lj, src);
                mC_off = (unsigned long long)(unsigned)__shfl((int)(unsigned)mB_off, src) | ((unsigned long long)(unsigned)__shfl((int)(unsigned)(mB_off >> 32), src) << 32);
            }
        }
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            // one dword per lane: bytes 4l .. 4l+3 of the read (what the fast staging above takes); the dword that holds the read's
            // last bytes is loaded so that it ENDS at the last byte and shifted down -- nothing behind the read is touched.  Reads
            // of fewer than 4 or more than 256 bases and reverse-complemented ones are staged by c2_commit_task from memory.
            const int v = __builtin_amdgcn_readlane(mC_valid, s);
            unsigned b4 = 0;
            if (v) {
                const uint64_t off = lane64(mC_off, s);
                const int Lj = __builtin_amdgcn_readlane(mC_lj, s);
                const int p = 4 * lane;
                if (Lj >= 4 && p < Lj) {
                    const int q = p < Lj - 4 ? p : Lj - 4;
                    uint32_t w;
                    __builtin_memcpy(&w, A.reads + off + (uint64_t)q, 4);
                    b4 = w >> (8 * (p - q));
                }
            }
            b4s(s) = b4;
        }
        // ---- B: descriptors of the group after that
        mB_valid = mA_valid; mB_task = mA_task; mB_off = 0; mB_off1 = 0; mB_ref = 0; mB_rc = 0;
        if (lane < NA && mA_valid) {
            unsigned read_id;
            if (A.all_refs) { read_id = mA_task / (unsigned)A.n_refs; mB_ref = (int)(mA_task % (unsigned)A.n_refs); }
            else            { read_id = mA_task; mB_ref = A.ref_ids ? (int)A.ref_ids[mA_task] : 0; }
            mB_rc = A.strands ? (int)A.strands[mA_task] : 0;
            mB_off = A.offsets[read_id]; mB_off1 = A.offsets[read_id + 1];
        }
        // ---- A: task indices of the group after that;  A0: the work counter for the one after
        mA_valid = 0; mA_task = 0;
        if (pend_valid) {
            const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pend >> 32)) << 32) |
                                  (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pend & 0xffffffffull));
            const uint64_t it = base + (uint64_t)lane;
            if (pend_atomic) { blk_next = base + (uint64_t)NA; blk_left = grab - 1; }
            if (base >= n_iter) { exhausted = true; blk_left = 0; }
            if (lane < NA && it < n_iter) {
                if (pair_order) {
                    // consecutive positions 2m, 2m+1 (the two slots of a lane group): reads 2q, 2q+1 against the same reference
                    const uint64_t blk = it / (2u * (uint64_t)A.n_refs), i = it - blk * 2u * (uint64_t)A.n_refs;
                    const uint64_t rd = 2u * blk + (i & 1u);
                    if (rd < n_reads_po) { mA_valid = 1; mA_task = (unsigned)(rd * (uint64_t)A.n_refs + (i >> 1)); }
                } else { mA_valid = 1; mA_task = A.task_list ? A.task_list[it] : (unsigned)it; }
            }
        }
        pend = 0; pend_valid = false;
        if (!exhausted) {
            if (blk_left > 0) { pend = blk_next; blk_next += (uint64_t)NA; --blk_left; pend_atomic = false; }
            else { if (lane == 0) pend = atomicAdd(A.work_counter, (unsigned long long)(NA * grab)); pend_atomic = true; }
            pend_valid = true;
        }
        __syncthreads();
        if (!have_group) continue;

        // ---- band of every alignment; the tables its lanes read; the wave-uniform loop limits.  Lane s does the arithmetic of slot s (sixteen
        //      slots one after the other, each on the scalar unit with its own wait for the reference's record, were a sixth of the score-only
        //      launch's instructions); three reductions over the first NA lanes give the loop limits
        bool any_ok = false;
        int gA = 0, gC = 0x7fffffff, g_end = 0;
        {
            int* T = sTab + (lane < NA ? lane : 0) * C2X_INTS;
            const int Li = T[C2X_LI], Lj = T[C2X_LJ];
            bool ok = false;
            int D = 0, d0 = 0, cb = 0, minsc = 0, lastpos = 0, rowBase = C2_DIAG_ROW_PAD;   // (idle slot: row 0 of the buffer's first table)
            int gA_s = 0, gC_s = 0x7fffffff, ge_s = 0;
            if (lane < NA && T[C2X_VALID] && T[C2X_STATUS] == 0) {
                const c2_dev_ref* rf = A.refs + T[C2X_REF];
                const c2_diag_row* drows = rf->diag_rows;
                D = Li - Lj;
                d0 = ((D - BANDW + 3) >> 1) & ~1;              // even; band = d0 .. d0 + BANDW - 1, the first diagonals outside it (d0 - 1, d0 + BANDW) as
                                                               // symmetric about D / 2 as an even d0 allows: the two sides of c2_outside_band_bound are then equal
                cb = (go > ge ? go : ge) + rf->gap_incentive_max;         // the most one gap base can add to a score
                lastpos = rf->gap_incentive_last_pos;
                ok = T[C2X_PACKED] && drows != nullptr && cb < 0 && d0 <= 0 && d0 + BANDW - 1 >= 0 &&
                     D >= d0 && D <= d0 + BANDW - 1;
                if (PK && ok) ok = rf->pk_ok != 0;              // the reference must be admitted to the int16 fill (c2_pk_eligible); pairs were formed by the staging
                if (SCORE && ok) ok = Li == Lj;                  // (the score-only fill finishes nothing but the main diagonal)
                if (ok) {
                    minsc = (int)(uint32_t)((uint64_t)(int64_t)go * (uint64_t)Lj * (uint64_t)Li);
                    rowBase = (int)(drows - A.diag_base);
                    const int max_start = (d0 + BANDW - 1 > -d0 ? d0 + BANDW - 1 : -d0) + 2;
                    gA_s = ((max_start + 1) >> 1) >> 2;                    // groups 0..gA contain lanes that have not started
                    gC_s = ((2 * Lj + d0) >> 1) >> 2;                      // first group in which some lane is on the last column
                    ge_s = ((Li + Lj) >> 1) >> 2;                          // group of the cell (Li, Lj)
                }
            }
            if (lane < NA) {
                T[C2X_OK] = ok ? 1 : 0; T[C2X_D] = D; T[C2X_D0] = d0; T[C2X_CB] = cb; T[C2X_MINSC] = minsc; T[C2X_ROWBASE] = rowBase; T[C2X_LASTPOS] = lastpos;
                T[C2X_BAND_LI] = ok ? Li : 0; T[C2X_BAND_LJ] = ok ? Lj : 0;
            }
            any_ok = __ballot(ok) != 0ull;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {                 // (NA <= 16; the lanes behind hold the neutral values)
                gA_s = c2_imax(gA_s, __shfl_xor(gA_s, m)); ge_s = c2_imax(ge_s, __shfl_xor(ge_s, m));
                const int o = __shfl_xor(gC_s, m); gC_s = gC_s < o ? gC_s : o;
            }
            gA = __builtin_amdgcn_readfirstlane(gA_s); gC = __builtin_amdgcn_readfirstlane(gC_s); g_end = __builtin_amdgcn_readfirstlane(ge_s);
        }
        __syncthreads();
        c2_phase_mark<0>(A.phase_cycles, PH);

        int Hcap = C2_DIAG_NEG;
        c2_gapfree GF; GF.acc = 0; GF.cap = 0xffffffffu;
        c2_pk_cap CAP; CAP.H = 0u; CAP.gf = 0u;                    // (H = 0 in both halves: below every bound; gf = 0: not gap-free)
        if (A.reserved & 4) g_end >>= 1;                           // (debug knob C2_DEBUG_HALF_FILL: what half of the fill costs; nothing is certified then)
        if (PK && any_ok) {
            const int* T = sTab + slot * C2X_INTS;                 // the lane group's first alignment (the second one, if any, has the same geometry)
            const int vLi = T[C2X_BAND_LI], vLj = T[C2X_BAND_LJ], vd0 = T[C2X_D0], vg0 = T[C2X_G0];
            const int vrow = T[C2X_ROWBASE], vcode = (int)(P.pcodes0 + (uint32_t)grp * P.pcodes_bytes) + C2_DIAG_CODE_PAD;
            C2_LANES_ACTIVE_BEGIN(sl != NL && grp < NG)
            const int hE = (vd0 >> 1) + sl;
            const int dE = 2 * hE, dO = dE + 1;
            c2_pk_state S;
            S.acc = 0u; S.gf = 0xffffffffu;
            {
                // boundary cells (pyx:153-176) with the bias; the sentinel min_score is the number 0 here
                // (beta: the boundary cell of diagonal d lies on anti-diagonal |d|)
                const int bE = ((dE == 0) ? 0 : (ge + beta) * (dE > 0 ? dE : -dE) + vg0) + PKB;
                const int mE = (dE == 0) ? PKB : 0, iE = (dE < 0) ? bE : 0, jE = (dE > 0) ? bE : 0;
                S.ME = c2_pk_dup(mE); S.IE = c2_pk_dup(iE); S.JE = c2_pk_dup(jE); S.HE = c2_pk_dup(c2_imax(c2_imax(mE, iE), jE));
                const int bO = (ge + beta) * (dO > 0 ? dO : -dO) + vg0 + PKB;
                const int iO = (dO < 0) ? bO : 0, jO = (dO > 0) ? bO : 0;
                S.MO = 0u; S.IO = c2_pk_dup(iO); S.JO = c2_pk_dup(jO); S.HO = c2_pk_dup(c2_imax(iO, jO));
            }
            c2_diagx_lane L;
            L.rowOff = (unsigned)((vrow + hE) * (int)sizeof(c2_diag_row));
            L.rowMax = (unsigned)((vrow + vLi + 1 + C2_DIAG_ROW_PAD - 5) * (int)sizeof(c2_diag_row));
            L.colOff = (unsigned)(vcode - hE);
            L.colMax = (unsigned)(vcode + vLj + 2);
            L.kLast = vLj + hE;
            L.kCap = (vLi + vLj) >> 1; L.capOdd = ((vLi + vLj) & 1) != 0;
            L.startE = (dE > 0 ? dE : -dE) + 2; L.startO = (dO > 0 ? dO : -dO) + 2;
            const c2_diag_row* rows = A.diagpk_base;
            c2_diag_row RA[5], RB[5];
            int CA[4], CB[4];
            c2_diagx_fetch<true>(0, L, rows, c2_smem, RA, CA);
            unsigned* wordsA = gWords + slot * slotWords + sl;
            unsigned* wordsB = wordsA + slotWords;
            int g = 0;
            const int gA_stop = gA < g_end ? gA : g_end;
            unsigned ge2 = c2_pk_dup(ge + beta);
            const unsigned lutBase = P.pairlut;
            if (gC <= gA_stop) {
                c2_pk_groups<true, true, ROWDPP, ADD32, SCORE>(S, g, gA_stop, L, ge2, CAP, RA, CA, RB, CB, rows, c2_smem, lutBase, wordsA, wordsB, LPW);
            } else {
                c2_pk_groups<true, false, ROWDPP, ADD32, SCORE>(S, g, gA_stop, L, ge2, CAP, RA, CA, RB, CB, rows, c2_smem, lutBase, wordsA, wordsB, LPW);
                c2_pk_groups<false, false, ROWDPP, ADD32, SCORE>(S, g, (gC - 1 < g_end ? gC - 1 : g_end), L, ge2, CAP, RA, CA, RB, CB, rows, c2_smem, lutBase, wordsA, wordsB, LPW);
            }
            c2_pk_groups<false, true, ROWDPP, ADD32, SCORE>(S, g, g_end, L, ge2, CAP, RA, CA, RB, CB, rows, c2_smem, lutBase, wordsA, wordsB, LPW);
            C2_LANES_ACTIVE_END()
        }
        if (!PK && any_ok) {
            const int* T = sTab + slot * C2X_INTS;                 // this lane's alignment
            const int vLi = T[C2X_BAND_LI], vLj = T[C2X_BAND_LJ], vd0 = T[C2X_D0], vg0 = T[C2X_G0], vmin = T[C2X_MINSC];
            const int vrow = T[C2X_ROWBASE], vcode = (int)(P.slot0 + (uint32_t)slot * P.slot_bytes + P.codes) + C2_DIAG_CODE_PAD;
            C2_LANES_ACTIVE_BEGIN(sl != NL && grp < NG)            // (grp >= NG: the lanes 64 / NG does not use up, e.g. 60..63 of five groups of 12)
            // ---- per-lane diagonals and their boundary cells (pyx:153-176), as in c2_align_diag_kernel
            const int hE = (vd0 >> 1) + sl;                    // dE = 2*hE, dO = 2*hE + 1
            const int dE = 2 * hE, dO = dE + 1;
            c2_diag_state S;
            S.bits = 0;
            {
                const int ms = vmin + C2_DIAG_BIAS;
                const int bE = ((dE == 0) ? 0 : ge * (dE > 0 ? dE : -dE) + vg0) + C2_DIAG_BIAS;
                S.ME = (dE == 0) ? C2_DIAG_BIAS : ms;
                S.IE = (dE < 0) ? bE : ms;
                S.JE = (dE > 0) ? bE : ms;
                S.HE = c2_imax(c2_imax(S.ME, S.IE), S.JE);
                const int bO = ge * (dO > 0 ? dO : -dO) + vg0 + C2_DIAG_BIAS;   // dO is odd, never 0
                S.MO = ms;
                S.IO = (dO < 0) ? bO : ms;
                S.JO = (dO > 0) ? bO : ms;
                S.HO = c2_imax(c2_imax(S.MO, S.IO), S.JO);
            }
            c2_diagx_lane L;
            L.rowOff = (unsigned)((vrow + hE) * (int)sizeof(c2_diag_row));
            L.rowMax = (unsigned)((vrow + vLi + 1 + C2_DIAG_ROW_PAD - 5) * (int)sizeof(c2_diag_row));
            L.colOff = (unsigned)(vcode - hE);
            L.colMax = (unsigned)(vcode + vLj + 2);
            L.kLast = vLj + hE;
            L.kCap = (vLi + vLj) >> 1; L.capOdd = ((vLi + vLj) & 1) != 0;
            L.startE = (dE > 0 ? dE : -dE) + 2; L.startO = (dO > 0 ? dO : -dO) + 2;
            const c2_diag_row* rows = A.diag_base;
            c2_diag_row RA[5], RB[5];
            int CA[4], CB[4];
            c2_diagx_fetch<true>(0, L, rows, c2_smem, RA, CA);
            unsigned* myWords = gWords + slot * slotWords + sl;
            int g = 0;
            const int gA_stop = gA < g_end ? gA : g_end;
            int geV = ge;                                          // gap_extend in a VGPR (second source of a DPP add)
            C2_KEEP_IN_VGPR(geV);
            if (gC <= gA_stop) {
                c2_diagx_groups<true, true>(S, g, gA_stop, L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, LPW);
            } else {
                c2_diagx_groups<true, false>(S, g, gA_stop, L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, LPW);
                c2_diagx_groups<false, false>(S, g, (gC - 1 < g_end ? gC - 1 : g_end), L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, LPW);
            }
            c2_diagx_groups<false, true>(S, g, g_end, L, geV, Hcap, GF, RA, CA, RB, CB, rows, c2_smem, myWords, LPW);
            C2_LANES_ACTIVE_END()
        }
        if (!SCORE) {
            __syncthreads();                                       // (waits for the plane stores)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drop this CU's stale L1 lines of the plane before reading it back
        }                                                          // (the score-only fill stores nothing: its row table stays in L1 from one group of alignments to the next)
        c2_phase_mark<1>(A.phase_cycles, PH);

        // ---- per alignment: optimality certificate (see c2_align_diag_kernel), then one of three ends:
        //        gap-free   the main-diagonal lane's c2_gapfree word is 0: the strings are the read and the reference, no pointer
        //                   word is ever read back (most reads of an amplicon run)
        //        traced     the alignment's pointer words come back from the scratch plane into LDS, traceback, output
        //        handed on  no certificate: the task goes to the next launch's list
        //      First pass: the decision of every slot (wave-uniform bit masks) -- the words of the NEXT traced slot have to be
        //      requested before the current one is written out (a load issued after those stores would wait for them: one
        //      vmcnt for loads and stores), so the traced slots must be known before the first one is handled.
        //      (Lane s decides for slot s; ballots make the masks.)
        unsigned m_valid, m_full, m_gapfree, m_trace;
        {
            const int s = lane < NA ? lane : 0;
            const int* T = sTab + s * C2X_INTS;
            const bool valid = lane < NA && T[C2X_VALID] != 0;
            const bool live = valid && T[C2X_STATUS] == 0;
            const bool okb = live && T[C2X_OK] != 0;
            const int Li = T[C2X_LI], Lj = T[C2X_LJ];
            const int D = T[C2X_D], d0 = T[C2X_D0], cb = T[C2X_CB];
            const int lane_end = ((PK ? (s >> 1) : s) * LPA + ((D - d0) >> 1)) & 63;
            int Hend;
            bool gapfree;
            if (PK) {
                const int half = 16 * (s & 1);
                Hend = (int)(int16_t)(((unsigned)__shfl((int)CAP.H, lane_end) >> half) & 0xffffu) - PKB - beta * (Li + Lj);
                const unsigned gfw = (unsigned)__shfl((int)CAP.gf, lane_end) >> half;
                gapfree = SCORE ? ((gfw & 0x8000u) != 0u) : ((gfw & 0x1111u) == 0x1111u);   // the E cells (cells 0 and 2 of a word): bits 0 and 4 of both bytes; SCORE: the AND of their sign bits
            } else {
                Hend = __shfl(Hcap, lane_end);
                gapfree = __shfl((int)GF.cap, lane_end) == 0;
            }
            const int dhi1 = d0 + BANDW, dlo1 = d0 - 1;               // first diagonals outside the band
            const int U = c2_outside_band_bound(A.max_score, Li, Lj, D, dhi1, dlo1, cb, go, ge, T[C2X_LASTPOS]);
            const bool cert = okb && Hend > U && !(A.reserved & 2);    // (debug knob C2_DEBUG_SKIP_EPILOGUE: certified, nothing written)
            const bool gf = cert && Li == Lj && gapfree;
            const bool full = (live && !okb) || (okb && !(Hend > U)) || (SCORE && cert && !gf);   // (SCORE: certified, but not the main diagonal: the launch with the pointer words takes it)
            m_valid = (unsigned)__ballot(valid); m_full = (unsigned)__ballot(full);
            m_gapfree = (unsigned)__ballot(gf); m_trace = (unsigned)__ballot(!SCORE && cert && !gf);
        }
#pragma nounroll
        for (int s = 0; s < NA; ++s) {
            if (!((m_valid >> s) & 1u)) continue;
            if (!SCORE && ((m_trace >> s) & 1u)) continue;           // traced: all of them at once, below (c2_group_epilogue)
            const int* T = sTab + s * C2X_INTS;
            const int tv = c2_tab_load(T, lane);
            const uint64_t task = (uint64_t)(unsigned)C2_TF(tv, C2X_TASK_LO) | ((uint64_t)(unsigned)C2_TF(tv, C2X_TASK_HI) << 32);
            const int Li = C2_TF(tv, C2X_LI), Lj = C2_TF(tv, C2X_LJ), g0 = C2_TF(tv, C2X_G0);
            int status = C2_TF(tv, C2X_STATUS);
            c2_aln_record rec;
            c2_clear_record(rec, C2_TF(tv, C2X_RC), C2_TF(tv, C2X_REF));
            bool need_full = (m_full >> s) & 1u;
            if ((m_gapfree >> s) & 1u) {
                if (rows_aligned && Li <= 256) c2_emit_gapless4(A, wg_of(s), win_of(s), task, Li, lane, rec);
                else c2_emit_gapless(A, wg_of(s), task, Li, lane, rec);
            }
            if (need_full) {
                status |= C2_STATUS_NEED_FULL;
                const bool to_unpaired = PK && A.un_list && C2_TF(tv, C2X_UNPAIRED);
                if (lane == 0) {
                    if (to_unpaired) { const unsigned q = atomicAdd(A.un_count, 1u); A.un_list[q] = (uint32_t)task; }   // same band, 32-bit kernel
                    else { const unsigned q = atomicAdd(A.fb_count, 1u); A.fb_list[q] = (uint32_t)task; }
                }
            }
            rec.status = (uint8_t)status;
            if (lane == 0) A.records[task] = rec;
            c2_phase_mark<3>(A.phase_cycles, PH);
        }
        if constexpr (!SCORE) {
            if (m_trace && !(A.reserved & 32)) {                       // (32: debug knob C2_DEBUG_SKIP_TRACE)
                __syncthreads();
                c2_group_epilogue<NA, PK, LPW, NL>(A, P, lane, m_trace, 0, sTab, gWords, slotWords, rows_aligned);
                if constexpr (NA > 8) { if (m_trace >> 8) c2_group_epilogue<NA, PK, LPW, NL>(A, P, lane, m_trace, 8, sTab, gWords, slotWords, rows_aligned); }
                c2_phase_mark<2>(A.phase_cycles, PH);
            }
        }
    }
    c2_phase_flush(A.phase_cycles, PH, lane);
}

template <int NA>
__global__ __launch_bounds__(64, 3) void c2_align_diagx_kernel(c2_align_args A) { c2_diagx_body<NA, false>(A); }

// NA alignments per wavefront, two per lane group, int16 DP values (see "Packed fill")
template <int NA, bool ADD32 = false>
__global__ __launch_bounds__(64, 3) void c2_align_diagp_kernel(c2_align_args A) { c2_diagx_body<NA, true, ADD32>(A); }

// The same packed fill WITHOUT pointer bits (c2_pk_pair<.., SCORE>): per cell pair 4 instructions for the gap-free predicate instead of 21 for eight
// pointer bits, no pointer word stored.  It can finish exactly one kind of alignment -- read and reference of one length, optimal path = the main
// diagonal, certificate holds -- which is most reads of an amplicon run; everything else it hands on, untouched, to the launch that keeps pointers.
// The host puts it in front of the first band tier over the tasks c2_align_partition_kernel expects to be of that kind.
template <int NA, bool ADD32 = false>
__global__ __launch_bounds__(64, 3) void c2_align_diags_kernel(c2_align_args A) { c2_diagx_body<NA, true, ADD32, true>(A); }

// Hardware self-test of the cross-lane primitives the DP relies on (wave_shr:1 with `old` kept in lane 0).
__global__ __launch_bounds__(64) void c2_selftest_kernel(int* out)
{
    const int lane = threadIdx.x;
    out[lane] = c2_shr1(-7, lane * 3 + 1);                      // expect lane0=-7, lane n = 3(n-1)+1
    out[64 + lane] = __builtin_amdgcn_readlane(lane * 5, 17);   // expect 85 everywhere
    const unsigned long long m = __ballot(lane % 3 == 0);
    out[128 + lane] = __popcll(m) + (lane == 0 ? __builtin_ctzll(~m) : 0);
    // a lane that is switched off in EXEC is an invalid DPP source: its neighbour keeps `old` (what isolates the lane
    // groups of c2_align_diagx_kernel from each other).  expect: lanes 31 -> -99, 32 (shr) / 30 (shl) -> -7
    int r = -99, l = -99;
    C2_LANES_ACTIVE_BEGIN(lane != 31)
        r = c2_shr1(-7, lane * 3 + 1);
        l = c2_shl1(-7, lane * 3 + 1);
    C2_LANES_ACTIVE_END()
    out[192 + lane] = r;
    out[256 + lane] = l;
    // the form the diagonal kernels use: bound_ctrl set, folded into an add (v_add_u32_dpp) -- a lane without a source
    // (wavefront end, or source switched off in EXEC) must read 0.  expect 1000 in lanes 0 and 32 (shr) / 63 and 30 (shl)
    int rz = -99, lz = -99;
    int k1000 = 1000;
    C2_KEEP_IN_VGPR(k1000);
    C2_LANES_ACTIVE_BEGIN(lane != 31)
        rz = c2_shr1z(lane * 3 + 1) + k1000;
        lz = c2_shl1z(lane * 3 + 1) + k1000;
    C2_LANES_ACTIVE_END()
    out[320 + lane] = rz;
    out[384 + lane] = lz;
}

// Which launch of the chain should see a task first?  One lane per task, two cheap looks at the read against its reference (forward strand,
// reference admitted to the packed fill; everything else goes where it went before: to the first band tier):
//   class 0  read and reference of one length (32 .. 256 bases) whose LAST 32 columns differ in at most `max_mismatch` places: a read with an
//            indel is shifted against the reference behind the indel and differs there in most columns, a read without one in hardly any
//            -> c2_align_diags_kernel (the fill without pointer bits; it can finish a main-diagonal alignment only)
//   otherwise the DIAGONAL the middle of the read lies on: the 32 bases from column Lj/2 + 16 on, as a 64-bit word of 2-bit codes, against every
//            window of the reference within `max_shift` of the same place (one shift + one byte per window); the best window with at most
//            `probe_max_mismatch` differing bases puts the path on diagonal s there, so a band has to hold the diagonals 0 (start), s and
//            D = Li - Lj (end), with `margin` diagonals to spare on either side (the certificate needs them: a path that leaves the band must
//            be worse than the one found, and every mismatch of the read brings the two closer):
//   class 1  ... fit a band of bandw[0] diagonals (14: c2_align_diagp_kernel<16>, sixteen alignments per wavefront)
//   class 2  ... bandw[1] (32: the first band tier)     -- also: no window found, reverse strand, reference not admitted
//   class 3  ... bandw[2] (40: round 5's tier between them -- three lane groups of 21 lanes; reads with an overhang at both ends, §DESIGN 3.1)
//   class 4  ... bandw[3] (62)         class 5  ... bandw[4] (126 / anything wider: the last band tier)
//   (bandw[k] = 0: the chain has no such launch; the next wider one that exists takes the task.)
// A prediction only: every launch verifies what it finishes (certificate) and hands on what it cannot, so a wrong class costs that alignment a
// second fill and nothing else -- but a read with a 20-base deletion no longer pays for a fill in a band that cannot hold it.
//   class 6  (round 5, `direct_full`) the read matches the reference NOWHERE: neither the middle window nor the windows a quarter and three quarters
//            into the read find a place with at most `probe_max_mismatch` differing bases.  No band will certify such a read (an unrelated
//            sequence, a chimera): it goes straight to the list of the LAST launch, the full matrix, instead of being filled and handed on by
//            every band tier on its way there (5 + 9 + 19 ns before the 71 ns it cannot avoid).
// When the middle window finds nothing (a breakpoint inside it), the two other windows place the path: the band then has to hold every diagonal found.
// Lists in slot order inside a chunk of C2_PART_CHUNK tasks, the chunks in the order of their atomics (one per chunk and class).  Slot order is task
// order -- unless the chunk's reads differ in LENGTH (round 5): then the chunk's slots are first ordered by read length (a counting sort in LDS),
// because the packed kernels put two alignments into one lane group only if they share reference AND read length, and neighbours of a ragged input
// in task order almost never do (reads of lengths U[200, 250]: everything went to the 32-bit kernels, at half the rate).
#define C2_PART_CLASSES 7                          // 0 score-only, 1 .. 5 the band launches by width, 6 the full-matrix launch
struct c2_partition_args {
    c2_align_args A;
    uint32_t* list[C2_PART_CLASSES]; uint32_t* count[C2_PART_CLASSES];      // per class: the launch's task list and its length (classes may share a list)
    uint32_t* class_count;                      // [C2_PART_CLASSES] tasks per class (statistics)
    int32_t bandw[5];                           // diagonals of the launches behind classes 1 .. 5 (14, 32, 40, 62, 128), 0: the chain has no such launch
    int32_t max_mismatch, probe_max_mismatch, margin, max_shift;
    int32_t direct_full, sort_by_length;        // the last class exists; order a ragged chunk's slots by read length
    int32_t check_cut;                          // class 0 also looks at the 32 columns around the cut site (batches with several candidate amplicons)
    int32_t exact_copies;                       // the output rows can be written as dwords: a class-0 read that EQUALS its reference is finished here (class_count[7])
    int32_t route_cert;                         // c2_part_probe: a band is chosen only if its launch can be expected to certify the alignment (round 6; 0: the geometric test alone)
};

#ifndef C2_PART_CHUNK
#define C2_PART_CHUNK 4096                         // tasks per workgroup and set of atomics (one per task and list serialises in L2: 28 ms for 10 M tasks)
#endif
#define C2_PART_LEN_BINS 512                       // read lengths 0 .. 510 have a bin of their own, longer reads share the last
// flags | per-wavefront scan words | the slots to probe, later the slots in length order (uint16 each) | the slots' read lengths (uint16) | length histogram
// | the reads' byte offsets relative to the chunk's first read (uint32 each)
// | the slots' reference ids (uint16; batches whose reads are tagged with their reference)
#define C2_PART_LDS (C2_PART_CHUNK + 160 + 2 * C2_PART_CHUNK + 2 * C2_PART_CHUNK + 4 * C2_PART_LEN_BINS + 4 * C2_PART_CHUNK + 2 * C2_PART_CHUNK)

// 32 bases from p on as 2-bit codes ((c >> 1) & 3: A 0, C 1, T 2, G 3; anything else aliases one of them -- this is a predictor), base k in bits 2k+1 .. 2k
__device__ __forceinline__ uint64_t c2_code32(const uint8_t* p) {
    uint64_t code = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t w;
        __builtin_memcpy(&w, p + 4 * q, 4);
        const uint32_t t = (w >> 1) & 0x03030303u;
        code |= (uint64_t)((t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xffu) << (8 * q);
    }
    return code;
}

__device__ __forceinline__ bool c2_band_holds(const int bandw, const int D, const int lo, const int hi, const int margin) {
    const int d0 = ((D - bandw + 3) >> 1) & ~1;                  // (the kernels' own placement: c2_diagx_body)
    return bandw > 0 && lo - d0 >= margin && d0 + bandw - 1 - hi >= margin;
}

// a task's read and reference as the partition needs them
struct c2_part_task { const uint8_t* rd; const uint8_t* f; const c2_dev_ref* ref; uint64_t off; int Li, Lj, rc, pk_ok, cut, ref_id, diag_kmax; };
__device__ __forceinline__ c2_part_task c2_part_load(const c2_align_args& A, const uint64_t task) {
    uint64_t read_id; int ref_id;
    if (A.all_refs) { read_id = task / (uint64_t)A.n_refs; ref_id = (int)(task % (uint64_t)A.n_refs); }
    else            { read_id = task; ref_id = A.ref_ids ? (int)A.ref_ids[task] : 0; }
    c2_part_task t;
    t.rc = A.strands ? (int)A.strands[task] : 0;
    const uint64_t off = A.offsets[read_id];
    t.Lj = (int)(A.offsets[read_id + 1] - off);
    t.off = off;
    const c2_dev_ref* rf = A.refs + ref_id;
    t.Li = rf->len; t.pk_ok = rf->pk_ok; t.rd = A.reads + off; t.f = rf->seq; t.cut = rf->first_incentive_pos;
    t.ref_id = ref_id; t.diag_kmax = rf->diag_kmax; t.ref = rf;
    return t;
}
// how a chunk's slots map to tasks (see the kernel): reference-major for an all-references batch of several references
struct c2_part_walk { uint32_t k, rpc, chunk_tasks; };
__host__ __device__ __forceinline__ uint32_t c2_part_chunk_tasks(const int all_refs, const int n_refs) {
    return (all_refs && n_refs > 1) ? (uint32_t)(C2_PART_CHUNK / n_refs) * (uint32_t)n_refs : (uint32_t)C2_PART_CHUNK;
}
__device__ __forceinline__ c2_part_walk c2_part_walk_of(const c2_align_args& A) {
    c2_part_walk w;
    w.k = (A.all_refs && A.n_refs > 1) ? (uint32_t)A.n_refs : 1u;
    w.rpc = (uint32_t)C2_PART_CHUNK / w.k;
    w.chunk_tasks = w.rpc * w.k;
    return w;
}
// -> the task in slot `slot` of the chunk that starts at task `chunk`, or ~0 (no task there)
__device__ __forceinline__ uint64_t c2_part_task_of(const c2_part_walk& w, const c2_align_args& A, const uint64_t chunk, const int slot) {
    if (w.k == 1u) return chunk + (uint64_t)slot;
    const uint32_t ref = (uint32_t)slot / w.rpc, rd = (uint32_t)slot - ref * w.rpc;
    if (ref >= w.k) return ~0ull;
    const uint64_t task = chunk + (uint64_t)rd * w.k + ref;
    return task < A.n_tasks ? task : ~0ull;
}
__device__ __forceinline__ bool c2_part_probes(const c2_partition_args& P, const c2_part_task& t) {
    return P.max_shift > 0 && t.Lj >= 96 && t.Li >= 32;
}

// the diagonal the 32 bases of the read from column p on lie on: the window of the reference within max_shift of the same place that differs
// from them in the fewest bases -> differing bases of that window (64: no window to look at), its shift in s.  The windows are visited from
// shift 0 OUTWARDS (+1, -1, +2, -2, ...: two rolling words, one base shifted in per step each) and the walk ends at the first window that
// equals the read's bases: a read's shift is a few bases (an indel, an overhang), so it is found after a dozen windows instead of 129
// (the probe was 2.0 of 37 ms on reads that all need it).  Ties between imperfect windows: the one nearer to shift 0.
__device__ __forceinline__ int c2_part_window(const c2_partition_args& P, const c2_part_task& t, const int p, int& best_s) {
    const int s_lo = -P.max_shift > -p ? -P.max_shift : -p;
    const int s_hi = P.max_shift < t.Li - 32 - p ? P.max_shift : t.Li - 32 - p;
    best_s = 0;
    if (s_lo > s_hi || p < 0 || p + 32 > t.Lj) return 64;
    const uint64_t rcode = c2_code32(t.rd + p);
    const uint64_t m55 = 0x5555555555555555ull;
    int s0 = 0;                                                     // the shift the walk starts at: 0 where the reference has a window there
    if (s0 < s_lo) s0 = s_lo;
    if (s0 > s_hi) s0 = s_hi;
    const int reach = (s_hi - s0) > (s0 - s_lo) ? (s_hi - s0) : (s0 - s_lo);
    int best_mm;
    if (const uint32_t* s2 = t.ref->seq2) {
        // Round 6: the reference as 2-bit codes, 16 bases per word (c2_dev_ref.seq2).  A step of the walk takes its base out of a word held in a
        // register and loads the next word a whole word ahead -- a byte of t.f per step was a load the next step waited for (an L1 hit is some
        // hundred cycles here, and the wavefront walks as far as its farthest lane: 44 % of the partition's time on the headline batch).
        const int i0 = p + s0;
        uint64_t up, dn;
        {
            const int d = i0 >> 4, sh = (i0 & 15) * 2;
            const uint64_t lo = (uint64_t)s2[d] | ((uint64_t)s2[d + 1] << 32);
            up = sh ? (lo >> sh) | ((uint64_t)s2[d + 2] << (64 - sh)) : lo;
            dn = up;
        }
        {
            const uint64_t x = up ^ rcode;
            best_mm = __popcll((x | (x >> 1)) & m55);
            best_s = s0;
        }
        int bu = i0 + 32, bd = i0 - 1;                              // the base that enters at the top / at the bottom next
        uint32_t wu = s2[bu >> 4], wu_n = s2[(bu >> 4) + 1], wd = s2[bd >> 4], wd_n = s2[(bd >> 4) - 1];
        for (int k = 1; k <= reach && best_mm > 0; ++k) {
            if (s0 + k <= s_hi) {
                up = (up >> 2) | ((uint64_t)((wu >> ((bu & 15) * 2)) & 3u) << 62);
                ++bu;
                if (!(bu & 15)) { wu = wu_n; wu_n = s2[(bu >> 4) + 1]; }
                const uint64_t x = up ^ rcode;
                const int m2 = __popcll((x | (x >> 1)) & m55);
                if (m2 < best_mm) { best_mm = m2; best_s = s0 + k; }
            }
            if (s0 - k >= s_lo) {
                dn = (dn << 2) | (uint64_t)((wd >> ((bd & 15) * 2)) & 3u);
                --bd;
                if ((bd & 15) == 15) { wd = wd_n; wd_n = s2[(bd >> 4) - 1]; }
                const uint64_t x = dn ^ rcode;
                const int m2 = __popcll((x | (x >> 1)) & m55);
                if (m2 < best_mm) { best_mm = m2; best_s = s0 - k; }
            }
        }
        return best_mm;
    }
    uint64_t up = c2_code32(t.f + p + s0), dn = up;                 // the windows at s0 + k and s0 - k
    {
        const uint64_t x = up ^ rcode;
        best_mm = __popcll((x | (x >> 1)) & m55);
        best_s = s0;
    }
    for (int k = 1; k <= reach && best_mm > 0; ++k) {
        if (s0 + k <= s_hi) {                                       // the base that enters at the top: reference position p + s0 + k + 31
            up = (up >> 2) | ((uint64_t)(((unsigned)t.f[p + s0 + k + 31] >> 1) & 3u) << 62);
            const uint64_t x = up ^ rcode;
            const int m2 = __popcll((x | (x >> 1)) & m55);
            if (m2 < best_mm) { best_mm = m2; best_s = s0 + k; }
        }
        if (s0 - k >= s_lo) {                                       // the base that enters at the bottom: reference position p + s0 - k
            dn = (dn << 2) | (uint64_t)(((unsigned)t.f[p + s0 - k] >> 1) & 3u);
            const uint64_t x = dn ^ rcode;
            const int m2 = __popcll((x | (x >> 1)) & m55);
            if (m2 < best_mm) { best_mm = m2; best_s = s0 - k; }
        }
    }
    return best_mm;
}

// the class of a task by the diagonals its read lies on (see above); `widest`: the class that takes what no band holds
// Which launch should see a probed task first.  Round 6: the band must not only HOLD the path's diagonals (c2_band_holds) -- its launch must be able to
// CERTIFY the alignment (c2_outside_band_bound, the test the kernels apply after the fill), or the fill is spent for nothing and the task is filled again
// by the next tier (measured on FANC-shaped reads, round 5: the 40-diagonal tier handed 17 % of its tasks on, the 62-diagonal one 21 %, the 126-diagonal
// one 24 %).  The path is modelled from two windows -- a quarter into the read and behind its middle --: a leading run to the first window's diagonal
// s1, one run of s2 - s1 in between, a trailing run to D; end runs cost gap_extend per base (pyx:153-176, 234-317), the one in between gap_open + gap_extend
// per base; every other column is taken as a match but for an allowance of two mismatches.  A prediction only: every launch verifies what it finishes.
__device__ __forceinline__ int c2_part_probe(const c2_partition_args& P, const c2_part_task& t, const int widest) {
    const c2_align_args& A = P.A;
    const int D = t.Li - t.Lj;
    int s = 0;
    int lo = D < 0 ? D : 0, hi = D > 0 ? D : 0;
    int s1 = 0, s2 = 0;
    const int mm = c2_part_window(P, t, (t.Lj >> 1) + 16, s);
    bool have1 = false, have2 = false;
    if (mm <= P.probe_max_mismatch) { s2 = s; have2 = true; }
    int m1 = 64, m2 = 64;
    bool looked1 = false;
    if (!have2 || (P.route_cert && D != 0)) {
        looked1 = true;                        // (a read as long as its reference: the front part is taken to lie on the main diagonal -- the second look cost the headline batch 0.5 ms)
        m1 = c2_part_window(P, t, t.Lj >> 2, s1);
        have1 = m1 <= P.probe_max_mismatch;
    }
    if (!have2) {
        // nothing behind the middle (a breakpoint inside the window, a noisy stretch -- or a read that is not this amplicon's at all): three quarters into the read
        m2 = c2_part_window(P, t, ((3 * t.Lj) >> 2) - 16, s2);
        have2 = m2 <= P.probe_max_mismatch;
        if (!have1 && !have2) return (P.direct_full && mm < 64 && m1 < 64 && m2 < 64) ? C2_PART_CLASSES - 1 : 2;      // (a window that could not be looked at says nothing)
    }
    if (have1) { lo = s1 < lo ? s1 : lo; hi = s1 > hi ? s1 : hi; }
    if (have2) { lo = s2 < lo ? s2 : lo; hi = s2 > hi ? s2 : hi; }
    const int go = A.gap_open, ge = A.gap_extend;
    const int cb = (go > ge ? go : ge) + t.ref->gap_incentive_max;
    if (!(P.route_cert && cb < 0 && A.max_score > 0)) {              // the geometric test alone (before round 6; C2_NO_ROUTE_CERT=1)
        int cls = widest;
        for (int k = 4; k >= 0; --k) if (c2_band_holds(P.bandw[k], D, lo, hi, P.margin)) cls = k + 1;
        return cls;
    }
    // the model path's score without a mismatch ...
    const int a1 = have1 ? s1 : ((looked1 && have2) ? s2 : 0), a2 = have2 ? s2 : a1;         // the diagonal in front of / behind the middle (not looked at in front: the main diagonal)
    const int g_lead = a1, g_mid = a2 - a1, g_trail = D - a2;
    const int nv = (g_lead > 0 ? g_lead : 0) + (g_mid > 0 ? g_mid : 0) + (g_trail > 0 ? g_trail : 0);
    const int h0 = t.ref->gap_incentive_max + A.max_score * (t.Li - nv) + ge * ((g_lead < 0 ? -g_lead : g_lead) + (g_trail < 0 ? -g_trail : g_trail)) +
                   (g_mid != 0 ? go + ge * (g_mid < 0 ? -g_mid : g_mid) : 0);
    // ... and the launch to start with: the one with the smallest EXPECTED cost, where a launch that cannot certify the alignment hands it to the next
    // wider one.  A launch certifies iff the read has at most m_k = (h0 - U_k - 1) / (a mismatch's cost) mismatches; their number is taken as Poisson with
    // mean 1 (0.5 % of 250 bases, a quarter of which put the same base back); the costs are the launches' measured times per task relative to the 32-diagonal one (profiles/r06).
    const float cdf[6] = {0.368f, 0.736f, 0.920f, 0.981f, 0.996f, 0.999f};
    const float cost[5] = {0.85f, 1.0f, 1.6f, 2.26f, 4.4f};           // 14 (opt-in), 32, 40, 62, 126 diagonals
    const int per_mm = A.max_score + 4;                              // (a mismatch instead of a match: EDNAFULL 5 + 4; a guess for other matrices -- this is a prediction)
    // q[k]: the probability that launch k cannot certify (1: its band does not even hold the path); a task that enters the chain at launch e is filled by
    // e, and by every later launch k with probability q[k - 1] (the room grows with the band: failing k - 1 implies having failed the ones before)
    float q[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int W = P.bandw[k];
        q[k] = 1.0f;
        if (W > 0 && c2_band_holds(W, D, lo, hi, 1)) {
            const int d0 = ((D - W + 3) >> 1) & ~1;
            const int U = c2_outside_band_bound(A.max_score, t.Li, t.Lj, D, d0 + W, d0 - 1, cb, go, ge, t.ref->gap_incentive_last_pos);
            const int room = h0 - U - 1;
            if (room >= 0) { const int mk = room / per_mm; q[k] = 1.0f - cdf[mk < 5 ? mk : 5]; }
        }
    }
    // backwards over the launches the chain has: S = what the launches BEHIND launch k cost a task that enters at k (the full-matrix launch: 9)
    int cls = widest, later = -1;
    float S = 0.0f, best = 1.0e30f;
    for (int k = 4; k >= 0; --k) {
        if (P.bandw[k] <= 0) continue;
        S = later < 0 ? 9.0f * q[k] : (cost[later] * q[k] + S);
        const float E_k = cost[k] + S;
        if (E_k <= best) { best = E_k; cls = k + 1; }
        later = k;
    }
    if (P.direct_full && best > 9.0f + 0.5f) cls = C2_PART_CLASSES - 1;      // no band launch is worth its fill (a read much shorter than its reference, a shift beyond every band): the full-matrix launch at once
    return cls;
}

#ifndef C2_PART_WAVES                              // wavefronts per SIMD the register budget allows: 3 (168 VGPRs, 16 spilled outside the loops; measured, round 6: 2.34 ms per
#define C2_PART_WAVES 3                            // 10 M headline reads against 2.55 ms with the 203 registers the compiler takes by itself, 2 per SIMD) -- an A/B knob
#endif
__global__ __launch_bounds__(256, C2_PART_WAVES) void c2_align_partition_kernel(c2_partition_args P)
{
    const c2_align_args& A = P.A;
    uint8_t* const flag = c2_smem;                                  // [C2_PART_CHUNK] the task's class, 7: no such task, 8: still to be probed
    unsigned* const part = (unsigned*)(c2_smem + C2_PART_CHUNK);    // [4 * 4] per wavefront: packed class counts; [16 .. 22]: bases of the seven lists; [24]: the chunk's first read length; [25]: ragged
    unsigned* const okmask = part + 32;                             // [8] a bit per byte value: a base the main-diagonal shortcut may meet in a read (code_of_char < 5)
    uint16_t* const todo = (uint16_t*)(c2_smem + C2_PART_CHUNK + 160);   // [C2_PART_CHUNK] the chunk's tasks that need the probe, densely; afterwards: the slots in length order
    uint16_t* const len16 = todo + C2_PART_CHUNK;                   // [C2_PART_CHUNK] read length of the slot's task (capped at the last bin)
    unsigned* const hist = (unsigned*)(len16 + C2_PART_CHUNK);      // [C2_PART_LEN_BINS]
    uint32_t* const roff = (uint32_t*)(hist + C2_PART_LEN_BINS);    // [C2_PART_CHUNK] where the slot's read starts, relative to the chunk's first read
    uint16_t* const rid16 = (uint16_t*)(roff + C2_PART_CHUNK);       // [C2_PART_CHUNK] the slot's reference
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int PT = C2_PART_CHUNK / 256;                          // tasks per thread of a chunk
    int widest = 2;                                                 // the class that takes what no band holds: the widest band launch the chain has
    for (int k = 2; k < 5; ++k) if (P.bandw[k] > 0) widest = k + 1;
    // An all-references batch (task = read * n_refs + reference) is walked REFERENCE-MAJOR inside a chunk: slot s of a chunk of `rpc` reads is read
    // s % rpc against reference s / rpc.  The lists keep slot order, so their neighbours are two reads against the SAME reference -- what the packed
    // kernels need to put two alignments into one lane group (a pair shares reference and read length); task order would put (read r, reference 0),
    // (r, 1), (r, 2) next to each other and nothing would pair.  Every other batch: slot = task.
    const c2_part_walk WK = c2_part_walk_of(A);
    const bool may_sort = P.sort_by_length && WK.k == 1u;
    {   // the bases a read finished on the main diagonal may differ in: A C G T N where the matrix scores them (code_of_char < 5), as a bit per byte value
        const unsigned long long okb = __ballot(A.code_of_char[tid] < 5);
        if (lane == 0) { okmask[2 * wv] = (unsigned)okb; okmask[2 * wv + 1] = (unsigned)(okb >> 32); }
    }
    for (uint64_t chunk = (uint64_t)blockIdx.x * WK.chunk_tasks; chunk < A.n_tasks; chunk += (uint64_t)gridDim.x * WK.chunk_tasks) {
#ifdef C2_PART_PHASES
        unsigned long long pt_last = (unsigned long long)clock64();
#endif
        // ---- one lane per task: the look at the last 32 columns
        const uint64_t roff_base = A.offsets[A.all_refs ? chunk / (uint64_t)A.n_refs : chunk];
        int ragged = 0;
        unsigned n_exact = 0;                                       // (wave-uniform) class-0 tasks this wavefront finished itself
        for (int r = 0; r < C2_PART_CHUNK / 256; ++r) {
            const int slot = r * 256 + tid;
            const uint64_t task = c2_part_task_of(WK, A, chunk, slot);
            int cls = 7, lj = -1;
            if (task < A.n_tasks) {
                cls = 2;
                const c2_part_task t = c2_part_load(A, task);
                lj = t.Lj;
                roff[slot] = (uint32_t)(t.off - roff_base);
                rid16[slot] = (uint16_t)t.ref_id;
                if (!t.rc && t.pk_ok && t.Lj >= 32) {
                    int mm = 0x10000;
                    if (t.Lj == t.Li && t.Lj <= 256) {
                        mm = 0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            uint32_t a, b;
                            __builtin_memcpy(&a, t.rd + (t.Lj - 32) + 4 * q, 4); __builtin_memcpy(&b, t.f + (t.Lj - 32) + 4 * q, 4);
                            const uint32_t x = a ^ b;
                            mm += __builtin_popcount((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
                        }
                    }
                    // ... and the 32 columns around the cut site (where the edits are, and where a candidate amplicon differs from the others: a read of
                    // the wild type against the prime-edited amplicon of the same length is a dozen mismatches in a row there -- gap-free, but more
                    // than the 14-diagonal certificate of the score-only launch allows: its fill would be for nothing)
                    if (P.check_cut && mm <= P.max_mismatch && t.cut >= 0) {
                        int c0 = t.cut - 16;
                        if (c0 > t.Lj - 64) c0 = t.Lj - 64;                 // (the last 32 columns have been looked at)
                        if (c0 < 0) c0 = 0;
                        if (c0 + 32 <= t.Lj - 32) {
                            int m2 = 0;
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                uint32_t a, b;
                                __builtin_memcpy(&a, t.rd + c0 + 4 * q, 4); __builtin_memcpy(&b, t.f + c0 + 4 * q, 4);
                                const uint32_t x = a ^ b;
                                m2 += __builtin_popcount((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
                            }
                            if (m2 > P.max_mismatch) mm = 0x10000;
                        }
                    }
                    if (mm <= P.max_mismatch) cls = 0;
                    else if (c2_part_probes(P, t)) cls = 8;
                }
            }
            flag[slot] = (uint8_t)cls;
            if (may_sort) {
                len16[slot] = (uint16_t)(lj < 0 ? 0 : (lj < C2_PART_LEN_BINS - 1 ? lj : C2_PART_LEN_BINS - 1));
                if (slot == 0) { part[24] = (unsigned)lj; part[25] = 0u; }
            }
        }
        __syncthreads();
        C2_PART_MARK(0);
        // ---- class 0, and the read lies ON its reference's main diagonal: a byte-for-byte copy (the unedited, error-free read: the commonest read of
        //      an amplicon run) or one that differs in one or two bases.  Where c2_main_diagonal_certificate (host) proves that the diagonal beats
        //      every other path there is nothing to fill: the aligned strings are the read and the reference, the only events substitutions.
        //      EIGHT lanes per candidate, 32 bytes each (the last lane's block overlaps the one before it: nothing is read behind the read): the
        //      read against the reference, and -- for the paths that are ONE other diagonal from end to end -- against the reference shifted by
        //      +-1 and +-2 over the 16-byte halves that lie well inside both (their equal bytes, every other column taken as equal, must stay
        //      under the host's limit).  The eight lanes write both rows from the registers they compared (256 contiguous bytes per row) and
        //      their first lane the record; the slot's flag becomes 9, in no list.  The candidates are first gathered densely, as the probe's are.
        if (P.exact_copies) {
            unsigned n0 = 0;
#pragma unroll
            for (int k = 0; k < PT; ++k) n0 += flag[PT * tid + k] == 0u;
            unsigned incl = n0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const unsigned o = (unsigned)__shfl_up((int)incl, d); if (lane >= d) incl += o; }
            if (lane == 63) part[wv] = incl;
            __syncthreads();
            unsigned before = 0, total = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) { const unsigned x = part[v]; if (v < wv) before += x; total += x; }
            unsigned pos = before + incl - n0;
#pragma unroll
            for (int k = 0; k < PT; ++k) if (flag[PT * tid + k] == 0u) todo[pos++] = (uint16_t)(PT * tid + k);
            __syncthreads();
            auto flags = [](const uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; };     // bit 7 of every non-zero byte
            const int q = tid & 7, grp = (lane >> 3) << 3;          // this lane's 32-byte block; the first lane of its group of eight
            // Round 6: nothing in a candidate's way depends on a load but the read's own 32 bytes per lane.  Where the read lies comes from LDS (roff,
            // written by the look at the last 32 columns); the differing bases, their window membership and the equal bases of the diagonals +-1 / +-2
            // come out of the registers that were compared (the shifted diagonals as 2-bit codes of the reference block and four bits of either
            // neighbour's -- a code aliases N / IUPAC / lower case to one of A C G T, which can only COUNT MORE equal bases: the safe side of the
            // host's limit); and with ONE reference in the batch its side (ref_t) is made once per workgroup.
            struct ref_t { uint4 y0, y1; uint64_t w[4]; uint32_t win; int mmax[4]; int L, nq, at, ov, kmax, ref_id; const c2_dev_ref* ref; bool live, has_win; };
            struct cand_t { bool act; int slot, ref_id; uint64_t task; uint4 x0, x1; };
            auto regs_code32 = [](const uint4& a, const uint4& b) {  // c2_code32 of 32 bytes held in registers
                const uint32_t wds[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                uint64_t code = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t t_ = (wds[k] >> 1) & 0x03030303u;
                    code |= (uint64_t)((t_ | (t_ >> 6) | (t_ >> 12) | (t_ >> 18)) & 0xffu) << (8 * k);
                }
                return code;
            };
            // the reference's side of a candidate of length L (every lane of the wavefront calls it together: it exchanges codes with its neighbours)
            auto ref_side = [&](const c2_dev_ref* rf, const int ref_id, const int L, const bool live, const bool want_win) {
                ref_t R;
                R.ref = rf; R.ref_id = ref_id; R.L = L; R.live = live; R.has_win = want_win;
                R.kmax = live ? rf->diag_kmax : -1;
                R.nq = (L + 31) >> 5;                               // 32 <= L <= 256: 1 .. 8 blocks
                R.at = (32 * q + 32 <= L) ? 32 * q : L - 32;        // (the last block starts at L - 32)
                R.ov = 32 * R.nq - L;                               // bytes at the start of the last block that the block before it holds too
                R.y0 = uint4{0u, 0u, 0u, 0u}; R.y1 = R.y0; R.win = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) R.mmax[k] = 0;
                const bool mine = live && q < R.nq;
                if (mine) {
                    __builtin_memcpy(&R.y0, rf->seq + R.at, 16); __builtin_memcpy(&R.y1, rf->seq + R.at + 16, 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) R.mmax[k] = rf->diag_mmax[k];
                    if (want_win) {                                 // which of the block's 32 positions are in the quantification window (COREResources.pyx:116)
                        const uint16_t* ip = rf->inc_prefix + R.at;
                        unsigned prev = ip[0];
#pragma unroll 8
                        for (int k = 0; k < 32; ++k) { const unsigned nx = ip[k + 1]; R.win |= (unsigned)(nx != prev) << k; prev = nx; }
                    }
                }
                const uint64_t fc = regs_code32(R.y0, R.y1);
                const uint32_t pv_hi = (uint32_t)__shfl((int)(uint32_t)(fc >> 32), (lane + 63) & 63);      // (only its top four bits are used)
                const uint64_t pv = (uint64_t)pv_hi << 32;
                const uint64_t nx = (uint64_t)(uint32_t)__shfl((int)(uint32_t)fc, (lane + 1) & 63) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(fc >> 32), (lane + 1) & 63) << 32);
                const uint64_t nlo = (q + 1 == R.nq - 1) ? (nx >> (2 * (R.ov & 31))) : nx;          // (the next block is the overlapping last one: its bases from R.ov on)
                R.w[0] = (fc << 4) | ((pv >> 60) & 0xfull);         // reference bases at - 2 .. at + 29
                R.w[1] = (fc << 2) | ((pv >> 62) & 0x3ull);         //                 at - 1 .. at + 30
                R.w[2] = (fc >> 2) | ((nlo & 0x3ull) << 62);        //                 at + 1 .. at + 32
                R.w[3] = (fc >> 4) | ((nlo & 0xfull) << 60);        //                 at + 2 .. at + 33
                return R;
            };
            const bool one_ref = A.n_refs == 1 || (!A.all_refs && A.ref_ids == nullptr);
            ref_t R0;
            {
                const c2_dev_ref* rf0 = A.refs;
                const int L0 = rf0->len;
                R0 = ref_side(rf0, 0, (L0 >= 32 && L0 <= 256) ? L0 : 32, one_ref && L0 >= 32 && L0 <= 256 && rf0->diag_kmax >= 0, true);
            }
            // a candidate's loads: the read's 32 bytes of this lane.  One reference in the batch: where the read lies comes from LDS, and two candidates
            // are being looked at while the next two are on their way; else the candidate's reference side is made from its own loads first.
            auto fetch1 = [&](const unsigned i) {
                cand_t c;
                c.act = i < total;
                c.slot = c.act ? (int)todo[i] : 0;
                c.task = c.act ? c2_part_task_of(WK, A, chunk, c.slot) : 0ull;
                c.ref_id = 0;
                c.x0 = uint4{0u, 0u, 0u, 0u}; c.x1 = c.x0;
                if (c.act && R0.live && q < R0.nq) {
                    const uint8_t* rd = A.reads + (roff_base + (uint64_t)roff[c.slot]) + R0.at;
                    __builtin_memcpy(&c.x0, rd, 16); __builtin_memcpy(&c.x1, rd + 16, 16);
                }
                return c;
            };
            // several references: the slot's reference from LDS, its length from the reference's record (one load in front of the read's, an L1 hit); the
            // reference's side is kept per lane group and made again when a group meets another reference (the lists are in slot order: reference-major
            // chunks of an all-references batch, runs of one amplicon in a pooled one -- a few times per chunk)
            auto fetchM = [&](const unsigned i) {
                cand_t c;
                c.act = i < total;
                c.slot = c.act ? (int)todo[i] : 0;
                c.task = c.act ? c2_part_task_of(WK, A, chunk, c.slot) : 0ull;
                c.ref_id = c.act ? (int)rid16[c.slot] : 0;
                c.x0 = uint4{0u, 0u, 0u, 0u}; c.x1 = c.x0;
                if (c.act) {
                    const int L = A.refs[c.ref_id].len;                 // (a class-0 task: its read is as long as its reference, 32 .. 256)
                    const int nq = (L + 31) >> 5, at = (32 * q + 32 <= L) ? 32 * q : L - 32;
                    if (q < nq) {
                        const uint8_t* rd = A.reads + (roff_base + (uint64_t)roff[c.slot]) + at;
                        __builtin_memcpy(&c.x0, rd, 16); __builtin_memcpy(&c.x1, rd + 16, 16);
                    }
                }
                return c;
            };
            ref_t Rc = R0;                                           // (not live unless the batch has one reference)
            Rc.ref_id = one_ref ? 0 : -1;
            auto ensure = [&](const cand_t& c) {                     // every lane of the wavefront together (ref_side exchanges codes between lanes)
                const bool need = c.act && c.ref_id != Rc.ref_id;
                if (__ballot(need) != 0ull) {
                    const int rid = need ? c.ref_id : (Rc.ref_id >= 0 ? Rc.ref_id : 0);
                    const c2_dev_ref* rf = A.refs + rid;
                    const int L = rf->len;
                    const bool ok = (need || Rc.ref_id >= 0) && L >= 32 && L <= 256 && rf->diag_kmax >= 0;
                    const int keep = (need || Rc.ref_id >= 0) ? rid : -1;
                    Rc = ref_side(rf, rid, (L >= 32 && L <= 256) ? L : 32, ok, false);   // (the window bits of a differing base: two loads where one is met -- a batch whose neighbours
                                                                                          //  are of different amplicons comes through here for every candidate)
                    Rc.ref_id = keep;
                }
            };
            auto finish = [&](const cand_t& c, const ref_t& R) {
                const bool live = c.act && R.live;
                const int L = R.L, nq = R.nq, at = R.at, ov = R.ov;
                const uint4 &x0 = c.x0, &x1 = c.x1, &y0 = R.y0, &y1 = R.y1;
                unsigned kl = 0, eqw = 0, halves = 0;               // this lane's differing bases; equal bases on the four shifted diagonals (a byte each); halves counted
                unsigned wa = 0, wb = 0;                            // its first two differing bases: position | read byte << 9 | in the window << 17 | valid << 18
                const bool mine = live && q < nq;
                if (mine) {
                    const uint32_t xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    uint32_t f[8] = {flags(x0.x ^ y0.x), flags(x0.y ^ y0.y), flags(x0.z ^ y0.z), flags(x0.w ^ y0.w),
                                     flags(x1.x ^ y1.x), flags(x1.y ^ y1.y), flags(x1.z ^ y1.z), flags(x1.w ^ y1.w)};
                    if (q == nq - 1 && ov) {                        // its first `ov` bytes were counted by the lane before
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            const int drop = ov - 4 * w;            // bytes of this dword to forget
                            if (drop >= 4) f[w] = 0u; else if (drop > 0) f[w] &= ~0u << (8 * drop);
                        }
                    }
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        uint32_t g = f[w];
                        kl += (unsigned)__builtin_popcount(g);
                        while (g) {
                            const int by = __builtin_ctz(g) >> 3, in = 4 * w + by;
                            g &= g - 1u;
                            if (wa && wb) continue;
                            unsigned inw = (R.win >> in) & 1u;
                            if (!R.has_win) { const uint16_t* ip = R.ref->inc_prefix + at + in; inw = ip[1] != ip[0]; }
                            const unsigned e = (unsigned)(at + in) | (((xs[w] >> (8 * by)) & 0xffu) << 9) | (inw << 17) | (1u << 18);
                            if (!wa) wa = e; else wb = e;
                        }
                    }
                    if (R.kmax > 0 && q < nq - 1) {                 // (not the last block: it overlaps)
                        const uint64_t rc = regs_code32(x0, x1), m55 = 0x5555555555555555ull;
                        const bool h0 = at >= 2 && at + 18 <= L, h1 = at + 34 <= L;          // reference[o - 2 .. o + 17] exists, o = at, at + 16
                        halves = (unsigned)h0 + (unsigned)h1;
#pragma unroll
                        for (int sft = 0; sft < 4; ++sft) {
                            const uint64_t d = rc ^ R.w[sft];
                            const uint64_t ne = (d | (d >> 1)) & m55;   // a bit per differing base
                            const unsigned e = (h0 ? 16u - (unsigned)__builtin_popcount((uint32_t)ne) : 0u) + (h1 ? 16u - (unsigned)__builtin_popcount((uint32_t)(ne >> 32)) : 0u);
                            eqw |= e << (8 * sft);
                        }
                    }
                }
                // the group's sums (every lane of the wavefront takes part: no branch around the exchanges)
                unsigned ks = kl < 3u ? kl : 3u;
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) {
                    ks += (unsigned)__shfl_xor((int)ks, d); eqw += (unsigned)__shfl_xor((int)eqw, d); halves += (unsigned)__shfl_xor((int)halves, d);
                }
                // ... and its (at most two) differing bases: the first lane that holds one, then that lane's second or the next lane's first
                const unsigned gm = (unsigned)(__ballot(kl > 0u) >> grp) & 0xffu;
                const int l1 = gm ? __builtin_ctz(gm) : 0, l2 = (gm & (gm - 1u)) ? __builtin_ctz(gm & (gm - 1u)) : l1;
                const unsigned e1 = (unsigned)__shfl((int)wa, grp + l1), e1b = (unsigned)__shfl((int)wb, grp + l1), e2n = (unsigned)__shfl((int)wa, grp + l2);
                const int k = (int)ks;
                const unsigned e2 = e1b ? e1b : (l2 != l1 ? e2n : 0u);
                const int p1 = e1 ? (int)(e1 & 0x1ffu) : -1, p2 = e2 ? (int)(e2 & 0x1ffu) : -1;
                bool done = live && k <= R.kmax;
                int n_all_sub = 0, n_win_sub = 0, irregular = 0;
                if (done && k > 0) {
                    const int counted = 16 * (int)halves;
                    const int e_m2 = (int)(eqw & 0xffu), e_m1 = (int)((eqw >> 8) & 0xffu), e_p1 = (int)((eqw >> 16) & 0xffu), e_p2 = (int)(eqw >> 24);
                    const int m1 = (e_m1 > e_p1 ? e_m1 : e_p1) + (L - 1 - counted), m2 = (e_m2 > e_p2 ? e_m2 : e_p2) + (L - 2 - counted);
                    if (m1 > (k == 1 ? R.mmax[0] : R.mmax[2]) || m2 > (k == 1 ? R.mmax[1] : R.mmax[3])) done = false;
                    // the differing bases: A C G T N only (anything else keeps its launch: status words, IUPAC scores); the substitutions among them
                    const unsigned ee[2] = {e1, k > 1 ? e2 : 0u};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (!ee[e] || !done) continue;
                        const int pos_ = (int)(ee[e] & 0x1ffu);
                        const unsigned rb = (ee[e] >> 9) & 0xffu;
                        if (!((okmask[rb >> 5] >> (rb & 31u)) & 1u)) { done = false; continue; }
                        if (rb != 'N') {                            // COREResources.pyx:113-118
                            ++n_all_sub;
                            n_win_sub += (int)((ee[e] >> 17) & 1u);
                        }
                        if (pos_ == 0 || pos_ == L - 1) irregular = 1;
                    }
                }
                if (done) {
                    uint8_t* outR = A.aln_read + c.task * (uint64_t)A.aln_stride;
                    uint8_t* outF = A.aln_ref + c.task * (uint64_t)A.aln_stride;
                    if (!(A.reserved & 1)) {
                        if (q < nq) {
                            __builtin_memcpy(outR + at, &x0, 16); __builtin_memcpy(outR + at + 16, &x1, 16);
                            __builtin_memcpy(outF + at, &y0, 16); __builtin_memcpy(outF + at + 16, &y1, 16);
                        }
                        if (q == nq - 1 && (L & 3)) {               // (a partial last dword is padded with zeros, as c2_emit_gapless4 pads it: the last block's top bytes)
                            const uint32_t w = x1.w >> (8 * (4 - (L & 3))), v = y1.w >> (8 * (4 - (L & 3)));
                            __builtin_memcpy(outR + (L & ~3), &w, 4); __builtin_memcpy(outF + (L & ~3), &v, 4);
                        }
                    }
                    if (q == 0) {
                        c2_aln_record rec;
                        c2_clear_record(rec, 0, R.ref_id);
                        rec.aln_len = (uint16_t)L; rec.matches = (uint16_t)(L - k);                             // pyx:375-376
                        rec.substitution_n = (uint16_t)n_win_sub; rec.all_substitutions = (uint16_t)n_all_sub;
                        rec.irregular_ends = (uint8_t)irregular;
                        A.records[c.task] = rec;
                        flag[c.slot] = 9u;
                        if (A.diag_hints) {                         // the alignment once more, as one word (c2_batch.diag_hints): the count pass need not read the rows back
                            unsigned h = C2_HINT_VALID | ((unsigned)k << 24);
                            if (k > 0) h |= (unsigned)p1 | ((((e1 >> 9) >> 1) & 7u) << 9);
                            if (k > 1) h |= ((unsigned)p2 << 12) | ((((e2 >> 9) >> 1) & 7u) << 21);
                            A.diag_hints[4u * c.task] = h;               // (words 1 .. 3 stay 0)
                        }
                    }
                }
                n_exact += (unsigned)__popcll(__ballot(done && q == 0));
            };
            // 64 candidates per step of the workgroup: two per group of eight lanes; the next step's loads are issued before this step's are looked at
            if (one_ref) {
                cand_t ca = fetch1((unsigned)(tid >> 3)), cb = fetch1(32u + (unsigned)(tid >> 3));
                for (unsigned i0 = 0; i0 < total; i0 += 64u) {
                    const cand_t na = fetch1(i0 + 64u + (unsigned)(tid >> 3));
                    const cand_t nb = fetch1(i0 + 96u + (unsigned)(tid >> 3));
                    finish(ca, R0);
                    finish(cb, R0);
                    ca = na; cb = nb;
                }
            } else {
                cand_t ca = fetchM((unsigned)(tid >> 3)), cb = fetchM(32u + (unsigned)(tid >> 3));
                for (unsigned i0 = 0; i0 < total; i0 += 64u) {
                    const cand_t na = fetchM(i0 + 64u + (unsigned)(tid >> 3));
                    const cand_t nb = fetchM(i0 + 96u + (unsigned)(tid >> 3));
                    ensure(ca);
                    finish(ca, Rc);
                    ensure(cb);
                    finish(cb, Rc);
                    ca = na; cb = nb;
                }
            }
            if (lane == 0 && n_exact && P.class_count) { atomicAdd(P.class_count + 0, n_exact); atomicAdd(P.class_count + 7, n_exact); }
            __syncthreads();
        }
        C2_PART_MARK(1);
        if (may_sort) {                                             // do the chunk's reads differ in length?
            const int l0 = (int)part[24];
            for (int r = 0; r < C2_PART_CHUNK / 256; ++r) {
                const int slot = r * 256 + tid;
                if (flag[slot] != 7u && (int)len16[slot] != (l0 < C2_PART_LEN_BINS - 1 ? l0 : C2_PART_LEN_BINS - 1)) part[25] = 1u;
            }
            __syncthreads();
            ragged = (int)part[25];
        }
        // ---- the tasks that need the probe, densely (about a third of an amplicon run's reads, scattered: probing them where they stand would
        //      keep every wavefront in the loop with a third of its lanes), then one lane per such task
        {
            unsigned n8 = 0;
#pragma unroll
            for (int k = 0; k < PT; ++k) n8 += flag[PT * tid + k] == 8u;
            unsigned incl = n8;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const unsigned o = (unsigned)__shfl_up((int)incl, d); if (lane >= d) incl += o; }
            if (lane == 63) part[wv] = incl;
            __syncthreads();
            unsigned before = 0, total = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) { const unsigned x = part[v]; if (v < wv) before += x; total += x; }
            unsigned pos = before + incl - n8;
#pragma unroll
            for (int k = 0; k < PT; ++k) if (flag[PT * tid + k] == 8u) todo[pos++] = (uint16_t)(PT * tid + k);
            __syncthreads();
            for (unsigned k = (unsigned)tid; k < total; k += 256u) {
                const int slot = (int)todo[k];
                const c2_part_task t = c2_part_load(A, c2_part_task_of(WK, A, chunk, slot));
                flag[slot] = (uint8_t)c2_part_probe(P, t, widest);
            }
            __syncthreads();
        }
        C2_PART_MARK(2);
        // ---- a ragged chunk: its slots in the order of their reads' lengths (counting sort; the order among equal lengths is that of the atomics --
        //      no result depends on the order of a list).  `todo` holds the permutation from here on; a uniform chunk keeps slot order.
        if (ragged) {
            for (int b = tid; b < C2_PART_LEN_BINS; b += 256) hist[b] = 0u;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PT; ++k) atomicAdd(&hist[len16[PT * tid + k]], 1u);
            __syncthreads();
            {   // exclusive scan of the bins: thread t owns bins 2t, 2t + 1
                const unsigned h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
                unsigned incl = h0 + h1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const unsigned o = (unsigned)__shfl_up((int)incl, d); if (lane >= d) incl += o; }
                if (lane == 63) part[wv] = incl;
                __syncthreads();
                unsigned before = 0;
#pragma unroll
                for (int v = 0; v < 4; ++v) if (v < wv) before += part[v];
                const unsigned e = before + incl - (h0 + h1);
                hist[2 * tid] = e; hist[2 * tid + 1] = e + h0;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PT; ++k) { const int slot = PT * tid + k; todo[atomicAdd(&hist[len16[slot]], 1u)] = (uint16_t)slot; }
            __syncthreads();
        }
        // ---- thread t owns positions PT t .. PT t + PT - 1 of the chunk's order: list positions by a scan over the workgroup, one atomic per chunk and class
        constexpr int NC = C2_PART_CLASSES, NW = (C2_PART_CLASSES + 1) / 2;
        unsigned n[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) n[c] = 0u;
        unsigned mine[PT];                                          // the slots in this thread's positions
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            mine[k] = ragged ? (unsigned)todo[PT * tid + k] : (unsigned)(PT * tid + k);
            const unsigned f = flag[mine[k]];
#pragma unroll
            for (int c = 0; c < NC; ++c) n[c] += f == (unsigned)c;
        }
        unsigned pk[NW], incl[NW];                                  // two 16-bit counters per word (every count <= 4096, every sum <= 4096)
#pragma unroll
        for (int w = 0; w < NW; ++w) { pk[w] = n[2 * w] | ((2 * w + 1 < NC ? n[2 * w + 1 < NC ? 2 * w + 1 : 0] : 0u) << 16); incl[w] = pk[w]; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int w = 0; w < NW; ++w) { const unsigned o = (unsigned)__shfl_up((int)incl[w], d); if (lane >= d) incl[w] += o; }
        }
        if (lane == 63) {
#pragma unroll
            for (int w = 0; w < NW; ++w) part[NW * wv + w] = incl[w];
        }
        __syncthreads();
        unsigned before[NW], total[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) { before[w] = 0u; total[w] = 0u; }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int w = 0; w < NW; ++w) { const unsigned x = part[NW * v + w]; if (v < wv) before[w] += x; total[w] += x; }
        if (tid < NC) {
            const unsigned t = (tid & 1) ? (total[tid >> 1] >> 16) : (total[tid >> 1] & 0xffffu);
            part[16 + tid] = t ? atomicAdd(P.count[tid], t) : 0u;
            if (t && P.class_count) atomicAdd(P.class_count + tid, t);
        }
        __syncthreads();
        unsigned pos[NC];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned e = before[w] + incl[w] - pk[w];
            pos[2 * w] = part[16 + 2 * w] + (e & 0xffffu);
            if (2 * w + 1 < NC) pos[2 * w + 1 < NC ? 2 * w + 1 : 0] = part[16 + 2 * w + 1] + (e >> 16);
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const unsigned f = flag[mine[k]];
            const uint32_t task = (uint32_t)c2_part_task_of(WK, A, chunk, (int)mine[k]);     // (f < NC only for a slot that holds a task)
#pragma unroll
            for (int c = 0; c < NC; ++c) if (f == (unsigned)c) P.list[c][pos[c]++] = task;
        }
        __syncthreads();                                            // (the flags are overwritten by the next chunk)
        C2_PART_MARK(3);
    }
}

// ... and of the row forms (c2_rshr1z / c2_rshl1z: the hand-off of the 16-lane groups of c2_align_diagp_kernel<8>), folded into an
// add like the kernels use them: out[lane] = value of lane - 1 (0 at the start of a row of 16) + 1000, out[64 + lane] = value of
// lane + 1 (0 at the end of a row) + 1000.
__global__ __launch_bounds__(64) void c2_selftest_rows_kernel(int* out)
{
    const int lane = threadIdx.x;
    int k1000 = 1000;
    C2_KEEP_IN_VGPR(k1000);
    out[lane] = c2_rshr1z(lane * 3 + 1) + k1000;
    out[64 + lane] = c2_rshl1z(lane * 3 + 1) + k1000;
}
