// Host-side preparation shared by the C ABI (c2_api_*.hip) and the test-only wave emulator
// harness (tests/emu/): plain C++, no HIP.  Marshals the reference's Python-level inputs
// (int64 score matrix indexed by ord(char), include_idxs list) into the kernel's compact tables.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>
#include "c2_device.h"

struct c2_scoring_tables {
    uint8_t code_of_char[256];
    std::vector<int16_t> tbl;     // n_codes x n_codes, [ref code][read code]
    std::vector<uint32_t> pk;     // per ref code: signed 4-bit scores against read codes 0..7 (empty if some score is outside [-8,7])
    int n_codes = 0;
    int first_ext_code = 0;       // codes >= this: READ characters with dim <= ord < 128 (see c2_build_scoring); invalid in a reference
    int mat_dim = 0;
};

// matrix: row-major int64[dim][dim], indexed [ord(ref)][ord(read)] (CRISPResso2Align.pyx:212).
// Characters with an all-zero row and column share one code.  The reference indexes the matrix with bounds checking off, so
// a READ character with dim <= ord < 128 reads the flat element ci * dim + cj -- a later row of the same buffer (its own
// main() does that on every default run: the flexiguide sequence defaults to the string "None", CRISPRessoCORE.py:3095-3105;
// lower-case reads do it too).  Such characters get codes of their own, >= first_ext_code, whose table COLUMN holds those
// elements: one shared code when they are all zero (EDNAFULL, BLOSUM62, make_matrix: always), else one code each.  Whether
// the element exists at all depends on the largest reference character (c2_dev_ref.max_char): the kernels check
// max_char * dim + cj < dim * dim per alignment.  A reference character >= dim, and any byte >= 128 (`char` is signed in
// the reference), is an out-of-bounds read there and gets C2_INVALID_CODE / C2_STATUS_OOB_CHAR here.
inline bool c2_build_scoring(const int64_t* matrix, int dim, c2_scoring_tables& out, std::string& err) {
    if (!matrix || dim <= 0) { err = "score matrix missing"; return false; }
    std::fill(out.code_of_char, out.code_of_char + 256, (uint8_t)C2_INVALID_CODE);
    const int lim = dim < 128 ? dim : 128;
    std::vector<int> syms;
    bool any_zero = false;
    std::vector<uint8_t> scoring(lim, 0), is_sym(lim, 0);
    for (int c = 0; c < lim; ++c) {
        bool nz = false;
        for (int k = 0; k < dim && !nz; ++k) nz = matrix[(size_t)c * dim + k] != 0 || matrix[(size_t)k * dim + c] != 0;
        scoring[c] = nz; is_sym[c] = nz; if (!nz) any_zero = true;
    }
    // the bases that make up real reads get the lowest codes, so that the packed 8-symbol score rows cover them
    for (const char* q = "ACGTN"; *q; ++q) if (*q < lim && scoring[(int)*q]) { syms.push_back(*q); scoring[(int)*q] = 0; }
    for (int c = 0; c < lim; ++c) if (scoring[c]) syms.push_back(c);
    // read characters beyond the matrix: the flat elements ci * dim + c, for every reference character ci that leaves them inside the buffer
    std::vector<int> ext_nz;
    bool ext_zero = false;
    std::vector<int> ext_kind(128, 0);                     // 0 invalid, 1 all zero, 2 own column
    for (int c = dim; c < 128; ++c) {
        bool nz_sym = false, nz_other = false;
        for (int ci = 0; ci < lim; ++ci) {
            const size_t e = (size_t)ci * dim + c;
            if (e >= (size_t)dim * dim) break;
            if (matrix[e] != 0) { if (is_sym[ci]) nz_sym = true; else nz_other = true; }
        }
        if (nz_other) continue;                            // (a non-zero score against a symbol of the shared zero code: not representable, stays invalid)
        ext_kind[c] = nz_sym ? 2 : 1;
        if (nz_sym) ext_nz.push_back(c); else ext_zero = true;
    }
    const int zero_code = (int)syms.size();
    const int first_ext = zero_code + (any_zero ? 1 : 0);
    const int n = first_ext + (ext_zero ? 1 : 0) + (int)ext_nz.size();
    if (n > C2_MAX_CODES) { err = "score matrix has more than " + std::to_string(C2_MAX_CODES - 1) + " scoring symbols"; return false; }
    out.n_codes = n;
    out.first_ext_code = first_ext;
    out.mat_dim = dim;
    out.tbl.assign((size_t)n * n, 0);
    for (size_t a = 0; a < syms.size(); ++a) out.code_of_char[syms[a]] = (uint8_t)a;
    if (any_zero) for (int c = 0; c < lim; ++c) if (out.code_of_char[c] == C2_INVALID_CODE) out.code_of_char[c] = (uint8_t)zero_code;
    for (int c = dim; c < 128; ++c) if (ext_kind[c] == 1) out.code_of_char[c] = (uint8_t)first_ext;
    for (size_t x = 0; x < ext_nz.size(); ++x) out.code_of_char[ext_nz[x]] = (uint8_t)(first_ext + (ext_zero ? 1 : 0) + (int)x);
    for (size_t a = 0; a < syms.size(); ++a) {
        for (size_t b = 0; b < syms.size(); ++b) {
            const int64_t v = matrix[(size_t)syms[a] * dim + syms[b]];
            if (v < -32768 || v > 32767) { err = "score matrix entry outside int16"; return false; }
            out.tbl[a * n + b] = (int16_t)v;
        }
        for (size_t x = 0; x < ext_nz.size(); ++x) {
            const size_t e = (size_t)syms[a] * dim + ext_nz[x];
            const int64_t v = e < (size_t)dim * dim ? matrix[e] : 0;
            if (v < -32768 || v > 32767) { err = "score matrix entry outside int16"; return false; }
            out.tbl[a * n + (size_t)(first_ext + (ext_zero ? 1 : 0)) + x] = (int16_t)v;
        }
    }
    bool nib = true;
    for (int16_t v : out.tbl) if (v < -8 || v > 7) nib = false;
    out.pk.clear();
    if (nib) {
        out.pk.assign(n, 0);
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < 8 && b < n; ++b) out.pk[a] |= ((uint32_t)out.tbl[(size_t)a * n + b] & 0xFu) << (4 * b);
    }
    return true;
}

// The two 8-entry byte tables the multi-alignment kernels' read staging looks bases up in with v_perm_b32, indexed by
// (ch >> 1) & 7 -- a perfect hash of A (0), C (1), T (2), G (3), N (7): the score-table code of the base, and the base itself
// (to verify that the byte really was that base).  An entry whose base has no packed code (>= 8, or not a scoring symbol of
// this matrix) holds 0xFF in the letter table, so such reads never pass the check and take the general staging path.
inline void c2_build_base_luts(const c2_scoring_tables& sc, uint32_t& code_lo, uint32_t& code_hi, uint32_t& chr_lo, uint32_t& chr_hi) {
    unsigned char code[8], chr[8];
    for (int k = 0; k < 8; ++k) { code[k] = 0xFF; chr[k] = 0xFF; }
    for (const char* q = "ACTGN"; *q; ++q) {
        const int k = (*q >> 1) & 7;
        const unsigned char c = sc.code_of_char[(unsigned char)*q];
        if (c < 8 && !sc.pk.empty()) { code[k] = c; chr[k] = (unsigned char)*q; }
    }
    code_lo = code[0] | (code[1] << 8) | (code[2] << 16) | ((uint32_t)code[3] << 24);
    code_hi = code[4] | (code[5] << 8) | (code[6] << 16) | ((uint32_t)code[7] << 24);
    chr_lo = chr[0] | (chr[1] << 8) | (chr[2] << 16) | ((uint32_t)chr[3] << 24);
    chr_hi = chr[4] | (chr[5] << 8) | (chr[6] << 16) | ((uint32_t)chr[7] << 24);
}

// c2_dev_ref.seq2: the reference as 2-bit codes, 16 per word, with two zero words on either side (out[2] is word 0)
inline void c2_build_seq2(const char* seq, int Li, std::vector<uint32_t>& out) {
    out.assign((size_t)((Li + 15) / 16) + 4, 0u);
    for (int k = 0; k < Li; ++k) out[2 + (k >> 4)] |= (uint32_t)(((unsigned char)seq[k] >> 1) & 3u) << (2 * (k & 15));
}

// inc_prefix[x] = number of distinct include idxs < x, x in [0, Li+1]
inline void c2_build_inc_prefix(const int32_t* inc, int n_inc, int Li, std::vector<uint16_t>& out) {
    std::vector<uint8_t> bit((size_t)Li + 2, 0);
    for (int k = 0; k < n_inc; ++k) if (inc[k] >= 0 && inc[k] <= Li) bit[inc[k]] = 1;
    out.assign((size_t)Li + 2, 0);
    uint16_t run = 0;
    for (int x = 0; x < Li + 2; ++x) { out[x] = run; run = (uint16_t)(run + bit[x]); }
}

// Row records of the diagonal-band kernels for one reference (rows 0 .. Li+1, rows 0 and Li+1 zero, plus padding); empty if the
// scoring has no packed form.  g32: the gap incentives already truncated to int32.
inline void c2_build_diag_rows(const char* seq, int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend,
                               std::vector<c2_diag_row>& out) {
    out.clear();
    if (sc.pk.empty()) return;
    // C2_DIAG_ROW_PAD zero records on either side: lanes whose diagonal has not entered the matrix yet (or has left it) index
    // past the ends; row i is out[C2_DIAG_ROW_PAD + i]
    out.assign((size_t)Li + 2 + 2 * C2_DIAG_ROW_PAD, c2_diag_row{0, 0, 0, 0u});
    for (int i = 1; i <= Li; ++i) {
        const int open = (i == Li) ? gap_extend : gap_open;        // last row: gap_open -> gap_extend (CRISPResso2Align.pyx:277-317)
        c2_diag_row r;
        r.a = open + g32[i]; r.b = gap_extend + g32[i]; r.c = open + g32[i - 1];
        const uint8_t code = sc.code_of_char[(unsigned char)seq[i - 1]];
        r.prof = code == C2_INVALID_CODE ? 0u : sc.pk[code];
        out[C2_DIAG_ROW_PAD + i] = r;
    }
}

// ---- the packed (int16, two alignments per lane) fill of c2_align_diagp_kernel -------------------------------------------
// A reference is admitted if every DP value the banded fill can hold provably fits an int16 around the kernel's bias
// (C2_PK_BIAS = 16384) with room for the pointer-bit differences, and if the reference's finite sentinel
// min_score = gap_open * Li * Lj lies below every real value anyway (so that replacing it by -bias changes no comparison):
//   hi  = most a real cell value can be: max(0, max score) per diagonal step
//   lo  = how far below zero a real cell value can lie: every in-band cell is reached by a near-diagonal in-band path
//         (diagonal steps at the worst score, one gap run of at most band + 4 bases, a few opens)
// over alignments whose read is within `band` bases of the reference's length (wider differences never enter this kernel).
inline bool c2_pk_eligible(const char* seq, int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend, int band) {
    if (sc.pk.empty() || Li <= 0) return false;
    for (int i = 0; i < Li; ++i) if (sc.code_of_char[(unsigned char)seq[i]] >= 5) return false;      // pair-score tables exist for codes 0..4 (A C G T N)
    int64_t smax = 0, smin = 0, gabs = 0;
    for (int16_t v : sc.tbl) { smax = std::max<int64_t>(smax, v); smin = std::min<int64_t>(smin, v); }
    for (int i = 0; i <= Li; ++i) gabs = std::max<int64_t>(gabs, g32[i] < 0 ? -(int64_t)g32[i] : (int64_t)g32[i]);
    const int64_t go = gap_open < 0 ? -(int64_t)gap_open : gap_open, ge = gap_extend < 0 ? -(int64_t)gap_extend : gap_extend;
    if (go + gabs > 2000 || ge + gabs > 2000) return false;
    // `hi` below counts diagonal steps only: no gap step may ADD score (an incentive larger than |gap_extend| or |gap_open| would let
    // runs at several cut sites stack positive excursions on top of hi).  The diagonal chain asks for the same (geometry()); stated
    // here too so that the range proof stands on its own.
    {
        int64_t gpos = 0;
        for (int i = 0; i <= Li; ++i) gpos = std::max<int64_t>(gpos, g32[i]);
        if (std::max<int64_t>(gap_open, gap_extend) + gpos >= 0) return false;
    }
    const int64_t L = (int64_t)Li + band;
    const int64_t hi = smax * L, lo = L * (-smin) + 4 * go + (band + 4) * (ge + gabs);
    if (hi + lo > 14000) return false;
    if (gap_open >= 0 || (int64_t)(-gap_open) * Li * std::max<int64_t>(1, Li - band) <= hi + lo + 64) return false;   // the sentinel must be out of reach
    return true;
}

// ---- a read that lies on its reference's main diagonal: no matrix where the diagonal provably wins ------------------------------------
// A read as long as its reference that differs from it in k <= 2 places (k = 0: a byte-for-byte copy -- the unedited, error-free read, the
// commonest read of an amplicon run) needs no fill when every other path from (0, 0) to (L, L) provably scores LESS than the main diagonal:
// the pointer walk (CRISPResso2Align.pyx:338-421) then stays in state M from (L, L) to (0, 0) -- at every cell (i, i) the M predecessor
// strictly beats the I and J predecessors (a better or equal I / J prefix, completed along the diagonal, would be another path with a score
// >= the diagonal's), at (L, L) mScore strictly beats iScore and jScore -- so the aligned strings are the read and the reference themselves.
//
// Scores.  smax = the largest matrix entry (>= 0), soff = the largest entry that pairs a reference base with a DIFFERENT base of A C G T N,
// delta = the most one differing read base (of A C G T N) costs against the same base, S0 = sum_i s(ref_i, ref_i); the diagonal of a read
// with k differing bases scores >= S0 - k delta.  A gap column adds at most gcol = max(gap_open, gap_extend) + gpos (gpos = the largest
// incentive, >= 0): an opening pays gap_open -- gap_extend on row 0 / column 0 (pyx:153-176) and on the last row / column (pyx:234-317) --
// plus its row's incentive, an extension gap_extend (plus the incentive for an insertion).  gcol < 0 (c2_pk_eligible asks for it).
//
// Every other path has a >= 1 columns with a gap in the reference and a with a gap in the read (both sequences have L bases), and is of
// one of two kinds (a gap run cannot follow a gap run of the other kind without a paired column between them: iScore comes from mScore
// and iScore only, pyx:187-228):
//   A  at least one of its gap runs is opened INSIDE the matrix, at gap_open:  score <= smax (L - a) + (gap_open + gpos) + (2a - 1) gcol,
//      largest at a = 1:  boundA = smax (L - 1) + gap_open + gpos + gcol.
//   B  its runs are a leading one (row 0 or column 0) and a trailing one (last column or last row) only: the read shifted by a against the
//      reference, ONE diagonal d = +-a from end to end.  score <= P_d + 2 a gcol with P_d the pair scores along that diagonal,
//      P_d <= smax m_d + soff (L - a - m_d) for m_d columns of equal bytes on it (<= smax (L - a) in any case).
// k = 0:  S0 > smax (L - 1) + 2 gcol covers A and B at once (EDNAFULL, -20 / -2, incentive 1: 5 L against 5 L - 7).
// k = 1, 2:  S0 - k delta > boundA; for a >= 3 kind B is beaten by its smax (L - a) bound; for a = 1, 2 the kernel COUNTS the equal bytes
// of the four shifted diagonals (an upper bound: whole 16-byte blocks in the middle of the read counted, every other column taken as
// equal) and compares with mmax[k - 1][a - 1], the largest count the inequality allows -- a homopolymer run or a short tandem repeat
// shifts onto itself and is refused, an ordinary amplicon has a quarter of its bytes equal there and passes by a wide margin.
// (EDNAFULL: 5 L - 9 k against boundA = 5 L - 25 -- k <= 2 -- and against 5 m + (L - a - m) - 2 a on the shifted diagonals.)
// Differing read characters other than A C G T N (IUPAC codes, anything the matrix does not hold) are left to the launches: their status
// words and scores are theirs.  Sentinel-derived values stay below real ones (c2_pk_eligible's last condition), as in the fill kernels.
struct c2_diag_cert { int kmax; int mmax[4]; };      // kmax: -1 none, 0 exact copies, 1 / 2 that many differing bases; mmax[2 (k - 1) + (a - 1)]
inline c2_diag_cert c2_main_diagonal_certificate(const char* seq, int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend) {
    c2_diag_cert C;
    C.kmax = -1; C.mmax[0] = C.mmax[1] = C.mmax[2] = C.mmax[3] = -1;
    if (Li < 2 || sc.tbl.empty() || sc.n_codes <= 0) return C;
    const int nc = sc.n_codes;
    int64_t smax = 0, gpos = 0, s0 = 0;
    for (int16_t v : sc.tbl) smax = std::max<int64_t>(smax, v);
    for (int i = 0; i <= Li; ++i) gpos = std::max<int64_t>(gpos, g32[i]);
    const int64_t gcol = std::max<int64_t>(gap_open, gap_extend) + gpos;
    if (gcol >= 0) return C;
    bool present[5] = {false, false, false, false, false};
    for (int i = 0; i < Li; ++i) {
        const uint8_t c = sc.code_of_char[(unsigned char)seq[i]];
        if ((int)c >= nc || c >= 5) return C;
        present[c] = true;
        s0 += sc.tbl[(size_t)c * (size_t)nc + c];
    }
    const int64_t L = Li;
    if (!(s0 > smax * (L - 1) + 2 * gcol)) return C;
    C.kmax = 0;
    // k >= 1: a differing byte must be a differing CODE (one character per code among the reference's), soff below smax
    int64_t soff = INT64_MIN, delta = 0;
    for (int r = 0; r < 5 && r < nc; ++r) {
        if (!present[r]) continue;
        int chars = 0;
        for (int b = 0; b < 256; ++b) if (sc.code_of_char[b] == r) ++chars;
        if (chars != 1) return C;
        for (int c = 0; c < 5 && c < nc; ++c) {                    // (every character of a read the kernel lets through is one of the five: its bases
            if (c == r) continue;                                   //  equal the reference's, or differ from it and are checked)
            soff = std::max<int64_t>(soff, sc.tbl[(size_t)r * nc + c]);
            delta = std::max<int64_t>(delta, (int64_t)sc.tbl[(size_t)r * nc + r] - sc.tbl[(size_t)r * nc + c]);
        }
    }
    if (soff == INT64_MIN || smax - soff <= 0 || Li < 8) return C;
    const int64_t boundA = smax * (L - 1) + gap_open + gpos + gcol;
    for (int k = 1; k <= 2; ++k) {
        const int64_t lhs = s0 - k * delta;
        if (!(lhs > boundA) || !(lhs > smax * (L - 3) + 6 * gcol)) break;
        bool ok = true;
        for (int a = 1; a <= 2; ++a) {                              // the largest m with smax m + soff (L - a - m) + 2 a gcol < lhs
            const int64_t num = lhs - 2 * a * gcol - soff * (L - a) - 1;
            if (num < 0) { ok = false; break; }
            C.mmax[2 * (k - 1) + (a - 1)] = (int)std::min<int64_t>(num / (smax - soff), L);
        }
        if (!ok) break;
        C.kmax = k;
    }
    return C;
}

// ---- the packed fill with plain 32-bit adds (c2_align_diagp_kernel<NA, true>) --------------------------------------------------
// On gfx950 v_pk_add_i16 issues at half the rate of v_add_u32 (profiles/r03/valu_microbench4.txt: 4.15 against 2.3 cycles per
// wave64 instruction), and ten of a cell pair's ~25 instructions are such adds.  A 32-bit add of two int16 pairs is the packed add
// as long as no carry leaves the low half -- true when both halves of both operands are non-negative and stay below 2^15.  The
// values are (bias 16384 + score >= 0, or the exact 0 of "outside the band"); the constants become non-negative by a bias of
// beta per ANTI-DIAGONAL: a gap step (one anti-diagonal) adds its constant + beta, a diagonal step (two) its score + 2 beta.  Every
// value that enters one cell's comparisons lies on the same anti-diagonal, so all of them carry the same beta * (i + j): no
// comparison changes, H(Li, Lj) is read back minus beta * (Li + Lj).  beta = the largest magnitude among the gap constants of
// the admitted references (20 for the default gap_open -20).
// Admitted when, on top of c2_pk_eligible: gap_open <= gap_extend (the last-column correction b - a is then non-negative too) and
// the biased values stay inside int16: bias + hi + beta * (anti-diagonals of the largest admitted alignment) <= 32000, with the
// bias as small as c2_pk_eligible's margin allows (lo + hi + 2384 instead of the fixed 16384: 6,272 for a 250-bp amplicon with the
// default scoring, which admits amplicons up to ~410 bp at gap_open -20).
// Sentinel-derived values (chains that start from the 0 of a cell outside the band) grow by at most beta + maxS / 2 per anti-diagonal
// while every real value carries beta per anti-diagonal on top of 16384 - lo, so they stay below the real ones exactly as without
// the bias (c2_pk_eligible's margin); differences of two values of one cell are bounded by the largest value, < 2^15.
inline int c2_pk_beta_needed(int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend) {
    int64_t need = gap_extend < 0 ? -(int64_t)gap_extend : 0;
    for (int i = 1; i <= Li; ++i) {
        const int64_t open = (i == Li) ? gap_extend : gap_open;
        need = std::max<int64_t>(need, -(open + g32[i]));
        need = std::max<int64_t>(need, -((int64_t)gap_extend + g32[i]));
        need = std::max<int64_t>(need, -(open + g32[i - 1]));
    }
    int64_t smin = 0;
    for (int16_t v : sc.tbl) smin = std::min<int64_t>(smin, v);
    need = std::max<int64_t>(need, (-smin + 1) / 2);
    return need > 30000 ? 30000 : (int)need;
}
// hi / lo of c2_pk_eligible for one reference (the most a real cell value can be above, and lie below, the bias)
inline void c2_pk_range(int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend, int band, int64_t& hi, int64_t& lo) {
    int64_t smax = 0, smin = 0, gabs = 0;
    for (int16_t v : sc.tbl) { smax = std::max<int64_t>(smax, v); smin = std::min<int64_t>(smin, v); }
    for (int i = 0; i <= Li; ++i) gabs = std::max<int64_t>(gabs, g32[i] < 0 ? -(int64_t)g32[i] : (int64_t)g32[i]);
    const int64_t go = gap_open < 0 ? -(int64_t)gap_open : gap_open, ge = gap_extend < 0 ? -(int64_t)gap_extend : gap_extend;
    const int64_t L = (int64_t)Li + band;
    hi = smax * L; lo = L * (-smin) + 4 * go + (band + 4) * (ge + gabs);
}
// The bias the 32-bit-add variant needs for this reference: real values must stay above every sentinel-derived one by the margin
// c2_pk_eligible keeps (bias - lo >= hi + 2384); and whether, with the context's `bias` and `beta`, its largest biased value
// bias + hi + beta * (anti-diagonals of its longest admitted read, Lj <= Li + band) still fits an int16 with room (<= 32000).
inline int c2_pk_add32_bias_needed(int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend, int band) {
    int64_t hi, lo;
    c2_pk_range(Li, g32, sc, gap_open, gap_extend, band, hi, lo);
    const int64_t b = lo + hi + 2384;
    return b > 32000 ? 32000 : (int)((b + 63) & ~(int64_t)63);
}
inline bool c2_pk_add32_ok(int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend, int band, int beta, int bias) {
    if (beta <= 0 || gap_open > gap_extend) return false;
    int64_t hi, lo;
    c2_pk_range(Li, g32, sc, gap_open, gap_extend, band, hi, lo);
    const int64_t L = (int64_t)Li + band;
    return bias >= lo + hi + 2384 && (int64_t)bias + hi + (int64_t)beta * (2 * L + 8) <= 32000;
}

// Row records of the packed kernel for one reference, same indexing as c2_build_diag_rows: {a, b, c} duplicated into both
// int16 halves, prof = byte offset of the reference symbol's pair-score table (code * C2_PK_LUT_STRIDE; table 5 = zeros for rows 0, Li+1
// and the padding, so that cells outside the matrix add nothing -- as with the 32-bit records' empty score row).
inline void c2_build_diag_rows_pk(const char* seq, int Li, const int32_t* g32, const c2_scoring_tables& sc, int gap_open, int gap_extend,
                                  std::vector<c2_diag_row>& out, int beta = 0) {
    // beta > 0: the constants of the 32-bit-add variant (see c2_pk_add32_ok): every gap constant + beta, all non-negative
    out.assign((size_t)Li + 2 + 2 * C2_DIAG_ROW_PAD, c2_diag_row{0, 0, 0, 5u * C2_PK_LUT_STRIDE});
    auto dup = [](int x) { return (int32_t)(((uint32_t)x & 0xffffu) | ((uint32_t)x << 16)); };
    for (int i = 1; i <= Li; ++i) {
        const int open = (i == Li) ? gap_extend : gap_open;
        c2_diag_row r;
        r.a = dup(open + g32[i] + beta); r.b = dup(gap_extend + g32[i] + beta); r.c = dup(open + g32[i - 1] + beta);
        const uint8_t code = sc.code_of_char[(unsigned char)seq[i - 1]];
        r.prof = (code < 5 ? (uint32_t)code : 5u) * C2_PK_LUT_STRIDE;
        out[C2_DIAG_ROW_PAD + i] = r;
    }
}

// rows per lane for the systolic sweep: smallest R in 1..4 whose single pass covers max_li, else 4
inline int c2_choose_rows_per_lane(int max_li) {
    for (int R = 1; R <= 4; ++R) if (max_li <= 64 * R) return R;
    return 4;
}
