// Plain-data structures shared by the host side (c2_api_*.hip) and the kernels (c2_k_*.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "crispresso2_amd.h"   // c2_aln_record, C2_STATUS_* (public ABI)

// traceback / DP states (same numbering as the reference: CRISPResso2Align.pyx:23)
#define C2_ST_M 1
#define C2_ST_I 2
#define C2_ST_J 3

#define C2_MAX_CODES 32            // distinct score-matrix symbols (+1 shared "scores 0" code) the kernel keeps in LDS
#define C2_INVALID_CODE 255
#define C2_PTR_PAD 2               // halfword padding of one pointer column (breaks the 128-B bank stride)
#define C2_LANES 64
#define C2_DIAG_NEG (-(1 << 30))     // diagonal-band kernels: "no score yet" of the certificate's H(Li, Lj)
#define C2_DIAG_BIAS (1 << 30)       // diagonal-band kernels: added to every DP value, so that a cell outside the band reads as 0
#define C2_DIAG_STORE_LO 0           // diagonal-band kernel: lanes STORE_LO .. STORE_LO+STORE_N-1 keep their pointer words in LDS (all 64:
#define C2_DIAG_STORE_N 64           //   16 KB for 500 anti-diagonals; its 175 VGPRs allow 8 workgroups per CU, which 19 KB of LDS each still fit)
#define C2_DIAG_ROW_PAD 128          // zero row records in front of row 0 and behind row Li+1 of every reference's table
#define C2_DIAG_CODE_PAD 35          // zero column symbols in front of column 0 (multi-alignment kernel's LDS tables); PAD + 1 is a multiple of 4: column 1 is dword-aligned
#ifndef C2_TIER0_NA
#define C2_TIER0_NA 4               // alignments per wavefront in the first launch of the chain (4, or 5: lane groups of 12)
#endif
#define C2_TASK_CHUNK 4              // tasks a workgroup takes per atomic
#define C2_STATUS_NEED_FULL 64     // internal: banded launch could not finish the traceback; the full-plane launch overwrites the record
#define C2_STATUS_SHAPE 128        // c2_classify_records_kernel: a pair of strings the count route's classifier does not take (a gap in both strings
                                   // of one column, or an insertion column next to a deletion column: shapes the aligner never emits, a consensus of two reads can)

// Row constants of the diagonal-band kernel: I opened from M (a), I extended (b), J opened from M (c) -- gap_open, gap_extend
// and the gap incentives of the row folded in, last-row rule included -- and the packed score row of the reference base.
typedef struct c2_diag_row { int32_t a, b, c; uint32_t prof; } c2_diag_row;   // 16 bytes: one dwordx4 load

// Device-resident description of one reference amplicon.
typedef struct c2_dev_ref {
    const uint8_t* seq;           // Li bytes
    const int32_t* gap_incentive; // Li+1 (int64 input truncated to int32 exactly as the reference's int arithmetic does)
    const uint16_t* inc_prefix;   // Li+2: inc_prefix[x] = number of include idxs < x  (window membership and range hits)
    const c2_diag_row* diag_rows; // row 0 of Li+2 records (rows 0 and Li+1 are zero) with C2_DIAG_ROW_PAD zero records on either side; NULL when the scoring has no packed form
    int32_t len;                  // Li
    int32_t gap_incentive_max;    // max(0, max_i gap_incentive[i]); max(gap_open, gap_extend) + this bounds what one gap base adds to a score
    int32_t gap_incentive_last_pos; // gap_incentive[Li] > 0: insertions along the last row collect an incentive without paying an open
    int32_t max_char;             // largest byte of seq (a read character >= the matrix dimension is defined iff max_char * dim + it < dim * dim)
    int32_t pk_ok;                // admitted to the int16 fill of c2_align_diagp_kernel (c2_pk_eligible); its packed row table sits at the same index in diagpk_base
    int32_t first_incentive_pos;  // smallest i with gap_incentive[i] > 0 (the cut site the caller marked), -1: none.  A hint for c2_align_partition_kernel only
    int32_t diag_kmax;            // c2_main_diagonal_certificate: a read of this reference's length with at most this many differing bases (0: a byte-for-byte
                                  // copy) aligns to it along the main diagonal, provably -- no fill needed; -1: no such proof
    int32_t diag_mmax[4];         // ... for k = 1, 2 differing bases: [2 (k - 1) + (a - 1)] = the most equal bytes the diagonals +-a may hold
    int32_t reserved_pad;
    const uint32_t* seq2;         // seq as 2-bit codes ((c >> 1) & 3), 16 bases per word, base k of a word in bits 2k+1 .. 2k; two zero words in front of [0] and
                                  // two behind the last (c2_build_seq2).  What c2_align_partition_kernel's probe walks; NULL: it reads seq byte by byte
} c2_dev_ref;

// Kernel arguments for the fused align + traceback + classify kernel.
typedef struct c2_align_args {
    const uint8_t* reads;         // byte arena
    const uint64_t* offsets;      // n_reads+1 byte offsets into reads
    const uint16_t* ref_ids;      // per read reference id, or NULL
    const uint8_t* strands;       // per task: 1 => align the reverse complement, or NULL
    const c2_dev_ref* refs;
    const int16_t* score_tbl;     // n_codes x n_codes, [ref code][read code]
    const uint32_t* score_pk;     // n_codes words: signed 4-bit scores of read codes 0..7 for each ref code, or NULL
    const uint8_t* code_of_char;  // 256 entries -> code, C2_INVALID_CODE if ord >= matrix dim
    uint8_t* aln_read;            // n_tasks x aln_stride
    uint8_t* aln_ref;             // n_tasks x aln_stride
    c2_aln_record* records;       // n_tasks
    uint64_t n_tasks;
    uint32_t aln_stride;
    int32_t n_refs;
    int32_t all_refs;             // 1: task t = (read t / n_refs, ref t % n_refs); 0: one task per read
    int32_t n_codes;
    int32_t gap_open, gap_extend;
    int32_t max_lj;               // LDS plan: longest read of this launch
    int32_t max_passes;           // LDS plan: ceil(max Li / (64*R))
    int32_t band_lanes;           // banded row-strip kernel: lanes kept on each side of the main-diagonal lane
    int32_t max_score;            // diagonal-band kernel: largest entry of the score table (>= 0)
    int32_t max_li;               // diagonal-band kernel: LDS plan, longest reference
    int32_t reserved;
    int32_t pair_order;           // packed kernels, all-references batches without a task list: positions are mapped to tasks so that the two slots of a
                                  // lane group hold two consecutive reads against the SAME reference (position 2k*q + i -> read 2q + (i & 1), reference i >> 1)
    int32_t legacy;               // fused classification follows find_indels_substitutions_legacy (COREResources.pyx:190-315)
    uint32_t* fb_count;           // banded kernel: number of tasks whose traceback left the band ...
    uint32_t* fb_list;            // ... and their task indices (capacity n_tasks)
    uint32_t* un_count;           // packed kernels: tasks that found no partner for their lane group (another reference or read length next to them) ...
    uint32_t* un_list;            // ... go to this list, which a 32-bit kernel of the SAME band runs next (NULL: they go to fb_list)
    const uint32_t* task_list;    // full kernel, second launch: run only these tasks (NULL = tasks 0..n_tasks-1)
    const uint32_t* task_count;   // device-resident length of task_list
    unsigned long long* work_counter; // device counter the workgroups pull task chunks from; zero before every launch
    unsigned long long* phase_cycles; // optional: 4 counters of per-phase shader cycles (profiling), else NULL
    uint32_t* plane;              // multi-alignment diagonal kernel: pointer words in HBM/L2, plane_words_per_wg per workgroup
    uint32_t plane_words_per_wg;
    uint32_t pk_beta;             // packed kernels: 0, or the per-anti-diagonal bias of the 32-bit-add variant (c2_pk_add32_ok): its row tables, score pairs
                                  // and boundary values carry it, H(Li, Lj) comes back minus pk_beta * (Li + Lj)
    int32_t mat_dim;              // dimension of the reference's score matrix (CRISPResso2Align.pyx:212 reads the flat element ci * dim + cj)
    int32_t first_ext_code;       // codes >= this belong to read characters with ord >= mat_dim (c2_build_scoring); never valid in a reference
    uint32_t lut_code_lo, lut_code_hi;  // 8-entry byte tables indexed by (ch >> 1) & 7 (A 0, C 1, T 2, G 3, N 7): the score-table code of that base ...
    uint32_t lut_chr_lo, lut_chr_hi;    // ... and the base itself (0xFF where the entry is no base or its code is not a packed one): v_perm_b32 look-ups
    const struct c2_diag_row* diag_base;   // start of the buffer every reference's diag_rows points into
    const struct c2_diag_row* diagpk_base; // the packed kernels' row tables, same indexing: {a, b, c} as int16 pairs, prof = LDS offset of the symbol's pair-score table
    uint32_t pk_bias;             // the 32-bit-add variant's value bias (c2_pk_add32_bias_needed); the packed-add variant uses the constant C2_PK_BIAS
    int32_t list_gate;            // full-matrix kernel in list mode: > 0: run only if the list holds at most that many tasks; < 0: only if it holds more than -list_gate; 0: always
    uint32_t* diag_hints;         // c2_batch.diag_hints (or NULL): c2_align_partition_kernel leaves a C2_HINT_VALID word for every task it finishes itself
} c2_align_args;

// Kernel arguments for the per-call classifier (find_indels_substitutions / _legacy with full lists).
typedef struct c2_classify_args {
    const uint8_t* read_al;       // n bytes
    const uint8_t* ref_al;        // n bytes
    const int32_t* include_sorted;// sorted unique include idxs
    int32_t n;
    int32_t n_include;
    int32_t legacy;
    int32_t cap;                  // capacity (int32 entries) of each of the C2_LIST_COUNT output lists
    int32_t* lists;               // C2_LIST_COUNT x cap
    int32_t* list_len;            // C2_LIST_COUNT true lengths (may exceed cap: caller retries)
    int64_t* counts;              // insertion_n, deletion_n, substitution_n
} c2_classify_args;

// Batched classifier (c2_classify_lists_batch_kernel): one lane per alignment, two passes.
typedef struct c2_classify_batch_args {
    const uint8_t* aln_read;      // n x stride
    const uint8_t* aln_ref;
    const int32_t* lens;          // n: columns of each alignment (<= stride)
    const uint16_t* set_ids;      // n: which include set, or NULL (set 0)
    const int32_t* include_sorted;// the include sets, each sorted and unique, back to back
    const int64_t* include_off;   // n_sets + 1 offsets into include_sorted
    uint64_t n;
    uint32_t stride;
    int32_t legacy;
    int32_t pass;                 // 0: count list lengths, 1: write the lists
    int32_t reserved;
    int32_t* scratch_rp;          // pass 0: n x stride reference positions (the walk reads them back)
    int32_t* list_len;            // n x C2_LIST_COUNT: written by pass 0, read by pass 1
    const int64_t* list_off;      // n x C2_LIST_COUNT offsets into values (pass 1)
    int32_t* values;              // pass 1 output
    int64_t* counts;              // n x 3: insertion_n, deletion_n, substitution_n (pass 0)
} c2_classify_batch_args;

// Paired-read consensus (c2_consensus_pairs_kernel): one lane per read pair.
typedef struct c2_consensus_args {
    const uint8_t* s1; const uint8_t* f1;   // aligned read 1 / its aligned reference: n x stride
    const uint8_t* s2; const uint8_t* f2;   // aligned read 2 / its aligned reference
    const uint8_t* q1; const uint8_t* q2;   // quality strings (one character per non-gap read base): n x qstride
    const int32_t* n1; const int32_t* n2;   // alignment lengths
    const int32_t* lq1; const int32_t* lq2; // quality lengths
    const uint8_t* best1;                   // score_r1 >= score_r2
    uint64_t n;
    uint32_t stride, qstride, ostride, reserved;
    uint8_t* o_aln; uint8_t* o_ref; uint8_t* o_qual;   // n x ostride (ostride >= 2 * stride)
    int32_t* o_info;                        // n x 4: consensus length, quality length, matching columns, flags (1 caching_is_ok, 2 IndexError in the reference)
} c2_consensus_args;

// ---- seed test that picks the strand(s) of an alignment (CRISPRessoCORE.py:656-687) on the device ----
struct c2_strand_args {
    const uint8_t* reads; const uint64_t* offsets; uint64_t n_reads;
    const uint8_t* seed_blob;         // all seeds back to back
    const int32_t* seed_off;          // [n_refs][2 (forward, reverse complement)][max_seeds]: byte offset into seed_blob
    const int32_t* seed_len;          // same shape: length (0 = the empty seed: Python's '' in s is True)
    const int32_t* n_seeds;           // [n_refs]: seeds that take part (min(aln_seed_count, seeds of the reference))
    int32_t n_refs, max_seeds, seed_min, max_read_len;
    uint8_t* plan;                    // [n_reads][n_refs]: 0 forward only, 1 reverse complement only, 2 both
    int32_t seed_table;               // 1: every seed is at most C2_SEED_SLOT bytes and the table of all of them fits LDS behind the read rows --
                                      // the kernel copies them there once and compares four bytes at a time; 0: byte by byte from global memory
    int32_t reserved;
};
#define C2_SEED_SLOT 32u               // bytes of one seed in the LDS table (zero padded)
// LDS of c2_strand_plan_kernel: one row per wavefront (the read, + 16 bytes that the window reads may touch), then the seed table
#define c2_strand_row_bytes(max_read_len) ((uint32_t)(((max_read_len) + 15) & ~15) + 16u)

// ---- FASTQ framing + exact de-duplication on the device (the readline loop of process_fastq, CRISPRessoCORE.py:1820-1849) ----
// The text (no '\r' in it: the host falls back to its own parser otherwise) lies in HBM; lines end at '\n'; line k is ended by
// newline k (0-based); record r = lines 4r .. 4r + 3; its sequence is line 4r + 1: it starts behind newline 4r and ends at newline 4r + 1.
#define C2_FQ_LDS_BYTES 1024u          // dynamic LDS of the two framing kernels
#define C2_FQ_DEDUP_LDS_BYTES 2064u    // ... of the de-duplication kernel (its per-workgroup count table, c2_kernels.hip)
#define C2_FQ_TILE 16384u              // bytes of text per workgroup of the two framing kernels (256 threads x 64 bytes)
struct c2_fq_frame_args {
    const uint8_t* text;              // the whole text; bytes [0, hi) are resident
    uint64_t lo, hi;                  // this launch frames [lo, hi) in tiles of C2_FQ_TILE bytes from lo on; lo is a multiple of 16
    uint32_t* tile_newlines;          // per tile of [lo, hi): '\n' bytes               (count kernel out, lines kernel in as EXCLUSIVE prefix, 64-bit)
    uint32_t* tile_empty;             // per tile: '\n' bytes preceded by '\n' (or at text position 0): the empty lines `grep -c .` does not count
    const uint64_t* tile_base;        // lines kernel: number of newlines in front of every tile (from the start of the TEXT)
    uint32_t* flags;                  // bit 0: a '\r' was seen
    uint64_t* seq_start;              // per record: first byte of its sequence line
    uint64_t* seq_end;                // per record: the newline that ends it
    uint64_t n_records_cap;           // entries of the two arrays
    uint64_t* qual_start;             // (optional, both or neither) per record: first byte of its quality line ...
    uint64_t* qual_end;               // ... and the newline that ends it
};
// Paired input (process_paired_fastq's reading loop, CRISPRessoCORE.py:1309-1334) over two framed texts: record r of file 1 with record r of file 2.
// lengths kernel (one thread per record): the four lines str.strip()ped -> s1 / q1 / s2 / q2 [r] = start << 24 | length, key_len[r] = len(seq1) + 1 +
// len(seq2), qual_len[r] likewise.  write kernel (one wavefront per record): key_out[key_off[r] ..) = seq1 + '+' + reverse_complement(seq2),
// qual_out[qual_off[r] ..) = qual1 + ' ' + qual2[::-1]; flags bit 0: a line of 2^24 bytes or more / a text of 2^40 or more, bit 1: a character
// of seq2 outside ACGTN_- (either case; CRISPRessoShared.py:399-403's KeyError).
struct c2_fq_pair_args {
    const uint8_t* text1; const uint8_t* text2;
    const uint64_t* seq_start1; const uint64_t* seq_end1; const uint64_t* qual_start1; const uint64_t* qual_end1;
    const uint64_t* seq_start2; const uint64_t* seq_end2; const uint64_t* qual_start2; const uint64_t* qual_end2;
    uint64_t n;
    unsigned long long* s1; unsigned long long* q1; unsigned long long* s2; unsigned long long* q2;
    int64_t* key_len; int64_t* qual_len;               // lengths kernel out
    const int64_t* key_off; const int64_t* qual_off;   // write kernel in (exclusive prefix sums of the lengths)
    uint8_t* key_out; uint8_t* qual_out;
    uint32_t* flags;
};
// one wavefront per record (grid-stride): strip() the sequence line, look it up in / add it to the table
struct c2_fq_dedup_args {
    const uint8_t* text;
    const uint64_t* seq_start; const uint64_t* seq_end;
    const uint64_t* range;            // device memory: records [range[0], range[1]) -- written by the stream that counted the newlines, so
                                      // that no host round trip sits between the framing of a chunk and its de-duplication
    uint64_t n_records_cap;           // entries of the per-record arrays (a range beyond it sets flag bit 2)
    unsigned long long* slots;        // open addressing: 0 = empty, else (stripped start << 24 | length) of the key's representative
    uint64_t mask;                    // slots - 1
    uint32_t* count;                  // per slot: occurrences
    uint32_t* first;                  // per slot: smallest record number (first-seen order); initialised to 0xffffffff
    uint32_t* slot_of;                // per record: its slot
    unsigned long long* rinfo;        // per record: (stripped start << 24 | length)
    uint32_t* flags;                  // bit 1: a sequence line of 2^24 bytes or more, or a text position beyond 2^40; bit 2: more records than
                                      // the arrays hold (the host falls back on any of them, and when stats[0] exceeds half the table)
    uint32_t* stats;                  // [0] += keys created by this launch, [1] = max(., their lengths), [2] += 1 for the empty key
};
// out[out_offsets[i] ..] = the bytes info[records ? records[i] : i] names (start << 24 | length)
struct c2_fq_gather_args {
    const uint8_t* text;
    const unsigned long long* info;
    const int64_t* records;           // may be null
    const int64_t* out_offsets;       // n + 1 byte offsets into out
    uint8_t* out;
    uint64_t n;
};

// which unique read equals reverse_complement(unique read i) (the count merge of CRISPRessoCORE.py:3970-3975; CRISPRessoShared.py:399-403:
// upper-cased first, ACGTN_- only): looked up in the table the de-duplication built -- one wavefront per unique read
struct c2_fq_rc_args {
    const uint8_t* text;
    const unsigned long long* info;   // per record: start << 24 | length
    const int64_t* records;           // the unique reads: record numbers
    uint64_t n;
    const unsigned long long* slots; uint64_t mask;
    int32_t* partner_slot;            // per unique read: the slot of the key that equals its reverse complement, -1: none (or a character outside the alphabet)
};

// ---- per-amplicon count tensor (what CRISPRessoCORE.py:3865-3901 keeps per reference and :4016-4115 fills) ----
// One int64 block per reference: C2_CNT_VECTORS vectors of (lmax + 1) entries, then C2_CNT_SCALARS scalars,
// then C2_CNT_HISTS histograms of hl entries.  crispresso2_amd/counts.py names the slices.
enum {
    C2_V_ALL_INSERTION = 0, C2_V_ALL_INSERTION_LEFT, C2_V_ALL_DELETION, C2_V_ALL_SUBSTITUTION,
    C2_V_INSERTION, C2_V_DELETION, C2_V_SUBSTITUTION,
    C2_V_ALL_SUB_BASE_A, C2_V_ALL_SUB_BASE_C, C2_V_ALL_SUB_BASE_G, C2_V_ALL_SUB_BASE_T, C2_V_ALL_SUB_BASE_N,
    C2_V_BASE_A, C2_V_BASE_C, C2_V_BASE_G, C2_V_BASE_T, C2_V_BASE_N, C2_V_BASE_GAP,
    C2_V_INSERTION_LENGTH, C2_V_DELETION_LENGTH,
    C2_CNT_VECTORS
};
enum {
    C2_S_TOTAL = 0, C2_S_MODIFIED, C2_S_UNMODIFIED, C2_S_DISCARDED, C2_S_INSERTION, C2_S_DELETION, C2_S_SUBSTITUTION,
    C2_S_ONLY_INSERTION, C2_S_ONLY_DELETION, C2_S_ONLY_SUBSTITUTION, C2_S_INSERTION_AND_DELETION,
    C2_S_INSERTION_AND_SUBSTITUTION, C2_S_DELETION_AND_SUBSTITUTION, C2_S_INSERTION_AND_DELETION_AND_SUBSTITUTION,
    C2_S_N_GLOBAL_SUBS, C2_S_N_SUBS_OUTSIDE_WINDOW, C2_S_N_MODS_IN_WINDOW, C2_S_N_MODS_OUTSIDE_WINDOW,
    C2_S_N_READS_IRREGULAR_ENDS, C2_S_ALIGNMENTS_COUNTED, C2_S_RESERVED0, C2_S_RESERVED1, C2_S_RESERVED2, C2_S_RESERVED3,
    C2_CNT_SCALARS
};
enum { C2_H_INSERTED_N = 0, C2_H_DELETED_N, C2_H_SUBSTITUTED_N, C2_H_EFFECTIVE_LEN, C2_CNT_HISTS };

// count kernel geometry: C2_CNT_WAVES wavefronts share one LDS accumulator block; after the block come
// C2_CNT_CTL_INTS control words and the current reference's inc_prefix (lmax + 2 uint16)
#ifndef C2_CNT_WAVES
#define C2_CNT_WAVES 8                 // wavefronts that share one LDS block
#endif
#ifndef C2_CNT_OCC
#define C2_CNT_OCC 8                   // waves per SIMD the count kernels are compiled for (C2_CNT_WAVES x resident workgroups / 4): the kernel waits for
                                       // memory round trips, one alignment per wave at a time -- 8 waves at 64 VGPRs (some spilled) beat 5 at 96 by 15 %
                                       // (profiles/r03/ab_count_occupancy.txt)
#endif
#ifndef C2_CNT_TASKS_PER_WAVE
#define C2_CNT_TASKS_PER_WAVE 32       // records a wavefront holds per chunk, one per lane (32 or 64)
#endif
#define C2_CNT_CTL_BASE_INTS 96     // >= 16 + 4 * C2_CNT_WAVES
#define C2_CNT_CTL_INTS (C2_CNT_CTL_BASE_INTS + C2_CNT_WAVES * C2_CNT_TASKS_PER_WAVE)   // + per task of a chunk: the part of a heavy weight that is still to be added
#define C2_CNT_LOAD_BUDGET (1u << 30) // sum of weight x alignment length an LDS block may take between two flushes (its entries are int32)
// Staging: the strings of C2_CNT_STAGE alignments per wavefront are copied to LDS by the memory system itself (global_load_lds: no
// register holds them) BEFORE the first of them is walked -- so a wavefront has C2_CNT_STAGE alignments' loads in flight instead of one,
// which is what the kernel waits for (profiles/r03: 65 % of its wave cycles).  A slot holds C2_CNT_STAGE_ROW columns of both strings;
// longer alignments are walked window by window.
#ifndef C2_CNT_STAGE
#define C2_CNT_STAGE 1
#endif
#ifndef C2_CNT_SWAR
#define C2_CNT_SWAR 1                  // alignments with gaps that fit a slot: eight columns per lane in one pass (0: the 64-column chunk walk for all of them)
#endif
#ifndef C2_CNT_GROUPED
#define C2_CNT_GROUPED 1               // gap-free alignments of at most 256 columns: eight at a time, eight lanes each (0: one at a time, staged)
#endif
#ifndef C2_CNT_STAGE_ROW
#define C2_CNT_STAGE_ROW 320           // a multiple of 64: a 250-bp read with up to 70 gap columns is walked in one pass (eight columns per lane)
#endif
#define C2_CNT_STAGE_BYTES ((size_t)C2_CNT_WAVES * C2_CNT_STAGE * 2u * C2_CNT_STAGE_ROW)
// LDS of the variant whose accumulator block lives in HBM: the difference array, the control words, the window prefix, the staging slots
static inline size_t c2_count_lds_tail_bytes(int lmax) {               // what follows the block: cov, dcov, control words, inc_prefix (padded to 16), staging
    return ((2 * ((size_t)lmax + 1) + C2_CNT_CTL_INTS) * sizeof(int) + (((size_t)lmax + 2 + 1) / 2) * 4 + 15) / 16 * 16 + C2_CNT_STAGE_BYTES;
}
static inline size_t c2_count_lds_bytes_hbm(int lmax) { return c2_count_lds_tail_bytes(lmax); }
static inline size_t c2_count_lds_bytes(size_t per_ref, int lmax) {      // block + the LDS-only `cov` vector (lmax + 1) + control words + inc_prefix + staging
    return per_ref * sizeof(int) + c2_count_lds_tail_bytes(lmax);
}

// packed (int16) fill: byte stride between the pair-score tables of two reference symbols in LDS.  65 dwords, not 64: lanes that hold the
// same pair of read symbols against DIFFERENT reference symbols -- the usual case, the reads resemble each other -- then hit different banks
#define C2_PK_LUT_STRIDE 260u

#define C2_CNT_FLAG_IGNORE_SUBSTITUTIONS 1
#define C2_CNT_FLAG_IGNORE_INSERTIONS 2
#define C2_CNT_FLAG_IGNORE_DELETIONS 4
#define C2_CNT_FLAG_DISCARD_INDEL_READS 8
#define C2_CNT_FLAG_ALL_REFS_LAYOUT 16     // the tasks are an all-references batch: task = read * n_refs + reference
#define C2_CNT_FLAG_LEGACY 32              // find_indels_substitutions_legacy's positions

typedef struct c2_count_args {
    const uint8_t* aln_read;      // n_tasks x aln_stride (outputs of the align kernel)
    const uint8_t* aln_ref;
    const c2_aln_record* records;
    const uint32_t* weights;      // per task read multiplicity; 0 = not counted; NULL = 1
    const uint16_t* min_matches;  // n_refs x (max_t + 1): smallest `matches` whose score exceeds refs[name]['min_aln_score'], or NULL
    const c2_dev_ref* refs;
    long long* counts;            // n_refs x per_ref int64, accumulated into (caller zeroes)
    unsigned long long* work_counter;
    uint64_t n_tasks;
    uint32_t aln_stride;
    int32_t n_refs;
    int32_t lmax;                 // longest reference
    int32_t hl;                   // histogram length (>= longest reference + longest read + 1)
    int32_t max_t;                // longest alignment the min_matches table covers
    int32_t flags;                // C2_CNT_FLAG_*
    const uint32_t* order;        // optional: tasks grouped by reference (position -> task), else NULL = task order
    int32_t* block_scratch;       // c2_count_vectors_hbm_kernel: one int32 accumulator block per workgroup in HBM (amplicons whose block does not fit LDS)
    uint64_t block_ints;          // ... its size in ints
    const uint32_t* hints;        // optional (one reference only): c2_batch.diag_hints of the batch -- a task with a usable hint is counted by c2_count_hinted_kernel
                                  // from the hint words, c2_count_vectors_kernel skips it
    uint32_t* rest_list;          // c2_count_hinted_kernel: the tasks it does NOT take (and whose weight is not 0), densely -- c2_count_vectors_kernel then runs over
    uint32_t* rest_count;         // this list (as its `order`) instead of looking at every task; rest_count: its device-resident length
    const uint32_t* n_tasks_dev;  // c2_count_vectors_kernel: if set, the number of positions to process is read from here (the list above) instead of n_tasks
    const uint32_t* ref_ends;     // c2_count_hinted_kernel with several references (`order` groups the tasks by reference): position behind the last task of every reference;
    uint32_t hint_gx;             // ... and the workgroups per reference: workgroup b works on reference b / hint_gx
} c2_count_args;
// LDS of c2_count_hinted_kernel: the int32 position vectors, histograms and two difference arrays; 16 + C2_COUNT_SCALARS 64-bit totals; inc_prefix; scan carries
static inline size_t c2_count_hinted_lds_bytes(int lmax, int hl) {
    const size_t ints = (size_t)C2_COUNT_VECTORS * (size_t)(lmax + 1) + (size_t)C2_COUNT_HISTS * (size_t)hl + 2u * (size_t)(lmax + 1);
    return (ints * sizeof(int32_t) + 15) / 16 * 16 + (16 + C2_COUNT_SCALARS) * sizeof(uint64_t) + (size_t)((lmax + 2 + 7) / 8) * 8 * 2 + 64 + 16 * 256 * sizeof(uint32_t) + (size_t)((lmax + 1 + 15) / 16) * 16;   // (+ lrest: C2_HCNT_FLUSH_ROUNDS x 256; + the reference's bytes)
}

// ---- best-reference selection on the device (CRISPRessoCORE.py:683, :697-707, :779-785) ----
// One lane per read over the k alignments of that read (all-references layout: task = read * n_refs + ref).
#define C2_SEL_MODE_DROP_AMBIGUOUS 0      // default: a read that ties between amplicons counts for none (class AMBIGUOUS)
#define C2_SEL_MODE_FIRST 1               // --assign_ambiguous_alignments_to_first_reference
#define C2_SEL_MODE_EXPAND 2              // --expand_ambiguous_alignments
#define C2_SEL_FLAG_ALIGNED 1
#define C2_SEL_FLAG_AMBIGUOUS 2
enum { C2_SEL_N_COMPUTED_ALN = 0, C2_SEL_N_COMPUTED_NOTALN, C2_SEL_N_CACHED_ALN, C2_SEL_N_CACHED_NOTALN, C2_SEL_N_GLOBAL_SUBS,
       C2_SEL_N_SUBS_OUTSIDE_WINDOW, C2_SEL_N_MODS_IN_WINDOW, C2_SEL_N_MODS_OUTSIDE_WINDOW, C2_SEL_N_READS_IRREGULAR_ENDS,
       C2_SEL_N_BAD_STATUS, C2_SEL_FIRST_BAD_STATUS, C2_SEL_STATS };
typedef struct c2_select_args {
    const c2_aln_record* records;   // n_reads x n_refs: the alignments on the strand the seeds asked for
    const c2_aln_record* records2;  // reverse-complement alignments of the (read, reference) pairs aligned on both strands, or NULL
    const int32_t* slot2;           // n_reads x n_refs: index into records2, -1 = none; NULL when records2 is NULL
    const uint32_t* min_mscore;     // n_refs: smallest milli-score (1000 * score) that exceeds refs[name]['min_aln_score']
    const uint32_t* raw_counts;     // per read multiplicity before the reverse-complement merge (aln_stats), NULL = 1
    const uint32_t* counts;         // per read multiplicity the weights are formed from, NULL = 1
    unsigned long long* member;     // out, per read: bit r = reference r is a best match (may be NULL)
    unsigned long long* use2;       // out, per read: bit r = the reverse-complement alignment won against reference r (may be NULL)
    uint8_t* flags;                 // out, per read: C2_SEL_FLAG_* (may be NULL)
    uint32_t* weights;              // out, n_reads x n_refs: multiplicity with which task (read, r) enters the count pass (may be NULL)
    uint32_t* weights2;             // out, one per records2 entry (may be NULL)
    unsigned long long* stats;      // out, C2_SEL_STATS sums (aln_stats of process_fastq, CRISPRessoCORE.py:1974-1979), may be NULL
    uint64_t n_reads;
    int32_t n_refs;
    int32_t mode;                   // C2_SEL_MODE_*
} c2_select_args;

// ---- allele frequency table on the device (c2_k_alleles.hip; CRISPRessoCORE.py:3964-4010, :4298-4303, :4498-4530) ----
// where the two strings of a table row live: batch 1 (row = read * n_refs + reference) or, with bit 31 of `src` set, batch 2
struct c2_allele_strings {
    const uint8_t* a1; const uint8_t* f1; const uint8_t* a2; const uint8_t* f2;
    uint32_t stride1, stride2;
};
struct c2_allele_jobs_args {
    c2_allele_src S;
    const uint64_t* offsets;          // NULL: pass 1, njobs[read] = rows this read gives; else pass 2: its first row
    uint32_t* njobs;
    c2_allele_row* rows;
};
// the text of one table line: aligned \t reference \t label \t status \t n_deleted \t n_inserted \t n_mutated \t #Reads \t %Reads [\t dsODN \t fragment] \n
struct c2_allele_text_args {
    c2_allele_strings X;
    const c2_allele_row* rows; const uint32_t* order;      // sorted position q -> row
    uint64_t m;                       // rows of the table
    const uint32_t* run_start;        // n_runs ascending positions: rows [run_start[u], run_start[u + 1]) share one #Reads value
    const uint32_t* pct_off; const uint8_t* pct_len; const uint8_t* pct_blob;   // ... and its %Reads text
    uint32_t n_runs;
    const uint32_t* label_off; const uint8_t* label_blob;  // label l = label_blob[label_off[l] .. label_off[l + 1])
    const uint8_t* probe_bits;        // per sorted position: bit 0 contains dsODN, bit 1 contains the fragment; NULL: no such columns
    uint32_t* lengths;                // lengths pass: bytes of line q
    const uint64_t* offsets;          // emit pass: first byte of line q in the whole text
    uint64_t q0, q1;                  // emit pass: lines [q0, q1) go to out[offsets[q] - offsets[q0]]
    uint8_t* out;
};
struct c2_allele_probe_args {
    c2_allele_strings X;
    const c2_allele_row* rows; const uint32_t* order; uint64_t m;
    const uint8_t* probe_blob; uint32_t probe_off[5];     // four probes back to back
    uint8_t* probe_bits;
};
struct c2_allele_fetch_args {
    c2_allele_strings X;
    const c2_allele_row* rows; const uint32_t* order; uint64_t m;
    c2_allele_row* out_rows; uint8_t* out_a; uint8_t* out_f; uint32_t stride;
};
// around-cut windows: key = window of the aligned read (W bytes, zero padded), window of the reference (W), Unedited, n_deleted, n_inserted,
// n_mutated (big endian) -- byte order of the key = order of the reference's group key; key_bytes a multiple of 8
struct c2_allele_window_args {
    c2_allele_strings X;
    const c2_allele_row* rows; const uint32_t* order; uint64_t m;
    int32_t label, cut_point, left, right;
    uint32_t W, key_bytes;
    const uint64_t* sub_index;        // NULL: flag pass (flag[q] = row q carries the label); else exclusive scan of the flags
    uint32_t* flag;
    uint8_t* keys;                    // [subset][key_bytes]
    uint32_t* sub_reads;              // [subset] #Reads of the row
    uint32_t* error;                  // set when a row's reference string holds no base cut_point
};
struct c2_allele_group_args {
    const uint8_t* keys; uint32_t key_bytes; uint64_t ms;
    const uint32_t* perm;             // sorted position j -> subset index
    uint32_t* head;                   // pass 0: head[j] = key differs from the one before
    const uint64_t* head_scan;        // pass 1: exclusive scan of head -> group of position j = head_scan[j] + head[j] - 1
    uint32_t* gid;                    // pass 1: per subset index
    uint8_t* gkeys;                   // pass 1: [group][key_bytes], in key order
};

// ---- records of GIVEN aligned strings (c2_classify_records_kernel): what the fused classifier of the align kernels writes for the
// alignments it emits, for strings that come from somewhere else -- the consensus alignments of read pairs (c2_consensus_pairs_kernel)
struct c2_records_args {
    const uint8_t* aln_read; const uint8_t* aln_ref;   // n x stride
    const int32_t* info;              // n x 4 (c2_consensus_pairs_kernel's): [0] columns, [2] matching columns, [3] flags
    const uint16_t* ref_ids;          // per item, or NULL: all-references layout (item = unit * n_refs + reference)
    const uint8_t* strands;           // per item (copied into the record), or NULL
    const c2_dev_ref* refs;
    c2_aln_record* records;
    uint64_t n;
    uint32_t stride;
    int32_t n_refs;
    int32_t legacy;
    int32_t reserved;
};
