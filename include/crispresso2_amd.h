/*
 * crispresso2_amd.h -- C ABI of libcrispresso2_amd.so
 *
 * MI355X (gfx950) implementation of CRISPResso2's per-read align + classify hot path.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository, pinellolab/CRISPResso2 v2.3.4).  Plain pointers and sizes only;
 * nothing here depends on Python or torch.  INTEGRATION.md shows the ctypes binding the
 * reference side uses (crispresso2_amd/_native.py is that binding).
 *
 * Conventions
 *   - functions return 0 on success, a negative C2_E_* code on failure; c2_last_error()
 *     returns a human-readable message for the last failure on that context
 *     (c2_last_error(NULL) for failures of c2_create itself).
 *   - "device pointer" = address in the HBM of the context's GPU; "host pointer" = ordinary
 *     process memory.  All arguments are borrowed for the duration of the call only.
 *   - every computation runs on the GPU.  There is no CPU fallback: without a usable
 *     device c2_create fails.
 */
#ifndef CRISPRESSO2_AMD_H
#define CRISPRESSO2_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: struct c2_batch grew by `min_read_len` (round 4).  A caller built against version 1 passes the shorter struct: check c2_abi_version()
 * against the header you compiled with before the first batch call.  Zero-initialise every struct c2_batch you fill in; fields added later
 * are hints whose zero means "not known". */
#define C2_ABI_VERSION 3
/* 3: struct c2_batch grew by `diag_hints` (round 6), c2_partition_info writes 7 class counts (since round 5; it wrote 5 under ABI 1), new entry
 * c2_count_vectors_hinted_device. */

/* error codes */
#define C2_E_INVALID   -1   /* bad argument */
#define C2_E_DEVICE    -2   /* HIP runtime error (no device, launch failure, ...) */
#define C2_E_NOMEM     -3
#define C2_E_STATE     -4   /* scoring / references not set */
#define C2_E_TOO_LARGE -5   /* problem does not fit the kernel's LDS plan (see DESIGN.md limits) */
#define C2_E_OVERFLOW  -6   /* caller's output buffer too small; required sizes are reported */

/* per-alignment status bits, c2_aln_record.status (0 = valid alignment) */
#define C2_STATUS_EMPTY         1
#define C2_STATUS_OOB_CHAR      2
#define C2_STATUS_SENTINEL_PATH 4
#define C2_STATUS_UNINIT_PTR    8
#define C2_STATUS_RC_CHAR       16
#define C2_STATUS_TOO_LONG      32

typedef struct c2_ctx c2_ctx;

/* One record per executed (read, reference) alignment; 32 bytes; layout is ABI. */
typedef struct c2_aln_record {
    uint16_t aln_len;              /* columns; length of both aligned strings */
    uint16_t matches;              /* score = round(100*matches/float(aln_len), 3): CRISPResso2Align.pyx:433-434 */
    uint16_t insertion_n;          /* CRISPRessoCOREResources.pyx:165 */
    uint16_t deletion_n;           /* pyx:164 */
    uint16_t substitution_n;       /* pyx:163 */
    uint16_t all_insertion_events; /* len(all_insertion_left_positions) */
    uint16_t win_insertion_events; /* len(insertion_sizes) */
    uint16_t all_deletion_events;  /* len(all_deletion_coordinates) */
    uint16_t win_deletion_events;  /* len(deletion_coordinates) */
    uint16_t all_deletion_bases;   /* len(all_deletion_positions) */
    uint16_t all_substitutions;    /* len(all_substitution_positions) */
    uint8_t  irregular_ends;       /* CRISPRessoCORE.py:729-733 */
    uint8_t  status;               /* C2_STATUS_* */
    uint8_t  strand;               /* 0 '+', 1 '-' */
    uint8_t  reserved0;
    uint16_t ref_id;
    uint32_t reserved2;
} c2_aln_record;
#ifdef __cplusplus
static_assert(sizeof(c2_aln_record) == 32, "c2_aln_record is ABI: 32 bytes");
#endif

/* ---- lifecycle ------------------------------------------------------------------------ */

/* Number of visible GPUs (0 if none / HIP unusable). */
int c2_device_count(void);

/* Create a context bound to GPU `device` (one context per process per GPU is the intended
 * use: CRISPRessoMultiProcessing's process pool becomes one process per GPU). */
int c2_create(int device, c2_ctx** out);
void c2_destroy(c2_ctx* ctx);
const char* c2_last_error(const c2_ctx* ctx);
int c2_abi_version(void);

/* ---- scoring and references ----------------------------------------------------------- */

/* Score matrix as produced by CRISPResso2Align.read_matrix / make_matrix (pyx:33-99):
 * row-major int64[mat_dim][mat_dim] indexed [ord(ref char)][ord(read char)] (pyx:212),
 * plus the two gap parameters of global_align (pyx:104-105). */
int c2_set_scoring(c2_ctx* ctx, const int64_t* matrix, int32_t mat_dim, int32_t gap_open, int32_t gap_extend);

/* The reference amplicons of a run: what CRISPRessoCORE.py:3236-3268 keeps in refs[name]
 * ('sequence', 'gap_incentive' int64[len+1], 'include_idxs').  include_idx[r] may be NULL
 * when n_include[r] == 0. */
int c2_set_refs(c2_ctx* ctx, int32_t n_refs, const char* const* seqs, const int32_t* lens,
                const int64_t* const* gap_incentives,
                const int32_t* const* include_idx, const int32_t* n_include);

/* ---- batch path: replaces the per-read loop of CRISPRessoCORE.py:1957-1981 / :1226-1232 - */

typedef struct c2_batch {
    uint64_t n_reads;
    const uint8_t*  reads;     /* byte arena holding the read sequences back to back */
    const uint64_t* offsets;   /* n_reads+1 byte offsets; read k = reads[offsets[k] .. offsets[k+1]) */
    const uint16_t* ref_ids;   /* per read amplicon id (CRISPRessoPooled semantics) or NULL = reference 0 */
    const uint8_t*  strands;   /* per task: 1 = align reverse_complement(read) (CRISPRessoCORE.py:672), or NULL */
    int32_t all_refs;          /* 1: align every read to every reference (CRISPRessoCORE.py:653);
                                  task t = read t / n_refs against reference t % n_refs; ref_ids ignored */
    int32_t max_read_len;      /* longest read of the batch; sizes the LDS plan.  Device path: required
                                  (0 = derive the bound aln_stride - longest reference); host path: ignored */
    /* outputs, n_tasks = n_reads * (all_refs ? n_refs : 1) entries each */
    uint8_t* aln_read;         /* n_tasks x aln_stride: aligned read  (global_align()[0]) */
    uint8_t* aln_ref;          /* n_tasks x aln_stride: aligned reference (global_align()[1]) */
    uint32_t aln_stride;       /* >= longest read + longest reference */
    uint32_t flags;            /* C2_BATCH_*: bit 0 = the record's window counts follow find_indels_substitutions_legacy
                                  (--use_legacy_insertion_quantification, CRISPRessoCORE.py:721-722) */
    c2_aln_record* records;    /* n_tasks */
    int32_t min_read_len;      /* shortest read of the batch, or 0 = not known.  A hint that never changes results: a band tier of the launch chain
                                  that no read of [min_read_len, max_read_len] against any reference can use (|len(ref) - len(read)| outside its band, e.g.
                                  150-bp mates against a 250-bp amplicon) is not launched instead of being passed through task by task */
    uint32_t* diag_hints;      /* optional output (DEVICE path only; NULL = none), FOUR words per task (n_tasks x 4): zeros, or a summary of the alignment that the count pass
                                  can use instead of reading the two aligned strings back (c2_count_vectors_hinted_device):
                                    C2_HINT_VALID   the read lies on its reference's main diagonal with at most two differing bases -- what c2_align_partition_kernel
                                                    finishes without a matrix (round 5); word 0 only
                                    C2_HINT_GAPPED  an alignment of at most five runs (M / insertion / deletion) and at most three differing columns, as the lane-group
                                                    epilogue of the band kernels leaves it (round 6; not under the legacy classifier)
                                  A hint restates the aligned strings of its task (it IS derived from them).  Every word is written (0 where there is nothing to say). */
} c2_batch;
#define C2_BATCH_LEGACY_CLASSIFIER 1u
/* a hint word: bit 31 valid | bits 24..25 differing bases k (0, 1, 2) | first: position bits 0..8, read base bits 9..11 | second: position bits 12..20, read base
 * bits 21..23; read bases as codes 0..4 = A C G T N.  The alignment: aln_len = matches + k = len(reference) = len(read), both strings without a gap. */
#define C2_HINT_VALID 0x80000000u
/* C2_HINT_GAPPED: word 0 bits 0..2 runs n (1..5), bits 3..4 differing columns m (0..3), run 0 bits 5..15, run 1 bits 16..26; word 1 run 2 bits 0..10, run 3 bits
 * 11..21; word 2 run 4 bits 0..10, differing column 0 bits 11..22; word 3 differing columns 1 and 2 (bits 0..11, 12..23).  A run, in the order of the strings:
 * state (1 M, 2 insertion = gap in the reference string, 3 deletion = gap in the read's) | length << 2.  A differing column (both strings have a base there):
 * reference index | read base code << 9 (codes: (ch >> 1) & 7 -- A 0, C 1, T 2, G 3, N 7). */
#define C2_HINT_GAPPED 0x40000000u

/* All pointers in `b` are DEVICE pointers; the launch is enqueued on `hip_stream` (a hipStream_t; NULL is HIP's
 * default stream, exactly as in hipLaunchKernelGGL) and the call returns without waiting. */
int c2_align_classify_batch_device(c2_ctx* ctx, const c2_batch* b, void* hip_stream);

/* All pointers in `b` are HOST pointers; stages through the context's device buffers
 * (H2D, launch, D2H) and returns when the results are in host memory. */
int c2_align_classify_batch_host(c2_ctx* ctx, const c2_batch* b);

/* Block until everything enqueued on `hip_stream` and on the context's own stream (host-path staging) is done. */
int c2_synchronize(c2_ctx* ctx, void* hip_stream);

/* Kernel timing with HIP events on the launch stream: enable, run launches, then read the
 * accumulated kernel milliseconds and launch count (synchronises the recorded events). */
int c2_timing_enable(c2_ctx* ctx, int on);
int c2_timing_read(c2_ctx* ctx, double* total_ms, int64_t* launches, int reset);
/* Same, and also the summed time of the FIRST kernel of every batch's launch chain (the kernel that sees every task). */
int c2_timing_read_split(c2_ctx* ctx, double* total_ms, double* first_kernel_ms, int64_t* launches, int reset);

/* Launch geometry chosen for the current references / longest read (for DESIGN/bench reporting). */
int c2_launch_info(c2_ctx* ctx, int32_t max_read_len, int32_t* rows_per_lane, int32_t* passes,
                   int32_t* lds_bytes, int32_t* workgroups_per_cu, int32_t* compute_units);

/* ---- per-amplicon count tensor: the device side of CRISPRessoCORE.py:3964-4115 (+ aln_stats :1974-1979) ------------
 * One int64 block per reference: C2_COUNT_VECTORS vectors of (lmax+1) entries [lmax = longest reference], then
 * C2_COUNT_SCALARS scalars, then C2_COUNT_HISTS histograms of `hl` entries (hl >= longest reference + longest read + 1).
 * crispresso2_amd/counts.py names every slice.  The tensor is ACCUMULATED into (zero it first), so batches and streams
 * add up, and it is what the multi-GPU path sums with one RCCL all-reduce. */
#define C2_COUNT_VECTORS 20
#define C2_COUNT_SCALARS 24
#define C2_COUNT_HISTS 4
#define C2_COUNT_IGNORE_SUBSTITUTIONS 1   /* --ignore_substitutions */
#define C2_COUNT_IGNORE_INSERTIONS    2   /* --ignore_insertions */
#define C2_COUNT_IGNORE_DELETIONS     4   /* --ignore_deletions */
#define C2_COUNT_DISCARD_INDEL_READS  8   /* --discard_indel_reads */
#define C2_COUNT_ALL_REFS_LAYOUT      16  /* the tasks are one all_refs batch (task = read * n_refs + reference; n_tasks a multiple of n_refs) */
#define C2_COUNT_LEGACY_CLASSIFIER    32  /* positions as find_indels_substitutions_legacy gives them (COREResources.pyx:190-315); the records must come
                                             from a batch aligned with C2_BATCH_LEGACY_CLASSIFIER */

/* All d_* are device pointers (outputs of c2_align_classify_batch_device); d_weights: per task read multiplicity, 0 = do not
 * count this alignment, NULL = 1.  h_min_matches: HOST table n_refs x (max_t+1) of the smallest `matches` whose score
 * round(100*matches/len,3) exceeds refs[name]['min_aln_score'] for each alignment length (CRISPRessoCORE.py:697), or NULL
 * to count every alignment with weight > 0.  d_counts: n_refs x per_ref int64.  Enqueued on hip_stream. */
int c2_count_vectors_device(c2_ctx* ctx, uint64_t n_tasks, const uint8_t* d_aln_read, const uint8_t* d_aln_ref,
                            uint32_t aln_stride, const c2_aln_record* d_records, const uint32_t* d_weights,
                            const uint16_t* h_min_matches, int32_t max_t, int32_t flags, int32_t hl,
                            int64_t* d_counts, void* hip_stream);
/* The same with the hints of the batch (c2_batch.diag_hints of the call that wrote d_records; NULL = c2_count_vectors_device): a task with a valid hint is
 * counted from its hint word alone -- c2_count_hinted_kernel, a lane per task, neither its record nor its strings are read -- and the kernel above skips it.
 * The tensor is the same either way (CRISPRessoCORE.py:3996-4115 semantics; tests/test_counts_emulated.py, tests/test_gpu_parity.py).  The hints are used for a
 * context with one reference and for several references with every task against its own (ref_ids; the hinted kernel then runs per reference over the tasks
 * grouped by reference), and with the all-references layout (the weights are c2_select_best_device's: a reference's tasks by the layout's arithmetic).  With
 * several references the tasks the hinted kernel leaves are listed per reference and closed up into one list the column walk runs over. */
int c2_count_vectors_hinted_device(c2_ctx* ctx, uint64_t n_tasks, const uint8_t* d_aln_read, const uint8_t* d_aln_ref,
                                   uint32_t aln_stride, const c2_aln_record* d_records, const uint32_t* d_weights, const uint32_t* d_hints,
                                   const uint16_t* h_min_matches, int32_t max_t, int32_t flags, int32_t hl,
                                   int64_t* d_counts, void* hip_stream);

/* ---- Alleles_frequency_table on the device (second half) ----------------------------------------------------
 * Replaces the reference's per-variant Python loop that fills alleles_list (CRISPRessoCORE.py:3964-4010: one row per unique read and
 * reference it counts for; 'AMBIGUOUS_<first best reference>' rows for ambiguous reads, 'DISCARDED_<first best reference>' rows under
 * --discard_indel_reads), the frame's sort and %Reads (:4298-4303: #Reads descending, then Aligned_Sequence and Reference_Sequence
 * ascending, stable; %Reads = #Reads / N_TOTAL * 100), the text of Alleles_frequency_table.txt (:4498-4530, the non-detailed
 * columns, with --dsODN the two "contains dsODN" columns of :4512-4524) and <ref>Alleles_frequency_table_around_<guide>.txt
 * (CRISPRessoShared.py:1513-1531 get_dataframe_around_cut_asymmetrical, window plots/data_prep.py:285-301, file CRISPRessoCORE.py:5250-5273).
 * Input: what the count route left in HBM -- the aligned strings and records of the all-references batch (row = read * n_refs +
 * reference) and of the both-strand batch, c2_select_best_device's masks and flags, the multiplicities after the reverse-complement
 * transfer.  Nothing but the finished text (or, for c2_allele_table_fetch, the sorted rows) crosses the link. */
typedef struct c2_allele_src {
    uint64_t n_reads;                 /* unique reads */
    int32_t n_refs;
    int32_t mode;                     /* as c2_select_best_device: 0 ambiguous reads give one AMBIGUOUS_ row, 1 first best reference only, 2 one row per best reference */
    const uint8_t* d_aln_read1;       /* n_reads * n_refs rows of stride1 bytes */
    const uint8_t* d_aln_ref1;
    const c2_aln_record* d_records1;
    const uint8_t* d_aln_read2;       /* rows of the both-strand batch (stride2), or NULL */
    const uint8_t* d_aln_ref2;
    const c2_aln_record* d_records2;
    const int32_t* d_slot2;           /* n_reads * n_refs: row in batch 2 or -1; NULL without batch 2 */
    const uint64_t* d_member;         /* ceil(n_refs / 64) words per read: best references */
    const uint64_t* d_use2;           /* ... whose reverse-complement alignment (batch 2) won */
    const uint8_t* d_flags;           /* bit 0: aligned */
    const uint32_t* d_counts;         /* multiplicity per read after the reverse-complement transfer (:3970-3975); 0 = gave its copies away */
    const uint8_t* d_scaffold_hit;    /* per read: counted for 'Scaffold-incorporated' with its alignment against reference scaffold_ref (:786-796), or NULL */
    uint32_t stride1, stride2;
    int32_t scaffold_ref;
    uint32_t flags;                   /* C2_COUNT_IGNORE_SUBSTITUTIONS / _INSERTIONS / _DELETIONS (Read_Status), C2_COUNT_DISCARD_INDEL_READS */
} c2_allele_src;

/* One row of the table; `label` indexes the caller's label list: [0, k) reference r; [k, 2k) 'AMBIGUOUS_' + reference r; [2k, 3k)
 * 'DISCARDED_' + reference r; 3k 'Scaffold-incorporated'; 3k + 1 'DISCARDED_Scaffold-incorporated'  (k = n_refs). */
typedef struct c2_allele_row {
    uint32_t src;                     /* row of batch 1, or with bit 31 set of batch 2 */
    uint32_t reads;                   /* #Reads */
    uint32_t read;                    /* unique read */
    uint16_t aln_len, label, n_deleted, n_inserted, n_mutated;
    uint8_t modified, reserved;
} c2_allele_row;
#ifdef __cplusplus
static_assert(sizeof(c2_allele_row) == 24, "c2_allele_row is ABI: 24 bytes");
#endif
#define C2_ALLELE_LABELS(k) (3 * (k) + 2)
#define C2_ALLELE_MAX_WINDOW 124      /* left + right columns of an around-cut window */

typedef struct c2_allele_table c2_allele_table;
/* Rows built and sorted on the device (enqueued on hip_stream, complete when it returns).  The table refers to the caller's device
 * buffers (strings, records): they must outlive it.  n_reads * n_refs and the number of rows must stay below 2^31. */
int c2_allele_table_build(c2_ctx* ctx, const c2_allele_src* src, c2_allele_table** out, void* hip_stream);
uint64_t c2_allele_table_rows(const c2_allele_table* t);
/* Alleles_frequency_table.txt: header + one line per row, text formed on the device chunk by chunk, copied through pinned buffers and
 * written by `threads` host threads (pwrite).  labels: C2_ALLELE_LABELS(n_refs) strings.  probes: NULL, or 4 strings -- dsODN, its
 * reverse complement, dsODN[3:-3], its reverse complement -- for the two "contains dsODN" columns (`str.find(..) > 0`: an occurrence at
 * column 0 hides every later one).  %Reads is printed as Python prints the double #Reads / n_total * 100 (shortest round-trip repr). */
int c2_allele_table_write(c2_allele_table* t, const char* path, const char* const* labels, int64_t n_total,
                          const char* const* probes, int32_t threads, uint64_t* bytes_written);
/* The same text as the one member `member` of the zip archive `zip_path` -- what the reference's run leaves behind: it zips the table
 * (ZIP_DEFLATED, allowZip64) and removes the .txt (CRISPRessoCORE.py:4531-4533).  The chunks that come off the device are deflated by
 * `threads` host threads (zlib `level`, 1 = fastest): every slice from a fresh window, ended with a sync flush, so the slices concatenate into
 * ONE deflate stream (the pigz scheme); CRC-32 per slice, combined; zip64 records when the text is 4 GiB or more.  Only the compressed
 * bytes are written.  *text_bytes: the table's size as text, *zip_bytes: the archive's. */
int c2_allele_table_write_zip(c2_allele_table* t, const char* zip_path, const char* member, const char* const* labels, int64_t n_total,
                              const char* const* probes, int32_t threads, int32_t level, uint64_t* text_bytes, uint64_t* zip_bytes);
/* The sorted rows for a caller that wants them in memory: rows[m], and the two strings of every row zero-padded to `stride` bytes
 * (>= the longest alignment) in aligned[m * stride] / reference[m * stride] (either may be NULL). */
int c2_allele_table_fetch(c2_allele_table* t, c2_allele_row* rows, uint8_t* aligned, uint8_t* reference, uint32_t stride);
/* <ref>Alleles_frequency_table_around_<guide>.txt for the rows labelled `label`: every allele cut down to plot_window_size columns either
 * side of the column that holds reference base `cut_point` (clipped at the amplicon's ends, ref_len), equal windows (same strings,
 * Unedited, n_deleted, n_inserted, n_mutated) merged -- #Reads summed, %Reads summed in table order with pandas' Kahan compensation --
 * sorted by #Reads descending, then the merged key ascending.  C2_E_INVALID with "<cut_point> is not in list" when a row's reference
 * string has no such base.  Grouping on the device; sums, order and text on the host. */
int c2_allele_table_around_cut_write(c2_allele_table* t, int32_t label, int32_t cut_point, int32_t ref_len, int32_t plot_window_size,
                                     int64_t n_total, const char* path, int32_t threads, uint64_t* n_groups);
void c2_allele_table_free(c2_allele_table* t);
/* repr() of a Python float (David Gay shortest round trip, 'r' format: exponent form below 1e-4 and from 1e16): out >= 32 bytes, returns the length */
int c2_format_float_repr(double v, char* out);

/* ---- multi-GPU: replaces CRISPRessoMultiProcessing's process pool + the variants_<k>.tsv exchange (CRISPRessoCORE.py:1870-1985) ----
 * One process per GPU, reads sharded by contiguous ranges, no data-path collective; the per-amplicon count tensor of
 * c2_count_vectors_device is the only thing exchanged: one RCCL all-reduce (sum, int64) over xGMI.
 * c2_comm_unique_id: rank 0 obtains the 128-byte communicator id and hands it to the other ranks by any means (a file, MPI,
 * torch.distributed's store); c2_comm_init: collective over all `world` ranks, binds the communicator to the context's GPU
 * (librccl.so is bound at run time: the copy already loaded in the process, else the ROCm installation's);
 * c2_reduce_counts: in-place all-reduce of n_elements int64 at the device pointer d_counts, enqueued on hip_stream. */
#define C2_COMM_ID_BYTES 128
int c2_comm_unique_id(uint8_t* out_id);
int c2_comm_init(c2_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id);
int c2_reduce_counts(c2_ctx* ctx, int64_t* d_counts, uint64_t n_elements, void* hip_stream);
int c2_comm_destroy(c2_ctx* ctx);

/* Strand and best-reference choice of get_new_variant_object (CRISPRessoCORE.py:683, :697-707, :710, :779-785) over the
 * records of an all-references batch: d_records is n_reads x n_refs (task = read * n_refs + reference).  Optional second
 * batch: d_records2 holds the reverse-complement alignments of the (read, reference) pairs whose seeds were inconclusive,
 * d_slot2 (n_reads x n_refs, -1 = none) says where.  h_min_mscore: HOST table of n_refs thresholds -- the smallest integer
 * k with k/1000.0 > refs[name]['min_aln_score'] (scores are round(100*matches/len, 3), compared as 1000 x score).
 * mode: 0 ambiguous reads count for no reference, 1 --assign_ambiguous_alignments_to_first_reference,
 * 2 --expand_ambiguous_alignments.  Outputs (device pointers, any may be NULL): d_member / d_use2 ceil(n_refs / 64) 64-bit words per read
 * (bit r % 64 of word r / 64: reference r is a best match / its reverse-complement alignment won), d_flags one byte per read (1 aligned,
 * 2 ambiguous), d_weights (n_reads x n_refs) and d_weights2 (one per d_records2 entry): the multiplicity with which each
 * alignment enters c2_count_vectors_device, formed from d_counts (NULL = 1); d_stats: 11 uint64 sums, the caller zeroes
 * them -- N_COMPUTED_ALN, N_COMPUTED_NOTALN, N_CACHED_ALN, N_CACHED_NOTALN, N_GLOBAL_SUBS, N_SUBS_OUTSIDE_WINDOW,
 * N_MODS_IN_WINDOW, N_MODS_OUTSIDE_WINDOW, N_READS_IRREGULAR_ENDS (CRISPRessoCORE.py:1974-1979, weighted with d_raw_counts),
 * records with a non-zero status, one such status.  Alignments of 8000 columns or more are refused
 * (C2_E_TOO_LARGE: max_aln_len states the bound the caller guarantees).  Enqueued on hip_stream. */
#define C2_SELECT_STATS 11
int c2_select_best_device(c2_ctx* ctx, uint64_t n_reads, int32_t n_refs, const c2_aln_record* d_records,
                          const c2_aln_record* d_records2, const int32_t* d_slot2, const uint32_t* h_min_mscore,
                          const uint32_t* d_raw_counts, const uint32_t* d_counts, int32_t mode, int32_t max_aln_len,
                          uint64_t* d_member, uint64_t* d_use2, uint8_t* d_flags, uint32_t* d_weights, uint32_t* d_weights2,
                          uint64_t* d_stats, void* hip_stream);

/* Pointer-plane banding of the batch kernel (a pure performance knob; results never depend on it).
 * band_lanes: -1 automatic (default), 0 off, n > 0 keep the pointer words of n lanes on each side of the main diagonal.
 * Alignments whose traceback leaves the band are redone in the same call by the full-plane kernel.
 * target_workgroups_per_cu (> 0) steers the automatic choice. */
int c2_set_band(c2_ctx* ctx, int32_t band_lanes, int32_t target_workgroups_per_cu);
/* Kernel chain: 0 automatic (when the scoring allows it: diagonal-band kernels with optimality certificate, 4 then 2 then 1
 * alignments per wavefront, each over the tasks the previous one could not certify; else the banded row-strip kernel),
 * 1 banded row-strip kernel, 2 full-plane row-strip kernel only, 3 single-alignment diagonal-band kernel only, 4 diagonal
 * tiers 2 -> 1.  Every chain ends with the full-plane kernel over what is left.  Results never depend on the mode. */
int c2_set_kernel_mode(c2_ctx* ctx, int32_t mode);
/* Band in use for reads up to max_read_len (-1: diagonal-band kernels come first), and how many tasks of the most recent launch needed the full-plane pass. */
int c2_band_info(c2_ctx* ctx, int32_t max_read_len, int32_t* band_lanes, int32_t* fallback_tasks_last_launch);
/* Most recent batch: number of banded launches in front of the full-plane launch, and how many tasks each of them left over. */
int c2_tier_info(c2_ctx* ctx, int32_t* n_tiers, int32_t* left_over4);
/* The launch chain a batch with reads up to max_read_len would get: *kernels bit 0 c2_align_diagp_kernel<8>, 1 c2_align_diagx_kernel<4>,
 * 2 diagp<4>, 3 diagx<2>, 4 diagp<2>, 5 c2_align_diag_kernel, 6 banded row-strip first launch, 7 the last launch keeps its pointer plane
 * in HBM scratch, 8 the packed kernels run their 32-bit-add variant (c2_pk_add32_ok: sums as v_add_u32 under a per-anti-diagonal bias); ref_packed_ok (n_refs bytes, may be NULL): 1 where the packed int16 fill admits the reference (its DP values provably
 * fit, c2_pk_eligible) -- the int32 kernels run everything else (the reference's C ints, CRISPResso2Align.pyx:142-147). */
int c2_chain_info(c2_ctx* ctx, int32_t max_read_len, uint32_t* kernels, uint8_t* ref_packed_ok);
/* (*kernels bit 9: the score-only stage is in front of the first band tier -- c2_align_partition_kernel + c2_align_diags_kernel<16>: the tasks whose read is as
 * long as its reference and agrees with it in its last 32 columns (an indel in front of them would shift them) go through the packed fill WITHOUT pointer bits, which finishes those whose alignment is the
 * main diagonal and hands the rest to the first tier; round 5: all-references batches of up to 64 references too.)
 * c2_score_stage_info: did it run for the most recent batch, how many tasks it took, how many it finished. */
int c2_score_stage_info(c2_ctx* ctx, int32_t* ran, int64_t* tasks, int64_t* finished);
/* The partition in front of the chain (c2_align_partition_kernel, same conditions as the score-only stage) gives every task a class: 0 the
 * score-only launch; otherwise, by the diagonal the middle of the read lies on (a 32-base window of the read against the windows of the reference
 * within 64 bases of the same place) -- 1: the path needs few diagonals, c2_align_diagp_kernel<16> (14 diagonals, sixteen alignments per
 * wavefront) takes it first; 2: the first band tier (also: nothing found); 3 / 4: the path needs more diagonals than the first / second tier's
 * band has, the task goes straight to the list of the second / third tier.  Every launch verifies what it finishes and hands on what it cannot:
 * the class only decides where a task is tried FIRST.  With this, left_over[t] of c2_tier_info is the length of the list the launch behind tier
 * t reads: what tier t left plus what the partition put there.
 * Round 5: the last class -- the read matches its reference nowhere (three 32-base windows of it find no place within 64 bases of their own with at most
 * four differing bases): no band will certify it, it goes straight to the list of the last launch (the full matrix).  A batch whose reads differ
 * in length has the slots of every 4,096-task chunk ordered by read length first, so that the lists' neighbours can share a lane group of the
 * packed kernels (same reference AND read length); an all-references batch of several references is walked reference-major for the same reason.
 * Seven classes in all: 0 score-only, 1 the 14-diagonal launch (opt-in), 2 the first band tier (32 diagonals), 3 the 40-diagonal tier (round 5: six
 * alignments per wavefront, for reads that overhang their amplicon at both ends), 4 the 62-diagonal tier, 5 the 126/128-diagonal tier, 6 the
 * full-matrix launch.
 * c2_partition_info: *ran bit 0 the partition ran for the most recent batch, bit 1 the 14-diagonal launch too; class_tasks7: tasks per class;
 * finished2: tasks the score-only launch and the 14-diagonal launch finished. */
int c2_partition_info(c2_ctx* ctx, int32_t* ran, int64_t* class_tasks7, int64_t* finished2);
/* Round 5: a class-0 read that lies on its reference's main diagonal -- a byte-for-byte copy (the unedited, error-free read of an amplicon run) or
 * one that differs from it in one or two bases (A C G T N) -- is finished by the partition itself when the scoring proves that the diagonal beats
 * every other path (c2_main_diagonal_certificate, csrc/c2_host_prep.h): a path with a gap run opened inside the matrix loses gap_open against at
 * most two mismatches; a path that is one other diagonal from end to end (a leading and a trailing run only) is bounded by the equal bytes the
 * kernel counts on the diagonals +-1 and +-2.  CRISPResso2Align.pyx:338-421 then walks (L, L) -> (0, 0) in state M: the aligned strings are the
 * two sequences, the events their substitutions (COREResources.pyx:113-118).  Such tasks count in class_tasks7[0]; *n = how many of them there
 * were.  C2_NO_EXACT_COPIES=1: none (every class-0 task goes through the score-only launch); C2_DIAG_CERT_KMAX=0: byte-for-byte copies only. */
int c2_partition_finished(c2_ctx* ctx, int64_t* n);
/* The same for up to 8 tiers, plus per tier the number of tasks its packed (int16) kernel could not pair and handed to the 32-bit
 * kernel of the same band: a tier with a packed kernel finished at least tasks_in - unpaired - left_over tasks in int16 arithmetic. */
int c2_tier_info_ex(c2_ctx* ctx, int32_t* n_tiers, int32_t* left_over8, int32_t* unpaired8);

/* ---- per-call path: same contract as the reference's Cython functions ------------------ */

/* global_align(pystr_seqj, pystr_seqi, matrix, gap_incentive, gap_open, gap_extend), pyx:103-105.
 * Host pointers.  out_read_aln / out_ref_aln need Lj+Li bytes.  n_gap_incentive != Li+1 is
 * reported through *out_status = -1 (the reference prints and returns 0, pyx:124-126);
 * otherwise *out_status holds the C2_STATUS_* bits of the alignment. */
int c2_global_align(c2_ctx* ctx, const char* read, int32_t Lj, const char* ref, int32_t Li,
                    const int64_t* matrix, int32_t mat_dim,
                    const int64_t* gap_incentive, int32_t n_gap_incentive,
                    int32_t gap_open, int32_t gap_extend,
                    char* out_read_aln, char* out_ref_aln,
                    int32_t* out_len, int32_t* out_matches, int32_t* out_status);

/* find_indels_substitutions(read_seq_al, ref_seq_al, _include_indx), COREResources.pyx:71, and
 * find_indels_substitutions_legacy (pyx:193) when legacy != 0.  Any two equal-length strings are
 * accepted (including shapes the aligner never emits).  Results come back as flat int32 lists:
 * `out` is a caller buffer of `out_cap` int32; `out_index[2*k]`/`out_index[2*k+1]` receive offset and
 * length of list k (C2_LIST_* order).  Returns C2_E_OVERFLOW and the needed size in *out_needed if
 * out_cap is too small. */
enum {
    C2_LIST_REF_POSITIONS = 0,
    C2_LIST_ALL_INSERTION_POSITIONS,
    C2_LIST_ALL_INSERTION_LEFT_POSITIONS,
    C2_LIST_INSERTION_POSITIONS,
    C2_LIST_INSERTION_COORDINATES,      /* flattened (start, end) pairs */
    C2_LIST_INSERTION_SIZES,
    C2_LIST_ALL_DELETION_POSITIONS,
    C2_LIST_ALL_DELETION_COORDINATES,   /* pairs */
    C2_LIST_DELETION_POSITIONS,
    C2_LIST_DELETION_COORDINATES,       /* pairs */
    C2_LIST_DELETION_SIZES,
    C2_LIST_ALL_SUBSTITUTION_POSITIONS,
    C2_LIST_ALL_SUBSTITUTION_VALUES,    /* character codes */
    C2_LIST_SUBSTITUTION_POSITIONS,
    C2_LIST_SUBSTITUTION_VALUES,        /* character codes */
    C2_LIST_COUNT
};
int c2_find_indels_substitutions(c2_ctx* ctx, const char* read_aln, const char* ref_aln, int32_t n,
                                 const int32_t* include_idx, int32_t n_include, int32_t legacy,
                                 int32_t* out, int32_t out_cap, int32_t* out_index /* 2*C2_LIST_COUNT */,
                                 int64_t* out_counts /* insertion_n, deletion_n, substitution_n */,
                                 int32_t* out_needed);

/* calculate_homology(a, b), COREResources.pyx:318-327: matches over strlen(a), float32 accumulator. */
int c2_calculate_homology(c2_ctx* ctx, const char* a, const char* b, int32_t n, double* out);

/* Batched find_indels_substitutions[_legacy] (CRISPRessoCOREResources.pyx:68-187 / :190-315): the per-read classifier calls of
 * get_new_variant_object (CRISPRessoCORE.py:721-724) for n alignments in a few launches.  Host pointers.  aln_read / aln_ref:
 * n rows of `stride` bytes, lens[t] columns valid in row t; set_ids[t] (NULL = 0) picks one of n_sets include sets,
 * include_idx[include_off[k] .. include_off[k+1]) (any integers, any order).  The result owns three arrays:
 * index[n * C2_LIST_COUNT + 1] (list k of alignment t is values[index[t*C2_LIST_COUNT+k] .. index[t*C2_LIST_COUNT+k+1]), C2_LIST_* order),
 * values, counts[n][3] = insertion_n, deletion_n, substitution_n. */
typedef struct c2_lists c2_lists;
int c2_classify_lists_batch(c2_ctx* ctx, uint64_t n, const uint8_t* aln_read, const uint8_t* aln_ref, uint32_t stride,
                            const int32_t* lens, const uint16_t* set_ids, const int32_t* include_idx, const int64_t* include_off,
                            int32_t n_sets, int32_t legacy, c2_lists** out);
uint64_t c2_lists_total(const c2_lists* r);
const int64_t* c2_lists_index(const c2_lists* r);
const int32_t* c2_lists_values(const c2_lists* r);
const int64_t* c2_lists_counts(const c2_lists* r);
void c2_lists_free(c2_lists* r);

/* Profiling aid: while enabled, every launch adds the shader cycles each workgroup spends in the four phases of a task
 * (0 fetch, 1 DP fill, 2 traceback, 3 output+classification) to four device counters.  The call first copies the
 * counters to out4 (may be NULL) and clears them, then sets the mode. */
int c2_phase_profile(c2_ctx* ctx, int enable, uint64_t* out4);

/* Paired reads: get_consensus_alignment_from_pairs (CRISPRessoCORE.py:829-984) for n pairs at once.  Host pointers.
 * s1/f1, s2/f2: aligned read / aligned reference of read 1 and read 2 (n rows of `stride` bytes, n1[t] / n2[t] columns);
 * q1/q2: the reads' quality strings (n rows of `qstride`, lq1[t] / lq2[t] characters); best1[t] = (score_r1 >= score_r2).
 * stride and qstride are multiples of 4 (the kernel reads the rows as dwords).
 * Outputs: n rows of `ostride` >= 2*stride bytes each for the consensus aligned sequence, reference and quality, and
 * out_info[t] = {length of sequence and reference, length of the quality string, columns where they are equal, flags:
 * 1 = caching_is_ok, 2 = the reference raises IndexError on these inputs}. */
int c2_consensus_pairs_batch(c2_ctx* ctx, uint64_t n, const uint8_t* s1, const uint8_t* f1, const uint8_t* s2, const uint8_t* f2,
                             uint32_t stride, const int32_t* n1, const int32_t* n2, const uint8_t* q1, const uint8_t* q2,
                             uint32_t qstride, const int32_t* lq1, const int32_t* lq2, const uint8_t* best1,
                             uint8_t* out_aln, uint8_t* out_ref, uint8_t* out_qual, uint32_t ostride, int32_t* out_info);
/* The same with every pointer a DEVICE address (launch only, on hip_stream): the paired route keeps both reads' alignments on the
 * device -- the rows c2_align_classify_batch_device wrote -- and only the qualities travel from the host. */
int c2_consensus_pairs_device(c2_ctx* ctx, uint64_t n, const uint8_t* d_s1, const uint8_t* d_f1, const uint8_t* d_s2, const uint8_t* d_f2,
                              uint32_t stride, const int32_t* d_n1, const int32_t* d_n2, const uint8_t* d_q1, const uint8_t* d_q2,
                              uint32_t qstride, const int32_t* d_lq1, const int32_t* d_lq2, const uint8_t* d_best1,
                              uint8_t* d_out_aln, uint8_t* d_out_ref, uint8_t* d_out_qual, uint32_t ostride, int32_t* d_out_info, void* hip_stream);

/* Records of GIVEN aligned strings on the device: for every item (two rows of `stride` bytes, d_info[4 * item] columns of them, d_info[4 * item + 2]
 * matching columns -- c2_consensus_pairs_device's d_out_info) the 32-byte c2_aln_record the align kernels' fused classifier would write:
 * window and whole-amplicon counts of find_indels_substitutions (CRISPRessoCOREResources.pyx:68-187; legacy != 0: pyx:190-315), irregular ends
 * (CRISPRessoCORE.py:1106-1110).  d_ref_ids NULL: all-references layout (item = unit * n_refs + reference).  An item whose strings have a shape
 * the aligner never emits (a gap in both strings of a column, an insertion column next to a deletion column) gets status 128: the count route
 * does not take it, the list-based classifier (c2_classify_lists_batch) does.  A consensus the reference raises IndexError for (info flag 2): status 1.
 * This is what puts get_new_variant_object_from_paired's consensus alignments (CRISPRessoCORE.py:987-1169) on the count route. */
int c2_classify_records_device(c2_ctx* ctx, uint64_t n, const uint8_t* d_aln_read, const uint8_t* d_aln_ref, uint32_t stride, const int32_t* d_info,
                               const uint16_t* d_ref_ids, const uint8_t* d_strands, int32_t legacy, c2_aln_record* d_records, void* hip_stream);

/* ---- FASTQ ingest + exact de-duplication (host code, no GPU): the first pass of process_fastq,
 * CRISPRessoCORE.py:1820-1849 -- every 4-line record's sequence line, str.strip()'ed, counted per distinct sequence in
 * first-seen order; plain or gzip'ed input, universal newlines.  The result owns: the unique sequences back to back
 * (arena + n_unique+1 offsets: what c2_align_classify_batch_* take) and counts[n_unique].  Errors: c2_fastq_last_error(). */
typedef struct c2_fastq c2_fastq;
int c2_fastq_unique(const char* path, c2_fastq** out);
/* c2_fastq_unique with the reference's read filter fused in front (CRISPResso2/filterFastqs.py:128-226, called from
 * CRISPRessoCORE.py:3696-3717 for --min_single_bp_quality / --min_average_read_quality / --min_bp_quality_or_N; <= 0 = not
 * set): the records that pass, with low-quality bases masked to 'N', are what gets de-duplicated -- the same bytes the
 * reference writes to its *_filtered.fastq.gz, without the file.  c2_fastq_n_reads = reads after the filter;
 * *nonempty_lines_in_input = what `grep -c .` counts in the input (the reference's N_READS_INPUT is int(that / 4.0),
 * CRISPRessoShared.py:743-748).  The reference's own failures (empty quality line under a minimum, sequence / quality
 * length mismatch under masking, the read-only buffer of its single-bp + mask combination) return C2_E_INVALID. */
int c2_fastq_unique_filtered(const char* path, int32_t min_bp_qual_in_read, int32_t min_av_read_qual, int32_t min_bp_qual_or_N,
                             c2_fastq** out, uint64_t* nonempty_lines_in_input);
uint64_t c2_fastq_n_unique(const c2_fastq* r);
uint64_t c2_fastq_n_reads(const c2_fastq* r);
/* Lines with at least one byte in the text that was parsed ('\n'-terminated, as `grep -c .` counts them): the reference's
 * get_n_reads_fastq (CRISPRessoShared.py:743-748) is int(that / 4.0) -- N_READS_INPUT of an unfiltered file, and
 * N_READS_AFTER_PREPROCESSING of the filtered text when the read filter ran in front (c2_fastq_unique_filtered). */
uint64_t c2_fastq_nonempty_lines(const c2_fastq* r);
uint64_t c2_fastq_arena_bytes(const c2_fastq* r);
const uint8_t* c2_fastq_arena(const c2_fastq* r);
const uint64_t* c2_fastq_offsets(const c2_fastq* r);
const uint32_t* c2_fastq_counts(const c2_fastq* r);
void c2_fastq_free(c2_fastq* r);
const char* c2_fastq_last_error(void);
/* The same ingest chunk by chunk (replaces the reference's one readline loop over the whole file, CRISPRessoCORE.py:1820-1849, when
 * the caller wants to overlap it with the device): _open takes the path and the three read-filter thresholds (all 0: no filter;
 * plain text is then pread() chunk by chunk, .gz / filtered input is held in memory); every _next parses one more chunk and
 * reports the number of unique reads and arena bytes so far and whether the text is exhausted.  Everything below those marks is
 * final: arena bytes (the pointer never moves), offsets (the pointer is valid until the next _next), first-seen order.
 * Multiplicities keep growing until done; _counts copies them out (n = the current number of unique reads). */
typedef struct c2_fastq_stream c2_fastq_stream;
int c2_fastq_stream_open(const char* path, int32_t min_bp_qual_in_read, int32_t min_av_read_qual, int32_t min_bp_qual_or_N, c2_fastq_stream** out);
int c2_fastq_stream_next(c2_fastq_stream* s, uint64_t* n_unique, uint64_t* arena_bytes, int32_t* done);
const uint8_t* c2_fastq_stream_arena(const c2_fastq_stream* s);
const uint64_t* c2_fastq_stream_offsets(const c2_fastq_stream* s);
uint64_t c2_fastq_stream_text_bytes(const c2_fastq_stream* s);
/* the text that _next parses, when it is in memory (inflated .gz / BGZF input, the read filter's output, a mapped plain file; NULL for a
 * plain file that is pread() in chunks): a caller that frames it on the device (c2_fq_*_device below) uploads it from here and never
 * calls _next -- the host then has inflated / filtered, the device parses */
const uint8_t* c2_fastq_stream_text(const c2_fastq_stream* s);
uint64_t c2_fastq_stream_n_reads(const c2_fastq_stream* s);
uint64_t c2_fastq_stream_nonempty_lines(const c2_fastq_stream* s);          /* of the parsed text (after the filter) */
uint64_t c2_fastq_stream_nonempty_lines_input(const c2_fastq_stream* s);    /* of the text in front of the filter (0 without one) */
int c2_fastq_stream_counts(c2_fastq_stream* s, uint32_t* out, uint64_t n);
/* c2_rc_partners (below) for the stream's unique reads, answered from the table the ingest built (after done; n = number of unique reads) */
int c2_fastq_stream_rc_partners(c2_fastq_stream* s, int64_t* partner, uint64_t n);
void c2_fastq_stream_close(c2_fastq_stream* s);
/* Paired input: the first pass of process_paired_fastq's n_processes > 1 route, CRISPRessoCORE.py:1296-1334 -- the two
 * files read in lockstep, key = seq1 + '+' + reverse_complement(seq2) (both str.strip()'ed; CRISPRessoShared.py:399-403's
 * reverse complement: a character outside ACGTN_- fails like its KeyError), counted per distinct key in first-seen order;
 * the aux arena holds, per key, the quality pair qual1 + ' ' + qual2[::-1] of its FIRST occurrence.
 * c2_fastq_paired_occurrences is the second pass of that route (:1452-1513): every occurrence, in file order, of the keys
 * with selected[k] != 0 -- out's counts[j] = k and aux entry j = the occurrence's own quality pair. */
int c2_fastq_unique_paired(const char* path1, const char* path2, c2_fastq** out);
int c2_fastq_paired_occurrences(const char* path1, const char* path2, const c2_fastq* uniq, const uint8_t* selected, c2_fastq** out);
uint64_t c2_fastq_aux_bytes(const c2_fastq* r);
const uint8_t* c2_fastq_aux(const c2_fastq* r);
const uint64_t* c2_fastq_aux_offsets(const c2_fastq* r);
/* BGZF input (bgzip / htslib: gzip members of <= 64 KiB of text that name their own size) member range by member range, for a caller
 * that uploads the text while it is inflated (crispresso2_amd/fastq_device.py): _open maps the file and indexes the members from their
 * headers (C2_E_INVALID "not a BGZF file" for anything else: padding, foreign members, truncation -- the whole-file routes above decide
 * about those); _text_offsets: n_blocks + 1 offsets of the members' text; _inflate: members [b0, b1) into dst on `threads` threads
 * (0: the CPUs the process may use), CRC and length of every member checked. */
typedef struct c2_bgzf c2_bgzf;
int c2_bgzf_open(const char* path, c2_bgzf** out);
uint64_t c2_bgzf_n_blocks(const c2_bgzf* h);
const uint64_t* c2_bgzf_text_offsets(const c2_bgzf* h);
int c2_bgzf_inflate(c2_bgzf* h, uint64_t b0, uint64_t b1, uint8_t* dst, uint64_t cap, int32_t threads);
void c2_bgzf_close(c2_bgzf* h);
/* The same handle over ONE ordinary gzip member (`gzip -6 reads.fastq`: no index): the member is cut into segments at deflate block starts found
 * by search, a first decode gives every segment's size and the 32 KiB in front of it (c2_gz_parallel.h); c2_bgzf_n_blocks / _text_offsets /
 * _inflate / _close then treat the segments as they treat BGZF members -- a caller that uploads the text while it is inflated needs no copy of
 * it in host memory.  The member's CRC-32 is checked when its last segment has been inflated (c2_bgzf_inflate fails then, as gzip.py raises at
 * the end of a member, CRISPRessoCORE.py:1820-1823).  threads <= 0: the CPUs this process may use; chunk_bytes 0: the library's choice.
 * C2_E_INVALID "not applicable (...)": several members, trailing bytes, a small file, no block start found -- inflate the file serially. */
int c2_gzseg_open(const char* path, int32_t threads, uint64_t chunk_bytes, c2_bgzf** out);

/* ONE ordinary gzip member (the whole file: `gzip -6 reads.fastq`, pigz) inflated by `threads` host threads (<= 0: the CPUs this process may
 * use) -- what gzip.open(fastq_filename, 'rt') does on one thread, CRISPResso2/CRISPRessoCORE.py:1820-1823.  The file routes above take this
 * path by themselves for files of a few megabytes and more (C2_GZ_PARALLEL=0 switches it off); this entry is the same code with the caller's
 * segment size (`chunk_bytes` of compressed data per segment, 0: the library's choice) and buffer.  Block boundaries are found by search,
 * every segment is decoded twice (symbols, then bytes), the member's CRC-32 and ISIZE are checked (c2_gz_parallel.h).
 *   0              the text is in dst, *n_out bytes
 *   C2_E_OVERFLOW  cap is too small; *n_out = the size of the text
 *   C2_E_INVALID   not applicable or not certain (several members, trailing bytes, no block boundary found, a damaged stream, too small to
 *                  cut): nothing is known about dst; inflate the file serially -- that route's text and errors are the result.
 * stats8 (may be NULL): segments, block starts found, bytes out, fell back (0/1), microseconds of: search, first pass, windows, second pass. */
int c2_gz_inflate_parallel(const uint8_t* gz, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* n_out, int32_t threads, uint64_t chunk_bytes,
                           uint64_t* stats8);
/* the same eight numbers for the last .gz file one of the file routes (c2_fastq_unique*, c2_fastq_stream_open) opened on the calling thread:
 * all zero when this route was not tried (BGZF input, a small file, C2_GZ_PARALLEL=0), fell back = 1 when it declined and the serial route ran */
void c2_gz_parallel_last(uint64_t* stats8);
/* Host-side bookkeeping between ingest and kernels, over the same arena/offsets layout (errors: c2_fastq_last_error()):
 * the seed test that picks the strand(s) a read is aligned on (CRISPRessoCORE.py:656-687) -> out_plan[n] in {0 forward,
 * 1 reverse complement, 2 both}, and the reverse-complement merge of read counts (CRISPRessoCORE.py:3970-3975), in place. */
int c2_strand_plan(const uint8_t* arena, const uint64_t* offsets, uint64_t n, const char* const* fw_seeds, const char* const* rc_seeds,
                   int32_t n_seeds, int32_t seed_min, uint8_t* out_plan);
int c2_merge_reverse_complements(const uint8_t* arena, const uint64_t* offsets, uint64_t n, const uint8_t* aligned, int64_t* counts);
/* c2_strand_plan for reads that are already on the device, all references at once (one wavefront per read).  d_reads / d_offsets:
 * the arena and its offsets in device memory; the seeds come from the host: seed_blob holds their bytes, seed_off / seed_len
 * [n_refs][2 (forward, reverse complement)][max_seeds] locate them, n_seeds[r] <= max_seeds is how many of reference r take part
 * (min(aln_seed_count, seeds the reference has)).  d_plan: uint8 [n_reads][n_refs], as c2_strand_plan's.  Enqueued on hip_stream. */
int c2_strand_plan_device(c2_ctx* ctx, uint64_t n_reads, const uint8_t* d_reads, const uint64_t* d_offsets, int32_t max_read_len,
                          int32_t n_refs, int32_t max_seeds, const int32_t* n_seeds, const uint8_t* seed_blob, int32_t blob_bytes,
                          const int32_t* seed_off, const int32_t* seed_len, int32_t seed_min, uint8_t* d_plan, void* hip_stream);
/* The same merge in two steps: c2_rc_partners (independent of the alignments: partner[i] = index of the read that equals
 * reverse_complement(read i), or -1; a host thread can run it while the device aligns) and the sequential count transfer over it. */
int c2_rc_partners(const uint8_t* arena, const uint64_t* offsets, uint64_t n, int64_t* partner);
int c2_merge_counts_with_partners(uint64_t n, const uint8_t* aligned, const int64_t* partner, int64_t* counts);
/* The reads idx[0..m) of an arena packed back to back (repeats allowed) -> out_offsets[m + 1] and, unless out_arena is NULL (sizes
 * only), their bytes: the second batch of the count route (reads aligned on both strands, CRISPRessoCORE.py:675-687). */
int c2_gather_reads(const uint8_t* arena, const uint64_t* offsets, const int64_t* idx, uint64_t m, uint8_t* out_arena, uint64_t* out_offsets);

/* ---- FASTQ framing and exact de-duplication ON THE DEVICE (the readline loop of process_fastq, CRISPRessoCORE.py:1825-1849, for text
 * that is uploaded as it lies in the file; the host parser above is a quarter as fast on a 16-CPU host as the link delivers the text).
 * Semantics = c2_fastq_stream's for text WITHOUT carriage returns (Python's text mode would translate them: flag bit 0 tells the host
 * to use its own parser): records are four consecutive '\n'-terminated lines from the top, the second one str.strip()ped.
 * All buffers are device memory owned by the caller; every call only enqueues on hip_stream.  The host-side driver is
 * crispresso2_amd/fastq_device.py.
 *   c2_fq_count_device  text [lo, hi) (lo a multiple of 16; bytes [0, hi) resident) -> per tile of C2_FQ_TILE_BYTES = 16384 bytes from lo on: the
 *                       number of '\n' and the number of '\n' that end an EMPTY line; flags |= 1 if a '\r' was seen.
 *   c2_fq_lines_device  the same range again, with tile_base[t] = number of '\n' in the text in front of tile t (the caller's
 *                       prefix sum): seq_start[r] = first byte of record r's sequence line (behind newline 4r), seq_end[r] = the
 *                       newline that ends it (newline 4r + 1); entries for records >= n_records_cap are not written.
 *   c2_fq_dedup_device  records [range[0], range[1]) (two uint64 in DEVICE memory): stripped sequence -> rinfo[r] = start << 24 |
 *                       length; looked up in / inserted into the open-addressing table `slots` (n_slots a power of two, zeroed by the
 *                       caller; `first` filled with 0xff): count[slot] += 1, first[slot] = min(first[slot], r), slot_of[r] = slot,
 *                       stats[0] += new keys, stats[1] = max(stats[1], their lengths), stats[2] += 1 if the empty sequence became a
 *                       key (three uint32).  Equal means equal bytes (compared, not hashed).  flags |= 2: a line of 2^24 bytes or
 *                       more / text beyond 2^40; |= 4: range beyond n_records_cap.  The caller keeps stats[0] below n_slots / 2.
 *   c2_fq_gather_device out[out_offsets[i] ..) = the bytes info[records ? records[i] : i] names (start << 24 | length) in `text`.
 *   c2_fq_rc_partner_device  for the n unique reads records[0..n): partner_slot[i] = the table slot whose key equals
 *                       reverse_complement(read i) (CRISPRessoShared.py:399-403: upper-cased, ACGTN_- only), -1 if there is none or the
 *                       read has another character -- c2_rc_partners through the table the de-duplication built. */
#define C2_FQ_TILE_BYTES 16384
int c2_fq_count_device(c2_ctx* ctx, const uint8_t* d_text, uint64_t lo, uint64_t hi, uint32_t* d_tile_newlines, uint32_t* d_tile_empty,
                       uint32_t* d_flags, void* hip_stream);
int c2_fq_lines_device(c2_ctx* ctx, const uint8_t* d_text, uint64_t lo, uint64_t hi, const uint64_t* d_tile_base, uint64_t* d_seq_start,
                       uint64_t* d_seq_end, uint64_t n_records_cap, void* hip_stream);
int c2_fq_dedup_device(c2_ctx* ctx, const uint8_t* d_text, const uint64_t* d_seq_start, const uint64_t* d_seq_end, const uint64_t* d_range,
                       uint64_t n_records_cap, uint64_t* d_slots, uint64_t n_slots, uint32_t* d_count, uint32_t* d_first,
                       uint32_t* d_slot_of, uint64_t* d_rinfo, uint32_t* d_flags, uint32_t* d_stats, void* hip_stream);
int c2_fq_gather_device(c2_ctx* ctx, const uint8_t* d_text, const uint64_t* d_info, const int64_t* d_records, const int64_t* d_out_offsets,
                        uint8_t* d_out, uint64_t n, void* hip_stream);
int c2_fq_rc_partner_device(c2_ctx* ctx, const uint8_t* d_text, const uint64_t* d_info, const int64_t* d_records, uint64_t n,
                            const uint64_t* d_slots, uint64_t n_slots, int32_t* d_partner_slot, void* hip_stream);
/* Paired input on the device: the reading loop of process_paired_fastq (CRISPRessoCORE.py:1309-1334 -- record r of R1 with record r of R2, key =
 * seq1 + '+' + reverse_complement(seq2), qualities qual1 + ' ' + qual2[::-1], all four lines str.strip()ped) over two texts in HBM.
 *   c2_fq_lines4_device        c2_fq_lines_device that also records the quality line: qual_start[r] behind newline 4r + 2, qual_end[r] = newline 4r + 3.
 *                              (A line the text ends in without a newline, or that is not there at all, keeps what the caller put into the arrays:
 *                              the text's length makes it end there / empty, as readline() at the end of a file.)
 *   c2_fq_pair_lengths_device  d_lines1 / d_lines2: HOST arrays of the four device pointers {seq_start, seq_end, qual_start, qual_end} of a text; per
 *                              record r < n the stripped lines -> s1 / q1 / s2 / q2 [r] = start << 24 | length, key_len[r] = len(seq1) + 1 + len(seq2),
 *                              qual_len[r] likewise.  flags |= 1: a line of 2^24 bytes or more, or a text of 2^40.
 *   c2_fq_pair_write_device    key_out[key_off[r] ..) = the key, qual_out[qual_off[r] ..) = the quality pair (offsets: the caller's exclusive prefix sums
 *                              of the lengths).  flags |= 2: a character of seq2 outside ACGTN_- in either case (CRISPRessoShared.py:399-403's KeyError).
 * The keys are then de-duplicated with c2_fq_dedup_device over the key arena (first-seen order, multiplicities); the second pass of the reference's
 * route (every occurrence of a key whose consensus depended on its qualities, :1452-1513) is an index selection on what is already in HBM --
 * crispresso2_amd/paired_device.py. */
int c2_fq_lines4_device(c2_ctx* ctx, const uint8_t* d_text, uint64_t lo, uint64_t hi, const uint64_t* d_tile_base, uint64_t* d_seq_start,
                        uint64_t* d_seq_end, uint64_t* d_qual_start, uint64_t* d_qual_end, uint64_t n_records_cap, void* hip_stream);
int c2_fq_pair_lengths_device(c2_ctx* ctx, const uint8_t* d_text1, const uint8_t* d_text2, const uint64_t* const* d_lines1, const uint64_t* const* d_lines2,
                              uint64_t n, uint64_t* d_s1, uint64_t* d_q1, uint64_t* d_s2, uint64_t* d_q2, int64_t* d_key_len, int64_t* d_qual_len,
                              uint32_t* d_flags, void* hip_stream);
int c2_fq_pair_write_device(c2_ctx* ctx, const uint8_t* d_text1, const uint8_t* d_text2, uint64_t n, const uint64_t* d_s1, const uint64_t* d_q1,
                            const uint64_t* d_s2, const uint64_t* d_q2, const int64_t* d_key_off, const int64_t* d_qual_off, uint8_t* d_key_out,
                            uint8_t* d_qual_out, uint32_t* d_flags, void* hip_stream);

/* Hardware self-test of the cross-lane primitives (DPP wave_shr:1 / wave_shl:1 with and without bound_ctrl, also with a
 * lane switched off in EXEC, readlane, ballot) the DP depends on; writes 448 int32 (see c2_selftest_kernel).  Used by the
 * GPU test-suite. */
int c2_selftest(c2_ctx* ctx, int32_t* out448);
/* ... and of the row forms (DPP row_shr:1 / row_shl:1 with bound_ctrl: the hand-off inside the 16-lane groups of the packed
 * kernel): out[lane] = 1000 + value of lane - 1 (0 at the start of a row of 16 lanes), out[64 + lane] = 1000 + value of lane + 1
 * (0 at the end of a row), values 3 * lane + 1. */
int c2_selftest_rows(c2_ctx* ctx, int32_t* out128);

#ifdef __cplusplus
}
#endif
#endif
