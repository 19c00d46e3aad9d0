// Plain-data structures shared by the host side (c2_api.hip) and the kernels (c2_kernels.hip).
#pragma once
#include <stdint.h>
#include "crispresso2_amd.h"   // c2_aln_record, C2_STATUS_* (public ABI)

// traceback / DP states (same numbering as the reference: CRISPResso2Align.pyx:23)
#define C2_ST_M 1
#define C2_ST_I 2
#define C2_ST_J 3

#define C2_MAX_CODES 32            // distinct score-matrix symbols (+1 shared "scores 0" code) the kernel keeps in LDS
#define C2_INVALID_CODE 255
#define C2_PTR_PAD 2               // halfword padding of one pointer column (breaks the 128-B bank stride)
#define C2_LANES 64
#define C2_TASK_CHUNK 4              // tasks a workgroup takes per atomic
#define C2_STATUS_NEED_FULL 64     // internal: banded launch could not finish the traceback; the full-plane launch overwrites the record

// Device-resident description of one reference amplicon.
typedef struct c2_dev_ref {
    const uint8_t* seq;           // Li bytes
    const int32_t* gap_incentive; // Li+1 (int64 input truncated to int32 exactly as the reference's int arithmetic does)
    const uint16_t* inc_prefix;   // Li+2: inc_prefix[x] = number of include idxs < x  (window membership and range hits)
    int32_t len;                  // Li
    int32_t reserved;
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
    int32_t band_lanes;           // banded kernel: lanes kept on each side of the main-diagonal lane
    int32_t reserved;
    uint32_t* fb_count;           // banded kernel: number of tasks whose traceback left the band ...
    uint32_t* fb_list;            // ... and their task indices (capacity n_tasks)
    const uint32_t* task_list;    // full kernel, second launch: run only these tasks (NULL = tasks 0..n_tasks-1)
    const uint32_t* task_count;   // device-resident length of task_list
    unsigned long long* work_counter; // device counter the workgroups pull task chunks from; zero before every launch
    unsigned long long* phase_cycles; // optional: 4 counters of per-phase shader cycles (profiling), else NULL
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
