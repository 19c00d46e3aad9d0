// c2_kernels.hip -- CDNA4 (gfx950) kernels for CRISPResso2's align + classify hot path.
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
// This file is the umbrella over the kernel units (the wave emulator of tests/emu compiles all of them in one go); the library builds
// each unit in its own translation unit next to the host code that launches it (c2_api_*.hip, Makefile).
#include "c2_k_common.h"
#include "c2_k_align.hip"
#include "c2_k_classify.hip"
#include "c2_k_select.hip"
#include "c2_k_count.hip"
#include "c2_k_fastq.hip"
#include "c2_k_alleles.hip"
