"""Batched equivalent of the reference's per-read driver `get_new_variant_object`
(CRISPRessoCORE.py:627-798) and of the unique-read loop of `process_fastq` around it (:1825-1849, :1957-1981).

Same inputs (the CRISPResso `args` namespace, `refs` dict, `ref_names`, score matrix) and the same per-read result dicts,
but every alignment of a batch of reads goes to the MI355X in a few launches (one per reference, forward and
reverse-complement tasks together) instead of one Cython call per read.  The strand / best-reference / ambiguity rules
are the reference's; only where they are evaluated changed.
"""
from copy import deepcopy

import numpy as np

from . import CRISPRessoCOREResources
from .batch import BatchAligner, score_from_counts
from . import _native


def _strand_plan(args, seq, ref):
    """Seed test of CRISPRessoCORE.py:656-687 -> 0: forward only, 1: reverse complement only, 2: both."""
    found_fw = found_rc = 0
    n = min(args.aln_seed_count, len(ref['fw_seeds']))
    for k in range(n):
        if ref['fw_seeds'][k] in seq:
            found_fw += 1
        if ref['rc_seeds'][k] in seq:
            found_rc += 1
    if found_fw > args.aln_seed_min and found_rc == 0:
        return 0
    if found_fw == 0 and found_rc > args.aln_seed_min:
        return 1
    return 2


def align_all(args, fastq_seqs, refs, ref_names, aln_matrix, ctx=None):
    """-> per read, per reference: (s1, s2, score, strand) chosen as the reference chooses them (:666-687)."""
    ctx = ctx or _native.default_context()
    aligner = BatchAligner([refs[n]['sequence'] for n in ref_names], [refs[n]['gap_incentive'] for n in ref_names],
                           [refs[n]['include_idxs'] for n in ref_names], aln_matrix,
                           args.needleman_wunsch_gap_open, args.needleman_wunsch_gap_extend, ctx=ctx)
    n = len(fastq_seqs)
    out = [[None] * len(ref_names) for _ in range(n)]
    for r, name in enumerate(ref_names):
        plan = [_strand_plan(args, s, refs[name]) for s in fastq_seqs]
        fw_idx = [k for k in range(n) if plan[k] != 1]
        rc_idx = [k for k in range(n) if plan[k] != 0]
        reads = [fastq_seqs[k] for k in fw_idx] + [fastq_seqs[k] for k in rc_idx]
        if not reads:
            continue
        strands = np.array([0] * len(fw_idx) + [1] * len(rc_idx), dtype=np.uint8)
        res = aligner.align(reads, ref_ids=np.full(len(reads), r, dtype=np.uint16), strands=strands)
        bad = res.records['status'] != 0
        if bad.any():
            t = int(np.nonzero(bad)[0][0])
            st = int(res.records['status'][t])
            if st & _native.STATUS_RC_CHAR:
                raise KeyError("reverse_complement: character outside ACGTN_- in read %r" % reads[t])
            raise Exception('global_align: undefined alignment (status %d) for read %r' % (st, reads[t]))
        scores = res.scores
        fw_of = {k: t for t, k in enumerate(fw_idx)}
        rc_of = {k: len(fw_idx) + t for t, k in enumerate(rc_idx)}
        for k in range(n):
            if plan[k] == 0:
                t = fw_of[k]
                s1, s2 = res.strings(t)
                out[k][r] = (s1, s2, float(scores[t]), '+')
            elif plan[k] == 1:
                t = rc_of[k]
                s1, s2 = res.strings(t)
                out[k][r] = (s1, s2, float(scores[t]), '-')
            else:
                tf, tr = fw_of[k], rc_of[k]
                if scores[tr] > scores[tf]:                       # strict: ties keep the forward alignment (:683)
                    s1, s2 = res.strings(tr)
                    out[k][r] = (s1, s2, float(scores[tr]), '-')
                else:
                    s1, s2 = res.strings(tf)
                    out[k][r] = (s1, s2, float(scores[tf]), '+')
    return out


def get_new_variant_objects(args, fastq_seqs, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None):
    """One result dict per read, equal to get_new_variant_object(args, seq, refs, ref_names, aln_matrix, pe_info).
    Alignments: one batch per reference (align_all); classifier payloads of all best alignments: one batched call
    (c2_classify_lists_batch) instead of one find_indels_substitutions launch per read."""
    ctx = ctx or _native.default_context()
    per_read = align_all(args, fastq_seqs, refs, ref_names, aln_matrix, ctx=ctx)
    ref_index = {name: r for r, name in enumerate(ref_names)}
    chosen = []                                                    # per read: scores, details and the best matches (:689-707)
    jobs, job_sets = [], []                                        # (s1, s2) of every best match, and its reference
    for k in range(len(fastq_seqs)):
        aln_scores = []
        best_match_score = -1
        best_match_s1s, best_match_s2s, best_match_names, best_match_strands = [], [], [], []
        ref_aln_details = []
        for r, ref_name in enumerate(ref_names):
            s1, s2, score, strand = per_read[k][r]
            ref_aln_details.append((ref_name, s1, s2, score))
            aln_scores.append(score)
            # best reference: strictly better and above that reference's min_aln_score; equal score -> ambiguous (:697-707)
            if score > best_match_score and score > refs[ref_name]['min_aln_score']:
                best_match_score = score
                best_match_s1s, best_match_s2s = [s1], [s2]
                best_match_names, best_match_strands = [ref_name], [strand]
            elif score == best_match_score:
                best_match_s1s.append(s1)
                best_match_s2s.append(s2)
                best_match_names.append(ref_name)
                best_match_strands.append(strand)
        first_job = len(jobs)
        if best_match_score > 0:
            for idx, name in enumerate(best_match_names):
                jobs.append((best_match_s1s[idx], best_match_s2s[idx]))
                job_sets.append(ref_index[name])
        chosen.append((aln_scores, ref_aln_details, best_match_score, best_match_s1s, best_match_s2s, best_match_names,
                       best_match_strands, first_job))
    payloads = CRISPRessoCOREResources.find_indels_substitutions_batch(
        jobs, [refs[name]['include_idxs'] for name in ref_names], set_ids=np.array(job_sets, dtype=np.uint16),
        legacy=bool(args.use_legacy_insertion_quantification), ctx=ctx)
    variants = []
    for k in range(len(fastq_seqs)):
        aln_scores, ref_aln_details, best_match_score, best_match_s1s, best_match_s2s, best_match_names, best_match_strands, first_job = chosen[k]
        new_variant = {'count': 1}
        if best_match_score <= 0:                                  # not aligned: scores only (:767-773)
            new_variant['aln_scores'] = aln_scores
            new_variant['ref_aln_details'] = ref_aln_details
            new_variant['best_match_score'] = best_match_score
            variants.append(new_variant)
            continue
        new_variant['aln_ref_names'] = best_match_names
        new_variant['aln_scores'] = aln_scores
        new_variant['ref_aln_details'] = ref_aln_details
        new_variant['best_match_score'] = best_match_score
        class_names = []
        for idx, best_match_name in enumerate(best_match_names):
            s1, s2 = best_match_s1s[idx], best_match_s2s[idx]
            payload = payloads[first_job + idx]                    # find_indels_substitutions[_legacy](s1, s2, include_idxs), :721-724
            payload['ref_name'] = best_match_name
            payload['aln_scores'] = aln_scores
            payload['irregular_ends'] = bool(s1[0] == '-' or s2[0] == '-' or s1[0] != s2[0]
                                             or s1[-1] == '-' or s2[-1] == '-' or s1[-1] != s2[-1])          # :729-733
            payload['insertions_outside_window'] = int((len(payload['all_insertion_positions']) / 2) - (len(payload['insertion_positions']) / 2))
            payload['deletions_outside_window'] = len(payload['all_deletion_coordinates']) - len(payload['deletion_coordinates'])
            payload['substitutions_outside_window'] = len(payload['all_substitution_positions']) - len(payload['substitution_positions'])
            payload['total_mods'] = int((len(payload['all_insertion_positions']) / 2) + len(payload['all_deletion_positions']) + len(payload['all_substitution_positions']))
            payload['mods_in_window'] = payload['substitution_n'] + payload['deletion_n'] + payload['insertion_n']
            payload['mods_outside_window'] = payload['total_mods'] - payload['mods_in_window']
            is_modified = False                                    # :746-760 (the elif chain only matters for which test fires first)
            if not args.ignore_deletions and payload['deletion_n'] > 0:
                is_modified = True
            elif not args.ignore_insertions and payload['insertion_n'] > 0:
                is_modified = True
            elif not args.ignore_substitutions and payload['substitution_n'] > 0:
                is_modified = True
            class_names.append(best_match_name + ("_MODIFIED" if is_modified else "_UNMODIFIED"))
            payload['classification'] = 'MODIFIED' if is_modified else 'UNMODIFIED'
            payload['aln_seq'] = s1
            payload['aln_ref'] = s2
            payload['aln_strand'] = best_match_strands[idx]
            new_variant['variant_' + best_match_name] = payload
            new_variant['best_match_name'] = best_match_name
        new_variant['class_name'] = "&".join(class_names)
        if len(best_match_names) > 1:                              # ambiguous alignments (:779-785)
            if args.assign_ambiguous_alignments_to_first_reference:
                new_variant['class_name'] = class_names[0]
                new_variant['aln_ref_names'] = [best_match_names[0]]
            elif not args.expand_ambiguous_alignments:
                new_variant['class_name'] = 'AMBIGUOUS'
        if getattr(args, 'prime_editing_pegRNA_scaffold_seq', '') and 'Prime-edited' in best_match_names:   # :789-796
            loc = new_variant['variant_Prime-edited']['ref_positions'].index(pe_scaffold_dna_info[0] - 1) + 1
            if new_variant['variant_Prime-edited']['aln_seq'][loc:(loc + len(pe_scaffold_dna_info[1]))] == pe_scaffold_dna_info[1]:
                new_variant['aln_ref_names'] = ["Scaffold-incorporated"]
                new_variant['class_name'] = "Scaffold-incorporated"
                old_payload = deepcopy(new_variant['variant_Prime-edited'])
                old_payload['ref_name'] = "Scaffold-incorporated"
                new_variant['variant_' + "Scaffold-incorporated"] = old_payload
        variants.append(new_variant)
    return variants


def read_fastq_unique(path):
    """First pass of process_fastq (CRISPRessoCORE.py:1825-1849): sequence line of every 4-line record -> dict seq -> count
    (insertion-ordered, like the reference's variantCache before alignment).  Plain or gzip.  Parsing and de-duplication
    run in the native library (c2_fastq_unique); an empty sequence (blank line / truncated record: the reference would
    hand '' to global_align, which is undefined there) is dropped."""
    arena, offsets, counts, _ = _native.fastq_unique(path)
    buf = arena.tobytes()
    cache = {}
    for k in range(len(counts)):
        a, b = int(offsets[k]), int(offsets[k + 1])
        if b > a:
            cache[buf[a:b].decode('utf-8')] = int(counts[k])
    return cache


def process_fastq(path, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None):
    """process_fastq equivalent: unique reads aligned in one batch; returns (variantCache, not_aligned_variants, aln_stats)
    with the reference's bookkeeping (CRISPRessoCORE.py:1957-2000)."""
    counts = read_fastq_unique(path)
    seqs = list(counts.keys())
    variants = get_new_variant_objects(args, seqs, refs, ref_names, aln_matrix, pe_scaffold_dna_info, ctx=ctx)
    variantCache, not_aligned = {}, {}
    st = dict(N_TOT_READS=0, N_CACHED_ALN=0, N_CACHED_NOTALN=0, N_COMPUTED_ALN=0, N_COMPUTED_NOTALN=0, N_GLOBAL_SUBS=0,
              N_SUBS_OUTSIDE_WINDOW=0, N_MODS_IN_WINDOW=0, N_MODS_OUTSIDE_WINDOW=0, N_READS_IRREGULAR_ENDS=0, READ_LENGTH=0)
    for seq, variant in zip(seqs, variants):
        c = counts[seq]
        st['N_TOT_READS'] += c
        variant['count'] = c
        if variant['best_match_score'] <= 0:
            st['N_COMPUTED_NOTALN'] += 1
            st['N_CACHED_NOTALN'] += c - 1
            not_aligned[seq] = variant
            continue
        variantCache[seq] = variant
        st['N_COMPUTED_ALN'] += 1
        st['N_CACHED_ALN'] += c - 1
        p = variant['variant_' + variant['best_match_name']]
        if st['READ_LENGTH'] == 0:
            st['READ_LENGTH'] = len(p['aln_seq'])
        st['N_GLOBAL_SUBS'] += (p['substitution_n'] + p['substitutions_outside_window']) * c
        st['N_SUBS_OUTSIDE_WINDOW'] += p['substitutions_outside_window'] * c
        st['N_MODS_IN_WINDOW'] += p['mods_in_window'] * c
        st['N_MODS_OUTSIDE_WINDOW'] += p['mods_outside_window'] * c
        if p['irregular_ends']:
            st['N_READS_IRREGULAR_ENDS'] += c
    return variantCache, not_aligned, st


def process_fastq_write_out(fastq_input, fastq_output, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None):
    """process_fastq_write_out equivalent (CRISPRessoCORE.py:2283-2350): process_fastq, then the input FASTQ written again
    (gzip) with every read's alignment summary on its '+' line.  -> (variantCache, not_aligned_variants, aln_stats)"""
    from . import variant_io
    cache, not_aligned, st = process_fastq(fastq_input, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info, ctx=ctx)
    variant_io.write_annotated_fastq(fastq_input, fastq_output, cache, not_aligned)
    return cache, not_aligned, st


def process_single_fastq_write_bam_out(fastq_input, bam_output, bam_header, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None):
    """process_single_fastq_write_bam_out equivalent (CRISPRessoCORE.py:2351-2515) up to the SAM text: process_fastq, then
    `bam_output + ".sam"` with the header and one line per input read (`refs[name]` must carry aln_chr / aln_start /
    aln_strand, as the reference requires).  Sorting and indexing that file into `bam_output` is `samtools`' job in the
    reference (:2503) and is left to the caller here.  -> (variantCache, not_aligned_variants, aln_stats)"""
    from . import variant_io
    cache, not_aligned, st = process_fastq(fastq_input, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info, ctx=ctx)
    variant_io.write_annotated_sam(fastq_input, bam_output + ".sam", bam_header, cache, not_aligned, refs)
    return cache, not_aligned, st


def process_fastq_sharded(path, args, refs, ref_names, aln_matrix, variants_dir, pe_scaffold_dna_info=None, ctx=None,
                          rank=None, world=None, get_variants=None):
    """The reference's n_processes > 1 route (CRISPRessoCORE.py:1870-1985) with one GPU rank in place of each worker
    process: every rank de-duplicates the FASTQ (host work, identical on all ranks), computes the variants of ITS slice of
    the unique reads (get_variant_cache_equal_boundaries) on its GPU, writes them as variants_<rank>.tsv in the reference's
    format; after a barrier rank 0 merges the files with the parent's bookkeeping.  Like the reference, fewer unique
    reads than ranks falls back to the one-process route (on rank 0).
    -> (variantCache, not_aligned_variants, aln_stats) on rank 0, None elsewhere."""
    import os
    from . import variant_io
    from .distributed import shard_boundaries
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    get_variants = get_variants or get_new_variant_objects
    counts = read_fastq_unique(path)
    seqs = list(counts.keys())
    if world <= 1 or len(seqs) <= world:
        if rank != 0:
            return None
        if get_variants is get_new_variant_objects:
            return process_fastq(path, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info, ctx=ctx)
        world = 1
    b = shard_boundaries(len(seqs), world)
    mine = seqs[b[rank]:b[rank + 1]]
    variants = get_variants(args, mine, refs, ref_names, aln_matrix, pe_scaffold_dna_info, ctx=ctx)
    variant_io.write_variant_file(variants_dir, rank, mine, variants)
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
    if rank != 0:
        return None
    paths = [os.path.join(variants_dir, "variants_%d.tsv" % k) for k in range(world)]
    st, not_aligned = variant_io.merge_variant_files(paths, counts, args)
    return counts, not_aligned, st
