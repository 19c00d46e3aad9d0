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


class _Winners:
    """The best-reference rule for one read over its candidate amplicons (CRISPRessoCORE.py:697-707; for pairs :1066-1081): a score that is
    strictly better than the best so far AND above that amplicon's min_aln_score takes over; an equal score joins it (an ambiguous read)."""
    __slots__ = ("top", "entries", "scores", "details")

    def __init__(self):
        self.top = -1
        self.entries = []                # (reference name, aligned read, aligned reference, strand or None) of the amplicons that share the top score
        self.scores = []                 # every amplicon's score, in ref_names order        -> 'aln_scores'
        self.details = []                # every amplicon's (name, strings, score[, qual])   -> 'ref_aln_details'

    def offer(self, ref_name, aligned_read, aligned_ref, score, min_aln_score, strand=None, detail=None):
        self.scores.append(score)
        self.details.append(detail if detail is not None else (ref_name, aligned_read, aligned_ref, score))
        if score > self.top and score > min_aln_score:
            self.top = score
            self.entries = [(ref_name, aligned_read, aligned_ref, strand)]
        elif score == self.top:
            self.entries.append((ref_name, aligned_read, aligned_ref, strand))

    @property
    def aligned(self):
        return self.top > 0

    def names(self):
        return [e[0] for e in self.entries]


def _complete_payload(payload, args, ref_name, aligned_read, aligned_ref, all_scores, paired=False):
    """the fields the reference derives from the classifier's payload for one winning amplicon (:726-760; pairs :1093-1131, which keep the
    two halves-of-a-list counts as floats) -> the class label of the read for this amplicon"""
    payload['ref_name'] = ref_name
    payload['aln_scores'] = all_scores
    a, f = aligned_read, aligned_ref
    payload['irregular_ends'] = bool(a[0] == '-' or f[0] == '-' or a[0] != f[0] or a[-1] == '-' or f[-1] == '-' or a[-1] != f[-1])
    ins_all, ins_win = len(payload['all_insertion_positions']) / 2, len(payload['insertion_positions']) / 2
    n_del_pos, n_sub_pos = len(payload['all_deletion_positions']), len(payload['all_substitution_positions'])
    payload['insertions_outside_window'] = (ins_all - ins_win) if paired else int(ins_all - ins_win)
    payload['deletions_outside_window'] = len(payload['all_deletion_coordinates']) - len(payload['deletion_coordinates'])
    payload['substitutions_outside_window'] = n_sub_pos - len(payload['substitution_positions'])
    payload['total_mods'] = (ins_all + n_del_pos + n_sub_pos) if paired else int(ins_all + n_del_pos + n_sub_pos)
    payload['mods_in_window'] = payload['substitution_n'] + payload['deletion_n'] + payload['insertion_n']
    payload['mods_outside_window'] = payload['total_mods'] - payload['mods_in_window']
    edited = ((not args.ignore_deletions and payload['deletion_n'] > 0) or (not args.ignore_insertions and payload['insertion_n'] > 0) or
              (not args.ignore_substitutions and payload['substitution_n'] > 0))
    payload['classification'] = 'MODIFIED' if edited else 'UNMODIFIED'
    payload['aln_seq'] = aligned_read
    payload['aln_ref'] = aligned_ref
    return ref_name + ("_MODIFIED" if edited else "_UNMODIFIED")


def _settle_read(result, labels, winners, args, pe_scaffold_dna_info):
    """what the reference does once every winning amplicon has its payload: the read's class name, the ambiguity flags (:779-785) and the
    prime-editing scaffold rule (:789-796: a read whose best amplicon is 'Prime-edited' and that carries the scaffold's first bases behind the
    extension is re-labelled)"""
    names = winners.names()
    result['class_name'] = "&".join(labels)
    if len(names) > 1:
        if args.assign_ambiguous_alignments_to_first_reference:
            result['class_name'] = labels[0]
            result['aln_ref_names'] = [names[0]]
        elif not args.expand_ambiguous_alignments:
            result['class_name'] = 'AMBIGUOUS'
    if getattr(args, 'prime_editing_pegRNA_scaffold_seq', '') and 'Prime-edited' in names:
        pe = result['variant_Prime-edited']
        at = pe['ref_positions'].index(pe_scaffold_dna_info[0] - 1) + 1
        if pe['aln_seq'][at:(at + len(pe_scaffold_dna_info[1]))] == pe_scaffold_dna_info[1]:
            result['aln_ref_names'] = ["Scaffold-incorporated"]
            result['class_name'] = "Scaffold-incorporated"
            twin = deepcopy(pe)
            twin['ref_name'] = "Scaffold-incorporated"
            result['variant_' + "Scaffold-incorporated"] = twin


def get_new_variant_objects(args, fastq_seqs, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None):
    """One result dict per read, equal to get_new_variant_object(args, seq, refs, ref_names, aln_matrix, pe_info).
    Two passes around two batched device calls: alignments, one batch per reference (align_all); then, with every read's winning
    amplicons known, the classifier payloads of all of them in one call (c2_classify_lists_batch) instead of one
    find_indels_substitutions launch per read."""
    ctx = ctx or _native.default_context()
    per_read = align_all(args, fastq_seqs, refs, ref_names, aln_matrix, ctx=ctx)
    set_of = {name: r for r, name in enumerate(ref_names)}
    picked = []                                                    # per read: its _Winners and where its classifier jobs start
    jobs, job_sets = [], []                                        # (aligned read, aligned reference) of every winning amplicon, and its include set
    for k in range(len(fastq_seqs)):
        w = _Winners()
        for r, ref_name in enumerate(ref_names):
            aligned_read, aligned_ref, score, strand = per_read[k][r]
            w.offer(ref_name, aligned_read, aligned_ref, score, refs[ref_name]['min_aln_score'], strand=strand)
        picked.append((w, len(jobs)))
        if w.aligned:
            for name, a, f, _ in w.entries:
                jobs.append((a, f))
                job_sets.append(set_of[name])
    payloads = CRISPRessoCOREResources.find_indels_substitutions_batch(
        jobs, [refs[name]['include_idxs'] for name in ref_names], set_ids=np.array(job_sets, dtype=np.uint16),
        legacy=bool(args.use_legacy_insertion_quantification), ctx=ctx)
    variants = []
    for w, first_job in picked:
        result = {'count': 1}
        if not w.aligned:                                          # not aligned: scores only (:767-773)
            result['aln_scores'] = w.scores
            result['ref_aln_details'] = w.details
            result['best_match_score'] = w.top
            variants.append(result)
            continue
        result['aln_ref_names'] = w.names()
        result['aln_scores'] = w.scores
        result['ref_aln_details'] = w.details
        result['best_match_score'] = w.top
        labels = []
        for q, (name, a, f, strand) in enumerate(w.entries):
            payload = payloads[first_job + q]                      # find_indels_substitutions[_legacy](aligned read, aligned reference, include_idxs), :721-724
            labels.append(_complete_payload(payload, args, name, a, f, w.scores))
            payload['aln_strand'] = strand
            result['variant_' + name] = payload
            result['best_match_name'] = name
        _settle_read(result, labels, w, args, pe_scaffold_dna_info)
        variants.append(result)
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
