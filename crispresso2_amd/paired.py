"""Paired reads: batched equivalents of the reference's `get_consensus_alignment_from_pairs` (CRISPRessoCORE.py:829-984) and
`get_new_variant_object_from_paired` (:987-1169).

Both reads of a pair are aligned to every reference on the device (one batch; forward and/or reverse-complement as the
seed test over BOTH reads decides, :1024-1064), the two alignments are merged column by column by
c2_consensus_pairs_kernel (one lane per pair: the reference's two-pointer walk with its quality rule, :800-826), and the
consensus alignments of the best references are classified by the batched classifier.  The per-pair dicts equal the
reference's, including what it leaves out for pairs (no 'aln_strand', no 'best_match_name', float
'insertions_outside_window' / 'total_mods') and `caching_is_ok` of the LAST consensus call of the pair.
"""
import ctypes

import numpy as np

from . import CRISPRessoCOREResources, _native
from .batch import BatchAligner


def _rows(strings, stride):
    a = np.zeros((len(strings), stride), dtype=np.uint8)
    for k, s in enumerate(strings):
        b = s.encode() if isinstance(s, str) else bytes(s)
        a[k, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return a


def consensus_batch(items, ctx=None, errors="raise"):
    """items: sequence of (aln_seq_r1, aln_ref_r1, score_r1, qual_r1, aln_seq_r2, aln_ref_r2, score_r2, qual_r2).
    -> list of (final_aln, final_qual, final_ref, score, caching_is_ok) as the reference returns them; an item on which the
    reference raises IndexError (a quality index past the end of its string) raises IndexError here -- or, with errors="skip",
    gives None in its place."""
    if errors == "skip":
        good = [k for k, it in enumerate(items) if len(it[0]) >= len(it[1]) and len(it[4]) >= len(it[5])]
        out = [None] * len(items)
        for k, c in zip(good, consensus_batch([items[k] for k in good], ctx=ctx, errors="none")):
            out[k] = c
        return out
    n = len(items)
    if n == 0:
        return []
    ctx = ctx or _native.default_context()
    n1 = np.array([len(it[1]) for it in items], dtype=np.int32)
    n2 = np.array([len(it[5]) for it in items], dtype=np.int32)
    for it in items:
        if len(it[0]) < len(it[1]) or len(it[4]) < len(it[5]):
            raise IndexError('string index out of range')
    lq1 = np.array([len(it[3]) for it in items], dtype=np.int32)
    lq2 = np.array([len(it[7]) for it in items], dtype=np.int32)
    stride = max(16, (int(max(n1.max(), n2.max())) + 15) // 16 * 16)
    qstride = max(16, (int(max(lq1.max(), lq2.max())) + 15) // 16 * 16)
    ostride = 2 * stride
    s1, f1 = _rows([it[0][:len(it[1])] for it in items], stride), _rows([it[1] for it in items], stride)
    s2, f2 = _rows([it[4][:len(it[5])] for it in items], stride), _rows([it[5] for it in items], stride)
    q1, q2 = _rows([it[3] for it in items], qstride), _rows([it[7] for it in items], qstride)
    best1 = np.array([1 if it[2] >= it[6] else 0 for it in items], dtype=np.uint8)        # is_best_aln_r1, :876
    oa = np.zeros((n, ostride), dtype=np.uint8)
    orf = np.zeros((n, ostride), dtype=np.uint8)
    oq = np.zeros((n, ostride), dtype=np.uint8)
    info = np.zeros((n, 4), dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ctx.check(ctx.lib.c2_consensus_pairs_batch(ctx.handle, ctypes.c_uint64(n), p(s1), p(f1), p(s2), p(f2), ctypes.c_uint32(stride),
                                               p(n1), p(n2), p(q1), p(q2), ctypes.c_uint32(qstride), p(lq1), p(lq2), p(best1),
                                               p(oa), p(orf), p(oq), ctypes.c_uint32(ostride), p(info)), "c2_consensus_pairs_batch")
    out = []
    for k in range(n):
        ln, lq, hom, fl = (int(x) for x in info[k])
        if fl & 2:
            if errors == "none":
                out.append(None)
                continue
            raise IndexError('string index out of range')
        out.append((oa[k, :ln].tobytes().decode(), oq[k, :lq].tobytes().decode(), orf[k, :ln].tobytes().decode(),
                    round(float(100 * hom / float(ln)), 3), bool(fl & 1)))
    return out


def get_consensus_alignment_from_pairs(aln_seq_r1, aln_ref_r1, score_r1, qual_r1, aln_seq_r2, aln_ref_r2, score_r2, qual_r2):
    """Same signature and return value as the reference function (CRISPRessoCORE.py:829)."""
    return consensus_batch([(aln_seq_r1, aln_ref_r1, score_r1, qual_r1, aln_seq_r2, aln_ref_r2, score_r2, qual_r2)])[0]


def _pair_plan(args, s1, s2, ref):
    """0 forward, 1 reverse complement, 2 both: the seed test over both reads of the pair (:1024-1036)."""
    found_fw = found_rc = 0
    for k in range(min(args.aln_seed_count, len(ref['fw_seeds']))):
        if ref['fw_seeds'][k] in s1 or ref['fw_seeds'][k] in s2:
            found_fw += 1
        if ref['rc_seeds'][k] in s1 or ref['rc_seeds'][k] in s2:
            found_rc += 1
    if found_fw > args.aln_seed_min and found_rc == 0:
        return 0
    if found_fw == 0 and found_rc > args.aln_seed_min:
        return 1
    return 2


def get_new_variant_objects_from_paired(args, pairs, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None):
    """pairs: sequence of (fastq1_seq, fastq2_seq, fastq1_qual, fastq2_qual).  One dict per pair, equal to
    get_new_variant_object_from_paired(args, *pair, refs, ref_names, aln_matrix, pe_scaffold_dna_info)."""
    ctx = ctx or _native.default_context()
    aligner = BatchAligner([refs[n]['sequence'] for n in ref_names], [refs[n]['gap_incentive'] for n in ref_names],
                           [refs[n]['include_idxs'] for n in ref_names], aln_matrix,
                           args.needleman_wunsch_gap_open, args.needleman_wunsch_gap_extend, ctx=ctx)
    npairs, k = len(pairs), len(ref_names)
    # ---- every alignment the pairs need: (pair, reference, strand) x (read 1, read 2)
    plans = [[_pair_plan(args, p[0], p[1], refs[name]) for name in ref_names] for p in pairs]
    reads, rids, strands, slot = [], [], [], {}
    for i, p in enumerate(pairs):
        for r in range(k):
            for st in ((0,) if plans[i][r] == 0 else (1,) if plans[i][r] == 1 else (0, 1)):
                slot[(i, r, st)] = len(reads)
                reads += [p[0], p[1]]
                rids += [r, r]
                strands += [st, st]
    res = aligner.align(reads, ref_ids=np.array(rids, dtype=np.uint16), strands=np.array(strands, dtype=np.uint8)) if reads else None
    if res is not None:
        bad = res.records['status'] != 0
        if bad.any():
            st = int(res.records['status'][np.nonzero(bad)[0][0]])
            if st & _native.STATUS_RC_CHAR:
                raise KeyError("reverse_complement: character outside ACGTN_-")
            raise Exception('global_align: undefined alignment (status %d)' % st)
        scores = res.scores
    # ---- consensus of every (pair, reference, strand)
    items, item_of = [], {}
    for key, t in slot.items():
        i = key[0]
        a1, a2 = res.strings(t), res.strings(t + 1)
        item_of[key] = len(items)
        items.append((a1[0], a1[1], float(scores[t]), pairs[i][2], a2[0], a2[1], float(scores[t + 1]), pairs[i][3]))
    cons = consensus_batch(items, ctx=ctx)
    # ---- per pair: the winning amplicons (variants._Winners: :1066-1081) -> classifier jobs
    from .variants import _Winners, _complete_payload, _settle_read
    picked, jobs, job_sets = [], [], []
    set_of = {name: r for r, name in enumerate(ref_names)}
    for i in range(npairs):
        w = _Winners()
        cache_ok = True
        for r, ref_name in enumerate(ref_names):
            pl = plans[i][r]
            if pl == 0 or pl == 1:
                a, qual, f, score, cache_ok = cons[item_of[(i, r, pl)]]
            else:
                a, qual, f, score, _ = cons[item_of[(i, r, 0)]]
                rv_a, rv_qual, rv_f, rv_score, cache_ok = cons[item_of[(i, r, 1)]]        # caching flag of the LAST consensus call, :1049
                if rv_score > score:
                    a, qual, f, score = rv_a, rv_qual, rv_f, rv_score
            w.offer(ref_name, a, f, score, refs[ref_name]['min_aln_score'], detail=(ref_name, a, f, score, qual))
        picked.append((w, cache_ok, len(jobs)))
        if w.aligned:
            for name, a, f, _ in w.entries:
                jobs.append((a, f))
                job_sets.append(set_of[name])
    payloads = CRISPRessoCOREResources.find_indels_substitutions_batch(
        jobs, [refs[name]['include_idxs'] for name in ref_names], set_ids=np.array(job_sets, dtype=np.uint16),
        legacy=bool(args.use_legacy_insertion_quantification), ctx=ctx)
    variants = []
    for w, cache_ok, first_job in picked:
        result = {'count': 1}
        if not w.aligned:                                          # :1145-1152
            result['aln_scores'] = w.scores
            result['ref_aln_details'] = w.details
            result['best_match_score'] = w.top
            result['caching_is_ok'] = cache_ok
            variants.append(result)
            continue
        result['aln_ref_names'] = w.names()
        result['aln_scores'] = w.scores
        result['ref_aln_details'] = w.details
        result['best_match_score'] = w.top
        result['caching_is_ok'] = cache_ok
        labels = []
        for q, (name, a, f, _) in enumerate(w.entries):
            payload = payloads[first_job + q]
            labels.append(_complete_payload(payload, args, name, a, f, w.scores, paired=True))     # (pairs: no 'aln_strand', float counts, :1093-1131)
            result['variant_' + name] = payload
        _settle_read(result, labels, w, args, pe_scaffold_dna_info)
        variants.append(result)
    return variants


def read_paired_fastq_unique(fastq1_filename, fastq2_filename):
    """First pass of process_paired_fastq (CRISPRessoCORE.py:1296-1334) in the native library (c2_fastq_unique_paired):
    -> (variantCache {seq1 + '+' + reverse_complement(seq2): [copies, qual1 + ' ' + qual2[::-1] of the first occurrence]},
    the PairedFastq handle for the second pass)."""
    pf = _native.PairedFastq(fastq1_filename, fastq2_filename)
    return {k: [int(c), q] for k, c, q in zip(pf.keys, pf.counts, pf.quals)}, pf


def _split_pair(key, quals):
    fastq1_seq, fastq2_seq = key.split('+')                         # :1225-1226 (a '+' inside a read is a ValueError there too)
    fastq1_qual, fastq2_qual = quals.split(' ')
    return fastq1_seq, fastq2_seq, fastq1_qual, fastq2_qual


def process_paired_fastq(fastq1_filename, fastq2_filename, args, refs, ref_names, aln_matrix, pe_scaffold_dna_info=None, ctx=None,
                         variants_dir=None, rank=None, world=None, get_variants=None):
    """process_paired_fastq, the n_processes > 1 route (CRISPRessoCORE.py:1296-1516), with GPU ranks as the workers:
    unique pairs (native ingest) -> every rank computes the variants of its slice of the keys with the qualities of each
    key's first occurrence (device alignments, consensus kernel, batched classifier) -> with `variants_dir`, the slices
    travel as variants_<rank>.tsv in the reference's format and rank 0 merges them; keys seen more than once whose
    consensus had to choose a base by quality (`caching_is_ok` false) are computed again for every occurrence with that
    occurrence's own qualities; pair keys are replaced by the consensus read.
    -> (variantCache, not_aligned_variants, aln_stats) on rank 0, None elsewhere."""
    import os
    from . import variant_io
    from .distributed import shard_boundaries
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    get_variants = get_variants or get_new_variant_objects_from_paired
    cache, pf = read_paired_fastq_unique(fastq1_filename, fastq2_filename)
    keys = list(cache.keys())
    if world > 1 and variants_dir is None:
        raise ValueError("variants_dir is needed to exchange the variants of %d ranks" % world)
    if len(keys) < world:
        raise Exception("The number of unique sequences is less than the number of processes. Please reduce the number of processes.")
    b = shard_boundaries(len(keys), world) if keys else [0] * (world + 1)
    mine = keys[b[rank]:b[rank + 1]]
    variants = get_variants(args, [_split_pair(k, cache[k][1]) for k in mine], refs, ref_names, aln_matrix, pe_scaffold_dna_info, ctx=ctx)
    if variants_dir is not None:
        variant_io.write_variant_file(variants_dir, rank, mine, variants)
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()
        if rank != 0:
            return None
        computed = (kv for k in range(world) for kv in variant_io.read_variant_file(os.path.join(variants_dir, "variants_%d.tsv" % k)))
    else:
        computed = zip(mine, variants)
    # ---- the parent's merge (:1382-1437)
    st = variant_io.new_aln_stats()
    expand = args.expand_ambiguous_alignments
    re_aln, not_aln = {}, {}
    for seq, variant in computed:
        if cache[seq][0] > 1 and not variant["caching_is_ok"]:
            re_aln[seq] = variant
            del cache[seq]
            continue
        count = cache[seq][0]
        st['N_TOT_READS'] += count
        variant['count'] = count
        if variant['best_match_score'] <= 0:
            st['N_COMPUTED_NOTALN'] += 1
            st['N_CACHED_NOTALN'] += count - 1
            not_aln[seq] = variant
        else:
            cache[seq] = variant
            st['N_COMPUTED_ALN'] += 1
            st['N_CACHED_ALN'] += count - 1
            if len(variant['aln_ref_names']) == 1 or expand:
                variant_io.account_variant(st, variant, count, variant['aln_ref_names'])
    for seq in not_aln:
        del cache[seq]
    for key in list(cache.keys()):                                  # pair key -> consensus read (:1439-1448)
        if '+' in key:
            variant = cache.pop(key)
            new_key = variant["variant_" + variant['aln_ref_names'][0]]['aln_seq']
            if new_key in cache:
                cache[new_key]['count'] += variant['count']
            else:
                cache[new_key] = variant
    if re_aln:                                                      # second pass over the files (:1450-1513)
        index_of = {k: i for i, k in enumerate(keys)}
        selected = np.zeros(len(keys), dtype=np.uint8)
        for k in re_aln:
            selected[index_of[k]] = 1
        idx, quals = pf.occurrences(selected)
        again = get_variants(args, [_split_pair(keys[int(i)], q) for i, q in zip(idx, quals)], refs, ref_names, aln_matrix,
                             pe_scaffold_dna_info, ctx=ctx)
        for variant in again:
            st['N_TOT_READS'] += 1
            if variant['best_match_score'] <= 0:
                st['N_COMPUTED_NOTALN'] += 1
                continue
            st['N_COMPUTED_ALN'] += 1
            aln_seq = variant["variant_" + variant['aln_ref_names'][0]]['aln_seq']
            if aln_seq in cache:
                cache[aln_seq]['count'] += 1
            else:
                cache[aln_seq] = variant
            if len(variant['aln_ref_names']) == 1 or expand:
                variant_io.account_variant(st, variant, 1, variant['aln_ref_names'])
    pf.close()
    return cache, not_aln, st
