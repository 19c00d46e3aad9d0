"""Paired FASTQ -> per-amplicon count tensors on the device, without a Python object per pair.

The reference's paired route (process_paired_fastq, CRISPRessoCORE.py:1245-1733: unique pair keys seq1 + '+' + reverse_complement(seq2) with the
qualities of their first occurrence; per key get_new_variant_object_from_paired :987-1169 -- seed test over BOTH reads, both reads aligned
against every amplicon on the strand(s) it asks for, get_consensus_alignment_from_pairs :829-984, best amplicon by the consensus' score,
classification of the consensus alignment; keys whose consensus had to choose a base by quality and that occur more than once are computed
again for every occurrence with its own qualities :1450-1513; pair keys are replaced by the consensus read :1439-1448) followed by the
aggregation loop every run shares (:3964-4115).  paired.process_paired_fastq mirrors it dict by dict; here the same run is a few batches:

    the two texts -> HBM, framed                        fastq_device.upload_whole, c2_fq_count / c2_fq_lines4 kernels (sequence AND quality line of every record)
    pair keys and quality pairs of every record         c2_fq_pair_lengths / c2_fq_pair_write kernels (strip(), '+', reverse complement, reversed qualities)
    exact de-duplication of the keys, first-seen order  c2_fq_dedup_kernel over the key arena
        (input the kernels do not take -- carriage returns, small files, anything the reference raises an error for: c2_fastq_unique_paired on the host)
    seed test over both reads                           c2_strand_plan_kernel on the KEY (a seed cannot span the '+')
    alignments of read 1 and of read 2                  the align launch chain, all amplicons, + one batch for the both-strand pairs
    consensus of the two alignments                     c2_consensus_pairs_kernel on the rows that are in HBM
    records of the consensus alignments                 c2_classify_records_kernel (what the fused classifier writes for an alignment)
    best amplicon / ambiguity / aln_stats               c2_select_best_kernel
    second pass for quality-dependent keys              the same batches over their occurrences: an index selection on the arenas in HBM
    pair key -> consensus read                          the consensus strings de-duplicated in first-seen order (c2_fq_dedup_kernel), copies added up
    reverse-complement merge, weights, count vectors    as pipeline.quantify_unique; the allele table through alleles.AlleleTable

-> pipeline.QuantResult.  Not taken here (PairedDeviceUnavailable; paired.process_paired_fastq handles them): the prime-editing scaffold rule,
the first-amplicon view of HDR / prime-editing runs, and a consensus whose shape the count route's classifier does not cover (a gap in both
strings of a column, an insertion column next to a deletion column).
"""
import ctypes
import os

import numpy as np

from . import _native
from . import counts as C
from .hostcopy import to_host, to_device


FORCE_HOST_PARSER = os.environ.get("C2_PAIRED_HOST_PARSER", "") not in ("", "0")    # the pair keys from c2_fastq_unique_paired instead of the device


class PairedDeviceUnavailable(Exception):
    """this run goes through paired.process_paired_fastq (the reason is the message)"""


# ---- the two launches this module adds (tests replace them with the wave emulator's entries) ----
def consensus_device(ctx, n, s1, f1, s2, f2, stride, n1, n2, q1, q2, qstride, lq1, lq2, best1, oa, orf, oq, ostride, info, stream):
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_consensus_pairs_device(ctx.handle, ctypes.c_uint64(n), V(s1), V(f1), V(s2), V(f2), ctypes.c_uint32(stride), V(n1), V(n2), V(q1), V(q2),
                                                ctypes.c_uint32(qstride), V(lq1), V(lq2), V(best1), V(oa), V(orf), V(oq), ctypes.c_uint32(ostride), V(info),
                                                V(stream or 0)), "c2_consensus_pairs_device")


def classify_records_device(ctx, n, aln_read, aln_ref, stride, info, ref_ids, strands, legacy, records, stream, refs=None, ref_names=None):
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_classify_records_device(ctx.handle, ctypes.c_uint64(n), V(aln_read), V(aln_ref), ctypes.c_uint32(stride), V(info), V(ref_ids or 0),
                                                 V(strands or 0), int(bool(legacy)), V(records), V(stream or 0)), "c2_classify_records_device")


def _mscore(matches, T):
    """1000 x round(100 * matches / T, 3) as integers (c2_mscore: exact, round half to even), for int64 tensors"""
    num = 100000 * matches
    Tc = T.clamp(min=1)
    q, r = num // Tc, num % Tc
    up = (2 * r > Tc) | ((2 * r == Tc) & ((q & 1) == 1))
    return q + up.to(q.dtype)


class _Units:
    """the tensors one pass leaves on the device: consensus strings + records of every (pair, amplicon) in the all-amplicons layout, the same for
    the reverse-complement alignments of the pairs aligned on both strands, and which pairs' consensus depended on the qualities"""
    pass


def _dev_gather(ctx, dev, stream, d_text, starts, lens, row_stride=0):
    """bytes d_text[starts[i] : starts[i] + lens[i]] (int64 device tensors) gathered ON THE DEVICE (c2_fq_gather_kernel): back to back -> (uint8 tensor,
    int64 offsets tensor [m + 1]); row_stride > 0: as zero-padded rows of that many bytes -> uint8 tensor [m, row_stride]"""
    import torch
    from . import fastq_device
    m = int(starts.numel())
    info = ((starts << 24) | lens).contiguous()
    if row_stride:
        out_off = torch.arange(m + 1, dtype=torch.int64, device=dev) * row_stride
        out = torch.zeros((max(m, 1), row_stride), dtype=torch.uint8, device=dev)
    else:
        out_off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        if m:
            torch.cumsum(lens, 0, out=out_off[1:])
        out = torch.empty(max(int(out_off[-1].item()), 1), dtype=torch.uint8, device=dev)
    if m:
        fastq_device.fq_gather(ctx, d_text.data_ptr(), info.data_ptr(), None, out_off.data_ptr(), out.data_ptr(), m, stream)
    return (out[:m] if row_stride else out), out_off


def _front(ctx, aligner, dev, stream, refs, ref_names, args, legacy, d_ka, k_start, k_plus, k_end, d_qa, q_start, q_space, q_end):
    """One pass over m pairs: key bytes d_ka[k_start[i] : k_end[i]] (on the device) with the '+' at k_plus[i]; quality pair
    d_qa[q_start[i] : q_end[i]] with the blank at q_space[i].  The index arrays are int64 tensors on the device (numpy arrays are uploaded); the
    reads, keys and quality rows are cut out of the two arenas there."""
    import torch
    from . import fastq_device
    as_dev = lambda x: x if isinstance(x, torch.Tensor) else to_device(np.ascontiguousarray(x, dtype=np.int64), dev)
    k_start, k_plus, k_end, q_start, q_space, q_end = (as_dev(x) for x in (k_start, k_plus, k_end, q_start, q_space, q_end))
    m, k = int(k_start.numel()), len(ref_names)
    U = _Units()
    U.m = m
    l1, l2 = k_plus - k_start, k_end - k_plus - 1
    if m and (int(l1.min().item()) <= 0 or int(l2.min().item()) <= 0):
        raise Exception('global_align: undefined alignment (status %d)' % _native.STATUS_EMPTY)     # (an empty read: the reference indexes seq[-1])
    d_key, d_koff = _dev_gather(ctx, dev, stream, d_ka, k_start, k_end - k_start)
    d_r1, d_o1 = _dev_gather(ctx, dev, stream, d_ka, k_start, l1)
    d_r2, d_o2 = _dev_gather(ctx, dev, stream, d_ka, k_plus + 1, l2)
    max_l = int(torch.maximum(l1.max(), l2.max()).item()) if m else 1
    min_l = int(torch.minimum(l1.min(), l2.min()).item()) if m else 0
    max_key = int((k_end - k_start).max().item()) if m else 1
    # ---- seed test over both reads of the pair (:1024-1036): "seed in read 1 or seed in read 2" = "seed in key" (no seed holds a '+')
    d_plan = torch.empty(m * k, dtype=torch.uint8, device=dev)
    C.strand_plan_device(ctx, m, d_key.data_ptr(), d_koff.data_ptr(), max_key, refs, ref_names, args.aln_seed_count, args.aln_seed_min, d_plan.data_ptr(), stream=stream)
    d_str = (d_plan == 1).to(torch.uint8)
    stride = aligner.stride_for(max_l)
    ostride = 2 * stride
    u8 = torch.uint8

    def align(n_units, d_reads, d_off, max_len, **kw):
        n_items = n_units * (k if kw.get("all_refs") else 1)
        a = torch.empty((n_items, stride), dtype=u8, device=dev)
        f = torch.empty((n_items, stride), dtype=u8, device=dev)
        r = torch.empty((n_items, 32), dtype=u8, device=dev)
        aligner.align_device(n_units, d_reads.data_ptr(), d_off.data_ptr(), a.data_ptr(), f.data_ptr(), r.data_ptr(), stride, max_len, stream=stream, legacy=legacy,
                             min_read_len=min_l, **kw)
        return a, f, r
    A1, F1, R1 = align(m, d_r1, d_o1, max_l, d_strands=d_str.data_ptr(), all_refs=True)
    A2, F2, R2 = align(m, d_r2, d_o2, max_l, d_strands=d_str.data_ptr(), all_refs=True)
    # qualities: one row per read
    ql1, ql2 = q_space - q_start, q_end - q_space - 1
    qstride = max(16, (int(torch.maximum(ql1.max(), ql2.max()).item()) + 15) // 16 * 16) if m else 16
    d_q1, _ = _dev_gather(ctx, dev, stream, d_qa, q_start, ql1, row_stride=qstride)
    d_q2, _ = _dev_gather(ctx, dev, stream, d_qa, q_space + 1, ql2, row_stride=qstride)
    d_lq1, d_lq2 = ql1.to(torch.int32).contiguous(), ql2.to(torch.int32).contiguous()

    def consensus(n_items, a1, f1, r1, a2, f2, r2, q1, q2, lq1, lq2, ref_ids, strands):
        """-> consensus strings (two tensors of ostride-byte rows), records, info"""
        rec1, rec2 = r1.view(torch.int16).view(n_items, 16).to(torch.int64) & 0xffff, r2.view(torch.int16).view(n_items, 16).to(torch.int64) & 0xffff
        bad = ((r1[:, 23] != 0) | (r2[:, 23] != 0))                     # the records' status byte
        if n_items and bool(bad.any().item()):
            st = int(torch.maximum(r1[:, 23], r2[:, 23]).max().item())
            if st & _native.STATUS_RC_CHAR:
                raise KeyError("reverse_complement: a read has a character outside ACGTN_-")
            raise Exception('global_align: undefined alignment (status %d)' % st)
        n1, n2 = rec1[:, 0].to(torch.int32).contiguous(), rec2[:, 0].to(torch.int32).contiguous()
        best1 = (_mscore(rec1[:, 1], rec1[:, 0]) >= _mscore(rec2[:, 1], rec2[:, 0])).to(u8).contiguous()       # is_best_aln_r1, :876
        ca = torch.empty((max(n_items, 1), ostride), dtype=u8, device=dev)
        cf = torch.empty((max(n_items, 1), ostride), dtype=u8, device=dev)
        cq = torch.empty((max(n_items, 1), ostride), dtype=u8, device=dev)
        info = torch.zeros((max(n_items, 1), 4), dtype=torch.int32, device=dev)
        rec = torch.zeros((max(n_items, 1), 32), dtype=u8, device=dev)
        if n_items:
            consensus_device(ctx, n_items, a1.data_ptr(), f1.data_ptr(), a2.data_ptr(), f2.data_ptr(), stride, n1.data_ptr(), n2.data_ptr(), q1.data_ptr(), q2.data_ptr(),
                             qstride, lq1.data_ptr(), lq2.data_ptr(), best1.data_ptr(), ca.data_ptr(), cf.data_ptr(), cq.data_ptr(), ostride, info.data_ptr(), stream)
            if bool(((info[:n_items, 3] & 2) != 0).any().item()):
                raise IndexError('string index out of range')          # (a quality index past the end of its string, as in the reference)
            classify_records_device(ctx, n_items, ca.data_ptr(), cf.data_ptr(), ostride, info.data_ptr(), None if ref_ids is None else ref_ids.data_ptr(),
                                    None if strands is None else strands.data_ptr(), legacy, rec.data_ptr(), stream, refs=refs, ref_names=ref_names)
        return ca[:n_items], cf[:n_items], rec[:n_items], info[:n_items]
    rep = lambda t: t if k == 1 else t.repeat_interleave(k, dim=0)
    U.a, U.f, U.r, info1 = consensus(m * k, A1, F1, R1, A2, F2, R2, rep(d_q1), rep(d_q2), rep(d_lq1), rep(d_lq2), None, d_str)
    del A1, F1, R1, A2, F2, R2
    # ---- the pairs aligned on both strands: reverse-complement alignments of both reads, their consensus
    both = torch.nonzero(d_plan.view(m, k) == 2)
    U.bi, U.br = both[:, 0].contiguous(), both[:, 1].contiguous()
    nb = int(both.shape[0])
    U.nb = nb
    if nb:
        bi_h = to_host(U.bi)
        g1, go1, ml1 = fastq_device.gather_reads_device(ctx, d_r1, d_o1, bi_h, dev, stream)
        g2, go2, ml2 = fastq_device.gather_reads_device(ctx, d_r2, d_o2, bi_h, dev, stream)
        d_rid = U.br.to(torch.int16).contiguous()
        ones = torch.ones(nb, dtype=u8, device=dev)
        B1, BF1, BR1 = align(nb, g1, go1, max(ml1, 1), d_ref_ids=d_rid.data_ptr(), d_strands=ones.data_ptr())
        B2, BF2, BR2 = align(nb, g2, go2, max(ml2, 1), d_ref_ids=d_rid.data_ptr(), d_strands=ones.data_ptr())
        U.a2, U.f2, U.r2, info2 = consensus(nb, B1, BF1, BR1, B2, BF2, BR2, d_q1[U.bi], d_q2[U.bi], d_lq1[U.bi], d_lq2[U.bi], d_rid, ones)
    else:
        z = torch.zeros((0, ostride), dtype=u8, device=dev)
        U.a2, U.f2, U.r2, info2 = z, z.clone(), torch.zeros((0, 32), dtype=u8, device=dev), torch.zeros((0, 4), dtype=torch.int32, device=dev)
    # caching_is_ok of a pair: of the LAST consensus call made for it (:1049): the last amplicon's, its reverse-complement one if both strands ran
    last = (torch.arange(m, device=dev) * k + (k - 1))
    ok = (info1[last, 3] & 1) != 0 if m else torch.zeros(0, dtype=torch.bool, device=dev)
    if nb:
        sel = U.br == (k - 1)
        ok = ok.clone()
        ok[U.bi[sel]] = (info2[sel, 3] & 1) != 0
    U.caching_ok = ok
    U.stride, U.ostride = stride, ostride
    return U


def quantify_paired_fastq(fastq1, fastq2, refs, ref_names, aln_matrix, args, ctx=None, device=0, timings=None):
    """Two FASTQ files read in lock step -> pipeline.QuantResult (per-amplicon count tensors, aln_stats, N_TOTAL / N_AMBIGUOUS, the allele
    table on the device): what the reference's process_paired_fastq + aggregation loop produce, without a dict per pair."""
    import time
    import torch
    from . import fastq_device
    from .batch import BatchAligner
    from .pipeline import QuantResult
    t_last = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
            t_last[0] = now
    legacy = bool(getattr(args, 'use_legacy_insertion_quantification', False))
    if getattr(args, 'prime_editing_pegRNA_scaffold_seq', '') and 'Prime-edited' in ref_names:
        raise PairedDeviceUnavailable("the prime-editing scaffold rule")
    k = len(ref_names)
    if k > 1 and (getattr(args, 'expected_hdr_amplicon_seq', '') or getattr(args, 'prime_editing_pegRNA_extension_seq', '')):
        raise PairedDeviceUnavailable("the first-amplicon view of HDR / prime-editing runs")
    if legacy and any(set(refs[nm]['sequence']) - set('ACGTN') for nm in ref_names):
        raise PairedDeviceUnavailable("use_legacy_insertion_quantification with a reference character outside ACGTN")
    ctx = ctx or _native.default_context()
    dev = torch.device("cuda", device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    aligner = BatchAligner([refs[nm]['sequence'] for nm in ref_names], [refs[nm]['gap_incentive'] for nm in ref_names],
                           [refs[nm]['include_idxs'] for nm in ref_names], aln_matrix, args.needleman_wunsch_gap_open, args.needleman_wunsch_gap_extend, ctx=ctx)
    L = [len(refs[nm]['sequence']) for nm in ref_names]
    front = lambda *idx: _front(ctx, aligner, dev, stream, refs, ref_names, args, legacy, *idx)
    P, why_host = None, None
    if not FORCE_HOST_PARSER:
        # the two texts uploaded as they lie in the files (BGZF: as they are inflated), framed, keyed and de-duplicated on the device: the second
        # pass below is then an index selection on arenas that are already in HBM
        try:
            with fastq_device.IngestSource(fastq1) as S1, fastq_device.IngestSource(fastq2) as S2:
                if S1.source is None or S2.source is None:
                    why_host = S1.why_not or S2.why_not
                else:
                    P = fastq_device.ingest_pairs(S1.source, S2.source, ctx, dev, timings=timings)
        except (fastq_device.DeviceIngestUnavailable, _native.NativeError) as e:   # (carriage returns, a damaged gzip stream, something the reference
            why_host = str(e)                                                      #  raises an error for: the host parser reproduces it)
    else:
        why_host = "C2_PAIRED_HOST_PARSER"
    if P is not None:
        n = P.n_unique
        raw_all = to_host(P.counts).astype(np.int64)
        lap("paired_ingest")
        u = P.uniq_rec
        k_start, q_start = P.key_off[u], P.qual_off[u]
        k_plus, k_end, q_space, q_end = k_start + P.l1[u], P.key_off[u + 1], q_start + P.lq1[u], P.qual_off[u + 1]
        first = front(P.d_keys, k_start, k_plus, k_end, P.d_quals, q_start, q_space, q_end)
        lap("first_pass")
        # ---- keys seen more than once whose consensus chose a base by quality: every occurrence again, with its own qualities (:1450-1513) --
        # the records whose key is one of them, in file order
        d_again = (P.counts > 1) & ~first.caching_ok if n else torch.zeros(0, dtype=torch.bool, device=dev)
        again = to_host(d_again.to(torch.uint8)).astype(bool)
        second = None
        if again.any():
            occ = torch.nonzero(d_again[P.rec_key]).reshape(-1)
            ko = P.rec_key[occ]
            second = front(P.d_keys, k_start[ko], k_plus[ko], k_end[ko], P.d_quals, P.qual_off[occ], P.qual_off[occ] + P.lq1[occ], P.qual_off[occ + 1])
        lap("second_pass")
        ingest_route = "paired, device"
        P = None
    else:
        pf = _native.PairedFastq(fastq1, fastq2)
        try:
            n = pf.n_unique
            raw_all = pf.counts.astype(np.int64)
            ka, ko, qa, qo = pf.arrays()
            ko, qo = ko.astype(np.int64), qo.astype(np.int64)
            plus, blank = np.flatnonzero(ka == 43), np.flatnonzero(qa == 32)
            if len(plus) != n or len(blank) != n or (n and (((plus < ko[:-1]) | (plus >= ko[1:])).any() or ((blank < qo[:-1]) | (blank >= qo[1:])).any())):
                raise ValueError("too many values to unpack (expected 2)")             # key.split('+') / quals.split(' ') of the reference (:1225-1226)
            lap("paired_ingest")
            d_ka = to_device(ka if ka.size else np.zeros(1, dtype=np.uint8), dev)
            d_qa = to_device(qa if qa.size else np.zeros(1, dtype=np.uint8), dev)
            first = front(d_ka, ko[:-1], plus, ko[1:], d_qa, qo[:-1], blank, qo[1:])
            lap("first_pass")
            again = (raw_all > 1) & ~to_host(first.caching_ok.to(torch.uint8)).astype(bool) if n else np.zeros(0, dtype=bool)
            second = None
            if again.any():
                idx_occ, (qa2, qo2) = pf.occurrences(again, as_arrays=True)
                idx_occ, qo2 = idx_occ.astype(np.int64), qo2.astype(np.int64)
                blank2 = np.flatnonzero(qa2 == 32)
                if len(blank2) != len(idx_occ):
                    raise ValueError("too many values to unpack (expected 2)")
                d_qa2 = to_device(qa2 if qa2.size else np.zeros(1, dtype=np.uint8), dev)
                second = front(d_ka, ko[:-1][idx_occ], plus[idx_occ], ko[1:][idx_occ], d_qa2, qo2[:-1], blank2, qo2[1:])
                del d_qa2
            lap("second_pass")
            del d_ka, d_qa
        finally:
            pf.close()
        ingest_route = "paired, keys from the host parser (%s)" % why_host
    # ---- the run's entries: the kept keys in key order, then the occurrences in file order (the order of the reference's cache)
    kept = np.flatnonzero(~again)
    d_kept = to_device(kept, dev)
    kk = torch.arange(k, device=dev)
    rows1 = (d_kept[:, None] * k + kk[None, :]).reshape(-1)
    stride = max(first.ostride, second.ostride if second is not None else 0)
    widen = lambda x, st: x if st == stride else torch.nn.functional.pad(x, (0, stride - st))
    a1, f1, r1 = widen(first.a.index_select(0, rows1), first.ostride), widen(first.f.index_select(0, rows1), first.ostride), first.r.index_select(0, rows1)
    new_of = torch.full((max(n, 1),), -1, dtype=torch.int64, device=dev)
    new_of[d_kept] = torch.arange(len(kept), device=dev)
    keep_b = new_of[first.bi] >= 0
    bi, br = new_of[first.bi][keep_b], first.br[keep_b]
    a2, f2, r2 = widen(first.a2[keep_b], first.ostride), widen(first.f2[keep_b], first.ostride), first.r2[keep_b]
    raw = raw_all[kept]
    N = len(kept)
    if second is not None:
        a1, f1, r1 = torch.cat([a1, widen(second.a, second.ostride)]), torch.cat([f1, widen(second.f, second.ostride)]), torch.cat([r1, second.r])
        bi, br = torch.cat([bi, second.bi + N]), torch.cat([br, second.br])
        a2, f2, r2 = torch.cat([a2, widen(second.a2, second.ostride)]), torch.cat([f2, widen(second.f2, second.ostride)]), torch.cat([r2, second.r2])
        raw = np.concatenate([raw, np.ones(second.m, dtype=np.int64)])
        N += second.m
    del first, second
    n2 = int(bi.numel())
    stats = dict(N_TOT_READS=int(raw.sum()), N_CACHED_ALN=0, N_CACHED_NOTALN=0, N_COMPUTED_ALN=0, N_COMPUTED_NOTALN=0, N_GLOBAL_SUBS=0, N_SUBS_OUTSIDE_WINDOW=0,
                 N_MODS_IN_WINDOW=0, N_MODS_OUTSIDE_WINDOW=0, N_READS_IRREGULAR_ENDS=0, N_TOTAL=0, N_AMBIGUOUS=0)
    max_read = stride                                                 # (the histograms' span: a consensus read is at most as long as its alignment, <= stride columns)
    layout = C.CountLayout(k, max(L), max_read)
    d_counts = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)
    flags = ((C.FLAG_IGNORE_SUBSTITUTIONS if args.ignore_substitutions else 0) | (C.FLAG_IGNORE_INSERTIONS if args.ignore_insertions else 0) |
             (C.FLAG_IGNORE_DELETIONS if args.ignore_deletions else 0) | (C.FLAG_DISCARD_INDEL_READS if getattr(args, 'discard_indel_reads', False) else 0) |
             (C.FLAG_LEGACY_CLASSIFIER if legacy else 0))

    def result(state):
        host = d_counts.cpu().numpy()
        per_ref = {nm: layout.unpack(host, r, L[r]) for r, nm in enumerate(ref_names)}
        res = QuantResult(per_ref, stats, layout, d_counts, state)
        res.stats['N_READS_INPUT'] = res.stats['N_READS_AFTER_PREPROCESSING'] = n_pairs
        res.ingest_route = ingest_route
        return res
    n_pairs = int(raw_all.sum())
    if N == 0:
        return result(None)
    if stride > C.SELECT_MAX_ALN_LEN:
        raise PairedDeviceUnavailable("consensus alignments of %d columns or more" % C.SELECT_MAX_ALN_LEN)
    # ---- best amplicon, ambiguity, aln_stats (:1066-1081, :1382-1437): the selection kernel on the consensus records
    d_slot2 = None
    if n2:
        d_slot2 = torch.full((N * k,), -1, dtype=torch.int32, device=dev)
        d_slot2[bi * k + br] = torch.arange(n2, dtype=torch.int32, device=dev)
    words = (k + 63) // 64
    d_member = torch.zeros((N, words), dtype=torch.int64, device=dev)
    d_use2 = torch.zeros((N, words), dtype=torch.int64, device=dev)
    d_flags = torch.zeros(N, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(len(C.SELECT_STATS), dtype=torch.int64, device=dev)
    if raw.max() > 0x7FFFFFFF:
        raise OverflowError("a read multiplicity exceeds 2^31 - 1")
    d_raw = to_device(raw.astype(np.uint32).view(np.int32), dev)
    min_mscore = C.min_mscore_table([refs[nm]['min_aln_score'] for nm in ref_names])
    mode = C.select_mode(args)
    C.select_best_device(ctx, N, k, r1.data_ptr(), min_mscore, mode, stride, d_records2=r2.data_ptr() if n2 else None, d_slot2=d_slot2.data_ptr() if n2 else None,
                         d_raw_counts=d_raw.data_ptr(), d_member=d_member.data_ptr(), d_use2=d_use2.data_ptr(), d_flags=d_flags.data_ptr(),
                         d_stats=d_stats.data_ptr(), stream=stream)
    st = dict(zip(C.SELECT_STATS, d_stats.cpu().numpy().tolist()))
    if st["n_bad_status"]:
        if int(st["a_bad_status"]) & 128:
            raise PairedDeviceUnavailable("a consensus alignment with a gap in both strings or an insertion next to a deletion")
        raise Exception('global_align: undefined alignment (status %d)' % int(st["a_bad_status"]))
    for q in ('N_COMPUTED_ALN', 'N_COMPUTED_NOTALN', 'N_CACHED_ALN', 'N_CACHED_NOTALN'):
        stats[q] = int(st[q])
    lap("selection")
    # aln_stats of the paired route (:1420-1437): an aligned entry is accounted under every amplicon it is counted for -- unless it is ambiguous
    # and ambiguous alignments are not expanded (the single-read route accounts the last best amplicon whatever the ambiguity)
    cols = torch.arange(k, device=dev)
    mem = ((d_member[:, cols >> 6] >> (cols & 63)[None, :]) & 1).to(torch.bool)
    u2 = ((d_use2[:, cols >> 6] >> (cols & 63)[None, :]) & 1).to(torch.bool)
    aligned = (d_flags & 1) != 0
    nbest = mem.sum(dim=1)
    first_ref = torch.argmax(mem.to(torch.uint8), dim=1)
    names = mem.clone()
    if mode == C.SELECT_FIRST:
        names = torch.zeros_like(mem)
        names[torch.arange(N, device=dev), first_ref] = True
        names &= mem
    accounted = names & aligned[:, None] & ((names.sum(dim=1) == 1) | (mode == C.SELECT_EXPAND))[:, None]
    recs1 = r1.view(torch.int16).view(N, k, 16).to(torch.int64) & 0xffff
    recs = recs1
    if n2:
        recs2 = r2.view(torch.int16).view(n2, 16).to(torch.int64) & 0xffff
        slot = d_slot2.view(N, k).to(torch.int64).clamp(min=0)
        recs = torch.where(u2[:, :, None], recs2[slot], recs1)
    irregular = recs[:, :, 11] & 0xff
    w_acc = accounted.to(torch.int64) * (d_raw.to(torch.int64) & 0xffffffff)[:, None]
    sub_all, sub_win = recs[:, :, 10], recs[:, :, 4]
    in_win = sub_win + recs[:, :, 3] + recs[:, :, 2]
    total_mods = recs[:, :, 5] + recs[:, :, 9] + sub_all
    sums = torch.stack([(sub_all * w_acc).sum(), ((sub_all - sub_win) * w_acc).sum(), (in_win * w_acc).sum(), ((total_mods - in_win) * w_acc).sum(),
                        ((irregular != 0).to(torch.int64) * w_acc).sum()]).cpu().numpy().tolist()
    for q, v in zip(('N_GLOBAL_SUBS', 'N_SUBS_OUTSIDE_WINDOW', 'N_MODS_IN_WINDOW', 'N_MODS_OUTSIDE_WINDOW', 'N_READS_IRREGULAR_ENDS'), sums):
        stats[q] = int(v)
    # ---- pair key -> consensus read (:1439-1448, :1497-1504): the aligned entries' consensus strings (against their first amplicon), first-seen
    # order, equal strings merged into the first of them with their copies added up
    ent = torch.nonzero(aligned).reshape(-1)                          # entries of the cache, in its order
    E = int(ent.numel())
    d_cnt = torch.zeros(N, dtype=torch.int64, device=dev)
    if E:
        r0 = first_ref[ent]
        in2 = u2[ent, r0]
        row1 = ent * k + r0
        T = recs[ent, r0, 0]
        text = torch.cat([a1.reshape(-1), a2.reshape(-1)]) if n2 else a1.reshape(-1)
        base2 = a1.numel()
        start = torch.where(in2, base2 + (d_slot2.to(torch.int64)[row1].clamp(min=0) if n2 else row1 * 0) * stride, row1 * stride)
        seq_s, seq_e = start.contiguous(), (start + T).contiguous()
        n_slots = 1 << 12
        while n_slots < 2 * E:
            n_slots <<= 1
        i64, i32 = torch.int64, torch.int32
        slots = torch.zeros(n_slots, dtype=i64, device=dev)
        count = torch.zeros(n_slots, dtype=i32, device=dev)
        firstrec = torch.full((n_slots,), -1, dtype=i32, device=dev)
        slot_of = torch.zeros(E + 1, dtype=i32, device=dev)
        rinfo = torch.zeros(E + 1, dtype=i64, device=dev)
        dstat = torch.zeros(4, dtype=i32, device=dev)
        dflag = torch.zeros(1, dtype=i32, device=dev)
        rng_t = torch.tensor([0, E], dtype=i64, device=dev)
        fastq_device.fq_dedup(ctx, text.data_ptr(), seq_s.data_ptr(), seq_e.data_ptr(), rng_t.data_ptr(), E, slots.data_ptr(), n_slots, count.data_ptr(),
                              firstrec.data_ptr(), slot_of.data_ptr(), rinfo.data_ptr(), dflag.data_ptr(), dstat.data_ptr(), stream)
        if int(dflag.item()):
            raise _native.NativeError("paired route: the de-duplication of the consensus reads raised flags %d" % int(dflag.item()))
        so = slot_of[:E].to(i64)
        is_rep = (firstrec[so].to(i64) & 0xffffffff) == torch.arange(E, dtype=i64, device=dev)
        wsum = torch.zeros(n_slots, dtype=i64, device=dev)
        wsum.index_add_(0, so, (d_raw.to(i64) & 0xffffffff)[ent])
        c_ent = torch.where(is_rep, wsum[so], torch.zeros(E, dtype=i64, device=dev))
        # reverse-complement merge over the cache (:3970-3975) -- reverse_complement() of an aligned read keeps its '-'
        e_ix = torch.arange(E, dtype=i64, device=dev)
        pslot = torch.empty(E, dtype=i32, device=dev)
        fastq_device.fq_rc_partner(ctx, text.data_ptr(), rinfo.data_ptr(), e_ix.data_ptr(), E, slots.data_ptr(), n_slots, pslot.data_ptr(), stream)
        rep_of_slot = firstrec.to(i64) & 0xffffffff
        partner = torch.where(pslot >= 0, rep_of_slot[pslot.to(i64).clamp(min=0)], torch.full((E,), -1, dtype=i64, device=dev))
        partner = torch.where(is_rep, partner, torch.full((E,), -1, dtype=i64, device=dev))
        c_host = np.ascontiguousarray(to_host(c_ent))
        _native.merge_counts_with_partners(np.ones(E, dtype=bool), np.ascontiguousarray(to_host(partner)), c_host)
        if E and c_host.max() > 0x7FFFFFFF:
            raise OverflowError("a read multiplicity exceeds 2^31 - 1")
        d_cnt[ent] = to_device(c_host, dev)
    lap("rekey_and_merge")
    amb = (d_flags & 2) != 0
    stats['N_TOTAL'] = int((d_cnt * aligned).sum().item())
    if mode == C.SELECT_DROP_AMBIGUOUS:
        stats['N_AMBIGUOUS'] = int((d_cnt * amb).sum().item())
    # ---- weights and count vectors, as on the single-read route
    d_cnt32 = d_cnt.to(torch.int32)
    d_w1 = torch.zeros(N * k, dtype=torch.int32, device=dev)
    d_w2 = torch.zeros(n2, dtype=torch.int32, device=dev) if n2 else None
    C.select_best_device(ctx, N, k, r1.data_ptr(), min_mscore, mode, stride, d_records2=r2.data_ptr() if n2 else None, d_slot2=d_slot2.data_ptr() if n2 else None,
                         d_counts=d_cnt32.data_ptr(), d_weights=d_w1.data_ptr(), d_weights2=d_w2.data_ptr() if n2 else None, stream=stream)
    C.accumulate_device(ctx, layout, N * k, a1.data_ptr(), f1.data_ptr(), stride, r1.data_ptr(), d_counts.data_ptr(), d_weights=d_w1.data_ptr(),
                        flags=flags | C.FLAG_ALL_REFS_LAYOUT, stream=stream)
    if n2:
        C.accumulate_device(ctx, layout, n2, a2.data_ptr(), f2.data_ptr(), stride, r2.data_ptr(), d_counts.data_ptr(), d_weights=d_w2.data_ptr(), flags=flags, stream=stream)
    torch.cuda.synchronize(dev)
    lap("count_kernels")
    state = dict(ctx=ctx, stream=stream, args=args, ref_names=list(ref_names), n=N, mode=mode, flags=flags & 15, a1=a1, f1=f1, r1=r1, stride=stride,
                 a2=a2 if n2 else None, f2=f2 if n2 else None, r2=r2 if n2 else None, stride2=stride, d_slot2=d_slot2, d_member=d_member, d_use2=d_use2,
                 d_flags=d_flags, d_cnt=d_cnt32, d_scaffold_hit=None, scaffold_ref=-1)
    return result(state)
