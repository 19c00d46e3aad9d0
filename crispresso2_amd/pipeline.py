"""FASTQ -> per-amplicon count tensors, end to end on the MI355X: the count outputs of a CRISPResso run
(`process_fastq` CRISPRessoCORE.py:1735-2000 + the "Quantifying indels/substitutions" loop :3964-4115) without building a
Python object per read.

    ingest + exact de-duplication      c2_fastq_unique (host, native)                         :1820-1849
    strand plan per (read, reference)  the seed test of get_new_variant_object                  :656-687
    alignments                         one all-references batch on the device (+ one small batch for the pairs whose
                                       seeds are inconclusive: those are aligned on both strands, :675-687)
    best reference / ambiguity         c2_select_best_kernel on the device: the reference's score comparisons as integers (1000 x the
                                       rounded score), strand choice, ambiguity, aln_stats                        :683, :689-707, :779-785
    reverse-complement merge           :3970-3975
    count vectors                      c2_count_vectors_kernel with the read multiplicities as weights  :3996-4115
    aln_stats                          :1974-1979 from the 32-byte records

Per read 17 bytes come back to the host (two 64-bit masks and a flag byte); records and aligned strings stay in HBM for the
count kernel (QuantResult.alleles() fetches the rows it prints).  The per-read
dict path (variants.get_new_variant_objects) remains for callers that need the reference's per-read payloads.
"""
import os

import numpy as np

from . import _native
from . import counts as C
from .hostcopy import to_host, to_device
from .batch import BatchAligner, score_from_counts

def strand_plans(arena, offsets, refs, ref_names, args):
    """uint8 [n, k]: 0 forward only, 1 reverse complement only, 2 both (the seed test of CRISPRessoCORE.py:656-687), by the
    native threaded c2_strand_plan."""
    n = len(offsets) - 1
    plan = np.empty((n, len(ref_names)), dtype=np.uint8)
    for r, name in enumerate(ref_names):
        m = min(args.aln_seed_count, len(refs[name]['fw_seeds']))
        plan[:, r] = _native.strand_plan(arena, offsets, refs[name]['fw_seeds'][:m], refs[name]['rc_seeds'][:m], args.aln_seed_min)
    return plan


class QuantResult:
    """per_ref[name]: the dict of counts.CountLayout.unpack (vectors named after the reference's variables);
    stats: N_TOT_READS, N_CACHED_ALN, ... (process_fastq's aln_stats) plus N_TOTAL and N_AMBIGUOUS of the aggregation loop;
    allele_table(): the allele frequency table on the device (alleles.AlleleTable: sorted rows, the text of the files);
    alleles(): its rows as tuples."""
    def __init__(self, per_ref, stats, layout, tensor, state=None, first_ref_view=None):
        self.per_ref, self.stats, self.layout, self.tensor = per_ref, stats, layout, tensor
        # what the allele table is built from: a dict of the device tensors the run left (alleles.AlleleTable's arguments), or None
        self._state = state
        self._table = None
        # {name: all_* count vectors of the reads counted for that amplicon, in the coordinates of the FIRST amplicon}
        # (CRISPRessoCORE.py:4195-4270; built for runs with an expected HDR amplicon / prime-editing extension), else None
        self.first_ref_view = first_ref_view

    @property
    def align_ref_names(self):
        """the amplicons reads were aligned to (the allele table's labels index them)"""
        return None if self._state is None else self._state["ref_names"]

    def allele_table(self):
        """alleles.AlleleTable over this run's alignments: rows built and sorted on the device (c2_allele_table_build) the first time it
        is asked for; None for a run without reads (an empty shard)."""
        if self._table is None and self._state is not None:
            from .alleles import AlleleTable
            S = self._state
            self._table = AlleleTable(S["ctx"], S["n"], len(S["ref_names"]), S["mode"], S["flags"], S["a1"], S["f1"], S["r1"], S["stride"],
                                      S["d_member"], S["d_flags"], S["d_cnt"], a2=S["a2"], f2=S["f2"], r2=S["r2"], stride2=S["stride2"],
                                      slot2=S["d_slot2"], use2=S["d_use2"], scaffold_hit=S["d_scaffold_hit"], scaffold_ref=S["scaffold_ref"],
                                      stream=S["stream"], keep=S)
        return self._table

    def host_view(self):
        """the selection as numpy arrays, for callers that look at single reads: member / use2 bool [n, k] (best references; those whose
        reverse-complement alignment won), aligned bool [n], cnt int64 [n] (multiplicities after the reverse-complement transfer),
        slot2 int64 [n, k] (row in the both-strand batch, -1: none)"""
        S = self._state
        n, k = S["n"], len(S["ref_names"])
        cols = np.arange(k)

        def unpack(t):
            w = to_host(t).view(np.uint64).reshape(n, -1)
            return ((w[:, cols >> 6] >> (cols & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)
        slot2 = np.full((n, k), -1, dtype=np.int64) if S["d_slot2"] is None else to_host(S["d_slot2"]).astype(np.int64).reshape(n, k)
        return dict(member=unpack(S["d_member"]), use2=unpack(S["d_use2"]) if S["d_use2"] is not None else np.zeros((n, k), dtype=bool),
                    aligned=(to_host(S["d_flags"]) & 1).astype(bool), cnt=to_host(S["d_cnt"]).view(np.uint32).astype(np.int64), slot2=slot2)

    def allele_rows(self):
        """alleles.AlleleRows: the sorted table as numpy columns in host memory (aligned strings of every aligned unique read)"""
        t = self.allele_table()
        if t is None:
            return None
        return t.rows(self._state["ref_names"], self.stats["N_TOTAL"])

    def alleles(self, gather=False):
        """Rows (Aligned_Sequence, Reference_Sequence, Reference_Name, Read_Status, n_deleted, n_inserted, n_mutated, #Reads,
        %Reads) of Alleles_frequency_table.txt: one per (read, reference it counts for), 'AMBIGUOUS_<first reference>' rows
        for ambiguous reads, 'DISCARDED_<first reference>' rows under --discard_indel_reads (CRISPRessoCORE.py:3926-4010,
        :4298-4303), sorted as the reference sorts them (#Reads descending, then the two sequences ascending) -- on the device
        (allele_table); this brings the sorted rows to the host as columns and zips them into tuples.
        gather (sharded run, a collective: every rank calls it): the rows of all ranks' shards, on every rank."""
        rows = self.allele_rows()
        if not gather:
            return [] if rows is None else rows.tuples()
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return [] if rows is None else rows.tuples()
        from .alleles import AlleleRows
        mine = None if rows is None else (rows.rows, rows.aligned, rows.reference, rows.labels)
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, mine)
        parts = [p_ for p_ in parts if p_ is not None and len(p_[0])]
        if not parts:
            return []
        width = max(p_[1].dtype.itemsize for p_ in parts)
        S = "S%d" % width
        rec = np.concatenate([p_[0] for p_ in parts])
        a = np.concatenate([p_[1].astype(S) for p_ in parts])
        f = np.concatenate([p_[2].astype(S) for p_ in parts])
        order = np.lexsort((f, a, -rec["reads"].astype(np.int64)))       # (stable: ties keep rank order, then shard order)
        return AlleleRows(rec[order], a[order], f[order], parts[0][3], self.stats["N_TOTAL"]).tuples()


def _pack_masks(b):
    """bool [n, k] -> int64 [n, ceil(k / 64)] (bit r % 64 of word r / 64: column r), the selection kernel's mask layout"""
    n, k = b.shape
    words = (k + 63) // 64
    padded = np.zeros((n, words * 64), dtype=np.uint8)
    padded[:, :k] = b
    return np.packbits(padded, axis=1, bitorder='little').view('<u8').view(np.int64).reshape(n, words)


FORCE_HOST_SELECTION = False        # tests: run the host restatement of the selection (_select_on_host) instead of c2_select_best_kernel
FORCE_HOST_STRAND_PLAN = False      # tests: the host's c2_strand_plan instead of c2_strand_plan_device
FORCE_HOST_MERGE = False            # tests: the reverse-complement count transfer and the weights on the host (the reference's sequential loop) instead of the device pass


def _select_on_host(r1, r2, n, k, n2, bi, br, slot2, min_scores, raw, stats):
    """Strand / best-reference choice and aln_stats from the records on the host, with the reference's float comparisons on
    scores formed by its own expression -- the route for alignments of 8000 columns and more (what c2_select_best_kernel's
    integer scores do not cover), and the restatement the tests compare the kernel with (FORCE_HOST_SELECTION).  -> member, use2, aligned"""
    rec1 = r1.cpu().numpy().view(_native.REC_DTYPE).reshape(n, k)
    rec2 = r2.cpu().numpy().view(_native.REC_DTYPE).reshape(-1) if n2 else None
    for rec in (rec1.reshape(-1), rec2 if rec2 is not None else rec1.reshape(-1)[:0]):
        bad = rec["status"] != 0
        if bad.any():
            st = int(rec["status"][np.nonzero(bad)[0][0]])
            if st & _native.STATUS_RC_CHAR:
                raise KeyError("reverse_complement: a read has a character outside ACGTN_-")
            raise Exception('global_align: undefined alignment (status %d)' % st)
    score = score_from_counts(rec1["matches"].reshape(-1), rec1["aln_len"].reshape(-1)).reshape(n, k)
    use2 = np.zeros((n, k), dtype=bool)
    if n2:
        s2 = score_from_counts(rec2["matches"], rec2["aln_len"])
        better = s2 > score[bi, br]                              # strict: ties keep the forward alignment
        use2[bi[better], br[better]] = True
        score[bi[better], br[better]] = s2[better]
    best = np.full(n, -1.0)
    member = np.zeros((n, k), dtype=bool)
    for r in range(k):
        s_r = score[:, r]
        c1 = (s_r > best) & (s_r > min_scores[r])
        best = np.where(c1, s_r, best)
        member[c1, :] = False
        member[c1, r] = True
        member[~c1 & (s_r == best), r] = True
    aligned = best > 0
    member[~aligned, :] = False
    stats['N_COMPUTED_ALN'] = int(aligned.sum())
    stats['N_COMPUTED_NOTALN'] = int(n - aligned.sum())
    stats['N_CACHED_ALN'] = int((raw[aligned] - 1).sum())
    stats['N_CACHED_NOTALN'] = int((raw[~aligned] - 1).sum())
    # aln_stats use the payload of the LAST best match (new_variant['best_match_name'], :764) and the raw multiplicity
    last_best = np.where(aligned, k - 1 - np.argmax(member[:, ::-1], axis=1), 0)
    ii = np.nonzero(aligned)[0]

    def field(name):
        v1 = rec1[name][ii, last_best[ii]].astype(np.int64)
        if n2:
            u = use2[ii, last_best[ii]]
            v1[u] = rec2[name][slot2[ii[u], last_best[ii[u]]]].astype(np.int64)
        return v1
    c_raw = raw[ii]
    sub_all, sub_win = field("all_substitutions"), field("substitution_n")
    total_mods = field("all_insertion_events") + field("all_deletion_bases") + sub_all
    in_win = sub_win + field("deletion_n") + field("insertion_n")
    stats['N_GLOBAL_SUBS'] = int((sub_all * c_raw).sum())
    stats['N_SUBS_OUTSIDE_WINDOW'] = int(((sub_all - sub_win) * c_raw).sum())
    stats['N_MODS_IN_WINDOW'] = int((in_win * c_raw).sum())
    stats['N_MODS_OUTSIDE_WINDOW'] = int(((total_mods - in_win) * c_raw).sum())
    stats['N_READS_IRREGULAR_ENDS'] = int((field("irregular_ends") * c_raw).sum())
    return member, use2, aligned


RC_PARTNERS_ON_DEVICE_MIN = 200_000   # unique reads (all of one length) from which the reverse-complement partner search runs on the device


class _DevicePartners:
    """what rc_partners_device enqueued: result() waits for it -> int64 [n] on the host, or None (a proposed pair failed the byte
    comparison: two different reads with one hash -- the caller takes the host search)"""
    def __init__(self, partner, bad):
        self._partner, self._bad = partner, bad

    def result(self):
        if bool(self._bad.item()):
            return None
        return to_host(self._partner)

    def partner_tensor(self):
        """the same on the device (int64 [n]), or None"""
        return None if bool(self._bad.item()) else self._partner


def rc_partners_device(d_reads2d):
    """The partner search of the count merge (CRISPRessoCORE.py:3970-3975: which unique read equals reverse_complement(read i);
    CRISPRessoShared.py:399-403: upper-cased first, ACGTN_- only) for reads of ONE length that are already in HBM as a [n, L] byte
    matrix: a 64-bit hash of every read and of every reverse complement (weighted sum of their 8-byte words), one sort + binary
    search to pair equal hashes, then the candidate's BYTES compared with the reverse complement -- the hash only proposes, equality
    decides.  Everything is enqueued on the current stream and nothing waits: -> _DevicePartners, whose result() is asked for when
    the merge needs it (a few device passes over n x L bytes, ~20 ms for 3.5 M reads, while the host prepares the next launches;
    the host search costs ~0.5 us per read and core)."""
    import torch
    n, L = d_reads2d.shape
    dev = d_reads2d.device
    R = d_reads2d
    comp = torch.zeros_like(R)
    for src, dst in (("A", "T"), ("C", "G"), ("G", "C"), ("T", "A"), ("N", "N"), ("_", "_"), ("-", "-"), ("a", "T"), ("c", "G"), ("g", "C"), ("t", "A"), ("n", "N")):
        comp.masked_fill_(R == ord(src), ord(dst))
    valid = (comp != 0).all(dim=1)
    rcb = comp.flip(1).contiguous()
    del comp
    W = (L + 7) // 8
    g = torch.Generator(device="cpu")
    g.manual_seed(0x5eed)
    w = (torch.randint(1, 1 << 62, (W,), generator=g, dtype=torch.int64) * 2 + 1).to(dev)

    def hash_rows(M):                                                  # [n, L] bytes -> int64 [n]: the rows' 8-byte words, mixed, weighted, summed (wrap-around)
        P = M if L == 8 * W else torch.nn.functional.pad(M, (0, 8 * W - L))
        x = P.contiguous().view(torch.int64).view(n, W)
        x = x ^ (x >> 29)
        return (x * w).sum(dim=1)
    h, hr = hash_rows(R), hash_rows(rcb)
    hs, order = torch.sort(h)
    pos = torch.searchsorted(hs, hr).clamp(max=n - 1)
    hit = (hs[pos] == hr) & valid
    cand = order[pos]
    same = torch.empty(n, dtype=torch.bool, device=dev)
    CH = 1 << 20
    for a0 in range(0, n, CH):
        a1 = min(n, a0 + CH)
        same[a0:a1] = (R[cand[a0:a1]] == rcb[a0:a1]).all(dim=1)
    partner = torch.where(hit & same, cand, torch.full_like(cand, -1))
    return _DevicePartners(partner, (hit & ~same).any())


STREAM_MIN_BATCH = 200_000          # unique reads: smaller arrivals wait for the next chunk (a launch chain per chunk is not free)


def _enqueue_first_batch(aligner, ctx, dev, refs, ref_names, args, legacy, m, d_reads, d_off, max_lj, stream):
    """m reads that are on the device (arena + int64 offsets): the seed test (c2_strand_plan_kernel) and their alignments against every
    reference on the strand it asks for, enqueued on `stream` -> (aligned reads, aligned refs, records, plan, stride, reads, offsets, strands)"""
    import torch
    k = len(ref_names)
    d_plan = torch.empty(m * k, dtype=torch.uint8, device=dev)
    C.strand_plan_device(ctx, m, d_reads.data_ptr(), d_off.data_ptr(), max_lj, refs, ref_names, args.aln_seed_count, args.aln_seed_min,
                         d_plan.data_ptr(), stream=stream)
    d_str = (d_plan == 1).to(torch.uint8)
    stride = aligner.stride_for(max_lj)
    a = torch.empty((m * k, stride), dtype=torch.uint8, device=dev)
    f = torch.empty((m * k, stride), dtype=torch.uint8, device=dev)
    r = torch.empty((m * k, 32), dtype=torch.uint8, device=dev)
    aligner.align_device(m, d_reads.data_ptr(), d_off.data_ptr(), a.data_ptr(), f.data_ptr(), r.data_ptr(), stride, max_lj,
                         d_strands=d_str.data_ptr(), all_refs=True, stream=stream, legacy=legacy)
    return (a, f, r, d_plan, stride, d_reads, d_off, d_str)


def _join_first_batches(parts, aligner, dev):
    """the batches' outputs as one (rows of narrower batches padded to the widest stride) -> a1, f1, r1, d_plan, stride"""
    import torch
    stride = max([p_[4] for p_ in parts], default=aligner.stride_for(1))

    def widen(x, st):
        return x if st == stride else torch.nn.functional.pad(x, (0, stride - st))
    if len(parts) == 1:
        return parts[0][0], parts[0][1], parts[0][2], parts[0][3], stride
    if parts:
        return (torch.cat([widen(p_[0], p_[4]) for p_ in parts]), torch.cat([widen(p_[1], p_[4]) for p_ in parts]),
                torch.cat([p_[2] for p_ in parts]), torch.cat([p_[3] for p_ in parts]), stride)
    return (torch.empty((0, stride), dtype=torch.uint8, device=dev), torch.empty((0, stride), dtype=torch.uint8, device=dev),
            torch.empty((0, 32), dtype=torch.uint8, device=dev), torch.empty(0, dtype=torch.uint8, device=dev), stride)


def _device_front(path, aligner, ctx, dev, refs, ref_names, args, legacy, timings):
    """_stream_front for text that is framed and de-duplicated ON the device (fastq_device.ingest_file): whenever STREAM_MIN_BATCH new
    unique reads are final, their seed test and alignments are enqueued behind the de-duplication kernels -- the device works on them
    while the host copies the next chunks of the text into pinned memory and the link carries them.  The reads never are on the host
    (`arena` is None; "device_reads" carries them for the steps that need their bytes).  Raises DeviceIngestUnavailable."""
    import time
    import torch
    from . import fastq_device
    t0 = time.perf_counter()
    k = len(ref_names)
    parts = []
    stream = torch.cuda.current_stream(dev).cuda_stream

    def on_batch(m, d_reads, d_off, max_len):
        parts.append(_enqueue_first_batch(aligner, ctx, dev, refs, ref_names, args, legacy, m, d_reads, d_off, max(max_len, 1), stream))
    ing = fastq_device.ingest_file(path, ctx, dev, timings=timings, on_batch=on_batch, min_batch=STREAM_MIN_BATCH)
    n = ing["n_unique"]
    a1, f1, r1, d_plan, stride = _join_first_batches(parts, aligner, dev)
    plan = to_host(d_plan).reshape(n, k)
    both = to_host(torch.nonzero(d_plan.view(n, k) == 2)) if n else np.zeros((0, 2), dtype=np.int64)   # (the few pairs aligned on both strands)
    if len(parts) == 1:
        d_reads_all, d_off_all = parts[0][5], parts[0][6]
    elif parts:
        d_reads_all = torch.cat([p_[5][:b] for p_, b in zip(parts, ing["batch_bytes"])])
        d_off_all = torch.from_numpy(ing["offsets"].view(np.int64)).to(dev)
    else:
        d_reads_all, d_off_all = torch.zeros(1, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    del parts
    if timings is not None:
        timings["stream_batches"] = len(ing["batch_bytes"])
        timings["device_front"] = time.perf_counter() - t0
    ing["d_reads"], ing["d_off"] = d_reads_all, d_off_all
    # (the reverse-complement partners come with the ingest: looked up in its table -- no second search)
    return dict(arena=None, offsets=ing["offsets"], counts=ing["counts"], plan=plan, stride=stride, a1=a1, f1=f1, r1=r1,
                rc_partners=lambda: ing["rc_partner"], d_reads_all=None, device_reads=ing, both=both)


def _stream_front(fq, aligner, ctx, dev, refs, ref_names, args, legacy, timings):
    """Ingest and device overlapped (replaces "parse the whole file, then align", CRISPRessoCORE.py:1825-1849 + :1957-1981).
    A host thread drives the native chunked parser (_native.FastqStream.next: all cores, GIL released); whenever it has brought
    STREAM_MIN_BATCH new unique reads, this thread copies them through pinned staging to the device on a copy stream and enqueues
    the seed test (c2_strand_plan_kernel) and the all-references alignment batch for exactly those reads on the compute stream.
    Nothing waits for the device until the file is exhausted.  -> the unique reads (arena, offsets, final multiplicities), the
    strand plan and the concatenated outputs of batch 1 (task = read * n_refs + reference, as in the one-batch flow)."""
    import queue
    import threading
    import time
    import torch
    t_start = time.perf_counter()
    k = len(ref_names)
    on_gpu = dev.type == "cuda"
    compute = torch.cuda.current_stream(dev)
    copy_stream = torch.cuda.Stream(device=dev) if on_gpu else compute
    q = queue.Queue()

    stop = threading.Event()                                          # set when the consumer gave up: the parser ends after the chunk in hand

    def producer():
        try:
            done = fq.done
            seen = 0
            while not done and not stop.is_set():
                nu, done = fq.next()
                if nu - seen >= STREAM_MIN_BATCH or (done and nu > seen):
                    q.put((seen, nu, fq.offsets_slice(seen, nu)))
                    seen = nu
            q.put(None)
        except BaseException as e:
            q.put(e)
    th = threading.Thread(target=producer, name="c2-fastq-stream")
    th.start()
    staging = [None, None]                # pinned (arena bytes, offsets) buffers, alternating; an event says when the copy out of one is done
    staged_ev = [None, None]
    parts, off_parts, dropped = [], [], []
    turn = 0
    t_ingest_done = None
    try:
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            n0, n1, off = item
            empty = np.nonzero(off[1:] == off[:-1])[0]
            if len(empty):
                # the empty sequence (blank line / truncated record; at most one unique read): dropped, as in quantify_fastq's one-batch
                # flow (the reference's aligner indexes seq[-1] of an empty string: undefined there)
                dropped.extend((n0 + empty).tolist())
                off = np.delete(off, empty + 1)
            m = len(off) - 1
            if m == 0:
                continue
            base = int(off[0])
            nbytes = int(off[-1]) - base
            lens = (off[1:] - off[:-1]).astype(np.int64)
            max_lj = max(int(lens.max()), 1)
            rel = (off - off[0]).astype(np.int64)
            i = turn & 1
            turn += 1
            if staged_ev[i] is not None:
                staged_ev[i].synchronize()
            if on_gpu:
                if staging[i] is None or staging[i][0].numel() < nbytes or staging[i][1].numel() < m + 1:
                    staging[i] = (torch.empty(max(nbytes, 1) * 5 // 4, dtype=torch.uint8, pin_memory=True),
                                  torch.empty((m + 1) * 5 // 4, dtype=torch.int64, pin_memory=True))
                hb, ho = staging[i][0][:max(nbytes, 1)], staging[i][1][:m + 1]
                hb.numpy()[:nbytes] = fq.arena[base:base + nbytes]
                ho.numpy()[:] = rel
                with torch.cuda.stream(copy_stream):
                    d_reads = hb.to(dev, non_blocking=True)
                    d_off = ho.to(dev, non_blocking=True)
                    staged_ev[i] = torch.cuda.Event()
                    staged_ev[i].record(copy_stream)
                compute.wait_event(staged_ev[i])
                d_reads.record_stream(compute)
                d_off.record_stream(compute)
            else:                                                     # (tests: the "device" is host memory)
                d_reads = torch.from_numpy(np.ascontiguousarray(fq.arena[base:base + max(nbytes, 1)]).copy())
                d_off = torch.from_numpy(rel.copy())
            parts.append(_enqueue_first_batch(aligner, ctx, dev, refs, ref_names, args, legacy, m, d_reads, d_off, max_lj, compute.cuda_stream))
            off_parts.append(off)
    except BaseException:
        stop.set()
        raise
    finally:
        th.join()
    t_ingest_done = time.perf_counter()
    n = fq.n_unique - len(dropped)
    if off_parts:
        offsets = np.concatenate([off_parts[0]] + [o[1:] for o in off_parts[1:]]).astype(np.uint64)
    else:
        offsets = np.zeros(1, dtype=np.uint64)
    counts = fq.counts()
    if dropped:
        counts = np.delete(counts, dropped)
    arena = fq.arena[:int(offsets[-1])]
    a1, f1, r1, d_plan, stride = _join_first_batches(parts, aligner, dev)
    plan = to_host(d_plan).reshape(n, k)                               # (waits for the last batch)
    d_reads_all = None
    if parts and n >= RC_PARTNERS_ON_DEVICE_MIN:                       # (kept for the partner search on the device)
        d_reads_all = parts[0][5] if len(parts) == 1 else torch.cat([p_[5][:int(o[-1] - o[0])] for p_, o in zip(parts, off_parts)])
    del parts
    if timings is not None:
        timings["ingest_dedup_streamed"] = t_ingest_done - t_start
        timings["stream_tail_device"] = time.perf_counter() - t_ingest_done
        timings["stream_batches"] = turn
    return dict(arena=arena, offsets=offsets, counts=counts, plan=plan, stride=stride, a1=a1, f1=f1, r1=r1,
                rc_partners=None if dropped else fq.rc_partners, d_reads_all=d_reads_all)


def quantify_unique(arena, offsets, read_counts, refs, ref_names, aln_matrix, args, ctx=None, device=0, reduce_across_ranks=False,
                    timings=None, pe_scaffold_dna_info=None, shard=None, fastq_stream=None, device_reads=None):
    """Unique reads -> QuantResult; see _UniqueRun for the arguments and for the stages.  (This wrapper also makes sure that the host thread the
    run starts -- it reads `arena`, which may be a view of native memory the caller frees -- has ended before control returns, also when the
    run raises.)"""
    run = _UniqueRun(refs, ref_names, aln_matrix, args, ctx, device, reduce_across_ranks, timings, pe_scaffold_dna_info)
    try:
        return run.run(arena, offsets, read_counts, shard, fastq_stream, device_reads)
    finally:
        for t in run.threads:
            t.join()


class _UniqueRun:
    """One run of the count route over a list of unique reads, as named stages over explicit state (the attributes below).

    arena / offsets / read_counts: the unique reads (c2_fastq_unique layout) and their multiplicities.
    shard: None -- this process aligns every read it was given.  "mine" (with device_reads): this rank's contiguous range
    (distributed.my_shard) of the unique reads the device ingest finds.  Otherwise arena / offsets / read_counts are the WHOLE run's unique
    reads (every rank holds the same list, as every worker of the reference sees the parent's variantCache keys) and `shard` names
    the part this rank aligns: (lo, hi) for a contiguous range (distributed.my_shard = get_variant_cache_equal_boundaries) or an
    index array.  The reverse-complement merge (:3970-3975) then runs over the whole list exactly as in one process: partners are
    looked up among ALL unique reads, the ranks exchange which of their reads aligned (one byte per unique read, all-reduced),
    and every rank applies the reference's sequential count transfer to the global counts before it weighs its own reads -- so the
    all-reduced tensors equal the single-process ones also when a read and its reverse complement land in different shards.
    Implies reduce_across_ranks.
    fastq_stream: a _native.FastqStream instead of arena / offsets / read_counts -- the file is parsed chunk by chunk on a host thread
    while the device already runs the seed test and the alignments of the unique reads the previous chunks brought (_stream_front).
    device_reads: instead of arena / offsets / read_counts -- the path of a FASTQ file that is framed and de-duplicated on the device
    with batch 1 running under its upload (_device_front; raises fastq_device.DeviceIngestUnavailable), or fastq_device.ingest_file's
    result (one batch).  The unique reads never are on the host (`arena` stays None; the rare host-side uses download them).
    timings: optional dict that receives the wall seconds of every stage.
    pe_scaffold_dna_info: (index, dna) of get_pe_scaffold_search for runs with --prime_editing_pegRNA_scaffold_seq: reads whose
    alignment against 'Prime-edited' carries `dna` right after reference base index-1 are counted for 'Scaffold-incorporated'
    (a copy of the Prime-edited amplicon that nothing is aligned to, CRISPRessoCORE.py:786-796, :3759-3764) -- the result then
    has that extra amplicon.

    Stages, in the order run() calls them:
        take_input            the reads (host arena / parsed stream / device ingest) and, for a sharded run, this rank's part of them
        prepare               tensor layout agreed across ranks, count tensor, statistics; an empty shard ends here (empty_shard)
        reads_to_device       the arena uploaded (unless a front already did)
        start_partner_search  which read is the reverse complement of which: on the device, or on a host thread under the alignments
        align_all_references  seed test + batch 1 (every read x every amplicon on the strand the seeds ask for)
        align_both_strands    batch 2 (the pairs whose seeds were inconclusive, reverse complement)
        select                best amplicon, strand, ambiguity, aln_stats: c2_select_best_kernel (or the host's comparisons)
        finish_on_device      the usual run's tail without the host: count transfer over partner pairs, weights, count launches
        merge_on_host         otherwise: the reference's sequential count transfer (over the WHOLE list when sharded), ambiguity rules
        scaffold_hits         the prime-editing scaffold rule
        count                 weights + count launches (+ the 'Scaffold-incorporated' tensor)
        first_amplicon_view   every amplicon's reads in the first amplicon's coordinates (HDR / prime-editing runs)
        reduce                all-reduce of the tensors and statistics
        result                -> QuantResult with the device state the allele table is built from
    """

    def __init__(self, refs, ref_names, aln_matrix, args, ctx, device, reduce_across_ranks, timings, pe_scaffold_dna_info):
        import time
        import torch
        self.refs, self.ref_names, self.args, self.timings = refs, list(ref_names), args, timings
        self.reduce_across_ranks, self.pe_scaffold_dna_info = reduce_across_ranks, pe_scaffold_dna_info
        self.threads = []
        self._t_last = time.perf_counter()
        self.legacy = bool(getattr(args, 'use_legacy_insertion_quantification', False))
        if self.legacy:
            # find_indels_substitutions_legacy (COREResources.pyx:190-315) on the count route: the fused classifier and the count kernel
            # follow its rules (an insertion counts when EITHER flank is in the window; its reference coordinates of a deletion that starts
            # in column 0 / 1 or reaches the end).  Its `nucSet` treats any other reference character as a gap, which the kernels do not.
            # CRISPRessoCORE never gets here with such an amplicon: it refuses a reference character outside ACGTN before any read is
            # aligned (CRISPRessoCORE.py:3054-3059, NTException "contains invalid characters"), so only a direct caller of this module can.
            for name in ref_names:
                if set(refs[name]['sequence']) - set('ACGTN'):
                    raise NotImplementedError("use_legacy_insertion_quantification with a reference character outside ACGTN: "
                                              "use variants.process_fastq (per-read route) for this run")
        self.scaffold_rule = bool(getattr(args, 'prime_editing_pegRNA_scaffold_seq', '')) and 'Prime-edited' in ref_names
        if self.scaffold_rule and (pe_scaffold_dna_info is None or pe_scaffold_dna_info[1] is None):
            raise ValueError("prime_editing_pegRNA_scaffold_seq needs pe_scaffold_dna_info = (index, dna) of get_pe_scaffold_search")
        self.pe = ref_names.index('Prime-edited') if self.scaffold_rule else -1
        self.ctx = ctx or _native.default_context()
        self.dev = torch.device("cuda", device)
        self.aligner = BatchAligner([refs[name]['sequence'] for name in ref_names], [refs[name]['gap_incentive'] for name in ref_names],
                                    [refs[name]['include_idxs'] for name in ref_names], aln_matrix,
                                    args.needleman_wunsch_gap_open, args.needleman_wunsch_gap_extend, ctx=self.ctx)
        self.k = len(ref_names)
        self.L = [len(refs[name]['sequence']) for name in ref_names]
        self.want_view = self.k > 1 and bool(getattr(args, 'expected_hdr_amplicon_seq', '') or getattr(args, 'prime_editing_pegRNA_extension_seq', ''))
        self.flags = ((C.FLAG_IGNORE_SUBSTITUTIONS if args.ignore_substitutions else 0) | (C.FLAG_IGNORE_INSERTIONS if args.ignore_insertions else 0) |
                      (C.FLAG_IGNORE_DELETIONS if args.ignore_deletions else 0) | (C.FLAG_DISCARD_INDEL_READS if getattr(args, 'discard_indel_reads', False) else 0) |
                      (C.FLAG_LEGACY_CLASSIFIER if self.legacy else 0))
        self.front = None                                              # what _stream_front / _device_front already did (seed test, batch 1)
        self.d_view = self.d_scaffold = None
        self.scaffold_hit = None

    # ---- helpers ----
    def lap(self, name):
        import time
        import torch
        if self.timings is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.timings[name] = self.timings.get(name, 0.0) + now - self._t_last
            self._t_last = now

    def _restart_clock(self):
        import time
        self._t_last = time.perf_counter()

    def host_arena(self):
        """the reads' bytes on the host (device_reads: downloaded when a host-side step needs them after all)"""
        return self.arena if self.arena is not None else self.device_reads["d_reads"][:int(self.offsets[-1])].cpu().numpy()

    def host_slot2(self):
        """[n, k] -> index of the (read, reference) pair in the both-strand batch, -1: none.  Built on the host only when something there
        reads it (the host selection, the scaffold rule): the kernels take the device copy"""
        if self._host_slot2 is None:
            self._host_slot2 = np.full((self.n, self.k), -1, dtype=np.int64)
            if self.n2:
                self._host_slot2[self.bi, self.br] = np.arange(self.n2)
        return self._host_slot2

    def reduce_stats(self):
        """the integer statistics summed over the ranks (each rank counted its own shard)"""
        import torch
        if self.reduce_across_ranks:
            keys = sorted(self.stats)
            v = C.all_reduce(torch.tensor([self.stats[q] for q in keys], dtype=torch.int64, device=self.dev)).cpu().numpy()
            for q, x in zip(keys, v):
                self.stats[q] = int(x)

    def exchange_aligned(self, aligned_local):
        """sharded run: which unique reads of the WHOLE list aligned -- every rank contributes its part (one byte per read)"""
        import torch
        t = torch.zeros(max(self.n_global, 1), dtype=torch.uint8, device=self.dev)
        if self.n:
            idx_t = (torch.arange(self.shard_idx.start, self.shard_idx.stop, device=self.dev) if isinstance(self.shard_idx, slice)
                     else torch.from_numpy(self.shard_idx).to(self.dev))
            t[idx_t] = torch.from_numpy(np.ascontiguousarray(aligned_local, dtype=np.uint8)).to(self.dev)
        C.all_reduce(t)
        return t.cpu().numpy()[:self.n_global] != 0

    def _select_kernel(self, **out):
        """c2_select_best_kernel over this run's records (first call: masks + statistics from the raw multiplicities; second call: the weights
        of the count pass from the merged ones)"""
        n2 = self.n2
        C.select_best_device(self.ctx, self.n, self.k, self.r1.data_ptr(), self.min_mscore, self.mode, max(self.stride, self.stride2),
                             d_records2=self.r2.data_ptr() if n2 else None, d_slot2=self.d_slot2.data_ptr() if n2 else None, stream=self.stream, **out)

    def _count_launches(self, d_out, d_w1, d_w2, flags):
        """the count kernel over batch 1 (all-references layout) and batch 2 with the given weights, into d_out"""
        C.accumulate_device(self.ctx, self.layout, self.n1, self.a1.data_ptr(), self.f1.data_ptr(), self.stride, self.r1.data_ptr(), d_out.data_ptr(),
                            d_weights=d_w1.data_ptr(), flags=flags | C.FLAG_ALL_REFS_LAYOUT, stream=self.stream,
                            d_hints=None if getattr(self, "h1", None) is None else self.h1.data_ptr())
        if self.n2 and d_w2 is not None:
            C.accumulate_device(self.ctx, self.layout, self.n2, self.a2.data_ptr(), self.f2.data_ptr(), self.stride2, self.r2.data_ptr(), d_out.data_ptr(),
                                d_weights=d_w2.data_ptr(), flags=flags, stream=self.stream)

    # ---- the run ----
    def run(self, arena, offsets, read_counts, shard, fastq_stream, device_reads):
        self.take_input(arena, offsets, read_counts, shard, fastq_stream, device_reads)
        self.prepare()
        if self.n == 0:
            return self.empty_shard()
        self.reads_to_device()
        self.start_partner_search()
        self.align_all_references()
        self.align_both_strands()
        self.select()
        if self.on_device and self.shard is None and not self.scaffold_rule and not self.want_view and not FORCE_HOST_MERGE:
            res = self.finish_on_device()
            if res is not None:
                return res
        self.merge_on_host()
        self.scaffold_hits()
        self.count()
        self.first_amplicon_view()
        self.reduce()
        return self.result(self.device_state())

    def take_input(self, arena, offsets, read_counts, shard, fastq_stream, device_reads):
        refs, ref_names, args = self.refs, self.ref_names, self.args
        if fastq_stream is not None:
            if shard is not None:
                raise ValueError("a sharded run cuts the list of ALL unique reads: it cannot start before the file is parsed")
            self.front = _stream_front(fastq_stream, self.aligner, self.ctx, self.dev, refs, ref_names, args, self.legacy, self.timings)
            arena, offsets, read_counts = self.front["arena"], self.front["offsets"], self.front["counts"]
            self._restart_clock()
        shard_of_all = isinstance(shard, str) and shard == "mine"      # (the rank's range of a list whose length is not known yet)
        if isinstance(shard, str) and not (shard_of_all and device_reads is not None):
            raise ValueError('shard is None, (lo, hi), an index array, or "mine" together with device_reads')
        if device_reads is not None:
            if fastq_stream is not None or (shard is not None and not shard_of_all and not isinstance(shard, tuple)):
                raise ValueError("device_reads takes no fastq_stream and only contiguous shards")
            if isinstance(device_reads, (str, os.PathLike, np.ndarray, _native.BgzfFile)):
                source = device_reads if isinstance(device_reads, (np.ndarray, _native.BgzfFile)) else os.fspath(device_reads)
                if shard is None:
                    # a FASTQ file, or FASTQ text in host memory: framed, de-duplicated AND aligned (batch 1) chunk by chunk under its upload (_device_front)
                    self.front = _device_front(source, self.aligner, self.ctx, self.dev, refs, ref_names, args, self.legacy, self.timings)
                    device_reads = self.front["device_reads"]
                else:
                    # sharded: every rank frames and de-duplicates the whole text on ITS device (the parent's variantCache of the reference,
                    # which all workers see) and aligns its range of the unique reads
                    from . import fastq_device
                    device_reads = fastq_device.ingest_file(source, self.ctx, self.dev, timings=self.timings)
                self._restart_clock()
            arena, offsets, read_counts = None, device_reads["offsets"], device_reads["counts"]
        if shard_of_all:                                              # (this rank's contiguous range of however many unique reads there are)
            import torch.distributed as dist
            from . import distributed as D
            shard = D.my_shard(len(read_counts), dist.get_rank(), dist.get_world_size())
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        # the whole run's unique reads (what the reverse-complement partner search looks at) and the part this process aligns
        self.g_arena, self.g_offsets, self.g_raw = arena, offsets, np.asarray(read_counts, dtype=np.int64)
        self.n_global = len(self.g_raw)
        self.shard_idx = None
        if shard is not None:
            self.reduce_across_ranks = True
            if isinstance(shard, tuple):
                lo, hi = int(shard[0]), int(shard[1])
                self.shard_idx = slice(lo, hi)
                if device_reads is not None:                          # (the reads are on the device: the shard is a view of them)
                    device_reads = dict(device_reads, d_reads=device_reads["d_reads"][int(offsets[lo]):max(int(offsets[hi]), int(offsets[lo]) + 1)],
                                        d_off=device_reads["d_off"][lo:hi + 1] - device_reads["d_off"][lo],
                                        d_counts=None if device_reads.get("d_counts") is None else device_reads["d_counts"][lo:hi].contiguous())
                else:
                    arena = np.asarray(arena)[int(offsets[lo]):int(offsets[hi])]
                offsets = (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64)
            else:
                self.shard_idx = np.asarray(shard, dtype=np.int64)
                ln_ = (offsets[1:] - offsets[:-1]).astype(np.int64)[self.shard_idx]
                new_off = np.zeros(len(self.shard_idx) + 1, dtype=np.uint64)
                new_off[1:] = np.cumsum(ln_)
                src = np.repeat(offsets[:-1].astype(np.int64)[self.shard_idx] - new_off[:-1].astype(np.int64), ln_) + np.arange(int(new_off[-1]), dtype=np.int64)
                arena = np.asarray(arena)[src] if len(src) else np.zeros(0, dtype=np.uint8)
                offsets = new_off
            read_counts = self.g_raw[self.shard_idx]
        self.arena, self.offsets, self.read_counts, self.shard, self.device_reads = arena, offsets, read_counts, shard, device_reads

    def prepare(self):
        import torch
        self.n = len(self.read_counts)
        self.n1 = self.n * self.k
        self.lens = (self.offsets[1:] - self.offsets[:-1]).astype(np.int64)
        self.max_lj = int(self.lens.max()) if self.n else 1
        if self.reduce_across_ranks:
            # every rank must build the SAME tensor (the histogram length depends on the longest read): agree on it first
            self.max_lj = C.all_reduce_max(self.max_lj, self.dev)
        self.layout = C.CountLayout(self.k, max(self.L), self.max_lj)
        self.d_counts = torch.zeros(self.layout.shape(), dtype=torch.int64, device=self.dev)
        self.stats = dict(N_TOT_READS=int(np.asarray(self.read_counts, dtype=np.int64).sum()), N_CACHED_ALN=0, N_CACHED_NOTALN=0, N_COMPUTED_ALN=0,
                          N_COMPUTED_NOTALN=0, N_GLOBAL_SUBS=0, N_SUBS_OUTSIDE_WINDOW=0, N_MODS_IN_WINDOW=0, N_MODS_OUTSIDE_WINDOW=0,
                          N_READS_IRREGULAR_ENDS=0, N_TOTAL=0, N_AMBIGUOUS=0)
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream

    def empty_shard(self):
        """an empty shard still takes part in every collective of the other ranks (same tensors, zeros, same order) and gets the same result
        as they do"""
        import torch
        k, dev, shape = self.k, self.dev, tuple(self.layout.shape())
        if self.reduce_across_ranks:
            if self.shard is not None:
                self.exchange_aligned(None)
            C.all_reduce(self.d_counts)
            if self.want_view:
                self.d_view = C.all_reduce(torch.zeros((k + (1 if self.scaffold_rule else 0),) + shape, dtype=torch.int64, device=dev))
            if self.scaffold_rule:
                self.d_scaffold = C.all_reduce(torch.zeros(shape, dtype=torch.int64, device=dev))
            self.reduce_stats()
        elif self.scaffold_rule:
            self.d_scaffold = torch.zeros(shape, dtype=torch.int64, device=dev)
            self.d_view = torch.zeros((k + 1,) + shape, dtype=torch.int64, device=dev) if self.want_view else None
        elif self.want_view:
            self.d_view = torch.zeros((k,) + shape, dtype=torch.int64, device=dev)
        return self.result(None)

    def reads_to_device(self):
        import torch
        if self.device_reads is not None:
            self.d_reads, self.d_off = self.device_reads["d_reads"], self.device_reads["d_off"]
        else:
            self.arena = np.ascontiguousarray(self.arena, dtype=np.uint8)
        if self.front is None and self.device_reads is None:
            if not self.arena.flags.writeable:
                self.arena = self.arena.copy()                        # torch.from_numpy wants a writable array
            # the reads go to the device now: the copy is in flight while the host tests the seeds
            self.d_reads = torch.from_numpy(self.arena if self.arena.size else np.zeros(1, dtype=np.uint8)).to(self.dev, non_blocking=True)
            self.d_off = torch.from_numpy(self.offsets.astype(np.int64)).to(self.dev, non_blocking=True)
        elif self.front is not None and self.device_reads is None:
            self.d_reads = self.d_off = None                          # (the streamed front sent them chunk by chunk; batch 2 gathers from the host arena)
        self.lap("setup")

    def start_partner_search(self):
        """which read is the reverse complement of which (for the count merge of :3970-3975) does not depend on the alignments: reads of one
        length that are already in HBM are searched there (milliseconds), anything else by a host thread while the device aligns"""
        import threading
        self.partners = partners = {}
        device_reads, front, shard = self.device_reads, self.front, self.shard

        def find():
            try:
                if partners.get('device') is not None:
                    return                                            # (enqueued on the device, below; fetched at the merge)
                if device_reads is not None and device_reads.get("rc_partner") is not None:
                    partners['index'] = device_reads["rc_partner"]     # (looked up in the device ingest's table, over ALL unique reads)
                elif front is not None and front.get("rc_partners") is not None:
                    partners['index'] = front["rc_partners"]()        # (from the table the streamed ingest built: no second hash of every read)
                else:
                    partners['index'] = (_native.rc_partners(self.host_arena(), self.offsets) if shard is None else
                                         _native.rc_partners(np.ascontiguousarray(self.g_arena, dtype=np.uint8), self.g_offsets))
            except BaseException as e:                               # re-raised by the main thread at the join
                partners['error'] = e
        if (shard is None and self.n >= RC_PARTNERS_ON_DEVICE_MIN and int(self.lens.min()) == self.max_lj
                and not (device_reads is not None and device_reads.get("rc_partner") is not None)):
            d_all = front["d_reads_all"] if front is not None else self.d_reads
            if d_all is not None and d_all.numel() >= self.n * self.max_lj:
                partners['device'] = rc_partners_device(d_all[:self.n * self.max_lj].view(self.n, self.max_lj))
        self.partner_thread = threading.Thread(target=find, name="c2-rc-partners")
        self.threads.append(self.partner_thread)
        self.partner_thread.start()

    def join_partner_search(self):
        self.partner_thread.join()
        if 'error' in self.partners:
            raise self.partners['error']

    def align_all_references(self):
        """the seed test that picks the strand(s) of every (read, reference) alignment (:656-687) and batch 1: every read against every
        reference, on the strand the seeds ask for (forward when they ask for both)"""
        import torch
        n, k, dev = self.n, self.k, self.dev
        if self.front is not None:
            # seed test and batch 1 ran chunk by chunk while the file was parsed / uploaded
            f = self.front
            self.plan, self.stride, self.a1, self.f1, self.r1 = f["plan"], f["stride"], f["a1"], f["f1"], f["r1"]
            self.h1 = None                                           # (batches enqueued under the upload: no hint words)
        else:
            # on the device, over the reads that were just sent there (FORCE_HOST_STRAND_PLAN: the host's threaded c2_strand_plan instead --
            # the tests compare the two)
            if FORCE_HOST_STRAND_PLAN:
                self.plan = strand_plans(self.arena, self.offsets, self.refs, self.ref_names, self.args)
            else:
                d_plan = torch.empty(n * k, dtype=torch.uint8, device=dev)
                C.strand_plan_device(self.ctx, n, self.d_reads.data_ptr(), self.d_off.data_ptr(), self.max_lj, self.refs, self.ref_names,
                                     self.args.aln_seed_count, self.args.aln_seed_min, d_plan.data_ptr(), stream=self.stream)
                self.plan = to_host(d_plan).reshape(n, k)
            self.lap("strand_plan")
            self.stride = self.aligner.stride_for(self.max_lj)
            d_str1 = to_device((self.plan == 1).astype(np.uint8).reshape(-1), dev)
            self.a1 = torch.empty((self.n1, self.stride), dtype=torch.uint8, device=dev)
            self.f1 = torch.empty((self.n1, self.stride), dtype=torch.uint8, device=dev)
            self.r1 = torch.empty((self.n1, 32), dtype=torch.uint8, device=dev)
            # one amplicon: the hint words of the reads the partition finishes itself (c2_batch.diag_hints) -- the count pass takes those from the word alone
            self.h1 = torch.empty(self.n1 * 4, dtype=torch.int32, device=dev) if (k == 1 and self.n1) else None
            self.aligner.align_device(n, self.d_reads.data_ptr(), self.d_off.data_ptr(), self.a1.data_ptr(), self.f1.data_ptr(), self.r1.data_ptr(),
                                      self.stride, self.max_lj, d_strands=d_str1.data_ptr(), all_refs=True, stream=self.stream, legacy=self.legacy,
                                      min_read_len=int(self.lens.min()) if n else 0, d_hints=None if self.h1 is None else self.h1.data_ptr())
        self.lap("h2d_align")

    def align_both_strands(self):
        """batch 2: the (read, reference) pairs aligned on both strands -- their reverse-complement alignments"""
        import torch
        dev = self.dev
        if self.front is not None and self.front.get("both") is not None:
            self.bi, self.br = self.front["both"][:, 0].copy(), self.front["both"][:, 1].copy()   # (found on the device, row-major like np.nonzero)
        else:
            self.bi, self.br = np.nonzero(self.plan == 2)
        self.n2 = n2 = len(self.bi)
        self._host_slot2 = None
        self.lap("both_strand_list")
        self.r2 = self.a2 = self.f2 = None
        self.stride2 = self.stride
        if n2:
            # gather the bytes of those reads (a read that is undecided for several references is repeated)
            max_lj2 = int(self.lens[self.bi].max())
            self.stride2 = self.aligner.stride_for(max_lj2)
            if self.device_reads is not None:
                from . import fastq_device
                d_reads2, d_off2, _ = fastq_device.gather_reads_device(self.ctx, self.d_reads, self.d_off, self.bi, dev, self.stream)
            else:
                arena2, off2 = _native.gather_reads(self.arena, self.offsets, self.bi)
                d_reads2 = torch.from_numpy(arena2 if arena2.size else np.zeros(1, dtype=np.uint8)).to(dev)
                d_off2 = torch.from_numpy(off2.astype(np.int64)).to(dev)
            d_rid2 = torch.from_numpy(self.br.astype(np.int16)).to(dev)
            d_str2 = torch.ones(n2, dtype=torch.uint8, device=dev)
            self.a2 = torch.empty((n2, self.stride2), dtype=torch.uint8, device=dev)
            self.f2 = torch.empty((n2, self.stride2), dtype=torch.uint8, device=dev)
            self.r2 = torch.empty((n2, 32), dtype=torch.uint8, device=dev)
            self.aligner.align_device(n2, d_reads2.data_ptr(), d_off2.data_ptr(), self.a2.data_ptr(), self.f2.data_ptr(), self.r2.data_ptr(), self.stride2,
                                      max_lj2, d_ref_ids=d_rid2.data_ptr(), d_strands=d_str2.data_ptr(), stream=self.stream, legacy=self.legacy,
                                      min_read_len=int(self.lens[self.bi].min()))
        self.lap("both_strand_pairs")
        self.d_slot2 = None
        if n2:
            self.d_slot2 = torch.full((self.n * self.k,), -1, dtype=torch.int32, device=dev)
            self.d_slot2[to_device(np.asarray(self.bi, dtype=np.int64) * self.k + np.asarray(self.br, dtype=np.int64), dev)] = \
                torch.arange(n2, dtype=torch.int32, device=dev)

    def select(self):
        """strand and reference choice (:683, :697-707), ambiguity, aln_stats: on the device.  Host selection (the same comparisons on Python
        floats) remains for what the kernel's exact integer scores do not cover (alignments of SELECT_MAX_ALN_LEN columns or more)."""
        import torch
        n, k, dev, stats = self.n, self.k, self.dev, self.stats
        self.raw = np.asarray(self.read_counts, dtype=np.int64)
        if self.raw.size and self.raw.max() > 0x7FFFFFFF:
            raise OverflowError("a read multiplicity exceeds 2^31 - 1")
        self.min_scores = [self.refs[name]['min_aln_score'] for name in self.ref_names]
        self.mode = C.select_mode(self.args)
        self.on_device = max(self.stride, self.stride2) <= C.SELECT_MAX_ALN_LEN and not FORCE_HOST_SELECTION
        self.member = self.use2 = self.aligned = None                  # (host copies of the masks: only when a host-side stage needs them)
        if not self.on_device:
            self.member, self.use2, self.aligned = _select_on_host(self.r1, self.r2, n, k, self.n2, self.bi if self.n2 else None, self.br if self.n2 else None,
                                                                   self.host_slot2(), self.min_scores, self.raw, stats)
            self.lap("selection_and_stats")
            return
        words = (k + 63) // 64                                        # 64-bit words of a read's masks (bit r % 64 of word r / 64: reference r)
        self.d_member = torch.zeros((n, words), dtype=torch.int64, device=dev)
        self.d_use2 = torch.zeros((n, words), dtype=torch.int64, device=dev)
        self.d_flags = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_stats = torch.zeros(len(C.SELECT_STATS), dtype=torch.int64, device=dev)
        dr = self.device_reads
        self.d_raw = (dr["d_counts"] if dr is not None and dr.get("d_counts") is not None else to_device(self.raw.astype(np.uint32).view(np.int32), dev))
        self.min_mscore = C.min_mscore_table(self.min_scores)
        self.lap("selection_inputs")
        self._select_kernel(d_raw_counts=self.d_raw.data_ptr(), d_member=self.d_member.data_ptr(), d_use2=self.d_use2.data_ptr(),
                            d_flags=self.d_flags.data_ptr(), d_stats=d_stats.data_ptr())
        st = dict(zip(C.SELECT_STATS, d_stats.cpu().numpy().tolist()))
        self.lap("selection_kernel")
        if st["n_bad_status"]:
            if int(st["a_bad_status"]) & _native.STATUS_RC_CHAR:
                raise KeyError("reverse_complement: a read has a character outside ACGTN_-")
            raise Exception('global_align: undefined alignment (status %d)' % int(st["a_bad_status"]))
        for q in ('N_COMPUTED_ALN', 'N_COMPUTED_NOTALN', 'N_CACHED_ALN', 'N_CACHED_NOTALN', 'N_GLOBAL_SUBS', 'N_SUBS_OUTSIDE_WINDOW',
                  'N_MODS_IN_WINDOW', 'N_MODS_OUTSIDE_WINDOW', 'N_READS_IRREGULAR_ENDS'):
            stats[q] = int(st[q])

    def masks_to_host(self):
        """the kernel's 64-bit masks (bit r: reference r) taken apart on the device: n x k bytes cross the link, not 16 n"""
        import torch
        ref_ix = torch.arange(self.k, dtype=torch.int64, device=self.dev)
        word_of, bit_of = ref_ix >> 6, (ref_ix & 63)[None, :]
        self.member = to_host(((self.d_member[:, word_of] >> bit_of) & 1).to(torch.uint8)).view(bool)
        self.use2 = to_host(((self.d_use2[:, word_of] >> bit_of) & 1).to(torch.uint8)).view(bool)
        self.aligned = to_host(self.d_flags & 1).view(bool)

    def finish_on_device(self):
        """The rest of the usual run without the host in it: the reverse-complement count transfer (:3970-3975), N_TOTAL / N_AMBIGUOUS, the
        weights and the count pass on the device; what alleles() reads comes to the host when it is asked for.  The transfer is sequential in
        the reference; over partner PAIRS (i <-> j, or i its own partner) it is independent work: when both reads aligned and have copies the
        earlier one takes the later one's (a palindrome doubles).  That needs the partner relation to be symmetric -- lower-case reads break
        it (their reverse complement is upper case): then (-> None) the host applies the loop."""
        import torch
        n, dev, stats, args = self.n, self.dev, self.stats, self.args
        self.join_partner_search()
        partners, dr = self.partners, self.device_reads
        if partners.get('device') is not None:
            d_partner = partners['device'].partner_tensor()
        elif dr is not None and dr.get("d_rc_partner") is not None:
            d_partner = dr["d_rc_partner"]
        else:
            d_partner = to_device(np.ascontiguousarray(partners['index'], dtype=np.int64), dev)
        if d_partner is None:
            return None
        ix = torch.arange(n, dtype=torch.int64, device=dev)
        has = d_partner >= 0
        pc = d_partner.clamp(min=0)
        asym = (has & (d_partner[pc] != ix)).sum()
        c0 = self.d_raw.to(torch.int64) & 0xffffffff
        d_al = (self.d_flags & 1) != 0
        ok = d_al & (c0 > 0)
        takes = has & (d_partner > ix) & ok & ok[pc]                  # the earlier read of an aligned pair with copies
        gives = has & (d_partner < ix) & takes[pc]                    # ... and its partner
        own = has & (d_partner == ix) & ok
        c1 = torch.where(gives, torch.zeros_like(c0), c0 + torch.where(takes, c0[pc], torch.zeros_like(c0)) + torch.where(own, c0, torch.zeros_like(c0)))
        d_amb = (self.d_flags & 2) != 0
        sums = to_host(torch.stack([asym, (c1 * d_al).sum(), (c1 * d_amb).sum(), c1.max() if n else asym * 0]))
        if int(sums[0]) != 0:
            return None
        if int(sums[3]) > 0x7FFFFFFF:
            raise OverflowError("a read multiplicity exceeds 2^31 - 1")
        stats['N_TOTAL'] = int(sums[1])
        if not args.assign_ambiguous_alignments_to_first_reference and not args.expand_ambiguous_alignments:
            stats['N_AMBIGUOUS'] = int(sums[2])
        self.lap("rc_merge_weights")
        d_cnt = c1.to(torch.int32)
        d_w1 = torch.zeros(self.n1, dtype=torch.int32, device=dev)
        d_w2 = torch.zeros(self.n2, dtype=torch.int32, device=dev) if self.n2 else None
        self._select_kernel(d_counts=d_cnt.data_ptr(), d_weights=d_w1.data_ptr(), d_weights2=d_w2.data_ptr() if self.n2 else None)
        self._count_launches(self.d_counts, d_w1, d_w2, self.flags)
        if self.reduce_across_ranks:
            C.all_reduce(self.d_counts)
            self.reduce_stats()
        torch.cuda.synchronize(dev)
        return self.result(self.device_state(d_cnt=d_cnt))

    def merge_on_host(self):
        """aggregation weights (:3964-4000): the reference's sequential reverse-complement count transfer -- over the WHOLE variantCache order
        with every rank's `aligned` flags when the run is sharded --, ambiguous reads, which references a read counts for"""
        args, stats = self.args, self.stats
        if self.member is None:
            self.masks_to_host()
        member, aligned = self.member, self.aligned
        n_best = member.sum(axis=1)
        self.lap("selection_and_stats")
        cnt = np.ascontiguousarray(self.raw.copy())
        self.join_partner_search()
        partners = self.partners
        if partners.get('device') is not None:
            partners['index'] = partners['device'].result()
            if partners['index'] is None:                            # (a hash collision among the reads: the host search decides)
                partners['index'] = _native.rc_partners(self.host_arena(), self.offsets)
        if self.shard is None:
            _native.merge_counts_with_partners(aligned, partners['index'], cnt)
        else:
            # a read whose reverse complement was aligned by another rank gives its copies to (or takes them from) that read exactly as in one process
            g_cnt = np.ascontiguousarray(self.g_raw.copy())
            _native.merge_counts_with_partners(self.exchange_aligned(aligned), partners['index'], g_cnt)
            cnt = np.ascontiguousarray(g_cnt[self.shard_idx])
        # (whole-array arithmetic instead of boolean fancy indexing: these are passes over millions of unique reads)
        stats['N_TOTAL'] = int(np.dot(cnt, aligned.astype(np.int64)))
        counted = member.copy()
        ambiguous = aligned & (n_best > 1)
        any_ambiguous = bool(ambiguous.any())
        if args.assign_ambiguous_alignments_to_first_reference:
            if any_ambiguous:
                first = np.argmax(member, axis=1)
                counted[ambiguous, :] = False
                counted[ambiguous, first[ambiguous]] = True
        elif not args.expand_ambiguous_alignments:
            if any_ambiguous:
                counted &= ~ambiguous[:, None]
            stats['N_AMBIGUOUS'] = int(np.dot(cnt, ambiguous.astype(np.int64))) if any_ambiguous else 0
        counted &= aligned[:, None]
        self.cnt, self.counted, self.ambiguous = cnt, counted, ambiguous

    def scaffold_hits(self):
        """prime-editing scaffold rule (:786-796): a read whose best amplicons include 'Prime-edited' and whose alignment against it shows the
        scaffold's first bases right after the extension is counted for 'Scaffold-incorporated' ONLY (ambiguous or not), with that alignment.
        The aligned strings of the candidate reads come to the host for the substring test."""
        import torch
        n, k, pe, dev = self.n, self.k, self.pe, self.dev
        self.scaffold_hit = scaffold_hit = np.zeros(n, dtype=bool)
        if self.scaffold_rule:
            cand = np.nonzero(self.aligned & self.member[:, pe])[0]
            if len(cand):
                idx0, dna = int(self.pe_scaffold_dna_info[0]) - 1, self.pe_scaffold_dna_info[1]
                in2 = self.use2[cand, pe]
                pairs = [None] * len(cand)

                def pull(a, f, rows, where, lens_):
                    ah, fh = a.index_select(0, rows).cpu().numpy(), f.index_select(0, rows).cpu().numpy()
                    for q, j in enumerate(where):
                        pairs[j] = (ah[q, :int(lens_[q])].tobytes().decode(), fh[q, :int(lens_[q])].tobytes().decode())
                w_1 = np.nonzero(~in2)[0]
                if len(w_1):
                    t_ = torch.from_numpy(cand[w_1] * k + pe).to(dev)
                    pull(self.a1, self.f1, t_, w_1, self.r1.index_select(0, t_).cpu().numpy().view(_native.REC_DTYPE).reshape(-1)["aln_len"])
                w_2 = np.nonzero(in2)[0]
                if len(w_2):
                    sl = torch.from_numpy(self.host_slot2()[cand[w_2], pe]).to(dev)
                    pull(self.a2, self.f2, sl, w_2, self.r2.index_select(0, sl).cpu().numpy().view(_native.REC_DTYPE).reshape(-1)["aln_len"])
                for j, (s_read, s_ref) in enumerate(pairs):
                    seen, col = -1, -1
                    for c_, ch in enumerate(s_ref):                # ref_positions.index(idx0): the column of reference base idx0
                        if ch != '-':
                            seen += 1
                            if seen == idx0:
                                col = c_
                                break
                    if col < 0:
                        raise ValueError("%d is not in list" % idx0)
                    if s_read[col + 1:col + 1 + len(dna)] == dna:
                        scaffold_hit[cand[j]] = True
            self.counted[scaffold_hit, :] = False
            if not self.args.assign_ambiguous_alignments_to_first_reference and not self.args.expand_ambiguous_alignments:
                self.stats['N_AMBIGUOUS'] = int(self.cnt[self.ambiguous & ~scaffold_hit].sum())
        if self.cnt.max() > 0x7FFFFFFF:
            raise OverflowError("a read multiplicity exceeds 2^31 - 1")      # (the count kernel's weights are int32)
        self.lap("rc_merge_weights")

    def count(self):
        """the weight of every alignment and the count launches; with the scaffold rule one more pair of launches for 'Scaffold-incorporated'"""
        import torch
        n, k, pe, dev, cnt, n2 = self.n, self.k, self.pe, self.dev, self.cnt, self.n2
        scaffold_hit, use2 = self.scaffold_hit, self.use2
        d_w2 = None
        if self.on_device:
            # the kernel again, now with the merged multiplicities (a read the scaffold rule took away counts for no amplicon here)
            d_cnt = to_device((np.where(scaffold_hit, 0, cnt) if self.scaffold_rule else cnt).astype(np.uint32).view(np.int32), dev)
            d_w1 = torch.zeros(self.n1, dtype=torch.int32, device=dev)
            if n2:
                d_w2 = torch.zeros(n2, dtype=torch.int32, device=dev)
            self._select_kernel(d_counts=d_cnt.data_ptr(), d_weights=d_w1.data_ptr(), d_weights2=d_w2.data_ptr() if n2 else None)
        else:
            w1 = np.where(self.counted & ~use2, cnt[:, None], 0).astype(np.uint32).reshape(-1)
            d_w1 = torch.from_numpy(w1.view(np.int32)).to(dev)
            if n2:
                w2 = np.where(self.counted[self.bi, self.br] & use2[self.bi, self.br], cnt[self.bi], 0).astype(np.uint32)
                d_w2 = torch.from_numpy(w2.view(np.int32)).to(dev)
        self._count_launches(self.d_counts, d_w1, d_w2, self.flags)
        if self.scaffold_rule:
            self.d_scaffold = torch.zeros(self.layout.shape(), dtype=torch.int64, device=dev)     # row `pe` = the 'Scaffold-incorporated' amplicon
            ws = np.zeros((n, k), dtype=np.uint32)
            ws[:, pe] = np.where(scaffold_hit & ~use2[:, pe], cnt, 0)
            d_ws = torch.from_numpy(ws.reshape(-1).view(np.int32)).to(dev)
            d_ws2 = None
            if n2:
                ws2 = np.where((self.br == pe) & scaffold_hit[self.bi] & use2[self.bi, self.br], cnt[self.bi], 0).astype(np.uint32)
                d_ws2 = torch.from_numpy(ws2.view(np.int32)).to(dev)
            self._count_launches(self.d_scaffold, d_ws, d_ws2, self.flags)
            torch.cuda.synchronize(dev)

    def first_amplicon_view(self):
        """Every amplicon's reads in the coordinates of the FIRST amplicon (:4195-4270; runs with an expected HDR amplicon or a prime-editing
        extension).  The alignment of every read against the first amplicon is already on the device (ref_aln_details[0]); the reference
        classifies it again and adds its all_* positions and bases into arrays of the amplicon the read is counted for.  Here: one more count
        launch per other amplicon r over those alignments, weighted with the multiplicities of the reads counted for r; row 0 of that launch's
        tensor is amplicon r's view.  No ignore_* / discard flags: the reference's loop has none."""
        import torch
        if not self.want_view:
            return
        n, k, dev, cnt, n2, use2 = self.n, self.k, self.dev, self.cnt, self.n2, self.use2
        rows = k + (1 if self.scaffold_rule else 0)
        vflags = C.FLAG_LEGACY_CLASSIFIER if self.legacy else 0
        self.d_view = torch.zeros((rows,) + tuple(self.layout.shape()), dtype=torch.int64, device=dev)
        for r in range(1, rows):
            for_r = self.counted[:, r] if r < k else self.scaffold_hit   # row k: the reads counted for 'Scaffold-incorporated'
            wv = np.zeros((n, k), dtype=np.uint32)
            wv[:, 0] = np.where(for_r & ~use2[:, 0], cnt, 0)
            d_wv = torch.from_numpy(wv.reshape(-1).view(np.int32)).to(dev)
            d_wv2 = None
            if n2:
                wv2 = np.where((self.br == 0) & for_r[self.bi] & use2[self.bi, self.br], cnt[self.bi], 0).astype(np.uint32)
                if wv2.any():
                    d_wv2 = torch.from_numpy(wv2.view(np.int32)).to(dev)
            self._count_launches(self.d_view[r], d_wv, d_wv2, vflags)
            torch.cuda.synchronize(dev)                              # the weight tensors of this round are done with

    def reduce(self):
        import torch
        if self.reduce_across_ranks:
            C.all_reduce(self.d_counts)
            if self.d_view is not None:
                C.all_reduce(self.d_view)
            if self.d_scaffold is not None:
                C.all_reduce(self.d_scaffold)
            self.reduce_stats()
        torch.cuda.synchronize(self.dev)

    def device_state(self, d_cnt=None):
        """what the allele table is built from (QuantResult.allele_table): the selection's masks as the kernel left them (or the host selection's,
        packed the same way), the merged multiplicities of this rank's reads, the scaffold rule's hits"""
        dev = self.dev
        if self.on_device:
            t_member, t_use2, t_flags = self.d_member, self.d_use2, self.d_flags
        else:
            t_member, t_use2 = to_device(_pack_masks(self.member), dev), to_device(_pack_masks(self.use2), dev)
            t_flags = to_device(self.aligned.astype(np.uint8), dev)
        if d_cnt is None:
            d_cnt = to_device(self.cnt.astype(np.uint32).view(np.int32), dev)
        hit = to_device(self.scaffold_hit.astype(np.uint8), dev) if (self.scaffold_rule and self.scaffold_hit is not None) else None
        return dict(ctx=self.ctx, stream=self.stream, args=self.args, ref_names=list(self.ref_names), n=self.n, mode=self.mode, flags=self.flags & 15,
                    a1=self.a1, f1=self.f1, r1=self.r1, stride=self.stride, a2=self.a2, f2=self.f2, r2=self.r2, stride2=self.stride2, d_slot2=self.d_slot2,
                    d_member=t_member, d_use2=t_use2, d_flags=t_flags, d_cnt=d_cnt, d_scaffold_hit=hit, scaffold_ref=self.pe)

    def result(self, state):
        """the tensors (already all-reduced when sharded) -> QuantResult; the same for a rank whose shard is empty"""
        layout, L, ref_names = self.layout, self.L, self.ref_names
        host = self.d_counts.cpu().numpy()
        self.lap("count_kernels")
        per_ref = {name: layout.unpack(host, r, L[r]) for r, name in enumerate(ref_names)}
        out_names = list(ref_names)
        if self.scaffold_rule:
            per_ref['Scaffold-incorporated'] = layout.unpack(self.d_scaffold.cpu().numpy(), self.pe, L[self.pe])
            out_names.append('Scaffold-incorporated')
        first_ref_view = None
        if self.d_view is not None:
            view_keys = (["all_insertion_count_vectors", "all_insertion_left_count_vectors", "all_deletion_count_vectors",
                          "all_substitution_count_vectors"] + ["all_base_count_vectors_" + x for x in "ACGTN-"])
            host_view = self.d_view.cpu().numpy()
            first_ref_view = {ref_names[0]: {kk: per_ref[ref_names[0]][kk] for kk in view_keys}}
            for r in range(1, len(out_names)):
                u = layout.unpack(host_view[r], 0, L[0])
                first_ref_view[out_names[r]] = {kk: u[kk] for kk in view_keys}
            for v in first_ref_view.values():
                v["all_indelsub_count_vectors"] = (v["all_insertion_count_vectors"] + v["all_deletion_count_vectors"]
                                                   + v["all_substitution_count_vectors"])
        self.lap("unpack")
        res = QuantResult(per_ref, self.stats, layout, self.d_counts, state, first_ref_view=first_ref_view)
        if self.device_reads is not None:
            res.device_ingest = {q: self.device_reads[q] for q in ("n_reads", "nonempty_lines", "n_unique", "n_empty_records") if q in self.device_reads}
        return res


def quantify_fastq(path, refs, ref_names, aln_matrix, args, ctx=None, device=0, timings=None, pe_scaffold_dna_info=None,
                   shard_across_ranks=False, stream=True):
    """FASTQ file -> QuantResult.  Empty sequences (blank line / truncated record) are dropped, as in variants.read_fastq_unique
    (the reference's aligner indexes seq[-1] of an empty string: undefined there), and N_TOT_READS counts the records that are left.
    shard_across_ranks (torch.distributed initialised, one process per GPU): every rank ingests ITS byte range of the text, the ranks
    reconcile their unique reads into the run's list -- the parent's variantCache of the reference, which all its workers see --, every
    rank aligns its contiguous range of that list (get_variant_cache_equal_boundaries, CRISPRessoCORE.py:1172-1195) and the count tensors
    are all-reduced; the result is the single-process one on every rank (quantify_unique, `shard`).
    stream (default; one process): the file is parsed in chunks on a host thread while the device already aligns the unique reads of
    the chunks before (_stream_front) -- same result as the one-batch flow (stream=False), which sharded runs keep."""
    import time
    t0 = time.perf_counter()
    # --min_single_bp_quality / --min_average_read_quality / --min_bp_quality_or_N: the reference filters the file first
    # (CRISPRessoCORE.py:3696-3717); here the filter runs inside the ingest
    flt = [int(getattr(args, k_, 0) or 0) for k_ in ('min_single_bp_quality', 'min_average_read_quality', 'min_bp_quality_or_N')]
    ingest_stats = {}
    sharded = False
    if shard_across_ranks:
        import torch.distributed as dist
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    run = lambda **kw: quantify_unique(None, None, None, refs, ref_names, aln_matrix, args, ctx=ctx, device=device, timings=timings,
                                       pe_scaffold_dna_info=pe_scaffold_dna_info, **kw)

    def reads_as_the_reference_counts_them(res, n_reads=None):
        # non-empty lines / 4 of the input and of the text that was parsed (get_n_reads_fastq) -- these differ from the number of records
        # only for files with blank lines or a truncated tail
        res.stats['N_READS_INPUT'] = ingest_stats['N_READS_INPUT'] if n_reads is None else ingest_stats.get('N_READS_INPUT', int(n_reads))
        res.stats['N_READS_AFTER_PREPROCESSING'] = (ingest_stats['N_READS_AFTER_PREPROCESSING'] if n_reads is None
                                                    else ingest_stats.get('N_READS_AFTER_PREPROCESSING', int(n_reads)))
        return res

    def note(why):
        if timings is not None:
            timings["host_parser_because"] = why
    if (stream or sharded) and not FORCE_HOST_STRAND_PLAN:
        # The text framed and de-duplicated on the device (fastq_device): a plain file as it lies on disk, BGZF members as they are inflated,
        # other compressed or quality-filtered input after the host has inflated / filtered it into memory (fastq_device.IngestSource says
        # which).  One process: the whole text under its upload, batch 1 running under it (_device_front).  Sharded: every rank uploads,
        # frames and de-duplicates ITS byte range, the ranks all-gather their unique reads and reconcile them into the run's list
        # (fastq_device.ingest_shard), each aligns its range of that list.  Anything the kernels cannot take (carriage returns, small files,
        # ...; sharded: on ANY rank -- the vote is a collective): the host parser.
        import torch
        from . import fastq_device
        S = None
        try:
            S = fastq_device.IngestSource(path, flt)
        except _native.NativeError:
            if not sharded:
                raise                                                 # (sharded: the vote below must still happen; the host parser raises it again)
        try:
            source = S.source if S is not None else None
            if source is not None and S.filtered_lines_input is not None:
                ingest_stats["N_READS_INPUT"] = int(float(S.filtered_lines_input) / 4.0)
            if sharded:
                dev_ = torch.device("cuda", device)
                ing = None
                if C.all_reduce_max(1 if source is None else 0, dev_) == 0:
                    try:
                        ing = fastq_device.ingest_shard(source, ctx or _native.default_context(), dev_, timings=timings)
                    except fastq_device.DeviceIngestUnavailable as e:
                        note(str(e))
                if ing is not None:
                    res = run(device_reads=ing, shard="mine")
                    _native._line_stats(ingest_stats, ing["nonempty_lines"])
                    res.ingest_route = "device, sharded by byte range"
                    res.shard_ingest = {q: ing[q] for q in ("shard_bytes", "text_bytes", "shard_records", "shard_unique", "gathered_unique", "gathered_bytes")}
                    return reads_as_the_reference_counts_them(res)
                ingest_stats.pop("N_READS_INPUT", None)               # (sharded runs keep the one-batch host flow below)
            else:
                why_not = S.why_not
                if source is not None:
                    try:
                        res = run(device_reads=source)
                        _native._line_stats(ingest_stats, res.device_ingest["nonempty_lines"])
                        res.ingest_route = S.route
                        return reads_as_the_reference_counts_them(res)
                    except fastq_device.DeviceIngestUnavailable as e:
                        why_not = str(e)                              # (e.g. carriage returns: the host parser takes the file)
                        ingest_stats.pop("N_READS_INPUT", None)
                note(why_not)
                fq = S.host_stream()                                  # parsed chunk by chunk on a host thread while the device aligns the chunks before
                if fq is not None:
                    res = run(fastq_stream=fq)
                    fq.line_stats(ingest_stats)
                    n_reads = fq.n_reads
                    t_free = time.perf_counter()
                    S.close()
                    if timings is not None:
                        timings["free_ingest"] = time.perf_counter() - t_free
                    return reads_as_the_reference_counts_them(res, n_reads)
        finally:
            if S is not None:
                S.close()
    with _native.FastqUnique(path, *flt, stats=ingest_stats) as fq:      # views of the native arena: nothing is copied on the host
        arena, offsets, counts, n_reads = fq.arena, fq.offsets, fq.counts, fq.n_reads
        if timings is not None:
            timings["ingest_dedup"] = time.perf_counter() - t0
        lens = offsets[1:] - offsets[:-1]
        if len(counts) and (lens == 0).any():
            keep = np.nonzero(lens > 0)[0]                          # at most one empty key
            new_off = np.zeros(len(keep) + 1, dtype=np.uint64)
            new_off[1:] = np.cumsum(lens[keep])
            arena = np.concatenate([arena[int(offsets[i]):int(offsets[i + 1])] for i in keep]) if len(keep) else np.zeros(0, dtype=np.uint8)
            offsets, counts = new_off, counts[keep]
        if timings is not None:
            timings["drop_empty_key"] = time.perf_counter() - t0 - timings["ingest_dedup"]
        shard = None
        if shard_across_ranks:
            import torch.distributed as dist
            from . import distributed as D
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                shard = D.my_shard(len(counts), dist.get_rank(), dist.get_world_size())
        res = quantify_unique(arena, offsets, counts, refs, ref_names, aln_matrix, args, ctx=ctx, device=device, timings=timings,
                              pe_scaffold_dna_info=pe_scaffold_dna_info, shard=shard)
        t_free = time.perf_counter()
    if timings is not None:
        timings["free_ingest"] = time.perf_counter() - t_free
    return reads_as_the_reference_counts_them(res, n_reads)
