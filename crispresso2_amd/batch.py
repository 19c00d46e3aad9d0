"""Batch align + classify on the MI355X: the replacement for the reference's per-read hot loop
(CRISPRessoCORE.py:1957-1981 serial, :1226-1232 in the forked workers).

`BatchAligner` owns one GPU context, the score matrix and the reference amplicons of a run
(what CRISPRessoCORE.py:3236-3268 keeps in `refs[name]`), and turns lists of reads into

  * the two aligned strings of every executed (read, reference) alignment -- byte-identical to
    `CRISPResso2Align.global_align(read, ref, ...)[0:2]`,
  * the alignment score `round(100*matches/len, 3)`,
  * a fixed 32-byte record per alignment with the counts `find_indels_substitutions` and
    CRISPRessoCORE.py:726-760 derive (window / outside-window counts, irregular ends, class).

Host pointers go through `align()`; `align_device()` takes device pointers (torch tensors'
`data_ptr()`), which is what bench.py times.  No CPU fallback.
"""
import ctypes

import numpy as np

from . import _native
from ._native import REC_DTYPE


def score_from_counts(matches, aln_len):
    """The reference's score, formed exactly as CRISPResso2Align.pyx:433-434 forms it:
    Python's round() on the double 100*matches/float(len).  Vectorised through a lookup table
    over the distinct (matches, len) pairs so that every value comes from that same expression."""
    matches = np.asarray(matches, dtype=np.int64)
    aln_len = np.asarray(aln_len, dtype=np.int64)
    out = np.zeros(matches.shape, dtype=np.float64)
    ok = aln_len > 0
    key = matches[ok] * 65536 + aln_len[ok]
    uniq, inv = np.unique(key, return_inverse=True)
    vals = np.array([round(100 * int(k >> 16) / float(int(k & 65535)), 3) for k in uniq], dtype=np.float64)
    out[ok] = vals[inv]
    return out


class BatchResult:
    """Outputs of one batch, in host memory."""

    def __init__(self, aln_read, aln_ref, records, n_reads, n_refs, all_refs):
        self.aln_read = aln_read      # uint8 [n_tasks, stride]
        self.aln_ref = aln_ref        # uint8 [n_tasks, stride]
        self.records = records        # REC_DTYPE [n_tasks]
        self.n_reads = n_reads
        self.n_refs = n_refs
        self.all_refs = all_refs
        self._scores = None

    def __len__(self):
        return len(self.records)

    def strings(self, t):
        """(aligned_read, aligned_ref) of task t, as the reference returns them."""
        n = int(self.records['aln_len'][t])
        return self.aln_read[t, :n].tobytes().decode('utf-8'), self.aln_ref[t, :n].tobytes().decode('utf-8')

    @property
    def scores(self):
        if self._scores is None:
            self._scores = score_from_counts(self.records['matches'], self.records['aln_len'])
        return self._scores

    def derived(self, ignore_substitutions=False, ignore_insertions=False, ignore_deletions=False):
        """The per-read counters CRISPRessoCORE.py:734-760 derives from the classifier payload."""
        r = self.records
        ins_out = r['all_insertion_events'].astype(np.int64) - r['win_insertion_events']
        del_out = r['all_deletion_events'].astype(np.int64) - r['win_deletion_events']
        sub_out = r['all_substitutions'].astype(np.int64) - r['substitution_n']
        total = r['all_insertion_events'].astype(np.int64) + r['all_deletion_bases'] + r['all_substitutions']
        in_win = r['substitution_n'].astype(np.int64) + r['deletion_n'] + r['insertion_n']
        modified = np.zeros(len(r), dtype=bool)
        if not ignore_deletions:
            modified |= r['deletion_n'] > 0
        if not ignore_insertions:
            modified |= r['insertion_n'] > 0
        if not ignore_substitutions:
            modified |= r['substitution_n'] > 0
        return {'insertions_outside_window': ins_out, 'deletions_outside_window': del_out,
                'substitutions_outside_window': sub_out, 'total_mods': total, 'mods_in_window': in_win,
                'mods_outside_window': total - in_win, 'modified': modified}


def pack_reads(reads):
    """list of str/bytes -> (uint8 arena, uint64 offsets[n+1])"""
    bs = [r.encode('utf-8') if isinstance(r, str) else bytes(r) for r in reads]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    arena = np.frombuffer(b''.join(bs), dtype=np.uint8) if bs else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(arena), offsets


class BatchAligner:
    def __init__(self, ref_seqs, gap_incentives, include_idxs, matrix, gap_open, gap_extend, device=None, ctx=None):
        """ref_seqs: list of reference amplicon strings; gap_incentives: list of int64[len+1];
        include_idxs: list of iterables (quantification window positions, refs[name]['include_idxs'])."""
        if ctx is None:
            ctx = _native.Context(device if device is not None else 0)
        self.ctx = ctx
        self.ref_seqs = list(ref_seqs)
        self.n_refs = len(self.ref_seqs)
        self.max_ref_len = max(len(s) for s in self.ref_seqs)
        ctx.set_scoring(matrix, gap_open, gap_extend)
        ctx.set_refs(self.ref_seqs, gap_incentives, include_idxs)

    def stride_for(self, max_read_len):
        return (self.max_ref_len + int(max_read_len) + 15) // 16 * 16

    def align(self, reads, ref_ids=None, strands=None, all_refs=False, legacy=False):
        """reads: list of str, or (arena uint8, offsets uint64).  Returns a BatchResult (host memory).
        legacy: the records' window counts follow find_indels_substitutions_legacy (--use_legacy_insertion_quantification)."""
        arena, offsets = reads if isinstance(reads, tuple) else pack_reads(reads)
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        n_tasks = n * (self.n_refs if all_refs else 1)
        max_lj = int((offsets[1:] - offsets[:-1]).max()) if n else 1
        stride = self.stride_for(max_lj)
        aln_read = np.zeros((n_tasks, stride), dtype=np.uint8)
        aln_ref = np.zeros((n_tasks, stride), dtype=np.uint8)
        records = np.zeros(n_tasks, dtype=REC_DTYPE)
        if n == 0:
            return BatchResult(aln_read, aln_ref, records, 0, self.n_refs, all_refs)
        rid = None
        if ref_ids is not None and not all_refs:
            rid = np.ascontiguousarray(ref_ids, dtype=np.uint16)
            if len(rid) != n:
                raise ValueError('ref_ids must have one entry per read')
        st = None
        if strands is not None:
            st = np.ascontiguousarray(strands, dtype=np.uint8)
            if len(st) != n_tasks:
                raise ValueError('strands must have one entry per task')
        b = _native.Batch()
        b.n_reads = n
        b.reads = arena.ctypes.data if arena.size else offsets.ctypes.data
        b.offsets = offsets.ctypes.data
        b.ref_ids = rid.ctypes.data if rid is not None else None
        b.strands = st.ctypes.data if st is not None else None
        b.all_refs = 1 if all_refs else 0
        b.max_read_len = max_lj
        b.aln_read = aln_read.ctypes.data
        b.aln_ref = aln_ref.ctypes.data
        b.aln_stride = stride
        b.records = records.ctypes.data
        b.flags = 1 if legacy else 0
        self.ctx.align_classify_host(b)
        return BatchResult(aln_read, aln_ref, records, n, self.n_refs, all_refs)

    def align_device(self, n_reads, d_reads, d_offsets, d_aln_read, d_aln_ref, d_records, aln_stride, max_read_len,
                     d_ref_ids=None, d_strands=None, all_refs=False, stream=None, legacy=False, min_read_len=0, d_hints=None):
        """All d_* are device addresses (ints).  Enqueues one launch on `stream` and returns immediately.
        min_read_len: the shortest read of the batch if the caller knows it (c2_batch.min_read_len: band tiers no read of that length range can use
        are not launched -- never changes a result)."""
        b = _native.Batch()
        b.n_reads = int(n_reads)
        b.reads = d_reads
        b.offsets = d_offsets
        b.ref_ids = d_ref_ids
        b.strands = d_strands
        b.all_refs = 1 if all_refs else 0
        b.max_read_len = int(max_read_len)
        b.aln_read = d_aln_read
        b.aln_ref = d_aln_ref
        b.aln_stride = int(aln_stride)
        b.records = d_records
        b.flags = 1 if legacy else 0
        b.min_read_len = int(min_read_len)
        b.diag_hints = d_hints                                      # (c2_batch.diag_hints: n_tasks x 4 uint32 on the device, or None)
        self.ctx.align_classify_device(b, stream)
