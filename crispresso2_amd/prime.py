"""Batch benefit for the UNCHANGED caller.

The reference's hot loop calls `CRISPResso2Align.global_align` once per unique read and candidate amplicon (and once more for
the reverse complement when the seeds ask for it) and `find_indels_substitutions` once per best alignment
(CRISPRessoCORE.py:667-679, :721-724, loop :1957-1981, workers :1226-1232).  Behind the per-call shim every such call is one
kernel launch plus a synchronisation.  This module lets the two shim modules answer those calls from ONE device batch:

    register_reads(fastq path | list of reads)      the reads the run is going to align (a speculation, never a requirement)
    prime(reads, refs, matrix, gap_incentives, go, ge, include_idxs=None)      the same, eagerly, for known amplicons

With reads registered, `global_align(read, ref, ...)` first looks in the memo of (ref, scoring, gap parameters, gap incentive);
the first miss whose read IS a registered read aligns ALL registered unique reads against that reference in one batch (the
default launch chain, `BatchAligner.align`) and the calls that follow are dictionary look-ups.  A miss whose read is the reverse
complement of a registered read does the same for the reverse complements (the reference aligns `reverse_complement(read)` for
the other strand, :672-679); any other call -- the run's set-up alignments -- stays per call.  `find_indels_substitutions(aligned_read, aligned_ref, include_idxs)` works the same way over the primed
alignments of the reference that `aligned_ref` spells (`c2_classify_lists_batch`: two launches per 32 k alignments); payload
objects are built at look-up time from the flat lists, so nothing is materialised for alignments nobody asks about.

Every answer from the memo is keyed by the exact argument strings and parameter values, and was computed by the same kernels
as the per-call path (results never depend on batching: tests/test_gpu_parity.py), so a hit returns what the call would have
returned; anything else -- other reads, other parameters, quality-filtered or trimmed reads that differ from the registered
file -- is a miss and takes the per-call path.  No CPU fallback, no oracle.

By default (no environment variable, no code change in the caller) the module watches sys.argv: when the process was started with the
reference's own command line -- `-r1 / --fastq_r1 <file>` and no `-r2 / --fastq_r2` -- that file is registered, so an unmodified
`CRISPResso -r1 ...` run gets its hot loop's alignments from one batch per amplicon.  It is a speculation and fails silently: a file
that cannot be read, reads the run filters or trims before it aligns them, more unique reads than C2_PRIME_MAX_READS -- every call
that finds no primed answer takes the per-call path, and the results are the same either way.  C2_PRIME_FROM_ARGV=0 switches the
watching off; C2_PRIME_FASTQ=<path> registers a file at import whatever the command line says; `counters()` (and C2_PRIME_REPORT=1: a
line on stderr at exit) say how many calls were answered from a batch.
"""
import os
import sys

import numpy as np

from . import _native
from .refs import reverse_complement

MAX_READS = int(os.environ.get("C2_PRIME_MAX_READS", 4_000_000))      # unique reads; larger runs belong on pipeline.quantify_fastq

_WATCH_ARGV = os.environ.get("C2_PRIME_FROM_ARGV", "1").strip().lower() not in ("", "0", "no", "off", "false")
_state = {"reads": None, "source": None, "read_set": frozenset()}
_align_memo = {}                        # key -> _Primed
_classify_memo = {}                     # (ref sequence, include key, legacy) -> _PrimedLists
_miss = {}
stats = {"batches": 0, "classify_batches": 0, "align_hits": 0, "align_misses": 0, "classify_hits": 0, "classify_misses": 0,
         "per_call_align": 0, "per_call_classify": 0, "not_primed": 0}


def counters():
    """hit / miss counters of the run so far: batches launched, calls answered from them (align_hits, classify_hits), calls that went per
    call, whether the registered file could not be used (not_primed)"""
    out = dict(stats)
    out["registered"] = _state["source"] if isinstance(_state["source"], (str, bytes, os.PathLike)) else (None if _state["source"] is None else "<reads>")
    return out


def _report_at_exit():
    if os.environ.get("C2_PRIME_REPORT") and (_state["source"] is not None or stats["per_call_align"]):
        sys.stderr.write("crispresso2_amd.prime: %s\n" % " ".join("%s=%s" % kv for kv in sorted(counters().items())))


import atexit
atexit.register(_report_at_exit)


def clear():
    """Forget the registered reads and every memo (a new run)."""
    _state["reads"] = _state["source"] = None
    _state["read_set"] = frozenset()
    _align_memo.clear()
    _classify_memo.clear()
    _miss.clear()
    for k in stats:
        stats[k] = 0


def register_reads(source):
    """source: a FASTQ path (plain or .gz; de-duplicated by the native parser, as process_fastq does, CRISPRessoCORE.py:1820-1849)
    or an iterable of read strings.  Replaces what was registered before."""
    clear()
    _state["source"] = source


def _reads():
    """-> list of unique read strings (loaded on first use: the run may never reach the hot loop)"""
    if _state["reads"] is None and _state["source"] is not None:
        src = _state["source"]
        if isinstance(src, (str, bytes, os.PathLike)):
            try:
                arena, offsets, counts, n_reads = _native.fastq_unique(os.fspath(src))
            except (_native.NativeError, OSError) as e:               # (a speculation: the run itself will say what is wrong with its input)
                _state["reads"] = []
                stats["not_primed"] = 1
                _state["why_not"] = str(e)
                return _state["reads"]
            if len(counts) > MAX_READS:
                _state["reads"] = []
                sys.stderr.write("crispresso2_amd.prime: %d unique reads exceed C2_PRIME_MAX_READS=%d -- not priming (use "
                                 "pipeline.quantify_fastq for runs of this size)\n" % (len(counts), MAX_READS))
                return _state["reads"]
            buf = arena.tobytes()
            seqs = []
            for i in range(len(counts)):
                try:
                    seqs.append(buf[int(offsets[i]):int(offsets[i + 1])].decode("utf-8"))
                except UnicodeDecodeError:
                    pass
            _state["reads"] = [s for s in seqs if s]
        else:
            _state["reads"] = list(dict.fromkeys(s for s in src if s))
        _state["read_set"] = frozenset(_state["reads"])
    return _state["reads"] or []


def _find(seqi, matrix, gap_incentive, gap_open, gap_extend, create=False):
    """the memo entry of (reference, gap parameters, score matrix, gap incentive) -- the arrays compared by CONTENT with the copies
    taken when the entry was made (8100 + L int64 compares, a few microseconds: no stale hit after an in-place change)"""
    key = (seqi, int(gap_open), int(gap_extend), matrix.shape)
    for m_, g_, P in _align_memo.get(key, ()):
        if np.array_equal(g_, gap_incentive) and np.array_equal(m_, matrix):
            return key, P
    if not create:
        return key, None
    P = _Primed()
    _align_memo.setdefault(key, []).append((matrix.copy(), gap_incentive.copy(), P))
    return key, P


class _Primed:
    """the alignments of a set of reads against one reference under one parameter set, as the batch returned them"""

    def __init__(self):
        self.index = {}                  # read string -> (part, row)
        self.parts = []                  # BatchResult per batch
        self.done_rc = False
        self.done_fw = False

    def add(self, reads, seqi, matrix, gap_incentive, gap_open, gap_extend):
        from .batch import BatchAligner
        if not reads:
            return
        al = BatchAligner([seqi], [gap_incentive], [[]], matrix, gap_open, gap_extend, ctx=_native.default_context())
        res = al.align(reads)
        stats["batches"] += 1
        part = len(self.parts)
        self.parts.append(res)
        for row, rd in enumerate(reads):
            self.index.setdefault(rd, (part, row))

    def get(self, seqj):
        hit = self.index.get(seqj)
        if hit is None:
            return None
        res = self.parts[hit[0]]
        rec = res.records[hit[1]]
        if rec["status"] != 0:
            return None                  # the per-call path raises the reference's error for it
        s1, s2 = res.strings(hit[1])
        return s1, s2, round(100 * int(rec["matches"]) / float(int(rec["aln_len"])), 3)


def lookup_alignment(pystr_seqj, pystr_seqi, matrix, gap_incentive, gap_open, gap_extend):
    """-> (aligned_read, aligned_ref, score) from a primed batch, or None (the caller then takes the per-call path)"""
    if _WATCH_ARGV:
        _from_environment()
    if _state["source"] is None and not _align_memo:
        return None
    key, P = _find(pystr_seqi, matrix, gap_incentive, gap_open, gap_extend)
    if P is not None:
        got = P.get(pystr_seqj)
        if got is not None:
            stats["align_hits"] += 1
            return got
    stats["align_misses"] += 1
    if _state["source"] is None:
        return None
    reads = _reads()
    if not reads:
        return None
    # only a call of the hot loop starts a batch: its read is one of the registered reads, or the reverse complement of one (the
    # reference aligns reverse_complement(read) for the other strand, CRISPRessoCORE.py:672-679) -- the run's set-up alignments
    # (guides against the amplicon, amplicons against each other) stay per call
    read_set = _state["read_set"]
    stage = None
    if pystr_seqj in read_set:
        stage = "fw"
    else:
        try:
            if reverse_complement(pystr_seqj) in read_set:
                stage = "rc"
        except KeyError:
            pass
    if stage is None:
        return None
    if P is None:
        key, P = _find(pystr_seqi, matrix, gap_incentive, gap_open, gap_extend, create=True)
    if stage == "fw" and not P.done_fw:
        P.done_fw = True
        P.add(reads, pystr_seqi, matrix, gap_incentive, gap_open, gap_extend)
    elif stage == "rc" and not P.done_rc:
        P.done_rc = True
        P.add([r for r in _reverse_complements(reads) if r not in P.index], pystr_seqi, matrix, gap_incentive, gap_open, gap_extend)
    got = P.get(pystr_seqj)
    if got is not None:
        stats["align_hits"] += 1
        stats["align_misses"] -= 1
    return got


def _reverse_complements(reads):
    out = []
    for rd in reads:
        try:
            out.append(reverse_complement(rd))
        except KeyError:                                             # a character outside ACGTN_-: the reference raises for it
            pass
    return list(dict.fromkeys(out))


def prime(reads, ref_seqs, matrix, gap_incentives, gap_open, gap_extend, both_strands=True):
    """Eager form: align `reads` (list of strings or a FASTQ path) against every sequence of ref_seqs now, one batch per
    reference (and one more for the reverse complements), so that the run's global_align calls are look-ups from the start."""
    register_reads(reads)
    rds = _reads()
    m = np.ascontiguousarray(matrix, dtype=np.int64)
    for seqi, g in zip(ref_seqs, gap_incentives):
        g = np.ascontiguousarray(g, dtype=np.int64)
        key, P = _find(seqi, m, g, gap_open, gap_extend, create=True)
        if P.done_fw:
            continue
        P.done_fw = True
        P.add(rds, seqi, m, g, gap_open, gap_extend)
        if both_strands:
            P.done_rc = True
            P.add([r for r in _reverse_complements(rds) if r not in P.index], seqi, m, g, gap_open, gap_extend)


class _PrimedLists:
    """the classifier lists of every primed alignment against one reference for one include set, flat as the batch returned them"""

    def __init__(self):
        self.index = {}                  # (aligned read, aligned ref) -> row
        self.flat = None                 # (index int64, values int32, counts int64 [n, 3])


def lookup_payload(read_seq_al, ref_seq_al, include_idx, legacy, build):
    """-> the classifier's result for this pair from a batched run over the primed alignments, or None.  build(res, counts) is the
    shim's own payload constructor (ResultsSlotsDict / dict)."""
    if not _align_memo or not isinstance(read_seq_al, str) or not isinstance(ref_seq_al, str):
        return None
    try:
        inc_arr = include_idx if isinstance(include_idx, np.ndarray) else np.asarray(list(include_idx))
        inc_arr = inc_arr.astype(np.int64, copy=False)
    except (TypeError, ValueError):
        return None
    ref = ref_seq_al.replace("-", "")
    ckey = (ref, inc_arr.tobytes(), bool(legacy))
    inc = inc_arr.tolist
    PL = _classify_memo.get(ckey)
    if PL is None:
        sources = [P for key, entries in _align_memo.items() if key[0] == ref for _, _, P in entries if P.parts]
        if not sources:
            return None
        rd = read_seq_al.replace("-", "")
        if not any(rd in P.index for P in sources):                   # not an alignment of a primed read: stays per call
            stats["classify_misses"] += 1
            return None
        PL = _classify_memo[ckey] = _PrimedLists()
        rows_a, rows_f, lens = [], [], []
        for P in sources:
            for res in P.parts:
                ok = np.nonzero(res.records["status"] == 0)[0]
                if len(ok) == 0:
                    continue
                rows_a.append((res.aln_read, ok))
                rows_f.append((res.aln_ref, ok))
                lens.append(res.records["aln_len"][ok].astype(np.int32))
        if not lens:
            return None
        stride = max(a.shape[1] for a, _ in rows_a)
        n = int(sum(len(ok) for _, ok in rows_a))
        A = np.zeros((n, stride), dtype=np.uint8)
        F = np.zeros((n, stride), dtype=np.uint8)
        pos = 0
        for (a, ok), (f, _) in zip(rows_a, rows_f):
            A[pos:pos + len(ok), :a.shape[1]] = a[ok]
            F[pos:pos + len(ok), :f.shape[1]] = f[ok]
            pos += len(ok)
        ln = np.concatenate(lens)
        PL.flat = _native.default_context().classify_lists_batch(A, F, ln, None, [inc()], legacy=bool(legacy))
        stats["classify_batches"] += 1
        for row in range(n):
            T = int(ln[row])
            PL.index.setdefault((A[row, :T].tobytes(), F[row, :T].tobytes()), row)
    try:
        row = PL.index.get((read_seq_al.encode("utf-8"), ref_seq_al.encode("utf-8")))
    except UnicodeEncodeError:
        row = None
    if row is None:
        stats["classify_misses"] += 1
        return None
    stats["classify_hits"] += 1
    index, values, counts = PL.flat
    from .CRISPRessoCOREResources import _LISTS, _PAIRS, _CHARS
    res = {}
    base = row * _native.LIST_COUNT
    for k, name in enumerate(_LISTS):
        v = values[index[base + k]:index[base + k + 1]].tolist()
        if name in _PAIRS:
            v = [(v[i], v[i + 1]) for i in range(0, len(v), 2)]
        elif name in _CHARS:
            v = np.array([chr(c) for c in v])
        res[name] = v
    return build(res, counts[row])


_argv_seen = [None]


def _from_environment():
    """C2_PRIME_FASTQ, or (C2_PRIME_FROM_ARGV=1) the -r1 / --fastq_r1 of the command line the process runs -- looked at again
    whenever sys.argv has changed (a host that runs several CRISPResso commands in one interpreter)."""
    path = os.environ.get("C2_PRIME_FASTQ")
    if not path and _WATCH_ARGV:
        argv = list(sys.argv)
        if argv == _argv_seen[0]:
            return
        _argv_seen[0] = argv
        for flag in ("-r1", "--fastq_r1"):
            if flag in argv[:-1]:
                path = argv[argv.index(flag) + 1]
        for a in argv:
            if a.startswith("--fastq_r1="):
                path = a.split("=", 1)[1]
        # paired input is merged / aligned pair by pair (process_paired_fastq): the reads of R1 are not what the hot loop aligns
        if any(a in ("-r2", "--fastq_r2") or a.startswith("--fastq_r2=") for a in argv):
            path = None
    elif path and _state["source"] is not None:
        return
    if path and os.path.exists(path) and _state["source"] != path:
        register_reads(path)


_from_environment()
