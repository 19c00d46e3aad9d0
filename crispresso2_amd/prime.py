"""Batch benefit for the UNCHANGED caller.

The reference's hot loop calls `CRISPResso2Align.global_align` once per unique read and candidate amplicon (and once more for
the reverse complement when the seeds ask for it) and `find_indels_substitutions` once per best alignment
(CRISPRessoCORE.py:667-679, :721-724, loop :1957-1981, workers :1226-1232; for pairs 2-4 alignments per pair and amplicon and the
classifier on the pair's consensus, :1032-1048, :1086-1089).  Behind the per-call shim every such call is one kernel launch plus
a synchronisation.  This module lets the two shim modules answer those calls from ONE device batch per amplicon and strand.

Where the reads of the batch come from, in the order they are tried:

  1. `prime(...)` / `register_reads(...)`: the host says so.
  2. the CALLER'S OWN FRAMES (`_discover`).  When a call misses the memo, the stack is searched for the reference's read loops by
     the names of their locals -- `process_fastq` (:1735: `variantCache`, whose keys ARE the unique reads it is about to align),
     `process_paired_fastq` (:1245: `fastq1_filename` / `fastq2_filename`, or its `variantCache` of 'read1+read2' keys with
     first-occurrence qualities on the `-p N` route), `process_single_fastq_write_bam_out` (:2351: `fastq_input`) -- each of which
     also holds `args`, `refs`, `ref_names`, `aln_matrix`.  With those in hand everything the loop will ask for is computed at
     once: every read against every amplicon on the strand(s) the reference's seed test picks, the classifier over all of these
     alignments with each amplicon's `include_idxs`, and for pairs the consensus of the two reads' alignments (the device's
     restatement of `get_consensus_alignment_from_pairs`) with ITS classification.
  3. the same search runs in an `os.register_at_fork` hook BEFORE the reference forks its `-p N` workers (:1870-1898, :1357-1378):
     a forked child cannot use the HIP runtime it inherits (nor start another), so what the workers will ask for has to be in the
     parent's memory when they are forked.  The children answer from the inherited memo; a call that still misses there is served
     by a spawned helper process (`_native.forked_child_helper`), never by the inherited GPU context.
  4. sys.argv (`-r1 / --fastq_r1 <file>` without `-r2`), C2_PRIME_FASTQ: a host that calls the per-read functions from a loop of
     its own.  Lazy: the first miss whose read is a registered read aligns all registered reads against that reference.

Every answer from the memo is keyed by the exact argument strings and parameter values, and was computed by the same kernels
as the per-call path (results never depend on batching: tests/test_gpu_parity.py), so a hit returns what the call would have
returned; anything else -- other reads, other parameters, quality-filtered or trimmed reads that differ from the registered
file -- is a miss and takes the per-call path.  All of it is a speculation that fails silently: a frame that does not look as
expected, a file that cannot be read, more unique reads than C2_PRIME_MAX_READS -- the calls go per call and the results are the
same either way.  No CPU fallback, no oracle.

C2_PRIME_FROM_ARGV=0 switches the argv watching off, C2_PRIME_FROM_FRAMES=0 the frame search (and with it the fork hook);
`counters()` (and C2_PRIME_REPORT=1: a line on stderr at exit) say how many calls were answered from a batch.
"""
import os
import sys

import numpy as np

from . import _native
from .refs import reverse_complement

MAX_READS = int(os.environ.get("C2_PRIME_MAX_READS", 4_000_000))      # unique reads; larger runs belong on pipeline.quantify_fastq


def _on(name):
    return os.environ.get(name, "1").strip().lower() not in ("", "0", "no", "off", "false")


_WATCH_ARGV = _on("C2_PRIME_FROM_ARGV")
_WATCH_FRAMES = _on("C2_PRIME_FROM_FRAMES")
_state = {"reads": None, "source": None, "read_set": frozenset()}
_pairs = {"keys": {}, "items": []}       # 'read1+read2' -> index; (read1, read2, qual1, qual2) of the pairs a caller's frames named
_align_memo = {}                        # key -> [(matrix, gap_incentive, _Primed)]
_classify_memo = {}                     # (ref sequence, include key, legacy) -> _PrimedLists
_discovered = set()
stats = {"batches": 0, "classify_batches": 0, "consensus_batches": 0, "align_hits": 0, "align_misses": 0, "classify_hits": 0,
         "classify_misses": 0, "per_call_align": 0, "per_call_classify": 0, "not_primed": 0, "from_frames": 0, "before_fork": 0}


def counters():
    """hit / miss counters of the run so far: batches launched, calls answered from them (align_hits, classify_hits), calls that went per
    call, whether the registered file could not be used (not_primed), how often a caller's frames named the reads (from_frames; of these
    before a fork: before_fork)"""
    out = dict(stats)
    out["registered"] = _state["source"] if isinstance(_state["source"], (str, bytes, os.PathLike)) else (None if _state["source"] is None else "<reads>")
    return out


def _report_at_exit():
    if os.environ.get("C2_PRIME_REPORT") and (_state["source"] is not None or stats["per_call_align"] or stats["from_frames"]):
        sys.stderr.write("crispresso2_amd.prime: %s\n" % " ".join("%s=%s" % kv for kv in sorted(counters().items())))


import atexit
atexit.register(_report_at_exit)


def clear():
    """Forget the registered reads and every memo (a new run)."""
    _state["reads"] = _state["source"] = None
    _state["read_set"] = frozenset()
    _pairs["keys"].clear()
    del _pairs["items"][:]
    _align_memo.clear()
    _classify_memo.clear()
    _discovered.clear()
    for k in stats:
        stats[k] = 0


def register_reads(source):
    """source: a FASTQ path (plain or .gz; de-duplicated by the native parser, as process_fastq does, CRISPRessoCORE.py:1820-1849)
    or an iterable of read strings.  Replaces what was registered before."""
    clear()
    _state["source"] = source


def _say_too_many(n, what):
    """one line on stderr, on every route, when C2_PRIME_MAX_READS stops priming (the calls then take the per-call path: same results, slower)"""
    sys.stderr.write("crispresso2_amd.prime: %d unique %s exceed C2_PRIME_MAX_READS=%d -- not priming (use pipeline.quantify_fastq / "
                     "paired_device.quantify_paired_fastq for runs of this size)\n" % (n, what, MAX_READS))


def _unique_reads_of_file(path):
    """-> list of the file's unique read strings in first-seen order, or None (and a note in stats) when it cannot be used"""
    try:
        arena, offsets, counts, n_reads = _native.fastq_unique(os.fspath(path))
    except (_native.NativeError, OSError) as e:                       # (a speculation: the run itself will say what is wrong with its input)
        stats["not_primed"] = 1
        _state["why_not"] = str(e)
        return None
    if len(counts) > MAX_READS:
        stats["not_primed"] = 1
        _say_too_many(len(counts), "reads of %s" % (path,))
        return None
    buf = arena.tobytes()
    seqs = []
    for i in range(len(counts)):
        try:
            seqs.append(buf[int(offsets[i]):int(offsets[i + 1])].decode("utf-8"))
        except UnicodeDecodeError:
            pass
    return [s for s in seqs if s]


def _reads():
    """-> list of unique read strings (loaded on first use: the run may never reach the hot loop)"""
    if _state["reads"] is None and _state["source"] is not None:
        src = _state["source"]
        if isinstance(src, (str, bytes, os.PathLike)):
            _state["reads"] = _unique_reads_of_file(src) or []
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
        reads = [r for r in dict.fromkeys(reads) if r and r not in self.index]
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
    key, P = _find(pystr_seqi, matrix, gap_incentive, gap_open, gap_extend)
    if P is not None:
        got = P.get(pystr_seqj)
        if got is not None:
            stats["align_hits"] += 1
            return got
    if _native.in_forked_child():                                     # nothing can be launched from here (see the module text, 3.)
        stats["align_misses"] += 1
        return None
    if _WATCH_FRAMES and _discover(sys._getframe(1)):                 # a read loop of the reference is on the stack and has not been seen yet
        key, P = _find(pystr_seqi, matrix, gap_incentive, gap_open, gap_extend)
        got = P.get(pystr_seqj) if P is not None else None
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
        P.add(_reverse_complements(reads), pystr_seqi, matrix, gap_incentive, gap_open, gap_extend)
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
            P.add(_reverse_complements(rds), seqi, m, g, gap_open, gap_extend)


# ---------------------------------------------------------------- what a caller's frames say (module text, 2. and 3.)
_READ_LOOPS = ("process_fastq", "process_paired_fastq", "process_single_fastq_write_bam_out")
_RUN_LOCALS = ("args", "refs", "ref_names")


def _discover(frame, before_fork=False):
    """Search the stack from `frame` outwards for one of the reference's read loops; prime what it is going to ask for.
    -> True when something new was primed."""
    f = frame
    depth = 0
    while f is not None and depth < 64:
        if f.f_code.co_name in _READ_LOOPS:
            loc = f.f_locals
            if all(k in loc for k in _RUN_LOCALS):
                try:
                    return _prime_read_loop(f.f_code.co_name, loc, before_fork)
                except Exception as e:                                # (a speculation: whatever is odd about the frame, the calls go per call)
                    stats["not_primed"] = 1
                    _state["why_not"] = "%s: %s" % (type(e).__name__, e)
                    return False
        f = f.f_back
        depth += 1
    return False


def _matrix_of(loc):
    m = loc.get("aln_matrix")
    if m is None:                                                     # (read again from where the loop itself reads it, CRISPRessoCORE.py:1814-1816)
        return None
    return np.ascontiguousarray(m, dtype=np.int64)


def _prime_read_loop(name, loc, before_fork):
    args, refs, ref_names = loc["args"], loc["refs"], loc["ref_names"]
    m = _matrix_of(loc)
    if m is None:
        return False
    cache = loc.get("variantCache")
    reads = pairs = None
    if name == "process_paired_fastq":
        if isinstance(cache, dict) and cache and all(isinstance(v, list) for v in cache.values()):
            mark = ("cache", id(cache), len(cache))                   # the -p N route: keys + first-occurrence qualities are in the cache (:1296-1334)
            if mark in _discovered:
                return False
            pairs = []
            for k, v in cache.items():
                a = k.split('+')
                q = v[1].split(' ')
                if len(a) == 2 and len(q) == 2:
                    pairs.append((a[0], a[1], q[0], q[1]))
        else:
            f1, f2 = loc.get("fastq1_filename"), loc.get("fastq2_filename")
            mark = ("files", f1, f2)
            if not f1 or not f2 or mark in _discovered:
                return False
            _discovered.add(mark)
            pf = _native.PairedFastq(os.fspath(f1), os.fspath(f2))
            try:
                if len(pf.counts) > MAX_READS:
                    stats["not_primed"] = 1
                    _say_too_many(len(pf.counts), "read pairs of %s / %s" % (f1, f2))
                    return False
                pairs = []
                for k, q in zip(pf.keys, pf.quals):
                    a, b = k.split('+'), q.split(' ')
                    if len(a) == 2 and len(b) == 2:
                        pairs.append((a[0], a[1], b[0], b[1]))
            finally:
                pf.close()
    elif isinstance(cache, dict) and cache and name == "process_fastq":
        mark = ("cache", id(cache), len(cache))
        if mark in _discovered:
            return False
        reads = [k for k, v in cache.items() if isinstance(k, str) and k and not isinstance(v, dict)]
    else:
        path = loc.get("fastq_input") or loc.get("fastq_filename")
        mark = ("file", path)
        if not path or mark in _discovered or not os.path.exists(os.fspath(path)):
            return False
        _discovered.add(mark)
        reads = _unique_reads_of_file(path)
    _discovered.add(mark)
    n = len(pairs) if pairs is not None else len(reads or ())
    if n > MAX_READS:
        stats["not_primed"] = 1
        _say_too_many(n, "reads of the caller's %s" % name)
    if n == 0 or n > MAX_READS:
        return False
    prime_for_caller(args, refs, ref_names, m, reads=reads, pairs=pairs)
    stats["from_frames"] += 1
    if before_fork:
        stats["before_fork"] += 1
    return True


def _strands_needed(args, ref, seqs):
    """the reference's seed test (CRISPRessoCORE.py:655-679; over both reads of a pair :1024-1048) -> (forward wanted, reverse wanted)"""
    found_fw = found_rc = 0
    fw_seeds, rc_seeds = ref.get('fw_seeds', ()), ref.get('rc_seeds', ())
    for k in range(min(int(args.aln_seed_count), len(fw_seeds))):
        if any(fw_seeds[k] in s for s in seqs):
            found_fw += 1
        if any(rc_seeds[k] in s for s in seqs):
            found_rc += 1
    if found_fw > args.aln_seed_min and found_rc == 0:
        return True, False
    if found_fw == 0 and found_rc > args.aln_seed_min:
        return False, True
    return True, True


def _rc_or_none(s):
    try:
        return reverse_complement(s)
    except KeyError:
        return None


def prime_for_caller(args, refs, ref_names, aln_matrix, reads=None, pairs=None):
    """Everything `get_new_variant_object` (reads) / `get_new_variant_object_from_paired` (pairs: (read1, read2, qual1, qual2) with read2
    already in read1's orientation, as process_paired_fastq hands them over) will ask the two modules for, computed now:
    alignments per amplicon and strand, classifier lists with each amplicon's include_idxs, for pairs the consensus alignments and
    theirs.  Adds to what is primed already."""
    m = np.ascontiguousarray(aln_matrix, dtype=np.int64)
    go, ge = args.needleman_wunsch_gap_open, args.needleman_wunsch_gap_extend
    legacy = bool(getattr(args, "use_legacy_insertion_quantification", False))
    units = [(p[0], p[1]) for p in pairs] if pairs is not None else [(r,) for r in reads]
    if pairs is not None:
        for p in pairs:
            k = p[0] + '+' + p[1]
            if k not in _pairs["keys"]:
                _pairs["keys"][k] = len(_pairs["items"])
                _pairs["items"].append(tuple(p))
    for name in ref_names:
        ref = refs[name]
        seqi = ref['sequence']
        g = np.ascontiguousarray(ref['gap_incentive'], dtype=np.int64)
        fw, rc = [], []
        for u in units:
            want_fw, want_rc = _strands_needed(args, ref, u)
            if want_fw:
                fw.extend(u)
            if want_rc:
                rc.extend(x for x in (_rc_or_none(s) for s in u) if x is not None)
        key, P = _find(seqi, m, g, go, ge, create=True)
        P.add(fw, seqi, m, g, go, ge)
        P.add(rc, seqi, m, g, go, ge)
    for name in ref_names:
        inc_arr = _include_array(refs[name]['include_idxs'])
        if inc_arr is None:
            continue
        ckey = (refs[name]['sequence'], inc_arr.tobytes(), legacy)
        PL = _classify_memo.get(ckey)
        if PL is None:
            PL = _classify_memo[ckey] = _PrimedLists(refs[name]['sequence'], inc_arr, legacy)
        PL.extend()


# ---------------------------------------------------------------- the classifier over what is primed
class _PrimedLists:
    """the classifier lists of every primed alignment against one reference (and of every registered pair's consensus alignments) for
    one include set, flat as the batches returned them; grows when more is primed"""

    def __init__(self, ref, inc_arr, legacy):
        self.ref, self.inc, self.legacy = ref, inc_arr, legacy
        self.index = {}                  # (aligned read, aligned ref) -> (batch, row)
        self.flats = []                  # (index int64, values int32, counts int64 [n, 3]) per batch
        self.seen = {}                   # id(_Primed) -> number of its parts that are in
        self.pairs_in = set()            # (pair index, strand) whose consensus is in

    def sources(self):
        return [P for key, entries in _align_memo.items() if key[0] == self.ref for _, _, P in entries if P.parts]

    def _run(self, A, F, ln):
        flat = _native.default_context().classify_lists_batch(A, F, ln, None, [self.inc.tolist()], legacy=self.legacy)
        stats["classify_batches"] += 1
        b = len(self.flats)
        self.flats.append(flat)
        for row in range(len(ln)):
            T = int(ln[row])
            self.index.setdefault((A[row, :T].tobytes(), F[row, :T].tobytes()), (b, row))

    def extend(self):
        """classify what has been primed since the last call -> True when rows were added"""
        grew = False
        rows_a, rows_f, lens = [], [], []
        srcs = self.sources()
        for P in srcs:
            for res in P.parts[self.seen.get(id(P), 0):]:
                ok = np.nonzero(res.records["status"] == 0)[0]
                if len(ok):
                    rows_a.append(res.aln_read[ok])
                    rows_f.append(res.aln_ref[ok])
                    lens.append(res.records["aln_len"][ok].astype(np.int32))
            self.seen[id(P)] = len(P.parts)
        if lens:
            stride = max(a.shape[1] for a in rows_a)
            n = int(sum(len(x) for x in lens))
            A = np.zeros((n, stride), dtype=np.uint8)
            F = np.zeros((n, stride), dtype=np.uint8)
            pos = 0
            for a, f in zip(rows_a, rows_f):
                A[pos:pos + len(a), :a.shape[1]] = a
                F[pos:pos + len(f), :f.shape[1]] = f
                pos += len(a)
            self._run(A, F, np.concatenate(lens))
            grew = True
        if _pairs["items"] and srcs:
            grew = self._extend_pairs(srcs) or grew
        return grew

    def _extend_pairs(self, srcs):
        """the consensus alignments of the registered pairs whose two reads are both primed on a strand (the device's
        get_consensus_alignment_from_pairs, CRISPRessoCORE.py:829-984, with the qualities the caller will pass) -> classified"""
        from .paired import consensus_batch

        def get(s):
            for P in srcs:
                hit = P.get(s)
                if hit is not None:
                    return hit
            return None

        items, marks = [], []
        for i, (r1, r2, q1, q2) in enumerate(_pairs["items"]):
            for strand in (0, 1):
                if (i, strand) in self.pairs_in:
                    continue
                a, b = (r1, r2) if strand == 0 else (_rc_or_none(r1), _rc_or_none(r2))
                if a is None or b is None:
                    continue
                h1, h2 = get(a), get(b)
                if h1 is None or h2 is None:
                    continue
                items.append((h1[0], h1[1], h1[2], q1, h2[0], h2[1], h2[2], q2))
                marks.append((i, strand))
        if not items:
            return False
        cons = consensus_batch(items, ctx=_native.default_context(), errors="skip")
        stats["consensus_batches"] += 1
        self.pairs_in.update(marks)
        todo = list(dict.fromkeys((c[0], c[2]) for c in cons if c is not None and len(c[0]) == len(c[2]) and (c[0].encode(), c[2].encode()) not in self.index))
        if not todo:
            return True
        ln = np.array([len(f) for _, f in todo], dtype=np.int32)
        stride = max(16, (int(ln.max()) + 15) // 16 * 16)
        A = np.zeros((len(todo), stride), dtype=np.uint8)
        F = np.zeros((len(todo), stride), dtype=np.uint8)
        for k, (a, f) in enumerate(todo):
            A[k, :ln[k]] = np.frombuffer(a.encode(), dtype=np.uint8)
            F[k, :ln[k]] = np.frombuffer(f.encode(), dtype=np.uint8)
        self._run(A, F, ln)
        return True


def _include_array(include_idx):
    try:
        inc_arr = include_idx if isinstance(include_idx, np.ndarray) else np.asarray(list(include_idx))
        return np.ascontiguousarray(inc_arr.astype(np.int64, copy=False))
    except (TypeError, ValueError):
        return None


def lookup_payload(read_seq_al, ref_seq_al, include_idx, legacy, build):
    """-> the classifier's result for this pair from a batched run over the primed alignments, or None.  build(res, counts) is the
    shim's own payload constructor (ResultsSlotsDict / dict)."""
    if not _align_memo or not isinstance(read_seq_al, str) or not isinstance(ref_seq_al, str):
        return None
    inc_arr = _include_array(include_idx)
    if inc_arr is None:
        return None
    ref = ref_seq_al.replace("-", "")
    ckey = (ref, inc_arr.tobytes(), bool(legacy))
    try:
        pair_key = (read_seq_al.encode("utf-8"), ref_seq_al.encode("utf-8"))
    except UnicodeEncodeError:
        return None
    PL = _classify_memo.get(ckey)
    hit = PL.index.get(pair_key) if PL is not None else None
    if hit is None and not _native.in_forked_child():
        if PL is None:
            sources = [P for key, entries in _align_memo.items() if key[0] == ref for _, _, P in entries if P.parts]
            if not sources:
                return None
            rd = read_seq_al.replace("-", "")
            if not _pairs["items"] and not any(rd in P.index for P in sources):      # not an alignment of a primed read: stays per call
                stats["classify_misses"] += 1
                return None
            PL = _classify_memo[ckey] = _PrimedLists(ref, inc_arr, bool(legacy))
        if PL.extend():                                               # (alignments primed since the lists were made: the other strand's batch)
            hit = PL.index.get(pair_key)
    if hit is None:
        stats["classify_misses"] += 1
        return None
    stats["classify_hits"] += 1
    index, values, counts = PL.flats[hit[0]]
    row = hit[1]
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


# ---------------------------------------------------------------- fork (module text, 3.)
_fork_note = [False]


def _before_fork():
    """os.register_at_fork(before=...): what the caller's workers will ask for is computed in the parent, once, before they are forked (a forked child of a
    HIP process cannot use the GPU).  C2_PRIME_AT_FORK=0 switches the hook off (every miss of a worker then goes through its spawned helper: one more
    process and GPU context per worker); the first time the hook primes it says so on stderr -- it may read the whole FASTQ file and
    run GPU batches inside os.fork(), and it opens the GPU in the parent, which is what makes the workers children of a HIP process."""
    if not _WATCH_FRAMES or _native.in_forked_child() or os.environ.get("C2_PRIME_AT_FORK", "1") == "0":
        return
    try:
        if _discover(sys._getframe(1), before_fork=True) and not _fork_note[0]:
            _fork_note[0] = True
            sys.stderr.write("crispresso2_amd.prime: computed the alignments of the caller's reads before its fork() (%d device batches so far; C2_PRIME_AT_FORK=0 "
                             "switches this off)\n" % stats.get("batches", 0))
    except Exception:                                                 # never in the way of the caller's fork
        pass


if hasattr(os, "register_at_fork"):
    os.register_at_fork(before=_before_fork)


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
        # paired input is merged / aligned pair by pair (process_paired_fastq): the reads of R1 are not what the hot loop aligns -- that
        # route is primed from the caller's frames.  (`--fastq_r2=<file>` included; a shorter prefix is ambiguous with --fastq_r1 and argparse refuses it.)
        if any(a == "-r2" or a.startswith("--fastq_r2") for a in argv):
            path = None
    elif path and _state["source"] is not None:
        return
    if path and os.path.exists(path) and _state["source"] != path:
        register_reads(path)


_from_environment()
