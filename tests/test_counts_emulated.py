"""c2_count_vectors_kernel (compiled for the host by the wave emulator) against oracle/aggregate.py, the CPU restatement of
the reference's aggregation loop (CRISPRessoCORE.py:3964-4115) -- CPU only."""
import numpy as np
import pytest

import emu_driver as E
import oracle
from oracle import aggregate
from helpers import load_golden, matrices
from crispresso2_amd import counts as C


@pytest.fixture(scope="module")
def aligned():
    """realistic.json's 250-bp amplicon reads (incl. adjacent insertions added here), aligned by the emulated kernel."""
    E.build()
    mats = matrices()
    vecs = [v for v in load_golden("realistic.json") if len(v["seqi"]) == 250]
    amp, g, inc = vecs[0]["seqi"], vecs[0]["gap_incentive"], list(range(100, 150))
    reads = [v["seqj"] for v in vecs]
    # two insertions one reference base apart (numpy's fancy += counts the shared position once), a window-edge deletion
    reads.append(amp[:120] + "TT" + amp[120:121] + "GGG" + amp[121:])
    reads.append(amp[:95] + amp[103:])
    reads.append(amp[:-30])
    st = {}
    res, rec = E.align_batch(reads, [amp], [g], [inc], mats["EDNAFULL"], -20, -2, stats=st)
    return amp, inc, reads, res, rec, st["raw"]


def payloads(res, inc):
    out = []
    for s1, s2 in res:
        p = oracle.find_indels_substitutions(s1, s2, inc)
        p["aln_seq"], p["aln_ref"] = s1, s2
        out.append(p)
    return out


def compare(got, exp, L):
    for k, v in exp.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(got[k][:L], v), k
        else:
            assert got[k] == v, (k, got[k], v)


@pytest.mark.parametrize("flags", [0, C.FLAG_IGNORE_SUBSTITUTIONS, C.FLAG_IGNORE_INSERTIONS | C.FLAG_IGNORE_DELETIONS,
                                   C.FLAG_DISCARD_INDEL_READS])
def test_count_vectors_match_reference_aggregation(aligned, flags):
    amp, inc, reads, res, rec, (o1, o2) = aligned
    rng = np.random.default_rng(1)
    w = rng.integers(1, 50, len(reads)).astype(np.uint32)
    w[::11] = 0                                           # reads not assigned to this reference
    counts, lay = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, flags=flags)
    got = lay.unpack(counts, 0, len(amp))
    items = [(p, int(c)) for p, c in zip(payloads(res, inc), w) if c > 0]
    exp = aggregate.aggregate(items, len(amp), ignore_substitutions=bool(flags & 1), ignore_insertions=bool(flags & 2),
                              ignore_deletions=bool(flags & 4), discard_indel_reads=bool(flags & 8))
    compare(got, exp, len(amp))
    assert got["counts_total"] + got["counts_discarded"] == int(w.sum())


def test_count_vectors_min_alignment_score_gate(aligned):
    amp, inc, reads, res, rec, (o1, o2) = aligned
    thr = 97.5
    mm = C.min_matches_table([thr], int(rec["aln_len"].max()))
    counts, lay = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), min_matches=mm)
    got = lay.unpack(counts, 0, len(amp))
    keep = [round(100 * int(r["matches"]) / float(int(r["aln_len"])), 3) > thr for r in rec]
    assert 0 < sum(keep) < len(keep)
    exp = aggregate.aggregate([(p, 1) for p, k in zip(payloads(res, inc), keep) if k], len(amp))
    compare(got, exp, len(amp))


def test_count_vectors_interleaved_references_and_heavy_weights(aligned):
    """Tasks of two references interleaved inside one workgroup chunk (rounds per reference, flush on change) and read
    multiplicities beyond the int32 budget of the LDS accumulators (one-task-at-a-time path)."""
    amp, inc, reads, res, rec, (o1, o2) = aligned
    rng = np.random.default_rng(5)
    rec2 = rec.copy()
    rec2["ref_id"] = rng.integers(0, 2, len(rec2)).astype(np.uint16)
    w = rng.integers(1, 9, len(reads)).astype(np.uint32)
    w[3] = 30_000_000                  # beyond the load budget (weight x alignment length <= 2^30 per flush): added in pieces
    w[40] = 2_000_000_000              # close to the int32 limit of a weight: ~270 pieces
    w[41] = 25_000_000
    w[len(reads) - 3] = 1_900_000_000  # the read with two insertions (length sums are weight x size)
    w[len(reads) - 1] = 1_700_000_000  # a 30-base trailing deletion (difference arrays, deletion length sums)
    counts, lay = E.count_vectors(o1, o2, rec2, [amp, amp], [inc, inc], max(len(r) for r in reads), weights=w, grid=3)
    P = payloads(res, inc)
    for r in range(2):
        got = lay.unpack(counts, r, len(amp))
        items = [(p, int(c)) for p, c, rid in zip(P, w, rec2["ref_id"]) if rid == r]
        compare(got, aggregate.aggregate(items, len(amp)), len(amp))


def test_min_matches_table_is_the_reference_expression():
    mm = C.min_matches_table([60.0, 0.0], 300)
    for T in (1, 7, 64, 255, 300):
        for thr, row in ((60.0, mm[0]), (0.0, mm[1])):
            m = int(row[T])
            assert m > T or round(100 * m / float(T), 3) > thr
            assert m == 0 or not (round(100 * (m - 1) / float(T), 3) > thr)


# ---- --use_legacy_insertion_quantification on the count route -----------------------------------------------------------
_LIST_NAMES = ("ref_positions", "all_insertion_positions", "all_insertion_left_positions", "insertion_positions", "insertion_coordinates",
               "insertion_sizes", "all_deletion_positions", "all_deletion_coordinates", "deletion_positions", "deletion_coordinates",
               "deletion_sizes", "all_substitution_positions", "all_substitution_values", "substitution_positions", "substitution_values")


def legacy_payload(s1, s2, inc):
    """find_indels_substitutions_legacy's payload: from the reference's own module when oracle/_ref is built, else from the emulated
    c2_classify_lists_kernel with legacy = 1 (itself pinned against reference-generated vectors, test_kernel_emulated.py)."""
    ref = oracle.ref()
    if ref is not None:
        p = dict(ref[1].find_indels_substitutions_legacy(s1, s2, inc))
        p["insertion_n"], p["deletion_n"] = int(p["insertion_n"]), int(p["deletion_n"])
    else:
        lists, cnt = E.classify_lists(s1, s2, inc, legacy=True)
        p = dict(zip(_LIST_NAMES, lists))
        for k in ("insertion_coordinates", "all_deletion_coordinates", "deletion_coordinates"):
            p[k] = [(p[k][i], p[k][i + 1]) for i in range(0, len(p[k]), 2)]
        p["insertion_n"], p["deletion_n"], p["substitution_n"] = cnt
    p["aln_seq"], p["aln_ref"] = s1, s2
    return p


def test_legacy_classifier_on_the_count_route(monkeypatch):
    """Fused classifier + count kernel with the legacy rules (COREResources.pyx:190-315): an insertion counts when EITHER flank is in
    the window; a deletion that starts in column 0 or 1 gets reference start 0; one that reaches the end of the alignment stops
    one base short.  Reads built to hit exactly those cases, window edges next to the events; records and every count vector
    against the reference's legacy classifier + the restated aggregation loop."""
    E.build()
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(2024)
    amp = "".join(rng.choice(list("ACGT"), 120))
    g = np.zeros(121, dtype=np.int64); g[61] = 1
    reads = []
    for _ in range(60):
        t = list(amp)
        for _ in range(int(rng.integers(0, 3))):
            t[int(rng.integers(0, 120))] = str(rng.choice(list("ACGTN")))
        t = "".join(t)
        k = rng.random()
        if k < 0.2:   t = t[0] + t[int(rng.integers(2, 9)):]                           # deletion that starts in column 1
        elif k < 0.35: t = t[int(rng.integers(1, 7)):]                                  # leading deletion
        elif k < 0.55: t = t[:120 - int(rng.integers(1, 8))]                            # trailing deletion
        elif k < 0.75:
            p0 = int(rng.integers(40, 80)); t = t[:p0] + "".join(rng.choice(list("ACGT"), int(rng.integers(1, 6)))) + t[p0:]   # insertion
        elif k < 0.9:
            p0 = int(rng.integers(40, 80)); t = t[:p0] + t[p0 + int(rng.integers(1, 12)):]
        reads.append(t)
    reads += [amp[0] + amp[3:118], amp[:119], amp[1:]]                                  # column-1 start AND trailing; one-base trailing; one-base leading
    for inc in ([59, 60], list(range(55, 66)), [0, 1, 2], [117, 118, 119], list(range(0, 120))):
        monkeypatch.setenv("C2_EMU_LEGACY", "1")
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
        o1, o2 = st["raw"]
        P = [legacy_payload(s1, s2, inc) for s1, s2 in res]
        for k, p in enumerate(P):
            r = rec[k]
            assert r["status"] == 0
            assert (int(r["insertion_n"]), int(r["deletion_n"]), int(r["substitution_n"])) == (p["insertion_n"], p["deletion_n"], p["substitution_n"]), (inc[:3], k, res[k])
            assert int(r["all_deletion_bases"]) == len(p["all_deletion_positions"]), (k, res[k])
            assert int(r["win_insertion_events"]) == len(p["insertion_sizes"]) and int(r["all_insertion_events"]) == len(p["all_insertion_left_positions"])
            assert int(r["win_deletion_events"]) == len(p["deletion_sizes"]) and int(r["all_deletion_events"]) == len(p["all_deletion_coordinates"])
        w = rng.integers(1, 9, len(reads)).astype(np.uint32)
        counts, lay = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, flags=C.FLAG_LEGACY_CLASSIFIER)
        got = lay.unpack(counts, 0, len(amp))
        exp = aggregate.aggregate([(p, int(c)) for p, c in zip(P, w)], len(amp))
        compare(got, exp, len(amp))
    # and the default classifier disagrees on these reads (the test would not notice a flag that does nothing)
    monkeypatch.delenv("C2_EMU_LEGACY")
    res2, rec2 = E.align_batch(reads, [amp], [g], [list(range(55, 66))], m, -20, -2, band_lanes=-87)
    assert res2 == res
    assert (rec2["all_deletion_bases"] != [len(legacy_payload(s1, s2, [0])["all_deletion_positions"]) for s1, s2 in res]).any()


@pytest.mark.parametrize("flags", [0, C.FLAG_IGNORE_INSERTIONS | C.FLAG_IGNORE_DELETIONS, C.FLAG_DISCARD_INDEL_READS])
def test_count_vectors_with_the_accumulator_block_in_hbm(aligned, flags, monkeypatch):
    """c2_count_vectors_hbm_kernel: the workgroup's int32 block in global memory instead of LDS (amplicons beyond ~1,650 bp take it;
    forced here): same tensors, incl. the interleaved-references / heavy-weight rounds (flush per reference, weights in pieces)."""
    monkeypatch.setenv("C2_EMU_COUNT_HBM", "1")
    amp, inc, reads, res, rec, (o1, o2) = aligned
    rng = np.random.default_rng(1)
    w = rng.integers(1, 50, len(reads)).astype(np.uint32)
    w[::11] = 0
    counts, lay = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, flags=flags)
    got = lay.unpack(counts, 0, len(amp))
    items = [(p, int(c)) for p, c in zip(payloads(res, inc), w) if c > 0]
    exp = aggregate.aggregate(items, len(amp), ignore_substitutions=bool(flags & 1), ignore_insertions=bool(flags & 2),
                              ignore_deletions=bool(flags & 4), discard_indel_reads=bool(flags & 8))
    compare(got, exp, len(amp))
    if flags == 0:
        rec2 = rec.copy()
        rec2["ref_id"] = rng.integers(0, 2, len(rec2)).astype(np.uint16)
        w2 = rng.integers(1, 9, len(reads)).astype(np.uint32)
        w2[3] = 30_000_000
        w2[40] = 2_000_000_000
        counts, lay = E.count_vectors(o1, o2, rec2, [amp, amp], [inc, inc], max(len(r) for r in reads), weights=w2, grid=3)
        P = payloads(res, inc)
        for r in range(2):
            items = [(p, int(c)) for p, c, rid in zip(P, w2, rec2["ref_id"]) if rid == r]
            compare(lay.unpack(counts, r, len(amp)), aggregate.aggregate(items, len(amp)), len(amp))


@pytest.mark.parametrize("L,shift", [(300, 0), (560, 0), (700, 0), (300, 1)])
def test_alignments_longer_than_a_staging_window(L, shift):
    """The count kernel stages C2_CNT_STAGE_ROW (320) columns of an alignment's strings in LDS at a time: amplicons of 300 / 560 / 700 bp
    give alignments of one window (the eight-columns-per-lane walk, beyond 256 columns), 2 and 3 windows (the 64-column chunk walk) -- gap-free reads (dword walk), deletions and insertions that straddle a window's edge (the walk's
    carried state), a trailing deletion; shift 1: string rows that do not start on a dword (byte-wise copies)."""
    E.build()
    mats = matrices()
    rng = np.random.default_rng(L + shift)
    amp = "".join(rng.choice(list("ACGT"), L))
    inc = list(range(L // 2 - 20, L // 2 + 20)) + list(range(250, 262)) + ([510, 511, 512, 513] if L > 520 else [])
    g = np.zeros(L + 1, dtype=np.int64)
    g[L // 2 + 1] = 1

    def mut(s, k):
        s = list(s)
        for p in rng.integers(0, len(s), k):
            s[p] = "ACGTN"[rng.integers(0, 5)]
        return "".join(s)
    reads = [amp, mut(amp, 3), mut(amp, 9)]
    for edge in [256] + ([512] if L > 520 else []):
        reads += [amp[:edge - 5] + amp[edge + 6:], amp[:edge - 1] + amp[edge + 1:], amp[:edge] + "TTGTT" + amp[edge:], amp[:edge - 2] + "GG" + amp[edge - 2:],
                  mut(amp[:edge - 12] + amp[edge - 3:], 2), amp[:edge + 1] + "ACA" + amp[edge + 1:]]
    reads += [amp[:-25], amp[20:], mut(amp[:L // 2 - 4] + amp[L // 2 + 9:], 4)]
    res, rec = E.align_batch(reads, [amp], [g], [inc], mats["EDNAFULL"], -20, -2, stats=(st := {}))
    o1, o2 = st["raw"]
    assert int(rec["aln_len"].max()) > 256
    if shift:                                                          # rows that start one byte into a dword
        def shifted(o):
            buf = np.zeros(o.size + 8, dtype=np.uint8)
            v = buf[shift:shift + o.size].reshape(o.shape)
            v[:] = o
            return v
        o1, o2 = shifted(o1), shifted(o2)
        assert o1.ctypes.data % 4 == shift
    w = rng.integers(1, 9, len(reads)).astype(np.uint32)
    counts, lay = count_vectors_raw(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), w)
    got = lay.unpack(counts, 0, L)
    exp = aggregate.aggregate([(p, int(c)) for p, c in zip(payloads(res, inc), w)], L)
    compare(got, exp, L)


def count_vectors_raw(a1, a2, rec, ref_seqs, includes, max_read_len, weights, flags=0):
    """E.count_vectors without its np.ascontiguousarray of the string arrays (that would re-align shifted rows)"""
    import ctypes
    from crispresso2_amd.counts import CountLayout
    n, nrefs = len(rec), len(ref_seqs)
    lens = np.array([len(x) for x in ref_seqs], dtype=np.int32)
    lay = CountLayout(nrefs, int(lens.max()), max_read_len)
    counts = np.zeros(lay.shape(), dtype=np.int64)
    inc = [np.ascontiguousarray(np.asarray(list(x), dtype=np.int64).astype(np.int32)) for x in includes]
    ip = (ctypes.c_void_p * nrefs)(*[x.ctypes.data for x in inc])
    ninc = np.array([len(x) for x in inc], dtype=np.int32)
    seq_ptrs = (ctypes.c_char_p * nrefs)(*[x.encode() for x in ref_seqs])
    wq = np.ascontiguousarray(weights, dtype=np.uint32)
    rec = np.ascontiguousarray(rec)
    assert a1.strides == (a1.shape[1], 1) and a2.strides == (a2.shape[1], 1)
    rc = E.lib().emu_count_vectors(ctypes.c_uint64(n), ctypes.c_void_p(a1.ctypes.data), ctypes.c_void_p(a2.ctypes.data), ctypes.c_uint32(a1.shape[1]),
                                   rec.ctypes.data_as(ctypes.c_void_p), wq.ctypes.data_as(ctypes.c_void_p), None, 0, nrefs,
                                   lens.ctypes.data_as(ctypes.c_void_p), ip, ninc.ctypes.data_as(ctypes.c_void_p), int(flags), int(lay.hl),
                                   counts.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(2), seq_ptrs)
    assert rc == 0
    return counts, lay


@pytest.mark.parametrize("legacy", [False, True])
def test_event_walk_at_every_offset_of_an_eight_column_unit(legacy, monkeypatch):
    """An alignment with gaps that fits its staging slot is walked eight columns per lane in one pass: deletions and insertions of 1 .. 17 bases
    that start at every offset of a lane's eight columns (the run is accounted by the lane that holds the column BEHIND it, which walks back over
    the staged string), two events a few columns apart, adjacent insertions, events in the first and the last columns, a trailing deletion --
    each against the oracle's aggregation."""
    E.build()
    mats = matrices()
    rng = np.random.default_rng(77)
    L = 160
    amp = "".join(rng.choice(list("ACGT"), L))
    inc = list(range(60, 100))
    g = np.zeros(L + 1, dtype=np.int64)
    g[81] = 1
    reads = []
    for start in range(40, 56):
        for ln in (1, 3, 7, 8, 9, 16, 17):
            reads.append(amp[:start] + amp[start + ln:])                                   # deletion
            reads.append(amp[:start] + "".join(rng.choice(list("ACGT"), ln)) + amp[start:])   # insertion
    reads += [amp[:70] + amp[75:90] + "GG" + amp[90:], amp[:66] + "T" + amp[66:72] + amp[74:], amp[3:], amp[:-9], amp[:5] + amp[9:], amp[:L - 12] + amp[L - 8:],
              amp[:80] + amp[88:96] + amp[104:]]
    if legacy:
        monkeypatch.setenv("C2_EMU_LEGACY", "1")
    res, rec = E.align_batch(reads, [amp], [g], [inc], mats["EDNAFULL"], -20, -2, stats=(st := {}))
    o1, o2 = st["raw"]
    w = rng.integers(1, 6, len(reads)).astype(np.uint32)
    counts, lay = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, flags=C.FLAG_LEGACY_CLASSIFIER if legacy else 0)
    got = lay.unpack(counts, 0, L)
    if legacy:
        ref = oracle.ref()
        if ref is None:
            pytest.skip("the reference's compiled classifier (oracle/_ref) is not built here")
        items = []
        for (s1, s2), c in zip(res, w):
            p = dict(ref[1].find_indels_substitutions_legacy(s1, s2, inc))
            p["aln_seq"], p["aln_ref"] = s1, s2
            items.append((p, int(c)))
    else:
        items = [(p, int(c)) for p, c in zip(payloads(res, inc), w)]
    compare(got, aggregate.aggregate(items, L), L)


@pytest.mark.parametrize("L", [250, 256, 97, 33])
def test_gap_free_alignments_eight_at_a_time(L):
    """Gap-free alignments of at most 256 columns are walked eight at a time, eight lanes each, 32 columns per lane as four 64-bit words: a
    substitution at EVERY column (so every lane, word and byte position, the last partial word included), several per read, N in the read, reads
    equal to the amplicon in between, more than eight and fewer than eight per round, different weights -- against the oracle's aggregation.
    (Rows on 16-byte boundaries: what the route needs and what the batch aligner writes.)"""
    E.build()
    mats = matrices()
    rng = np.random.default_rng(L)
    amp = "".join(rng.choice(list("ACGT"), L))
    inc = list(range(max(0, L // 2 - 10), min(L, L // 2 + 10)))
    g = np.zeros(L + 1, dtype=np.int64)
    g[L // 2 + 1] = 1
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    reads = []
    for c in range(L):
        s = list(amp)
        s[c] = other[s[c]]
        if c % 7 == 3 and c + 9 < L:
            s[c + 9] = "N"
        if c % 5 == 1 and c >= 30:
            s[c - 30] = other[s[c - 30]]
        reads.append("".join(s))
        if c % 11 == 0:
            reads.append(amp)
    res, rec = E.align_batch(reads, [amp], [g], [inc], mats["EDNAFULL"], -20, -2, stats=(st := {}))
    o1, o2 = st["raw"]
    assert (rec["aln_len"] == L).all()                                  # (all of them gap-free)
    stride = (o1.shape[1] + 15) // 16 * 16

    def rows16(o):
        buf = np.zeros(o.shape[0] * stride + 16, dtype=np.uint8)
        off = (-buf.ctypes.data) % 16
        v = buf[off:off + o.shape[0] * stride].reshape(o.shape[0], stride)
        v[:, :o.shape[1]] = o
        return v
    a1, a2 = rows16(o1), rows16(o2)
    assert a1.ctypes.data % 16 == 0 and a2.ctypes.data % 16 == 0 and stride % 16 == 0
    w = rng.integers(1, 9, len(reads)).astype(np.uint32)
    counts, lay = count_vectors_raw(a1, a2, rec, [amp], [inc], L, w)
    compare(lay.unpack(counts, 0, L), aggregate.aggregate([(p, int(c)) for p, c in zip(payloads(res, inc), w)], L), L)
    # the same rows one byte off a 16-byte boundary take the one-at-a-time walk: same tensor
    buf = np.zeros(a1.size + 32, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16 + 4
    b1 = buf[off:off + a1.size].reshape(a1.shape)
    b1[:] = a1
    counts2, _ = count_vectors_raw(b1, a2, rec, [amp], [inc], L, w)
    assert np.array_equal(counts, counts2)


@pytest.mark.parametrize("flags", [0, C.FLAG_IGNORE_SUBSTITUTIONS, C.FLAG_IGNORE_INSERTIONS | C.FLAG_IGNORE_DELETIONS, C.FLAG_DISCARD_INDEL_READS])
@pytest.mark.parametrize("L", [250, 151])
def test_hinted_tasks_are_counted_from_their_hint_word_alone(L, flags):
    """Round 6: c2_align_partition_kernel leaves a hint word for every read it finishes itself (main diagonal, at most two
    differing bases) -- c2_batch.diag_hints --, and c2_count_vectors_hinted_device counts those tasks from the word alone (c2_count_hinted_kernel:
    neither the rows nor the record are read), the others as ever.  The tensor equals the one without hints entry by entry, and the reference's
    aggregation loop (oracle/aggregate.py) -- with weights incl. 0, one above the LDS limit (70,000) and one above 2^31 (clamped, as the kernel clamps),
    differing bases in and outside the window, at both ends, N's, and with the score gate at a value that two differing bases fail."""
    E.build()
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(4200 + L)
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = list(range(L // 2 - 8, L // 2 + 8))
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    sub = lambda s_, q, c=None: s_[:q] + (c or other[s_[q]]) + s_[q + 1:]
    reads = [amp] * 5
    for q in (0, 1, L // 2 - 9, L // 2 - 8, L // 2, L // 2 + 7, L // 2 + 8, L - 2, L - 1, 33):
        reads += [sub(amp, q), sub(amp, q, "N")]
    reads += [sub(sub(amp, 0), L - 1), sub(sub(amp, L // 2), L // 2 + 1), sub(sub(amp, L // 2 - 1, "N"), L // 2 + 2), sub(sub(amp, 10), 200 % L, "N"),
              sub(sub(amp, L // 2 - 30), L // 2 + 3), sub(sub(sub(amp, 5), 50), 100)]
    reads += [amp[:L // 2] + amp[L // 2 + 4:] + "ACGT", amp[:L // 2] + "TG" + amp[L // 2:-2], amp[:-3]]      # not on the main diagonal
    reads += [reads[int(k)] for k in rng.integers(0, len(reads), 300)]
    st = {"want_hints": True}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    o1, o2 = st["raw"]
    hints4 = st["hints"]
    hints = hints4[:, 0]
    n_hinted = int((hints >> 31).sum())
    assert n_hinted == st["exact_copies"] > 200 and (hints4[:, 1:][(hints >> 30) == 0] == 0).all() and (hints[(hints >> 30) == 0] == 0).all()   # every word is written: 0 where there is nothing to say
    for k in np.nonzero(hints >> 31)[0][:60]:                             # a hint restates its alignment
        kk = int((hints[k] >> 24) & 3)
        diff = [c for c in range(L) if reads[k][c] != amp[c]]
        assert len(reads[k]) == L and kk == len(diff) and int(rec[k]["aln_len"]) == L and int(rec[k]["matches"]) == L - kk
        for e, c in enumerate(diff):
            assert int((hints[k] >> (12 * e)) & 0x1ff) == c and "ACTG???N"[int((hints[k] >> (12 * e + 9)) & 7)] == reads[k][c]
    w = rng.integers(1, 50, len(reads)).astype(np.uint32)
    w[::11] = 0
    w[3] = 70000                                                          # (above c2_count_hinted_kernel's LDS limit: straight to the tensor)
    w[7] = 0x90000000                                                     # (clamped to 2^31 - 1 by both kernels)
    for mm in (None, C.min_matches_table([99.3], L + L)):                 # 99.3: two differing bases of 250 (99.2) fail, one (99.6) passes
        plain, lay = E.count_vectors(o1, o2, rec, [amp], [inc], L, weights=w, min_matches=mm, flags=flags)
        hinted, _ = E.count_vectors(o1, o2, rec, [amp], [inc], L, weights=w, min_matches=mm, flags=flags, hints=hints4)
        assert np.array_equal(plain, hinted), np.nonzero(plain != hinted)
        # ... and with the rows of the hinted tasks wiped: they are not read
        w1, w2 = o1.copy(), o2.copy()
        taken = ((hints >> 31) == 1) | (((hints >> 30) & 1) == 1) & (w < 1024)       # (a gapped hint is used below the hinted kernel's weight limit)
        w1[taken] = 0x58; w2[taken] = 0x59
        wiped, _ = E.count_vectors(w1, w2, rec, [amp], [inc], L, weights=w, min_matches=mm, flags=flags, hints=hints4)
        assert np.array_equal(plain, wiped)
    got = lay.unpack(hinted, 0, L)
    thr = 99.3
    items = [(p, int(min(c, 0x7fffffff))) for p, c, r in zip(payloads(res, inc), w, rec) if c > 0 and round(100 * int(r["matches"]) / float(int(r["aln_len"])), 3) > thr]
    exp = aggregate.aggregate(items, L, ignore_substitutions=bool(flags & 1), ignore_insertions=bool(flags & 2),
                              ignore_deletions=bool(flags & 4), discard_indel_reads=bool(flags & 8))
    compare(got, exp, L)


def _gapped_reads(rng, amp, n):
    """reads with one or two indels anywhere (also at the window's edges, in front, at the end), 0 .. 4 differing bases (N's among them, next to the gaps too)"""
    L = len(amp)
    out = []
    for k in range(n):
        t = list(amp)
        for _ in range(int(rng.integers(0, 5))):
            t[int(rng.integers(0, L))] = str(rng.choice(list("ACGTN")))
        for _ in range(int(rng.integers(1, 3))):
            kind = int(rng.integers(0, 6))
            p = int(rng.integers(0, len(t)))
            d = int(rng.integers(1, 25))
            if kind <= 1: del t[p:p + d]                                           # deletion
            elif kind <= 3: t[p:p] = list(rng.choice(list("ACGT"), d))             # insertion
            elif kind == 4: t = t[d:]                                              # the read starts inside the amplicon
            else: t = t[:len(t) - d] + list(rng.choice(list("ACGT"), int(rng.integers(0, 12))))   # ends early / runs over
        out.append("".join(t) if len(t) >= 40 else amp)
    return out


@pytest.mark.parametrize("flags", [0, C.FLAG_IGNORE_SUBSTITUTIONS, C.FLAG_IGNORE_INSERTIONS | C.FLAG_IGNORE_DELETIONS, C.FLAG_DISCARD_INDEL_READS])
@pytest.mark.parametrize("L,seed", [(250, 1), (250, 2), (120, 3)])
def test_gapped_hints_count_like_the_column_walk(L, seed, flags):
    """Round 6: the lane-group epilogue leaves a four-word hint for an alignment of at most five runs and three differing columns (C2_HINT_GAPPED: its runs, and
    where / what its differing columns are); c2_count_hinted_kernel counts it from the hint and its record -- the position vectors run by run, in closed
    form -- instead of walking the strings.  700 reads with indels of every kind: the tensor with the hints equals the tensor without them entry by entry
    (also with the hinted tasks' rows wiped), and the reference's aggregation loop; a hint restates its alignment's runs."""
    E.build()
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(9000 + seed)
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = list(range(L // 2 - 6, L // 2 + 7))
    reads = _gapped_reads(rng, amp, 700)
    st = {"want_hints": True}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    assert (rec["status"] == 0).all()
    o1, o2 = st["raw"]
    hints4 = st["hints"]
    h0 = hints4[:, 0]
    gapped = ((h0 >> 30) & 3) == 1
    assert gapped.sum() > 300, int(gapped.sum())
    for k in np.nonzero(gapped)[0][:120]:                                          # the runs of a hint are the runs of its strings
        s1, s2 = res[k]
        runs, cur, ln = [], None, 0
        for a_, b_ in zip(s1, s2):
            stt = 3 if a_ == "-" else (2 if b_ == "-" else 1)
            if stt == cur: ln += 1
            else:
                if cur is not None: runs.append((cur, ln))
                cur, ln = stt, 1
        runs.append((cur, ln))
        hw = [int(x) for x in hints4[k]]
        assert (hw[0] & 7) == len(runs) <= 5, (k, runs, hw)
        flds = [(hw[0] >> 5) & 0x7ff, (hw[0] >> 16) & 0x7ff, hw[1] & 0x7ff, (hw[1] >> 11) & 0x7ff, hw[2] & 0x7ff]
        assert [(f_ & 3, f_ >> 2) for f_ in flds[:len(runs)]] == runs, (k, runs, flds)
        diff = [(i, a_) for i, (a_, b_) in enumerate(zip(s1, s2)) if a_ != "-" and b_ != "-" and a_ != b_]
        assert ((hw[0] >> 3) & 3) == len(diff) <= 3
    w = rng.integers(1, 30, len(reads)).astype(np.uint32)
    w[::13] = 0
    gi = np.nonzero(gapped)[0]
    w[gi[0]] = 1023; w[gi[1]] = 1024; w[gi[2]] = 70000                             # at / above the hinted kernel's weight limit: the column walk's
    for mm in (None, C.min_matches_table([90.0], L + max(len(r) for r in reads))):
        plain, lay = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, min_matches=mm, flags=flags)
        hinted, _ = E.count_vectors(o1, o2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, min_matches=mm, flags=flags, hints=hints4)
        bad = np.nonzero(plain != hinted)
        assert np.array_equal(plain, hinted), (bad, plain[bad][:8], hinted[bad][:8])
        taken = ((h0 >> 31) == 1) | (gapped & (w < 1024))
        w1, w2 = o1.copy(), o2.copy()
        w1[taken] = 0x58; w2[taken] = 0x59
        wiped, _ = E.count_vectors(w1, w2, rec, [amp], [inc], max(len(r) for r in reads), weights=w, min_matches=mm, flags=flags, hints=hints4)
        assert np.array_equal(plain, wiped)
    got = lay.unpack(hinted, 0, L)
    items = [(p, int(c)) for p, c, r in zip(payloads(res, inc), w, rec) if c > 0 and round(100 * int(r["matches"]) / float(int(r["aln_len"])), 3) > 90.0]
    exp = aggregate.aggregate(items, L, ignore_substitutions=bool(flags & 1), ignore_insertions=bool(flags & 2),
                              ignore_deletions=bool(flags & 4), discard_indel_reads=bool(flags & 8))
    compare(got, exp, L)


def test_a_differing_column_that_is_no_base_leaves_no_hint():
    """a hint names a read base by (ch >> 1) & 7 -- A C G T N and nothing else: a read with an IUPAC code in a differing column must not leave one (the
    count pass then walks its strings, as it did); the tensor with the batch's hints equals the tensor without them"""
    E.build()
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(77)
    L = 200
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = list(range(L // 2 - 5, L // 2 + 5))
    reads = []
    for k in range(120):
        t = list(amp)
        p = int(rng.integers(5, L - 30))
        t[p] = "RYKMSWBDHV"[k % 10] if k % 3 else str(rng.choice(list("ACGT")))
        if k % 2: del t[L // 2 - 3:L // 2 + 4]                                          # ... with and without a deletion at the cut
        reads.append("".join(t))
    st = {"want_hints": True}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    assert (rec["status"] == 0).all()
    o1, o2 = st["raw"]
    h0 = st["hints"][:, 0]
    for k in range(120):
        iupac = any(c not in "ACGTN" for c in reads[k])
        if iupac:
            assert h0[k] == 0, (k, reads[k], hex(int(h0[k])))
    assert (h0 != 0).sum() >= 20
    w = rng.integers(1, 9, len(reads)).astype(np.uint32)
    plain, _ = E.count_vectors(o1, o2, rec, [amp], [inc], L, weights=w)
    hinted, _ = E.count_vectors(o1, o2, rec, [amp], [inc], L, weights=w, hints=st["hints"])
    assert np.array_equal(plain, hinted)


@pytest.mark.parametrize("flags", [0, C.FLAG_IGNORE_SUBSTITUTIONS | C.FLAG_IGNORE_DELETIONS])
def test_hints_with_several_references_each_task_against_its_own(flags):
    """CRISPRessoPooled's shape (BASELINE configs[4]): every read tagged with its amplicon.  The hinted kernel runs per reference over that reference's range of
    the order grouped by reference; the tensor of all three references equals the one without hints, and the reference's aggregation per amplicon."""
    E.build()
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(31337)
    amps = ["".join(rng.choice(list("ACGT"), L)) for L in (200, 250, 180)]
    gs, incs = [], []
    for a in amps:
        g = np.zeros(len(a) + 1, dtype=np.int64); g[len(a) // 2 + 1] = 1
        gs.append(g); incs.append(list(range(len(a) // 2 - 5, len(a) // 2 + 6)))
    reads, rids = [], []
    for k in range(600):
        r = int(rng.integers(0, 3))
        rd = _gapped_reads(rng, amps[r], 1)[0] if k % 3 else amps[r]
        if k % 7 == 0:
            t = list(amps[r]); t[int(rng.integers(0, len(t)))] = "N"; rd = "".join(t)
        reads.append(rd); rids.append(r)
    rids = np.array(rids, dtype=np.uint16)
    st = {"want_hints": True}
    res, rec = E.align_batch(reads, amps, gs, incs, m, -20, -2, ref_ids=rids, band_lanes=-87, stats=st)
    assert (rec["status"] == 0).all() and (rec["ref_id"] == rids).all()
    o1, o2 = st["raw"]
    hints4 = st["hints"]
    assert ((hints4[:, 0] >> 30) != 0).sum() > 300
    w = rng.integers(0, 20, len(reads)).astype(np.uint32)
    max_len = max(len(r) for r in reads)
    plain, lay = E.count_vectors(o1, o2, rec, amps, incs, max_len, weights=w, flags=flags)
    hinted, _ = E.count_vectors(o1, o2, rec, amps, incs, max_len, weights=w, flags=flags, hints=hints4)
    assert np.array_equal(plain, hinted), np.nonzero(plain != hinted)
    pls = payloads(res, [0])                                          # (per-read payloads need the read's own window: below)
    for r in range(3):
        items = []
        for k in range(len(reads)):
            if rids[k] == r and w[k] > 0:
                p = oracle.find_indels_substitutions(res[k][0], res[k][1], incs[r])
                p["aln_seq"], p["aln_ref"] = res[k]
                items.append((p, int(w[k])))
        exp = aggregate.aggregate(items, len(amps[r]), ignore_substitutions=bool(flags & 1), ignore_insertions=bool(flags & 2),
                                  ignore_deletions=bool(flags & 4), discard_indel_reads=bool(flags & 8))
        compare(lay.unpack(hinted, r, len(amps[r])), exp, len(amps[r]))
