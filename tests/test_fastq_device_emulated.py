"""The device ingest (crispresso2_amd/fastq_device.py: c2_fq_count / lines / dedup / gather kernels) on the wave emulator, against the
oracle's restatement of the reference's readline loop (oracle/fastq.py, CRISPRessoCORE.py:1825-1849) and against the host parser's line
statistics: same unique reads, same first-seen order, same multiplicities, for well-formed text, blank lines, whitespace around the
sequence, truncated tails, text without a final newline, text cut into several chunks and tiles."""
import ctypes
import contextlib
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import emu_driver as E
from oracle import fastq as O
from crispresso2_amd import fastq_device as FD, _native


@contextlib.contextmanager
def emulated_fq_kernels(chunk=FD.TILE):
    V, U = ctypes.c_void_p, ctypes.c_uint64
    lib = E.lib()

    def count(ctx, d_text, lo, hi, nl, em, flags, stream):
        assert lib.emu_fq_count(V(d_text), U(lo), U(hi), V(nl), V(em), V(flags)) == 0

    def lines(ctx, d_text, lo, hi, base, s, e, cap, stream):
        assert lib.emu_fq_lines(V(d_text), U(lo), U(hi), V(base), V(s), V(e), U(cap)) == 0

    def dedup(ctx, d_text, s, e, rng, cap, slots, n_slots, cnt, first, slot_of, rinfo, flags, stats, stream):
        assert lib.emu_fq_dedup(V(d_text), V(s), V(e), V(rng), U(cap), V(slots), U(n_slots), V(cnt), V(first), V(slot_of), V(rinfo), V(flags),
                                V(stats)) == 0

    def gather(ctx, d_text, info, records, out_off, out, n, stream):
        assert lib.emu_fq_gather(V(d_text), V(info), V(records or 0), V(out_off), V(out), U(n)) == 0
    def rc_partner(ctx, d_text, info, records, n, slots, n_slots, pslot, stream):
        assert lib.emu_fq_rc_partner(V(d_text), V(info), V(records), U(n), V(slots), U(n_slots), V(pslot)) == 0

    def lines4(ctx, d_text, lo, hi, base, s, e, qs, qe, cap, stream):
        assert lib.emu_fq_lines4(V(d_text), U(lo), U(hi), V(base), V(s), V(e), V(qs), V(qe), U(cap)) == 0

    def pair_lengths(ctx, t1, t2, lines1, lines2, n, s1, q1, s2, q2, klen, qlen, flags, stream):
        L1, L2 = (V * 4)(*[V(x) for x in lines1]), (V * 4)(*[V(x) for x in lines2])
        assert lib.emu_fq_pair_lengths(V(t1), V(t2), L1, L2, U(n), V(s1), V(q1), V(s2), V(q2), V(klen), V(qlen), V(flags)) == 0

    def pair_write(ctx, t1, t2, n, s1, q1, s2, q2, koff, qoff, kout, qout, flags, stream):
        assert lib.emu_fq_pair_write(V(t1), V(t2), U(n), V(s1), V(q1), V(s2), V(q2), V(koff), V(qoff), V(kout), V(qout), V(flags)) == 0
    saved = (FD.fq_count, FD.fq_lines, FD.fq_dedup, FD.fq_gather, FD.CHUNK_BYTES, torch.cuda.current_stream, FD.fq_rc_partner)
    saved_pair = (FD.fq_lines4, FD.fq_pair_lengths, FD.fq_pair_write)
    FD.fq_lines4, FD.fq_pair_lengths, FD.fq_pair_write = lines4, pair_lengths, pair_write
    FD.fq_rc_partner = rc_partner

    class _S:
        cuda_stream = 0
    FD.fq_count, FD.fq_lines, FD.fq_dedup, FD.fq_gather, FD.CHUNK_BYTES = count, lines, dedup, gather, chunk
    torch.cuda.current_stream = lambda *a, **k: _S()
    try:
        yield
    finally:
        FD.fq_count, FD.fq_lines, FD.fq_dedup, FD.fq_gather, FD.CHUNK_BYTES, torch.cuda.current_stream, FD.fq_rc_partner = saved
        FD.fq_lines4, FD.fq_pair_lengths, FD.fq_pair_write = saved_pair


def device_unique(path, chunk=FD.TILE):
    with emulated_fq_kernels(chunk):
        out = FD.ingest_file(str(path), None, torch.device("cpu"))
    arena, off = out["d_reads"].numpy(), out["offsets"]
    reads = [arena[int(off[i]):int(off[i + 1])].tobytes().decode("latin-1") for i in range(out["n_unique"])]
    return reads, out["counts"].tolist(), out


def check(path, chunk=FD.TILE):
    want, n_reads = O.read_fastq_unique(str(path))
    reads, counts, out = device_unique(path, chunk)
    n_empty = want.pop("", 0)
    assert reads == list(want.keys())
    assert counts == list(want.values())
    assert out["n_reads"] == n_reads and out["n_empty_records"] == n_empty
    st = {}
    with _native.FastqStream(str(path)) as fq:
        while not fq.done:
            fq.next()
        fq.line_stats(st)
    assert int(float(out["nonempty_lines"]) / 4.0) == st["N_READS_AFTER_PREPROCESSING"]
    return out


def records(rng, n, length=40, pool=None):
    pool = pool or ["".join(rng.choice("ACGTN") for _ in range(rng.randint(1, length))) for _ in range(max(2, n // 3))]
    return "".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(rng.choice(pool) for _ in range(n)))


def test_well_formed_text_over_several_tiles_and_chunks(tmp_path):
    rng = random.Random(5)
    p = tmp_path / "a.fastq"
    p.write_text(records(rng, 1500))
    assert os.path.getsize(p) > 4 * FD.TILE
    out = check(p)
    assert out["n_reads"] == 1500
    check(p, chunk=2 * FD.TILE)
    check(p, chunk=1 << 30)


@pytest.mark.parametrize("tail", ["", "@x", "@x\n", "@x\nACGT", "@x\nACGT\n", "@x\nACGT\n+", "@x\nACGT\n+\n", "@x\nACGT\n+\nIIII", "\n", "\n\n",
                                  "@x\n  ACGT \t\n+\nIIII\n", "@x\n\x0bAC GT\x1c\n+\nIIII\n\n\n@y\n\n"])
def test_tails_and_blank_lines(tmp_path, tail):
    rng = random.Random(9)
    p = tmp_path / "b.fastq"
    p.write_text(records(rng, 40, pool=["ACGT", "ACGTT", "GGGG"]) + tail)
    check(p)


def test_blank_lines_shift_the_records_as_in_the_reference(tmp_path):
    # a blank line in the middle: every later "record" is four lines from there on, whatever they hold
    p = tmp_path / "c.fastq"
    p.write_text("@a\nAAAA\n+\nIIII\n\n@b\nCCCC\n+\nIIII\n@c\nGGGG\n+\nIIII\n" * 30)
    check(p)


def test_long_and_short_sequences_and_a_tile_boundary_inside_a_line(tmp_path):
    rng = random.Random(2)
    big = "".join(rng.choice("ACGT") for _ in range(3 * FD.TILE + 77))
    p = tmp_path / "d.fastq"
    p.write_text("@l\n%s\n+\n%s\n" % (big, "I" * len(big)) + records(rng, 30, length=5) + "@l2\n%s\n+\nI\n" % big + "@l3\n%sA\n+\nI\n" % big[:-1])
    out = check(p)
    assert out["max_len"] == len(big)


def test_carriage_returns_go_to_the_host_parser(tmp_path):
    p = tmp_path / "e.fastq"
    p.write_text("@a\r\nAAAA\r\n+\r\nIIII\r\n" * 3)
    with emulated_fq_kernels():
        with pytest.raises(FD.DeviceIngestUnavailable):
            FD.ingest_file(str(p), None, torch.device("cpu"))


def test_more_records_than_estimated_goes_to_the_host_parser(tmp_path):
    # the first MB has long lines, the rest short ones: the estimate of the table is too small, the kernels say so
    p = tmp_path / "f.fastq"
    p.write_text("@a\n%s\n+\n%s\n" % ("A" * 400000, "I" * 400000) * 2 + "@b\nC\n+\nI\n" * 60000)
    with emulated_fq_kernels(1 << 30):
        with pytest.raises(FD.DeviceIngestUnavailable):
            FD.ingest_file(str(p), None, torch.device("cpu"))


def test_applicable():
    assert FD.applicable("/nonexistent.fastq") is not None
    assert FD.applicable(__file__, (30, 0, 0)).startswith("in memory:")      # (the host filters, the device can frame the filtered text)
    assert FD.applicable(__file__) == "small file"
    assert FD.size_applicable(0) is not None and FD.size_applicable(FD.MIN_TEXT_BYTES) is None and FD.text_applicable(None) is not None


def test_bgzf_members_inflated_chunk_by_chunk(tmp_path):
    """a BGZF file as the source: members inflated range by range into the text buffer (c2_bgzf_*), framed as they arrive; chunk
    boundaries are member boundaries, so the framing launches start at multiples of 16 and carry the bytes in between"""
    from crispresso2_amd import synth
    rng = random.Random(21)
    plain = tmp_path / "b.fastq"
    plain.write_text(records(rng, 2500) + "@cut\nACGTAC")
    bz = tmp_path / "b.fastq.gz"
    synth.write_bgzf(str(plain), str(bz), workers=2, level=1)
    want, n_reads = O.read_fastq_unique(str(plain))
    for chunk in (FD.TILE, 100_000, 1 << 30):
        with _native.BgzfFile(str(bz)) as bg, emulated_fq_kernels(chunk):
            assert bg.text_bytes == os.path.getsize(plain) and bg.n_blocks >= 3
            out = FD.ingest_file(bg, None, torch.device("cpu"))
        arena, off = out["d_reads"].numpy(), out["offsets"]
        reads = [arena[int(off[i]):int(off[i + 1])].tobytes().decode() for i in range(out["n_unique"])]
        assert reads == list(want.keys()) and out["counts"].tolist() == list(want.values()) and out["n_reads"] == n_reads
    with pytest.raises(_native.NativeError):
        _native.BgzfFile(str(plain))


def test_one_gzip_member_inflated_segment_by_segment(tmp_path, monkeypatch):
    """an ORDINARY one-member .gz as the source (c2_gzseg_open: segments at deflate block starts found by search, sizes and windows from a first
    decode): inflated range by range into the text buffer through the same c2_bgzf_* entries a BGZF file uses, framed as it arrives.  The
    member's CRC is checked when its last segment has been inflated; what is not one clean member is declined (the host inflates it as a whole)."""
    import gzip
    rng = random.Random(23)
    text = (records(rng, 12000) + "@cut\nACGTAC").encode()
    plain = tmp_path / "g.fastq"
    plain.write_bytes(text)
    gz = tmp_path / "g.fastq.gz"
    gz.write_bytes(gzip.compress(text, 6))
    want, n_reads = O.read_fastq_unique(str(plain))
    monkeypatch.setenv("C2_GZ_PARALLEL_MIN", "0")
    monkeypatch.setenv("C2_GZ_PARALLEL_CHUNK", "32768")
    for chunk in (FD.TILE, 100_000, 1 << 30):
        with _native.GzSegFile(str(gz), threads=4) as sg, emulated_fq_kernels(chunk):
            assert sg.text_bytes == len(text) and sg.n_blocks >= 3 and int(sg.text_offsets[0]) == 0
            whole = np.empty(len(text), dtype=np.uint8)
            sg.inflate(0, sg.n_blocks, whole.ctypes.data, whole.size, 3)
            assert whole.tobytes() == text
            mid = sg.n_blocks // 2                                       # any range, into any place
            part = np.empty(int(sg.text_offsets[mid + 1] - sg.text_offsets[mid]), dtype=np.uint8)
            sg.inflate(mid, mid + 1, part.ctypes.data, part.size, 1)
            assert part.tobytes() == text[int(sg.text_offsets[mid]):int(sg.text_offsets[mid + 1])]
            out = FD.ingest_file(sg, None, torch.device("cpu"))
        arena, off = out["d_reads"].numpy(), out["offsets"]
        reads = [arena[int(off[i]):int(off[i + 1])].tobytes().decode() for i in range(out["n_unique"])]
        assert reads == list(want.keys()) and out["counts"].tolist() == list(want.values()) and out["n_reads"] == n_reads
    # IngestSource picks it by itself for a .gz that is no BGZF file
    with FD.IngestSource(str(gz)) as src:
        assert isinstance(src.source, _native.GzSegFile) and "one gzip member" in src.route or src.why_not is not None
    # a wrong CRC shows when the last segment has been inflated
    bad = bytearray(gz.read_bytes())
    bad[-8] ^= 1
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(bad))
    with _native.GzSegFile(str(tmp_path / "bad.fastq.gz"), threads=4) as sg:
        buf = np.empty(len(text), dtype=np.uint8)
        sg.inflate(0, sg.n_blocks - 1, buf.ctypes.data, buf.size, 2)
        with pytest.raises(_native.NativeError, match="CRC check failed"):
            sg.inflate(sg.n_blocks - 1, sg.n_blocks, buf.ctypes.data, buf.size, 2)
    for name, data in (("two.fastq.gz", gzip.compress(text[:9000]) + gzip.compress(text[9000:])), ("plain.fastq", text),
                       ("padded.fastq.gz", gzip.compress(text) + b"\x00" * 32), ("cut.fastq.gz", gzip.compress(text)[:-20000])):
        (tmp_path / name).write_bytes(data)
        with pytest.raises(_native.NativeError, match="not applicable|cannot"):
            _native.GzSegFile(str(tmp_path / name), threads=4)
    monkeypatch.setenv("C2_GZ_PARALLEL", "0")
    with pytest.raises(_native.NativeError, match="not applicable"):
        _native.GzSegFile(str(gz), threads=4)


def test_reverse_complement_partners_from_the_table(tmp_path):
    """partner[i] = the unique read that equals reverse_complement(read i): the device's look-up in its own table against the host search
    (c2_rc_partners) -- pairs, palindromes (their own partner), lower case (upper-cased before complementing: the partner is the
    upper-case reverse complement, not the other way round), characters outside the alphabet (no partner), reads without a partner"""
    from crispresso2_amd import refs as RF
    rng = random.Random(11)
    base = ["".join(rng.choice("ACGT") for _ in range(rng.randint(20, 90))) for _ in range(200)]
    seqs = base + [RF.reverse_complement(s) for s in base[:80]] + ["ACGT", "AATT", "acgt", "ACGTNN", "NNACGT", "AC-GT_", "_AC-GT", "ACXGT", "ACYGT",
                                                                  "ggatcc", "GGATCC", base[5].lower(), "A" * 70, "T" * 70, "A" * 69]
    rng.shuffle(seqs)
    p = tmp_path / "rc.fastq"
    p.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(seqs + seqs[:50])))
    reads, counts, out = device_unique(p, chunk=FD.TILE)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    want = _native.rc_partners(arena, out["offsets"])
    assert np.array_equal(out["rc_partner"], want)
    assert (want >= 0).sum() >= 160 and (want == np.arange(len(want))).sum() >= 3
    from helpers import rc_partner_witness                             # ... and both against the definition written out in the test helpers
    assert np.array_equal(np.asarray(out["rc_partner"], dtype=np.int64), rc_partner_witness(reads))


def fuzz_text(rng, n_lines, alphabet="ACGTN acgt\t\x0b\x0c\x1c+@I#", max_len=60, blank=0.15):
    """lines of random bytes (white space of every kind str.strip() knows, empty lines, no FASTQ structure at all), random tail"""
    out = []
    for _ in range(n_lines):
        if rng.random() < blank:
            out.append("")
        else:
            out.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, max_len))))
    text = "\n".join(out)
    return text + rng.choice(["", "\n", "\n\n", " ", "\nAC", "\n \n"])


@pytest.mark.parametrize("seed", range(6))
def test_fuzzed_text_without_any_structure(tmp_path, seed):
    """the framing is "four lines are a record, whatever they hold": random lines, blank lines, white-space-only lines, every tail --
    the kernels = the reference's readline loop (oracle) on all of it"""
    rng = random.Random(1000 + seed)
    p = tmp_path / "fz.fastq"
    p.write_text(fuzz_text(rng, rng.randint(50, 2500)))
    want, n_reads = O.read_fastq_unique(str(p))
    reads, counts, out = device_unique(p, chunk=rng.choice([FD.TILE, 3 * FD.TILE, 1 << 30]))
    n_empty = want.pop("", 0)
    assert reads == list(want.keys()) and counts == list(want.values())
    assert out["n_reads"] == n_reads and out["n_empty_records"] == n_empty


# ---- paired input: fastq_device.ingest_pairs (c2_fq_lines4 / pair_lengths / pair_write / dedup kernels) vs the reference's lock-step loop ----
def device_pairs(p1, p2):
    """-> (keys, counts, first qualities, {key: [quality pairs of all its occurrences, in file order]}, n_records) from what the kernels left"""
    with emulated_fq_kernels(), FD.IngestSource(str(p1)) as S1, FD.IngestSource(str(p2)) as S2:
        assert S1.source is not None and S2.source is not None, (S1.why_not, S2.why_not)
        P = FD.ingest_pairs(S1.source, S2.source, None, torch.device("cpu"))
    ka, qa, ko, qo = P.d_keys.numpy(), P.d_quals.numpy(), P.key_off.numpy(), P.qual_off.numpy()
    key = lambda r: ka[int(ko[r]):int(ko[r + 1])].tobytes().decode("latin-1")
    qual = lambda r: qa[int(qo[r]):int(qo[r + 1])].tobytes().decode("latin-1")
    u = P.uniq_rec.numpy()
    keys = [key(int(r)) for r in u]
    for r in range(P.n_records):                                     # the '+' / the blank where l1 / lq1 say, every record's key index
        assert key(r)[int(P.l1[r])] == '+' and qual(r)[int(P.lq1[r])] == ' '
        assert keys[int(P.rec_key[r])] == key(r)
    occ = {}
    for r in range(P.n_records):
        occ.setdefault(key(r), []).append(qual(r))
    return keys, P.counts.tolist(), [qual(int(r)) for r in u], occ, P.n_records


def check_pairs_on_device(p1, p2):
    exp, n = O.read_paired_fastq_unique(str(p1), str(p2))
    keys, counts, quals, occ, n_rec = device_pairs(p1, p2)
    assert n_rec == n
    assert keys == list(exp.keys())
    assert counts == [v[0] for v in exp.values()]
    assert quals == [v[1] for v in exp.values()]
    wanted = {k for k, v in exp.items() if v[0] > 1}
    again = {}
    for k, q1, q2 in O.paired_occurrences(str(p1), str(p2), wanted):
        again.setdefault(k, []).append(q1 + ' ' + q2)
    assert {k: v for k, v in occ.items() if k in wanted} == again
    return n


def test_pairs_framed_keyed_and_deduplicated_on_the_device(tmp_path, monkeypatch):
    from test_fastq_ingest import random_pairs, paired_records
    monkeypatch.setenv("C2_FQ_INGEST", "device")                     # (files below MIN_TEXT_BYTES go to the host parser otherwise)
    rng = np.random.default_rng(31)
    pairs = random_pairs(1500, rng, pool=40)
    t1, t2 = paired_records(pairs)
    p1, p2 = tmp_path / "a_1.fastq", tmp_path / "a_2.fastq"
    p1.write_text(t1)
    p2.write_text(t2)
    assert check_pairs_on_device(p1, p2) == 1500
    # gzip'ed (one member: the host inflates, the device frames) and BGZF input, mixed with plain
    import gzip
    g1 = tmp_path / "a_1.fastq.gz"
    with gzip.open(g1, "wt") as fh:
        fh.write(t1)
    assert check_pairs_on_device(g1, p2) == 1500
    # file 2 shorter; cut in the middle of a record without a terminator; whitespace around the lines, blank id lines; an empty file
    lines2 = t2.splitlines(keepends=True)
    p2.write_text("".join(lines2[:4 * 150]))
    assert check_pairs_on_device(p1, p2) == 150
    p2.write_text("".join(lines2[:4 * 150 + 2]).rstrip("\n"))
    assert check_pairs_on_device(p1, p2) == 151
    p2.write_text("".join(lines2[:4 * 150 + 1]))                      # the last record of file 2 is an id line only
    assert check_pairs_on_device(p1, p2) == 151
    p1.write_text("@a\n  ACGT \t\n+\n IIII \n\n\x0bGGCC\x0c\n+\nJJJJ\n")
    p2.write_text("@a\n\tTTGA\n+\nABCD  \n\ncctt \n+\n EFGH\n")
    assert check_pairs_on_device(p1, p2) == 2
    keys, counts, quals, _, _ = device_pairs(p1, p2)
    assert keys == ["ACGT+TCAA", "GGCC+AAGG"] and quals == ["IIII DCBA", "JJJJ HGFE"]
    p1.write_text("")                                                 # an empty file is the host parser's
    with FD.IngestSource(str(p1)) as S1:
        assert S1.source is None and "0 bytes" in S1.why_not


def test_pairs_the_device_does_not_take_go_to_the_host_parser(tmp_path, monkeypatch):
    monkeypatch.setenv("C2_FQ_INGEST", "device")
    p1, p2 = tmp_path / "k_1.fastq", tmp_path / "k_2.fastq"
    p1.write_text("@a\nACGT\n+\nIIII\n")
    for bad2, why in (("@a\nACRT\n+\nIIII\n", "outside ACGTN_-"), ("@a\nAC+T\n+\nIIII\n", "outside ACGTN_-"), ("@a\nACGT\n+\nII I\n", "blank inside"),
                      ("@a\r\nACGT\r\n+\r\nIIII\r\n", "carriage returns")):
        p2.write_text(bad2)
        with emulated_fq_kernels(), FD.IngestSource(str(p1)) as S1, FD.IngestSource(str(p2)) as S2:
            with pytest.raises(FD.DeviceIngestUnavailable, match=why):
                FD.ingest_pairs(S1.source, S2.source, None, torch.device("cpu"))
    p2.write_text("@a\nACGT\n+\nIIII\n")
    p1.write_text("@a\nAC+T\n+\nIIII\n")
    with emulated_fq_kernels(), FD.IngestSource(str(p1)) as S1, FD.IngestSource(str(p2)) as S2:
        with pytest.raises(FD.DeviceIngestUnavailable, match="'\\+' inside a read"):
            FD.ingest_pairs(S1.source, S2.source, None, torch.device("cpu"))
