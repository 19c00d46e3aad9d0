"""c2_fastq_unique (native FASTQ ingest + exact de-duplication, host code) against oracle/fastq.py, the restatement of the
reference's readline loop (CRISPRessoCORE.py:1820-1849).  CPU only: the function needs the built library but no GPU."""
import gzip
import os

import numpy as np
import pytest

from oracle import fastq as ofq


def native(path):
    from crispresso2_amd import _native
    arena, offsets, counts, total = _native.fastq_unique(str(path))
    buf = arena.tobytes()
    seqs = [buf[int(offsets[k]):int(offsets[k + 1])].decode() for k in range(len(counts))]
    assert len(set(seqs)) == len(seqs)
    return dict(zip(seqs, (int(c) for c in counts))), list(seqs), total


def check(path):
    exp, n = ofq.read_fastq_unique(str(path))
    got, order, total = native(path)
    assert total == n
    assert got == exp
    assert order == list(exp.keys())                  # first-seen order = Python dict order


def records(seqs, nl="\n"):
    return "".join("@r%d%s%s%s+%s%s%s" % (k, nl, s, nl, nl, "I" * len(s), nl) for k, s in enumerate(seqs))


def random_seqs(n, rng, lo=20, hi=260, pool=200):
    base = ["".join(rng.choice(list("ACGTN"), int(rng.integers(lo, hi)))) for _ in range(pool)]
    return [base[int(rng.integers(0, pool))] for _ in range(n)]


def test_plain_gzip_and_duplicates(tmp_path):
    rng = np.random.default_rng(3)
    seqs = random_seqs(5000, rng)
    text = records(seqs)
    p = tmp_path / "a.fastq"
    p.write_text(text)
    check(p)
    g = tmp_path / "a.fastq.gz"
    with gzip.open(g, "wt") as fh:
        fh.write(text)
    check(g)
    # concatenated gzip members (bgzip-style output of many tools)
    g2 = tmp_path / "b.fastq.gz"
    half = text.index("@r2500\n")
    with open(g2, "wb") as fh:
        fh.write(gzip.compress(text[:half].encode()))
        fh.write(gzip.compress(text[half:].encode()))
    check(g2)


@pytest.mark.parametrize("nl", ["\r\n", "\r"])
def test_universal_newlines(tmp_path, nl):
    rng = np.random.default_rng(4)
    p = tmp_path / "crlf.fastq"
    p.write_bytes(records(random_seqs(700, rng), nl=nl).encode())
    check(p)


def test_whitespace_blank_lines_truncation_and_lowercase(tmp_path):
    cases = {
        "strip": "@a\n  ACGT\t \n+\nIIII\n@b\n\x0bACGT\x0c\n+\nIIII\n@c\nacgt\n+\nIIII\n",      # strip(), no upper-casing
        "no_final_newline": "@a\nACGT\n+\nIIII\n@b\nGGCC\n+\nIIII",
        "truncated_after_seq": "@a\nACGT\n+\nIIII\n@b\nGGCC",
        "truncated_after_id": "@a\nACGT\n+\nIIII\n@b\n",
        "truncated_id_no_newline": "@a\nACGT\n+\nIIII\n@b",
        "trailing_blank_line": "@a\nACGT\n+\nIIII\n\n",
        "blank_line_between": "@a\nACGT\n+\nIIII\n\n@b\nGGCC\n+\nIIII\n",                          # shifts the framing, as in the reference
        "empty_file": "",
        "only_newline": "\n",
        "interior_space": "@a\nAC GT\n+\nIIII\n",
        "cr_at_block_edge": "@a\r\nACGT\r\n+\r\nIIII\r",
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(text.encode())
        check(p)


@pytest.mark.parametrize("threads,range_bytes", [(1, 0), (2, 0), (3, 0), (7, 0), (16, 0), (64, 0), (1, 7), (2, 5), (3, 1), (5, 23), (16, 2), (4, 64)])
def test_parallel_ranges_edge_cases_and_random_files(tmp_path, monkeypatch, threads, range_bytes):
    """Plain files are parsed by byte ranges on several threads (line numbers from a terminator count); force many ranges
    onto small files so that range edges fall inside ids, sequences, terminators ("\\r|\\n"), blank lines and the tail.
    range_bytes > 0: the text is also cut into many CHUNKS of threads x range_bytes (the streaming engine: a chunk's ranges are
    counted, parsed and merged before the next chunk is read; line numbers, the per-thread tables and the global table carry over)."""
    monkeypatch.setenv("C2_FASTQ_THREADS", str(threads))
    if range_bytes:
        monkeypatch.setenv("C2_FASTQ_RANGE_BYTES", str(range_bytes))
    rng = np.random.default_rng(100 + threads)
    fixed = ["@a\nACGT\n+\nIIII\n@b\nGGCC\n+\nIIII", "@a\r\nACGT\r\n+\r\nIIII\r\n@b\r\nGG\r\n+\r\nII\r\n", "\n\n\n\n\n", "@a\n",
             "@a\nAC\n+\nII\n\n", "x", "\r", "\r\n", "@a\rACGT\r+\rIIII\r@b\rTTTT\r+\rIIII\r", "@a\n  AC GT \t\n+\nIIII\n@b\n\n+\n\n"]
    for k, text in enumerate(fixed):
        p = tmp_path / ("f%d.fastq" % k)
        p.write_bytes(text.encode())
        check(p)
    alphabet = ["A", "C", "G", "T", "N", " ", "@", "+", "\t", "a"]
    for k in range(60):
        lines = []
        for _ in range(int(rng.integers(0, 40))):
            lines.append("".join(rng.choice(alphabet, int(rng.integers(0, 12)))))
        nl = ["\n", "\r\n", "\r"][int(rng.integers(0, 3))] if k % 3 else None
        text = "".join(l + (nl or ["\n", "\r\n", "\r"][int(rng.integers(0, 3))]) for l in lines)
        if k % 4 == 0 and text:
            text = text.rstrip("\r\n")                           # unterminated tail
        p = tmp_path / ("r%d.fastq" % k)
        p.write_bytes(text.encode())
        check(p)
    big = records(random_seqs(3000 if not range_bytes or range_bytes >= 16 else 60, rng))      # (tiny chunks: a short file is enough)
    p = tmp_path / "big.fastq"
    p.write_text(big)
    check(p)


def test_long_lines_across_read_blocks(tmp_path):
    """Lines longer than, and straddling, the 4 MiB read block of the parser."""
    rng = np.random.default_rng(5)
    big = "".join(rng.choice(list("ACGT"), 5_000_000))
    seqs = [big, "ACGT" * 10, big, big[:-1]] + random_seqs(2000, rng)
    p = tmp_path / "long.fastq"
    p.write_text(records(seqs))
    check(p)


def bgzf_bytes(data, block=0xff00, level=6, eof=True):
    """BGZF as bgzip writes it: gzip members with a 'BC' extra subfield holding (member size - 1), <= 64 KiB of text each,
    and the empty end-of-file member."""
    import struct
    import zlib
    out = []
    chunks = [data[k:k + block] for k in range(0, len(data), block)] + ([b""] if eof else [])
    for c in chunks:
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = z.compress(c) + z.flush()
        size = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1)
                   + body + struct.pack("<II", zlib.crc32(c), len(c)))
    return b"".join(out)


GZ_ROUTES = ["auto", "zlib", "stream", "parallel"]


def set_gz_route(monkeypatch, route):
    """C2_FASTQ_GZ picks the route; "parallel" = auto with the one-member-on-all-threads route (c2_gz_parallel.h) forced onto files of any size,
    segments of 32 KiB of compressed data -- whatever it declines goes to the serial routes, so every expectation is the same"""
    if route == "parallel":
        monkeypatch.setenv("C2_FASTQ_GZ", "auto")
        monkeypatch.setenv("C2_GZ_PARALLEL_MIN", "0")
        monkeypatch.setenv("C2_GZ_PARALLEL_CHUNK", "32768")
        monkeypatch.setenv("C2_FASTQ_THREADS", "4")
    else:
        monkeypatch.setenv("C2_FASTQ_GZ", route)


@pytest.mark.parametrize("route", GZ_ROUTES)
def test_gzip_routes_bgzf_members_and_single_stream(tmp_path, monkeypatch, route):
    """.gz input takes one of three routes (whole-buffer BGZF on all threads, whole-buffer libdeflate, streaming zlib);
    every route must give the reference's dict for every kind of file, and files a route does not accept fall through."""
    set_gz_route(monkeypatch, route)
    rng = np.random.default_rng(11)
    text = records(random_seqs(6000, rng)).encode()
    crlf = records(random_seqs(900, rng), nl="\r\n").encode()
    files = {
        "bgzf.fastq.gz": bgzf_bytes(text),
        "bgzf_small_blocks.fastq.gz": bgzf_bytes(text, block=997),          # block edges inside ids, sequences, terminators
        "bgzf_no_eof.fastq.gz": bgzf_bytes(text, eof=False),
        "bgzf_crlf.fastq.gz": bgzf_bytes(crlf, block=1234),
        "bgzf_then_plain_member.fastq.gz": bgzf_bytes(text[:100000], eof=False) + gzip.compress(text[100000:]),
        "one_member.fastq.gz": gzip.compress(text, compresslevel=1),
        "three_members.fastq.gz": gzip.compress(text[:70001]) + gzip.compress(text[70001:70002]) + gzip.compress(text[70002:]),
        "empty_member_inside.fastq.gz": gzip.compress(text[:5000]) + gzip.compress(b"") + gzip.compress(text[5000:]),
        "empty_text.fastq.gz": gzip.compress(b""),
        "empty_bgzf.fastq.gz": bgzf_bytes(b""),
        "truncated_record.fastq.gz": gzip.compress(b"@a\nACGT\n+\nIIII\n@b\nGGCC"),
        "stored_blocks.fastq.gz": gzip.compress(text[:200000], compresslevel=0),
        "with_name_and_comment.fastq.gz": None,
    }
    import io
    bio = io.BytesIO()
    with gzip.GzipFile(filename="reads_with_a_name.fastq", mode="wb", fileobj=bio, mtime=12345) as fh:
        fh.write(text[:50000])
    files["with_name_and_comment.fastq.gz"] = bio.getvalue()
    for name, blob in files.items():
        p = tmp_path / name
        p.write_bytes(blob)
        check(p)


def test_gzip_whole_buffer_growth_and_memory_budget(tmp_path, monkeypatch):
    """A highly compressible member outgrows the first buffer estimate (4 x the file) and is inflated again into a larger
    one; with a budget smaller than the text the file goes through the streaming route.  Same result either way."""
    seq = "ACGT" * 60
    text = records([seq] * 20000 + ["TTTT" * 50] * 3).encode()
    two = gzip.compress(text, compresslevel=9) + gzip.compress(text[:len(records([seq] * 100))], compresslevel=9)
    assert 4 * len(two) < len(text)                            # the ISIZE hint (last member) and 4 x file are both too small
    p = tmp_path / "grow.fastq.gz"
    p.write_bytes(two)
    check(p)
    monkeypatch.setenv("C2_FASTQ_INFLATE_MAX", "100000")
    check(p)
    q = tmp_path / "bgzf_budget.fastq.gz"
    q.write_bytes(bgzf_bytes(text))
    check(q)


@pytest.mark.parametrize("route", GZ_ROUTES)
def test_gzip_damaged_or_padded_files_behave_like_pythons_gzip_module(tmp_path, monkeypatch, route):
    """Whatever the whole-buffer routes cannot take as a clean run of members is left to the streaming route from the start,
    which follows gzip.py: zero bytes after a member are skipped (more members may follow), anything else there is
    BadGzipFile, input ending inside a member is EOFError, a wrong checksum is BadGzipFile -- errors on every route."""
    from crispresso2_amd import _native
    set_gz_route(monkeypatch, route)
    rng = np.random.default_rng(12)
    text = records(random_seqs(3000, rng)).encode()
    half = text.index(b"@r1500\n")
    good = gzip.compress(text)
    accepted = {"zero_padded.fastq.gz": good + b"\x00" * 700,
                "bgzf_zero_padded.fastq.gz": bgzf_bytes(text) + b"\x00" * 512,
                "zeros_between_members.fastq.gz": gzip.compress(text[:half]) + b"\x00" * 13 + gzip.compress(text[half:]) + b"\x00"}
    for name, blob in accepted.items():
        p = tmp_path / name
        p.write_bytes(blob)
        check(p)
    bad_crc = bytearray(good)
    bad_crc[-6] ^= 0x55
    bgzf = bytearray(bgzf_bytes(text))
    bgzf[len(bgzf) // 2] ^= 0xff                                 # damage inside a block's deflate data
    rejected = {"cut_short.fastq.gz": (good[:len(good) // 2], EOFError), "bad_crc.fastq.gz": (bytes(bad_crc), gzip.BadGzipFile),
                "bgzf_damaged.fastq.gz": (bytes(bgzf), Exception), "bgzf_cut_short.fastq.gz": (bgzf_bytes(text)[:-40 - 28], EOFError),
                "trailing_garbage.fastq.gz": (good + b"not a gzip member", gzip.BadGzipFile),
                "trailing_half_magic.fastq.gz": (good + b"\x1f", gzip.BadGzipFile),
                "second_member_cut_in_header.fastq.gz": (good + good[:7], EOFError),
                "garbage_after_zeros.fastq.gz": (good + b"\x00\x00\x00junk", gzip.BadGzipFile)}
    for name, (blob, exc) in rejected.items():
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(exc):                                    # what the reference's readline loop dies of
            ofq.read_fastq_unique(str(p))
        with pytest.raises(_native.NativeError):
            _native.fastq_unique(str(p))


def test_missing_file_raises(tmp_path):
    from crispresso2_amd import _native
    with pytest.raises(_native.NativeError):
        _native.fastq_unique(str(tmp_path / "nope.fastq"))


def test_variants_read_fastq_unique_drops_only_the_empty_key(tmp_path):
    from crispresso2_amd import variants
    p = tmp_path / "t.fastq"
    p.write_text("@a\nACGT\n+\nIIII\n@b\nACGT\n+\nIIII\n@c\n")
    exp, _ = ofq.read_fastq_unique(str(p))
    assert "" in exp
    got = variants.read_fastq_unique(str(p))
    exp.pop("")
    assert got == exp and list(got) == list(exp)


def test_native_strand_plan_equals_the_seed_test(tmp_path):
    """c2_strand_plan against the seed test of get_new_variant_object (CRISPRessoCORE.py:656-687) as restated in
    crispresso2_amd.variants._strand_plan (which the variant goldens pin to the reference's function)."""
    from types import SimpleNamespace
    from crispresso2_amd import _native, variants, refs as RF
    rng = np.random.default_rng(21)
    amp = "".join(rng.choice(list("ACGT"), 230))
    ref = RF.make_ref("R", amp, [110], [110, 111])
    reads = []
    for k in range(6000):
        s = list(amp)
        for _ in range(int(rng.integers(0, 12))):
            s[int(rng.integers(0, len(s)))] = "ACGTN"[int(rng.integers(0, 5))]
        s = "".join(s[int(rng.integers(0, 60)):len(s) - int(rng.integers(0, 60))])
        mode = k % 5
        if mode == 1:
            s = RF.reverse_complement(s)
        elif mode == 2:
            s = s[:len(s) // 2] + RF.reverse_complement(s[len(s) // 2:])        # seeds of both strands
        elif mode == 3:
            s = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 40))))      # nothing found
        reads.append(s)
    reads += ["", "A", ref["fw_seeds"][0], ref["rc_seeds"][0]]
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    for seed_count, seed_min in ((5, 2), (3, 0), (50, 1), (0, 2)):
        args = SimpleNamespace(aln_seed_count=seed_count, aln_seed_min=seed_min)
        m = min(seed_count, len(ref["fw_seeds"]))
        got = _native.strand_plan(arena, off, ref["fw_seeds"][:m], ref["rc_seeds"][:m], seed_min)
        exp = np.array([variants._strand_plan(args, s, ref) for s in reads], dtype=np.uint8)
        assert np.array_equal(got, exp)
        if seed_count == 5:
            assert set(np.unique(exp)) == {0, 1, 2}


@pytest.mark.parametrize("n_base", [400, 12000])
def test_native_reverse_complement_merge_equals_the_aggregation_loop(n_base):
    """c2_merge_reverse_complements against the statements of CRISPRessoCORE.py:3970-3975 on a dict, in cache order;
    palindromes (counted twice by the reference), lower case, reads with characters reverse_complement() cannot map.
    The larger case runs the partner search on several threads."""
    from crispresso2_amd import _native, refs as RF
    rng = np.random.default_rng(22)
    base = ["".join(rng.choice(list("ACGTN"), int(rng.integers(4, 30)))) for _ in range(n_base)]
    reads = list(dict.fromkeys(base + [RF.reverse_complement(b) for b in base[::2]] + ["ACGT", "AATT", "acgt", "AC-GT_N", "ACXGT", "GGCC"]))
    rng.shuffle(reads)
    counts = rng.integers(1, 50, len(reads)).astype(np.int64)
    counts[::13] = 0
    aligned = rng.random(len(reads)) < 0.8
    # the reference's loop (variantCache holds the aligned reads only)
    cache = {r: int(c) for r, c, a in zip(reads, counts, aligned) if a}
    for variant in cache:
        variant_count = cache[variant]
        if variant_count == 0:
            continue
        try:
            rc_variant = RF.reverse_complement(variant)
        except KeyError:
            continue
        if rc_variant in cache and cache[rc_variant] > 0:
            variant_count += cache[rc_variant]
            cache[rc_variant] = 0
            cache[variant] = variant_count
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    got = counts.copy()
    _native.merge_reverse_complements(arena, off, aligned, got)
    for k, r in enumerate(reads):
        assert got[k] == (cache[r] if aligned[k] else counts[k]), r
    assert got[reads.index("ACGT")] in (0, 2 * counts[reads.index("ACGT")]) or not aligned[reads.index("ACGT")]


# ---- paired input: c2_fastq_unique_paired / c2_fastq_paired_occurrences vs the reference's lockstep loop -------------

def check_paired(p1, p2):
    from crispresso2_amd import _native
    exp, n = ofq.read_paired_fastq_unique(str(p1), str(p2))
    pf = _native.PairedFastq(str(p1), str(p2))
    assert pf.n_pairs == n
    assert pf.keys == list(exp.keys())
    assert [int(c) for c in pf.counts] == [v[0] for v in exp.values()]
    assert pf.quals == [v[1] for v in exp.values()]
    # second pass: every occurrence of the keys seen more than once
    sel = pf.counts > 1
    idx, quals = pf.occurrences(sel)
    wanted = {k for k, v in exp.items() if v[0] > 1}
    occ = ofq.paired_occurrences(str(p1), str(p2), wanted)
    assert [pf.keys[int(i)] for i in idx] == [o[0] for o in occ]
    assert quals == [o[1] + ' ' + o[2] for o in occ]
    pf.close()
    return pf


def paired_records(pairs, nl="\n"):
    r1 = "".join("@p%d/1%s%s%s+%s%s%s" % (k, nl, p[0], nl, nl, p[2], nl) for k, p in enumerate(pairs))
    r2 = "".join("@p%d/2%s%s%s+%s%s%s" % (k, nl, p[1], nl, nl, p[3], nl) for k, p in enumerate(pairs))
    return r1, r2


def random_pairs(n, rng, pool=60):
    base = []
    for _ in range(pool):
        a, b = int(rng.integers(30, 150)), int(rng.integers(30, 150))
        base.append(("".join(rng.choice(list("ACGTN"), a)), "".join(rng.choice(list("ACGTNacgt"), b))))
    out = []
    for _ in range(n):
        s1, s2 = base[int(rng.integers(0, pool))]
        q = lambda m: "".join(chr(int(x)) for x in rng.integers(33, 74, m))
        out.append((s1, s2, q(len(s1)), q(len(s2))))
    return out


def test_paired_plain_gzip_crlf(tmp_path):
    rng = np.random.default_rng(21)
    pairs = random_pairs(3000, rng)
    for tag, nl in (("lf", "\n"), ("crlf", "\r\n"), ("cr", "\r")):
        t1, t2 = paired_records(pairs, nl)
        p1, p2 = tmp_path / (tag + "_1.fastq"), tmp_path / (tag + "_2.fastq")
        p1.write_bytes(t1.encode())
        p2.write_bytes(t2.encode())
        pf = check_paired(p1, p2)
        assert len(pf.keys) <= 60 and pf.n_pairs == 3000
    t1, t2 = paired_records(pairs)
    g1, g2 = tmp_path / "a_1.fastq.gz", tmp_path / "a_2.fastq.gz"
    with gzip.open(g1, "wt") as fh:
        fh.write(t1)
    with gzip.open(g2, "wt") as fh:
        fh.write(t2)
    check_paired(g1, g2)
    check_paired(g1, tmp_path / "lf_2.fastq")                       # one gzip'ed, one plain


def test_paired_unequal_lengths_whitespace_and_truncation(tmp_path):
    rng = np.random.default_rng(22)
    pairs = random_pairs(200, rng, pool=15)
    t1, t2 = paired_records(pairs)
    p1, p2 = tmp_path / "x_1.fastq", tmp_path / "x_2.fastq"
    # file 2 shorter: the loop stops with the shorter file
    p1.write_text(t1)
    p2.write_text("".join(t2.splitlines(keepends=True)[:4 * 150]))
    assert check_paired(p1, p2).n_pairs == 150
    # file 2 cut in the middle of a record (no quality line, no terminator): '' qualities, the record still counts
    p2.write_text("".join(t2.splitlines(keepends=True)[:4 * 150 + 2]).rstrip("\n"))
    assert check_paired(p1, p2).n_pairs == 151
    # whitespace around sequences / qualities, blank id lines ('\n' is a true value), empty files
    p1.write_text("@a\n  ACGT \t\n+\n IIII \n\n\x0bGGCC\x0c\n+\nJJJJ\n")
    p2.write_text("@a\n\tTTGA\n+\nABCD  \n\ncctt \n+\n EFGH\n")
    pf = check_paired(p1, p2)
    assert pf.keys == ["ACGT+TCAA", "GGCC+AAGG"] and pf.quals == ["IIII DCBA", "JJJJ HGFE"]
    p1.write_text("")
    assert check_paired(p1, p2).n_pairs == 0


def test_paired_bad_base_is_a_key_error(tmp_path):
    from crispresso2_amd import _native
    p1, p2 = tmp_path / "k_1.fastq", tmp_path / "k_2.fastq"
    p1.write_text("@a\nACGT\n+\nIIII\n")
    p2.write_text("@a\nACRT\n+\nIIII\n")                             # R: not in the reference's complement table
    with pytest.raises(KeyError):
        ofq.read_paired_fastq_unique(str(p1), str(p2))
    with pytest.raises(KeyError):
        _native.PairedFastq(str(p1), str(p2))


def test_paired_gzip_cut_short_is_an_error_like_pythons_eoferror(tmp_path):
    """gzip.open(...) raises EOFError when a stream stops before its end-of-stream marker; zlib's gzread only sets
    Z_BUF_ERROR and returns what it has -- the native reader turns that into an error for either file."""
    from crispresso2_amd import _native
    rng = np.random.default_rng(21)
    pairs = random_pairs(400, rng)
    t1, t2 = paired_records(pairs)
    g1, g2 = gzip.compress(t1.encode()), gzip.compress(t2.encode())
    for k, (b1, b2) in enumerate([(g1[:len(g1) // 2], g2), (g1, g2[:len(g2) // 2])]):
        p1, p2 = tmp_path / ("t%d_1.fastq.gz" % k), tmp_path / ("t%d_2.fastq.gz" % k)
        p1.write_bytes(b1)
        p2.write_bytes(b2)
        with pytest.raises(EOFError):
            ofq.read_paired_fastq_unique(str(p1), str(p2))
        with pytest.raises(_native.NativeError):
            _native.PairedFastq(str(p1), str(p2))


# ---- the read filter of filterFastqs.py, fused into the native ingest ------------------------------------------------------
def native_filtered(path, mbp, mrq, mbpn):
    from crispresso2_amd import _native
    st = {}
    arena, offsets, counts, total = _native.fastq_unique(str(path), min_single_bp_quality=mbp, min_average_read_quality=mrq,
                                                         min_bp_quality_or_N=mbpn, stats=st)
    buf = arena.tobytes()
    seqs = [buf[int(offsets[k]):int(offsets[k + 1])].decode() for k in range(len(counts))]
    return dict(zip(seqs, (int(c) for c in counts))), seqs, total, st["N_READS_INPUT"]


def check_filtered(path, tmp_path, mbp, mrq, mbpn):
    """the oracle writes the reference's intermediate file and reads it back with the reference's loop; the native route
    must give the same dict, order and counts from the ORIGINAL file in one pass"""
    mid = tmp_path / "oracle_filtered.fastq"
    n_in = ofq.filter_fastq(str(path), str(mid), mbp or None, mrq or None, mbpn or None)
    exp, n = ofq.read_fastq_unique(str(mid))
    got, order, total, n_input = native_filtered(path, mbp, mrq, mbpn)
    assert (total, n_input) == (n, n_in)
    assert got == exp and order == list(exp.keys())
    return n


def test_read_filter_reproduces_the_filtered_file_of_the_reference_params_run(tmp_path):
    """FANC.Cas9.fastq with -q 30 (CRISPResso_on_params): the oracle's filtered text is the content of the reference's own
    FANC.Cas9_filtered.fastq.gz (pins the restatement), and the fused native route sees exactly those reads."""
    from helpers import load_golden
    raw = tmp_path / "FANC.Cas9.fastq"
    raw.write_text(load_golden("fanc_run.json.gz")["fastq"])
    want = load_golden("params_run.json.gz")
    mid = tmp_path / "f.fastq"
    assert ofq.filter_fastq(str(raw), str(mid), None, 30, None) == want["alignment_stats"]["N_READS_INPUT"] == 250
    assert mid.read_text() == want["fastq_after_quality_filter"]
    assert check_filtered(raw, tmp_path, 0, 30, 0) == want["alignment_stats"]["N_READS_AFTER_PREPROCESSING"] == 231
    g = tmp_path / "FANC.Cas9.fastq.gz"
    with gzip.open(g, "wb") as fh:
        fh.write(raw.read_bytes())
    assert check_filtered(g, tmp_path, 0, 30, 0) == 231


@pytest.mark.parametrize("opts", [(0, 25, 0), (12, 0, 0), (0, 0, 15), (10, 22, 0), (0, 24, 14), (8, 20, 12)])
@pytest.mark.parametrize("route", ["auto", "stream"])
def test_read_filter_option_combinations_random_files(tmp_path, monkeypatch, opts, route):
    set_gz_route(monkeypatch, route)
    rng = np.random.default_rng(sum(opts))
    seqs = random_seqs(1500, rng, lo=15, hi=120, pool=40)
    recs = []
    for k, s in enumerate(seqs):
        q = rng.integers(33 + int(rng.integers(0, 30)), 75, len(s))
        if k % 17 == 0:
            q[int(rng.integers(0, len(s)))] = 33 + int(rng.integers(0, 10))        # one bad base
        pad = ["", " ", "\t ", "\r"][k % 4]                                        # rstrip() takes these off every line
        recs.append("@r%d%s\n%s%s\n+%s\n%s%s\n" % (k, pad, s, pad, pad, "".join(chr(int(x)) for x in q), pad))
    text = "".join(recs)
    p = tmp_path / "q.fastq"
    p.write_bytes(text.encode())
    n = check_filtered(p, tmp_path, *opts)
    assert 0 < n < len(seqs) or opts == (0, 0, 15)
    g = tmp_path / "q.fastq.gz"
    g.write_bytes(bgzf_bytes(text.encode(), block=5000) if sum(opts) % 2 else gzip.compress(text.encode()))
    assert check_filtered(g, tmp_path, *opts) == n


def test_read_filter_framing_wraparound_and_the_reference_failures(tmp_path):
    from crispresso2_amd import _native
    cases = {
        "blank_id_line_ends_the_file": ("@a\nACGT\n+\nIIII\n\n@b\nGGCC\n+\nIIII\n", (0, 20, 0)),
        "truncated_last_record": ("@a\nACGT\n+\nIIII\n@b\nGGCC\n", (0, 20, 0)),             # empty quality line: mean is nan -> dropped
        "no_final_newline": ("@a\nACGT\n+\nIIII\n@b\nGGCC\n+\nIII5", (0, 20, 15)),
        "qualities_below_33_wrap": ("@a\nACGT\n+\n I!I\n@b\nGGCC\n+\nIIII\n", (0, 60, 0)),       # ' ' - 33 = 255 in uint8
        "wrap_passes_the_minimum": ("@a\nACGT\n+\n  \x1f \n", (200, 0, 0)),
        "crlf": ("@a\r\nACGT\r\n+\r\nII#I\r\n@b\r\nGGCC\r\n+\r\nIIII\r\n", (0, 0, 20)),
        "cr_only_is_one_line": ("@a\rACGT\r+\rIIII\r", (0, 10, 0)),
        "nothing_passes": ("@a\nACGT\n+\n####\n", (0, 30, 0)),
        "lowercase_masked": ("@a\nacgt\n+\nI#I#\n", (0, 0, 10)),
        # numpy takes a boolean index of size 0 for any array: an EMPTY quality line under masking alone is written unmasked
        "empty_quality_under_masking_alone": ("@a\nACGT\n+\nI#II\n@b\nGGCC\n+\n\n@c\nTTAA\n+\n#III\n", (0, 0, 10)),
        "record_cut_after_the_plus_line_under_masking": ("@a\nACGT\n+\nI#II\n@b\nGGCC\n+\n", (0, 0, 10)),
    }
    for name, (text, opts) in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(text.encode("latin-1"))
        check_filtered(p, tmp_path, *opts)
    failing = {
        "empty_quality_under_minimum": ("@a\nACGT\n+\n\n", (5, 0, 0), ValueError),
        "length_mismatch_under_masking": ("@a\nACGT\n+\nIII\n", (0, 0, 10), IndexError),
        "read_only_view_of_run_mBP_mBPN": ("@a\nACGT\n+\nIIII\n", (5, 0, 10), ValueError),
    }
    for name, (text, opts, exc) in failing.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(text.encode())
        with pytest.raises(exc):
            ofq.filter_fastq(str(p), str(tmp_path / "o.fastq"), opts[0] or None, opts[1] or None, opts[2] or None)
        with pytest.raises(_native.NativeError):
            native_filtered(p, *opts)
    # the read-only failure needs a read that reaches the masking step: none does here, and neither side fails
    p = tmp_path / "mBP_mBPN_nothing_passes.fastq"
    p.write_bytes(b"@a\nACGT\n+\n#III\n")
    check_filtered(p, tmp_path, 5, 0, 10)


@pytest.mark.parametrize("threads", [2, 3, 7, 16, 61])
def test_read_filter_on_line_ranges(tmp_path, monkeypatch, threads):
    """The filter cuts the text into byte ranges (one thread each; record framing from a count of '\\n'); force many ranges onto
    small files: range edges inside any of the four lines, a blank id line (everything after it is dropped, also what later
    ranges produced), failures before and after such a stop (only the former count), CRLF, no final newline."""
    from crispresso2_amd import _native
    monkeypatch.setenv("C2_FASTQ_THREADS", str(threads))
    rng = np.random.default_rng(300 + threads)
    for trial in range(12):
        seqs = random_seqs(int(rng.integers(1, 90)), rng, lo=5, hi=60, pool=15)
        recs = []
        for k, s in enumerate(seqs):
            q = "".join(chr(int(x)) for x in rng.integers(33, 75, len(s)))
            nl = "\r\n" if trial % 3 == 2 else "\n"
            recs.append("@r%d%s%s%s+%s%s%s" % (k, nl, s, nl, nl, q, nl))
        if trial % 4 == 1 and len(recs) > 3:
            recs.insert(int(rng.integers(1, len(recs))), "\n")             # a blank id line: the reference's loop ends there
        text = "".join(recs)
        if trial % 5 == 3:
            text = text.rstrip("\r\n")
        p = tmp_path / ("t%d.fastq" % trial)
        p.write_bytes(text.encode())
        for opts in ((0, 30, 0), (5, 0, 0), (0, 20, 12), (4, 18, 9)):
            check_filtered(p, tmp_path, *opts)
    # a failure AFTER the blank id line is never reached; one BEFORE it is
    good = "@a\nACGT\n+\nIIII\n" * 9
    bad = "@b\nACGT\n+\nIII\n"                                        # sequence / quality length mismatch under masking
    p = tmp_path / "after_stop.fastq"
    p.write_bytes((good + "\n" + bad + good).encode())
    check_filtered(p, tmp_path, 0, 0, 10)
    p = tmp_path / "before_stop.fastq"
    p.write_bytes((good + bad + good + "\n" + good).encode())
    with pytest.raises(IndexError):
        ofq.filter_fastq(str(p), str(tmp_path / "o.fastq"), None, None, 10)
    with pytest.raises(_native.NativeError):
        native_filtered(p, 0, 0, 10)


@pytest.mark.parametrize("n_base,threads", [(400, None), (12000, "5"), (12000, "1")])
def test_native_partner_search_plus_count_transfer_equals_the_one_step_merge(n_base, threads, monkeypatch):
    """c2_rc_partners (parallel inserts into one table, lookup among ALL reads) + c2_merge_counts_with_partners (skips partners that
    are not aligned) = c2_merge_reverse_complements -- the pipeline runs the first on a thread while the device aligns."""
    from crispresso2_amd import _native, refs as RF
    if threads:
        monkeypatch.setenv("C2_HOST_THREADS", threads)
    rng = np.random.default_rng(23 + n_base)
    base = ["".join(rng.choice(list("ACGTN"), int(rng.integers(4, 30)))) for _ in range(n_base)]
    reads = list(dict.fromkeys(base + [RF.reverse_complement(b) for b in base[::2]] + ["ACGT", "AATT", "acgt", "AC-GT_N", "ACXGT", "GGCC", ""]))
    rng.shuffle(reads)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    partner = _native.rc_partners(arena, off)
    index = {r: k for k, r in enumerate(reads)}
    for k, r in enumerate(reads):
        try:
            exp = index.get(RF.reverse_complement(r), -1)
        except KeyError:
            exp = -1
        assert partner[k] == exp, r
    for seed in range(3):
        r2 = np.random.default_rng(seed)
        counts = r2.integers(0, 50, len(reads)).astype(np.int64)
        aligned = r2.random(len(reads)) < 0.7
        one = _native.merge_reverse_complements(arena, off, aligned, counts.copy())
        two = _native.merge_counts_with_partners(aligned, partner, counts.copy())
        assert np.array_equal(one, two)


def test_nonempty_line_count_is_what_get_n_reads_fastq_counts(tmp_path, monkeypatch):
    """c2_fastq_nonempty_lines = `grep -c .` of the parsed text (CRISPRessoShared.py:743-748) on every ingest route; the read counts
    the reference reports are int(that / 4.0) -- different from the number of records for blank lines and truncated tails."""
    import subprocess
    from crispresso2_amd import _native
    texts = {
        "plain": "@a\nACGT\n+\nIIII\n@b\nGGCC\n+\nIIII\n",
        "blank_lines": "@a\nACGT\n+\nIIII\n\n\n@b\nGGCC\n+\nIIII\n\n",
        "truncated": "@a\nACGT\n+\nIIII\n@b\nGG",
        "crlf": "@a\r\nACGT\r\n+\r\nIIII\r\n\r\n@b\r\nGGCC\r\n+\r\nIIII\r\n",       # a line holding only '\r' is not empty for grep
        "cr_only": "@a\rACGT\r+\rIIII\r",
        "empty_sequence": "@a\n\n+\n\n@b\nGGCC\n+\nIIII\n",
    }
    rng = np.random.default_rng(3)
    big = "".join("@r%d\n%s\n+\n%s\n%s" % (k, "".join(rng.choice(list("ACGT"), 30)), "I" * 30, "\n" if k % 97 == 0 else "") for k in range(4000))
    texts["big_with_blank_lines"] = big
    for name, text in texts.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(text.encode())
        want = int(subprocess.run("cat < %s | grep -c ." % p, shell=True, capture_output=True, text=True).stdout or 0)
        gz = tmp_path / (name + ".fastq.gz")
        with gzip.open(gz, "wb") as fh:
            fh.write(text.encode())
        for path, envs in ((p, [{}, {"C2_FASTQ_THREADS": "5"}]), (gz, [{"C2_FASTQ_GZ": "stream"}, {"C2_FASTQ_GZ": "auto"}])):
            for env in envs:
                for k_, v_ in env.items():
                    monkeypatch.setenv(k_, v_)
                st = {}
                _native.fastq_unique(str(path), stats=st)
                for k_ in env:
                    monkeypatch.delenv(k_)
                assert st["N_READS_AFTER_PREPROCESSING"] == st["N_READS_INPUT"] == int(float(want) / 4.0), (name, path, env, st, want)


@pytest.mark.parametrize("threads,range_bytes", [(1, 300), (3, 170), (8, 64), (8, 4096), (32, 100)])
def test_streamed_chunks_keep_first_seen_order_and_counts_across_chunks(tmp_path, monkeypatch, threads, range_bytes):
    """Many chunks over one file with heavy duplication ACROSS chunks (a few hot sequences everywhere, sequences that come back
    after thousands of records, a read that only ever appears as the last record): unique reads in global first-seen order,
    exact multiplicities, arena bytes -- the plain file through pread() and the same text through the in-memory route (.gz)."""
    monkeypatch.setenv("C2_FASTQ_THREADS", str(threads))
    monkeypatch.setenv("C2_FASTQ_RANGE_BYTES", str(range_bytes))
    rng = np.random.default_rng(threads * 1000 + range_bytes)
    pool = ["".join(rng.choice(list("ACGTN"), int(rng.integers(1, 70)))) for _ in range(400)]
    hot = pool[:5]
    seqs = []
    for k in range(6000):
        r = rng.random()
        seqs.append(hot[int(rng.integers(0, 5))] if r < 0.4 else pool[int(rng.integers(0, 400))] if r < 0.9 else
                    "".join(rng.choice(list("ACGT"), int(rng.integers(1, 90)))))
    seqs.append("TTTTGGGGCCCCAAAA" * 3)
    p = tmp_path / "many_chunks.fastq"
    p.write_text(records(seqs))
    check(p)
    gz = tmp_path / "many_chunks.fastq.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(records(seqs).encode())
    check(gz)


@pytest.mark.parametrize("threads,range_bytes", [(1, 0), (4, 512), (16, 0)])
def test_stream_rc_partners_equal_the_standalone_search(tmp_path, monkeypatch, threads, range_bytes):
    """c2_fastq_stream_rc_partners answers "which unique read is reverse_complement(read i)" from the table the ingest built; it must
    equal c2_rc_partners (its own table over the same reads): pairs in both orders, a read that is its own reverse complement,
    lower-case reads (upper-cased before complementing), characters outside ACGTN_- (no partner), reads without a partner."""
    from crispresso2_amd import _native
    monkeypatch.setenv("C2_FASTQ_THREADS", str(threads))
    if range_bytes:
        monkeypatch.setenv("C2_FASTQ_RANGE_BYTES", str(range_bytes))
    rng = np.random.default_rng(77 + threads)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "_": "_", "-": "-"}
    rc = lambda s_: "".join(comp[c] for c in reversed(s_.upper()))
    base = ["".join(rng.choice(list("ACGTN"), int(rng.integers(4, 80)))) for _ in range(300)]
    seqs = []
    for k, b in enumerate(base):
        seqs.append(b)
        if k % 3 == 0:
            seqs.append(rc(b))
        if k % 7 == 0:
            seqs.append(b.lower())
        if k % 11 == 0:
            seqs.append(b[:5] + "R" + b[5:])
    seqs += ["ACGT", "AATT", "GGCC_-", "acgt"]                          # palindromes: their own partners
    order = rng.permutation(len(seqs))
    p = tmp_path / "rc.fastq"
    p.write_text(records([seqs[k] for k in order] * 2))
    with _native.FastqStream(str(p)) as fq:
        while not fq.done:
            fq.next()
        got = fq.rc_partners()
        off = fq.offsets_slice(0, fq.n_unique)
        arena = fq.arena[:fq.arena_bytes].copy()
    want = _native.rc_partners(arena, off)
    assert np.array_equal(got, want)
    assert (got >= 0).sum() > 150 and (got == np.arange(len(got))).sum() >= 2 and (got < 0).sum() > 50
