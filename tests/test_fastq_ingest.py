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


@pytest.mark.parametrize("threads", [1, 2, 3, 7, 16, 64])
def test_parallel_ranges_edge_cases_and_random_files(tmp_path, monkeypatch, threads):
    """Plain files are parsed by byte ranges on several threads (line numbers from a terminator count); force many ranges
    onto small files so that range edges fall inside ids, sequences, terminators ("\\r|\\n"), blank lines and the tail."""
    monkeypatch.setenv("C2_FASTQ_THREADS", str(threads))
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
    big = records(random_seqs(3000, rng))
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
