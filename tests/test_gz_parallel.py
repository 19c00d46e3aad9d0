"""One ordinary gzip member inflated on all host threads (c2_gz_parallel.h, c2_gz_inflate_parallel): what gzip.open(fastq, 'rt') does on one
thread (CRISPResso2/CRISPRessoCORE.py:1820-1823).  The witness is Python's own gzip / zlib module on the same bytes.  The contract: rc 0 ->
the text is byte-identical; anything the route is not certain about -> C2_E_INVALID (the file routes then inflate serially, and the accepted
inputs / errors are the serial route's: tests/test_fastq_ingest.py runs its whole .gz table with this route forced on, route "parallel")."""
import ctypes
import gzip
import io
import os
import zlib

import numpy as np
import pytest

from crispresso2_amd import _native

C2_E_INVALID, C2_E_OVERFLOW = -1, -6


def inflate_parallel(gz, threads=4, chunk=32768, cap=None):
    lib = _native.load()
    src = np.frombuffer(bytes(gz), dtype=np.uint8)
    room = (1 << 16) if cap is None else cap
    buf = np.full(max(room, 1), 0xAB, dtype=np.uint8)
    n_out = ctypes.c_uint64(0)
    stats = (ctypes.c_uint64 * 8)()
    rc = lib.c2_gz_inflate_parallel(src.ctypes.data, src.size, buf.ctypes.data, room, ctypes.byref(n_out), threads, chunk, stats)
    if rc == C2_E_OVERFLOW and cap is None:                       # (first call with a small buffer: the size comes back)
        return inflate_parallel(gz, threads, chunk, cap=int(n_out.value))
    return rc, (buf[:n_out.value].tobytes() if rc == 0 else None), dict(segments=stats[0], found=stats[1], bytes=stats[2], fell_back=stats[3])


def fastq_text(n, L=100, seed=0, constant_quality=False):
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    amp = bases[rng.integers(0, 4, L)]
    out = []
    for i in range(n):
        r = amp.copy()
        k = int(rng.integers(0, 4))
        if k:
            r[rng.integers(0, L, k)] = bases[rng.integers(0, 4, k)]
        q = np.full(L, 73, np.uint8) if (constant_quality or i % 3 == 0) else rng.integers(33, 74, L).astype(np.uint8)
        out.append(b"@read%07d\n" % i + r.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("chunk", [32768, 50001, 1 << 18])
def test_identical_to_the_gzip_module(level, chunk):
    text = fastq_text(30000, seed=level)
    gz = gzip.compress(text, level)
    for threads in (2, 3, 8):
        rc, got, st = inflate_parallel(gz, threads, chunk)
        assert rc == 0, st
        assert got == text
        assert st["segments"] >= 2 and st["bytes"] == len(text) and st["fell_back"] == 0


def test_bench_shaped_text_constant_qualities_long_matches():
    """the end-to-end leg's file: every quality line is 'I' * L (matches of 258 at distance 1 and at one record's distance), ~40:1"""
    text = fastq_text(60000, L=250, seed=5, constant_quality=True)
    gz = gzip.compress(text, 6)
    assert len(text) > 25 * len(gz)
    rc, got, st = inflate_parallel(gz, 4, 32768)
    assert rc == 0 and got == text and st["segments"] >= 4


def test_matches_that_reach_the_whole_window_and_into_the_previous_segment():
    """a 32 KiB random block repeated: every match is 32,768 back, so each segment's first 32 KiB come out of the window in front of it;
    then single-byte runs (distance 1) across segment borders"""
    rng = np.random.default_rng(3)
    block = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    text = block * 40 + b"A" * 300000 + block[:1000] * 500 + bytes(rng.integers(65, 70, 400000, dtype=np.uint8))
    for level in (6, 9):
        gz = gzip.compress(text, level)
        rc, got, st = inflate_parallel(gz, 4, 32768)
        if rc == 0:
            assert got == text
        else:
            assert rc == C2_E_INVALID                            # (e.g. no dynamic block starts in any segment's search range)
    # at least the compressible tail must give the route something to cut
    gz = gzip.compress(text + fastq_text(20000, seed=9), 6)
    rc, got, st = inflate_parallel(gz, 4, 32768)
    assert rc == 0 and got == text + fastq_text(20000, seed=9) and st["segments"] >= 2


def test_flush_points_stored_and_fixed_blocks():
    """pigz / Z_SYNC_FLUSH style streams: empty stored blocks between the data blocks; level 0: stored blocks only; tiny pieces: fixed blocks"""
    text = fastq_text(20000, seed=11)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for a in range(0, len(text), 70001):
        parts.append(co.compress(text[a:a + 70001]))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH if (a // 70001) % 2 else zlib.Z_FULL_FLUSH))
    for a in range(0, 2000, 40):                                  # a run of tiny fixed-Huffman blocks at the end
        parts.append(co.compress(text[a:a + 40]))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    gz = b"".join(parts)
    want = gzip.decompress(gz)
    rc, got, st = inflate_parallel(gz, 4, 32768)
    assert rc == 0 and got == want and st["segments"] >= 2
    stored = gzip.compress(text[:400000], 0)
    rc, got, st = inflate_parallel(stored, 4, 32768)
    assert rc == C2_E_INVALID or got == text[:400000]            # no dynamic block to find: declined (the serial route inflates it)
    mixed = zlib.compressobj(6, zlib.DEFLATED, 31)
    rnd = bytes(np.random.default_rng(2).integers(0, 256, 300000, dtype=np.uint8))      # incompressible: zlib stores it
    gz = mixed.compress(text[:500000]) + mixed.compress(rnd) + mixed.compress(text[500000:]) + mixed.flush()
    rc, got, st = inflate_parallel(gz, 4, 32768)
    assert rc == 0 and got == text[:500000] + rnd + text[500000:]


def test_header_fields_are_skipped():
    text = fastq_text(20000, seed=13)
    bio = io.BytesIO()
    with gzip.GzipFile(filename="reads_with_a_name.fastq", mode="wb", fileobj=bio, mtime=12345) as fh:
        fh.write(text)
    rc, got, st = inflate_parallel(bio.getvalue(), 4, 32768)
    assert rc == 0 and got == text
    # FEXTRA + FCOMMENT + FHCRC by hand around the same deflate data
    plain = gzip.compress(text, 6)
    deflate, trailer = plain[10:-8], plain[-8:]
    head = bytes([0x1f, 0x8b, 8, 4 | 16 | 2, 0, 0, 0, 0, 0, 3]) + (5).to_bytes(2, "little") + b"xy\x01\x00z" + b"a comment\x00"
    head += (zlib.crc32(head) & 0xffff).to_bytes(2, "little")
    assert gzip.decompress(head + deflate + trailer) == text
    rc, got, st = inflate_parallel(head + deflate + trailer, 4, 32768)
    assert rc == 0 and got == text


def test_declines_whatever_is_not_one_clean_member():
    text = fastq_text(20000, seed=17)
    good = gzip.compress(text, 6)
    half = len(text) // 2
    cases = {
        "two members": gzip.compress(text[:half]) + gzip.compress(text[half:]),
        "trailing zeros": good + b"\x00" * 64,
        "trailing garbage": good + b"not a gzip member",
        "cut short": good[:len(good) // 2],
        "cut in the trailer": good[:-3],
        "wrong crc": good[:-8] + bytes([good[-8] ^ 1]) + good[-7:],
        "wrong isize": good[:-4] + ((len(text) + 1) & 0xffffffff).to_bytes(4, "little"),
        "not gzip": b"@r1\nACGT\n+\nIIII\n" * 10000,
        "too small": gzip.compress(text[:2000]),
        "empty": gzip.compress(b""),
    }
    damaged = bytearray(good)
    damaged[len(good) // 2] ^= 0x5a
    cases["a flipped byte in the middle"] = bytes(damaged)
    for name, gz in cases.items():
        rc, got, st = inflate_parallel(gz, 4, 32768)
        assert rc == C2_E_INVALID and got is None, name
        assert b"inflate the file serially" in _native.load().c2_fastq_last_error(), name
    # one thread: nothing to share out
    rc, got, st = inflate_parallel(good, 1, 32768)
    assert rc == C2_E_INVALID


def test_every_flipped_byte_is_either_declined_or_harmless():
    """damage anywhere: the route may only answer with the text Python's gzip module gives for the same bytes -- never with different text"""
    text = fastq_text(6000, seed=19)
    good = gzip.compress(text, 6)
    rng = np.random.default_rng(23)
    for at in sorted(set(int(x) for x in rng.integers(0, len(good), 60))):
        bad = bytearray(good)
        bad[at] ^= 1 << int(rng.integers(0, 8))
        try:
            want = gzip.decompress(bytes(bad))
        except Exception:
            want = None
        rc, got, st = inflate_parallel(bytes(bad), 4, 32768)
        if rc == 0:
            assert want is not None and got == want, at
        else:
            assert rc == C2_E_INVALID, at


def test_destination_too_small_reports_the_size():
    text = fastq_text(20000, seed=29)
    gz = gzip.compress(text, 6)
    rc, got, st = inflate_parallel(gz, 4, 32768, cap=len(text) - 1)
    assert rc == C2_E_OVERFLOW
    rc, got, st = inflate_parallel(gz, 4, 32768, cap=len(text))
    assert rc == 0 and got == text


def test_the_file_routes_take_it_and_give_the_serial_routes_text(tmp_path, monkeypatch):
    """FastqStream.text() and fastq_unique() over a one-member .gz: with the route forced on (any size, 32 KiB segments) and switched off"""
    text = fastq_text(40000, seed=31)
    p = tmp_path / "reads.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    monkeypatch.setenv("C2_GZ_PARALLEL", "0")
    with _native.FastqStream(str(p), 0, 0, 0) as fq:
        serial = fq.text().tobytes()
    a0, o0, c0, n0 = _native.fastq_unique(str(p))
    monkeypatch.delenv("C2_GZ_PARALLEL")
    monkeypatch.setenv("C2_GZ_PARALLEL_MIN", "0")
    monkeypatch.setenv("C2_GZ_PARALLEL_CHUNK", "32768")
    with _native.FastqStream(str(p), 0, 0, 0) as fq:
        parallel = fq.text().tobytes()
    a1, o1, c1, n1 = _native.fastq_unique(str(p))
    assert serial == text and parallel == text
    assert np.array_equal(a0, a1) and np.array_equal(o0, o1) and np.array_equal(c0, c1) and n0 == n1 == 40000


def test_symbol_is_declared_in_the_header():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "crispresso2_amd.h")).read()
    assert "int c2_gz_inflate_parallel(" in hdr and "CRISPRessoCORE.py:1820-1823" in hdr
