#!/usr/bin/env python3
"""Rate of the FASTQ ingest + de-duplication step in front of the kernels: the native parser (c2_fastq_unique) against the
reference's way of doing it, a Python readline loop (CRISPRessoCORE.py:1820-1849, restated inline below so that this
tool imports nothing from oracle/).  Host-only; no GPU needed.
    python tools/fastq_rate.py [--reads N] [--len L] [--dir DIR]"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def python_loop(path):
    opener = (lambda x: gzip.open(x, 'rt')) if path.endswith('.gz') else open
    cache = {}
    with opener(path) as fh:
        fastq_id = fh.readline()
        while fastq_id:
            seq = fh.readline().strip()
            fh.readline().strip()
            fh.readline()
            if seq in cache:
                cache[seq] += 1
            else:
                cache[seq] = 1
            fastq_id = fh.readline()
    return cache


def write_bgzf(src, dst, block=0xff00, level=4):
    """bgzip's format: gzip members of <= 64 KiB of text, each with its compressed size in a 'BC' extra subfield."""
    import struct
    import zlib
    with open(src, "rb") as fin, open(dst, "wb") as fout:
        while True:
            c = fin.read(block)
            z = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = z.compress(c) + z.flush()
            fout.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
                       + body + struct.pack("<II", zlib.crc32(c), len(c)))
            if not c:
                break


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=250, dest="L")
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    from crispresso2_amd import synth, _native
    reads = synth.make_reads(a.L, a.reads, workers=1)
    qual = b"I" * a.L
    d = a.dir or tempfile.mkdtemp(prefix="c2fq_")
    plain, gz = os.path.join(d, "r.fastq"), os.path.join(d, "r.fastq.gz")
    with open(plain, "wb") as fh:
        for k in range(a.reads):
            fh.write(b"@read%d\n%s\n+\n%s\n" % (k, reads[k].tobytes(), qual))
    with open(plain, "rb") as src, gzip.open(gz, "wb", compresslevel=4) as dst:
        while True:
            b = src.read(1 << 24)
            if not b:
                break
            dst.write(b)
    out = {"reads": a.reads, "read_len": a.L, "plain_bytes": os.path.getsize(plain), "gz_bytes": os.path.getsize(gz)}
    for name, path in (("plain", plain), ("gz", gz)):
        with open(path, "rb") as fh:                       # both contenders read from the page cache
            while fh.read(1 << 24):
                pass
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            arena, offsets, counts, total = _native.fastq_unique(path)
            t1 = time.perf_counter()
            best = t1 - t0 if best is None else min(best, t1 - t0)
        t0, t1 = 0.0, best
        t1b = time.perf_counter()
        ref = python_loop(path)
        t2 = t1 + (time.perf_counter() - t1b)
        assert total == a.reads and len(ref) == len(counts) and sum(ref.values()) == int(counts.sum())
        out[name] = {"native_s": t1 - t0, "native_reads_per_s": a.reads / (t1 - t0), "python_loop_s": t2 - t1,
                     "python_loop_reads_per_s": a.reads / (t2 - t1), "speedup": (t2 - t1) / (t1 - t0), "unique": int(len(counts))}
    # the routes a .gz file can take (c2_fastq.cpp): whole-buffer libdeflate (default for a plain gzip stream), BGZF members
    # inflated by all threads (default for bgzip output), zlib streaming (the fallback)
    bgzf = os.path.join(d, "r.bgzf.fastq.gz")
    write_bgzf(plain, bgzf)
    out["bgzf_bytes"] = os.path.getsize(bgzf)
    routes = {}
    for name, path, route in (("gz_stream_zlib", gz, "stream"), ("gz_whole_libdeflate", gz, "auto"),
                              ("bgzf_threads_libdeflate", bgzf, "auto"), ("bgzf_threads_zlib", bgzf, "zlib"), ("bgzf_stream_zlib", bgzf, "stream")):
        os.environ["C2_FASTQ_GZ"] = route
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            arena, offsets, counts, total = _native.fastq_unique(path)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert total == a.reads and len(counts) == out["plain"]["unique"]
        routes[name] = {"native_s": best, "native_reads_per_s": a.reads / best}
    os.environ.pop("C2_FASTQ_GZ", None)
    out["gz_routes"] = routes
    # the reference's read filter (filterFastqs.py, -q) fused into the ingest: same files, every read passes (constant quality 'I')
    fused = {}
    for name, path in (("plain", plain), ("gz", gz), ("bgzf", bgzf)):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            arena, offsets, counts, total = _native.fastq_unique(path, min_average_read_quality=30, min_bp_quality_or_N=10)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert total == a.reads and len(counts) == out["plain"]["unique"]
        fused[name] = {"native_s": best, "native_reads_per_s": a.reads / best}
    out["fused_read_filter_q30_mask10"] = fused
    out["host_threads"] = os.cpu_count()
    for p in (plain, gz, bgzf):
        os.remove(p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
