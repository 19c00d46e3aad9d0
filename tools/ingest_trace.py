#!/usr/bin/env python3
"""Phase times of the native FASTQ ingest (C2_FASTQ_TRACE) on an N-read synthetic file in /dev/shm; host only.
python tools/ingest_trace.py [--reads N] [--threads 32,64,128] [--range-mb 4,16]"""
import argparse, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--threads", default="")
ap.add_argument("--range-mb", default="4")
a = ap.parse_args()
os.environ["C2_FASTQ_TRACE"] = "1"
from crispresso2_amd import synth, _native
reads = synth.make_reads(250, a.reads, workers=32)
d = tempfile.mkdtemp(prefix="c2tr_", dir="/dev/shm")
p = os.path.join(d, "r.fastq")
synth.write_fastq(reads, p)
for th in (a.threads.split(",") if a.threads else [""]):
    for rb in a.range_mb.split(","):
        if th:
            os.environ["C2_FASTQ_THREADS"] = th
        os.environ["C2_FASTQ_RANGE_BYTES"] = str(int(float(rb) * (1 << 20)))
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            with _native.FastqUnique(p) as fq:
                t1 = time.perf_counter()
                n = len(fq.counts)
            print("threads %s range %s MB: ingest %.3f s, unique %d" % (th or "auto", rb, t1 - t0, n), flush=True)
            time.sleep(0.5)
os.remove(p); os.rmdir(d)
