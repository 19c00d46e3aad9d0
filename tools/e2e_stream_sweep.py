#!/usr/bin/env python3
"""Round 3 measurement: the chunked ingest alone (threads x range bytes) and FASTQ -> count tensors streamed / one-batch, on a
10 M-read synthetic file in /dev/shm.  python tools/e2e_stream_sweep.py [--reads N] [--quick]"""
import argparse, json, os, sys, tempfile, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--quick", action="store_true")
a = ap.parse_args()
from crispresso2_amd import synth, _native, pipeline, refs as R, CRISPResso2Align as A
L = 250
reads = synth.make_reads(L, a.reads, workers=32)
d = tempfile.mkdtemp(prefix="c2sw_", dir="/dev/shm")
p = os.path.join(d, "r.fastq")
synth.write_fastq(reads, p)
del reads


def ingest(threads, rb):
    if threads:
        os.environ["C2_FASTQ_THREADS"] = str(threads)
    else:
        os.environ.pop("C2_FASTQ_THREADS", None)
    os.environ["C2_FASTQ_RANGE_BYTES"] = str(rb)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        with _native.FastqUnique(p) as fq:
            dt = time.perf_counter() - t0
            nu = len(fq.counts)
        best = min(best, dt)
        time.sleep(0.3)
    return best, nu

combos = [(0, 4 << 20), (0, 16 << 20)] if a.quick else [(t, rb) for t in (32, 64, 96, 128, 192, 256) for rb in (1 << 20, 4 << 20, 16 << 20)]
for t, rb in combos:
    dt, nu = ingest(t, rb)
    print(json.dumps({"ingest_only": {"threads": t, "range_bytes": rb, "seconds": round(dt, 4), "reads_per_s": round(a.reads / dt), "unique": nu}}), flush=True)

amp, g, inc = synth.amplicon_setup(L)
args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                       ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                       assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
ref = R.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
best_ingest = (128, 4 << 20)
for threads, rb, stream, minb in ([(0, 4 << 20, True, 200_000), (0, 4 << 20, True, 500_000), (0, 16 << 20, True, 500_000), (0, 4 << 20, False, 0)] if a.quick else
                                  [(128, 4 << 20, False, 0), (64, 4 << 20, True, 200_000), (128, 4 << 20, True, 200_000), (128, 4 << 20, True, 500_000),
                                   (128, 4 << 20, True, 1_000_000), (128, 1 << 20, True, 200_000), (192, 4 << 20, True, 500_000), (128, 16 << 20, True, 500_000)]):
    if threads:
        os.environ["C2_FASTQ_THREADS"] = str(threads)
    else:
        os.environ.pop("C2_FASTQ_THREADS", None)
    os.environ["C2_FASTQ_RANGE_BYTES"] = str(rb)
    pipeline.STREAM_MIN_BATCH = minb or 200_000
    runs = []
    for rep in range(3):
        tm = {}
        t0 = time.perf_counter()
        res = pipeline.quantify_fastq(p, {"Reference": ref}, ["Reference"], m, args, timings=tm, stream=stream)
        runs.append((time.perf_counter() - t0, tm))
        tot = res.per_ref["Reference"]["counts_total"]
        del res
        time.sleep(0.4)
    dt, tm = min(runs[1:], key=lambda x: x[0])
    print(json.dumps({"e2e": {"threads": threads, "range_bytes": rb, "stream": stream, "min_batch": minb, "seconds": round(dt, 4),
                              "reads_per_s": round(a.reads / dt), "all": [round(r[0], 3) for r in runs], "counts_total": tot,
                              "stages": {k: round(v, 4) for k, v in tm.items()}}}), flush=True)
os.remove(p); os.rmdir(d)
