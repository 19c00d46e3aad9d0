#!/usr/bin/env python3
"""End-to-end rate of crispresso2_amd.pipeline.quantify_fastq: a synthetic FASTQ file (bench.py's read model, qualities
'I') -> ingest + dedup -> strand plan -> device alignments -> reference choice -> count tensors, with the wall time of
every stage.  Run on the GPU box:  python tools/e2e_rate.py [--reads N] [--len L] [--gz]"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--len", type=int, default=250, dest="L")
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--repeat", type=int, default=3, help="timed runs (the fastest is reported, all are listed)")
    a = ap.parse_args()
    from crispresso2_amd import synth, pipeline, refs as R, CRISPResso2Align as A
    reads = synth.make_reads(a.L, a.reads)                   # (fork pool: before any HIP call)
    amp, g, inc = synth.amplicon_setup(a.L)
    d = tempfile.mkdtemp(prefix="c2e2e_")
    path = os.path.join(d, "reads.fastq" + (".gz" if a.gz else ""))
    qual = b"I" * a.L
    opener = (lambda p: gzip.open(p, "wb", compresslevel=4)) if a.gz else (lambda p: open(p, "wb"))
    with opener(path) as fh:
        for k in range(a.reads):
            fh.write(b"@read%d\n%s\n+\n%s\n" % (k, reads[k].tobytes(), qual))
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20,
                           needleman_wunsch_gap_extend=-2, ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    ref = R.make_ref("Reference", amp, [a.L // 2], inc, min_aln_score=60)
    m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
    pipeline.quantify_fastq(path, {"Reference": ref}, ["Reference"], m, args)          # warm-up (context, allocations)
    runs = []
    for rep in range(a.repeat):
        time.sleep(0.5)                                      # (the previous run's buffers are unmapped by helper threads: let them finish)
        tm = {}
        t0 = time.perf_counter()
        res = pipeline.quantify_fastq(path, {"Reference": ref}, ["Reference"], m, args, timings=tm)
        dt = time.perf_counter() - t0
        runs.append((dt, tm))
    size = os.path.getsize(path)
    os.remove(path)
    c = res.per_ref["Reference"]
    dt, tm = min(runs, key=lambda x: x[0])
    print(json.dumps({"reads": a.reads, "file_bytes_gz" if a.gz else "file_bytes": size, "seconds": dt, "reads_per_s": a.reads / dt,
                      "seconds_all_runs": [r[0] for r in runs],
                      "unique_reads": res.stats["N_COMPUTED_ALN"] + res.stats["N_COMPUTED_NOTALN"], "stage_seconds": tm,
                      "reads_aligned": c["counts_total"], "modified": c["counts_modified"], "N_TOTAL": res.stats["N_TOTAL"]}))


if __name__ == "__main__":
    main()
