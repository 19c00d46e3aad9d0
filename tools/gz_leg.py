#!/usr/bin/env python3
"""The end-to-end .gz leg alone: bench.py's headline reads as a FASTQ file, as ONE ordinary gzip member -> count tensors
(pipeline.quantify_fastq), timed like bench.py's e2e legs.  Run on the GPU box:  python tools/gz_leg.py [--reads N] [--repeat K]
C2_GZ_PARALLEL=0: the host inflates the member on one thread (libdeflate) first -- the route before round 5."""
import argparse
import json
import os
import shutil
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--workers", type=int, default=16)
    a = ap.parse_args()
    from crispresso2_amd import synth, pipeline, refs as RF, CRISPResso2Align as A, _native
    L = 250
    reads = synth.make_reads(L, a.reads, workers=a.workers)           # (fork pool: before any HIP call)
    files = bench._e2e_prepare(reads, a.workers, bgzf=True)
    try:
        amp, g, inc = synth.amplicon_setup(L)
        args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                               ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                               assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
        ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
        m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
        out = {"reads": a.reads, "file_bytes_gzip": files["bytes_gzip"], "file_bytes_plain": files["bytes_plain"]}
        for kind in ("plain", "gzip"):
            runs, route, tallies = [], None, None
            for r in range(a.repeat + 1):
                tm = {} if r == a.repeat else None               # (stage times: a device synchronisation per stage -- the last run only, not a timed one)
                t0 = time.perf_counter()
                res = pipeline.quantify_fastq(files[kind], {"Reference": ref}, ["Reference"], m, args, timings=tm)
                dt = time.perf_counter() - t0
                runs.append(dt)
                route = getattr(res, "ingest_route", "host")
                c = res.per_ref["Reference"]
                tallies = (res.stats["N_TOT_READS"], res.stats["N_TOTAL"], c["counts_total"], c["counts_modified"])
                del res
                time.sleep(0.3)
            out[kind] = {"seconds": min(runs[1:a.repeat] or runs[1:]), "reads_per_s": a.reads / min(runs[1:a.repeat] or runs[1:]), "seconds_all_runs": runs, "ingest_route": route, "tallies": tallies,
                         "stages_last_run": tm}
        out["gzip"]["inflate_plan"] = _native.gz_parallel_last()
        out["same_tallies"] = out["plain"]["tallies"] == out["gzip"]["tallies"]
        print(json.dumps(out))
    finally:
        shutil.rmtree(files["dir"], ignore_errors=True)


if __name__ == "__main__":
    main()
