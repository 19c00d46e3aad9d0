#!/usr/bin/env python3
"""For rocprofv3 --kernel-trace --stats: FASTQ -> count tensors on the device route, three runs over a 10 M-read file.
python tools/e2e_trace.py [reads]   (C2_WORKERS=1 under the profiler: no fork()ed data generation)"""
import json, os, sys, tempfile, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crispresso2_amd import synth, _native, pipeline, refs as R, CRISPResso2Align as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 250
reads = synth.make_reads(L, n, workers=int(os.environ.get("C2_WORKERS", "32")))
d = tempfile.mkdtemp(prefix="c2tr_", dir="/dev/shm")
p = os.path.join(d, "r.fastq")
synth.write_fastq(reads, p)
del reads
amp, g, inc = synth.amplicon_setup(L)
args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                       ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                       assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
ref = R.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
mat = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
try:
    for rep in range(3):
        t0 = time.perf_counter()
        res = pipeline.quantify_fastq(p, {"Reference": ref}, ["Reference"], mat, args, ctx=ctx)
        print(json.dumps({"seconds": round(time.perf_counter() - t0, 4), "route": getattr(res, "ingest_route", "host"), "N_TOTAL": res.stats["N_TOTAL"]}), flush=True)
        del res
        time.sleep(0.3)
finally:
    os.remove(p)
    os.rmdir(d)
