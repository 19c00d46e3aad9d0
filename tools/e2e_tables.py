#!/usr/bin/env python3
"""FASTQ -> every result table on disk (pipeline.quantify_fastq + tables.write_tables: the allele frequency table and the alleles around the
guide's cut among them), three runs over an N-read synthetic file in /dev/shm, stage times of the table part per run.
python tools/e2e_tables.py [reads]   (C2_WORKERS=1 under rocprofv3: no fork()ed data generation)"""
import json, os, shutil, sys, tempfile, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crispresso2_amd import synth, _native, pipeline, tables, refs as R, CRISPResso2Align as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 250
reads = synth.make_reads(L, n, workers=int(os.environ.get("C2_WORKERS", "32")))
d = tempfile.mkdtemp(prefix="c2tab_", dir=os.environ.get("C2_E2E_DIR", "/dev/shm"))
p = os.path.join(d, "r.fastq")
synth.write_fastq(reads, p)
del reads
amp, g, inc = synth.amplicon_setup(L)
args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                       ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                       assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
ref = R.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
ref["sgRNA_orig_sequences"] = [amp[L // 2 - 16:L // 2 + 4]]
mat = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
out = os.path.join(d, "tables")
try:
    for rep in range(int(os.environ.get("C2_REPS", "3"))):
        shutil.rmtree(out, ignore_errors=True)
        tt = {}
        t0 = time.perf_counter()
        res = pipeline.quantify_fastq(p, {"Reference": ref}, ["Reference"], mat, args, ctx=ctx)
        t1 = time.perf_counter()
        written = tables.write_tables(res, {"Reference": ref}, ["Reference"], out, timings=tt)
        t2 = time.perf_counter()
        sizes = {w: os.path.getsize(os.path.join(out, w)) for w in written if "llele" in w}
        print(json.dumps({"reads": n, "seconds": round(t2 - t0, 4), "quantify_fastq": round(t1 - t0, 4), "write_tables": round(t2 - t1, 4),
                          "stages": {k: round(v, 4) for k, v in tt.items()}, "rows": res.allele_table().n_rows, "files": len(written), "allele_files_bytes": sizes,
                          "route": getattr(res, "ingest_route", "host")}), flush=True)
        res.allele_table().close()
        del res
        time.sleep(0.3)
finally:
    shutil.rmtree(d, ignore_errors=True)
