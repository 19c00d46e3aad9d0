#!/usr/bin/env python3
"""FASTQ -> count tensors on a file several times the headline's size (default 30 M reads, 15.5 GB of text), device route against host parser;
with C2_LARGE_TABLES=1 the device route's last run goes on to every result table on disk (the allele table of ~9.7 M rows among them):
python tools/e2e_large.py [reads]"""
import json, os, sys, tempfile, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from crispresso2_amd import synth, _native, pipeline, tables, refs as R, CRISPResso2Align as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
L = 250
d = tempfile.mkdtemp(prefix="c2big_", dir="/dev/shm")
p = os.path.join(d, "r.fastq")
t0 = time.perf_counter()
with open(p, "wb") as out:                                            # written in parts of 10 M reads (consecutive blocks of the synthetic data set: the unique reads keep growing)
    for part in range(0, n, 10_000_000):
        m = min(10_000_000, n - part)
        reads = synth.make_reads(L, m, first_block=part // synth.BLOCK, workers=32)      # (later blocks of the same data set: new reads)
        tmp = p + ".part"
        synth.write_fastq(reads, tmp)
        with open(tmp, "rb") as fh:
            while True:
                b = fh.read(1 << 26)
                if not b:
                    break
                out.write(b)
        os.remove(tmp)
        del reads
print(json.dumps({"file_bytes": os.path.getsize(p), "written_in_s": round(time.perf_counter() - t0, 1)}), flush=True)
amp, g, inc = synth.amplicon_setup(L)
args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                       ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                       assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
ref = R.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
ref["sgRNA_orig_sequences"] = [amp[L // 2 - 16:L // 2 + 4]]
mat = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
tallies = {}
try:
    for route in ("device", "host"):
        os.environ["C2_FQ_INGEST"] = route
        runs = []
        for rep in range(3 if route == "device" else 2):
            t0 = time.perf_counter()
            res = pipeline.quantify_fastq(p, {"Reference": ref}, ["Reference"], mat, args, ctx=ctx)
            runs.append(time.perf_counter() - t0)
            c = res.per_ref["Reference"]
            tallies[route] = (res.stats["N_TOT_READS"], res.stats["N_TOTAL"], c["counts_total"], c["counts_modified"], c["counts_insertion"], c["counts_deletion"],
                              c["counts_substitution"], res.stats["N_READS_INPUT"], res.stats["N_COMPUTED_ALN"] + res.stats["N_COMPUTED_NOTALN"])
            if route == "device" and rep == 2 and os.environ.get("C2_LARGE_TABLES"):
                import shutil
                out, tt = os.path.join(d, "tables"), {}
                t1 = time.perf_counter()
                written = tables.write_tables(res, {"Reference": ref}, ["Reference"], out, timings=tt)
                t2 = time.perf_counter()
                print(json.dumps({"reads": n, "fastq_to_all_tables_seconds": round(runs[-1] + t2 - t1, 3), "write_tables": round(t2 - t1, 3),
                                  "stages": {k_: round(v_, 3) for k_, v_ in tt.items()}, "rows": res.allele_table().n_rows, "files": len(written),
                                  "bytes": sum(os.path.getsize(os.path.join(out, w_)) for w_ in written)}), flush=True)
                res.allele_table().close()
                shutil.rmtree(out, ignore_errors=True)
            del res
            time.sleep(0.5)
        print(json.dumps({"route": route, "reads": n, "seconds": [round(x, 3) for x in runs], "reads_per_s": round(n / min(runs[1:])), "tallies": tallies[route]}), flush=True)
    print(json.dumps({"same_tallies": tallies["device"] == tallies["host"]}))
finally:
    os.remove(p)
    os.rmdir(d)
