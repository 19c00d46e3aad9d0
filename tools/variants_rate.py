#!/usr/bin/env python3
"""Rate of the Python-level caller path (crispresso2_amd.variants.get_new_variant_objects = the reference's
get_new_variant_object per unique read, CRISPRessoCORE.py:627-798) on synthetic unique reads: batched alignment +
batched classifier lists + building the reference's per-read dicts.  Run on the GPU box:
    python tools/variants_rate.py [--reads N] [--len L]"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000)
    ap.add_argument("--len", type=int, default=250, dest="L")
    a = ap.parse_args()
    from crispresso2_amd import synth, variants, refs as R, CRISPResso2Align as A
    amp, g, inc = synth.amplicon_setup(a.L)
    raw = synth.make_reads(a.L, a.reads)
    seqs = list(dict.fromkeys(r.tobytes().decode() for r in raw))          # unique reads, first-seen order
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20,
                           needleman_wunsch_gap_extend=-2, use_legacy_insertion_quantification=False, ignore_deletions=False,
                           ignore_insertions=False, ignore_substitutions=False, assign_ambiguous_alignments_to_first_reference=False,
                           expand_ambiguous_alignments=False, prime_editing_pegRNA_scaffold_seq="")
    ref = R.make_ref("Reference", amp, [a.L // 2], inc, min_aln_score=60)
    m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
    variants.get_new_variant_objects(args, seqs[:2000], {"Reference": ref}, ["Reference"], m)      # warm-up
    t0 = time.perf_counter()
    out = variants.get_new_variant_objects(args, seqs, {"Reference": ref}, ["Reference"], m)
    dt = time.perf_counter() - t0
    n_mod = sum(1 for v in out if v.get("class_name", "").endswith("_MODIFIED"))
    print(json.dumps({"unique_reads": len(seqs), "seconds": dt, "unique_reads_per_s": len(seqs) / dt, "modified": n_mod,
                      "note": "single Python process; the reference's per-read function does ~1.3 k reads/s per core at 250 bp"}))


if __name__ == "__main__":
    main()
