#!/usr/bin/env python3
"""debug: the ragged + unrelated batch of test_ragged_and_unrelated_reads... twice (default, C2_FULL_PLANE_IN_LDS=1): which rows differ, where"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from crispresso2_amd import synth, _native, CRISPResso2Align as A
from crispresso2_amd.batch import BatchAligner
m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
L, n = 250, 6000
amp, g, inc = synth.amplicon_setup(L)
rng = np.random.default_rng(2025)
base = synth.make_reads(L, n)
reads = []
for k in range(n):
    s = base[k].tobytes().decode()[:int(rng.integers(200, L + 1))]
    if k % 10 == 7:
        s = "".join(rng.choice(list("ACGT"), len(s)))
    reads.append(s)
al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
res = al.align(reads)
for knob in ("", "C2_FULL_PLANE_IN_LDS", ""):
    if knob: os.environ[knob] = "1"
    other = al.align(reads)
    if knob: del os.environ[knob]
    print("knob", knob or "(none)", "records equal", np.array_equal(other.records, res.records))
    for name, a, b in (("read", res.aln_read, other.aln_read), ("ref", res.aln_ref, other.aln_ref)):
        bad = np.nonzero((a != b).any(axis=1))[0]
        print(name, "rows that differ:", len(bad))
        for k in bad[:12]:
            T = int(res.records["aln_len"][k]); pos = np.nonzero(a[k] != b[k])[0]
            print("  row", k, "Lj", len(reads[k]), "T", T, "junk" if k % 10 == 7 else "", "diff at", pos[:10].tolist(), "first", a[k][pos[:6]].tolist(), "second", b[k][pos[:6]].tolist())
