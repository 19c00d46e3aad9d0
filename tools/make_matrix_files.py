#!/usr/bin/env python3
"""Write crispresso2_amd/EDNAFULL and crispresso2_amd/BLOSUM62 in NCBI text format.

The numbers are the standard NCBI NUC.4.4 (EDNAFULL, with the extra U column CRISPResso2 ships)
and BLOSUM62 matrices; this script takes them from the reference's read_matrix() output
(oracle/_ref) so that `read_matrix(<our file>)` is array-equal to `read_matrix(<reference file>)`
(tests/test_oracle.py::test_matrix_files_match_reference checks it when oracle/_ref exists).
Run in the dev container only.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

A, _ = oracle.ref()
HEAD = {
    "EDNAFULL": "# NUC.4.4 / EDNAFULL nucleotide scoring matrix with IUPAC ambiguity codes (plus U)\n"
                "# match 5, mismatch -4; ambiguity entries are rounded expected scores\n",
    "BLOSUM62": "# BLOSUM62 amino-acid substitution matrix (NCBI), 1/2-bit units\n",
}
for name in ("EDNAFULL", "BLOSUM62"):
    m = A.read_matrix(os.path.join(ROOT, "oracle/_ref", name))
    syms = [c for c in range(m.shape[0]) if m[c].any() or m[:, c].any()]
    with open(os.path.join(ROOT, "crispresso2_amd", name), "w") as fh:
        fh.write(HEAD[name])
        fh.write("   " + "".join("%4s" % chr(c) for c in syms) + "\n")
        for r in syms:
            fh.write("%-3s" % chr(r) + "".join("%4d" % m[r, c] for c in syms) + "\n")
    print(name, len(syms), "symbols")
