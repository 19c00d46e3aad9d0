"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PAYLOAD_FIELDS = [
    "all_insertion_positions", "all_insertion_left_positions", "insertion_positions", "insertion_coordinates",
    "insertion_sizes", "insertion_n", "all_deletion_positions", "all_deletion_coordinates", "deletion_positions",
    "deletion_coordinates", "deletion_sizes", "deletion_n", "all_substitution_positions", "substitution_positions",
    "all_substitution_values", "substitution_values", "substitution_n", "ref_positions",
]


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        import gzip
        with gzip.open(path, "rt") as fh:
            return json.load(fh)
    with open(path) as fh:
        return json.load(fh)


def norm(v):
    """JSON-normal form of a payload value (tuples -> lists, numpy -> python)."""
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [norm(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def payload_diff(got, exp):
    """List of (field, got, expected) that differ. `got` may be a dict or a ResultsSlotsDict."""
    bad = []
    for f in PAYLOAD_FIELDS:
        g = norm(got[f])
        e = norm(exp[f])
        if g != e:
            bad.append((f, g, e))
    return bad


def matrices():
    """name -> int64 score matrix, read with the PRODUCT's read_matrix from the package's own files."""
    from crispresso2_amd import CRISPResso2Align as A
    d = os.path.dirname(A.__file__)
    return {"EDNAFULL": A.read_matrix(os.path.join(d, "EDNAFULL")),
            "BLOSUM62": A.read_matrix(os.path.join(d, "BLOSUM62"))}
