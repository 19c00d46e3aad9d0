"""Drop-in for the reference's Cython module `CRISPResso2.CRISPResso2Align`
(reference CRISPResso2/CRISPResso2Align.pyx): same names, signatures, return types and
error behaviour; `global_align` runs on the MI355X through the C ABI (no CPU fallback).
"""
import ctypes

import numpy as np

from . import _native
from . import prime as _prime


def read_matrix(path):
    """Score matrix file in NCBI format -> int64[max_ord+1, max_ord+1] with mat[ord(a), ord(b)] = score of a against b
    (reference pyx:33-61).  Same reading rules as the reference, quirks included: '#' lines before the column header are
    skipped (a blank line there is an IndexError, as in the reference); columns are the header's symbols in order; the k-th
    row of numbers belongs to the k-th header symbol whatever its own label says; the LAST character of every row line is
    dropped before it is split (the newline -- or a digit, if the file does not end with one)."""
    with open(path) as fh:
        rows = fh.readlines()
    cursor = 0
    while True:
        head = rows[cursor].strip() if cursor < len(rows) else ''
        cursor += 1
        if head[0] != '#':                                  # ('' -> IndexError: string index out of range)
            break
    columns = [ord(tok) for tok in head.split(' ') if tok]
    size = max(columns) + 1
    mat = np.zeros((size, size), dtype=np.int64)
    for row_no, text in enumerate(rows[cursor:]):
        scores = [int(tok) for tok in text[:-1].split(' ')[1:] if tok]
        for col, score in zip(columns, scores):
            mat[columns[row_no], col] = score               # (looked up per value: a blank line past the last row is harmless, as in the reference)
    return mat


def make_matrix(match_score=5, mismatch_score=-4, n_mismatch_score=-2, n_match_score=-1):
    """Match/mismatch score matrix over A,T,C,G,N (reference pyx:63-99); defaults equal EDNAFULL's values."""
    nuc_ords = [ord(x) for x in 'ATCG']
    n = ord('N')
    a = np.zeros((max(nuc_ords + [n]) + 1,) * 2, dtype=np.int64)
    for x in nuc_ords:
        for y in nuc_ords:
            a[x, y] = match_score if x == y else mismatch_score
        a[x, n] = n_mismatch_score
        a[n, x] = n_mismatch_score
    a[n, n] = n_match_score
    return a


def _as_int64(arr, ndim, name):
    """The reference's typed buffers reject anything but int64 ndarrays (ValueError: Buffer dtype mismatch)."""
    if not isinstance(arr, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(arr).__name__))
    if arr.dtype != np.int64:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_LONG' but got '%s'" % arr.dtype.name)
    if arr.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, arr.ndim))
    return np.ascontiguousarray(arr)


def global_align(pystr_seqj, pystr_seqi, matrix, gap_incentive, gap_open=-1, gap_extend=-1):
    """Global alignment (Needleman-Wunsch, affine gaps, per-position gap incentive) of the read
    `pystr_seqj` against the reference `pystr_seqi` (reference pyx:101-434).

    Returns (aligned_read, aligned_ref, round(100*matches/len, 3)).
    A gap_incentive of the wrong length prints the reference's message and returns 0 (pyx:124-126).
    """
    if not isinstance(pystr_seqj, str):
        raise TypeError("Argument 'pystr_seqj' has incorrect type (expected str, got %s)" % type(pystr_seqj).__name__)
    if not isinstance(pystr_seqi, str):
        raise TypeError("Argument 'pystr_seqi' has incorrect type (expected str, got %s)" % type(pystr_seqi).__name__)
    m = _as_int64(matrix, 2, 'matrix')
    g = _as_int64(gap_incentive, 1, 'gap_incentive')
    bj = pystr_seqj.encode('UTF-8')
    bi = pystr_seqi.encode('UTF-8')
    max_i = len(pystr_seqi)
    if len(g) != max_i + 1:
        print('\nERROR: Mismatch in gap_incentive length (gap_incentive: ' + str(len(g)) + ' ref: ' + str(max_i + 1) + '\n')
        return 0
    # a run whose reads were registered with crispresso2_amd.prime answers from one device batch (same kernels, same results)
    hit = _prime.lookup_alignment(pystr_seqj, pystr_seqi, m, g, gap_open, gap_extend)
    if hit is not None:
        return hit
    _prime.stats["per_call_align"] += 1
    if _native.in_forked_child():
        # a worker the reference fork()ed after the GPU was opened (CRISPRessoCORE.py:1870-1898): the call is served by a spawned helper
        return _native.forked_child_helper().call("global_align", pystr_seqj, pystr_seqi, m, g, gap_open, gap_extend)
    ctx = _native.default_context()
    cap = len(bi) + len(bj) + 1
    oj = ctypes.create_string_buffer(cap)
    oi = ctypes.create_string_buffer(cap)
    n = ctypes.c_int32(0)
    mt = ctypes.c_int32(0)
    st = ctypes.c_int32(0)
    rc = ctx.lib.c2_global_align(ctx.handle, bj, len(bj), bi, len(bi),
                                 m.ctypes.data_as(ctypes.c_void_p), int(m.shape[0]),
                                 g.ctypes.data_as(ctypes.c_void_p), int(g.shape[0]),
                                 int(gap_open), int(gap_extend), oj, oi,
                                 ctypes.byref(n), ctypes.byref(mt), ctypes.byref(st))
    ctx.check(rc, "c2_global_align")
    if st.value != 0:
        # outside the reference's defined domain (empty sequence, out-of-matrix character, or a traceback
        # through cells the reference never initialises): the reference raises 'wtf4!', reads garbage or
        # crashes there; this implementation always raises.
        raise Exception('global_align: undefined alignment (status %d) for seqj: %s seqi: %s' % (st.value, pystr_seqj, pystr_seqi))
    align_j = oj.raw[:n.value].decode('UTF-8', 'strict')
    align_i = oi.raw[:n.value].decode('UTF-8', 'strict')
    final_score = 100 * mt.value / float(n.value)
    return align_j, align_i, round(final_score, 3)
