"""CPU oracle for the align + classify hot path -- TEST INFRASTRUCTURE ONLY.

`oracle.global_align` / `oracle.find_indels_substitutions` wrap oracle/c2_oracle.c (a plain-C
restatement of CRISPResso2Align.pyx:101-434 and CRISPRessoCOREResources.pyx:68-187) and return
objects shaped like the reference's.  `oracle.ref()` returns the reference's own Cython modules
compiled by oracle/build_ref.py (oracle/_ref/, present when built in the dev container).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Parity status: pinned (see header of c2_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libc2_oracle.so")
_lib = None

ERR_GAP_INCENTIVE_LEN = 1
ERR_BAD_POINTER = 2
ERR_UNINIT_POINTER = 4
ERR_EMPTY = 8
ERR_OOB_CHAR = 16
WARN_SENTINEL_PATH = 32

_LIST_FIELDS = [
    "ref_positions", "all_insertion_positions", "all_insertion_left_positions", "insertion_positions",
    "insertion_coordinates", "insertion_sizes", "all_deletion_positions", "all_deletion_coordinates",
    "deletion_positions", "deletion_coordinates", "deletion_sizes", "all_substitution_positions",
    "all_substitution_values", "substitution_positions", "substitution_values",
]


class _Payload(ctypes.Structure):
    _fields_ = ([("cap", ctypes.c_int32)]
                + [f for name in _LIST_FIELDS
                   for f in (("n_" + name, ctypes.c_int32), (name, ctypes.POINTER(ctypes.c_int32)))]
                + [("insertion_n", ctypes.c_int64), ("deletion_n", ctypes.c_int64), ("substitution_n", ctypes.c_int64)])


def build(force=False):
    """Compile c2_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "c2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _LIB_PATH, src])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.c2o_global_align.restype = ctypes.c_int
        _lib.c2o_find_indels_substitutions.restype = ctypes.c_int
        _lib.c2o_calculate_homology.restype = ctypes.c_double
    return _lib


def ref():
    """The reference's own compiled Cython modules (oracle/_ref/c2ref), or None if not built."""
    import importlib
    import sys
    d = os.path.join(_HERE, "_ref")
    if not os.path.isdir(os.path.join(d, "c2ref")):
        return None
    if d not in sys.path:
        sys.path.insert(0, d)
    try:
        return (importlib.import_module("c2ref.CRISPResso2Align"),
                importlib.import_module("c2ref.CRISPRessoCOREResources"))
    except ImportError:
        return None


def global_align_raw(seqj, seqi, matrix, gap_incentive, gap_open=-1, gap_extend=-1):
    """-> (status, aligned_read, aligned_ref, matches, length)"""
    bj = seqj.encode("utf-8")
    bi = seqi.encode("utf-8")
    m = np.ascontiguousarray(matrix, dtype=np.int64)
    g = np.ascontiguousarray(gap_incentive, dtype=np.int64)
    cap = len(bi) + len(bj) + 1
    oj = ctypes.create_string_buffer(cap)
    oi = ctypes.create_string_buffer(cap)
    n = ctypes.c_int(0)
    mt = ctypes.c_int(0)
    st = lib().c2o_global_align(bj, len(bj), bi, len(bi),
                                m.ctypes.data_as(ctypes.c_void_p), int(m.shape[0]),
                                g.ctypes.data_as(ctypes.c_void_p), int(g.shape[0]),
                                int(gap_open), int(gap_extend), oj, oi, ctypes.byref(n), ctypes.byref(mt))
    return st, oj.raw[:n.value].decode(), oi.raw[:n.value].decode(), mt.value, n.value


def global_align(seqj, seqi, matrix, gap_incentive, gap_open=-1, gap_extend=-1):
    """Same return value as the reference: (aligned_read, aligned_ref, round(100*m/n, 3))."""
    st, s1, s2, m, n = global_align_raw(seqj, seqi, matrix, gap_incentive, gap_open, gap_extend)
    if st & ERR_GAP_INCENTIVE_LEN:
        return 0                                        # CRISPResso2Align.pyx:124-126
    if st & ~WARN_SENTINEL_PATH:
        raise Exception("oracle: alignment outside the reference's defined domain (status %d)" % st)
    return s1, s2, round(100 * m / float(n), 3)         # pyx:433-434


def find_indels_substitutions(read_seq_al, ref_seq_al, _include_indx):
    """dict with the 18 fields of the reference's ResultsSlotsDict (COREResources.pyx:167-187)."""
    br = read_seq_al.encode("utf-8")
    bf = ref_seq_al.encode("utf-8")
    n = len(bf)
    inc = np.ascontiguousarray(np.asarray(list(_include_indx), dtype=np.int64).astype(np.int32))
    cap = 2 * n + 8
    while True:
        p = _Payload()
        p.cap = cap
        bufs = {}
        for name in _LIST_FIELDS:
            bufs[name] = np.zeros(cap, dtype=np.int32)
            setattr(p, name, bufs[name].ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        lib().c2o_find_indels_substitutions(br, bf, n, inc.ctypes.data_as(ctypes.c_void_p), int(inc.shape[0]),
                                            ctypes.byref(p))
        need = max(getattr(p, "n_" + name) for name in _LIST_FIELDS)
        if need <= cap:
            break
        cap = need + 8      # the negative-coordinate quirk can make deletion ranges longer than the alignment
    out = {}
    for name in _LIST_FIELDS:
        out[name] = bufs[name][:getattr(p, "n_" + name)].tolist()
    for name in ("insertion_coordinates", "all_deletion_coordinates", "deletion_coordinates"):
        v = out[name]
        out[name] = [(v[k], v[k + 1]) for k in range(0, len(v), 2)]
    for name in ("all_substitution_values", "substitution_values"):
        out[name] = np.array([chr(c) for c in out[name]])
    out["insertion_n"] = int(p.insertion_n)
    out["deletion_n"] = int(p.deletion_n)
    out["substitution_n"] = int(p.substitution_n)
    return out


def calculate_homology(a, b):
    return lib().c2o_calculate_homology(a, b)
