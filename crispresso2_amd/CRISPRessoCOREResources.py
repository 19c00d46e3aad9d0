"""Drop-in for the reference's Cython module `CRISPResso2.CRISPRessoCOREResources`
(reference CRISPResso2/CRISPRessoCOREResources.pyx): same names, signatures and return types.
The column walk runs on the MI355X (c2_classify_lists_kernel) through the C ABI; this module
only turns the flat int32 lists that come back into the reference's Python objects.
There is no CPU fallback.
"""
import ctypes

import numpy as np

from . import _native
from . import prime as _prime


class ResultsSlotsDict():
    """Same slots and dict-style access as the reference class (pyx:18-65); CRISPRessoShared's JSON
    encoder/decoder type-check this class, so INTEGRATION.md installs *this* module under the
    reference's module name."""
    __slots__ = (
        'all_insertion_positions',
        'all_insertion_left_positions',
        'insertion_positions',
        'insertion_coordinates',
        'insertion_sizes',
        'insertion_n',
        'all_deletion_positions',
        'all_deletion_coordinates',
        'deletion_positions',
        'deletion_coordinates',
        'deletion_sizes',
        'deletion_n',
        'all_substitution_positions',
        'substitution_positions',
        'all_substitution_values',
        'substitution_values',
        'substitution_n',
        'ref_positions',
        'ref_name',
        'aln_scores',
        'classification',
        'aln_seq',
        'aln_ref',
        'aln_strand',
        'irregular_ends',
        'insertions_outside_window',
        'deletions_outside_window',
        'substitutions_outside_window',
        'total_mods',
        'mods_in_window',
        'mods_outside_window',
    )

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    @property
    def __dict__(self):
        return {key: getattr(self, key) for key in self.__slots__ if hasattr(self, key)}


# order of the flat lists in the C ABI (enum C2_LIST_* in include/crispresso2_amd.h)
_LISTS = ('ref_positions', 'all_insertion_positions', 'all_insertion_left_positions', 'insertion_positions',
          'insertion_coordinates', 'insertion_sizes', 'all_deletion_positions', 'all_deletion_coordinates',
          'deletion_positions', 'deletion_coordinates', 'deletion_sizes', 'all_substitution_positions',
          'all_substitution_values', 'substitution_positions', 'substitution_values')
_PAIRS = ('insertion_coordinates', 'all_deletion_coordinates', 'deletion_coordinates')
_CHARS = ('all_substitution_values', 'substitution_values')


def _classify(read_seq_al, ref_seq_al, _include_indx, legacy):
    br = read_seq_al.encode('utf-8') if isinstance(read_seq_al, str) else bytes(read_seq_al)
    bf = ref_seq_al.encode('utf-8') if isinstance(ref_seq_al, str) else bytes(ref_seq_al)
    n = len(bf)
    if len(br) < n:
        raise IndexError('string index out of range')          # the reference indexes read_seq_al[idx_c]
    inc = np.ascontiguousarray(np.asarray(list(_include_indx), dtype=np.int64).astype(np.int32))
    ctx = _native.default_context()
    cap = 8 * n + 64
    counts = np.zeros(3, dtype=np.int64)
    index = np.zeros(2 * _native.LIST_COUNT, dtype=np.int32)
    needed = ctypes.c_int32(0)
    while True:
        out = np.zeros(cap, dtype=np.int32)
        rc = ctx.lib.c2_find_indels_substitutions(
            ctx.handle, br, bf, n, inc.ctypes.data_as(ctypes.c_void_p), int(inc.size), int(legacy),
            out.ctypes.data_as(ctypes.c_void_p), cap, index.ctypes.data_as(ctypes.c_void_p),
            counts.ctypes.data_as(ctypes.c_void_p), ctypes.byref(needed))
        if rc == _native.E_OVERFLOW:
            cap = needed.value + 64
            continue
        ctx.check(rc, 'c2_find_indels_substitutions')
        break
    res = {}
    for k, name in enumerate(_LISTS):
        o, ln = int(index[2 * k]), int(index[2 * k + 1])
        v = out[o:o + ln].tolist()
        if name in _PAIRS:
            v = [(v[i], v[i + 1]) for i in range(0, ln, 2)]
        elif name in _CHARS:
            v = np.array([chr(c) for c in v])
        res[name] = v
    return res, counts


def _plain_include(_include_indx):
    """what crosses the pipe to a forked worker's helper: the include indices as a plain list (sets and arrays alike)"""
    return _include_indx if isinstance(_include_indx, np.ndarray) else list(_include_indx)


def _payload(res, counts):
    return ResultsSlotsDict(
        all_insertion_positions=res['all_insertion_positions'],
        all_insertion_left_positions=res['all_insertion_left_positions'],
        insertion_positions=res['insertion_positions'],
        insertion_coordinates=res['insertion_coordinates'],
        insertion_sizes=res['insertion_sizes'],
        insertion_n=int(counts[0]),

        all_deletion_positions=res['all_deletion_positions'],
        all_deletion_coordinates=res['all_deletion_coordinates'],
        deletion_positions=res['deletion_positions'],
        deletion_coordinates=res['deletion_coordinates'],
        deletion_sizes=res['deletion_sizes'],
        deletion_n=int(counts[1]),

        all_substitution_positions=res['all_substitution_positions'],
        substitution_positions=res['substitution_positions'],
        all_substitution_values=res['all_substitution_values'],
        substitution_values=res['substitution_values'],
        substitution_n=int(counts[2]),

        ref_positions=res['ref_positions'],
    )


def find_indels_substitutions(read_seq_al, ref_seq_al, _include_indx):
    """Reference pyx:68-187.  Returns a ResultsSlotsDict with the 18 classifier fields."""
    hit = _prime.lookup_payload(read_seq_al, ref_seq_al, _include_indx, 0, _payload)
    if hit is not None:
        return hit
    _prime.stats["per_call_classify"] += 1
    if _native.in_forked_child():
        return _native.forked_child_helper().call("find_indels_substitutions", read_seq_al, ref_seq_al, _plain_include(_include_indx))
    res, counts = _classify(read_seq_al, ref_seq_al, _include_indx, 0)
    return _payload(res, counts)


def find_indels_substitutions_legacy(read_seq_al, ref_seq_al, _include_indx):
    """Reference pyx:190-315 (--use_legacy_insertion_quantification): plain dict; deletion_n / insertion_n are
    numpy sums of the size lists, as in the reference (np.sum([]) is the float 0.0)."""
    hit = _prime.lookup_payload(read_seq_al, ref_seq_al, _include_indx, 1, _payload_legacy)
    if hit is not None:
        return hit
    _prime.stats["per_call_classify"] += 1
    if _native.in_forked_child():
        return _native.forked_child_helper().call("find_indels_substitutions_legacy", read_seq_al, ref_seq_al, _plain_include(_include_indx))
    res, counts = _classify(read_seq_al, ref_seq_al, _include_indx, 1)
    return _payload_legacy(res, counts)


def _payload_legacy(res, counts):
    return {
        'all_insertion_positions': res['all_insertion_positions'],
        'all_insertion_left_positions': res['all_insertion_left_positions'],
        'insertion_positions': res['insertion_positions'],
        'insertion_coordinates': res['insertion_coordinates'],
        'insertion_sizes': res['insertion_sizes'],
        'insertion_n': np.sum(res['insertion_sizes']),
        'all_deletion_positions': res['all_deletion_positions'],

        'deletion_positions': res['deletion_positions'],
        'deletion_coordinates': res['deletion_coordinates'],
        'all_deletion_coordinates': res['all_deletion_coordinates'],
        'deletion_sizes': res['deletion_sizes'],
        'deletion_n': np.sum(res['deletion_sizes']),

        'all_substitution_positions': res['all_substitution_positions'],
        'substitution_positions': res['substitution_positions'],
        'all_substitution_values': res['all_substitution_values'],
        'substitution_values': res['substitution_values'],
        'substitution_n': int(counts[2]),

        'ref_positions': res['ref_positions'],
    }


def find_indels_substitutions_batch(pairs, include_sets, set_ids=None, legacy=False, ctx=None):
    """The classifier calls of a whole batch of alignments in a few launches (c2_classify_lists_batch): `pairs` is a sequence
    of (read_seq_al, ref_seq_al), `include_sets` a list of include-index collections and set_ids[k] says which of them
    pair k uses (default: set 0).  Returns what find_indels_substitutions (or _legacy) returns for every pair, in order."""
    n = len(pairs)
    if n == 0:
        return []
    ctx = ctx or _native.default_context()
    enc = []
    for a, b in pairs:
        br = a.encode('utf-8') if isinstance(a, str) else bytes(a)
        bf = b.encode('utf-8') if isinstance(b, str) else bytes(b)
        if len(br) < len(bf):
            raise IndexError('string index out of range')
        enc.append((br, bf))
    lens = np.array([len(bf) for _, bf in enc], dtype=np.int32)
    stride = max(16, (int(lens.max()) + 15) // 16 * 16)
    a1 = np.zeros((n, stride), dtype=np.uint8)
    a2 = np.zeros((n, stride), dtype=np.uint8)
    for k, (br, bf) in enumerate(enc):
        a1[k, :len(bf)] = np.frombuffer(br, dtype=np.uint8)[:len(bf)]
        a2[k, :len(bf)] = np.frombuffer(bf, dtype=np.uint8)
    index, values, counts = ctx.classify_lists_batch(a1, a2, lens, set_ids, include_sets, legacy=legacy)
    out = []
    for t in range(n):
        res = {}
        base = t * _native.LIST_COUNT
        for k, name in enumerate(_LISTS):
            v = values[index[base + k]:index[base + k + 1]].tolist()
            if name in _PAIRS:
                v = [(v[i], v[i + 1]) for i in range(0, len(v), 2)]
            elif name in _CHARS:
                v = np.array([chr(c) for c in v])
            res[name] = v
        out.append(_payload_legacy(res, counts[t]) if legacy else _payload(res, counts[t]))
    return out


def calculate_homology(a, b):
    """Reference pyx:318-327: fraction of positions of `a` (bytes, up to its first NUL) equal to `b`."""
    if not isinstance(a, (bytes, bytearray)) or not isinstance(b, (bytes, bytearray)):
        raise TypeError('expected bytes, %s found' % type(a if not isinstance(a, (bytes, bytearray)) else b).__name__)
    a = bytes(a).split(b'\0', 1)[0]
    n = len(a)
    if n == 0:
        raise ZeroDivisionError('float division')
    b = bytes(b)[:n].ljust(n, b'\0')
    if _native.in_forked_child():
        return _native.forked_child_helper().call("calculate_homology", a, b)
    ctx = _native.default_context()
    out = ctypes.c_double(0)
    ctx.check(ctx.lib.c2_calculate_homology(ctx.handle, a, b, n, ctypes.byref(out)), 'c2_calculate_homology')
    return out.value
