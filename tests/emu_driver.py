"""ctypes driver for the wave emulator (tests/emu/) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_DIR = os.path.join(HERE, "emu")
LIB = os.path.join(EMU_DIR, "libc2_emu.so")

REC_DTYPE = np.dtype([
    ("aln_len", "<u2"), ("matches", "<u2"), ("insertion_n", "<u2"), ("deletion_n", "<u2"), ("substitution_n", "<u2"),
    ("all_insertion_events", "<u2"), ("win_insertion_events", "<u2"), ("all_deletion_events", "<u2"),
    ("win_deletion_events", "<u2"), ("all_deletion_bases", "<u2"), ("all_substitutions", "<u2"),
    ("irregular_ends", "u1"), ("status", "u1"), ("strand", "u1"), ("reserved0", "u1"),
    ("ref_id", "<u2"), ("reserved2", "<u4")])
assert REC_DTYPE.itemsize == 32

_lib = None


def build(force=False):
    srcs = [os.path.join(EMU_DIR, f) for f in ("emu_harness.cpp", "emu_runtime.h", "hip/hip_runtime.h")]
    srcs += [os.path.join(ROOT, "crispresso2_amd/csrc", f) for f in ("c2_kernels.hip", "c2_k_common.h", "c2_k_align.hip", "c2_k_classify.hip", "c2_k_select.hip",
                                                                      "c2_k_count.hip", "c2_k_fastq.hip", "c2_k_alleles.hip", "c2_alleles_host.h",
                                                                      "c2_device.h", "c2_host_prep.h")]
    srcs.append(os.path.join(ROOT, "include/crispresso2_amd.h"))
    def stale():
        return not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if force or stale():
        # one builder at a time (pytest-xdist workers all get here after a source change), and the library appears atomically
        import fcntl
        with open(LIB + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                tmp = LIB + ".tmp.%d" % os.getpid()
                subprocess.check_call(["g++", "-O1", "-U_FORTIFY_SOURCE", "-D_FORTIFY_SOURCE=0", "-std=c++17", "-shared", "-fPIC"] + os.environ.get("C2_EMU_CFLAGS", "").split() + ["-I", EMU_DIR,
                                       "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "crispresso2_amd/csrc"),
                                       "-x", "c++", os.path.join(EMU_DIR, "emu_harness.cpp"), "-o", tmp, "-lz"])
                os.replace(tmp, LIB)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def align_batch(reads, refs, gap_incentives, includes, matrix, gap_open, gap_extend,
                ref_ids=None, strands=None, all_refs=False, force_R=0, grid=0, no_packed=False, band_lanes=0, stats=None):
    """reads: list[str]; refs: list[str]; returns (list[(s1, s2)], records ndarray)"""
    n = len(reads)
    arena = "".join(reads).encode()
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    nrefs = len(refs)
    seqs = (ctypes.c_char_p * nrefs)(*[r.encode() for r in refs])
    lens = np.array([len(r) for r in refs], dtype=np.int32)
    g = [np.ascontiguousarray(x, dtype=np.int64) for x in gap_incentives]
    gp = (ctypes.c_void_p * nrefs)(*[x.ctypes.data for x in g])
    inc = [np.ascontiguousarray(np.asarray(list(x), dtype=np.int64).astype(np.int32)) for x in includes]
    ip = (ctypes.c_void_p * nrefs)(*[x.ctypes.data for x in inc])
    ninc = np.array([len(x) for x in inc], dtype=np.int32)
    m = np.ascontiguousarray(matrix, dtype=np.int64)
    ntasks = n * (nrefs if all_refs else 1)
    stride = (max(len(r) for r in reads) + int(lens.max()) + 15) // 16 * 16
    o1 = np.zeros((ntasks, stride), dtype=np.uint8)
    o2 = np.zeros((ntasks, stride), dtype=np.uint8)
    rec = np.zeros(ntasks, dtype=REC_DTYPE)
    rid = None if ref_ids is None else np.ascontiguousarray(ref_ids, dtype=np.uint16)
    st = None if strands is None else np.ascontiguousarray(strands, dtype=np.uint8)
    hints = None
    if stats is not None and stats.get("want_hints"):                # c2_batch.diag_hints: one word per task
        hints = np.full((ntasks, 4), 0xdeadbeef, dtype=np.uint32)       # four words per task
        lib().emu_set_hints_out(hints.ctypes.data_as(ctypes.c_void_p))
    nfb = ctypes.c_int(0)
    rc = lib().emu_align_batch(
        ctypes.c_uint64(n), arena, offs.ctypes.data_as(ctypes.c_void_p),
        None if rid is None else rid.ctypes.data_as(ctypes.c_void_p),
        None if st is None else st.ctypes.data_as(ctypes.c_void_p), int(all_refs),
        nrefs, seqs, lens.ctypes.data_as(ctypes.c_void_p), gp, ip, ninc.ctypes.data_as(ctypes.c_void_p),
        m.ctypes.data_as(ctypes.c_void_p), int(m.shape[0]), int(gap_open), int(gap_extend),
        o1.ctypes.data_as(ctypes.c_void_p), o2.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(stride),
        rec.ctypes.data_as(ctypes.c_void_p), int(force_R), ctypes.c_uint(grid), int(no_packed), int(band_lanes), ctypes.byref(nfb))
    if stats is not None:
        stats['unpaired'] = stats.get('unpaired', 0) + int(lib().emu_last_unpaired())
        stats['pk_beta'], stats['pk_bias'] = int(lib().emu_last_pk_beta()), int(lib().emu_last_pk_bias())
        cls, p16 = (ctypes.c_uint32 * 7)(), ctypes.c_uint32(0)
        lib().emu_last_partition(cls, ctypes.byref(p16))
        stats['classes'] = [a + int(b) for a, b in zip(stats.get('classes', [0] * 7), cls)]
        stats['p16_finished'] = stats.get('p16_finished', 0) + int(p16.value)
        stats['exact_copies'] = stats.get('exact_copies', 0) + int(lib().emu_last_exact_copies())
        stats['fallback'] = stats.get('fallback', 0) + max(nfb.value, 0)
        stats['tasks'] = stats.get('tasks', 0) + ntasks
    assert rc == 0, rc
    out = []
    for k in range(ntasks):
        L = int(rec["aln_len"][k])
        out.append((o1[k, :L].tobytes().decode(), o2[k, :L].tobytes().decode()))
    if stats is not None:
        stats["raw"] = (o1, o2)
        if hints is not None:
            stats["hints"] = hints
    return out, rec


N_LISTS = 15


def classify_lists(read_al, ref_al, include, legacy=False):
    n = len(ref_al)
    inc = np.ascontiguousarray(sorted(set(int(x) for x in include)), dtype=np.int32)
    cap = 2 * n + 8
    while True:
        lists = np.zeros((N_LISTS, cap), dtype=np.int32)
        lens = np.zeros(N_LISTS, dtype=np.int32)
        counts = np.zeros(3, dtype=np.int64)
        lib().emu_classify_lists(read_al.encode(), ref_al.encode(), n, inc.ctypes.data_as(ctypes.c_void_p), len(inc),
                                 int(legacy), cap, lists.ctypes.data_as(ctypes.c_void_p),
                                 lens.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p))
        if lens.max() <= cap:
            break
        cap = int(lens.max()) + 8
    return [lists[k, :lens[k]].tolist() for k in range(N_LISTS)], counts.tolist()


def classify_lists_batch(pairs, includes, legacy=False):
    """pairs: [(read_al, ref_al)], includes: one include collection per pair -> per pair ([15 lists], [3 counts])"""
    n = len(pairs)
    lens = np.array([len(b) for _, b in pairs], dtype=np.int32)
    stride = int(lens.max()) + 3
    a1 = np.zeros((n, stride), dtype=np.uint8); a2 = np.zeros((n, stride), dtype=np.uint8)
    for k, (a, b) in enumerate(pairs):
        a1[k, :len(b)] = np.frombuffer(a.encode(), dtype=np.uint8)[:len(b)]
        a2[k, :len(b)] = np.frombuffer(b.encode(), dtype=np.uint8)
    sets = [np.array(sorted(set(int(x) for x in inc)), dtype=np.int32) for inc in includes]
    off = np.zeros(n + 1, dtype=np.int64); off[1:] = np.cumsum([x.size for x in sets])
    flat = np.ascontiguousarray(np.concatenate(sets + [np.zeros(1, dtype=np.int32)]), dtype=np.int32)
    ids = np.arange(n, dtype=np.uint16)
    index = np.zeros(n * N_LISTS + 1, dtype=np.int64)
    cap = int(lens.sum()) * 12 + 1024
    values = np.zeros(cap, dtype=np.int32)
    counts = np.zeros((n, 3), dtype=np.int64)
    rc = lib().emu_classify_lists_batch(ctypes.c_uint64(n), a1.ctypes.data_as(ctypes.c_void_p), a2.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_uint32(stride), lens.ctypes.data_as(ctypes.c_void_p), ids.ctypes.data_as(ctypes.c_void_p),
                                        flat.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), int(legacy),
                                        index.ctypes.data_as(ctypes.c_void_p), values.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cap),
                                        counts.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return [([values[index[t * N_LISTS + k]:index[t * N_LISTS + k + 1]].tolist() for k in range(N_LISTS)], counts[t].tolist()) for t in range(n)]


def consensus_pairs(items):
    """items: (aln_seq_r1, aln_ref_r1, score_r1, qual_r1, aln_seq_r2, aln_ref_r2, score_r2, qual_r2) -> per item
    (final_aln, final_qual, final_ref, matching columns, caching_is_ok, index_error)"""
    n = len(items)

    def rows(strs, stride):
        a = np.zeros((n, stride), dtype=np.uint8)
        for k, x in enumerate(strs):
            a[k, :len(x)] = np.frombuffer(x.encode(), dtype=np.uint8)
        return a
    n1 = np.array([len(it[1]) for it in items], dtype=np.int32); n2 = np.array([len(it[5]) for it in items], dtype=np.int32)
    lq1 = np.array([len(it[3]) for it in items], dtype=np.int32); lq2 = np.array([len(it[7]) for it in items], dtype=np.int32)
    stride = int(max(n1.max(), n2.max())) + 1; qstride = int(max(lq1.max(), lq2.max())) + 1; ostride = 2 * stride
    s1, f1 = rows([it[0] for it in items], stride), rows([it[1] for it in items], stride)
    s2, f2 = rows([it[4] for it in items], stride), rows([it[5] for it in items], stride)
    q1, q2 = rows([it[3] for it in items], qstride), rows([it[7] for it in items], qstride)
    best1 = np.array([1 if it[2] >= it[6] else 0 for it in items], dtype=np.uint8)
    oa = np.zeros((n, ostride), dtype=np.uint8); orf = np.zeros((n, ostride), dtype=np.uint8); oq = np.zeros((n, ostride), dtype=np.uint8)
    info = np.zeros((n, 4), dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib().emu_consensus_pairs(ctypes.c_uint64(n), p(s1), p(f1), p(s2), p(f2), ctypes.c_uint32(stride), p(n1), p(n2), p(q1), p(q2),
                                   ctypes.c_uint32(qstride), p(lq1), p(lq2), p(best1), p(oa), p(orf), p(oq), ctypes.c_uint32(ostride), p(info))
    assert rc == 0
    return [(oa[k, :info[k, 0]].tobytes().decode(), oq[k, :info[k, 1]].tobytes().decode(), orf[k, :info[k, 0]].tobytes().decode(),
             int(info[k, 2]), bool(info[k, 3] & 1), bool(info[k, 3] & 2)) for k in range(n)]


def count_vectors(aln_read, aln_ref, records, ref_seqs, includes, max_read_len, weights=None, min_matches=None, flags=0, grid=2, hints=None):
    """aln_read/aln_ref: uint8 [n, stride]; records: REC_DTYPE [n]; ref_seqs: the reference strings.
    -> (counts int64 [n_refs, per_ref], layout)"""
    ref_lens = [len(x) for x in ref_seqs]
    seq_ptrs = (ctypes.c_char_p * len(ref_seqs))(*[x.encode() for x in ref_seqs])
    import sys
    sys.path.insert(0, ROOT)
    from crispresso2_amd.counts import CountLayout
    n = len(records)
    nrefs = len(ref_lens)
    lay = CountLayout(nrefs, max(ref_lens), max_read_len)
    counts = np.zeros(lay.shape(), dtype=np.int64)
    lens = np.array(ref_lens, dtype=np.int32)
    inc = [np.ascontiguousarray(np.asarray(list(x), dtype=np.int64).astype(np.int32)) for x in includes]
    ip = (ctypes.c_void_p * nrefs)(*[x.ctypes.data for x in inc])
    ninc = np.array([len(x) for x in inc], dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
    mm = None if min_matches is None else np.ascontiguousarray(min_matches, dtype=np.uint16)
    a1 = np.ascontiguousarray(aln_read); a2 = np.ascontiguousarray(aln_ref); rec = np.ascontiguousarray(records)
    hw = None
    if hints is not None:                                            # c2_count_vectors_hinted_device
        hw = np.ascontiguousarray(hints, dtype=np.uint32)
        lib().emu_set_count_hints(hw.ctypes.data_as(ctypes.c_void_p))
    rc = lib().emu_count_vectors(ctypes.c_uint64(n), a1.ctypes.data_as(ctypes.c_void_p), a2.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_uint32(a1.shape[1]), rec.ctypes.data_as(ctypes.c_void_p),
                                 None if w is None else w.ctypes.data_as(ctypes.c_void_p),
                                 None if mm is None else mm.ctypes.data_as(ctypes.c_void_p), 0 if mm is None else mm.shape[1] - 1,
                                 nrefs, lens.ctypes.data_as(ctypes.c_void_p), ip, ninc.ctypes.data_as(ctypes.c_void_p),
                                 int(flags), int(lay.hl), counts.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(grid), seq_ptrs)
    assert rc == 0
    return counts, lay
