"""TEST INFRASTRUCTURE ONLY -- runs crispresso2_amd.pipeline's host logic on a machine without a GPU by redirecting its three
device entry points to the wave emulator (tests/emu: the same HIP kernel source compiled for the host):

    torch "cuda" tensors        -> CPU tensors (their data_ptr()s are host addresses the emulator can write to)
    BatchAligner.align_device   -> emu_driver.align_batch over the bytes behind those addresses (default launch chain)
    counts.accumulate_device    -> emu_driver.count_vectors, added into the tensor behind d_counts

    BatchAligner.align / Context.classify_lists_batch (the per-read route of variants.py) -> the same emulator entry points

Everything else -- native ingest, strand plans, strand / best-amplicon selection, reverse-complement merge, weights, the
first-amplicon view, statistics, allele rows, the per-read dicts and their files -- is the product's own code.  Used by tests only; the product never imports this."""
import contextlib
import ctypes

import numpy as np

import emu_driver as E


def _view(addr, nbytes):
    return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(addr))


class EmulatedAligner:
    def __init__(self, seqs, gap_incentives, include_idxs, matrix, gap_open, gap_extend, ctx=None, device=None):
        self.seqs, self.g, self.inc = list(seqs), [np.asarray(x, dtype=np.int64) for x in gap_incentives], [list(x) for x in include_idxs]
        self.m, self.go, self.ge = matrix, gap_open, gap_extend
        self.max_ref_len = max(len(s) for s in seqs)

    def stride_for(self, max_read_len):
        return (self.max_ref_len + int(max_read_len) + 15) // 16 * 16

    def align_device(self, n_reads, d_reads, d_offsets, d_aln_read, d_aln_ref, d_records, aln_stride, max_read_len,
                     d_ref_ids=None, d_strands=None, all_refs=False, stream=None, legacy=False, min_read_len=0, d_hints=None):
        import os
        if legacy:
            os.environ["C2_EMU_LEGACY"] = "1"
        else:
            os.environ.pop("C2_EMU_LEGACY", None)
        n, k = int(n_reads), len(self.seqs)
        off = _view(d_offsets, 8 * (n + 1)).view(np.int64)
        arena = _view(d_reads, max(int(off[-1]), 1)).tobytes()
        reads = [arena[int(off[i]):int(off[i + 1])].decode() for i in range(n)]
        ntasks = n * k if all_refs else n
        strands = None if d_strands is None else _view(d_strands, ntasks).copy()
        rids = None if d_ref_ids is None else _view(d_ref_ids, 2 * n).view(np.int16).astype(np.uint16)
        st = {"want_hints": bool(d_hints)}
        _, rec = E.align_batch(reads, self.seqs, self.g, self.inc, self.m, self.go, self.ge, ref_ids=rids, strands=strands,
                               all_refs=all_refs, band_lanes=-87, stats=st)
        if d_hints:
            _view(d_hints, 16 * ntasks).view(np.uint32)[:] = st["hints"].reshape(-1)
        o1, o2 = st["raw"]
        w = min(o1.shape[1], aln_stride)
        assert int(rec["aln_len"].max()) <= w
        a = _view(d_aln_read, ntasks * aln_stride).reshape(ntasks, aln_stride)
        f = _view(d_aln_ref, ntasks * aln_stride).reshape(ntasks, aln_stride)
        a[:, :w] = o1[:, :w]
        f[:, :w] = o2[:, :w]
        _view(d_records, 32 * ntasks)[:] = rec.view(np.uint8).reshape(-1)


    # the host-memory API of the per-read route (variants.py, paired.py): BatchAligner.align -> BatchResult
    def align(self, reads, ref_ids=None, strands=None, all_refs=False):
        from crispresso2_amd.batch import BatchResult
        if isinstance(reads, tuple):
            arena, off = reads
            buf = np.asarray(arena, dtype=np.uint8).tobytes()
            reads = [buf[int(off[i]):int(off[i + 1])].decode() for i in range(len(off) - 1)]
        st = {}
        _, rec = E.align_batch(list(reads), self.seqs, self.g, self.inc, self.m, self.go, self.ge, ref_ids=ref_ids, strands=strands,
                               all_refs=all_refs, band_lanes=-87, stats=st)
        o1, o2 = st["raw"]
        return BatchResult(o1, o2, rec.view(_native_rec_dtype()), len(reads), len(self.seqs), all_refs)


def _native_rec_dtype():
    from crispresso2_amd import _native
    return _native.REC_DTYPE


class _EmulatedLib:
    @staticmethod
    def c2_consensus_pairs_batch(handle, *args):                    # same arguments as the emulator's entry point, minus the context
        return E.lib().emu_consensus_pairs(*args)


class EmulatedContext:
    """stands in for _native.Context where the per-read routes ask it for the batched classifier (c2_classify_lists_batch) and the
    paired-read consensus kernel (c2_consensus_pairs_batch)"""
    lib = _EmulatedLib()
    handle = None

    @staticmethod
    def check(rc, what):
        assert rc == 0, (what, rc)

    def classify_lists_batch(self, aln_read, aln_ref, lens, set_ids, include_sets, legacy=False):
        a1 = np.ascontiguousarray(aln_read, dtype=np.uint8)
        a2 = np.ascontiguousarray(aln_ref, dtype=np.uint8)
        n, stride = a1.shape
        ln = np.ascontiguousarray(lens, dtype=np.int32)
        ids = np.zeros(n, dtype=np.uint16) if set_ids is None else np.ascontiguousarray(set_ids, dtype=np.uint16)
        sets = [np.array(sorted(set(int(x) for x in inc)), dtype=np.int32) for inc in include_sets]
        off = np.zeros(len(sets) + 1, dtype=np.int64)
        off[1:] = np.cumsum([x.size for x in sets])
        flat = np.ascontiguousarray(np.concatenate(sets + [np.zeros(1, dtype=np.int32)]), dtype=np.int32)
        index = np.zeros(n * E.N_LISTS + 1, dtype=np.int64)
        cap = int(ln.sum()) * 12 + 1024
        values = np.zeros(cap, dtype=np.int32)
        counts = np.zeros((n, 3), dtype=np.int64)
        rc = E.lib().emu_classify_lists_batch(ctypes.c_uint64(n), a1.ctypes.data_as(ctypes.c_void_p), a2.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_uint32(stride), ln.ctypes.data_as(ctypes.c_void_p), ids.ctypes.data_as(ctypes.c_void_p),
                                              flat.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), int(bool(legacy)),
                                              index.ctypes.data_as(ctypes.c_void_p), values.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cap),
                                              counts.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, rc
        return index, values[:int(index[-1])].copy(), counts


def _accumulate(aligners):
    def accumulate_device(ctx, layout, n_tasks, d_aln_read, d_aln_ref, aln_stride, d_records, d_counts, d_weights=None,
                          min_matches=None, flags=0, stream=None, d_hints=None):
        al = aligners[-1]
        a = _view(d_aln_read, n_tasks * aln_stride).reshape(n_tasks, aln_stride)
        f = _view(d_aln_ref, n_tasks * aln_stride).reshape(n_tasks, aln_stride)
        rec = _view(d_records, 32 * n_tasks).view(E.REC_DTYPE).reshape(-1)
        w = None if not d_weights else _view(d_weights, 4 * n_tasks).view(np.uint32).copy()
        hints = None if not d_hints else _view(d_hints, 16 * n_tasks).view(np.uint32).copy().reshape(n_tasks, 4)
        counts, lay = E.count_vectors(a, f, rec, al.seqs, al.inc, layout.hl - layout.lmax - 2,
                                      weights=w, min_matches=min_matches, flags=flags, hints=hints)
        assert lay.shape() == layout.shape(), (lay.shape(), layout.shape())
        n64 = int(np.prod(layout.shape()))
        _view(d_counts, 8 * n64).view(np.int64)[:] += counts.reshape(-1)
    return accumulate_device


def _select(ctx, n_reads, n_refs, d_records, min_mscore, mode, max_aln_len, d_records2=None, d_slot2=None, d_raw_counts=None,
            d_counts=None, d_member=None, d_use2=None, d_flags=None, d_weights=None, d_weights2=None, d_stats=None, stream=None):
    """counts.select_best_device -> c2_select_best_kernel on the emulator (same arguments)"""
    assert max_aln_len < 8000
    mm = np.ascontiguousarray(min_mscore, dtype=np.uint32)
    P = lambda x: ctypes.c_void_p(x or 0)
    rc = E.lib().emu_select_best(ctypes.c_uint64(n_reads), int(n_refs), P(d_records), P(d_records2), P(d_slot2),
                                 mm.ctypes.data_as(ctypes.c_void_p), P(d_raw_counts), P(d_counts), int(mode), P(d_member), P(d_use2),
                                 P(d_flags), P(d_weights), P(d_weights2), P(d_stats))
    assert rc == 0


def _strand_plan(ctx, n_reads, d_reads, d_offsets, max_read_len, refs, ref_names, seed_count, seed_min, d_plan, stream=None):
    """counts.strand_plan_device -> c2_strand_plan_kernel on the emulator (same arguments)"""
    from crispresso2_amd import counts as C
    blob, off, ln, ns, S = C.seed_tables(refs, ref_names, seed_count)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else None
    rc = E.lib().emu_strand_plan(ctypes.c_uint64(int(n_reads)), ctypes.c_void_p(d_reads), ctypes.c_void_p(d_offsets), int(max_read_len),
                                 len(ref_names), int(S), P(ns), P(blob), int(blob.size), P(off), P(ln), int(seed_min), ctypes.c_void_p(d_plan))
    assert rc == 0


class _EmuAlleleCalls:
    """alleles.CALLS -> the emulator's allele-table entry points (the product's c2_alleles_host.h over an emulator backend)"""
    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            L = E.lib()
            L.emu_allele_last_error.restype = ctypes.c_char_p
            raise RuntimeError("%s: %s" % (what, (L.emu_allele_last_error() or b"").decode()))

    @staticmethod
    def build(ctx, src, stream):
        h = ctypes.c_void_p()
        _EmuAlleleCalls._chk(E.lib().emu_allele_table_build(ctypes.byref(src), ctypes.byref(h)), "emu_allele_table_build")
        return h

    @staticmethod
    def rows(ctx, h):
        E.lib().emu_allele_table_rows.restype = ctypes.c_uint64
        return int(E.lib().emu_allele_table_rows(h))

    @staticmethod
    def write(ctx, h, path, labels, n_total, probes, threads):
        nb = ctypes.c_uint64()
        _EmuAlleleCalls._chk(E.lib().emu_allele_table_write(h, path, labels, ctypes.c_int64(n_total), probes, int(threads), ctypes.byref(nb)), "emu_allele_table_write")
        return int(nb.value)

    @staticmethod
    def write_zip(ctx, h, zip_path, member, labels, n_total, probes, threads, level):
        nb, nz = ctypes.c_uint64(), ctypes.c_uint64()
        _EmuAlleleCalls._chk(E.lib().emu_allele_table_write_zip(h, zip_path, member, labels, ctypes.c_int64(n_total), probes, int(threads), int(level),
                                                                ctypes.byref(nb), ctypes.byref(nz)), "emu_allele_table_write_zip")
        return int(nb.value), int(nz.value)

    @staticmethod
    def fetch(ctx, h, rows, aligned, reference, stride):
        _EmuAlleleCalls._chk(E.lib().emu_allele_table_fetch(h, rows, aligned, reference, ctypes.c_uint32(stride)), "emu_allele_table_fetch")

    @staticmethod
    def around_cut_write(ctx, h, label, cut_point, ref_len, plot_window_size, n_total, path, threads):
        ng = ctypes.c_uint64()
        _EmuAlleleCalls._chk(E.lib().emu_allele_table_around_cut_write(h, int(label), int(cut_point), int(ref_len), int(plot_window_size), ctypes.c_int64(n_total),
                                                                       path, int(threads), ctypes.byref(ng)), "emu_allele_table_around_cut_write")
        return int(ng.value)

    @staticmethod
    def free(ctx, h):
        E.lib().emu_allele_table_free.restype = None
        E.lib().emu_allele_table_free(h)


def _emu_consensus_device(ctx, n, s1, f1, s2, f2, stride, n1, n2, q1, q2, qstride, lq1, lq2, best1, oa, orf, oq, ostride, info, stream):
    """paired_device.consensus_device -> c2_consensus_pairs_kernel on the emulator (same arguments; the addresses are host addresses here)"""
    V = ctypes.c_void_p
    rc = E.lib().emu_consensus_pairs(ctypes.c_uint64(n), V(s1), V(f1), V(s2), V(f2), ctypes.c_uint32(stride), V(n1), V(n2), V(q1), V(q2), ctypes.c_uint32(qstride),
                                     V(lq1), V(lq2), V(best1), V(oa), V(orf), V(oq), ctypes.c_uint32(ostride), V(info))
    assert rc == 0


def _emu_classify_records_device(ctx, n, aln_read, aln_ref, stride, info, ref_ids, strands, legacy, records, stream, refs=None, ref_names=None):
    """paired_device.classify_records_device -> c2_classify_records_kernel on the emulator"""
    V = ctypes.c_void_p
    nrefs = len(ref_names)
    lens = np.array([len(refs[nm]['sequence']) for nm in ref_names], dtype=np.int32)
    inc = [np.ascontiguousarray(np.asarray(sorted(set(int(x) for x in refs[nm]['include_idxs'])), dtype=np.int64).astype(np.int32)) for nm in ref_names]
    ip = (ctypes.c_void_p * nrefs)(*[x.ctypes.data for x in inc])
    ninc = np.array([len(x) for x in inc], dtype=np.int32)
    rc = E.lib().emu_classify_records(ctypes.c_uint64(n), V(aln_read), V(aln_ref), ctypes.c_uint32(stride), V(info), V(ref_ids or 0), V(strands or 0), int(bool(legacy)),
                                      V(records), nrefs, lens.ctypes.data_as(V), ip, ninc.ctypes.data_as(V))
    assert rc == 0


@contextlib.contextmanager
def emulated_device(made=None):
    """Inside the block pipeline.quantify_* run on the emulator; restored afterwards.  made: the caller's list of emulated aligners
    (bench_on_emulator shares its own, so that the count pass always sees the aligner of the batch it counts)."""
    import torch
    from crispresso2_amd import pipeline, variants, paired, counts as C, _native, alleles, paired_device
    made = [] if made is None else made
    saved_paired_device = (paired_device.consensus_device, paired_device.classify_records_device)
    paired_device.consensus_device, paired_device.classify_records_device = _emu_consensus_device, _emu_classify_records_device
    saved_allele_calls = alleles.CALLS
    alleles.CALLS = _EmuAlleleCalls

    def make_aligner(*a, **kw):
        made.append(EmulatedAligner(*a, **kw))
        return made[-1]

    class _Stream:
        cuda_stream = 0
    import crispresso2_amd.batch as _batch_mod
    saved_batch_aligner = _batch_mod.BatchAligner
    _batch_mod.BatchAligner = make_aligner                            # (paired_device imports it from there at call time)
    saved = (torch.device, torch.cuda.current_stream, torch.cuda.synchronize, pipeline.BatchAligner, C.accumulate_device, _native.default_context)
    saved_select = C.select_best_device
    C.select_best_device = _select
    saved_strand = C.strand_plan_device
    C.strand_plan_device = _strand_plan
    saved_variants_aligner, saved_paired_aligner = variants.BatchAligner, paired.BatchAligner
    variants.BatchAligner = paired.BatchAligner = make_aligner
    real_device = torch.device
    torch.device = lambda *a, **k: real_device("cpu")
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.synchronize = lambda *a, **k: None
    pipeline.BatchAligner = make_aligner
    C.accumulate_device = _accumulate(made)
    _native.default_context = lambda *a, **k: EmulatedContext()
    try:
        yield
    finally:
        torch.device, torch.cuda.current_stream, torch.cuda.synchronize, pipeline.BatchAligner, C.accumulate_device, _native.default_context = saved
        _batch_mod.BatchAligner = saved_batch_aligner
        variants.BatchAligner, paired.BatchAligner = saved_variants_aligner, saved_paired_aligner
        C.select_best_device = saved_select
        C.strand_plan_device = saved_strand
        alleles.CALLS = saved_allele_calls
        paired_device.consensus_device, paired_device.classify_records_device = saved_paired_device
