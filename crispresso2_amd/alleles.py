"""The allele frequency table, built, sorted and printed on the device (c2_allele_table_* of include/crispresso2_amd.h).

Reference: the per-variant loop that fills `alleles_list` (CRISPRessoCORE.py:3964-4010), `df_alleles`' sort and %Reads (:4298-4303),
Alleles_frequency_table.txt (:4498-4530) and <ref>Alleles_frequency_table_around_<guide>.txt (CRISPRessoShared.py:1513-1531,
CRISPRessoCORE.py:5250-5273).  Input: what pipeline.quantify_* left in HBM -- aligned strings, records, selection masks, merged
multiplicities.  The host sees the finished files, or (AlleleTable.rows) the sorted rows as numpy columns; there is no loop over rows
in Python anywhere on this route.
"""
import ctypes
import os

import numpy as np

ROW_DTYPE = np.dtype([("src", "<u4"), ("reads", "<u4"), ("read", "<u4"), ("aln_len", "<u2"), ("label", "<u2"), ("n_deleted", "<u2"),
                      ("n_inserted", "<u2"), ("n_mutated", "<u2"), ("modified", "u1"), ("reserved", "u1")])
assert ROW_DTYPE.itemsize == 24


class AlleleSrc(ctypes.Structure):
    """struct c2_allele_src"""
    _fields_ = [("n_reads", ctypes.c_uint64), ("n_refs", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("d_aln_read1", ctypes.c_void_p), ("d_aln_ref1", ctypes.c_void_p), ("d_records1", ctypes.c_void_p),
                ("d_aln_read2", ctypes.c_void_p), ("d_aln_ref2", ctypes.c_void_p), ("d_records2", ctypes.c_void_p),
                ("d_slot2", ctypes.c_void_p), ("d_member", ctypes.c_void_p), ("d_use2", ctypes.c_void_p), ("d_flags", ctypes.c_void_p),
                ("d_counts", ctypes.c_void_p), ("d_scaffold_hit", ctypes.c_void_p),
                ("stride1", ctypes.c_uint32), ("stride2", ctypes.c_uint32), ("scaffold_ref", ctypes.c_int32), ("flags", ctypes.c_uint32)]


def labels_for(ref_names):
    """the label list c2_allele_row.label indexes: reference names, their AMBIGUOUS_ / DISCARDED_ forms, the two scaffold labels"""
    return (list(ref_names) + ['AMBIGUOUS_' + r for r in ref_names] + ['DISCARDED_' + r for r in ref_names] +
            ['Scaffold-incorporated', 'DISCARDED_Scaffold-incorporated'])


class _NativeCalls:
    """the C ABI through ctypes (tests/pipeline_on_emulator.py puts the wave emulator's entry points here instead)"""
    @staticmethod
    def build(ctx, src, stream):
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.c2_allele_table_build(ctx.handle, ctypes.byref(src), ctypes.byref(h), ctypes.c_void_p(stream or 0)), "c2_allele_table_build")
        return h

    @staticmethod
    def rows(ctx, h):
        ctx.lib.c2_allele_table_rows.restype = ctypes.c_uint64
        return int(ctx.lib.c2_allele_table_rows(h))

    @staticmethod
    def write(ctx, h, path, labels, n_total, probes, threads):
        nb = ctypes.c_uint64()
        ctx.check(ctx.lib.c2_allele_table_write(h, path, labels, ctypes.c_int64(n_total), probes, int(threads), ctypes.byref(nb)), "c2_allele_table_write")
        return int(nb.value)

    @staticmethod
    def write_zip(ctx, h, zip_path, member, labels, n_total, probes, threads, level):
        nb, nz = ctypes.c_uint64(), ctypes.c_uint64()
        ctx.check(ctx.lib.c2_allele_table_write_zip(h, zip_path, member, labels, ctypes.c_int64(n_total), probes, int(threads), int(level),
                                                    ctypes.byref(nb), ctypes.byref(nz)), "c2_allele_table_write_zip")
        return int(nb.value), int(nz.value)

    @staticmethod
    def fetch(ctx, h, rows, aligned, reference, stride):
        ctx.check(ctx.lib.c2_allele_table_fetch(h, rows, aligned, reference, ctypes.c_uint32(stride)), "c2_allele_table_fetch")

    @staticmethod
    def around_cut_write(ctx, h, label, cut_point, ref_len, plot_window_size, n_total, path, threads):
        ng = ctypes.c_uint64()
        ctx.check(ctx.lib.c2_allele_table_around_cut_write(h, int(label), int(cut_point), int(ref_len), int(plot_window_size), ctypes.c_int64(n_total),
                                                           path, int(threads), ctypes.byref(ng)), "c2_allele_table_around_cut_write")
        return int(ng.value)

    @staticmethod
    def free(ctx, h):
        ctx.lib.c2_allele_table_free.restype = None
        ctx.lib.c2_allele_table_free(h)


CALLS = _NativeCalls


def default_threads():
    """host threads that write a chunk: the CPUs this process may use, at most 16"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                                                # (a cgroup quota below the affinity mask)
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(16, n))


class AlleleRows:
    """The sorted rows as numpy columns (what QuantResult.alleles() hands out): `aligned` / `reference` are fixed-width byte strings
    (numpy 'S': trailing zero padding is not part of the value), `label` indexes `labels`.  Iterating gives the tuples
    (Aligned_Sequence, Reference_Sequence, Reference_Name, Read_Status, n_deleted, n_inserted, n_mutated, #Reads, %Reads)."""
    def __init__(self, rows, aligned, reference, labels, n_total):
        self.rows, self.aligned, self.reference, self.labels, self.n_total = rows, aligned, reference, list(labels), n_total

    def __len__(self):
        return len(self.rows)

    def columns(self):
        r = self.rows
        names = np.array(self.labels, dtype=object)[r["label"]] if len(r) else np.zeros(0, dtype=object)
        status = np.where(r["modified"] != 0, 'MODIFIED', 'UNMODIFIED').astype(object)
        pct = r["reads"].astype(np.int64) / self.n_total * 100 if len(r) else np.zeros(0)
        return (np.char.decode(self.aligned, 'ascii').tolist(), np.char.decode(self.reference, 'ascii').tolist(), names.tolist(), status.tolist(),
                r["n_deleted"].tolist(), r["n_inserted"].tolist(), r["n_mutated"].tolist(), r["reads"].tolist(), pct.tolist())

    def tuples(self):
        return list(zip(*self.columns())) if len(self.rows) else []

    def __iter__(self):
        return iter(self.tuples())


class AlleleTable:
    """c2_allele_table: built from device tensors the caller keeps alive (`keep`: whatever owns them).  All tensors are torch tensors on
    the context's device; masks are int64 [n, ceil(k / 64)]; counts uint32 / int32 [n]."""
    def __init__(self, ctx, n_reads, n_refs, mode, flags, a1, f1, r1, stride1, member, flags_t, counts, a2=None, f2=None, r2=None, stride2=0,
                 slot2=None, use2=None, scaffold_hit=None, scaffold_ref=-1, stream=None, keep=None):
        self.ctx, self.n_refs = ctx, int(n_refs)
        self._keep = (keep, a1, f1, r1, member, flags_t, counts, a2, f2, r2, slot2, use2, scaffold_hit)
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        src = AlleleSrc(n_reads=int(n_reads), n_refs=int(n_refs), mode=int(mode), d_aln_read1=P(a1), d_aln_ref1=P(f1), d_records1=P(r1),
                        d_aln_read2=P(a2), d_aln_ref2=P(f2), d_records2=P(r2), d_slot2=P(slot2), d_member=P(member), d_use2=P(use2),
                        d_flags=P(flags_t), d_counts=P(counts), d_scaffold_hit=P(scaffold_hit), stride1=int(stride1), stride2=int(stride2),
                        scaffold_ref=int(scaffold_ref), flags=int(flags))
        self._h = CALLS.build(ctx, src, stream)
        self.n_rows = CALLS.rows(ctx, self._h)
        self.max_aln_len = max(int(stride1), int(stride2))

    def close(self):
        if self._h is not None:
            CALLS.free(self.ctx, self._h)
            self._h = None
            self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write(self, path, ref_names, n_total, dsODN="", threads=None, zip_member=None, zip_level=1):
        """Alleles_frequency_table.txt -> bytes written.  zip_member: `path` becomes a zip archive with the table as its one member of that name
        (what the reference's run leaves behind, CRISPRessoCORE.py:4531-4533) -> (bytes of the text, bytes of the archive)"""
        from .refs import reverse_complement
        labels = labels_for(ref_names)
        lab = (ctypes.c_char_p * len(labels))(*[x.encode() for x in labels])
        probes = None
        if dsODN != "":
            if len(dsODN) <= 6:
                raise KeyError("contains dsODN fragment")              # the reference selects a column it never made (:4519-4524)
            pr = [dsODN, reverse_complement(dsODN), dsODN[3:-3], reverse_complement(dsODN[3:-3])]
            probes = (ctypes.c_char_p * 4)(*[x.encode() for x in pr])
        if zip_member:
            return CALLS.write_zip(self.ctx, self._h, os.fsencode(path), zip_member.encode(), lab, int(n_total), probes, threads or default_threads(), zip_level)
        return CALLS.write(self.ctx, self._h, os.fsencode(path), lab, int(n_total), probes, threads or default_threads())

    def write_around_cut(self, path, label, cut_point, ref_len, plot_window_size, n_total, threads=None):
        """<ref>Alleles_frequency_table_around_<guide>.txt for the rows labelled `label` -> number of merged alleles"""
        try:
            return CALLS.around_cut_write(self.ctx, self._h, label, cut_point, ref_len, plot_window_size, int(n_total), os.fsencode(path),
                                          threads or default_threads())
        except Exception as e:
            if "is not in list" in str(e):
                raise ValueError("%d is not in list" % cut_point)       # ref_positions.index(cut_point)
            raise

    def rows(self, ref_names, n_total):
        """-> AlleleRows (the whole table in host memory)"""
        m, st = self.n_rows, self.max_aln_len
        rows = np.zeros(m, dtype=ROW_DTYPE)
        a = np.zeros((m, st), dtype=np.uint8)
        f = np.zeros((m, st), dtype=np.uint8)
        if m:
            CALLS.fetch(self.ctx, self._h, rows.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p), st)
        S = "S%d" % st
        return AlleleRows(rows, a.view(S).reshape(-1), f.view(S).reshape(-1), labels_for(ref_names), n_total)
