"""The allele table of the device route (crispresso2_amd/csrc/c2_k_alleles.hip + c2_alleles_host.h) on the wave emulator, against a pandas
restatement of the reference's frame, sort and files (oracle/aggregate.py: CRISPRessoCORE.py:3964-4010, :4298-4303, :4498-4530,
CRISPRessoShared.py:1513-1531).  Rows are made by hand -- the table does not care whether the strings are alignments: shared prefixes,
equal (#Reads, strings) ties, several references with every ambiguity mode, rows of the both-strand batch, scaffold hits, discarded
reads, dsODN probes that occur at column 0, chunks of a few hundred bytes."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import emu_driver as E                                       # noqa: E402
from oracle import aggregate as AG                           # noqa: E402
from crispresso2_amd import alleles as AL                    # noqa: E402
from pipeline_on_emulator import _EmuAlleleCalls             # noqa: E402


class _Ctx:
    lib = handle = None


def _rand_aln(rng, base, L):
    """a (read, reference) string pair of equal length derived from `base`: substitutions, a deletion, an insertion"""
    a, f = list(base), list(base)
    for _ in range(rng.integers(0, 3)):
        a[rng.integers(0, L)] = "ACGTN"[rng.integers(0, 5)]
    if rng.random() < 0.4:
        p, d = rng.integers(5, L - 12), rng.integers(1, 9)
        a[p:p + d] = "-" * d
    if rng.random() < 0.25:
        p, d = rng.integers(5, L - 5), rng.integers(1, 6)
        ins = "".join("ACGT"[x] for x in rng.integers(0, 4, d))
        a[p:p] = list(ins)
        f[p:p] = list("-" * d)
    return "".join(a), "".join(f)


def _build(rng, n, k, mode, flags, with_batch2, with_scaffold, L=60):
    """random device-side inputs of c2_allele_table_build + the rows the reference's loop would append (with names) in its order"""
    names = ["Ref%d" % r for r in range(k)]
    bases = ["".join("ACGT"[x] for x in rng.integers(0, 4, L)) for _ in range(k)]
    stride = ((2 * L + 15) // 16) * 16
    a1 = rng.integers(33, 120, (n * k, stride)).astype(np.uint8)          # garbage beyond aln_len, as in the product's buffers
    f1 = rng.integers(33, 120, (n * k, stride)).astype(np.uint8)
    r1 = np.zeros(n * k, dtype=E.REC_DTYPE)
    strs1 = {}
    pool = {}
    for t in range(n * k):
        r = t % k
        key = (r, int(rng.integers(0, max(2, n // 3))))                   # repeated strings: equal-key ties in the table
        if key not in pool:
            pool[key] = _rand_aln(rng, bases[r], L)
        a, f = pool[key]
        T = len(f)
        a1[t, :T] = np.frombuffer(a.encode(), dtype=np.uint8)
        f1[t, :T] = np.frombuffer(f.encode(), dtype=np.uint8)
        r1[t]["aln_len"] = T
        r1[t]["deletion_n"], r1[t]["insertion_n"], r1[t]["substitution_n"] = rng.integers(0, 3), rng.integers(0, 2), rng.integers(0, 4)
        strs1[t] = (a, f)
    n2 = n // 4 if with_batch2 else 0
    slot2 = np.full(n * k, -1, dtype=np.int32)
    a2 = f2 = r2 = None
    strs2 = {}
    if n2:
        a2 = rng.integers(33, 120, (n2, stride)).astype(np.uint8)
        f2 = rng.integers(33, 120, (n2, stride)).astype(np.uint8)
        r2 = np.zeros(n2, dtype=E.REC_DTYPE)
        where = rng.choice(n * k, n2, replace=False)
        for s, t in enumerate(where):
            slot2[t] = s
            a, f = _rand_aln(rng, bases[t % k], L)
            T = len(f)
            a2[s, :T] = np.frombuffer(a.encode(), dtype=np.uint8)
            f2[s, :T] = np.frombuffer(f.encode(), dtype=np.uint8)
            r2[s]["aln_len"] = T
            r2[s]["deletion_n"], r2[s]["insertion_n"], r2[s]["substitution_n"] = rng.integers(0, 3), rng.integers(0, 2), rng.integers(0, 4)
            strs2[s] = (a, f)
    member = rng.random((n, k)) < (0.6 if k > 1 else 1.0)
    member[np.arange(n), rng.integers(0, k, n)] = True
    use2 = (slot2.reshape(n, k) >= 0) & (rng.random((n, k)) < 0.7)
    aligned = rng.random(n) < 0.9
    cnt = np.where(rng.random(n) < 0.15, 0, rng.choice([1, 1, 1, 2, 3, 7, 40, 1000], n)).astype(np.uint32)
    hit = (rng.random(n) < 0.1) if with_scaffold else np.zeros(n, dtype=bool)
    pe = k - 1 if with_scaffold else -1
    from crispresso2_amd.pipeline import _pack_masks
    dev = dict(a1=a1, f1=f1, r1=r1, a2=a2, f2=f2, r2=r2, slot2=slot2 if n2 else None, member=_pack_masks(member), use2=_pack_masks(use2) if n2 else None,
               flags=aligned.astype(np.uint8), cnt=cnt, hit=hit.astype(np.uint8) if with_scaffold else None)
    # ---- the reference's loop (:3964-4010) over the same data
    ign_sub, ign_ins, ign_del, discard = bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8)
    rows = []
    for i in range(n):
        if not (aligned[i] and cnt[i] > 0):
            continue
        best = [int(r) for r in np.nonzero(member[i])[0]]

        def payload(r):
            t = i * k + r
            if n2 and use2[i, r]:
                s = int(slot2[t])
                return strs2[s], r2[s]
            return strs1[t], r1[t]

        def row(label, r, counted, first):
            (a, f), rec = payload(r)
            dn, inn, sn = int(rec["deletion_n"]), int(rec["insertion_n"]), int(rec["substitution_n"])
            mod = (not ign_del and dn > 0) or (not ign_ins and inn > 0) or (not ign_sub and sn > 0)
            if counted and discard and (dn > 0 or inn > 0):
                label = 'DISCARDED_Scaffold-incorporated' if label == 'Scaffold-incorporated' else 'DISCARDED_' + names[first]
            rows.append((a, f, label, 'MODIFIED' if mod else 'UNMODIFIED', dn, inn, sn, int(cnt[i])))
        if hit[i]:
            row('Scaffold-incorporated', pe, True, None)
        elif len(best) > 1 and mode == 0:
            row('AMBIGUOUS_' + names[best[0]], best[0], False, best[0])
        elif len(best) > 1 and mode == 1:
            row(names[best[0]], best[0], True, best[0])
        else:
            for r in best:
                row(names[r], r, True, best[0])
    return names, stride, dev, rows


class _Held:
    """numpy arrays standing in for device tensors (data_ptr = host address, as under pipeline_on_emulator)"""
    def __init__(self, a):
        self.a = None if a is None else np.ascontiguousarray(a)

    def data_ptr(self):
        return self.a.ctypes.data


def _table(names, stride, dev, n, k, mode, flags, pe):
    AL.CALLS, saved = _EmuAlleleCalls, AL.CALLS
    try:
        H = {q: _Held(v) for q, v in dev.items()}
        P = lambda q: None if H[q].a is None else H[q]
        return AL.AlleleTable(_Ctx(), n, k, mode, flags, P("a1"), P("f1"), P("r1"), stride, P("member"), P("flags"), P("cnt"), a2=P("a2"), f2=P("f2"), r2=P("r2"),
                              stride2=stride, slot2=P("slot2"), use2=P("use2"), scaffold_hit=P("hit"), scaffold_ref=pe, keep=H), saved
    except Exception:
        AL.CALLS = saved
        raise


@pytest.mark.parametrize("seed,n,k,mode,flags,b2,scaf,dsODN", [
    (1, 300, 1, 0, 0, False, False, ""),
    (2, 400, 3, 0, 0, True, False, ""),
    (3, 400, 3, 1, 8, True, True, ""),
    (4, 400, 2, 2, 8 | 1, True, True, "ACGTAGGTCA"),
    (5, 200, 70, 2, 2 | 4, False, False, ""),                # two mask words per read
    (6, 1, 1, 0, 0, False, False, "TTGACCAGTCCA"),
    (7, 5000, 1, 0, 0, False, False, ""),
])
def test_allele_table_text_equals_the_pandas_restatement(tmp_path, monkeypatch, seed, n, k, mode, flags, b2, scaf, dsODN):
    rng = np.random.default_rng(seed)
    names, stride, dev, rows = _build(rng, n, k, mode, flags, b2, scaf)
    if dsODN and rows:
        # plant the probe (and its fragment) at column 0 and further in, on either strand
        from oracle.fastq import reverse_complement
        for j, (pos, seq) in enumerate([(0, dsODN), (7, dsODN), (0, dsODN[3:-3]), (9, reverse_complement(dsODN)), (11, reverse_complement(dsODN[3:-3]))]):
            t = j % (n * k)
            T = int(dev["r1"][t]["aln_len"])
            if pos + len(seq) <= T:
                dev["a1"][t, pos:pos + len(seq)] = np.frombuffer(seq.encode(), dtype=np.uint8)
        names, stride, dev, rows = _rebuild_rows(names, stride, dev, n, k, mode, flags, scaf)
    n_total = max(1, int(sum(r[7] for r in rows)) + 17)
    pe = k - 1 if scaf else -1
    monkeypatch.setenv("C2_ALLELE_CHUNK_BYTES", "700" if n < 1000 else "65536")
    tab, saved = _table(names, stride, dev, n, k, mode, flags, pe)
    try:
        assert tab.n_rows == len(rows)
        out = tmp_path / "Alleles_frequency_table.txt"
        nb = tab.write(str(out), names, n_total, dsODN=dsODN, threads=3)
        want = AG.allele_table_text(rows, n_total, dsODN=dsODN) if rows else None
        got = out.read_text()
        assert nb == len(got.encode())
        if rows:
            assert got == want
        # the same table as the reference's run leaves it (CRISPRessoCORE.py:4531-4533): a zip archive with the .txt as its one member.  The
        # chunks are deflated slice by slice on several threads and concatenated into one stream: Python's zipfile must give the text back
        import zipfile
        zp = tmp_path / "Alleles_frequency_table.zip"
        nb_z, zip_bytes = tab.write(str(zp), names, n_total, dsODN=dsODN, threads=3, zip_member="Alleles_frequency_table.txt")
        assert nb_z == nb and zip_bytes == zp.stat().st_size
        with zipfile.ZipFile(str(zp)) as z:
            assert z.namelist() == ["Alleles_frequency_table.txt"] and z.testzip() is None
            info = z.getinfo("Alleles_frequency_table.txt")
            assert info.compress_type == zipfile.ZIP_DEFLATED and info.file_size == nb
            assert z.read("Alleles_frequency_table.txt").decode() == got
        if seed == 3:                                                  # the records a table of 4 GiB or more gets (zip64 extra fields, zip64 end record + locator)
            monkeypatch.setenv("C2_ZIP_FORCE_ZIP64", "1")
            tab.write(str(zp), names, n_total, dsODN=dsODN, threads=2, zip_member="Alleles_frequency_table.txt")
            monkeypatch.delenv("C2_ZIP_FORCE_ZIP64")
            with zipfile.ZipFile(str(zp)) as z:
                assert z.testzip() is None and z.read("Alleles_frequency_table.txt").decode() == got
            with open(str(zp), "rb") as fh:
                blob = fh.read()
            assert b"PK\x06\x06" in blob and b"PK\x06\x07" in blob
        # the rows in memory: same order, same values
        AR = tab.rows(names, n_total)
        tuples = AR.tuples()
        lines = got.split("\n")[1:-1]
        assert len(tuples) == len(lines)
        for tpl, line in zip(tuples[:200], lines[:200]):
            cells = line.split("\t")
            assert [tpl[0], tpl[1], tpl[2], tpl[3], str(tpl[4]), str(tpl[5]), str(tpl[6]), str(tpl[7]), repr(tpl[8])] == cells[:9]
        # around the cut, per reference label
        for r in range(min(k, 3)):
            cut = 30
            in_ref = [row + (row[7] / n_total * 100,) for row in _sorted(rows) if row[2] == names[r]]
            if not in_ref or any(sum(c != '-' for c in row[1]) <= cut for row in in_ref):
                continue
            p = tmp_path / ("around_%d.txt" % r)
            ng = tab.write_around_cut(str(p), r, cut, 60, 20, n_total, threads=2)
            want_ac = AG.alleles_around_cut(in_ref, names[r], cut, 60, 20)
            assert p.read_text() == want_ac
            assert ng == len(want_ac.split("\n")) - 2
    finally:
        tab.close()
        AL.CALLS = saved


def _sorted(rows):
    return sorted(rows, key=lambda t: (-t[7], t[0], t[1]))               # (stable, as pandas' lexsort)


def _rebuild_rows(names, stride, dev, n, k, mode, flags, scaf):
    """the reference rows again after dev['a1'] was edited: read them back from the arrays"""
    member = np.zeros((n, k), dtype=bool)
    w = dev["member"].view(np.uint64).reshape(n, -1)
    for r in range(k):
        member[:, r] = (w[:, r >> 6] >> np.uint64(r & 63)) & np.uint64(1)
    use2 = np.zeros((n, k), dtype=bool)
    if dev["use2"] is not None:
        w2 = dev["use2"].view(np.uint64).reshape(n, -1)
        for r in range(k):
            use2[:, r] = (w2[:, r >> 6] >> np.uint64(r & 63)) & np.uint64(1)
    aligned, cnt = dev["flags"].astype(bool), dev["cnt"]
    hit = dev["hit"].astype(bool) if dev["hit"] is not None else np.zeros(n, dtype=bool)
    pe = k - 1 if scaf else -1
    ign_sub, ign_ins, ign_del, discard = bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8)
    rows = []

    def strings(i, r):
        t = i * k + r
        if dev["slot2"] is not None and use2[i, r]:
            s = int(dev["slot2"][t])
            rec = dev["r2"][s]
            T = int(rec["aln_len"])
            return dev["a2"][s, :T].tobytes().decode(), dev["f2"][s, :T].tobytes().decode(), rec
        rec = dev["r1"][t]
        T = int(rec["aln_len"])
        return dev["a1"][t, :T].tobytes().decode(), dev["f1"][t, :T].tobytes().decode(), rec
    for i in range(n):
        if not (aligned[i] and cnt[i] > 0):
            continue
        best = [int(r) for r in np.nonzero(member[i])[0]]

        def row(label, r, counted, first):
            a, f, rec = strings(i, r)
            dn, inn, sn = int(rec["deletion_n"]), int(rec["insertion_n"]), int(rec["substitution_n"])
            mod = (not ign_del and dn > 0) or (not ign_ins and inn > 0) or (not ign_sub and sn > 0)
            if counted and discard and (dn > 0 or inn > 0):
                label = 'DISCARDED_Scaffold-incorporated' if label == 'Scaffold-incorporated' else 'DISCARDED_' + names[first]
            rows.append((a, f, label, 'MODIFIED' if mod else 'UNMODIFIED', dn, inn, sn, int(cnt[i])))
        if hit[i]:
            row('Scaffold-incorporated', pe, True, None)
        elif len(best) > 1 and mode == 0:
            row('AMBIGUOUS_' + names[best[0]], best[0], False, best[0])
        elif len(best) > 1 and mode == 1:
            row(names[best[0]], best[0], True, best[0])
        else:
            for r in best:
                row(names[r], r, True, best[0])
    return names, stride, dev, rows


def test_no_rows_gives_the_header_only(tmp_path):
    rng = np.random.default_rng(9)
    names, stride, dev, rows = _build(rng, 50, 2, 0, 0, False, False)
    dev["cnt"][:] = 0
    tab, saved = _table(names, stride, dev, 50, 2, 0, 0, -1)
    try:
        assert tab.n_rows == 0
        p = tmp_path / "t.txt"
        tab.write(str(p), names, 10)
        assert p.read_text() == "Aligned_Sequence\tReference_Sequence\tReference_Name\tRead_Status\tn_deleted\tn_inserted\tn_mutated\t#Reads\t%Reads\n"
        import zipfile
        tab.write(str(tmp_path / "t.zip"), names, 10, zip_member="Alleles_frequency_table.txt")
        with zipfile.ZipFile(str(tmp_path / "t.zip")) as z:
            assert z.testzip() is None and z.read("Alleles_frequency_table.txt").decode() == p.read_text()
        assert tab.rows(names, 10).tuples() == []
    finally:
        tab.close()
        AL.CALLS = saved


def test_cut_point_beyond_a_rows_reference_raises_as_list_index_does(tmp_path):
    rng = np.random.default_rng(11)
    names, stride, dev, rows = _build(rng, 40, 1, 0, 0, False, False)
    tab, saved = _table(names, stride, dev, 40, 1, 0, 0, -1)
    try:
        with pytest.raises(ValueError, match="59 is not in list|is not in list"):
            tab.write_around_cut(str(tmp_path / "x.txt"), 0, 75, 100, 20, 100)     # no row's reference string has 76 bases
    finally:
        tab.close()
        AL.CALLS = saved


def test_float_repr_is_pythons():
    L = E.lib()
    L.emu_format_float_repr.argtypes = [ctypes.c_double, ctypes.c_char_p]
    buf = ctypes.create_string_buffer(64)
    rng = np.random.default_rng(3)
    vals = [0.0, 1.0, 100.0, 1e-5, 1e-4, 1e16, 1e15, 1e22, 5e-324, 1.7976931348623157e308, 0.1, 1 / 3, 99.99999999999999, 2.5e-5, 12345678.9]
    vals += (rng.integers(1, 10**7, 20000) / rng.integers(1, 10**7, 20000) * 100).tolist()
    vals += [float(x) for x in rng.integers(0, 2**63, 20000).astype(np.uint64).view(np.float64) if np.isfinite(x)]
    for v in vals:
        n = L.emu_format_float_repr(v, buf)
        assert buf.raw[:n].decode() == repr(float(v))
