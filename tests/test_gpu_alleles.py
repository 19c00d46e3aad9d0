"""-m gpu: the allele table through the C ABI on the MI355X (c2_allele_table_build / _write / _fetch / _around_cut_write: rocPRIM's merge sort
with the table's comparator, the text kernels, the chunked writer) against the pandas restatement of the reference's frame and files
(oracle/aggregate.py) on the hand-made rows of tests/test_alleles_emulated.py, and on a large table whose order is checked by properties."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


def _gpu_table(names, stride, dev, n, k, mode, flags, pe):
    import torch
    from crispresso2_amd import _native, alleles as AL
    ctx = _native.default_context()
    d = torch.device("cuda", 0)

    def T(a, view=None):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.uint8) if a.dtype.names else a).to(d)
    member = T(dev["member"].view(np.int64))
    use2 = None if dev["use2"] is None else T(dev["use2"].view(np.int64))
    return AL.AlleleTable(ctx, n, k, mode, flags, T(dev["a1"]), T(dev["f1"]), T(dev["r1"]), stride, member, T(dev["flags"]), T(dev["cnt"].view(np.int32)),
                          a2=T(dev["a2"]), f2=T(dev["f2"]), r2=T(dev["r2"]), stride2=stride, slot2=T(dev["slot2"]), use2=use2,
                          scaffold_hit=T(dev["hit"]), scaffold_ref=pe)


@pytest.mark.parametrize("seed,n,k,mode,flags,b2,scaf,dsODN,chunk", [
    (1, 300, 1, 0, 0, False, False, "", 700),
    (2, 3000, 3, 0, 0, True, False, "", 65536),
    (3, 3000, 3, 1, 8, True, True, "", 1 << 20),
    (4, 2000, 2, 2, 9, True, True, "ACGTAGGTCA", 4096),
    (5, 500, 70, 2, 6, False, False, "", 1 << 20),
    (7, 40000, 1, 0, 0, False, False, "", 1 << 20),
])
def test_allele_table_on_the_device_equals_the_pandas_restatement(tmp_path, monkeypatch, seed, n, k, mode, flags, b2, scaf, dsODN, chunk):
    import test_alleles_emulated as TE
    from oracle import aggregate as AG
    rng = np.random.default_rng(seed)
    names, stride, dev, rows = TE._build(rng, n, k, mode, flags, b2, scaf)
    n_total = max(1, int(sum(r[7] for r in rows)) + 17)
    pe = k - 1 if scaf else -1
    monkeypatch.setenv("C2_ALLELE_CHUNK_BYTES", str(chunk))
    tab = _gpu_table(names, stride, dev, n, k, mode, flags, pe)
    try:
        assert tab.n_rows == len(rows)
        out = tmp_path / "Alleles_frequency_table.txt"
        nb = tab.write(str(out), names, n_total, dsODN=dsODN, threads=4)
        got = out.read_text()
        assert nb == len(got.encode())
        assert got == AG.allele_table_text(rows, n_total, dsODN=dsODN)
        tuples = tab.rows(names, n_total).tuples()
        lines = got.split("\n")[1:-1]
        assert len(tuples) == len(lines)
        for tpl, line in zip(tuples[:300], lines[:300]):
            cells = line.split("\t")
            assert [tpl[0], tpl[1], tpl[2], tpl[3], str(tpl[4]), str(tpl[5]), str(tpl[6]), str(tpl[7]), repr(tpl[8])] == cells[:9]
        for r in range(min(k, 3)):
            cut = 30
            in_ref = [row + (row[7] / n_total * 100,) for row in TE._sorted(rows) if row[2] == names[r]]
            if not in_ref or any(sum(c != '-' for c in row[1]) <= cut for row in in_ref):
                continue
            p = tmp_path / ("around_%d.txt" % r)
            ng = tab.write_around_cut(str(p), r, cut, 60, 20, n_total, threads=4)
            want = AG.alleles_around_cut(in_ref, names[r], cut, 60, 20)
            assert p.read_text() == want and ng == len(want.split("\n")) - 2
    finally:
        tab.close()


def test_empty_table_and_missing_cut_point(tmp_path):
    import test_alleles_emulated as TE
    rng = np.random.default_rng(9)
    names, stride, dev, rows = TE._build(rng, 50, 2, 0, 0, False, False)
    tab = _gpu_table(names, stride, dev, 50, 2, 0, 0, -1)
    try:
        with pytest.raises(ValueError, match="is not in list"):
            tab.write_around_cut(str(tmp_path / "x.txt"), 0, 75, 100, 20, 100)
    finally:
        tab.close()
    dev["cnt"][:] = 0
    tab = _gpu_table(names, stride, dev, 50, 2, 0, 0, -1)
    try:
        assert tab.n_rows == 0
        p = tmp_path / "t.txt"
        tab.write(str(p), names, 10)
        assert p.read_text().count("\n") == 1 and tab.rows(names, 10).tuples() == []
    finally:
        tab.close()


def test_whole_pipeline_table_of_synthetic_reads_is_sorted_and_complete(tmp_path):
    """200 k synthetic 250-bp reads through pipeline.quantify_fastq + tables.write_tables: the allele table on disk has one line per aligned
    unique read with copies, in the reference's order (checked line against line), its #Reads add up to N_TOTAL, and the file equals the
    pandas restatement fed with the same rows in unique-read order."""
    from types import SimpleNamespace
    from crispresso2_amd import pipeline, tables, refs as RF, synth, CRISPResso2Align as A
    from oracle import aggregate as AG
    L, n = 250, 200_000
    amp, g, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, n)
    fq = tmp_path / "reads.fastq"
    synth.write_fastq(reads, str(fq))
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                           ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [amp[L // 2 - 16:L // 2 + 4]]
    m = A.read_matrix(os.path.join(os.path.dirname(HERE), "crispresso2_amd", "EDNAFULL"))
    res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], m, args)
    out = tmp_path / "out"
    written = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert "Alleles_frequency_table.txt" in written and any(w.startswith("Alleles_frequency_table_around_") for w in written)
    text = (out / "Alleles_frequency_table.txt").read_text()
    lines = text.split("\n")[1:-1]
    cells = [ln.split("\t") for ln in lines]
    assert sum(int(c[7]) for c in cells) == res.stats["N_TOTAL"]
    keys = [(-int(c[7]), c[0], c[1]) for c in cells]
    assert keys == sorted(keys)
    # the same rows in unique-read order (the reference's alleles_list order) through pandas
    hv = res.host_view()
    AR = res.allele_rows()
    order = np.argsort(AR.rows["read"], kind="stable")
    cols = AR.columns()
    rows = [tuple(cols[c][j] for c in range(8)) for j in order.tolist()]
    assert len(rows) == int((hv["aligned"] & (hv["cnt"] > 0)).sum())
    assert text == AG.allele_table_text(rows, res.stats["N_TOTAL"])
    around = [w for w in written if w.startswith("Alleles_frequency_table_around_")][0]
    sorted_rows = [tuple(cols[c][j] for c in range(9)) for j in range(len(AR))]
    assert (out / around).read_text() == AG.alleles_around_cut(sorted_rows, "Reference", L // 2, L, 20)
