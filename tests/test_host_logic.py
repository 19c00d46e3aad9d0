"""Host-side logic and the C-ABI surface, no GPU needed: the shared library loads and exports every symbol the header
declares, fails loudly without a device, and the Python-side helpers (matrix parsing, seeds, score rounding, sharding,
synthetic data, gate tables) behave like the reference expressions they restate."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_declared_in_the_header():
    from crispresso2_amd import _native
    lib = _native.load()
    hdr = open(os.path.join(ROOT, "include", "crispresso2_amd.h")).read()
    declared = set(re.findall(r"\b(c2_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), "header declares %s but the library does not export it" % sym
    assert set(_native.SYMBOLS) <= declared
    assert lib.c2_abi_version() == 3 == _native.ABI_VERSION


def test_no_device_means_loud_failure_not_a_fallback():
    from crispresso2_amd import _native
    lib = _native.load()
    if lib.c2_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_native.NativeError) as e:
        _native.Context(0)
    assert "no CPU fallback" in str(e.value)
    from crispresso2_amd import CRISPResso2Align as A
    m = A.make_matrix()
    _native._default_ctx = None
    with pytest.raises(_native.NativeError):
        A.global_align("ACGT", "ACGT", matrix=m, gap_incentive=np.zeros(5, dtype=int))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "crispresso2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "c2_oracle" not in src, f


def test_record_struct_layout_matches_numpy_dtype():
    from crispresso2_amd import _native
    assert _native.REC_DTYPE.itemsize == 32
    assert _native.REC_DTYPE.fields["status"][1] == 23 and _native.REC_DTYPE.fields["ref_id"][1] == 26
    assert ctypes.sizeof(_native.Batch) == 96                         # (struct c2_batch: 80 bytes + min_read_len, padded to 8, + diag_hints: ABI 3)
    assert _native.Batch.min_read_len.offset == 80 and _native.Batch.diag_hints.offset == 88


def test_read_matrix_and_make_matrix():
    from crispresso2_amd import CRISPResso2Align as A
    d = os.path.dirname(A.__file__)
    m = A.read_matrix(os.path.join(d, "EDNAFULL"))
    assert m.shape == (90, 90) and m.dtype == np.int64
    assert m[ord("A"), ord("A")] == 5 and m[ord("A"), ord("T")] == -4 and m[ord("N"), ord("A")] == -2 and m[ord("N"), ord("N")] == -1
    b = A.read_matrix(os.path.join(d, "BLOSUM62"))
    assert b[ord("M"), ord("M")] == 5 and b[ord("W"), ord("W")] == 11
    mk = A.make_matrix()
    for x in "ACGTN":
        for y in "ACGTN":
            assert mk[ord(x), ord(y)] == m[ord(x), ord(y)]
    assert A.make_matrix(3, -2, -1, 0)[ord("N"), ord("N")] == 0


def test_alignment_seeds_match_the_reference_run():
    from crispresso2_amd import refs as RF
    for case in load_golden("variants.json.gz"):
        for r in case["refs"]:
            fw, rc = RF.alignment_seeds(r["sequence"])
            assert fw == r["fw_seeds"] and rc == r["rc_seeds"]
    assert RF.reverse_complement("acgtN_-") == "-_NACGT"
    with pytest.raises(KeyError):
        RF.reverse_complement("ACGR")


def test_score_from_counts_is_python_round():
    from crispresso2_amd.batch import score_from_counts
    rng = np.random.default_rng(0)
    n = rng.integers(1, 600, 5000)
    m = (rng.random(5000) * (n + 1)).astype(np.int64)
    got = score_from_counts(m, n)
    exp = [round(100 * int(a) / float(int(b)), 3) for a, b in zip(m, n)]
    assert got.tolist() == exp
    assert score_from_counts([0], [0])[0] == 0.0


def test_pack_reads_and_synth_are_deterministic():
    from crispresso2_amd.batch import pack_reads
    from crispresso2_amd import synth
    arena, off = pack_reads(["ACG", "", "TTTT"])
    assert arena.tobytes() == b"ACGTTTT" and off.tolist() == [0, 3, 3, 7]
    a = synth.make_reads(150, 1000)
    b = synth.make_reads(150, 70000)[:1000]
    assert np.array_equal(a, b)                                   # any prefix is reproducible
    assert set(np.unique(a).tolist()) <= set(b"ACGTN")
    amp, g, inc = synth.amplicon_setup(150)
    assert len(amp) == 150 and g[76] == 1 and g.sum() == 1 and inc == [75, 76]
    assert len(synth.make_variant(amp, "hdr")) == 153 and len(synth.make_variant(amp, "pe")) == 150


def test_count_layout_roundtrip():
    from crispresso2_amd import counts as C
    lay = C.CountLayout(2, 223, 250)
    x = np.zeros(lay.shape(), dtype=np.int64)
    x[1, 3 * lay.vl + 7] = 5                                      # all_substitution_count_vectors[7]
    x[1, C.N_VECTORS * lay.vl + 1] = 9                            # counts_modified
    x[1, C.N_VECTORS * lay.vl + C.N_SCALARS + 2 * lay.hl + 4] = 3  # substituted_n[4]
    u = lay.unpack(x, 1, 223)
    assert u["all_substitution_count_vectors"][7] == 5 and len(u["all_substitution_count_vectors"]) == 223
    assert u["counts_modified"] == 9 and u["substituted_n"] == {4: 3}


def test_tables_write_the_reference_formats_from_a_count_tensor(tmp_path):
    """tables.write_tables on a QuantResult built from oracle/aggregate.py output (no GPU): the FANC.Cas9 result files of the
    reference repository must come out byte for byte when the counts are right, and the other tables must be well formed."""
    import gzip
    import json
    import os
    import numpy as np
    import oracle
    from oracle import aggregate
    from crispresso2_amd import tables, counts as C, refs as RF
    from crispresso2_amd.pipeline import QuantResult
    from helpers import matrices
    here = os.path.dirname(os.path.abspath(__file__))
    with gzip.open(os.path.join(here, "golden", "fanc_run.json.gz"), "rt") as fh:
        g = json.load(fh)
    amp, cut = g["amplicon"], g["cut_point"]
    ref = RF.make_ref("Reference", amp, [cut], [cut, cut + 1], min_aln_score=60)
    m = matrices()["EDNAFULL"]
    lines = g["fastq"].split("\n")
    reads = [lines[k] for k in range(1, len(lines), 4) if lines[k]]
    # the reference's single-amplicon flow with the CPU oracle: forward strand only (the FANC reads are), min_aln_score 60
    items, n_total = [], 0
    for rd in reads:
        s1, s2, score = oracle.global_align(rd, amp, m, ref["gap_incentive"], -20, -2)
        if score > 60:
            p = oracle.find_indels_substitutions(s1, s2, ref["include_idxs"])
            p["aln_seq"], p["aln_ref"] = s1, s2
            items.append((p, 1))
            n_total += 1
    agg = aggregate.aggregate(items, len(amp))
    lay = C.CountLayout(1, len(amp), max(len(r) for r in reads))
    per_ref = {"Reference": agg}
    res = QuantResult(per_ref, {"N_TOT_READS": len(reads), "N_READS_INPUT": len(reads), "N_TOTAL": n_total}, lay, None)
    names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(tmp_path))
    for fn, text in g["expected_files"].items():
        assert fn in names and (tmp_path / fn).read_text() == text, fn
    win = (tmp_path / "Quantification_window_nucleotide_frequency_table.txt").read_text().split("\n")
    assert win[0] == "\t" + "\t".join(amp[cut:cut + 2]) and len(win) == 8
    pct = (tmp_path / "Nucleotide_percentage_table.txt").read_text().split("\n")[1].split("\t")
    assert pct[0] == "A" and abs(float(pct[4]) - 1.0) < 1e-12          # position 4 of the amplicon is an A in every aligned read
    assert tables.ref_plot_name(["Reference"], "Reference") == "" and tables.ref_plot_name(["A", "B"], "A") == "A."


def test_rc_partner_search_with_torch_ops_equals_the_native_search():
    """pipeline.rc_partners_device (reads of one length as a byte matrix; runs on the GPU in production, on CPU tensors here) against
    c2_rc_partners: pairs in both orders, palindromes, lower-case reads, characters outside ACGTN_-, reads without a partner."""
    import numpy as np
    import torch
    from crispresso2_amd import _native, pipeline
    rng = np.random.default_rng(11)
    L = 37
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "_": "_", "-": "-"}
    rc = lambda s_: "".join(comp[c] for c in reversed(s_.upper()))
    base = ["".join(rng.choice(list("ACGTN"), L)) for _ in range(400)]
    seqs = []
    for k, b in enumerate(base):
        seqs.append(b)
        if k % 3 == 0:
            seqs.append(rc(b))
        if k % 7 == 0:
            seqs.append(b.lower())
        if k % 11 == 0:
            seqs.append(b[:5] + "R" + b[6:])
        if k % 13 == 0:
            seqs.append(b[:9] + "-_" + b[11:])
    half = "".join(rng.choice(list("ACGT"), L // 2))
    seqs.append(half + "N" + rc(half))                                 # its own reverse complement
    seqs = list(dict.fromkeys(seqs))
    order = rng.permutation(len(seqs))
    seqs = [seqs[k] for k in order]
    arena = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    off = (np.arange(len(seqs) + 1, dtype=np.uint64) * L)
    want = _native.rc_partners(arena, off)
    got = pipeline.rc_partners_device(torch.from_numpy(arena).view(len(seqs), L)).result()
    assert got is not None and np.array_equal(got, want)
    assert (want >= 0).sum() > 200 and (want == np.arange(len(seqs))).sum() >= 1 and (want < 0).sum() > 100


def test_hostcopy_round_trip_on_the_cpu_device():
    """hostcopy.to_device / to_host on a non-GPU device: plain copies with the dtype conversions the pipeline asks for"""
    import torch
    from crispresso2_amd.hostcopy import to_device, to_host
    a = np.arange(2_000_000, dtype=np.int32)
    t = to_device(a, torch.device("cpu"))
    assert t.dtype == torch.int32 and t.shape == (2_000_000,)
    b = to_host(t, np.int64)
    assert b.dtype == np.int64 and np.array_equal(b, a)
    a[0] = 7                                                          # (copies, not views)
    assert int(t[0]) in (0, 7) and b[0] == 0
    e = to_host(torch.zeros((0, 3), dtype=torch.uint8))
    assert e.shape == (0, 3)
