"""Every result table of the reference's own end-to-end run (CRISPResso -r1 FANC.Cas9.fastq -a ... -g ...), byte for byte.

tests/golden/fanc_full_run.json.gz holds every .txt file the REFERENCE wrote for that run in the dev container
(make_golden.py --fanc-full: its main(), plot functions replaced by no-ops) plus the unzipped allele table.  tables.write_tables
must reproduce, from the count tensor: quantification of editing frequency, mapping statistics, nucleotide frequency /
percentage tables (amplicon and quantification window), modification count vectors (both), the four effect vectors, the
four histograms, the allele frequency table and the alleles around the guide's cut.  CPU: counts built by the oracle
(oracle/aggregate.py); GPU: the device-resident pipeline.  Not reproduced: Alleles_homology_scores.txt (data of a plot)."""
import gzip
import json
import os

import numpy as np
import pytest

from helpers import matrices

HERE = os.path.dirname(os.path.abspath(__file__))
NOT_REPRODUCED = {"Alleles_homology_scores.txt"}


def _golden():
    with gzip.open(os.path.join(HERE, "golden", "fanc_full_run.json.gz"), "rt") as fh:
        g = json.load(fh)
    with gzip.open(os.path.join(HERE, "golden", "fanc_run.json.gz"), "rt") as fh:
        g["fastq"] = json.load(fh)["fastq"]
    return g


def _compare(g, names, out_dir, skip=()):
    n = 0
    for fn, text in g["files"].items():
        if fn in NOT_REPRODUCED or fn in skip:
            continue
        assert fn in names, fn
        with open(os.path.join(out_dir, fn)) as fh:
            assert fh.read() == text, fn
        n += 1
    return n


def test_tables_from_oracle_counts_equal_every_file_of_the_reference_run(tmp_path):
    import oracle
    from oracle import aggregate
    from crispresso2_amd import tables, counts as C, refs as RF
    from crispresso2_amd.pipeline import QuantResult
    g = _golden()
    amp, cut = g["amplicon"], g["cut_point"]
    ref = RF.make_ref("Reference", amp, [cut], [cut, cut + 1], min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [g["guide"]]
    m = matrices()["EDNAFULL"]
    lines = g["fastq"].split("\n")
    reads = [lines[k] for k in range(1, len(lines), 4) if lines[k]]
    unique = {}
    for rd in reads:
        unique[rd] = unique.get(rd, 0) + 1
    # the reference's single-amplicon flow with the CPU oracle (forward strand only: the FANC reads are), process_fastq's statistics
    stats = {"N_TOT_READS": len(reads), "N_READS_INPUT": len(reads), "N_TOTAL": 0, "N_COMPUTED_ALN": 0, "N_CACHED_ALN": 0,
             "N_COMPUTED_NOTALN": 0, "N_CACHED_NOTALN": 0}
    items = []
    for rd, c in unique.items():
        s1, s2, score = oracle.global_align(rd, amp, m, ref["gap_incentive"], -20, -2)
        if score > 60:
            p = oracle.find_indels_substitutions(s1, s2, ref["include_idxs"])
            p["aln_seq"], p["aln_ref"] = s1, s2
            items.append((p, c))
            stats["N_TOTAL"] += c
            stats["N_COMPUTED_ALN"] += 1
            stats["N_CACHED_ALN"] += c - 1
        else:
            stats["N_COMPUTED_NOTALN"] += 1
            stats["N_CACHED_NOTALN"] += c - 1
    agg = aggregate.aggregate(items, len(amp))
    lay = C.CountLayout(1, len(amp), max(len(r) for r in reads))

    class WithAlleles(QuantResult):
        def alleles(self):                                          # get_allele_row per unique read, :3926-3959, sort and %Reads :4298-4303
            rows = []
            for p, c in items:
                mod = p["insertion_n"] + p["deletion_n"] + p["substitution_n"] > 0
                rows.append((p["aln_seq"], p["aln_ref"], "Reference", "MODIFIED" if mod else "UNMODIFIED", p["deletion_n"], p["insertion_n"],
                             p["substitution_n"], c, c / stats["N_TOTAL"] * 100))
            return sorted(rows, key=lambda t: (-t[7], t[0], t[1]))
    res = WithAlleles({"Reference": agg}, stats, lay, None)
    names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(tmp_path))
    assert _compare(g, names, str(tmp_path)) == 18
    # an amplicon without reads: zero effect vectors, the reference's histogram ranges
    empty = aggregate.aggregate([], len(amp))
    res0 = QuantResult({"Reference": empty}, dict(stats, N_TOTAL=0), lay, None)
    d0 = tmp_path / "empty"
    tables.write_tables(res0, {"Reference": ref}, ["Reference"], str(d0))
    eff = (d0 / "Effect_vector_combined.txt").read_text().split("\n")
    assert eff[0] == "# amplicon position\teffect" and eff[1] == "1\t0.000000000000000000e+00" and len(eff) == len(amp) + 2
    ind = (d0 / "Indel_histogram.txt").read_text().split("\n")
    assert ind[1] == "%d\t0" % -len(amp) and ind[-2] == "15\t0"             # min(ref_len - 15, min([] or [0])) = 0 -> starts at -ref_len
    assert (d0 / "Insertion_histogram.txt").read_text().split("\n")[16] == "15\t0"


@pytest.mark.gpu
def test_tables_from_the_device_pipeline_equal_every_file_of_the_reference_run(tmp_path):
    import argparse
    from crispresso2_amd import _native, pipeline, tables, refs as RF
    g = _golden()
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(g["fastq"])
    cut = g["cut_point"]
    ref = RF.make_ref("Reference", g["amplicon"], [cut], [cut, cut + 1], min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [g["guide"]]
    args = argparse.Namespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                              use_legacy_insertion_quantification=False, ignore_deletions=False, ignore_insertions=False,
                              ignore_substitutions=False, assign_ambiguous_alignments_to_first_reference=False,
                              expand_ambiguous_alignments=False, prime_editing_pegRNA_scaffold_seq="", discard_indel_reads=False)
    res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], args, ctx=_native.default_context())
    out = tmp_path / "CRISPResso_on_FANC.Cas9"
    names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert _compare(g, names, str(out)) == 18


def test_alleles_around_cut_random_tables_windows_at_the_amplicon_ends_and_merging(tmp_path):
    """tables.alleles_around_cut against the pandas restatement of the reference's function (oracle/aggregate.py): random
    allele tables with indels, cuts next to either end of the amplicon (clipped windows), several references, rows that
    coincide inside the window (merged: #Reads added, %Reads added with pandas' compensated float sum)."""
    import numpy as np
    from oracle import aggregate
    from crispresso2_amd import tables
    rng = np.random.default_rng(77)
    for trial in range(40):
        L = int(rng.integers(30, 90))
        ref = "".join(rng.choice(list("ACGT"), L))
        rows, total = [], 0
        for k in range(int(rng.integers(1, 60))):
            a, r = list(ref), list(ref)
            for _ in range(int(rng.integers(0, 4))):                         # substitutions, deletions, insertions as the aligner writes them
                kind, at = int(rng.integers(0, 3)), int(rng.integers(0, len(r)))
                if kind == 0 and a[at] != '-' and r[at] != '-':
                    a[at] = "ACGT"[int(rng.integers(0, 4))]
                elif kind == 1:
                    n = int(rng.integers(1, 6))
                    for q in range(at, min(len(r), at + n)):
                        if r[q] != '-' and a[q] != '-':
                            a[q] = '-'
                else:
                    ins = list(rng.choice(list("ACGT"), int(rng.integers(1, 5))))
                    if at > 0 and a[at - 1] != '-' and (at >= len(a) or a[at] != '-'):
                        a[at:at] = ins
                        r[at:at] = ['-'] * len(ins)
            a, r = "".join(a), "".join(r)
            dn, inn = a.count('-'), r.count('-')
            sn = sum(x != y and x != '-' and y != '-' for x, y in zip(a, r))
            reads = int(rng.integers(1, 50))
            total += reads
            rows.append([a, r, ["R1", "R2", "AMBIGUOUS_R1"][int(rng.integers(0, 3))], "MODIFIED" if dn + inn + sn else "UNMODIFIED", dn, inn, sn, reads])
        seen, uniq = set(), []
        for row in rows:                                                      # the allele table has one row per (strings, reference)
            if (row[0], row[1], row[2]) not in seen:
                seen.add((row[0], row[1], row[2]))
                uniq.append(tuple(row) + (row[7] / total * 100,))
        uniq.sort(key=lambda t: (-t[7], t[0], t[1]))
        for cut in (0, 1, int(rng.integers(0, L)), L - 2, L - 1):
            for name in ("R1", "R2"):
                if not any(u[2] == name for u in uniq):
                    continue
                w = int(rng.integers(1, 25)) if trial % 2 else 20
                got = tables.alleles_around_cut(uniq, name, cut, L, plot_window_size=w)
                p = tmp_path / "t.txt"
                tables.write_alleles_around_cut(got, str(p))
                assert p.read_text() == aggregate.alleles_around_cut(uniq, name, cut, L, w), (trial, cut, name, w)


# ---- the reference's second end-to-end test: CRISPResso_on_params (tests/Makefile) -------------------------------------------
def _params_golden(name="params_run.json.gz"):
    import numpy as np
    with gzip.open(os.path.join(HERE, "golden", name), "rt") as fh:
        g = json.load(fh)
    refs, names = {}, []
    for r in g["refs"]:
        d = dict(r)
        d["gap_incentive"] = np.array(r["gap_incentive"], dtype=int)
        d["include_idxs"] = np.array(r["include_idxs"])
        d["sequence_length"] = len(r["sequence"])
        refs[r["name"]] = d
        names.append(r["name"])
    return g, refs, names


def _compare_params(g, written, out_dir):
    for fn, text in g["files"].items():
        assert fn in written, fn
        with open(os.path.join(out_dir, fn)) as fh:
            assert fh.read() == text, fn
    return len(g["files"])


def test_params_run_tables_from_oracle_counts(tmp_path):
    """Two amplicons (FANC + the expected HDR allele), quantification window from coordinates, min_aln_score 80, three guides
    (named, flexible), --dsODN: the reads that survive the reference's quality filter + the per-amplicon records it derived
    (make_golden.py --params) -> oracle alignments, the reference's best-amplicon rule (CRISPRessoCORE.py:697-707), oracle
    aggregation -> 39 result files of the reference's run byte for byte (two of them are kept in its repository)."""
    import oracle
    from oracle import aggregate
    from crispresso2_amd import tables, counts as C
    from crispresso2_amd.pipeline import QuantResult
    g, refs, names = _params_golden()
    m = matrices()["EDNAFULL"]
    lines = g["fastq_after_quality_filter"].split("\n")
    reads = [lines[k] for k in range(1, len(lines) - 1, 4)]
    unique = {}
    for rd in reads:
        unique[rd] = unique.get(rd, 0) + 1
    go, ge = g["args"]["needleman_wunsch_gap_open"], g["args"]["needleman_wunsch_gap_extend"]
    stats = dict(N_TOT_READS=len(reads), N_READS_INPUT=g["alignment_stats"]["N_READS_INPUT"], N_READS_AFTER_PREPROCESSING=len(reads),
                 N_TOTAL=0, N_COMPUTED_ALN=0, N_CACHED_ALN=0, N_COMPUTED_NOTALN=0, N_CACHED_NOTALN=0)
    items, rows = {nm: [] for nm in names}, []
    to_first = {nm: [] for nm in names[1:]}
    for rd, c in unique.items():
        best, best_names, al = 0, [], {}
        for nm in names:                                            # forward strand: the reads of this file are
            s1, s2, score = oracle.global_align(rd, refs[nm]["sequence"], m, refs[nm]["gap_incentive"], go, ge)
            al[nm] = (s1, s2)
            if score > best and score > refs[nm]["min_aln_score"]:
                best, best_names = score, [nm]
            elif score == best:
                best_names.append(nm)
        if best <= 0:
            stats["N_COMPUTED_NOTALN"] += 1
            stats["N_CACHED_NOTALN"] += c - 1
            continue
        stats["N_COMPUTED_ALN"] += 1
        stats["N_CACHED_ALN"] += c - 1
        stats["N_TOTAL"] += c
        assert len(best_names) == 1                                 # no ambiguous read in this run
        nm = best_names[0]
        p = oracle.find_indels_substitutions(al[nm][0], al[nm][1], refs[nm]["include_idxs"])
        p["aln_seq"], p["aln_ref"] = al[nm]
        items[nm].append((p, c))
        if nm != names[0]:
            to_first[nm].append((al[names[0]][0], al[names[0]][1], c))        # ref_aln_details[0]: its alignment against the first amplicon
        mod = p["insertion_n"] + p["deletion_n"] + p["substitution_n"] > 0
        rows.append((al[nm][0], al[nm][1], nm, "MODIFIED" if mod else "UNMODIFIED", p["deletion_n"], p["insertion_n"], p["substitution_n"], c))
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS"):
        assert stats[k] == g["alignment_stats"][k], k
    per_ref = {nm: aggregate.aggregate(items[nm], len(refs[nm]["sequence"])) for nm in names}
    lay = C.CountLayout(2, max(len(refs[nm]["sequence"]) for nm in names), max(len(r) for r in reads))

    class WithAlleles(QuantResult):
        def alleles(self):
            return sorted([r + (r[7] / stats["N_TOTAL"] * 100,) for r in rows], key=lambda t: (-t[7], t[0], t[1]))
    res = WithAlleles(per_ref, stats, lay, None)
    # the run has an expected HDR amplicon: every amplicon's reads in the coordinates of the first one (:4195-4270)
    view = aggregate.remap_to_first_reference(per_ref[names[0]], to_first, len(refs[names[0]]["sequence"]), refs[names[0]]["include_idxs"])
    view[names[0]] = view.pop(None)
    res.first_ref_view = view
    written = tables.write_tables(res, refs, names, str(tmp_path), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(tmp_path)) == 39
    with pytest.raises(KeyError):                                   # the reference's own failure for an oligo of <= 6 bases
        tables.write_alleles_frequency_table(res, str(tmp_path / "x.txt"), dsODN="ACGTAC")


@pytest.mark.gpu
def test_params_run_tables_from_the_device_pipeline(tmp_path):
    """The same run through pipeline.quantify_fastq: native ingest, seeds, both amplicons aligned on the GPU, best-amplicon
    selection, count kernel -> the same 39 files."""
    import argparse
    from crispresso2_amd import _native, pipeline, tables
    g, refs, names = _params_golden()
    fq = tmp_path / "FANC.Cas9_filtered.fastq"
    fq.write_text(g["fastq_after_quality_filter"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], argparse.Namespace(**a), ctx=_native.default_context())
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
              "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS"):
        assert res.stats[k] == g["alignment_stats"][k], k
    res.stats["N_READS_INPUT"] = g["alignment_stats"]["N_READS_INPUT"]                 # reads before the quality filter (preprocessing: not here)
    res.stats["N_READS_AFTER_PREPROCESSING"] = g["alignment_stats"]["N_READS_AFTER_PREPROCESSING"]
    out = tmp_path / "CRISPResso_on_params"
    written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == 39


def test_params_run_tables_from_the_emulated_kernels(tmp_path):
    """The same run with the HIP kernels compiled for the host by the wave emulator (tests/emu): every read against both
    amplicons through the default launch chain, best amplicon from the 32-byte records with the reference's score expression,
    c2_count_vectors_kernel with the read multiplicities as weights -> the count tensor -> the 39 files.  (CPU stand-in for
    test_params_run_tables_from_the_device_pipeline: same kernels, same tables; only the host-side selection of pipeline.py is not exercised here.)"""
    import numpy as np
    import emu_driver as E
    from crispresso2_amd import tables
    from crispresso2_amd.pipeline import QuantResult
    g, refs, names = _params_golden()
    m = matrices()["EDNAFULL"]
    lines = g["fastq_after_quality_filter"].split("\n")
    reads = [lines[k] for k in range(1, len(lines) - 1, 4)]
    unique = {}
    for rd in reads:
        unique[rd] = unique.get(rd, 0) + 1
    ureads = list(unique)
    seqs = [refs[nm]["sequence"] for nm in names]
    st = {}
    res, rec = E.align_batch(ureads, seqs, [refs[nm]["gap_incentive"] for nm in names], [list(refs[nm]["include_idxs"]) for nm in names],
                             m, g["args"]["needleman_wunsch_gap_open"], g["args"]["needleman_wunsch_gap_extend"], all_refs=True, band_lanes=-87, stats=st)
    o1, o2 = st["raw"]
    k = len(names)
    weights = np.zeros(len(rec), dtype=np.uint32)
    stats = dict(N_TOT_READS=len(reads), N_READS_INPUT=g["alignment_stats"]["N_READS_INPUT"], N_READS_AFTER_PREPROCESSING=len(reads),
                 N_TOTAL=0, N_COMPUTED_ALN=0, N_CACHED_ALN=0, N_COMPUTED_NOTALN=0, N_CACHED_NOTALN=0)
    rows = []
    for i, rd in enumerate(ureads):
        best, best_refs = 0, []
        for r, nm in enumerate(names):
            t = rec[i * k + r]
            assert t["status"] == 0 and t["ref_id"] == r
            score = round(100 * int(t["matches"]) / float(int(t["aln_len"])), 3)
            if score > best and score > refs[nm]["min_aln_score"]:
                best, best_refs = score, [r]
            elif score == best:
                best_refs.append(r)
        c = unique[rd]
        if best <= 0:
            stats["N_COMPUTED_NOTALN"] += 1
            stats["N_CACHED_NOTALN"] += c - 1
            continue
        assert len(best_refs) == 1
        stats["N_COMPUTED_ALN"] += 1
        stats["N_CACHED_ALN"] += c - 1
        stats["N_TOTAL"] += c
        t = i * k + best_refs[0]
        weights[t] = c
        s1, s2 = res[t]
        mod = int(rec[t]["insertion_n"]) + int(rec[t]["deletion_n"]) + int(rec[t]["substitution_n"]) > 0
        rows.append((s1, s2, names[best_refs[0]], "MODIFIED" if mod else "UNMODIFIED", int(rec[t]["deletion_n"]), int(rec[t]["insertion_n"]),
                     int(rec[t]["substitution_n"]), c))
    counts, lay = E.count_vectors(o1, o2, rec, seqs, [list(refs[nm]["include_idxs"]) for nm in names], max(len(r) for r in ureads), weights=weights)
    per_ref = {nm: lay.unpack(counts, r, len(seqs[r])) for r, nm in enumerate(names)}

    class WithAlleles(QuantResult):
        def alleles(self):
            return sorted([r + (r[7] / stats["N_TOTAL"] * 100,) for r in rows], key=lambda t: (-t[7], t[0], t[1]))
    res_ = WithAlleles(per_ref, stats, lay, None)
    # the first-amplicon view, the way pipeline.py builds it: one more count launch per other amplicon over the alignments against
    # the FIRST amplicon, weighted with the multiplicities of the reads counted for that amplicon; row 0 of the result is the view
    view_keys = ["all_insertion_count_vectors", "all_insertion_left_count_vectors", "all_deletion_count_vectors",
                 "all_substitution_count_vectors"] + ["all_base_count_vectors_" + x for x in "ACGTN-"]
    view = {names[0]: {kk: per_ref[names[0]][kk] for kk in view_keys}}
    for r in range(1, k):
        w_r = np.zeros(len(rec), dtype=np.uint32)
        w_r[0::k] = weights[r::k]                                    # weight of task (i, r) moved onto task (i, 0)
        c_r, _ = E.count_vectors(o1, o2, rec, seqs, [list(refs[nm]["include_idxs"]) for nm in names], max(len(r_) for r_ in ureads), weights=w_r)
        u = lay.unpack(c_r, 0, len(seqs[0]))
        view[names[r]] = {kk: u[kk] for kk in view_keys}
    for nm in names:
        view[nm]["all_indelsub_count_vectors"] = (view[nm]["all_insertion_count_vectors"] + view[nm]["all_deletion_count_vectors"]
                                                  + view[nm]["all_substitution_count_vectors"])
    res_.first_ref_view = view
    written = tables.write_tables(res_, refs, names, str(tmp_path), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(tmp_path)) == 39


@pytest.mark.gpu
def test_params_run_from_the_unfiltered_fastq_with_the_fused_read_filter(tmp_path):
    """CRISPResso_on_params from its real input: FANC.Cas9.fastq with -q 30 fused into the ingest (no intermediate file) ->
    the same 39 files, READS IN INPUTS 250 / READS AFTER PREPROCESSING 231 from the library itself."""
    import argparse
    from crispresso2_amd import _native, pipeline, tables
    g, refs, names = _params_golden()
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(_golden()["fastq"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    a["min_average_read_quality"] = 30
    res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], argparse.Namespace(**a), ctx=_native.default_context())
    assert (res.stats["N_READS_INPUT"], res.stats["N_READS_AFTER_PREPROCESSING"]) == (250, 231)
    out = tmp_path / "CRISPResso_on_params"
    written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == 39


# ---- pipeline.py itself on the CPU: its device calls redirected to the wave emulator (tests/pipeline_on_emulator.py) --------
def _pipeline_args(extra=None):
    import argparse
    a = dict(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
             use_legacy_insertion_quantification=False, ignore_deletions=False, ignore_insertions=False,
             ignore_substitutions=False, assign_ambiguous_alignments_to_first_reference=False,
             expand_ambiguous_alignments=False, prime_editing_pegRNA_scaffold_seq="", discard_indel_reads=False)
    a.update(extra or {})
    return argparse.Namespace(**a)


def test_pipeline_host_logic_on_the_emulator_fanc_run(tmp_path):
    """pipeline.quantify_fastq end to end without a GPU: native ingest, seeds, emulated launch chain, selection, reverse-
    complement merge, emulated count kernel, allele rows from the string buffers -> the 18 files of the reference's FANC run."""
    from pipeline_on_emulator import emulated_device
    from crispresso2_amd import pipeline, tables, refs as RF
    g = _golden()
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(g["fastq"])
    cut = g["cut_point"]
    ref = RF.make_ref("Reference", g["amplicon"], [cut], [cut, cut + 1], min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [g["guide"]]
    with emulated_device():
        res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args())
        out = tmp_path / "out"
        names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert res.first_ref_view is None
    assert _compare(g, names, str(out)) == 18


def test_pipeline_host_logic_on_the_emulator_params_run_with_fused_filter_and_hdr_view(tmp_path):
    """The two-amplicon run from its raw FASTQ: read filter fused into the ingest, both amplicons per read, best-amplicon
    selection, count launches, the first-amplicon view launches (expected HDR amplicon) -> the 39 files."""
    from pipeline_on_emulator import emulated_device
    from crispresso2_amd import pipeline, tables
    g, refs, names = _params_golden()
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(_golden()["fastq"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    a["min_average_read_quality"] = 30
    with emulated_device():
        res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a))
        out = tmp_path / "out"
        written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert (res.stats["N_READS_INPUT"], res.stats["N_READS_AFTER_PREPROCESSING"]) == (250, 231)
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
              "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS"):
        assert res.stats[k] == g["alignment_stats"][k], k
    assert set(res.first_ref_view) == set(names)
    assert _compare_params(g, written, str(out)) == 39


def test_pipeline_on_the_emulator_both_strand_batch_feeds_counts_view_and_alleles(tmp_path):
    """Same run with every third unique read reverse-complemented and the seed test made inconclusive (aln_seed_min beyond the
    number of seeds), so that every (read, amplicon) pair is also aligned as its reverse complement in the second batch and the
    reverse-complemented reads take their alignments -- counts, first-amplicon view and allele rows -- from THAT batch.  No
    read coexists with its reverse complement, so by symmetry the reference's result files are the same 39."""
    from pipeline_on_emulator import emulated_device
    from crispresso2_amd import pipeline, tables, refs as RF
    g, refs, names = _params_golden()
    lines = g["fastq_after_quality_filter"].split("\n")
    recs, order = [], {}
    for k in range(0, len(lines) - 1, 4):
        rid, seq, plus, qual = lines[k:k + 4]
        if order.setdefault(seq, len(order)) % 3 == 1:               # all copies of a read together: no read meets its reverse complement
            seq, qual = RF.reverse_complement(seq), qual[::-1]
        recs.append("%s\n%s\n%s\n%s\n" % (rid, seq, plus, qual))
    fq = tmp_path / "mixed_strands.fastq"
    fq.write_text("".join(recs))
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    a["aln_seed_min"] = 1000
    with emulated_device():
        res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a))
        S = res.host_view()
        assert res._state["r2"] is not None and S["use2"][:, 0].sum() > 30 and (~S["use2"][:, 0]).sum() > 60
        # the seed test ran in c2_strand_plan_kernel; the host's c2_strand_plan gives the same run
        pipeline.FORCE_HOST_STRAND_PLAN = True
        try:
            res_h = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a))
        finally:
            pipeline.FORCE_HOST_STRAND_PLAN = False
        assert res_h.stats == res.stats and np.array_equal(res_h.host_view()["use2"], S["use2"])
        assert all(np.array_equal(res_h.per_ref[nm][kk], vv) if isinstance(vv, np.ndarray) else res_h.per_ref[nm][kk] == vv
                   for nm in names for kk, vv in res.per_ref[nm].items())
        res.stats["N_READS_INPUT"] = 250                              # the file fed here is the already filtered one
        out = tmp_path / "out"
        written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == 39


# ---- two different amplicons in one run: the reference's pooled test reads (FANC + HEK3) through CRISPResso core -----------------
def _both_run(tmp_path, ctx=None):
    from crispresso2_amd import pipeline, tables
    g, refs, names = _params_golden("both_run.json.gz")
    fq = tmp_path / "Both.Cas9.fastq"
    fq.write_text(g["fastq"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a), ctx=ctx)
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
              "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "N_READS_INPUT"):
        assert res.stats[k] == g["alignment_stats"][k], k
    assert res.first_ref_view is None                                # no HDR amplicon, no prime-editing extension
    out = tmp_path / "CRISPResso_on_Both.Cas9"
    written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == 33
    assert res.per_ref["FANC"]["counts_total"] > 200 and res.per_ref["HEK3"]["counts_total"] > 200


def test_pipeline_on_the_emulator_two_amplicon_run_of_the_pooled_test_reads(tmp_path):
    """500 reads of two unrelated amplicons (223 and 2xx bp), each read against both on the emulated launch chain (the reads of
    the other amplicon are the 'unrelated read' case of the band certificate), assigned to the better one -> the 33 result
    files of the reference's run with `-a A,B -an FANC,HEK3 -g gA,gB` (make_golden.py --both)."""
    from pipeline_on_emulator import emulated_device
    with emulated_device():
        _both_run(tmp_path)


@pytest.mark.gpu
def test_two_amplicon_run_of_the_pooled_test_reads_on_the_device(tmp_path):
    from crispresso2_amd import _native
    _both_run(tmp_path, ctx=_native.default_context())


# ---- a prime-editing run: Reference + Prime-edited amplicon, ambiguous reads, first-amplicon view -----------------------------
def _pe_run(tmp_path, ctx=None):
    from crispresso2_amd import pipeline, tables
    g, refs, names = _params_golden("pe_run.json.gz")
    assert names == ["Reference", "Prime-edited"] and g["args"]["prime_editing_pegRNA_extension_seq"]
    fq = tmp_path / "pe.fastq"
    fq.write_text(g["fastq"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a), ctx=ctx)
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
              "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "N_READS_INPUT"):
        assert res.stats[k] == g["alignment_stats"][k], k
    assert res.stats["N_AMBIGUOUS"] > 0 and set(res.first_ref_view) == set(names)
    out = tmp_path / "CRISPResso_on_pe"
    written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == 35
    amb = [l for l in (out / "Alleles_frequency_table.txt").read_text().split("\n") if "\tAMBIGUOUS_Reference\t" in l]
    assert len(amb) >= 5


def test_pipeline_on_the_emulator_prime_editing_run(tmp_path):
    """Reference's main() with a pegRNA spacer + extension (make_golden.py --pe): it derives the 'Prime-edited' amplicon; reads
    that do not reach the edited base tie between the two amplicons (AMBIGUOUS_ rows of the allele table, not counted for
    either), and the reads counted for 'Prime-edited' are also viewed in the first amplicon's coordinates -> its 35 files."""
    from pipeline_on_emulator import emulated_device
    with emulated_device():
        _pe_run(tmp_path)
    from crispresso2_amd import pipeline
    g, refs, names = _params_golden("pe_run.json.gz")
    a = dict(g["args"], prime_editing_pegRNA_scaffold_seq="GGCACCGAGTCGGTGC")
    with pytest.raises(ValueError):                                 # a scaffold sequence without the (index, dna) of get_pe_scaffold_search
        pipeline.quantify_unique(None, None, [1], refs, names, matrices()["EDNAFULL"], _pipeline_args(a))
    odd = {n: dict(refs[n], sequence=refs[n]["sequence"][:5] + "R" + refs[n]["sequence"][6:]) for n in names}
    with pytest.raises(NotImplementedError):                        # the legacy classifier's nucSet rule for a non-ACGTN reference base is not on the count route
        pipeline.quantify_unique(None, None, [1], odd, names, matrices()["EDNAFULL"], _pipeline_args(dict(g["args"], use_legacy_insertion_quantification=True)))


@pytest.mark.gpu
def test_prime_editing_run_on_the_device(tmp_path):
    from crispresso2_amd import _native
    _pe_run(tmp_path, ctx=_native.default_context())


# ---- --use_legacy_insertion_quantification: find_indels_substitutions_legacy on the count route ------------------------------------
def _legacy_run(tmp_path, ctx=None):
    from crispresso2_amd import pipeline, tables
    g, refs, names = _params_golden("legacy_run.json.gz")
    assert g["args"]["use_legacy_insertion_quantification"] is True
    fq = tmp_path / "legacy.fastq"
    fq.write_text(g["fastq"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a), ctx=ctx)
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
              "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "N_READS_INPUT"):
        assert res.stats[k] == g["alignment_stats"][k], k
    out = tmp_path / "CRISPResso_on_legacy"
    written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == len(g["files"])
    # the statistics come from c2_select_best_kernel's uint64 accumulators (stored as int64); the host restatement of the
    # selection must give the same numbers under the legacy classifier too (whose window counts can exceed the all_* counts:
    # N_MODS_OUTSIDE_WINDOW is then a difference of wrapped terms on both routes)
    pipeline.FORCE_HOST_SELECTION = True
    try:
        res_h = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a), ctx=ctx)
    finally:
        pipeline.FORCE_HOST_SELECTION = False
    assert res_h.stats == res.stats
    # the same reads under the default classifier give different tables (the golden pins the legacy rules, not the common part)
    res2 = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(dict(a, use_legacy_insertion_quantification=False)), ctx=ctx)
    assert res2.stats["N_MODS_IN_WINDOW"] != res.stats["N_MODS_IN_WINDOW"] or res2.stats["N_MODS_OUTSIDE_WINDOW"] != res.stats["N_MODS_OUTSIDE_WINDOW"]


def test_pipeline_on_the_emulator_legacy_insertion_quantification_run(tmp_path):
    """make_golden.py --legacy: the reference's main() with --use_legacy_insertion_quantification -w 2 over FANC reads plus reads
    built for the legacy corner cases (one-flank insertions, trailing / leading / column-1 deletions) -> its 18 result files,
    from the fused classifier's legacy rules and the count kernel's legacy coordinates."""
    from pipeline_on_emulator import emulated_device
    with emulated_device():
        _legacy_run(tmp_path)


@pytest.mark.gpu
def test_legacy_insertion_quantification_run_on_the_device(tmp_path):
    from crispresso2_amd import _native
    _legacy_run(tmp_path, ctx=_native.default_context())


# ---- prime editing with a scaffold sequence: the 'Scaffold-incorporated' amplicon nothing is aligned to -----------------------------
def _pe_scaffold_run(tmp_path, ctx=None):
    from crispresso2_amd import pipeline, tables
    g, refs, names = _params_golden("pe_scaffold_run.json.gz")
    assert names == ["Reference", "Prime-edited", "Scaffold-incorporated"]
    assert refs["Scaffold-incorporated"]["sequence"] == refs["Prime-edited"]["sequence"]          # the reference's deepcopy (:3759-3764)
    fq = tmp_path / "pes.fastq"
    fq.write_text(g["fastq"])
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    res = pipeline.quantify_fastq(str(fq), refs, names[:2], matrices()["EDNAFULL"], _pipeline_args(a), ctx=ctx,
                                  pe_scaffold_dna_info=tuple(g["pe_scaffold_dna_info"]))
    for k in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
              "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "N_READS_INPUT"):
        assert res.stats[k] == g["alignment_stats"][k], k
    assert res.per_ref["Scaffold-incorporated"]["counts_total"] >= 10 and set(res.first_ref_view) == set(names)
    out = tmp_path / "CRISPResso_on_pes"
    written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
    assert _compare_params(g, written, str(out)) == 51


def test_pipeline_on_the_emulator_prime_editing_scaffold_run(tmp_path):
    """make_golden.py --pe-scaffold: the reference's run with --prime_editing_pegRNA_scaffold_seq over reads in which reverse
    transcription ran on into the scaffold.  The count route pulls the aligned strings of the reads whose best amplicons
    include 'Prime-edited', applies the substring rule (:786-796), counts the hits for 'Scaffold-incorporated' in a launch of
    their own and labels their allele rows -> the reference's 51 result files (three amplicons)."""
    from pipeline_on_emulator import emulated_device
    with emulated_device():
        _pe_scaffold_run(tmp_path)


@pytest.mark.gpu
def test_prime_editing_scaffold_run_on_the_device(tmp_path):
    from crispresso2_amd import _native
    _pe_scaffold_run(tmp_path, ctx=_native.default_context())


def _variant_scaffold_check(ctx):
    """variants.get_new_variant_objects with the scaffold rule against the reference's get_new_variant_object (make_golden.py
    --variant-scaffold): every dict of the 121 unique reads of the scaffold run, eight of them re-labelled 'Scaffold-incorporated'."""
    import types
    from helpers import load_golden
    from test_gpu_parity import _variant_equal
    from crispresso2_amd import variants as V
    g, refs, names = _params_golden("pe_scaffold_run.json.gz")
    gold = load_golden("variants_scaffold.json.gz")
    args = types.SimpleNamespace(**g["args"])
    got = V.get_new_variant_objects(args, gold["reads"], refs, names[:2], matrices()["EDNAFULL"], tuple(g["pe_scaffold_dna_info"]), ctx=ctx)
    assert len(got) == len(gold["variants"])
    for a, e in zip(got, gold["variants"]):
        _variant_equal(a, e)
    assert sum(v.get("class_name") == "Scaffold-incorporated" for v in got) == gold["n_scaffold"] == 8


def test_per_read_scaffold_rule_on_the_emulator():
    from pipeline_on_emulator import EmulatedContext, emulated_device
    with emulated_device():
        _variant_scaffold_check(EmulatedContext())


@pytest.mark.gpu
def test_per_read_scaffold_rule_on_the_device():
    from crispresso2_amd import _native
    _variant_scaffold_check(_native.default_context())


# ---- the reference's other test FASTQs (HEK3 amplicon; untreated FANC sample) as single-amplicon runs --------------------------------
def _single_runs(tmp_path, ctx=None):
    from helpers import load_golden
    from crispresso2_amd import pipeline, tables, refs as RF
    for k, c in enumerate(load_golden("single_runs.json.gz")):
        ref = RF.make_ref("Reference", c["amplicon"], c["cut_points"], c["include_idxs"], min_aln_score=60)
        ref["sgRNA_orig_sequences"] = [c["guide"]]
        fq = tmp_path / c["fastq_name"]
        fq.write_text(c["fastq"])
        res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args(), ctx=ctx)
        for key in ("N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN", "N_TOT_READS", "N_GLOBAL_SUBS",
                    "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "N_READS_INPUT"):
            assert res.stats[key] == c["alignment_stats"][key], (c["fastq_name"], key)
        out = tmp_path / ("out%d" % k)
        written = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
        assert _compare_params(c, written, str(out)) == 18, c["fastq_name"]


def test_pipeline_on_the_emulator_hek3_and_untreated_runs(tmp_path):
    """make_golden.py --single-runs: tests/HEK3.Cas9.fastq (another amplicon, other read shapes) and tests/FANC.Untreated.fastq
    through the reference's main(); 18 result files each from pipeline.py on the emulated kernels."""
    from pipeline_on_emulator import emulated_device
    with emulated_device():
        _single_runs(tmp_path)


@pytest.mark.gpu
def test_hek3_and_untreated_runs_on_the_device(tmp_path):
    from crispresso2_amd import _native
    _single_runs(tmp_path, ctx=_native.default_context())


@pytest.mark.parametrize("run", ["fanc", "params", "one_batch"])
def test_streamed_ingest_in_many_small_batches_gives_the_same_files(tmp_path, monkeypatch, run):
    """pipeline.quantify_fastq's streamed flow forced into MANY batches (chunks of 3 x 2 KiB of text, a device batch for every 15 new
    unique reads: different longest reads per batch, so the batches' output strides differ and are widened when joined) -- the same
    18 / 39 files of the reference's runs; `one_batch` is the same file through stream=False, and both flows must agree on every
    statistic and tensor."""
    from pipeline_on_emulator import emulated_device
    from crispresso2_amd import pipeline, tables, refs as RF
    monkeypatch.setenv("C2_FASTQ_THREADS", "3")
    monkeypatch.setenv("C2_FASTQ_RANGE_BYTES", "2048")
    monkeypatch.setattr(pipeline, "STREAM_MIN_BATCH", 15)
    if run == "params":
        g, refs, names = _params_golden()
        fq = tmp_path / "FANC.Cas9.fastq"
        fq.write_text(_golden()["fastq"])
        a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
        a["min_average_read_quality"] = 30
        with emulated_device():
            res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a))
            out = tmp_path / "out"
            written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
        assert (res.stats["N_READS_INPUT"], res.stats["N_READS_AFTER_PREPROCESSING"]) == (250, 231)
        assert _compare_params(g, written, str(out)) == 39
        return
    g = _golden()
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(g["fastq"] + "@trailing_id_only\n")                 # + a record cut short after its id: the empty sequence, dropped
    cut = g["cut_point"]
    ref = RF.make_ref("Reference", g["amplicon"], [cut], [cut, cut + 1], min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [g["guide"]]
    with emulated_device():
        tm = {}
        res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args(), timings=tm,
                                      stream=(run != "one_batch"))
        assert ("ingest_dedup_streamed" in tm) == (run != "one_batch")
        assert run == "one_batch" or tm["stream_batches"] >= 8
        other = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args(),
                                        stream=(run == "one_batch"))
        assert other.stats == res.stats
        for kk, vv in res.per_ref["Reference"].items():
            ww = other.per_ref["Reference"][kk]
            assert np.array_equal(vv, ww) if isinstance(vv, np.ndarray) else vv == ww, kk
        assert res.alleles() == other.alleles()
        res.stats["N_READS_INPUT"] = res.stats["N_READS_AFTER_PREPROCESSING"] = 250       # (the extra id line is not part of the reference's run)
        out = tmp_path / "out"
        names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert _compare(g, names, str(out)) == 18


def test_partner_search_on_the_device_gives_the_same_run(tmp_path, monkeypatch):
    """Reads of one length with reverse-complement partners among them: pipeline.quantify_fastq with the partner search on the
    "device" (torch ops over the read matrix; RC_PARTNERS_ON_DEVICE_MIN lowered) -- streamed in several batches and in one batch --
    against the host search: same statistics, tensors and allele rows (the merged reads' copies go to the first of each pair)."""
    from pipeline_on_emulator import emulated_device
    from crispresso2_amd import pipeline, synth, refs as RF
    L = 150
    amp, g_, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, 90)
    seqs = [r.tobytes().decode() for r in reads]
    seqs = seqs + [RF.reverse_complement(s_) for s_ in seqs[:25]] + seqs[:10]
    fq = tmp_path / "rc.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (k, s_, "I" * L) for k, s_ in enumerate(seqs)))
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    monkeypatch.setenv("C2_FASTQ_THREADS", "2")
    monkeypatch.setenv("C2_FASTQ_RANGE_BYTES", "4096")
    monkeypatch.setattr(pipeline, "STREAM_MIN_BATCH", 20)
    results = []
    with emulated_device():
        for dev_min, stream in ((10**9, True), (1, True), (1, False)):
            monkeypatch.setattr(pipeline, "RC_PARTNERS_ON_DEVICE_MIN", dev_min)
            called = []
            orig = pipeline.rc_partners_device
            monkeypatch.setattr(pipeline, "rc_partners_device", lambda m_: (called.append(m_.shape), orig(m_))[1])     # (-> _DevicePartners)
            res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args(), stream=stream)
            monkeypatch.setattr(pipeline, "rc_partners_device", orig)
            assert bool(called) == (dev_min == 1)
            results.append((res.stats, res.per_ref["Reference"], res.alleles()))
    st0, pr0, al0 = results[0]
    assert st0["N_TOTAL"] > 100 and len(al0) < len(set(seqs))           # merged pairs: fewer rows than unique reads
    for st, pr, al in results[1:]:
        assert st == st0 and al == al0
        for kk, vv in pr0.items():
            assert np.array_equal(vv, pr[kk]) if isinstance(vv, np.ndarray) else vv == pr[kk], kk


def test_device_ingest_gives_the_same_files(tmp_path, monkeypatch):
    """pipeline.quantify_fastq with the text framed and de-duplicated by the c2_fq_* kernels (fastq_device; chunks of one tile) instead
    of the host parser: the 18 files of the reference's FANC.Cas9 run, and every statistic, tensor and allele row of the host-parser
    flow.  The file carries a record cut short after its id line (the empty sequence: dropped by both)."""
    from pipeline_on_emulator import emulated_device
    from test_fastq_device_emulated import emulated_fq_kernels
    from crispresso2_amd import pipeline, tables, refs as RF
    g = _golden()
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(g["fastq"] + "@trailing_id_only\n")
    cut = g["cut_point"]
    ref = RF.make_ref("Reference", g["amplicon"], [cut], [cut, cut + 1], min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [g["guide"]]
    monkeypatch.setattr(pipeline, "STREAM_MIN_BATCH", 15)             # (a batch of alignments whenever 15 new unique reads are final)
    with emulated_device(), emulated_fq_kernels():
        monkeypatch.setenv("C2_FQ_INGEST", "device")
        tm = {}
        res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args(), timings=tm)
        assert getattr(res, "ingest_route", None) == "device" and "host_parser_because" not in tm
        assert tm["stream_batches"] >= 4, tm
        monkeypatch.setenv("C2_FQ_INGEST", "host")
        tm = {}
        other = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args(), timings=tm)
        assert tm["host_parser_because"] == "C2_FQ_INGEST=host" and "ingest_dedup_streamed" in tm
        assert other.stats == res.stats
        for kk, vv in res.per_ref["Reference"].items():
            ww = other.per_ref["Reference"][kk]
            assert np.array_equal(vv, ww) if isinstance(vv, np.ndarray) else vv == ww, kk
        assert res.alleles() == other.alleles()
        res.stats["N_READS_INPUT"] = res.stats["N_READS_AFTER_PREPROCESSING"] = 250       # (the extra id line is not part of the reference's run)
        out = tmp_path / "out"
        names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert _compare(g, names, str(out)) == 18


def test_device_ingest_with_both_strand_reads_and_partners(tmp_path, monkeypatch):
    """reads whose seeds leave the strand open (aligned on both: the second batch is gathered ON the device from the device arena)
    and reverse-complement partners (device search): device ingest = host parser"""
    from pipeline_on_emulator import emulated_device
    from test_fastq_device_emulated import emulated_fq_kernels
    from crispresso2_amd import pipeline, synth, refs as RF
    L = 150
    amp, g_, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, 60)
    seqs = [r.tobytes().decode() for r in reads]
    seqs = seqs + [RF.reverse_complement(s_) for s_ in seqs[:15]] + seqs[:10] + ["ACGT" * 30, "TTTTGGGGCCCC" * 9]      # (the last two: no seed hits)
    fq = tmp_path / "rc.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (k, s_, "I" * len(s_)) for k, s_ in enumerate(seqs)))
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    results = []
    with emulated_device(), emulated_fq_kernels():
        for route, dev_min in (("host", 10**9), ("device", 10**9), ("device", 1)):
            monkeypatch.setenv("C2_FQ_INGEST", route)
            monkeypatch.setattr(pipeline, "RC_PARTNERS_ON_DEVICE_MIN", dev_min)
            res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args())
            assert (getattr(res, "ingest_route", None) == "device") == (route == "device")
            results.append((res.stats, res.per_ref["Reference"], res.alleles(), res.host_view()["slot2"]))
    st0, pr0, al0, slot2 = results[0]
    assert (slot2 >= 0).sum() >= 2                                   # (the second batch ran)
    for st, pr, al, _ in results[1:]:
        assert st == st0 and al == al0
        for kk, vv in pr0.items():
            assert np.array_equal(vv, pr[kk]) if isinstance(vv, np.ndarray) else vv == pr[kk], kk


def _many_references_run(tmp_path, ctx=None, n_refs=70, expand=True):
    """more references than one 64-bit mask holds: the device selection (two words per read) against the host restatement of the
    reference's loop (pipeline._select_on_host) -- statistics, tensors, allele rows; some references are identical (ties: ambiguous reads)"""
    from crispresso2_amd import pipeline, refs as RF
    rng = np.random.default_rng(70)
    L = 64
    seqs = ["".join(rng.choice(list("ACGT"), L)) for _ in range(n_refs)]
    seqs[66] = seqs[2]                                              # a tie across the word boundary
    seqs[69] = seqs[65]                                             # ... and inside the second word
    names = ["amp%d" % i for i in range(n_refs)]
    refs = {nm: RF.make_ref(nm, sq, [L // 2], [L // 2 - 1, L // 2], min_aln_score=60) for nm, sq in zip(names, seqs)}
    reads = []
    for i in (0, 2, 2, 63, 64, 65, 65, 66, 69, 33, 68):
        s_ = list(seqs[i])
        if i % 3 == 0:
            s_[10] = "A" if s_[10] != "A" else "C"
        reads.append("".join(s_))
    reads.append(seqs[40][:30] + seqs[40][34:])                     # a deletion
    reads.append("".join(rng.choice(list("ACGT"), L)))              # aligns to nothing
    fq = tmp_path / "many.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (k, s_, "I" * len(s_)) for k, s_ in enumerate(reads)))
    out = []
    runs = [(False, {}), (True, {}), (False, {"assign_ambiguous_alignments_to_first_reference": True}), (True, {"assign_ambiguous_alignments_to_first_reference": True})]
    if expand:                                                      # (the emulator run leaves this pair to the GPU test: a third of its time)
        runs += [(False, {"expand_ambiguous_alignments": True}), (True, {"expand_ambiguous_alignments": True})]
    for host, mode in runs:
        pipeline.FORCE_HOST_SELECTION = host
        try:
            res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(mode), ctx=ctx)
        finally:
            pipeline.FORCE_HOST_SELECTION = False
        out.append(res)
    for dev_res, host_res in zip(out[0::2], out[1::2]):
        assert dev_res.stats == host_res.stats
        assert dev_res.alleles() == host_res.alleles()
        for nm in names:
            for kk, vv in dev_res.per_ref[nm].items():
                ww = host_res.per_ref[nm][kk]
                assert np.array_equal(vv, ww) if isinstance(vv, np.ndarray) else vv == ww, (nm, kk)
    assert out[0].stats["N_AMBIGUOUS"] >= 5                          # the reads of the identical amplicons
    assert out[2].per_ref["amp66"]["counts_total"] == 0 and out[2].per_ref["amp2"]["counts_total"] >= 2   # first reference only
    if expand:
        assert out[4].per_ref["amp66"]["counts_total"] == out[4].per_ref["amp2"]["counts_total"] >= 2   # expanded: counted for both


def test_more_than_64_references_select_on_the_device_emulator(tmp_path):
    from pipeline_on_emulator import emulated_device
    with emulated_device():
        _many_references_run(tmp_path, expand=False)


@pytest.mark.gpu
def test_more_than_64_references_select_on_the_device(tmp_path):
    from crispresso2_amd import _native
    _many_references_run(tmp_path, ctx=_native.default_context())


@pytest.mark.parametrize("kind", ["filtered", "gz", "bgzf"])
def test_device_ingest_of_text_the_host_inflated_or_filtered(tmp_path, monkeypatch, kind):
    """compressed / quality-filtered input: the host inflates / filters into memory (c2_fastq_stream_open), the text is framed and
    de-duplicated by the c2_fq_* kernels from there (the host parser never runs) -- the reference's 39 files of its params run
    (-q 30: N_READS_INPUT 250, 231 after the filter) and the FANC run's 18 files from a gzip / BGZF file"""
    import gzip
    from pipeline_on_emulator import emulated_device
    from test_fastq_device_emulated import emulated_fq_kernels
    from crispresso2_amd import pipeline, tables, refs as RF, synth
    monkeypatch.setenv("C2_FQ_INGEST", "device")
    monkeypatch.setattr(pipeline, "STREAM_MIN_BATCH", 25)
    if kind == "filtered":
        g, refs, names = _params_golden()
        fq = tmp_path / "FANC.Cas9.fastq"
        fq.write_text(_golden()["fastq"])
        a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
        a["min_average_read_quality"] = 30
        with emulated_device(), emulated_fq_kernels():
            tm = {}
            res = pipeline.quantify_fastq(str(fq), refs, names, matrices()["EDNAFULL"], _pipeline_args(a), timings=tm)
            out = tmp_path / "out"
            written = tables.write_tables(res, refs, names, str(out), plot_window_size=g["args"]["plot_window_size"], dsODN=g["args"]["dsODN"])
        assert res.ingest_route == "device, text from host memory" and "host_parser_because" not in tm and tm["stream_batches"] >= 2
        assert (res.stats["N_READS_INPUT"], res.stats["N_READS_AFTER_PREPROCESSING"]) == (250, 231)
        assert _compare_params(g, written, str(out)) == 39
        return
    g = _golden()
    plain = tmp_path / "FANC.Cas9.fastq"
    plain.write_text(g["fastq"])
    fq = tmp_path / "FANC.Cas9.fastq.gz"
    if kind == "gz":
        with gzip.open(fq, "wt") as fh:
            fh.write(g["fastq"])
    else:
        synth.write_bgzf(str(plain), str(fq), workers=2, level=1)
    cut = g["cut_point"]
    ref = RF.make_ref("Reference", g["amplicon"], [cut], [cut, cut + 1], min_aln_score=60)
    ref["sgRNA_orig_sequences"] = [g["guide"]]
    with emulated_device(), emulated_fq_kernels():
        res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args())
        assert res.ingest_route == ("device, text from host memory" if kind == "gz" else "device, members inflated into the upload buffers")
        out = tmp_path / "out"
        names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert _compare(g, names, str(out)) == 18


@pytest.mark.parametrize("lower_case", [False, True])
def test_count_transfer_on_the_device_equals_the_sequential_loop(tmp_path, monkeypatch, lower_case):
    """the reverse-complement count transfer (CRISPRessoCORE.py:3970-3975), N_TOTAL / N_AMBIGUOUS and the weights of the count pass
    computed on the device over partner pairs (pipeline's default for unsharded runs) against the reference's sequential loop on the
    host (FORCE_HOST_MERGE): pairs both of which aligned, pairs one of which did not, a read that is its own reverse complement,
    reads with several copies.  lower_case: a lower-case read's reverse complement is upper case -- the partner relation is not
    symmetric then, and the device pass must leave the run to the host loop (same result either way)."""
    from pipeline_on_emulator import emulated_device
    from crispresso2_amd import pipeline, synth, refs as RF
    L = 120
    amp, g_, inc = synth.amplicon_setup(L)
    half = amp[:L // 2]
    pal = half + RF.reverse_complement(half)                          # its own reverse complement
    reads = [r.tobytes().decode() for r in synth.make_reads(L, 40)]
    seqs = reads + [RF.reverse_complement(s_) for s_ in reads[:12]] * 2 + reads[:8] * 3 + [pal] * 3
    seqs += ["ACGT" * 30, RF.reverse_complement("ACGT" * 30)]          # a pair neither of which aligns
    if lower_case:
        seqs += [reads[3].lower(), reads[4].lower()] * 2
    fq = tmp_path / "merge.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (k_, s_, "I" * len(s_)) for k_, s_ in enumerate(seqs)))
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    ref2 = RF.make_ref("Palindrome", pal, [L // 2], inc, min_aln_score=60)
    out = []
    with emulated_device():
        for host in (False, True):
            monkeypatch.setattr(pipeline, "FORCE_HOST_MERGE", host)
            tm = {}
            res = pipeline.quantify_fastq(str(fq), {"Reference": ref, "Palindrome": ref2}, ["Reference", "Palindrome"], matrices()["EDNAFULL"],
                                          _pipeline_args(), timings=tm)
            out.append((res.stats, res.per_ref, res.alleles(), res.host_view()["cnt"].tolist()))
    assert out[0][0] == out[1][0] and out[0][2] == out[1][2] and out[0][3] == out[1][3]
    for nm in ("Reference", "Palindrome"):
        for kk, vv in out[0][1][nm].items():
            ww = out[1][1][nm][kk]
            assert np.array_equal(vv, ww) if isinstance(vv, np.ndarray) else vv == ww, (nm, kk)
    if not lower_case:
        assert out[0][1]["Palindrome"]["counts_total"] == 6          # the palindrome's three copies, doubled (the reference's own arithmetic)
    assert sum(1 for c_ in out[0][3] if c_ == 0) >= 8                 # the partners that gave their copies away


def test_one_gzip_member_through_the_segment_route_gives_the_plain_files_result(tmp_path, monkeypatch):
    """pipeline.quantify_fastq over an ordinary one-member .gz whose segments are inflated straight into the (emulated) device text
    (c2_gzseg_open; thresholds lowered so that a file of a few hundred KB is cut into several segments) against the same text as a plain file and
    against the same .gz with the route switched off (the host inflates the whole member first): statistics and count tensors agree."""
    import gzip
    from pipeline_on_emulator import emulated_device
    from test_fastq_device_emulated import emulated_fq_kernels
    from crispresso2_amd import pipeline, refs as RF, synth
    monkeypatch.setenv("C2_FQ_INGEST", "device")
    monkeypatch.setattr(pipeline, "STREAM_MIN_BATCH", 100)
    L = 100
    amp, g_, inc = synth.amplicon_setup(L)
    uniq = [r.tobytes().decode() for r in synth.make_reads(L, 300)]
    rng = np.random.default_rng(5)
    order = rng.integers(0, len(uniq), 16000)
    text = "".join("@read_%07d_%d\n%s\n+\n%s\n" % (k, int(rng.integers(0, 1 << 30)), uniq[int(i)], "".join(chr(33 + int(q)) for q in rng.integers(20, 41, L)))
                   for k, i in enumerate(order))
    plain = tmp_path / "reads.fastq"
    plain.write_text(text)
    gz = tmp_path / "reads.fastq.gz"
    gz.write_bytes(gzip.compress(text.encode(), 6))
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    results = {}
    with emulated_device(), emulated_fq_kernels():
        for name, path, env in (("plain", plain, {}), ("segments", gz, {"C2_GZ_PARALLEL_MIN": "0", "C2_GZ_PARALLEL_CHUNK": "32768"}),
                                ("whole", gz, {"C2_GZ_PARALLEL": "0"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            res = pipeline.quantify_fastq(str(path), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], _pipeline_args())
            for k_ in env:
                monkeypatch.delenv(k_)
            results[name] = res
    assert results["segments"].ingest_route == "device, one gzip member inflated segment by segment into the upload buffers"
    assert results["whole"].ingest_route == "device, text from host memory"
    base = results["plain"]
    for name in ("segments", "whole"):
        other = results[name]
        assert other.stats == base.stats, name
        for kk, vv in base.per_ref["Reference"].items():
            ww = other.per_ref["Reference"][kk]
            assert np.array_equal(vv, ww) if isinstance(vv, np.ndarray) else vv == ww, (name, kk)
    assert base.stats["N_TOT_READS"] == 16000
