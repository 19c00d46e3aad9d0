"""Every result table of the reference's own end-to-end run (CRISPResso -r1 FANC.Cas9.fastq -a ... -g ...), byte for byte.

tests/golden/fanc_full_run.json.gz holds every .txt file the REFERENCE wrote for that run in the dev container
(make_golden.py --fanc-full: its main(), plot functions replaced by no-ops) plus the unzipped allele table.  tables.write_tables
must reproduce, from the count tensor: quantification of editing frequency, mapping statistics, nucleotide frequency /
percentage tables (amplicon and quantification window), modification count vectors (both), the four effect vectors, the
four histograms and the allele frequency table.  CPU: counts built by the oracle (oracle/aggregate.py); GPU: the
device-resident pipeline.  Not reproduced (plot data preparation, out of scope): Alleles_frequency_table_around_sgRNA_*.txt,
Alleles_homology_scores.txt."""
import gzip
import json
import os

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
        if fn in NOT_REPRODUCED or fn.startswith("Alleles_frequency_table_around_") or fn in skip:
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
    assert _compare(g, names, str(tmp_path)) == 17
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
    args = argparse.Namespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                              use_legacy_insertion_quantification=False, ignore_deletions=False, ignore_insertions=False,
                              ignore_substitutions=False, assign_ambiguous_alignments_to_first_reference=False,
                              expand_ambiguous_alignments=False, prime_editing_pegRNA_scaffold_seq="", discard_indel_reads=False)
    res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], args, ctx=_native.default_context())
    out = tmp_path / "CRISPResso_on_FANC.Cas9"
    names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    assert _compare(g, names, str(out)) == 17
