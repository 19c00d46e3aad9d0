"""paired_device.quantify_paired_fastq -- two FASTQ files -> count tensors on the device, no dict per pair -- against the run of the REFERENCE
recorded in tests/golden/paired_fastq.json.gz (its final variantCache after process_paired_fastq: consensus reads, copies, every payload, and
its aln_stats), through the restatement of the aggregation loop (oracle/aggregate.py: reverse-complement merge, ambiguity, per-amplicon
vectors).  On the CPU the device calls land on the wave emulator; with -m gpu on the MI355X."""
import json
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import load_golden, matrices          # noqa: E402


def expected_from_reference_cache(case, refs, names):
    """the reference's cache (keys in order, copies, payload dicts) -> what its aggregation loop makes of it (CRISPRessoCORE.py:3964-4115)"""
    from crispresso2_amd import variant_io as IO
    from oracle import aggregate as AG
    from oracle.fastq import reverse_complement
    res = case["result"]
    cache = {}
    for key, cnt, js in zip(res["aligned"], res["counts"], res["variants"]):
        v = json.loads(js, cls=IO.CRISPRessoJSONDecoder)
        v["count"] = cnt
        cache[key] = v
    a = case["args"]
    n_total = n_amb = 0
    items = {nm: [] for nm in names}
    rows = []
    for key in cache:
        c = cache[key]["count"]
        if c == 0:
            continue
        rc = reverse_complement(key)
        if rc in cache and cache[rc]["count"] > 0:
            c += cache[rc]["count"]
            cache[rc]["count"] = 0
            cache[key]["count"] = c
        n_total += c
        v = cache[key]
        if v["class_name"] == "AMBIGUOUS":
            n_amb += c
            p = v["variant_" + v["aln_ref_names"][0]]
            rows.append((p["aln_seq"], p["aln_ref"], "AMBIGUOUS_" + v["aln_ref_names"][0], p["classification"], p["deletion_n"], p["insertion_n"], p["substitution_n"], c))
            continue
        for nm in v["aln_ref_names"]:
            p = v["variant_" + nm]
            items[nm].append((p, c))
            rows.append((p["aln_seq"], p["aln_ref"], nm, p["classification"], p["deletion_n"], p["insertion_n"], p["substitution_n"], c))
    per_ref = {nm: AG.aggregate(items[nm], len(refs[nm]["sequence"]), ignore_substitutions=a["ignore_substitutions"], ignore_insertions=a["ignore_insertions"],
                                ignore_deletions=a["ignore_deletions"]) for nm in names}
    rows.sort(key=lambda t: (-t[7], t[0], t[1]))
    return per_ref, n_total, n_amb, rows


def run_cases(tmp_path):
    from crispresso2_amd import refs as RF, paired_device as PD
    gold = load_golden("paired_fastq.json.gz")
    p1, p2 = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    p1.write_text(gold["fastq1"])
    p2.write_text(gold["fastq2"])
    for case in gold["cases"]:
        args = types.SimpleNamespace(**case["args"])
        refs, names = {}, []
        for r in case["refs"]:
            refs[r["name"]] = RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])
            names.append(r["name"])
        tm = {}
        res = PD.quantify_paired_fastq(str(p1), str(p2), refs, names, matrices()["EDNAFULL"], args, timings=tm)
        exp_stats = case["result"]["aln_stats"]
        for q in ("N_TOT_READS", "N_CACHED_ALN", "N_CACHED_NOTALN", "N_COMPUTED_ALN", "N_COMPUTED_NOTALN", "N_GLOBAL_SUBS", "N_SUBS_OUTSIDE_WINDOW",
                  "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS"):
            assert res.stats[q] == exp_stats[q], (case["label"], q, res.stats[q], exp_stats[q])
        per_ref, n_total, n_amb, rows = expected_from_reference_cache(case, refs, names)
        assert res.stats["N_TOTAL"] == n_total and res.stats["N_AMBIGUOUS"] == n_amb, (case["label"], res.stats, n_total, n_amb)
        for nm in names:
            L = len(refs[nm]["sequence"])
            for key, v in per_ref[nm].items():
                got = res.per_ref[nm][key]
                if isinstance(v, np.ndarray):
                    assert np.array_equal(np.asarray(got)[:L], v), (case["label"], nm, key)
                else:
                    assert got == v, (case["label"], nm, key, got, v)
        # the allele table of the run: one row per consensus read and amplicon it counts for
        got_rows = [(a, r, nm, st, dn, inn, sn, c) for a, r, nm, st, dn, inn, sn, c, pct in res.alleles()]
        assert got_rows == rows, case["label"]
        assert "second_pass" in tm and len(case["second_pass"]) > 0
    return res.ingest_route


def test_paired_files_to_count_tensors_on_the_emulator(tmp_path, monkeypatch):
    import emu_driver as E
    E.build()
    from pipeline_on_emulator import emulated_device
    from test_fastq_device_emulated import emulated_fq_kernels
    with emulated_device(), emulated_fq_kernels():
        assert run_cases(tmp_path) == "paired, keys from the host parser (small file)"
        monkeypatch.setenv("C2_FQ_INGEST", "device")                 # the two texts framed, keyed and de-duplicated by the kernels
        assert run_cases(tmp_path) == "paired, device"


@pytest.mark.gpu
def test_paired_files_to_count_tensors_on_the_device(tmp_path, monkeypatch):
    assert run_cases(tmp_path).startswith("paired, keys from the host parser")
    monkeypatch.setenv("C2_FQ_INGEST", "device")
    assert run_cases(tmp_path) == "paired, device"
