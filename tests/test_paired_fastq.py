"""process_paired_fastq (the reference's -p 2 route, CRISPRessoCORE.py:1296-1516) on CPU: native paired ingest, slices of
the unique pairs as variants_<k>.tsv, the parent's merge, the per-occurrence second pass for pairs whose consensus depends
on the qualities, pair keys replaced by consensus reads -- against a run of the REFERENCE
(tests/golden/paired_fastq.json.gz, made by make_golden.py --paired-fastq).  The variant dicts themselves come from the
recorded reference run here (the device computes them in tests/test_gpu_parity.py)."""
import json
import os
import types

import pytest

from helpers import load_golden


@pytest.fixture(scope="module")
def gold():
    return load_golden("paired_fastq.json.gz")


def recorded_variants(case, cache):
    """(seq1, seq2, qual1, qual2) -> a fresh copy of the dict the reference computed for that call."""
    from crispresso2_amd import variant_io as IO
    text = {}
    for t in case["tsv"]:
        for line in t.splitlines():
            key, js = line.split("\t")
            s1, s2 = key.split("+")
            q1, q2 = cache[key][1].split(" ")
            text[(s1, s2, q1, q2)] = js
    for s1, s2, q1, q2, js in case["second_pass"]:
        text[(s1, s2, q1, q2)] = js

    def get_variants(args, pairs, refs, ref_names, aln_matrix, pe, ctx=None):
        return [json.loads(text[tuple(p)], cls=IO.CRISPRessoJSONDecoder) for p in pairs]
    return get_variants


def check_result(case, cache, not_aln, st):
    from crispresso2_amd import variant_io as IO
    exp = case["result"]
    assert st == exp["aln_stats"]
    assert list(not_aln) == exp["not_aligned"]
    assert list(cache) == exp["aligned"]
    assert [cache[k]["count"] for k in cache] == exp["counts"]
    assert [json.dumps(cache[k], cls=IO.CRISPRessoJSONEncoder) for k in cache] == exp["variants"]


@pytest.mark.parametrize("through_files", [False, True])
def test_paired_route_equals_the_reference_run(gold, tmp_path, through_files):
    from crispresso2_amd import paired as P
    p1, p2 = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    p1.write_text(gold["fastq1"])
    p2.write_text(gold["fastq2"])
    for case in gold["cases"]:
        cache0, pf = P.read_paired_fastq_unique(str(p1), str(p2))
        pf.close()
        args = types.SimpleNamespace(**case["args"])
        stub = recorded_variants(case, cache0)
        if not through_files:
            cache, not_aln, st = P.process_paired_fastq(str(p1), str(p2), args, None, None, None, get_variants=stub)
        else:
            d = tmp_path / ("v_" + case["label"].replace(" ", "_").replace("+", "_"))
            d.mkdir()
            assert P.process_paired_fastq(str(p1), str(p2), args, None, None, None, variants_dir=str(d), rank=1, world=2, get_variants=stub) is None
            cache, not_aln, st = P.process_paired_fastq(str(p1), str(p2), args, None, None, None, variants_dir=str(d), rank=0, world=2,
                                                        get_variants=stub)
            for k in range(2):
                assert (d / ("variants_%d.tsv" % k)).read_text() == case["tsv"][k]
        check_result(case, cache, not_aln, st)
        assert len(case["second_pass"]) > 0 and not any("+" in k for k in cache)


def test_more_ranks_than_unique_pairs_is_the_reference_error(tmp_path):
    from crispresso2_amd import paired as P
    p1, p2 = tmp_path / "a.fastq", tmp_path / "b.fastq"
    p1.write_text("@a\nACGT\n+\nIIII\n")
    p2.write_text("@a\nACGT\n+\nIIII\n")
    with pytest.raises(Exception, match="less than the number of processes"):
        P.process_paired_fastq(str(p1), str(p2), types.SimpleNamespace(expand_ambiguous_alignments=False), None, None, None,
                               variants_dir=str(tmp_path), rank=0, world=2, get_variants=lambda *a, **k: [])
