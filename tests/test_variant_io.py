"""Wire formats behind the path, CPU side: the JSON codec of variant dicts, variants_<k>.tsv files and the
--fastq_output annotation, against text the REFERENCE wrote (tests/golden/variant_io.json.gz, recorded by
tests/golden/make_golden.py --variant-io from process_fastq / process_fastq_write_out with -p 1 and -p 2)."""
import gzip
import json
import os
import socket
import sys
import types

import numpy as np
import pytest

from helpers import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def gold():
    return load_golden("variant_io.json.gz")


def _write(path, text):
    with open(path, "w") as fh:
        fh.write(text)
    return str(path)


def test_json_codec_round_trips_the_reference_lines_byte_for_byte(gold):
    from crispresso2_amd import variant_io as IO, CRISPRessoCOREResources as R
    n = n_slots = 0
    for case in gold["cases"]:
        for text in case["tsv"]:
            for line in text.splitlines(keepends=True):
                seq, js = line.rstrip("\n").split("\t")
                v = json.loads(js, cls=IO.CRISPRessoJSONDecoder)
                assert IO.variant_line(seq, v) == line
                for k, p in v.items():
                    if k.startswith("variant_"):
                        assert isinstance(p, R.ResultsSlotsDict)
                        n_slots += 1
                n += 1
    assert n > 600 and n_slots > 500


def test_codec_tagged_types():
    import argparse
    import datetime
    from crispresso2_amd import variant_io as IO
    obj = {"a": np.arange(3), "i": np.int64(4), "f": np.float64(0.5), "s": {3}, "r": range(2, 9, 3), "r2": range(1, 4),
           "t": datetime.datetime(2024, 6, 1, 12, 30, 5), "d": datetime.timedelta(days=1, seconds=2, microseconds=3),
           "ns": argparse.Namespace(x=1, y="z")}
    text = json.dumps(obj, cls=IO.CRISPRessoJSONEncoder)
    assert '{"_type": "np.ndarray", "value": [0, 1, 2]}' in text and '"i": 4' in text and '"_type": "range", "value": "range(2, 9, 3)"' in text
    back = json.loads(text, cls=IO.CRISPRessoJSONDecoder)
    assert np.array_equal(back["a"], obj["a"]) and back["i"] == 4 and back["f"] == 0.5 and back["s"] == {3}
    assert back["r"] == obj["r"] and back["r2"] == obj["r2"] and back["t"] == obj["t"] and back["d"] == obj["d"] and back["ns"] == obj["ns"]
    with pytest.raises(TypeError):
        json.dumps({"x": object()}, cls=IO.CRISPRessoJSONEncoder)


def test_merge_of_variant_files_reproduces_the_reference_bookkeeping(gold, tmp_path):
    from crispresso2_amd import variant_io as IO
    from crispresso2_amd.variants import read_fastq_unique
    fq = _write(tmp_path / "in.fastq", gold["fastq"])
    for case in gold["cases"]:
        paths = [_write(tmp_path / ("variants_%d.tsv" % k), t) for k, t in enumerate(case["tsv"])]
        cache = read_fastq_unique(fq)
        args = types.SimpleNamespace(**case["args"])
        st, not_aligned = IO.merge_variant_files(paths, cache, args)
        assert st == case["multi"]["aln_stats"]
        assert list(not_aligned.keys()) == case["multi"]["not_aligned"]
        assert list(cache.keys()) == case["multi"]["aligned"]
        assert [cache[k]["count"] for k in cache] == case["multi"]["counts"]
    # a file that lacks a read -> error, like the reference's count check
    cache = read_fastq_unique(fq)
    short = _write(tmp_path / "short.tsv", "".join(gold["cases"][0]["tsv"][0].splitlines(keepends=True)[:-1]))
    with pytest.raises(ValueError):
        IO.merge_variant_files([short, paths[1]], cache, args)


def test_annotated_fastq_equals_the_reference_output(gold, tmp_path):
    from crispresso2_amd import variant_io as IO
    from crispresso2_amd.variants import read_fastq_unique
    fq = _write(tmp_path / "in.fastq", gold["fastq"])
    for case in gold["cases"]:
        paths = [_write(tmp_path / ("variants_%d.tsv" % k), t) for k, t in enumerate(case["tsv"])]
        cache = read_fastq_unique(fq)
        _, not_aligned = IO.merge_variant_files(paths, cache, types.SimpleNamespace(**case["args"]))
        out = str(tmp_path / "out.fastq.gz")
        IO.write_annotated_fastq(fq, out, cache, not_aligned)
        with gzip.open(out, "rt") as fh:
            assert fh.read() == case["annotated"]
        assert all("crispresso2_annotation" in v for v in cache.values())
    # gzip'ed input, and a read nobody computed
    with gzip.open(tmp_path / "in.fastq.gz", "wt") as fh:
        fh.write(gold["fastq"])
    IO.write_annotated_fastq(str(tmp_path / "in.fastq.gz"), out, cache, not_aligned)
    with gzip.open(out, "rt") as fh:
        assert fh.read() == gold["cases"][-1]["annotated"]
    _write(tmp_path / "other.fastq", "@x\nACGT\n+\nIIII\n")
    with pytest.raises(KeyError):
        IO.write_annotated_fastq(str(tmp_path / "other.fastq"), out, cache, not_aligned)


# ---- the N > 1 route on CPU: two gloo ranks write variants_<rank>.tsv for their slice, rank 0 merges -------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from crispresso2_amd import distributed as D, variant_io as IO
    import crispresso2_amd.variants as V
    D.init("gloo")
    gold = load_golden("variant_io.json.gz")
    case = gold["cases"][1]
    known = {}
    for text in case["tsv"]:
        for line in text.splitlines():
            seq, js = line.split("\t")
            known[seq] = json.loads(js, cls=IO.CRISPRessoJSONDecoder)

    def recorded_variants(args, seqs, refs, ref_names, aln_matrix, pe, ctx=None):      # stands in for the GPU on this CPU box
        return [known[s] for s in seqs]

    out = V.process_fastq_sharded(os.path.join(tmp, "in.fastq"), types.SimpleNamespace(**case["args"]), None, None, None, tmp,
                                  get_variants=recorded_variants)
    if rank == 0:
        cache, not_aligned, st = out
        with open(os.path.join(tmp, "result.json"), "w") as fh:
            json.dump({"st": st, "not_aligned": list(not_aligned), "aligned": list(cache), "counts": [cache[k]["count"] for k in cache]}, fh)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_variant_files_merge_to_the_reference_result(gold, tmp_path):
    import torch.multiprocessing as mp
    _write(tmp_path / "in.fastq", gold["fastq"])
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    case = gold["cases"][1]
    for k in range(2):
        with open(tmp_path / ("variants_%d.tsv" % k)) as fh:
            assert fh.read() == case["tsv"][k]                        # same slices as the reference's two worker processes
    with open(tmp_path / "result.json") as fh:
        got = json.load(fh)
    assert got["st"] == case["multi"]["aln_stats"] and got["not_aligned"] == case["multi"]["not_aligned"]
    assert got["aligned"] == case["multi"]["aligned"] and got["counts"] == case["multi"]["counts"]


def test_sam_text_of_bam_output_equals_the_reference_file():
    """--bam_output (CRISPRessoCORE.py:2351-2515): the .sam text the reference writes before `samtools sort`, from the variant
    dicts IT computed (recorded as its own JSON lines by make_golden.py --sam): contigs on both strands, start offsets,
    unmapped reads, two assigned references per read (expand_ambiguous_alignments); every dict gets its 'sam_entry'."""
    from crispresso2_amd import variant_io as IO
    import tempfile
    gold = load_golden("sam_output.json.gz")
    n_lines = n_rev = n_unmapped = 0
    for case in gold["cases"]:
        cache, not_aln = {}, {}
        for bucket, lines in ((cache, case["variant_lines"]), (not_aln, case["not_aligned_lines"])):
            for line in lines:
                seq, js = line.split("\t")
                bucket[seq] = json.loads(js, cls=IO.CRISPRessoJSONDecoder)
        with tempfile.TemporaryDirectory() as tmp:
            fq = _write(os.path.join(tmp, "in.fastq"), gold["fastq"])
            sam = os.path.join(tmp, "out.bam.sam")
            IO.write_annotated_sam(fq, sam, case["header"], cache, not_aln, case["aln"])
            with open(sam) as fh:
                text = fh.read()
        assert text == case["sam"], case["label"]
        body = [l.split("\t") for l in text.splitlines() if not l.startswith("@")]
        n_lines += len(body)
        n_rev += sum(f[1] == "16" for f in body)
        n_unmapped += sum(f[1] == "4" for f in body)
        assert all(len(f) == 12 for f in body)
        assert all(v["sam_entry"][9] in (k, IO.sam_entry("x", k, "", v, case["aln"])[9]) for k, v in cache.items())
    assert n_lines == 4 * 302 and n_rev > 100 and n_unmapped == 4 * 20


def test_cigar_elements_and_unknown_symbols():
    from crispresso2_amd import variant_io as IO
    assert IO.cigar_elements("AC--GTNA", "ACGGGT-A") == ["2M", "2D", "2M", "1I", "1M"]
    assert IO.cigar_elements("", "") == []
    with pytest.raises(KeyError):                                   # the reference's table has no lower case / IUPAC / double gap
        IO.cigar_elements("AcG", "ACG")
    with pytest.raises(KeyError):
        IO.cigar_elements("A-G", "A-G")


@pytest.mark.gpu
def test_bam_output_sam_text_from_the_device_route(tmp_path):
    """process_single_fastq_write_bam_out with the variant dicts computed on the GPU (device alignments + device classifier):
    the reference's .sam text byte for byte, for every recorded case."""
    from helpers import matrices
    from crispresso2_amd import _native, refs as RF, variants as V
    ctx = _native.default_context()
    gold = load_golden("sam_output.json.gz")
    fq = _write(tmp_path / "in.fastq", gold["fastq"])
    for k, case in enumerate(gold["cases"]):
        args = types.SimpleNamespace(**case["args"])
        refs, names = {}, []
        for r in case["refs"]:
            refs[r["name"]] = RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])
            refs[r["name"]].update(case["aln"][r["name"]])
            names.append(r["name"])
        bam = str(tmp_path / ("out%d.bam" % k))
        cache, not_aligned, st = V.process_single_fastq_write_bam_out(fq, bam, case["header"], args, refs, names, matrices()["EDNAFULL"], ctx=ctx)
        with open(bam + ".sam") as fh:
            assert fh.read() == case["sam"], case["label"]
        assert st["N_TOT_READS"] == 302 and len(not_aligned) == len(case["not_aligned_lines"])
