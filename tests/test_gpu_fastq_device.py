"""The device ingest on the MI355X (c2_fq_count / lines / dedup / gather kernels through the C-ABI, driven by crispresso2_amd/fastq_device.py)
against the host parser (c2_fastq_stream, itself checked against the oracle's readline loop in the CPU suite): same unique reads in the
same order with the same multiplicities and line statistics -- many chunks, many workgroups racing for the same table slots -- and the
same run (statistics, count tensors, allele rows) from pipeline.quantify_fastq whichever route framed the file."""
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_unique(path):
    from crispresso2_amd import _native
    st = {}
    with _native.FastqStream(str(path)) as fq:
        while not fq.done:
            fq.next()
        n = fq.n_unique
        off = fq.offsets_slice(0, n).astype(np.int64)
        arena = np.array(fq.arena[:int(off[-1])], copy=True)
        counts = fq.counts().astype(np.int64)
        fq.line_stats(st)
        n_reads = fq.n_reads
    return arena, off, counts, st, n_reads


def _oracle_agrees(path, out, d_reads_host):
    """third party: the restatement of the reference's readline loop (oracle/fastq.py, CRISPRessoCORE.py:1820-1849) -- same unique reads
    (empty key aside) in first-seen order with the same multiplicities, same number of records"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import fastq as OF
    cache, num_reads = OF.read_fastq_unique(str(path))
    assert out["n_reads"] == num_reads
    keys = [q for q in cache if q != ""]
    assert out["n_unique"] == len(keys) and out["n_empty_records"] == cache.get("", 0)
    assert out["counts"].astype(np.int64).tolist() == [cache[q] for q in keys]
    off = out["offsets"].astype(np.int64)
    blob = d_reads_host.tobytes()
    assert blob[:int(off[-1])] == "".join(keys).encode("latin-1")
    assert np.array_equal(np.diff(off), np.array([len(q) for q in keys], dtype=np.int64))


def _write(path, n, L, rng, n_pool, tail=""):
    pool = ["".join(rng.choice("ACGT") for _ in range(rng.randint(L // 2, L))) for _ in range(n_pool)]
    pool[7] = "  " + pool[7] + "\t"                                   # (whitespace the reference strips; equals no other read)
    pool[9] = pool[8]
    with open(path, "w") as fh:
        for i in range(n):
            s = pool[min(int(rng.expovariate(1.0 / (n_pool / 6.0))), n_pool - 1)]
            fh.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
        fh.write(tail)


@pytest.mark.parametrize("tail", ["", "@cut\nACGTACGT", "@id_only\n", "\n\n@x\nAC\n+\n"])
def test_device_ingest_equals_the_host_parser(tmp_path, monkeypatch, tail):
    import torch
    from crispresso2_amd import fastq_device as FD, _native
    rng = random.Random(len(tail) + 3)
    p = tmp_path / "t.fastq"
    _write(p, 300_000, 120, rng, 40_000, tail)
    monkeypatch.setattr(FD, "CHUNK_BYTES", 1 << 20)                    # ~80 chunks
    ctx = _native.default_context()
    out = FD.ingest_file(str(p), ctx, torch.device("cuda", 0))
    arena, off, counts, st, n_reads = _host_unique(p)
    lens = off[1:] - off[:-1]
    keep = lens > 0
    assert out["n_reads"] == n_reads
    assert int(float(out["nonempty_lines"]) / 4.0) == st["N_READS_AFTER_PREPROCESSING"]
    assert out["n_empty_records"] == int(counts[~keep].sum())
    assert out["n_unique"] == int(keep.sum())
    assert np.array_equal(out["counts"], counts[keep])
    assert np.array_equal(np.diff(out["offsets"].astype(np.int64)), lens[keep])
    want = np.concatenate([arena[off[i]:off[i + 1]] for i in np.nonzero(keep)[0]]) if keep.any() else np.zeros(0, np.uint8)
    got_reads = out["d_reads"][:int(out["offsets"][-1])].cpu().numpy()
    assert np.array_equal(got_reads, want)
    assert int(counts.sum()) == n_reads
    _oracle_agrees(p, out, got_reads)
    assert np.array_equal(out["rc_partner"], _native.rc_partners(want, out["offsets"]))


def test_reverse_complement_partners_from_the_table(tmp_path, monkeypatch):
    import torch
    from crispresso2_amd import fastq_device as FD, _native, refs as RF
    rng = random.Random(4)
    base = ["".join(rng.choice("ACGT") for _ in range(rng.randint(60, 150))) for _ in range(30_000)]
    seqs = base + [RF.reverse_complement(s) for s in base[:12_000]] + ["ACGT", "AATT", "acgt", "ACGTNN", "NNACGT", "AC-GT_", "_AC-GT", "ACXGT", "ggatcc",
                                                                      "GGATCC", base[5].lower()]
    rng.shuffle(seqs)
    p = tmp_path / "rc.fastq"
    p.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(seqs + seqs[:9000])))
    monkeypatch.setattr(FD, "CHUNK_BYTES", 1 << 20)
    out = FD.ingest_file(str(p), _native.default_context(), torch.device("cuda", 0))
    arena = out["d_reads"][:int(out["offsets"][-1])].cpu().numpy()
    want = _native.rc_partners(arena, out["offsets"])
    assert np.array_equal(out["rc_partner"], want)
    assert (want >= 0).sum() >= 24_000
    from helpers import rc_partner_witness                             # ... and both against the definition written out in the test helpers
    off = np.asarray(out["offsets"], dtype=np.int64)
    reads = [arena[off[i]:off[i + 1]].tobytes().decode() for i in range(len(off) - 1)]
    assert np.array_equal(np.asarray(out["rc_partner"], dtype=np.int64), rc_partner_witness(reads))


def test_whole_run_is_the_same_on_either_route(tmp_path, monkeypatch):
    from crispresso2_amd import pipeline, synth, refs as RF, fastq_device as FD
    from helpers import matrices
    L = 150
    amp, g_, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, 60_000)
    seqs = [r.tobytes().decode() for r in reads]
    seqs = seqs + [RF.reverse_complement(s_) for s_ in seqs[:5000]] + seqs[:20_000] + ["ACGT" * 30, "TTTTGGGGCCCC" * 9] * 3
    fq = tmp_path / "run.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (k, s_, "I" * len(s_)) for k, s_ in enumerate(seqs)) + "@id_only\n")
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                           ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    import gzip
    gz = tmp_path / "run.fastq.gz"
    with gzip.open(gz, "wb", compresslevel=1) as fh:
        fh.write(fq.read_bytes())
    bz = tmp_path / "run.bgzf.fastq.gz"
    synth.write_bgzf(str(fq), str(bz), workers=4, level=1)
    monkeypatch.setattr(FD, "CHUNK_BYTES", 2 << 20)
    monkeypatch.setattr(pipeline, "STREAM_MIN_BATCH", 5000)           # (batches of alignments under the upload)
    results = []
    for route, path, want in (("host", fq, None), ("device", fq, "device"), ("device", gz, "device, text from host memory"), ("host", gz, None),
                              ("device", bz, "device, members inflated into the upload buffers"), ("host", bz, None)):
        monkeypatch.setenv("C2_FQ_INGEST", route)
        tm = {}
        res = pipeline.quantify_fastq(str(path), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], args, timings=tm)
        assert getattr(res, "ingest_route", None) == want
        assert want is None or tm["stream_batches"] >= 3, tm
        results.append((res.stats, res.per_ref["Reference"], res.alleles()))
    st0, pr0, al0 = results[0]
    assert st0["N_TOT_READS"] == len(seqs)
    for st1, pr1, al1 in results[1:]:
        assert st0 == st1 and al0 == al1
        for kk, vv in pr0.items():
            assert np.array_equal(vv, pr1[kk]) if isinstance(vv, np.ndarray) else vv == pr1[kk], kk


def test_filtered_input_on_either_route(tmp_path, monkeypatch):
    """--min_average_read_quality: the host filters into memory, the device frames the filtered text; = the host parser on the same text"""
    from crispresso2_amd import pipeline, synth, refs as RF, fastq_device as FD
    from helpers import matrices
    L = 150
    amp, g_, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, 40_000)
    rng = random.Random(8)
    fq = tmp_path / "q.fastq"
    with open(fq, "w") as fh:
        for k, r in enumerate(reads):
            s_ = r.tobytes().decode()
            fh.write("@r%d\n%s\n+\n%s\n" % (k, s_, ("I" if rng.random() < 0.8 else "#") * len(s_)))
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                           ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False, min_average_read_quality=30,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    monkeypatch.setattr(FD, "CHUNK_BYTES", 2 << 20)
    out = []
    for route in ("host", "device"):
        monkeypatch.setenv("C2_FQ_INGEST", route)
        res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], args)
        assert (getattr(res, "ingest_route", None) == "device, text from host memory") == (route == "device")
        out.append(res)
    assert out[0].stats == out[1].stats and out[0].stats["N_READS_INPUT"] == 40_000 and 25_000 < out[0].stats["N_READS_AFTER_PREPROCESSING"] < 38_000
    assert out[0].alleles() == out[1].alleles()


@pytest.mark.parametrize("seed", range(4))
def test_fuzzed_text_against_the_host_parser(tmp_path, monkeypatch, seed):
    """text without any FASTQ structure (random lines, blank lines, white space of every kind, every tail), ~2 MB in 64 KB chunks:
    the kernels = the host parser (= the reference's readline loop, CPU suite) in reads, order, multiplicities, record and line counts"""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_fastq_device_emulated import fuzz_text
    from crispresso2_amd import fastq_device as FD, _native
    rng = random.Random(500 + seed)
    p = tmp_path / "fz.fastq"
    p.write_text(fuzz_text(rng, 60_000, max_len=rng.choice([12, 60, 200]), blank=rng.choice([0.0, 0.1, 0.5])))
    monkeypatch.setattr(FD, "CHUNK_BYTES", 1 << 16)
    out = FD.ingest_file(str(p), _native.default_context(), torch.device("cuda", 0))
    arena, off, counts, st, n_reads = _host_unique(p)
    lens = off[1:] - off[:-1]
    keep = lens > 0
    assert out["n_reads"] == n_reads and out["n_unique"] == int(keep.sum())
    assert int(float(out["nonempty_lines"]) / 4.0) == st["N_READS_AFTER_PREPROCESSING"]
    assert np.array_equal(out["counts"], counts[keep]) and np.array_equal(np.diff(out["offsets"].astype(np.int64)), lens[keep])
    want = np.concatenate([arena[off[i]:off[i + 1]] for i in np.nonzero(keep)[0]]) if keep.any() else np.zeros(0, np.uint8)
    got_reads = out["d_reads"][:int(out["offsets"][-1])].cpu().numpy()
    assert np.array_equal(got_reads, want)
    _oracle_agrees(p, out, got_reads)


def _sharded_worker(rank, world, port, path, out_dir, route):
    import pickle
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), C2_FQ_INGEST=route)
    import torch.distributed as dist
    from helpers import matrices
    from crispresso2_amd import distributed as D, pipeline, synth, refs as RF
    if world > 1:
        D.init("gloo")                                             # (both ranks share the box's one GPU: RCCL needs a GPU per rank)
    L = 150
    amp, g_, inc = synth.amplicon_setup(L)
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                           ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    res = pipeline.quantify_fastq(path, {"Reference": ref}, ["Reference"], matrices()["EDNAFULL"], args, shard_across_ranks=world > 1)
    with open(os.path.join(out_dir, "w%d_r%d_%s.pkl" % (world, rank, route)), "wb") as fh:
        pickle.dump({"per_ref": res.per_ref, "stats": res.stats, "route": getattr(res, "ingest_route", None)}, fh)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_each_frame_the_text_on_the_device(tmp_path):
    """pipeline.quantify_fastq(shard_across_ranks=True) with two gloo ranks on the box's one GPU: each rank frames and de-duplicates the
    whole file with the c2_fq_* kernels, aligns its half of the unique reads, the tensors are all-reduced -- = one process, either route
    (reads with reverse-complement partners in the other rank's half included)"""
    import pickle
    import socket
    import torch.multiprocessing as mp
    from crispresso2_amd import synth, refs as RF
    L = 150
    reads = synth.make_reads(L, 50_000)
    seqs = [r.tobytes().decode() for r in reads]
    seqs = [RF.reverse_complement(s_) for s_ in seqs[-4000:]] + seqs + seqs[:15_000]
    fq = tmp_path / "sh.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (k, s_, "I" * len(s_)) for k, s_ in enumerate(seqs)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_sharded_worker, args=(1, 0, str(fq), str(tmp_path), "host"), nprocs=1, join=True)
    one = pickle.load(open(tmp_path / "w1_r0_host.pkl", "rb"))
    mp.spawn(_sharded_worker, args=(2, port, str(fq), str(tmp_path), "device"), nprocs=2, join=True)
    for rank in range(2):
        got = pickle.load(open(tmp_path / ("w2_r%d_device.pkl" % rank), "rb"))
        assert got["route"] == "device, sharded by byte range"
        assert got["stats"] == one["stats"], rank
        for key, v in one["per_ref"]["Reference"].items():
            w = got["per_ref"]["Reference"][key]
            assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, (rank, key)
