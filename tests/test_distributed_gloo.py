"""The N > 1 path on CPU: two gloo ranks each count their shard of the alignments (kernel run by the wave emulator, a test
harness), all-reduce the count tensors, and must end up with the single-process tensor."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import emu_driver as E
    from helpers import load_golden, matrices
    from crispresso2_amd import distributed as D
    D.init("gloo")
    vecs = [v for v in load_golden("realistic.json") if len(v["seqi"]) == 250]
    amp, g, inc = vecs[0]["seqi"], vecs[0]["gap_incentive"], vecs[0]["include"]
    reads = [v["seqj"] for v in vecs]
    lo, hi = D.my_shard(len(reads))
    st = {}
    res, rec = E.align_batch(reads[lo:hi], [amp], [g], [inc], matrices()["EDNAFULL"], -20, -2, stats=st)
    o1, o2 = st["raw"]
    counts, lay = E.count_vectors(o1, o2, rec, [amp], [inc], 250)
    t = torch.from_numpy(counts)
    D.reduce_counts(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), t.numpy())
        res_all, rec_all = E.align_batch(reads, [amp], [g], [inc], matrices()["EDNAFULL"], -20, -2, stats=st)
        o1, o2 = st["raw"]
        full, _ = E.count_vectors(o1, o2, rec_all, [amp], [inc], 250)
        np.save(os.path.join(out_dir, "single.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_count_reduce_equals_single_process(tmp_path):
    sys.path.insert(0, HERE)
    import emu_driver as E
    E.build()                                  # compile once, before the ranks start
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "reduced.npy")
    b = np.load(tmp_path / "single.npy")
    assert a.sum() > 0 and np.array_equal(a, b)


def test_shard_boundaries_match_reference_semantics():
    from crispresso2_amd.distributed import shard_boundaries
    # reference tests/unit_tests/test_CRISPRessoCORE.py:866-980 cases for get_variant_cache_equal_boundaries
    assert shard_boundaries(100, 4) == [0, 25, 50, 75, 100]
    assert shard_boundaries(101, 4) == [0, 25, 50, 75, 101]
    assert shard_boundaries(3, 3) == [0, 1, 2, 3]
    with pytest.raises(Exception):
        shard_boundaries(2, 3)


def _pipeline_worker(rank, world, port, out_dir):
    """pipeline.quantify_unique on this rank's contiguous shard of the unique reads with reduce_across_ranks=True (device calls on
    the wave emulator): count tensor, first-amplicon view and the Scaffold-incorporated tensor are all-reduced over gloo."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    from helpers import matrices
    from pipeline_on_emulator import emulated_device
    from test_whole_run_tables import _params_golden, _pipeline_args
    from crispresso2_amd import distributed as D, pipeline, _native
    D.init("gloo")
    g, refs, names = _params_golden("pe_scaffold_run.json.gz")
    fq = os.path.join(out_dir, "in%d.fastq" % rank)
    with open(fq, "w") as fh:
        fh.write(g["fastq"])
    arena, offsets, counts, n_reads = _native.fastq_unique(fq)
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    with emulated_device():
        res = pipeline.quantify_unique(arena, offsets, counts, refs, names[:2], matrices()["EDNAFULL"], _pipeline_args(a),
                                       shard=D.my_shard(len(counts)) if world > 1 else None,
                                       pe_scaffold_dna_info=tuple(g["pe_scaffold_dna_info"]))
    if rank == 0:
        with open(os.path.join(out_dir, "world%d.pkl" % world), "wb") as fh:
            pickle.dump({"per_ref": res.per_ref, "view": res.first_ref_view}, fh)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_pipeline_reduce_of_counts_view_and_scaffold_tensors(tmp_path):
    """Every tensor the count route exchanges (per-amplicon counts, first-amplicon view, Scaffold-incorporated) summed over two
    gloo ranks equals the single-process result, for the prime-editing scaffold run."""
    import pickle
    sys.path.insert(0, HERE)
    import emu_driver as E
    E.build()
    _pipeline_worker(0, 1, 0, str(tmp_path))
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = pickle.load(open(tmp_path / "world1.pkl", "rb"))
    two = pickle.load(open(tmp_path / "world2.pkl", "rb"))

    def same(a, b):
        assert set(a) == set(b)
        for k, v in a.items():
            if isinstance(v, dict) and v and isinstance(next(iter(v.values())), np.ndarray):
                same(v, b[k])
            elif isinstance(v, np.ndarray):
                assert np.array_equal(v, b[k]), k
            else:
                assert v == b[k], k
    for nm in ("Reference", "Prime-edited", "Scaffold-incorporated"):
        same(one["per_ref"][nm], two["per_ref"][nm])
        same(one["view"][nm], two["view"][nm])
    assert one["per_ref"]["Scaffold-incorporated"]["counts_total"] >= 10


def _ragged_worker(rank, world, port, out_dir):
    """Shards with DIFFERENT longest reads (the unique reads sorted by length, cut unevenly) and one EMPTY shard: every rank
    must still build the same tensors and take part in every collective, and the integer statistics come back summed."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    from helpers import matrices
    from pipeline_on_emulator import emulated_device
    from test_whole_run_tables import _params_golden, _pipeline_args
    from crispresso2_amd import distributed as D, pipeline, _native
    D.init("gloo")
    g, refs, names = _params_golden("pe_scaffold_run.json.gz")
    fq = os.path.join(out_dir, "in%d.fastq" % rank)
    with open(fq, "w") as fh:
        fh.write(g["fastq"])
    arena, offsets, counts, n_reads = _native.fastq_unique(fq)
    lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
    order = np.argsort(lens, kind="stable")
    n = len(order)
    assert lens[order[0]] < lens[order[-1]]
    cuts = [0, n] if world == 1 else [0, n // 3, n, n]            # rank 0: the short reads, rank 1: the rest, rank 2: nothing
    mine = order[cuts[rank]:cuts[rank + 1]]
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    with emulated_device():
        res = pipeline.quantify_unique(arena, offsets, counts, refs, names[:2], matrices()["EDNAFULL"], _pipeline_args(a),
                                       shard=mine if world > 1 else None, pe_scaffold_dna_info=tuple(g["pe_scaffold_dna_info"]))
    with open(os.path.join(out_dir, "ragged_world%d_rank%d.pkl" % (world, rank)), "wb") as fh:
        pickle.dump({"per_ref": res.per_ref, "view": res.first_ref_view, "stats": res.stats, "shape": tuple(res.layout.shape())}, fh)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_ragged_and_empty_shards_reduce_to_the_single_process_result(tmp_path):
    import pickle
    sys.path.insert(0, HERE)
    import emu_driver as E
    E.build()
    _ragged_worker(0, 1, 0, str(tmp_path))
    mp.spawn(_ragged_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    one = pickle.load(open(tmp_path / "ragged_world1_rank0.pkl", "rb"))
    for rank in range(3):
        got = pickle.load(open(tmp_path / ("ragged_world3_rank%d.pkl" % rank), "rb"))
        assert got["shape"] == one["shape"]                        # the layout was agreed on, whatever the shard held
        assert got["stats"] == one["stats"], rank                  # ... and the statistics are the run's, not the shard's
        assert set(got["per_ref"]) == set(one["per_ref"]) and set(got["view"]) == set(one["view"])
        for nm in one["per_ref"]:                                  # every rank -- the one with the empty shard too -- holds the run's result
            for key, v in one["per_ref"][nm].items():
                w = got["per_ref"][nm][key]
                assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, (rank, nm, key)
            for key, v in one["view"][nm].items():
                assert np.array_equal(v, got["view"][nm][key]), (rank, nm, key)


# ---- a read and its reverse complement in DIFFERENT shards: the merge of CRISPRessoCORE.py:3970-3975 must still be the whole run's ----
def _split_partner_fastq():
    """The params run's reads plus reverse-complemented copies placed so that partners are far apart in first-seen order: the
    reverse complements of early unique reads at the END of the file (the partner that keeps the copies is in the first shard,
    the one that gives them away in the last), those of late unique reads at the START (the other way round), with different
    multiplicities."""
    from crispresso2_amd import refs as RF
    from test_whole_run_tables import _params_golden
    g, refs, names = _params_golden()
    lines = g["fastq_after_quality_filter"].split("\n")
    recs = [lines[k:k + 4] for k in range(0, len(lines) - 1, 4)]
    uniq = list(dict.fromkeys(r[1] for r in recs))
    rc = lambda r, tag: ["@%s_%s" % (r[0][1:].split()[0], tag), RF.reverse_complement(r[1]), "+", r[3][::-1]]
    first_of = {}
    for r in recs:
        first_of.setdefault(r[1], r)
    head = [rc(first_of[u], "h%d" % q) for q, u in enumerate(uniq[-30:]) for _ in range(q % 3 + 1)]
    tail = [rc(first_of[u], "t%d" % q) for q, u in enumerate(uniq[:40]) for _ in range(q % 2 + 1)]
    text = "".join("%s\n%s\n%s\n%s\n" % tuple(r) for r in head + recs + tail)
    return g, refs, names, text


def _split_partner_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    from helpers import matrices
    from pipeline_on_emulator import emulated_device
    from test_whole_run_tables import _pipeline_args
    from crispresso2_amd import distributed as D, pipeline
    D.init("gloo")
    g, refs, names, text = _split_partner_fastq()
    fq = os.path.join(out_dir, "split_w%d_r%d.fastq" % (world, rank))
    with open(fq, "w") as fh:
        fh.write(text)
    kind = os.environ.get("C2_TEST_INPUT_KIND", "plain")
    if kind == "bgzf":                                                  # BGZF: a rank inflates only the members that cover its byte range
        from crispresso2_amd import synth
        synth.write_bgzf(fq, fq + ".bgzf.gz", workers=1, slice_bytes=65280 * 2)
        fq = fq + ".bgzf.gz"
    elif kind == "gz":                                                  # one gzip member: every rank's host inflates it, each uploads its slice
        import gzip
        with gzip.open(fq + ".gz", "wb") as fh:
            fh.write(text.encode())
        fq = fq + ".gz"
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    import contextlib
    device_ingest = bool(os.environ.get("C2_TEST_DEVICE_INGEST"))
    if device_ingest:
        from test_fastq_device_emulated import emulated_fq_kernels
        from crispresso2_amd import fastq_device
        os.environ["C2_FQ_INGEST"] = "device"
        fastq_device.SHARD_OVERLAP = int(os.environ.get("C2_TEST_SHARD_OVERLAP", "2048"))     # (a few lines: ranges and overlaps of a small file)
    with emulated_device(), (emulated_fq_kernels() if device_ingest else contextlib.nullcontext()):
        res = pipeline.quantify_fastq(fq, refs, names, matrices()["EDNAFULL"], _pipeline_args(a), shard_across_ranks=True)
        rows = res.alleles(gather=True)
    want_route = os.environ.get("C2_TEST_EXPECT_ROUTE", "device, sharded by byte range")
    assert (getattr(res, "ingest_route", None) == want_route) == (device_ingest and world > 1), getattr(res, "ingest_route", None)
    with open(os.path.join(out_dir, "split_world%d_rank%d.pkl" % (world, rank)), "wb") as fh:
        pickle.dump({"per_ref": res.per_ref, "view": res.first_ref_view, "stats": res.stats, "alleles": rows, "shard": getattr(res, "shard_ingest", None)}, fh)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_reverse_complement_partners_in_different_shards_merge_as_in_one_process(tmp_path):
    """VERDICT r02 item 1b.  2 and 3 gloo ranks, pipeline.quantify_fastq(shard_across_ranks=True): count tensors, first-amplicon
    view, statistics and the gathered allele rows equal the single-process run on a file whose reverse-complement partners sit
    in different shards (a rank-local merge leaves both partners' rows in the allele table)."""
    import pickle
    sys.path.insert(0, HERE)
    import emu_driver as E
    from crispresso2_amd import _native, distributed as D
    E.build()
    g, refs, names, text = _split_partner_fastq()
    fq = tmp_path / "probe.fastq"
    fq.write_text(text)
    arena, offsets, counts, n_reads = _native.fastq_unique(str(fq))
    partner = _native.rc_partners(np.ascontiguousarray(arena), offsets)
    for world in (2, 3):
        b = np.array(D.shard_boundaries(len(counts), world))
        shard_of = np.searchsorted(b, np.arange(len(counts)), side="right") - 1
        has = partner >= 0
        straddle = int((shard_of[has] != shard_of[partner[has]]).sum())
        assert straddle >= 60, (world, straddle)                    # the file does what it was built for
    _split_partner_worker(0, 1, 0, str(tmp_path))
    one = pickle.load(open(tmp_path / "split_world1_rank0.pkl", "rb"))
    assert len(one["alleles"]) > 50
    for world in (2, 3):
        mp.spawn(_split_partner_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
        for rank in range(world):
            got = pickle.load(open(tmp_path / ("split_world%d_rank%d.pkl" % (world, rank)), "rb"))
            assert got["stats"] == one["stats"], (world, rank)
            assert got["alleles"] == one["alleles"], (world, rank)
            for nm in one["per_ref"]:
                for key, v in one["per_ref"][nm].items():
                    w = got["per_ref"][nm][key]
                    assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, (world, rank, nm, key)
                for key, v in one["view"][nm].items():
                    assert np.array_equal(v, got["view"][nm][key]), (world, rank, nm, key)


def test_sharded_run_with_the_text_framed_on_every_ranks_device(tmp_path, monkeypatch):
    """VERDICT r03 item 2.  The same file and the same comparison with the SHARDED device ingest: rank r uploads, frames and de-duplicates
    byte range r of the text (+ a 2 KB overlap) with the c2_fq_* kernels (emulated), the ranks all-gather their unique reads and reconcile
    them into the run's first-seen list, each aligns its range of it -- duplicates and reverse-complement partners straddle the byte
    ranges (the file's head and tail are reverse complements of reads in its middle).  Tensors, view, statistics and gathered allele rows
    equal the single process (host parser); every rank touched ~1/world of the text."""
    import pickle
    sys.path.insert(0, HERE)
    import emu_driver as E
    E.build()
    _split_partner_worker(0, 1, 0, str(tmp_path))
    one = pickle.load(open(tmp_path / "split_world1_rank0.pkl", "rb"))
    monkeypatch.setenv("C2_TEST_DEVICE_INGEST", "1")
    for world in (2, 3):
        mp.spawn(_split_partner_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
        shards = []
        for rank in range(world):
            got = pickle.load(open(tmp_path / ("split_world%d_rank%d.pkl" % (world, rank)), "rb"))
            shards.append(got["shard"])
            assert got["stats"] == one["stats"], (world, rank)
            assert got["alleles"] == one["alleles"], (world, rank)
            for nm in one["per_ref"]:
                for key, v in one["per_ref"][nm].items():
                    w = got["per_ref"][nm][key]
                    assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, (world, rank, nm, key)
                for key, v in one["view"][nm].items():
                    assert np.array_equal(v, got["view"][nm][key]), (world, rank, nm, key)
        size = shards[0]["text_bytes"]
        assert all(sh["text_bytes"] == size for sh in shards) and size > 100_000
        for sh in shards:                                             # a rank's upload: its range, the overlap behind it, 16 bytes in front
            assert sh["shard_bytes"] <= -(-size // world // 16384) * 16384 + 2048 + 16, (world, sh)
        assert sum(sh["shard_records"] for sh in shards) == one["stats"]["N_READS_INPUT"]      # every record framed by exactly one rank
        assert sum(sh["shard_unique"] for sh in shards) == shards[0]["gathered_unique"] >= len(one["alleles"])


@pytest.mark.parametrize("kind", ["bgzf", "gz"])
def test_sharded_device_ingest_of_compressed_input(tmp_path, monkeypatch, kind):
    """the sharded ingest over compressed input, 2 ranks: BGZF (a rank inflates only the members that cover its byte range of the TEXT) and a
    single gzip member (the host inflates all of it on every rank, each uploads and frames its slice) -- same result as the single process"""
    import pickle
    sys.path.insert(0, HERE)
    import emu_driver as E
    E.build()
    _split_partner_worker(0, 1, 0, str(tmp_path))
    one = pickle.load(open(tmp_path / "split_world1_rank0.pkl", "rb"))
    monkeypatch.setenv("C2_TEST_DEVICE_INGEST", "1")
    monkeypatch.setenv("C2_TEST_INPUT_KIND", kind)
    mp.spawn(_split_partner_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = pickle.load(open(tmp_path / ("split_world2_rank%d.pkl" % rank), "rb"))
        assert got["stats"] == one["stats"] and got["alleles"] == one["alleles"], (kind, rank)
        assert got["shard"]["shard_bytes"] < 0.75 * got["shard"]["text_bytes"]
        for nm in one["per_ref"]:
            for key, v in one["per_ref"][nm].items():
                w = got["per_ref"][nm][key]
                assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, (kind, rank, nm, key)
