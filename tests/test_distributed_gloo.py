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
    lo, hi = D.my_shard(len(counts)) if world > 1 else (0, len(counts))
    sub_off = (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64)
    sub_arena = arena[int(offsets[lo]):int(offsets[hi])]
    with emulated_device():
        res = pipeline.quantify_unique(sub_arena, sub_off, counts[lo:hi], refs, names[:2], matrices()["EDNAFULL"], _pipeline_args(a),
                                       reduce_across_ranks=world > 1, pe_scaffold_dna_info=tuple(g["pe_scaffold_dna_info"]))
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
    sub_off = np.zeros(len(mine) + 1, dtype=np.uint64)
    sub_off[1:] = np.cumsum(lens[mine])
    sub_arena = (np.concatenate([arena[int(offsets[i]):int(offsets[i + 1])] for i in mine]) if len(mine) else np.zeros(0, dtype=np.uint8))
    a = {k: v for k, v in g["args"].items() if k not in ("plot_window_size", "dsODN")}
    with emulated_device():
        res = pipeline.quantify_unique(sub_arena, sub_off, counts[mine], refs, names[:2], matrices()["EDNAFULL"], _pipeline_args(a),
                                       reduce_across_ranks=world > 1, pe_scaffold_dna_info=tuple(g["pe_scaffold_dna_info"]))
    with open(os.path.join(out_dir, "ragged_world%d_rank%d.pkl" % (world, rank)), "wb") as fh:
        pickle.dump({"per_ref": res.per_ref, "stats": res.stats, "shape": tuple(res.layout.shape())}, fh)
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
        if rank < 2:                                               # (the empty rank returns before the scaffold amplicon is added)
            for nm in one["per_ref"]:
                for key, v in one["per_ref"][nm].items():
                    w = got["per_ref"][nm][key]
                    assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, (rank, nm, key)
