#!/usr/bin/env python3
"""Rates of the other BASELINE.json workload shapes on one GPU (config 3 is bench.py's default):
  config 2: 150 bp reads vs one 150 bp amplicon
  config 4: every read against 3 candidate amplicons (wild type, HDR, prime edit): 3 alignments per read
  config 5: 96 amplicons, every read tagged with its amplicon id (pooled): interleaved, and sorted by amplicon
Prints ms for the launch chain and for the count kernel.  python tools/config_rates.py [--reads N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def run(ctx, al, reads, L, ref_ids=None, all_refs=False, n_refs=1, label=""):
    import torch
    from crispresso2_amd import counts as C
    dev = torch.device("cuda", 0)
    n = len(reads)
    n_tasks = n * (n_refs if all_refs else 1)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    d_rid = None if ref_ids is None else torch.from_numpy(ref_ids.astype(np.int16)).to(dev)
    o1 = torch.empty((n_tasks, stride), dtype=torch.uint8, device=dev)
    o2 = torch.empty((n_tasks, stride), dtype=torch.uint8, device=dev)
    rec = torch.empty((n_tasks, 32), dtype=torch.uint8, device=dev)
    lay = C.CountLayout(n_refs, al.max_ref_len, L)
    d_counts = torch.zeros(lay.shape(), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def align():
        al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L,
                        d_ref_ids=None if d_rid is None else d_rid.data_ptr(), all_refs=all_refs, stream=s)

    def count():
        C.accumulate_device(ctx, lay, n_tasks, o1.data_ptr(), o2.data_ptr(), stride, rec.data_ptr(), d_counts.data_ptr(), stream=s)

    out = {}
    for name, fn in (("align_ms", align), ("count_ms", count)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t0) / 3
    out["alignments"] = n_tasks
    out["alignments_per_s"] = n_tasks / ((out["align_ms"] + out["count_ms"]) / 1e3)
    out["tasks_left_after_each_banded_launch"] = ctx.tier_info()
    print(json.dumps({label: out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    a = ap.parse_args()
    from crispresso2_amd import synth, _native, CRISPResso2Align as A
    from crispresso2_amd.batch import BatchAligner
    m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
    ctx = _native.Context(0)
    n = a.reads
    # config 2
    amp, g, inc = synth.amplicon_setup(150)
    run(ctx, BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx), synth.make_reads(150, n), 150, label="config2_150bp")
    # config 4
    L = 250
    amp, g, inc = synth.amplicon_setup(L)
    refs = [amp, synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")]
    gis = []
    for r in refs:
        x = np.zeros(len(r) + 1, dtype=np.int64); x[L // 2 + 1] = 1; gis.append(x)
    run(ctx, BatchAligner(refs, gis, [inc] * 3, m, -20, -2, ctx=ctx), synth.make_reads(L, n // 3), L, all_refs=True, n_refs=3, label="config4_3refs")
    # config 5
    n_amp = 96
    setups = [synth.amplicon_setup(L, 1000 + k) for k in range(n_amp)]
    per = n // n_amp
    reads = np.concatenate([synth.make_reads(L, per, amplicon_id=1000 + k, amplicon=setups[k][0]) for k in range(n_amp)])
    rids = np.repeat(np.arange(n_amp, dtype=np.uint16), per)
    al = BatchAligner([s[0] for s in setups], [s[1] for s in setups], [s[2] for s in setups], m, -20, -2, ctx=ctx)
    run(ctx, al, reads, L, ref_ids=rids, n_refs=n_amp, label="config5_96amplicons_sorted_by_amplicon")
    perm = np.random.default_rng(0).permutation(len(reads))
    run(ctx, al, reads[perm], L, ref_ids=rids[perm], n_refs=n_amp, label="config5_96amplicons_interleaved")


if __name__ == "__main__":
    main()
