#!/usr/bin/env python3
"""One of bench.py's robustness workloads (or the headline's) through the default chain, quickly: rate, tier shares, chain = full plane.
    python tools/robust_rate.py [--reads N] [--legs fanc_shaped,lengths_200_to_L,unrelated_10_percent,headline] [--steps K]
Environment knobs of the partition (C2_ROUTE_MARGIN, C2_NO_DIRECT_FULL, C2_NO_LENGTH_ORDER, ...) apply: this is the A/B tool for them."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=3_000_000)
    ap.add_argument("--legs", default="fanc_shaped,lengths_200_to_L,unrelated_10_percent")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    L = 250
    wls = bench.build_robust_workloads(L, a.reads, 0, 1)
    if "headline" in a.legs:
        wls["headline"] = dict(bench.build_workload(3, L, a.reads, 0, 1), max_len=L)
    import torch
    from crispresso2_amd import CRISPResso2Align as A, _native
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
    ctx = _native.Context(0)
    for name in a.legs.split(","):
        wl = wls[name]
        job = bench.Job(ctx, wl, wl["max_len"], m, dev, 1)
        tm = job.timed(1, a.steps)
        tiers, part = ctx.tier_info(), ctx.partition_info()
        out = {"leg": name, "reads": job.n, "reads_per_s": tm["reads_per_s"], "align_chain_ms": tm["align_ms"], "count_ms": tm["count_ms"],
               "classes": part["classes"], "score_only_finished": part["finished"][0], "finished_by_partition": part.get("finished_by_partition"), "lists_after_tiers": tiers,
               "env": {k: v for k, v in os.environ.items() if k.startswith("C2_")}}
        if not a.no_check:
            eq, tf = job.chain_equals_full_plane()
            out["chain_equals_full_plane"] = bool(eq == job.n_tasks)
        print(json.dumps(out))
        sys.stdout.flush()
        job.free()
        del job


if __name__ == "__main__":
    main()
