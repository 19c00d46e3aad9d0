#!/usr/bin/env python3
"""Where does c2_align_partition_kernel spend a chunk's time?  Needs the A/B build with -DC2_PART_PHASES (tools/build_variant.sh partphases
c2_api_align "-DC2_PART_PHASES"; C2_AMD_LIB=crispresso2_amd/lib/variants/lib_partphases.so): workgroup clock between the kernel's barriers, summed
over the workgroups -> share of: the look at the last 32 columns | main-diagonal reads finished | the probe | lists.
    python tools/part_phases.py [--reads N] [--leg headline|fanc_shaped|lengths_200_to_L]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--leg", default="headline")
    a = ap.parse_args()
    import torch
    from crispresso2_amd import CRISPResso2Align as A, _native
    L = 250
    if a.leg == "headline":
        wl = dict(bench.build_workload(3, L, a.reads, 0, 1), max_len=L)
    else:
        wl = bench.build_robust_workloads(L, a.reads, 0, 1)[a.leg]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    job = bench.Job(ctx, wl, wl["max_len"], A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL")), dev, 1)
    job.timed(1, 1)
    ctx.phase_profile(True)
    tm = job.timed(0, 1)
    cyc = ctx.phase_profile(False)
    names = ["tail_look", "main_diagonal", "probe", "lists"]
    tot = float(sum(cyc)) or 1.0
    print(json.dumps({"leg": a.leg, "reads": job.n, "align_chain_ms": tm["align_ms"], "clock_sum": dict(zip(names, cyc)),
                      "share": {k: round(v / tot, 4) for k, v in zip(names, cyc)}, "classes": ctx.partition_info()["classes"]}))


if __name__ == "__main__":
    main()
