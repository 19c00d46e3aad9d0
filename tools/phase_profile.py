#!/usr/bin/env python3
"""Where do a workgroup's cycles go?  Runs the bench workload with the kernel's optional per-phase cycle accounting
(c2_phase_profile) and prints shader cycles per alignment for: task fetch, DP fill, traceback, output+classification.
Run on the GPU box: python tools/phase_profile.py [--reads N] [--len L]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=250, dest="L")
    ap.add_argument("--band", type=int, default=-1)
    ap.add_argument("--band-wgs", type=int, default=0)
    ap.add_argument("--kernel", default="auto", help="auto | diag2 | diag1 | band | full")
    ap.add_argument("--only-gapped", action="store_true", help="keep only the reads whose last 32 columns differ from the amplicon's in more than 6 places "
                                                                  "(an indel in front of them): every task is traced -- the epilogue's share of the band tiers")
    a = ap.parse_args()
    from crispresso2_amd import synth, _native, CRISPResso2Align as A
    from crispresso2_amd.batch import BatchAligner
    import torch
    amp, g, inc = synth.amplicon_setup(a.L)
    reads = synth.make_reads(a.L, a.reads)
    if a.only_gapped:
        ampb = np.frombuffer(amp.encode(), dtype=np.uint8)
        keep = (reads[:, -32:] != ampb[None, -32:]).sum(axis=1) > 6
        reads = np.ascontiguousarray(reads[keep])
        a.reads = len(reads)
    ctx = _native.Context(0)
    ctx.set_band(a.band, a.band_wgs)
    ctx.set_kernel_mode(a.kernel)
    al = BatchAligner([amp], [g], [inc], A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL")), -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    n, L = a.reads, a.L
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    o1 = torch.empty((n, stride), dtype=torch.uint8, device=dev)
    o2 = torch.empty((n, stride), dtype=torch.uint8, device=dev)
    rec = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L, stream=s)
        torch.cuda.synchronize()

    run()
    ctx.phase_profile(True)
    run()
    cyc = ctx.phase_profile(False)
    names = ["fetch", "dp_fill", "traceback", "output_classify"]
    tot = sum(cyc)
    print(json.dumps({"reads": n, "len": L, "cycles_per_alignment": {k: v / n for k, v in zip(names, cyc)},
                      "fraction": {k: v / tot for k, v in zip(names, cyc)}, "launch": ctx.launch_info(L), "band": ctx.band_info(L)}))


if __name__ == "__main__":
    main()
