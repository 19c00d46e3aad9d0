#!/usr/bin/env python3
"""Rate of c2_consensus_pairs_batch (get_consensus_alignment_from_pairs for a batch of read pairs): synthetic pairs -- read 1 and
read 2 of the same fragment aligned to a 250-bp amplicon on the device, Phred qualities drawn at random -- through the C ABI
(host arrays in, host arrays out).  Run on the GPU box; under `rocprofv3 --kernel-trace --stats` the kernel's own time shows.
  python tools/consensus_rate.py [--pairs N]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=500_000)
    a = ap.parse_args()
    from crispresso2_amd import synth, _native, CRISPResso2Align as A
    from crispresso2_amd.batch import BatchAligner
    L, n = 250, a.pairs
    amp, g, inc = synth.amplicon_setup(L)
    r1 = synth.make_reads(L, n, workers=1)            # (no fork: forked children hang in rocprofv3's exit handler)
    rng = np.random.default_rng(7)
    r2 = r1.copy()                                             # read 2: the same fragment with its own sequencing errors
    err = rng.random(r2.shape) < 0.004
    r2[err] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(err.sum()))]
    ctx = _native.Context(0)
    al = BatchAligner([amp], [g], [inc], A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL")), -20, -2, ctx=ctx)
    off = np.arange(n + 1, dtype=np.uint64) * L
    a1 = al.align((r1.reshape(-1), off))
    a2 = al.align((r2.reshape(-1), off))
    stride = a1.aln_read.shape[1]
    n1 = a1.records["aln_len"].astype(np.int32)
    n2 = a2.records["aln_len"].astype(np.int32)
    qstride = 256
    q1 = rng.integers(35, 74, (n, qstride), dtype=np.uint8)
    q2 = rng.integers(35, 74, (n, qstride), dtype=np.uint8)
    lq = np.full(n, L, dtype=np.int32)
    best1 = (a1.records["matches"].astype(np.int64) * n2 >= a2.records["matches"].astype(np.int64) * n1).astype(np.uint8)
    ostride = 2 * stride
    oa = np.zeros((n, ostride), dtype=np.uint8); orf = np.zeros_like(oa); oq = np.zeros_like(oa)
    info = np.zeros((n, 4), dtype=np.int32)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)

    def call():
        ctx.check(ctx.lib.c2_consensus_pairs_batch(ctx.handle, ctypes.c_uint64(n), p(a1.aln_read), p(a1.aln_ref), p(a2.aln_read), p(a2.aln_ref),
                                                   ctypes.c_uint32(stride), p(n1), p(n2), p(q1), p(q2), ctypes.c_uint32(qstride), p(lq), p(lq), p(best1),
                                                   p(oa), p(orf), p(oq), ctypes.c_uint32(ostride), p(info)), "c2_consensus_pairs_batch")
    call()
    t0 = time.perf_counter()
    call()
    dt = time.perf_counter() - t0
    # the same pairs with every array resident on the device (c2_consensus_pairs_device: what the paired route would call on the rows
    # the align kernels wrote); results must be the host call's
    import torch
    dev = torch.device("cuda", 0)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d = [T(a1.aln_read), T(a1.aln_ref), T(a2.aln_read), T(a2.aln_ref), T(n1), T(n2), T(q1), T(q2), T(lq), T(best1)]
    d_oa = torch.zeros((n, ostride), dtype=torch.uint8, device=dev); d_or = torch.zeros_like(d_oa); d_oq = torch.zeros_like(d_oa)
    d_info = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream

    def call_dev():
        ctx.check(ctx.lib.c2_consensus_pairs_device(ctx.handle, ctypes.c_uint64(n), P(d[0]), P(d[1]), P(d[2]), P(d[3]), ctypes.c_uint32(stride), P(d[4]), P(d[5]),
                                                    P(d[6]), P(d[7]), ctypes.c_uint32(qstride), P(d[8]), P(d[8]), P(d[9]), P(d_oa), P(d_or), P(d_oq),
                                                    ctypes.c_uint32(ostride), P(d_info), ctypes.c_void_p(stream)), "c2_consensus_pairs_device")
        torch.cuda.synchronize()
    call_dev()
    t0 = time.perf_counter()
    call_dev()
    dt_dev = time.perf_counter() - t0
    same = bool(np.array_equal(d_info.cpu().numpy(), info))
    if same:
        cols = np.arange(ostride)[None, :]
        for got, want, k_ in ((d_oa, oa, 0), (d_or, orf, 0), (d_oq, oq, 1)):
            g_ = got.cpu().numpy()
            same = same and bool(((g_ == want) | (cols >= info[:, k_][:, None])).all())
    bytes_per_pair = 4 * stride + 2 * qstride + 16 + 17 + 3 * int(info[:, :2].max()) + 16
    print(json.dumps({"pairs": n, "seconds": dt, "pairs_per_s": n / dt, "bytes_moved_per_pair": int(bytes_per_pair),
                      "index_errors": int((info[:, 3] & 2).astype(bool).sum()), "caching_ok": int((info[:, 3] & 1).sum()),
                      "device_resident": {"seconds": dt_dev, "pairs_per_s": n / dt_dev, "equals_host_call": same},
                      "note": "whole C-ABI call: chunks of 65536 pairs through pinned staging on three streams (six input arrays H2D, kernel, lengths + the "
                              "written part of the three output arrays D2H); device_resident: c2_consensus_pairs_device on arrays that are already in HBM"}))


if __name__ == "__main__":
    main()
