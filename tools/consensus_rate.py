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
    bytes_per_pair = 4 * stride + 2 * qstride + 3 * ostride + 16 + 17
    print(json.dumps({"pairs": n, "seconds": dt, "pairs_per_s": n / dt, "bytes_moved_per_pair": int(bytes_per_pair),
                      "index_errors": int((info[:, 3] & 2).astype(bool).sum()), "caching_ok": int((info[:, 3] & 1).sum()),
                      "note": "whole C-ABI call: six input arrays H2D, kernel, three output arrays D2H, in chunks of 65536 pairs through pageable memory"}))


if __name__ == "__main__":
    main()
