#!/usr/bin/env python3
"""What the UNCHANGED caller gets: the reference's hot loop calls global_align and
find_indels_substitutions once per unique read (CRISPRessoCORE.py:1957-1981).  Measured here on the GPU box, through the two shim
modules' public functions exactly as CRISPRessoCORE calls them, for N unique synthetic reads (bench.py's read model, 250 bp):
  per_call   every call is a kernel launch + synchronisation (no reads registered)
  primed     crispresso2_amd.prime.register_reads(...) first: one device batch per amplicon, then dictionary look-ups
Both loops must return the same objects.  Run:  python tools/shim_call_rate.py [--reads N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--len", type=int, default=250, dest="L")
    a = ap.parse_args()
    import numpy as np
    from crispresso2_amd import synth, prime, CRISPResso2Align as A, CRISPRessoCOREResources as R
    reads_u8 = synth.make_reads(a.L, 4 * a.reads)
    reads = list(dict.fromkeys(r.tobytes().decode() for r in reads_u8))[:a.reads]
    amp, g, inc = synth.amplicon_setup(a.L)
    g = np.asarray(g, dtype=np.int64)
    inc = np.array(inc)
    m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))

    def loop(rs):
        out = []
        for rd in rs:
            s1, s2, sc = A.global_align(rd, amp, matrix=m, gap_incentive=g, gap_open=-20, gap_extend=-2)
            p = R.find_indels_substitutions(s1, s2, inc)
            out.append((s1, s2, sc, p["insertion_n"], p["deletion_n"], p["substitution_n"], tuple(p["all_deletion_positions"][:4])))
        return out
    loop(reads[:50])                                             # context, allocations
    n_pc = min(len(reads), 3000)
    t0 = time.perf_counter()
    ref = loop(reads[:n_pc])
    t_pc = time.perf_counter() - t0
    # latency of one call of each kind
    t0 = time.perf_counter()
    for rd in reads[:500]:
        A.global_align(rd, amp, matrix=m, gap_incentive=g, gap_open=-20, gap_extend=-2)
    t_align = (time.perf_counter() - t0) / 500
    s1, s2, _ = A.global_align(reads[0], amp, matrix=m, gap_incentive=g, gap_open=-20, gap_extend=-2)
    t0 = time.perf_counter()
    for _ in range(500):
        R.find_indels_substitutions(s1, s2, inc)
    t_cls = (time.perf_counter() - t0) / 500
    prime.register_reads(reads)
    t0 = time.perf_counter()
    got = loop(reads)
    t_pr = time.perf_counter() - t0
    st = dict(prime.stats)
    prime.clear()
    same = got[:n_pc] == ref
    print(json.dumps({"unique_reads": len(reads), "read_len": a.L,
                      "per_call": {"reads": n_pc, "seconds": t_pc, "reads_per_s": n_pc / t_pc, "global_align_call_us": 1e6 * t_align,
                                   "find_indels_substitutions_call_us": 1e6 * t_cls},
                      "primed": {"reads": len(reads), "seconds": t_pr, "reads_per_s": len(reads) / t_pr, "stats": st},
                      "identical": bool(same),
                      "note": "one Python process; the primed loop's time includes the two device batches (alignments, classifier lists) and "
                              "building the reference's Python objects at look-up time"}))


if __name__ == "__main__":
    main()
