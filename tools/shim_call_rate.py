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
    ap.add_argument("--procs", type=int, default=0, help="also: the -p N shape -- everything primed in the parent, then N fork()ed workers loop over their slices")
    ap.add_argument("--fork-reads", type=int, default=200000)
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
    forked = None
    if a.procs > 1:
        # the reference's -p N (CRISPRessoCORE.py:1870-1898): the parent has used the device, primes what the workers will ask for (crispresso2_amd.prime
        # does that by itself in a before-fork hook when the caller is the reference's process_fastq; here: the same call, made directly), forks,
        # and every worker answers its slice from the memory it inherited
        import multiprocessing as mp
        from types import SimpleNamespace
        from crispresso2_amd import refs as RF, _native
        big = list(dict.fromkeys(r.tobytes().decode() for r in synth.make_reads(a.L, 3 * a.fork_reads)))[:a.fork_reads]
        args = SimpleNamespace(aln_seed_count=5, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2, use_legacy_insertion_quantification=False)
        ref_obj = RF.make_ref("Reference", amp, [a.L // 2], inc, min_aln_score=60)
        ref_obj["gap_incentive"] = g
        t0 = time.perf_counter()
        prime.prime_for_caller(args, {"Reference": ref_obj}, ["Reference"], m, reads=big)
        t_prime = time.perf_counter() - t0
        b = [len(big) * k // a.procs for k in range(a.procs + 1)]

        def work(lo, hi, q):
            r = loop(big[lo:hi])
            q.put((len(r), dict(prime.stats)["align_hits"], _native._helper[1].calls if _native._helper[0] == os.getpid() else 0))
        ctx_ = mp.get_context("fork")
        q = ctx_.Queue()
        t0 = time.perf_counter()
        ps = [ctx_.Process(target=work, args=(b[k], b[k + 1], q)) for k in range(a.procs)]
        for p_ in ps:
            p_.start()
        res = [q.get() for _ in ps]
        for p_ in ps:
            p_.join()
        t_loop = time.perf_counter() - t0
        forked = {"procs": a.procs, "reads": len(big), "prime_seconds": t_prime, "loop_seconds": t_loop, "reads_per_s_loop": len(big) / t_loop,
                  "reads_per_s_with_priming": len(big) / (t_loop + t_prime), "helper_calls": sum(r[2] for r in res), "done": sum(r[0] for r in res),
                  "note": "the loop body here is two calls and a tuple per read -- lighter than the reference's get_new_variant_object + JSON line per read: the "
                          "shim's share of a -p N run, not the run"}
        prime.clear()
    print(json.dumps({"unique_reads": len(reads), "read_len": a.L,
                      "per_call": {"reads": n_pc, "seconds": t_pc, "reads_per_s": n_pc / t_pc, "global_align_call_us": 1e6 * t_align,
                                   "find_indels_substitutions_call_us": 1e6 * t_cls},
                      "primed": {"reads": len(reads), "seconds": t_pr, "reads_per_s": len(reads) / t_pr, "stats": st},
                      "identical": bool(same), "forked_workers": forked,
                      "note": "one Python process; the primed loop's time includes the two device batches (alignments, classifier lists) and "
                              "building the reference's Python objects at look-up time"}))


if __name__ == "__main__":
    main()
