#!/usr/bin/env python3
"""the reference AS SHIPPED on the benchmark's reads -- its own main() (`CRISPResso -r1 <fastq> -a <amplicon> -g <guide>
-p N --suppress_plots --suppress_report`) on a synthetic FASTQ of bench.py's read model, timed between its log lines
"Aligning sequences..." (CRISPRessoCORE.py:3731) and "Finished reads;" (:1987 / :1719).  Needs the reference sources
(/root/reference): it runs in the dev container only, whose CPU is NOT the GPU box's -- the number is a record of the as-shipped
path's overheads next to oracle/cpu_baseline.py's pure hot path on the same machine, not a baseline for the GPU figure.
  python tools/as_shipped_rate.py [--reads N] [--procs P]"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    import numpy as np
    from crispresso2_amd import synth
    L = 250
    reads = synth.make_reads(L, a.reads, workers=1)
    amp, _g, _inc = synth.amplicon_setup(L)
    c = L // 2
    guide = amp[c - 16:c + 4]                                   # cut point c = guide end - 3
    d = tempfile.mkdtemp(prefix="c2shipped_")
    fq = os.path.join(d, "reads.fastq")
    qual = b"I" * L
    with open(fq, "wb") as fh:
        for k in range(a.reads):
            fh.write(b"@read%d\n%s\n+\n%s\n" % (k, reads[k].tobytes(), qual))
    unique = len(np.unique(reads.view([("r", "V%d" % L)])))
    import make_golden as MG                                     # the reference's CRISPRessoCORE with its own compiled modules (oracle/_ref)
    core = MG.load_reference_core()
    import importlib
    P = importlib.import_module("CRISPResso2.plots.CRISPRessoPlot")
    for k_ in dir(P):
        if k_.startswith("plot_") and callable(getattr(P, k_)):
            setattr(P, k_, (lambda *x, **kw: None))
    marks = {}
    orig_info = core.info

    def info(msg, *x, **kw):
        s = str(msg)
        if s.startswith("Aligning sequences"):
            marks["start"] = time.perf_counter()
        if s.startswith("Finished reads"):
            marks["end"] = time.perf_counter()
        return orig_info(msg, *x, **kw)
    core.info = info
    argv = ["CRISPResso", "-r1", fq, "-a", amp, "-g", guide, "-p", str(a.procs), "--suppress_plots", "--suppress_report", "-o", d]
    old = sys.argv
    sys.argv = argv
    t0 = time.perf_counter()
    try:
        core.main()
    except SystemExit as e:
        assert e.code in (0, None), e.code
    finally:
        sys.argv = old
    total = time.perf_counter() - t0
    dt = marks["end"] - marks["start"]
    print(json.dumps({"reads": a.reads, "unique_reads": int(unique), "procs": a.procs, "host_cpus": os.cpu_count(),
                      "align_section_seconds": dt, "reads_per_s": a.reads / dt, "unique_reads_per_s": unique / dt,
                      "whole_run_seconds": total, "where": "dev container (not the GPU box)",
                      "command": "CRISPResso -r1 reads.fastq -a <250 bp amplicon> -g <guide> -p %d --suppress_plots --suppress_report" % a.procs}))


if __name__ == "__main__":
    main()
