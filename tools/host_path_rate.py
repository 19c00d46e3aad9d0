#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer API (c2_align_classify_batch_host): reads start in pageable host memory,
aligned strings and records end there.  Not the headline (bench.py times HBM-resident data); DESIGN.md quotes this."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from crispresso2_amd import synth, _native, CRISPResso2Align as A  # noqa: E402
from crispresso2_amd.batch import BatchAligner  # noqa: E402

L, n = 250, 2_000_000
amp, g, inc = synth.amplicon_setup(L)
reads = synth.make_reads(L, n, workers=16)
al = BatchAligner([amp], [g], [inc], A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL")), -20, -2, ctx=_native.Context(0))
off = np.arange(n + 1, dtype=np.uint64) * L
al.align((reads[:300000].reshape(-1), off[:300001]))           # warm-up: context, scratch, pinned staging
out = {"reads": n}
for label, env in (("pipelined", None), ("one_shot", str(1 << 40))):
    if env:
        os.environ["C2_HOST_PIPE_MIN_TASKS"] = env
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        res = al.align((reads.reshape(-1), off))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out[label] = {"host_path_reads_per_s": n / best, "seconds": best}
    os.environ.pop("C2_HOST_PIPE_MIN_TASKS", None)
out["bytes_moved_per_read"] = L + 8 + 2 * res.aln_read.shape[1] + 32
print(json.dumps(out))
