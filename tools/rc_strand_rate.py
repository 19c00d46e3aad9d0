#!/usr/bin/env python3
"""Round 3 measurement: the launch chain over reads that are aligned as they are against the same reads given reverse-complemented
with strand = 1 (the kernels complement them while staging): python tools/rc_strand_rate.py [reads]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from crispresso2_amd import synth, _native, CRISPResso2Align as A
from crispresso2_amd.batch import BatchAligner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
L = 250
amp, g, inc = synth.amplicon_setup(L)
m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
dev = torch.device("cuda", 0)
al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
reads = synth.make_reads(L, n, workers=int(os.environ.get("C2_WORKERS", "16")))
comp = np.zeros(256, dtype=np.uint8)
for a_, b_ in zip(b"ACGTN", b"TGCAN"):
    comp[a_] = b_
rc = comp[reads[:, ::-1]]
stride = al.stride_for(L)
d_off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
out = {}
recs = {}
for name, arr, strand in (("forward", reads, 0), ("reverse complement, strand 1", rc, 1)):
    d_reads = torch.from_numpy(np.ascontiguousarray(arr)).to(dev).reshape(-1)
    d_str = torch.full((n,), strand, dtype=torch.uint8, device=dev)
    a = torch.empty((n, stride), dtype=torch.uint8, device=dev); f = torch.empty_like(a)
    r = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), a.data_ptr(), f.data_ptr(), r.data_ptr(), stride, L, d_strands=d_str.data_ptr(), stream=s)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    out[name] = {"chain_ms": round(best, 3), "M_alignments_per_s": round(n / best / 1e3, 1)}
    recs[name] = r.cpu().numpy().copy()
out["records_identical"] = bool(np.array_equal(recs["forward"], recs["reverse complement, strand 1"]))
print(json.dumps(out))
