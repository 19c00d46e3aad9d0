#!/usr/bin/env python3
"""The device ingest alone on a 10 M-read synthetic file (for rocprofv3 --kernel-trace --stats): python tools/device_ingest_only.py [reads] [reps]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crispresso2_amd import synth, _native, fastq_device as FD
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reads = synth.make_reads(250, n, workers=int(os.environ.get("C2_WORKERS", "32")))
d = tempfile.mkdtemp(prefix="c2dio_", dir="/dev/shm")
p = os.path.join(d, "r.fastq")
synth.write_fastq(reads, p)
del reads
ctx = _native.default_context()
dev = torch.device("cuda", 0)
FD.CHUNK_BYTES = int(os.environ.get("C2_CHUNK_MB", "64")) << 20
try:
    for rep in range(reps):
        tm = {}
        t0 = time.perf_counter()
        out = FD.ingest_file(p, ctx, dev, timings=tm)
        torch.cuda.synchronize()
        print(json.dumps({"seconds": round(time.perf_counter() - t0, 4), "stages": {k: round(v, 4) for k, v in tm.items()}, "unique": out["n_unique"], "finish_trace_ms": out.get("finish_trace_ms")}), flush=True)
        del out
finally:
    os.remove(p)
    os.rmdir(d)
