#!/usr/bin/env python3
"""Round 3 measurement: where c2_count_vectors_kernel's time goes, by kind of alignment -- the same kernel over batches of ONE kind:
reads equal to the amplicon, reads with k substitutions and no gap, reads with one deletion, reads with one insertion, and the
benchmark's own mix.  python tools/count_kernel_split.py [reads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from crispresso2_amd import synth, _native, counts as C, CRISPResso2Align as A
from crispresso2_amd.batch import BatchAligner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
L = 250
amp, g, inc = synth.amplicon_setup(L)
m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
dev = torch.device("cuda", 0)
al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
rng = np.random.default_rng(3)
base = np.frombuffer(amp.encode(), dtype=np.uint8)
other = {65: 67, 67: 71, 71: 84, 84: 65}


def batch(kind, k=2):
    """-> uint8 [n, L]: n reads of one kind (256 distinct ones, tiled)"""
    rows = np.tile(base, (256, 1))
    for r in range(256):
        if kind == "subs":
            for p in rng.choice(L, k, replace=False):
                rows[r, p] = other[rows[r, p]]
        elif kind == "del":
            x, d = int(rng.integers(60, 180)), int(rng.integers(3, 12))
            rows[r] = np.concatenate([base[:x], base[x + d:], rng.choice([65, 67, 71, 84], d).astype(np.uint8)])
        elif kind == "ins":
            x, d = int(rng.integers(60, 180)), int(rng.integers(2, 8))
            rows[r] = np.concatenate([base[:x], rng.choice([65, 67, 71, 84], d).astype(np.uint8), base[x:L - d]])
    return np.tile(rows, (n // 256 + 1, 1))[:n]


def run(name, reads2d):
    d_reads = torch.from_numpy(np.ascontiguousarray(reads2d)).to(dev).reshape(-1)
    d_off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    stride = al.stride_for(L)
    a = torch.empty((n, stride), dtype=torch.uint8, device=dev); f = torch.empty_like(a)
    r = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), a.data_ptr(), f.data_ptr(), r.data_ptr(), stride, L, stream=s)
    layout = C.CountLayout(1, L, L)
    d_counts = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        C.accumulate_device(ctx, layout, n, a.data_ptr(), f.data_ptr(), stride, r.data_ptr(), d_counts.data_ptr(), stream=s)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    rec = r.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    print(json.dumps({"kind": name, "alignments": n, "count_ms": round(best, 3), "ns_per_alignment": round(best * 1e6 / n, 3),
                      "mean_aln_len": float(rec["aln_len"].mean()), "gap_free_share": float((rec["aln_len"] == L).mean())}), flush=True)


run("equal to the amplicon", batch("perfect"))
for k in (1, 2, 5):
    run("gap-free, %d substitutions" % k, batch("subs", k))
run("one deletion of 3-11 bases", batch("del"))
run("one insertion of 2-7 bases", batch("ins"))
mix = synth.make_reads(L, n, workers=int(os.environ.get("C2_WORKERS", "16")))
run("the benchmark's mix", np.stack([np.frombuffer(x.tobytes(), dtype=np.uint8) for x in mix]) if not isinstance(mix, np.ndarray) else mix)
