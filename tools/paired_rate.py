#!/usr/bin/env python3
"""Paired FASTQ -> count tensors on the device route (paired_device.quantify_paired_fastq): N synthetic pairs cut from the benchmark's 250-bp
amplicon reads (read 1 = the first 150 bases, read 2 = the reverse complement of the last 150: 50 overlapping bases; qualities from a small
alphabet so that the consensus has to choose), pairs/s over three runs, stage times of the last.   python tools/paired_rate.py [pairs]"""
import json, os, shutil, sys, tempfile, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from crispresso2_amd import synth, _native, paired_device, refs as R, CRISPResso2Align as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L, RL = 250, 150
reads = synth.make_reads(L, n, workers=int(os.environ.get("C2_WORKERS", "32")))
comp = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    comp[a] = b
r1 = reads[:, :RL]
r2 = comp[reads[:, L - RL:]][:, ::-1]
rng = np.random.default_rng(5)
d = tempfile.mkdtemp(prefix="c2pair_", dir=os.environ.get("C2_E2E_DIR", "/dev/shm"))
p1, p2 = os.path.join(d, "r1.fastq"), os.path.join(d, "r2.fastq")


def write(path, seqs, seed):
    q = np.frombuffer(b"FI:5", dtype=np.uint8)[np.random.default_rng(seed).integers(0, 4, (1 << 16, RL))]
    with open(path, "wb") as fh:
        for c0 in range(0, len(seqs), 1 << 16):
            blk = seqs[c0:c0 + (1 << 16)]
            m = len(blk)
            rec = np.empty((m, 2 * RL + 16), dtype=np.uint8)
            rec[:, :10] = np.frombuffer(b"@r00000000", dtype=np.uint8)
            ids = np.arange(c0, c0 + m)
            for k in range(8):
                rec[:, 9 - k] = 48 + (ids // 10 ** k) % 10
            rec[:, 10] = 10
            rec[:, 11:11 + RL] = blk
            rec[:, 11 + RL] = 10; rec[:, 12 + RL] = 43; rec[:, 13 + RL] = 10
            rec[:, 14 + RL:14 + 2 * RL] = q[:m]
            rec[:, 14 + 2 * RL] = 10
            fh.write(rec[:, :15 + 2 * RL].tobytes())


write(p1, r1, 1)
write(p2, r2, 2)
del reads, r1, r2
amp, g, inc = synth.amplicon_setup(L)
args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                       ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False, use_legacy_insertion_quantification=False,
                       assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False,
                       prime_editing_pegRNA_scaffold_seq='', prime_editing_pegRNA_extension_seq='')
ref = R.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
mat = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
ctx = _native.default_context()
try:
    for rep in range(5):
        tm = {} if rep >= 2 else None
        paired_device.FORCE_HOST_PARSER = rep >= 3                     # (the last two runs: the pair keys from the host's lock-step parser)
        t0 = time.perf_counter()
        res = paired_device.quantify_paired_fastq(p1, p2, {"Reference": ref}, ["Reference"], mat, args, ctx=ctx, timings=tm)
        dt = time.perf_counter() - t0
        c = res.per_ref["Reference"]
        print(json.dumps({"pairs": n, "seconds": round(dt, 4), "pairs_per_s": round(n / dt), "N_TOTAL": res.stats["N_TOTAL"], "unique_pairs_aligned": res.stats["N_COMPUTED_ALN"],
                          "modified": c["counts_modified"], "route": res.ingest_route, "stages": None if tm is None else {k: round(v, 4) for k, v in tm.items()}}), flush=True)
        del res
except paired_device.PairedDeviceUnavailable as e:
    print(json.dumps({"unavailable": str(e)}))
finally:
    shutil.rmtree(d, ignore_errors=True)
