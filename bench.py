#!/usr/bin/env python3
"""bench.py -- aligned+classified reads/sec of the align + classify hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic reads that is already resident in HBM: the launch chain
of the align kernels (c2_align_diagp_kernel<8> -> c2_align_diagp_kernel<4> -> c2_align_diagp_kernel<2> -> c2_align_classify_kernel,
each packed kernel followed by the 32-bit kernel of its band over the tasks it could not pair: NW fill with optimality
certificate, traceback, fused classification), for several candidate amplicons the strand / best-amplicon choice
(c2_select_best_kernel), the per-amplicon count tensor (c2_count_vectors_kernel) and its all-reduce over the ranks.
After the timed region: the checks (`checks` in the JSON line) -- an oracle sample, size-independent properties of every
alignment, the whole batch once more through the full-plane kernel alone compared byte for byte with what the chain wrote,
and every alignment of the cpu_baseline legs (the reference's own compiled code) against the device's.

  --config 3 (default)  BASELINE.json configs[2]: 10 M synthetic 250 bp reads vs one 250 bp amplicon -- the configuration
                        the metric is quoted on
  --config 2            configs[1]: 1 M x 150 bp reads vs one 150 bp amplicon
  --config 4            configs[3]: 10 M x 250 bp reads, every read against 3 candidate amplicons (wild type, HDR, prime edit;
                        60 / 20 / 20 % of the reads derive from them), best-amplicon choice inside the step
  --config 5            configs[4]: CRISPRessoPooled-style, 96 amplicons, reads tagged with their amplicon: this rank's
                        shard of the 100 M-read stream (12.5 M reads per GPU)
N > 1: every rank runs the same amount of work on its own shard of the read stream (weak scaling, no data-path collective;
the all-reduce of the per-amplicon count tensor is the only exchange and is inside the step).

After the headline's timed region and checks, the default run (no --config / --reads / --len / --kernel / --no-extras) adds, in
the same JSON line:
  int32_chain    the same batch through the 32-bit launch chain (c2_align_diagx_kernel<4> -> <2> -> c2_align_diag_kernel -> full plane:
                 the reference's C-int arithmetic, CRISPResso2Align.pyx:142-147), timed the same way -- the number that stands next to
                 the headline's packed-int16 first tier
  other_configs  BASELINE.json configs[1], [3], [4] (--config 2, 4, 5) at full size: 3 timed steps each and the chain = full-plane
                 comparison of every alignment
  e2e            FASTQ -> count tensors (pipeline.quantify_fastq: ingest + exact de-duplication, seed test, alignments, selection,
                 reverse-complement merge, count kernel) on the headline's reads written to /dev/shm: plain text and BGZF, each framed and
                 de-duplicated on the device under its upload (fastq_device) and, for comparison, through the native host parser

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 it runs one rank per GPU over RCCL: under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) it is one of the ranks; started bare it
spawns its own N ranks on this node (the reference's fan-out needs no launcher either, CRISPRessoCORE.py:1870-1898) and rank 0's
line is the output.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0                          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LINK_PEAK_GBS = 63.0                           # "Host link: PCIe Gen5 x16, 63 GB/s (spec)", MI355X_MICROARCH.md, one direction
N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9
# MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32
VALU_SIMD32_WAVE_INSTR_PER_S = N_SIMD * CLOCK_HZ / 2.0
# measured on this chip with counted cycles (tools/valu_microbench4.hip under rocprofv3 --pmc GRBM_GUI_ACTIVE, profiles/r03/valu_microbench4.txt;
# the clock during those kernels is 2.40 GHz): two issue classes.  Half rate, ~4.15 cycles of SIMD time per wave64 instruction: v_max/min,
# every VOP3 three-operand form, v_perm_b32, v_pk_*, DPP forms, carry / compare forms, v_lshlrev.  Full rate, ~2.3 cycles: v_add/sub_u32,
# v_and/or/xor, right shifts, v_mov, v_fma_f32 (the guide's reference point).  The DP cell is mostly half-rate opcodes.
VALU_MEASURED_CYCLES_PER_INSTR = 4.15
VALU_FULL_RATE_CYCLES_PER_INSTR = 2.3
VALU_MEASURED_SOURCE = "profiles/r03/valu_microbench4.txt (tools/valu_microbench4.hip, cycles from GRBM_GUI_ACTIVE; clock 2.40 GHz)"
PROFILE_ROUNDS = ["r06", "r05", "r04", "r03", "r02"]                 # the PMC summary of the newest round that has one
GO, GE, MIN_ALN_SCORE = -20, -2, 60.0

CONFIG_DEFAULTS = {2: (150, 1_000_000), 3: (250, 10_000_000), 4: (250, 10_000_000), 5: (250, 12_500_000)}


def build_workload(config, L, n, rank, workers):
    """-> dict(refs=[(seq, gap_incentive, include)], reads uint8 [n, L], ref_ids uint16 [n] or None, all_refs, text)"""
    from crispresso2_amd import synth
    blocks = (n + synth.BLOCK - 1) // synth.BLOCK
    if config in (2, 3):
        amp, g, inc = synth.amplicon_setup(L)
        reads = synth.make_reads(L, n, first_block=rank * blocks, workers=workers)
        return dict(refs=[(amp, g, inc)], reads=reads, ref_ids=None, all_refs=False,
                    text="%s synthetic %d bp reads vs one %d bp amplicon per GPU (BASELINE.json configs[%d] shape), every read aligned "
                         "(dedup off)" % ("{:,}".format(n), L, L, config - 1))
    if config == 4:
        amp, g, inc = synth.amplicon_setup(L)
        variants = [amp, synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")]
        refs = []
        for r in variants:
            x = np.zeros(len(r) + 1, dtype=np.int64)
            x[L // 2 + 1] = 1
            refs.append((r, x, inc))
        # 60 % of the reads derive from the amplicon, 20 % from each variant (its first L bases), in a fixed pattern of
        # BLOCK-read blocks [wt wt wt hdr pe]; the same error / indel model on all of them
        reads = np.empty((n, L), dtype=np.uint8)
        pattern = [0, 0, 0, 1, 2]
        n_blocks_of = [sum(1 for b in range(blocks) if pattern[b % 5] == src) for src in range(3)]
        pools = [synth.make_reads(L, n_blocks_of[src] * synth.BLOCK, amplicon_id=100 + src, amplicon=variants[src][:L],
                                  first_block=rank * blocks, workers=workers) if n_blocks_of[src] else None for src in range(3)]
        per_src = [0, 0, 0]
        for b in range(blocks):
            src = pattern[b % 5]
            a0, m = b * synth.BLOCK, min(synth.BLOCK, n - b * synth.BLOCK)
            reads[a0:a0 + m] = pools[src][per_src[src] * synth.BLOCK:per_src[src] * synth.BLOCK + m]
            per_src[src] += 1
        del pools
        return dict(refs=refs, reads=reads, ref_ids=None, all_refs=True,
                    text="%s synthetic %d bp reads per GPU, each against 3 candidate amplicons (wild type %d bp, HDR %d bp, prime edit %d bp; "
                         "60/20/20 %% of the reads derive from them; BASELINE.json configs[3] shape), best-amplicon choice in the step, "
                         "every read aligned (dedup off)" % ("{:,}".format(n), L, len(variants[0]), len(variants[1]), len(variants[2])))
    if config == 5:
        n_amp = 96
        setups = [synth.amplicon_setup(L, 1000 + k) for k in range(n_amp)]
        per = (n + n_amp - 1) // n_amp
        pb = (per + synth.BLOCK - 1) // synth.BLOCK
        reads = np.empty((n, L), dtype=np.uint8)
        rids = np.empty(n, dtype=np.uint16)
        pos = 0
        for k in range(n_amp):
            m = min(per, n - pos)
            if m <= 0:
                break
            reads[pos:pos + m] = synth.make_reads(L, m, amplicon_id=1000 + k, amplicon=setups[k][0], first_block=rank * pb,
                                                  workers=workers)
            rids[pos:pos + m] = k
            pos += m
        return dict(refs=setups, reads=reads, ref_ids=rids, all_refs=False,
                    text="%s synthetic %d bp reads per GPU = this rank's shard of the pooled stream, 96 amplicons of %d bp, every read "
                         "tagged with its amplicon and grouped by amplicon in the input (BASELINE.json configs[4] shape), every read "
                         "aligned (dedup off)" % ("{:,}".format(n), L, L))
    raise SystemExit("--config must be 2, 3, 4 or 5")


def build_robust_workloads(L, n, rank, workers):
    """The headline's read budget with inputs that are NOT the synthetic generator's best case: every read of the
    generator is exactly as long as the amplicon and most are gap-free, which is what the score-only stage and the first band tier are
    fastest on.  -> {name: workload dict with `reads` padded to a common width with 0 and `lens`}"""
    from crispresso2_amd import synth
    blocks = (n + synth.BLOCK - 1) // synth.BLOCK
    out = {}
    # (i) reads shaped like the reference's own test data: tests/FANC.Cas9.fastq resampled (crispresso2_amd/fanc_profile.json) -- a 4-base
    # overhang in front of the 223-bp amplicon, 23+ bases of genomic flank behind it, lengths 248-250 (a few much shorter), deletions at the
    # cut in a third of the reads, 5 % unrelated reads
    amp, g, inc = synth.fanc_setup()
    fr, fl = synth.make_fanc_reads(n, first_block=rank * blocks, workers=workers)
    out["fanc_shaped"] = dict(refs=[(amp, g, inc)], reads=fr, lens=fl, ref_ids=None, all_refs=False, max_len=int(fl.max()),
                              text="%s reads resampled from the length / overhang / indel signatures of the reference's tests/FANC.Cas9.fastq "
                                   "(4-base leading overhang, 23+ bases of flank behind the amplicon, 5 %% unrelated reads) vs the %d bp FANC amplicon"
                                   % ("{:,}".format(n), len(amp)))
    # (ii) the generator's reads cut to lengths U[200, L]: the read ends inside the amplicon (a trailing deletion of 0-50 bases)
    amp2, g2, inc2 = synth.amplicon_setup(L)
    base = synth.make_reads(L, n, first_block=(rank + 64) * blocks, workers=workers)
    rng = np.random.default_rng([20240604, rank])
    W = (L + 15) // 16 * 16
    vr = np.zeros((n, W), dtype=np.uint8)
    vr[:, :L] = base
    vl = rng.integers(min(200, L), L + 1, n).astype(np.int32)
    vr[np.arange(W, dtype=np.int32)[None, :] >= vl[:, None]] = 0
    out["lengths_200_to_L"] = dict(refs=[(amp2, g2, inc2)], reads=vr, lens=vl, ref_ids=None, all_refs=False, max_len=L,
                                   text="%s reads of the synthetic generator cut to lengths U[%d, %d] vs the %d bp amplicon" % ("{:,}".format(n), min(200, L), L, L))
    # (iii) one read in ten is unrelated to the amplicon (random bases): no band can certify it, it ends in the full-matrix launch
    ur = np.zeros((n, W), dtype=np.uint8)
    ur[:, :L] = base
    del base
    junk = np.nonzero(rng.random(n) < 0.10)[0]
    ur[junk, :L] = synth._bases(rng.integers(0, 4, (len(junk), L), dtype=np.uint8))
    out["unrelated_10_percent"] = dict(refs=[(amp2, g2, inc2)], reads=ur, lens=np.full(n, L, dtype=np.int32), ref_ids=None, all_refs=False, max_len=L,
                                       text="%s reads of the synthetic generator, 10 %% of them replaced by random sequences, vs the %d bp amplicon" % ("{:,}".format(n), L))
    return out


SPAWN_CMD = [sys.executable, os.path.abspath(__file__)]     # what a rank is (tests/bench_emulated_main.py puts its own script here)


def _spawn_ranks(n):
    """`python bench.py --gpus N` started bare: this process becomes the launcher of N ranks of itself on this node (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them; rank k uses GPU k).  The ranks inherit stdout, so
    rank 0's JSON line is this command's output.  A rank that fails takes the others down with it."""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), C2_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # the host driver only supports dmabuf IPC (RCCL needs it)
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        procs.append(subprocess.Popen(SPAWN_CMD + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:                                      # our own children, by handle
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class Job:
    """One workload resident in HBM: its reads, output buffers, count tensor, the step (launch chain -> best-amplicon choice ->
    count pass -> all-reduce), the timed run and the exhaustive chain-vs-full-plane comparison."""

    def __init__(self, ctx, wl, L, matrix, dev, world, kernel="auto", overlap_count=False):
        import torch
        from crispresso2_amd import counts as C
        from crispresso2_amd.batch import BatchAligner
        self.torch, self.C, self.ctx, self.wl, self.L, self.dev, self.world, self.kernel = torch, C, ctx, wl, L, dev, world, kernel
        refs, reads = wl["refs"], wl["reads"]
        self.all_refs = all_refs = wl["all_refs"]
        self.n = n = reads.shape[0]
        self.k = k = len(refs)
        self.n_tasks = n_tasks = n * (k if all_refs else 1)
        ctx.set_kernel_mode(kernel)
        self.al = BatchAligner([r[0] for r in refs], [r[1] for r in refs], [r[2] for r in refs], matrix, GO, GE, ctx=ctx)
        self.stride = stride = self.al.stride_for(L)
        self.Lmax = self.al.max_ref_len
        self.min_len = 0
        if wl.get("lens") is not None:                                # ragged reads: rows padded with 0 + their lengths -> arena + offsets
            from crispresso2_amd import synth
            arena, off = synth.pack_ragged(reads, wl["lens"])
            self.d_reads = torch.from_numpy(arena).to(dev)
            self.d_offsets = torch.from_numpy(off.view(np.int64)).to(dev)
            self.min_len = int(wl["lens"].min())
            del arena
        else:
            self.d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
            self.d_offsets = (torch.arange(n + 1, dtype=torch.int64, device=dev) * L)
        self.d_rids = None if wl["ref_ids"] is None else torch.from_numpy(wl["ref_ids"].astype(np.int16)).to(dev)
        # output buffers: one set; two with --overlap-count, so that batch k+1 is aligned while batch k is still being counted
        self.n_sets = 2 if overlap_count else 1
        self.out_sets = [(torch.empty((n_tasks, stride), dtype=torch.uint8, device=dev), torch.empty((n_tasks, stride), dtype=torch.uint8, device=dev),
                          torch.empty((n_tasks, 32), dtype=torch.uint8, device=dev)) for _ in range(self.n_sets)]
        # single amplicon: the hint words c2_align_partition_kernel leaves for the reads it finishes itself (c2_batch.diag_hints); the count pass takes those
        # tasks from the word alone (c2_count_hinted_kernel).  C2_BENCH_NO_HINTS=1: the step without them (A/B).
        # (an all-references batch can be counted from hints too -- c2_count_vectors_hinted_device takes the selection's weights; measured on config 4: the count
        #  pass 4.7 -> 1.4 ms, the chain + 3.3 ms for collecting and storing the hints of 30 M alignments of which 20 M are never counted: not used.  C2_BENCH_ALLREFS_HINTS=1: used)
        self.use_hints = (not (all_refs and k > 1) or bool(os.environ.get("C2_BENCH_ALLREFS_HINTS"))) and not os.environ.get("C2_BENCH_NO_HINTS")
        self.hint_sets = [torch.zeros(n_tasks * 4, dtype=torch.int32, device=dev) if self.use_hints else None for _ in range(self.n_sets)]      # four words per task
        self.t_align = torch.cuda.current_stream()
        self.t_count = torch.cuda.Stream(device=dev) if overlap_count else self.t_align
        self.stream = self.t_align.cuda_stream
        self.count_stream = self.t_count.cuda_stream
        self.aligned_ev = [torch.cuda.Event() for _ in range(self.n_sets)]
        self.counted_ev = [None] * self.n_sets
        # per-amplicon count tensor (CRISPRessoCORE.py:3865-4115 on the device) -- the only thing the GPUs exchange
        self.layout = C.CountLayout(k, self.Lmax, L)
        self.d_counts = torch.zeros(self.layout.shape(), dtype=torch.int64, device=dev)
        self.min_matches = C.min_matches_table([MIN_ALN_SCORE] * k, self.Lmax + L)          # --default_min_aln_score 60
        self.d_weights = self.d_selstats = None
        if all_refs:
            self.d_weights = torch.zeros(n_tasks, dtype=torch.int32, device=dev)
            self.d_selstats = torch.zeros(len(C.SELECT_STATS), dtype=torch.int64, device=dev)
            self.min_mscore = C.min_mscore_table([MIN_ALN_SCORE] * k)
        self.step_no = 0

    @property
    def outputs(self):
        return self.out_sets[0]

    def align_into(self, a_read, a_ref, recs, hints=None):
        self.al.align_device(self.n, self.d_reads.data_ptr(), self.d_offsets.data_ptr(), a_read.data_ptr(), a_ref.data_ptr(), recs.data_ptr(),
                             self.stride, self.L, d_ref_ids=None if self.d_rids is None else self.d_rids.data_ptr(), all_refs=self.all_refs,
                             stream=self.stream, min_read_len=self.min_len, d_hints=None if hints is None else hints.data_ptr())

    def step(self, e=None):
        """One batch: launch chain on the align stream into buffer set i; reference choice, count pass and all-reduce on the count
        stream (the same stream unless --overlap-count).  Set i is aligned into again only after its previous batch has been counted."""
        torch, C = self.torch, self.C
        i = self.step_no % self.n_sets
        self.step_no += 1
        a_read, a_ref, recs = self.out_sets[i]
        hints = self.hint_sets[i]
        if self.counted_ev[i] is not None:
            self.t_align.wait_event(self.counted_ev[i])
        if e: e[0].record(self.t_align)
        self.align_into(a_read, a_ref, recs, hints)
        if e: e[1].record(self.t_align)
        self.aligned_ev[i].record(self.t_align)
        self.t_count.wait_event(self.aligned_ev[i])
        with torch.cuda.stream(self.t_count):
            if e: e[4].record(self.t_count)
            if self.all_refs:
                # strand / best-amplicon choice on the device (CRISPRessoCORE.py:697-707) -> the weight of every alignment in the count pass
                self.d_selstats.zero_()
                C.select_best_device(self.ctx, self.n, self.k, recs.data_ptr(), self.min_mscore, C.SELECT_DROP_AMBIGUOUS, self.Lmax + self.L,
                                     d_weights=self.d_weights.data_ptr(), d_stats=self.d_selstats.data_ptr(), stream=self.count_stream)
            if e: e[2].record(self.t_count)
            self.d_counts.zero_()
            C.accumulate_device(self.ctx, self.layout, self.n_tasks, a_read.data_ptr(), a_ref.data_ptr(), self.stride, recs.data_ptr(),
                                self.d_counts.data_ptr(), d_weights=self.d_weights.data_ptr() if self.all_refs else None,
                                min_matches=None if self.all_refs else self.min_matches,
                                flags=C.FLAG_ALL_REFS_LAYOUT if self.all_refs else 0, stream=self.count_stream,
                                d_hints=None if hints is None else hints.data_ptr())
            C.all_reduce(self.d_counts)
            if e: e[3].record(self.t_count)
            self.counted_ev[i] = torch.cuda.Event()
            self.counted_ev[i].record(self.t_count)

    def fence(self):
        import torch.distributed as dist
        self.torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, warmup, steps):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; the time is the maximum over the ranks."""
        import torch.distributed as dist
        torch = self.torch
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
        for _ in range(warmup):
            self.step()
        self.fence()
        self.ctx.timing_enable(True)
        t0 = time.perf_counter()
        for s_ in range(steps):
            self.step(ev[s_])
        self.fence()
        dt = time.perf_counter() - t0
        kernel_ms, first_ms, launches = self.ctx.timing_read_split()
        self.ctx.timing_enable(False)
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ns = max(steps, 1)
        return dict(dt=dt, steps=steps, kernel_ms=kernel_ms, first_ms=first_ms, launches=launches,
                    align_ms=sum(e[0].elapsed_time(e[1]) for e in ev) / ns, select_ms=sum(e[4].elapsed_time(e[2]) for e in ev) / ns,
                    count_ms=sum(e[2].elapsed_time(e[3]) for e in ev) / ns,
                    reads_per_s=self.world * self.n * steps / dt, alignments_per_s=self.world * self.n_tasks * steps / dt)

    def chain_equals_full_plane(self):
        """The launch chain's certificates, exhaustively: the SAME batch through the full-plane row-strip kernel (every cell of every
        matrix computed, any path followed) into second buffers; every aligned string and every record must be equal.
        -> (tasks that are equal, seconds of the full-plane pass)"""
        torch = self.torch
        d_aln_read, d_aln_ref, d_records = self.outputs
        n_tasks, stride, dev = self.n_tasks, self.stride, self.dev
        b_read = torch.zeros((n_tasks, stride), dtype=torch.uint8, device=dev)
        b_ref = torch.zeros((n_tasks, stride), dtype=torch.uint8, device=dev)
        b_rec = torch.zeros((n_tasks, 32), dtype=torch.uint8, device=dev)
        self.ctx.set_kernel_mode("full")
        torch.cuda.synchronize()
        tf = time.perf_counter()
        self.align_into(b_read, b_ref, b_rec)
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf
        self.ctx.set_kernel_mode(self.kernel)
        equal_n = 0
        cols = torch.arange(stride, device=dev)[None, :]
        recs_t = d_records.view(torch.int16)
        for a0 in range(0, n_tasks, 1_000_000):
            a1 = min(n_tasks, a0 + 1_000_000)
            valid = cols < recs_t[a0:a1, 0].to(torch.int64)[:, None]
            same = (((d_aln_read[a0:a1] == b_read[a0:a1]) & (d_aln_ref[a0:a1] == b_ref[a0:a1])) | ~valid).all(1)
            same &= (d_records[a0:a1] == b_rec[a0:a1]).all(1)
            equal_n += int(same.sum().item())
        del b_read, b_ref, b_rec
        torch.cuda.empty_cache()
        return equal_n, tf

    def properties_hold(self):
        """Size-independent properties on EVERY alignment of the batch (device-side, in slices): no double-gap column; removing the gaps
        gives back the read and the amplicon (every base exactly once, in order); `matches` and `all_deletion_bases` of the record agree
        with the strings."""
        torch = self.torch
        d_aln_read, d_aln_ref, d_records = self.outputs
        refs, k, n, L, Lmax, stride, dev, n_tasks = self.wl["refs"], self.k, self.n, self.L, self.Lmax, self.stride, self.dev, self.n_tasks
        props = True
        Li_t = torch.tensor([len(r[0]) for r in refs], dtype=torch.int64, device=dev)
        d_amps = torch.zeros((k, Lmax), dtype=torch.uint8, device=dev)
        for r in range(k):
            d_amps[r, :len(refs[r][0])] = torch.from_numpy(np.frombuffer(refs[r][0].encode(), dtype=np.uint8).copy()).to(dev)
        d_reads2 = self.d_reads.view(n, L)
        recs_t = d_records.view(torch.int16)                      # aln_len, matches are the first two uint16 (< 32768 here)
        cols = torch.arange(stride, device=dev)[None, :]
        dash = ord("-")
        SL = 500_000
        for a0 in range(0, n_tasks, SL):
            a1 = min(n_tasks, a0 + SL)
            tt = torch.arange(a0, a1, device=dev)
            rid_ = ((tt % k) if self.all_refs else
                    (self.d_rids[a0:a1].to(torch.int64) if self.d_rids is not None else torch.zeros(a1 - a0, dtype=torch.int64, device=dev)))
            rd_ = (tt // k) if self.all_refs else tt
            Li_ = Li_t[rid_]
            T_ = recs_t[a0:a1, 0].to(torch.int64)
            mt_ = recs_t[a0:a1, 1].to(torch.int64)
            delb_ = recs_t[a0:a1, 9].to(torch.int64)
            R_, F_ = d_aln_read[a0:a1], d_aln_ref[a0:a1]
            valid = cols < T_[:, None]
            rgap = (R_ == dash) & valid
            fgap = (F_ == dash) & valid
            ok = not bool((rgap & fgap).any())
            keep_r, keep_f = valid & ~rgap, valid & ~fgap
            ok = ok and bool((keep_r.sum(1) == L).all()) and bool((keep_f.sum(1) == Li_).all())
            if ok:
                ok = bool(torch.equal(R_[keep_r].view(a1 - a0, L), d_reads2[rd_]))
                # the reference's bases in order: rank of every kept column -> compare with the amplicon at that rank
                rank_f = torch.cumsum(keep_f.to(torch.int32), 1) - 1
                want = d_amps[rid_][:, :].gather(1, rank_f.clamp(min=0, max=Lmax - 1).to(torch.int64))
                ok = ok and bool(((F_ == want) | ~keep_f).all())
                ok = ok and bool(torch.equal(((R_ == F_) & keep_r & keep_f).sum(1), mt_)) and bool(torch.equal(rgap.sum(1), delb_))
                del rank_f, want
            props = props and ok
            del valid, rgap, fgap, keep_r, keep_f
        return props

    def local_reads_aligned(self):
        """reads of THIS rank's shard the count pass accepted in the last step (the step's own tensor is all-reduced: this one is not)"""
        torch, C = self.torch, self.C
        a_read, a_ref, recs = self.out_sets[(self.step_no - 1) % self.n_sets]
        t = torch.zeros_like(self.d_counts)
        with torch.cuda.stream(self.t_count):
            C.accumulate_device(self.ctx, self.layout, self.n_tasks, a_read.data_ptr(), a_ref.data_ptr(), self.stride, recs.data_ptr(), t.data_ptr(),
                                d_weights=self.d_weights.data_ptr() if self.all_refs else None, min_matches=None if self.all_refs else self.min_matches,
                                flags=C.FLAG_ALL_REFS_LAYOUT if self.all_refs else 0, stream=self.count_stream)
        torch.cuda.synchronize()
        host = t.cpu().numpy()
        return int(sum(self.layout.unpack(host, r, len(self.wl["refs"][r][0]))["counts_total"] for r in range(self.k)))

    def count_tensor_equals_without_hints(self):
        """the step's count tensor (hinted tasks from their hint word, c2_count_hinted_kernel) against the count pass over the same rows and records
        WITHOUT the hints (every task's strings read back) -- entry by entry.  None when the step uses no hints."""
        torch, C = self.torch, self.C
        if not self.use_hints or self.world > 1:
            return None
        a_read, a_ref, recs = self.out_sets[(self.step_no - 1) % self.n_sets]
        t = torch.zeros_like(self.d_counts)
        with torch.cuda.stream(self.t_count):
            C.accumulate_device(self.ctx, self.layout, self.n_tasks, a_read.data_ptr(), a_ref.data_ptr(), self.stride, recs.data_ptr(), t.data_ptr(),
                                d_weights=self.d_weights.data_ptr() if self.all_refs else None,        # (the last step's selection)
                                min_matches=None if self.all_refs else self.min_matches,
                                flags=C.FLAG_ALL_REFS_LAYOUT if self.all_refs else 0, stream=self.count_stream)
        torch.cuda.synchronize()
        return bool(torch.equal(t, self.d_counts))

    def tallies(self):
        host = self.d_counts.cpu().numpy()
        return [self.layout.unpack(host, r, len(self.wl["refs"][r][0])) for r in range(self.k)]

    def free(self):
        self.out_sets = []
        self.hint_sets = []
        self.d_reads = self.d_offsets = self.d_rids = self.d_weights = self.d_counts = None
        self.torch.cuda.empty_cache()


def _dedup_on_leg(job, reads, steps):
    """The same batch the way the reference feeds its aligner -- "dedup on": only the UNIQUE reads are aligned, their multiplicities are the
    weights of the count pass (SURVEY 8d asks for both; the headline is dedup off).  The de-duplication itself (host, outside the timed
    region here; the e2e leg times it inside a FASTQ -> tensors run) is a 64-bit hash + an exact byte comparison of every read with its
    group's first member."""
    torch, C, ctx, dev, L, n, k, layout = job.torch, job.C, job.ctx, job.dev, job.L, job.n, job.k, job.layout
    stride, stream = job.stride, job.stream
    t_d = time.perf_counter()
    W = (L + 7) // 8
    padded = np.zeros((n, W * 8), dtype=np.uint8)
    padded[:, :L] = reads
    rng_h = np.random.default_rng(99)
    m1 = (rng_h.integers(1, 1 << 62, W, dtype=np.uint64) << np.uint64(1)) | np.uint64(1)
    m2 = (rng_h.integers(1, 1 << 62, W, dtype=np.uint64) << np.uint64(1)) | np.uint64(1)
    with np.errstate(over="ignore"):
        x = padded.view(np.uint64) * m1[None, :]                  # (a plain multiply-sum loses the high bytes of a word: two
        x ^= x >> np.uint64(32)                                   #  differences there cancel with probability 1/256 -- fold them down first)
        x *= m2[None, :]
        h = x.sum(axis=1, dtype=np.uint64)
    del padded, x
    _, first, inverse, mult_counts = np.unique(h, return_index=True, return_inverse=True, return_counts=True)
    for c0 in range(0, n, 1 << 20):                               # every read equals the first read of its hash group
        c1 = min(n, c0 + (1 << 20))
        if not bool((reads[c0:c1] == reads[first[inverse[c0:c1]]]).all()):
            return None
    nu = len(first)
    d_ureads = torch.from_numpy(np.ascontiguousarray(reads[first]).reshape(-1)).to(dev)
    d_uoff = torch.arange(nu + 1, dtype=torch.int64, device=dev) * L
    d_uw = torch.from_numpy(mult_counts.astype(np.uint32).view(np.int32)).to(dev)
    host_dedup_s = time.perf_counter() - t_d
    ua = torch.empty((nu, stride), dtype=torch.uint8, device=dev)    # (own buffers: the checks read the timed batch's outputs)
    uf = torch.empty((nu, stride), dtype=torch.uint8, device=dev)
    ur = torch.empty((nu, 32), dtype=torch.uint8, device=dev)
    d_ucounts = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)

    def dedup_step():
        job.al.align_device(nu, d_ureads.data_ptr(), d_uoff.data_ptr(), ua.data_ptr(), uf.data_ptr(), ur.data_ptr(), stride, L, stream=stream)
        d_ucounts.zero_()
        C.accumulate_device(ctx, layout, nu, ua.data_ptr(), uf.data_ptr(), stride, ur.data_ptr(), d_ucounts.data_ptr(),
                            d_weights=d_uw.data_ptr(), min_matches=job.min_matches, stream=stream)
    dedup_step()
    torch.cuda.synchronize()
    t_u = time.perf_counter()
    for _ in range(steps):
        dedup_step()
    torch.cuda.synchronize()
    dt_u = time.perf_counter() - t_u
    # the weighted tensor of the unique reads = the tensor of all reads (but for the scalar that counts ALIGNMENTS, not reads)
    tu, ta = d_ucounts.clone(), job.d_counts.clone()
    for t_ in (tu, ta):
        t_.view(k, -1)[:, layout.scalar_offset("alignments_counted")] = 0
    same_tensor = bool(torch.equal(tu, ta))
    out = {"unique_reads": int(nu), "reads_per_s": n * steps / dt_u, "ms_per_step": 1e3 * dt_u / steps,
           "count_tensor_equals_dedup_off": same_tensor, "host_dedup_seconds_not_timed": host_dedup_s,
           "note": "the batch's unique reads aligned once, multiplicities as weights of the count pass; reads/s counts every read of the batch"}
    del d_ureads, d_uoff, d_uw, ua, uf, ur
    torch.cuda.empty_cache()
    return out


def _e2e_sharded_leg(path, n_reads, ctx, L, matrix, dev, repeat=2):
    """N ranks, one FASTQ file: pipeline.quantify_fastq(shard_across_ranks=True) -- every rank uploads, frames and de-duplicates ITS byte range
    of the text, the ranks all-gather their unique reads, each aligns its range of the run's list, the count tensors are all-reduced.
    A collective: every rank calls it.  -> (rank 0) seconds, reads/s, what every rank uploaded."""
    import torch
    import torch.distributed as dist
    from types import SimpleNamespace
    from crispresso2_amd import pipeline, refs as RF, synth
    amp, g, inc = synth.amplicon_setup(L)
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=GO, needleman_wunsch_gap_extend=GE,
                           ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=MIN_ALN_SCORE)
    runs, last = [], None
    for rep in range(repeat + 1):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipeline.quantify_fastq(path, {"Reference": ref}, ["Reference"], matrix, args, ctx=ctx, shard_across_ranks=True)
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        runs.append(float(t.item()))
        c = res.per_ref["Reference"]
        last = (res.stats["N_TOT_READS"], res.stats["N_TOTAL"], c["counts_total"], c["counts_modified"], getattr(res, "ingest_route", "host"), getattr(res, "shard_ingest", None))
        del res
    shards = [None] * dist.get_world_size()
    dist.all_gather_object(shards, last[5])
    dt = min(runs[1:])
    return {"reads": n_reads, "seconds": dt, "reads_per_s": n_reads / dt, "seconds_all_runs": runs, "ingest_route": last[4],
            "tallies": dict(zip(("N_TOT_READS", "N_TOTAL", "counts_total", "modified"), last[:4])),
            "per_rank": shards,
            "note": "one FASTQ file, N ranks: each rank uploads + frames + de-duplicates its byte range (shard_bytes of text_bytes), the ranks' unique reads are "
                    "all-gathered and reconciled, each aligns its range of the run's unique reads, one all-reduce of the count tensor; max over ranks, best of %d "
                    "runs after a warm-up" % repeat}


def _e2e_prepare(reads, workers, bgzf=True):
    """(host, before HIP) the headline's reads as a FASTQ file in /dev/shm -- plain, and BGZF-compressed by a process pool -- sized down if
    the file system is too small.  -> dict(dir, plain, bgzf, reads, bytes_plain, bytes_bgzf, write_s) or {"skipped": reason}"""
    from crispresso2_amd import synth
    n, L = reads.shape
    per = 2 * L + 16
    want = os.environ.get("C2_BENCH_E2E_DIR")
    cands = [want] if want else ["/dev/shm", tempfile.gettempdir()]
    for base in cands:
        try:
            free = shutil.disk_usage(base).free
        except OSError:
            continue
        m = n
        if free < 2.2 * per * n:                                  # (the FASTQ, its BGZF and gzip copies, and the result tables of the with_all_tables legs)
            m = int(free / (2.2 * per))
            m -= m % 1000
        if m < min(n, 100_000):
            continue
        d = None
        try:
            d = tempfile.mkdtemp(prefix="c2bench_", dir=base)
            t0 = time.perf_counter()
            plain, bgzf_path = os.path.join(d, "reads.fastq"), os.path.join(d, "reads.bgzf.fastq.gz")
            b1 = synth.write_fastq(reads[:m], plain)
            t1 = time.perf_counter()
            b2 = synth.write_bgzf(plain, bgzf_path, workers=workers) if bgzf else 0
            t2 = time.perf_counter()
            # an ordinary single-member .gz of the same text (what `gzip -6` gives: ONE deflate stream; the reference opens it with gzip.open, CRISPRessoCORE.py:1820-1823)
            gz_path = os.path.join(d, "reads.fastq.gz")
            b3 = synth.write_gzip_member(plain, gz_path, workers=workers, level=6) if bgzf else 0
            return dict(dir=d, plain=plain, bgzf=bgzf_path if bgzf else None, gzip=gz_path if bgzf else None, reads=m, bytes_plain=b1, bytes_bgzf=b2, bytes_gzip=b3,
                        write_plain_s=t1 - t0, write_bgzf_s=t2 - t1, write_gzip_s=time.perf_counter() - t2)
        except OSError:
            if d:
                shutil.rmtree(d, ignore_errors=True)
            continue
    return {"skipped": "no file system with room for the FASTQ file (%s)" % ", ".join(cands)}


def _e2e_leg(files, ctx, L, matrix, repeat=2):
    """FASTQ -> count tensors with the wall time of every stage (pipeline.quantify_fastq), the plain file and the BGZF file."""
    from types import SimpleNamespace
    from crispresso2_amd import pipeline, refs as RF, synth, _native
    gz_inflate = None
    amp, g, inc = synth.amplicon_setup(L)
    args = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=GO, needleman_wunsch_gap_extend=GE,
                           ignore_deletions=False, ignore_insertions=False, ignore_substitutions=False,
                           assign_ambiguous_alignments_to_first_reference=False, expand_ambiguous_alignments=False, discard_indel_reads=False)
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=MIN_ALN_SCORE)
    out = {"reads": files["reads"], "file_bytes": files["bytes_plain"], "file_bytes_bgzf": files["bytes_bgzf"], "file_bytes_gzip": files.get("bytes_gzip"),
           "file_system": os.path.dirname(files["dir"]),
           "write_seconds_not_timed": {"plain": files["write_plain_s"], "bgzf": files["write_bgzf_s"], "gzip": files.get("write_gzip_s")}}
    tallies = {}
    kinds = [("plain", files["plain"]), ("plain_host_parser", files["plain"]), ("bgzf", files["bgzf"]), ("bgzf_host_parser", files["bgzf"])]
    if files.get("gzip"):
        kinds.append(("gzip", files["gzip"]))                        # single member: ONE deflate stream, inflated by all host threads (c2_gz_parallel.h), then the device frames the text
    for kind, path in kinds:
        # plain: the text is uploaded as it is and framed + de-duplicated by the c2_fq_* kernels (fastq_device); bgzf: the host inflates
        # (all usable threads) and the same kernels frame the text it then holds; *_host_parser: the same files through the native host
        # parser (what text with carriage returns uses), for comparison
        os.environ["C2_FQ_INGEST"] = "host" if kind.endswith("_host_parser") else "auto"
        runs = []
        for rep in range(repeat + 2):                                # the first run is the warm-up (context, allocations, page cache); the last one
            tm = {} if rep == repeat + 1 else None                    # collects the stage times (a device synchronisation per stage: not a timed run)
            t0 = time.perf_counter()
            res = pipeline.quantify_fastq(path, {"Reference": ref}, ["Reference"], matrix, args, ctx=ctx, timings=tm)
            runs.append((time.perf_counter() - t0, tm))
            c = res.per_ref["Reference"]
            tallies[kind] = (res.stats["N_TOT_READS"], res.stats["N_TOTAL"], c["counts_total"], c["counts_modified"], c["counts_insertion"],
                             c["counts_deletion"], c["counts_substitution"])
            uniq = res.stats["N_COMPUTED_ALN"] + res.stats["N_COMPUTED_NOTALN"]
            route = getattr(res, "ingest_route", "host")
            del res
            time.sleep(0.3)                                          # (the run's buffers are unmapped by helper threads: let them finish)
        dt = min(r[0] for r in runs[1:repeat + 1])
        tm = runs[-1][1]
        # what crosses the host link in this leg: the device-ingest routes upload the TEXT (plain: as it lies in the file; bgzf: as the host inflated
        # it); the host-parser routes upload the unique reads' arena + offsets + multiplicities.  Results coming back are kilobytes.
        if kind.endswith("_host_parser"):
            link_bytes = None
        elif kind == "gzip":
            link_bytes = files["bytes_plain"]                         # (the host inflates; the text crosses the link)
            gz_inflate = _native.gz_parallel_last()                   # (of the last run: segments, seconds of the search and the two passes)
        else:
            link_bytes = files["bytes_plain"]
        out[kind] = {"seconds": dt, "reads_per_s": files["reads"] / dt, "seconds_all_runs": [r[0] for r in runs], "stage_seconds": tm,
                     "link_bytes": link_bytes, "link_gbs": None if link_bytes is None else link_bytes / dt / 1e9,
                     "frac_of_link_peak": None if link_bytes is None else link_bytes / dt / 1e9 / LINK_PEAK_GBS,
                     "link_peak_gbs": LINK_PEAK_GBS,
                     "link_note": "text bytes uploaded / the whole run's seconds / 63 GB/s (PCIe Gen5 x16 spec, MI355X_MICROARCH.md): the run is one upload with the "
                                  "kernels underneath it, so this fraction is the leg's roofline",
                     "stage_seconds_note": "from one more run with a device synchronisation after every stage (the last of seconds_all_runs); the "
                                           "timed runs have none", "unique_reads": uniq, "ingest_route": route}
    os.environ.pop("C2_FQ_INGEST", None)
    # ---- a run that ends where the reference's ends: FASTQ -> every result table of tables.write_tables ON DISK (the allele frequency table --
    # one line per aligned unique read, sorted -- and the alleles around the guide's cut among them: rows, sort, grouping and text on the device).
    # `with_all_tables`: the allele table as the reference leaves it, Alleles_frequency_table.zip (CRISPRessoCORE.py:4531-4533: deflated, the .txt
    # removed) -- the chunks that come off the device are deflated on all host threads into ONE stream, only compressed bytes are written;
    # `with_all_tables_txt`: the same run writing the 2 GB .txt instead (round 4's leg).
    from crispresso2_amd import tables
    ref_t = dict(ref)
    ref_t["sgRNA_orig_sequences"] = [amp[L // 2 - 16:L // 2 + 4]]
    out_dir = os.path.join(files["dir"], "tables")
    txt_digest = None
    for leg, as_zip in (("with_all_tables_txt", False), ("with_all_tables", True)):
        runs = []
        for rep in range(repeat + 1):
            shutil.rmtree(out_dir, ignore_errors=True)
            tt = {}
            t0 = time.perf_counter()
            res = pipeline.quantify_fastq(files["plain"], {"Reference": ref_t}, ["Reference"], matrix, args, ctx=ctx)
            t1 = time.perf_counter()
            written = tables.write_tables(res, {"Reference": ref_t}, ["Reference"], out_dir, timings=tt, allele_table_zip=as_zip)
            t2 = time.perf_counter()
            runs.append((t2 - t0, t1 - t0, tt))
            rows = res.allele_table().n_rows
            table_bytes = {w: os.path.getsize(os.path.join(out_dir, w)) for w in written}
            res.allele_table().close()
            del res
            time.sleep(0.3)
        best = min(runs[1:], key=lambda r: r[0])
        entry = {"seconds": best[0], "reads_per_s": files["reads"] / best[0], "seconds_all_runs": [r[0] for r in runs],
                 "quantify_fastq_seconds": best[1], "write_tables_seconds": best[0] - best[1], "write_tables_stage_seconds": best[2],
                 "files_written": len(table_bytes), "bytes_written": int(sum(table_bytes.values())), "allele_table_rows": rows,
                 "alleles_around_cut_bytes": max([v for w, v in table_bytes.items() if "_around_" in w], default=None)}
        import hashlib
        import zipfile
        if not as_zip:
            with open(os.path.join(out_dir, "Alleles_frequency_table.txt"), "rb") as fh:
                head_lines = fh.read(1 << 16).split(b"\n")[:3]
                fh.seek(0)
                hh, nb_txt = hashlib.blake2b(digest_size=16), 0
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    hh.update(blk)
                    nb_txt += len(blk)
            txt_digest = (hh.hexdigest(), nb_txt)
            aw = best[2].get("allele_table_write") or 0.0
            entry.update({"allele_table_bytes": table_bytes.get("Alleles_frequency_table.txt"),
                          "allele_table_write_gbs": (table_bytes.get("Alleles_frequency_table.txt", 0) / aw / 1e9) if aw > 0 else None,
                          "second_line_of_the_allele_table_starts": head_lines[1][:40].decode("ascii", "replace") if len(head_lines) > 1 else None,
                          "note": "FASTQ (plain, device ingest) -> count tensors -> every .txt of tables.write_tables on the same file system; best of %d runs after a "
                                  "warm-up.  allele_table_write_gbs: bytes of the .txt / seconds of its stage (device text -> pinned chunks -> pwrite on all threads): "
                                  "host memory bandwidth into the page cache, not the link (2 GB at 63 GB/s would be 0.03 s)" % repeat})
        else:
            zp = os.path.join(out_dir, "Alleles_frequency_table.zip")
            with zipfile.ZipFile(zp) as z:                           # the reference's own reader gives the text back: compared with the .txt leg's bytes
                info = z.getinfo("Alleles_frequency_table.txt")
                hh, nb_z = hashlib.blake2b(digest_size=16), 0
                with z.open(info) as fh:
                    for blk in iter(lambda: fh.read(1 << 24), b""):
                        hh.update(blk)
                        nb_z += len(blk)
            aw = best[2].get("allele_table_write") or 0.0
            entry.update({"allele_table_zip_bytes": table_bytes.get("Alleles_frequency_table.zip"), "allele_table_text_bytes": info.file_size,
                          "zip_member_equals_the_txt_legs_file": bool(txt_digest is not None and (hh.hexdigest(), nb_z) == txt_digest),
                          "allele_table_deflate_gbs": (info.file_size / aw / 1e9) if aw > 0 else None,
                          "note": "the same run ending as the reference's does: Alleles_frequency_table.zip (one member, deflated by all host threads into one "
                                  "stream -- zlib level 1, slices ended by sync flushes --, only compressed bytes written) and no .txt; Python's zipfile read the member back "
                                  "and its bytes equal the .txt leg's file; best of %d runs after a warm-up" % repeat})
        out[leg] = entry
    shutil.rmtree(out_dir, ignore_errors=True)
    out["reads_per_s"] = out["plain"]["reads_per_s"]
    out["stage_seconds"] = out["plain"]["stage_seconds"]
    out["plain_equals_bgzf"] = tallies["plain"] == tallies["bgzf"] == tallies["plain_host_parser"] == tallies["bgzf_host_parser"]
    if "gzip" in tallies:
        out["plain_equals_gzip"] = tallies["plain"] == tallies["gzip"]
        out["gzip"]["note"] = ("an ordinary single-member .gz (one deflate stream, level 6): cut into segments at block starts found by search, decoded once for "
                               "sizes and 32 KiB windows (c2_gz_parallel.h, c2_gzseg_open), then inflated segment by segment by all host threads straight into the "
                               "pinned upload buffers -- the text never lies in host memory as a whole; CRC-32 and ISIZE checked at the last segment; framed on "
                               "the device like the plain file's.  Whatever is not one clean member goes to libdeflate on one thread")
        out["gzip"]["inflate"] = gz_inflate
    out["tallies"] = dict(zip(("N_TOT_READS", "N_TOTAL", "counts_total", "modified", "with_insertion", "with_deletion", "with_substitution"),
                              tallies["plain"]))
    out["note"] = ("pipeline.quantify_fastq on the headline's reads as a FASTQ file (qualities 'I'), page cache warm: ingest + exact "
                   "de-duplication (plain / bgzf: on the device, bgzf after the host inflated it; *_host_parser: the native host parser), seed test, alignments of the unique reads, selection, reverse-complement merge, count kernel; best of %d "
                   "runs after a warm-up; reads/s counts every read of the file" % repeat)
    return out


def _host_batch_leg(ctx, reads, L, matrix, n_reads=2_000_000, repeat=2):
    """The boundary handing over HOST buffers (c2_align_classify_batch_host, the call a C caller without device memory makes): the reads in the caller's
    pageable arrays, aligned strings + records back into the caller's pageable arrays -- chunks through pinned staging, copies both ways overlapped
    with the launch chains.  This is the PCIe-inclusive rate; it is never `value`."""
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    amp, g, inc = synth.amplicon_setup(L)
    al = BatchAligner([amp], [g], [inc], matrix, GO, GE, ctx=ctx)
    n = min(n_reads, reads.shape[0])
    arena = np.ascontiguousarray(reads[:n]).reshape(-1)
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    al.align((arena[:L * 4096], offsets[:4097]))                      # warm-up: staging buffers, row tables
    runs = []
    for _ in range(repeat):
        t0 = time.perf_counter()
        res = al.align((arena, offsets))
        runs.append(time.perf_counter() - t0)
        stride = res.aln_read.shape[1]
        ok = bool((res.records["status"] == 0).all())
        del res
    dt = min(runs)
    link = n * (L + 8) + n * (2 * stride + 32)
    return {"reads": n, "seconds": dt, "reads_per_s": n / dt, "seconds_all_runs": runs, "all_status_ok": ok,
            "link_bytes": link, "link_bytes_per_read": link / n, "link_gbs": link / dt / 1e9, "frac_of_link_peak": link / dt / 1e9 / LINK_PEAK_GBS,
            "link_peak_gbs": LINK_PEAK_GBS,
            "note": "c2_align_classify_batch_host: reads from and results into the caller's PAGEABLE numpy arrays (in: read bytes + offsets; out: two aligned "
                    "strings of `aln_stride` bytes + a 32-byte record per read), including the allocation of the 2 x n x stride output arrays' pages; "
                    "the PCIe-inclusive rate of the hot path -- bench.py's `value` has the inputs resident in HBM"}


def _compare_with_reference(legs, d_aln_read, d_aln_ref, rec, k, all_refs):
    """every alignment the reference's compiled code computed for `legs` (oracle/cpu_baseline.py: blake2b digests of the two aligned strings,
    the classifier's three window counts of the best alignment) against the device's output for the same reads -> (reads compared, identical)"""
    from oracle import cpu_baseline as cb
    compared = identical = 0
    CH = 65536
    for leg in legs:
        a0, cnt_r = leg["first_read"], leg["n_reads"]
        t0_, nt = a0 * (k if all_refs else 1), cnt_r * (k if all_refs else 1)
        dig = np.zeros(nt, dtype=np.uint64)
        for c0 in range(0, nt, CH):
            c1 = min(nt, c0 + CH)
            ar = d_aln_read[t0_ + c0:t0_ + c1].cpu().numpy()
            af = d_aln_ref[t0_ + c0:t0_ + c1].cpu().numpy()
            Ts = rec["aln_len"][t0_ + c0:t0_ + c1]
            for q in range(c1 - c0):
                T = int(Ts[q])
                dig[c0 + q] = cb.digest(ar[q, :T].tobytes(), af[q, :T].tobytes())
        same = dig == leg["digests"]
        rr = np.arange(cnt_r)
        tb = t0_ + (rr * k + leg["best_ref"].astype(np.int64) if all_refs else rr)
        same_cnt = ((rec["insertion_n"][tb] == leg["counts"][:, 0]) & (rec["deletion_n"][tb] == leg["counts"][:, 1]) &
                    (rec["substitution_n"][tb] == leg["counts"][:, 2]))
        if all_refs:
            same = same.reshape(cnt_r, k).all(axis=1)
        compared += cnt_r
        identical += int((same & same_cnt).sum())
    return compared, identical


def _r(x, sig=5):
    """a float with `sig` significant digits (the short line has no room for 17)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    return x


SHORT_LINE_LIMIT = 4096


def short_line(d, detail_path=None):
    """The ONE line on stdout: what the contract names, as scalars, under SHORT_LINE_LIMIT bytes; `d` is the full record (bench_detail.json)."""
    cfg, ck, rf, vl, cb = d["config"], d["checks"], d["roofline"], d["valu"], d.get("cpu_baseline")
    oc = d.get("other_configs") or {}
    e2e = d.get("e2e") or {}

    def leg(name, key):
        x = e2e.get(name)
        return _r(x.get(key)) if isinstance(x, dict) else None
    short = {
        "metric": d["metric"], "value": _r(d["value"], 6), "unit": d["unit"], "n_gpus": d["n_gpus"], "ranks_seen": d["ranks_seen"],
        "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": _r(d["ms_per_step"], 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": d["dtype"], "data": "synthetic", "collective_backend": d["collective_backend"],
        "config": {
            "workload": cfg["workload"], "baseline_config": cfg["baseline_config"], "reads_per_gpu_per_step": cfg["reads_per_gpu_per_step"],
            "alignments_per_gpu_per_step": cfg["alignments_per_gpu_per_step"], "n_amplicons": cfg["n_amplicons"],
            "finished_by_partition": cfg["finished_by_partition"], "packed_int16_share": _r(cfg["packed_int16_share"], 4),
            "int32_chain_reads_per_s": _r(cfg["int32_chain_reads_per_s"]),
            "robust_fanc_shaped_reads_per_s": _r(cfg["robust_fanc_shaped_reads_per_s"]),
            "robust_lengths_200_to_L_reads_per_s": _r(cfg["robust_lengths_200_to_L_reads_per_s"]),
            "robust_unrelated_10_percent_reads_per_s": _r(cfg["robust_unrelated_10_percent_reads_per_s"]),
            "robust_full_plane_floor_reads_per_s": _r(cfg["robust_full_plane_floor_reads_per_s"]),
            "other_configs_reads_per_s": {c_: _r(e_.get("reads_per_s")) for c_, e_ in oc.items() if isinstance(e_, dict) and "reads_per_s" in e_} or None,
            "other_configs_alignments_per_s": {c_: _r(e_.get("alignments_per_s")) for c_, e_ in oc.items() if isinstance(e_, dict) and e_.get("n_amplicons", 1) > 1 and "alignments_per_s" in e_ and e_.get("alignments_per_gpu_per_step") != e_.get("reads_per_gpu_per_step")} or None,
            "other_configs_chain_equals_full_plane": {c_: e_.get("chain_equals_full_plane") for c_, e_ in oc.items() if isinstance(e_, dict) and "chain_equals_full_plane" in e_} or None,
            "other_configs_reference_identical": cfg["other_configs_reference_identical"],
            "chain_equals_full_plane_n": cfg["chain_equals_full_plane_n"], "reference_identical_n": cfg["reference_identical_n"],
            "reference_compared_n": cfg["reference_compared_n"],
            "e2e_fastq_to_tensors_reads_per_s": _r(cfg["e2e_fastq_to_tensors_reads_per_s"]),
            "e2e_frac_of_link_peak": {"plain": leg("plain", "frac_of_link_peak"), "gzip": leg("gzip", "frac_of_link_peak")} if e2e and "plain" in e2e else None,
            "e2e_gzip_single_member_reads_per_s": _r(cfg["e2e_gzip_single_member_reads_per_s"]),
            "e2e_gzip_seconds": leg("gzip", "seconds"),
            "e2e_fastq_to_all_tables_seconds": _r(cfg["e2e_fastq_to_all_tables_seconds"]),
            "host_batch_pcie_inclusive_reads_per_s": _r(cfg["host_batch_pcie_inclusive_reads_per_s"])},
        "step_breakdown_ms": {k_: _r(v_, 4) for k_, v_ in d["step_breakdown_ms"].items() if k_ != "note"},
        "roofline": {"bound": rf["bound"], "achieved": _r(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"], "frac": _r(rf["frac"], 4),
                     "traffic": _r(rf["traffic"]), "kernel": rf["kernel"], "avg_launch_ms": _r(rf["avg_launch_ms"]),
                     "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"], "launches": rf["launches"],
                     "chain_avg_ms": _r(rf["chain_avg_ms"]), "traffic_source": rf.get("traffic_file")},
        "valu": {"frac_of_simd32_peak": _r(vl.get("frac_of_simd32_peak"), 4), "frac_of_measured_issue": _r(vl.get("frac_of_measured_issue"), 4),
                 "wave_instr_per_alignment": _r(vl.get("wave_instr_per_alignment")), "gcups": _r(vl.get("gcups"))},
        "cpu_baseline": None if not cb else {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                             "best_procs": cb.get("best_procs"), "sample": cb.get("sample_short") or str(cb.get("sample"))[:160]},
        "speedup_vs_cpu_baseline": _r(d.get("speedup_vs_cpu_baseline")),
        "checks": {k_: (_r(v_) if isinstance(v_, float) else v_) for k_, v_ in ck.items()},
        "per_rank_reads_aligned": d.get("per_rank_reads_aligned"),
        "reads_aligned_all_gpus": [c_["reads_aligned_all_gpus"] for c_ in d["counts"]],
        "side_legs_ok": d["side_legs_ok"], "side_legs_note": (d.get("side_legs_note") or "")[:200] or None,
        "detail": None if detail_path is None else (os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) and detail_path.startswith(ROOT) else detail_path),
    }
    line = json.dumps(short, separators=(",", ":"))
    if len(line) >= SHORT_LINE_LIMIT:                                  # (cannot happen with the fields above; if a text grew, the optional groups go first)
        for k_ in ("step_breakdown_ms", "side_legs_note", "per_rank_reads_aligned", "reads_aligned_all_gpus"):
            short.pop(k_, None)
        short["config"]["workload"] = short["config"]["workload"][:200]
        line = json.dumps(short, separators=(",", ":"))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5], help="BASELINE.json workload shape (3 = the headline)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (0 = the configuration's size)")
    ap.add_argument("--len", type=int, default=0, dest="L", help="read and amplicon length (0 = the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall seconds of the CPU baseline's sweep over pool sizes")
    ap.add_argument("--cpu-long-seconds", type=float, default=60.0, help="... then the best pool size once more for this long (0: skip); its rate is cpu_baseline.value")
    ap.add_argument("--ref-check-reads", type=int, default=100_000,
                    help="reads of each OTHER configuration (other_configs) aligned by the reference's compiled code before the timed region and compared with the device's output")
    ap.add_argument("--workers", type=int, default=0, help="processes generating the synthetic reads (0 = auto; 1 = no fork, for profiler runs)")
    ap.add_argument("--band", type=int, default=-1, help="pointer-plane band: -1 auto, 0 off, n lanes each side")
    ap.add_argument("--band-wgs", type=int, default=0, help="target workgroups per CU for the automatic band")
    ap.add_argument("--kernel", choices=["auto", "band", "full", "diag1", "diag2", "diag4"], default="auto",
                    help="kernel chain (auto = diagonal-band kernels with certificate: 8 alignments per wavefront in int16 pairs -> 2 -> 1; "
                         "diag4 = the 32-bit chain 4 -> 2 -> 1)")
    ap.add_argument("--check", type=int, default=300, help="reads compared with the C oracle after the timed region (0 = no checks at all)")
    ap.add_argument("--no-full-plane-check", action="store_true", help="skip the chain-vs-full-plane comparison of every alignment")
    ap.add_argument("--no-dedup-leg", action="store_true", help="skip the dedup-on measurement after the timed region (single-amplicon configurations)")
    ap.add_argument("--extras", choices=["auto", "on", "off"], default="auto",
                    help="the legs after the headline: int32 chain, the other BASELINE configurations, FASTQ -> tensors (auto: on for the "
                         "plain default run -- no --config / --reads / --len / --kernel)")
    ap.add_argument("--no-extras", dest="extras", action="store_const", const="off")
    ap.add_argument("--extra-reads", type=int, default=0, help="reads per GPU of the other configurations and of the FASTQ leg (0 = full size)")
    ap.add_argument("--extra-steps", type=int, default=3)
    ap.add_argument("--overlap-count", action="store_true",
                    help="run the count pass of batch k on a second stream while batch k+1 is aligned into a second set of output buffers "
                         "(measured on MI355X, profiles/r02/README.md: no gain -- the persistent workgroups of the launch chain leave the "
                         "count kernel nothing to run on, and it slows them; the default keeps one stream)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    L = args.L or CONFIG_DEFAULTS[args.config][0]
    n = args.reads or CONFIG_DEFAULTS[args.config][1]
    matrix_path = os.path.join(ROOT, "crispresso2_amd", "EDNAFULL")
    extras = args.extras == "on" or (args.extras == "auto" and args.config == 3 and not args.reads and not args.L and args.kernel == "auto"
                                     and not args.overlap_count)

    # ---------- everything that fork()s happens before this process touches HIP ----------
    ncpu = os.cpu_count() or 1
    workers = args.workers if args.workers > 0 else max(1, min(32, ncpu // max(world, 1)))
    t0 = time.perf_counter()
    wl = build_workload(args.config, L, n, rank, workers)
    t_gen = time.perf_counter() - t0
    reads, refs, all_refs = wl["reads"], wl["refs"], wl["all_refs"]
    k = len(refs)
    n_tasks = n * (k if all_refs else 1)
    n_u = min(n, 1_000_000)
    unique_fraction = float(len(np.unique(reads[:n_u].view([("r", "V%d" % L)]))) / n_u)
    cpu_baseline, cpu_legs = None, []
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline as cb
        cpu_baseline, cpu_legs = cb.run(reads, refs, matrix_path, GO, GE, ref_ids=wl["ref_ids"], all_refs=all_refs, cores=ncpu,
                                        target_seconds=args.cpu_seconds, long_seconds=args.cpu_long_seconds)
    other_wl, other_ref, e2e_files = {}, {}, None
    robust_wl, robust_ref, t_gen_robust = {}, {}, 0.0
    if extras:
        t0 = time.perf_counter()
        for cfg in (2, 4, 5):
            if cfg != args.config:
                Lc, nc = CONFIG_DEFAULTS[cfg]
                other_wl[cfg] = (Lc, build_workload(cfg, Lc, min(nc, args.extra_reads) if args.extra_reads else nc, rank, workers))
        t_gen_other = time.perf_counter() - t0
        if args.config == 3 and not all_refs:
            t0 = time.perf_counter()
            robust_wl = build_robust_workloads(L, min(n, args.extra_reads) if args.extra_reads else n, rank, workers)
            t_gen_robust = time.perf_counter() - t0
        if rank == 0 and world == 1 and not args.no_cpu_baseline and args.ref_check_reads > 0 and args.check > 0:
            # the first reads of every other configuration through the reference's compiled code (a checker, not a timing; before HIP: it forks)
            from oracle import cpu_baseline as cb
            procs_ = int((cpu_baseline or {}).get("best_procs") or min(ncpu, 16))
            for cfg, (Lc, wlc) in sorted(other_wl.items()):
                m_ = min(len(wlc["reads"]), args.ref_check_reads)
                try:
                    other_ref[cfg] = cb.reference_slice(wlc["reads"][:m_], wlc["refs"], matrix_path, GO, GE,
                                                        ref_ids=None if wlc["ref_ids"] is None else wlc["ref_ids"][:m_], all_refs=wlc["all_refs"], procs=procs_)
                except Exception as e:
                    other_ref[cfg] = ({"error": repr(e)}, None)
            for name, wlc in robust_wl.items():
                m_ = min(len(wlc["reads"]), args.ref_check_reads)
                try:
                    robust_ref[name] = cb.reference_slice(wlc["reads"][:m_], wlc["refs"], matrix_path, GO, GE, procs=procs_)
                except Exception as e:
                    robust_ref[name] = ({"error": repr(e)}, None)
        if rank == 0 and not all_refs and wl["ref_ids"] is None:
            try:                                                     # (N ranks: one plain file, read by all of them -- the sharded FASTQ leg)
                e2e_files = _e2e_prepare(reads[:args.extra_reads] if args.extra_reads else reads, min(64, max(workers, ncpu // 4)), bgzf=world == 1)
            except Exception as e:                                   # the headline must not die of a side leg
                e2e_files = {"skipped": "writing the FASTQ files failed: %r" % (e,)}

    import torch
    import torch.distributed as dist
    backend = os.environ.get("C2_BENCH_BACKEND", "nccl")               # ("gloo": the CPU test of the multi-rank plumbing, tests/test_bench_on_emulator.py)
    dev_index = local_rank
    if world > 1 and backend != "nccl" and torch.cuda.is_available() and torch.cuda.device_count() <= local_rank:
        dev_index = local_rank % torch.cuda.device_count()              # (plumbing runs only: several gloo ranks sharing the box's GPUs)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ranks_seen, rccl_version = 1, None
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        ranks_seen = dist.get_world_size()
    try:
        v = torch.cuda.nccl.version()
        rccl_version = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass

    from crispresso2_amd import CRISPResso2Align as A, _native
    from crispresso2_amd import counts as C
    m = A.read_matrix(matrix_path)
    ctx = _native.Context(dev_index)
    ctx.set_band(args.band, args.band_wgs)
    job = Job(ctx, wl, L, m, dev, world, kernel=args.kernel, overlap_count=args.overlap_count)
    stride, Lmax, layout = job.stride, job.Lmax, job.layout
    tm = job.timed(args.warmup, args.steps)
    dt, kernel_ms, first_ms, launches = tm["dt"], tm["kernel_ms"], tm["first_ms"], tm["launches"]
    per_rank_aligned = None
    if world > 1:                                                   # (the count tensor of the step is the all-reduced one: every rank's own share next to it)
        per_rank_aligned = [None] * world
        dist.all_gather_object(per_rank_aligned, job.local_reads_aligned())
    tiers = ctx.tier_info()
    score_info = ctx.score_stage_info() if hasattr(ctx, "score_stage_info") else (False, 0, 0)     # (of the timed batch: later launches overwrite it)
    part_info = ctx.partition_info() if hasattr(ctx, "partition_info") else None
    packed_share = None                                           # share of the batch's alignments the packed (int16 pair) kernels finished
    try:
        left_, unpaired_ = ctx.tier_info_ex()
        if args.kernel == "auto" and left_ and not os.environ.get("C2_NO_PACKED_FILL"):
            # what the int16 kernels THEMSELVES finished: a tier's tasks minus those it could not pair (they run its 32-bit twin) minus those it hands on
            tier_in = [n_tasks] + list(left_[:-1])
            # (... minus the main-diagonal reads the partition finished without any fill: config.finished_by_partition)
            by_part_ = (part_info or {}).get("finished_by_partition", 0) if (part_info and part_info.get("ran")) else 0
            packed_share = float(sum(tier_in[t] - unpaired_[t] - left_[t] for t in range(len(left_))) - by_part_) / float(n_tasks)
    except Exception:
        packed_share = None

    dedup_on = None
    if rank == 0 and world == 1 and not all_refs and wl["ref_ids"] is None and not args.no_dedup_leg:
        dedup_on = _dedup_on_leg(job, reads, args.steps)
    # ---------- algorithmic bytes of one launch, parity checks ----------
    while len(job.out_sets) > 1:                                  # (every set holds the same bytes: the checks read set 0)
        job.out_sets.pop()
    torch.cuda.empty_cache()
    d_aln_read, d_aln_ref, d_records = job.outputs
    d_rids = job.d_rids
    rec = d_records.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    ok_status = bool((rec["status"] == 0).all())
    aln_cols = int(rec["aln_len"].astype(np.int64).sum())
    bytes_in = n_tasks * (L + 8) + (0 if d_rids is None else 2 * n)
    bytes_out = 2 * aln_cols + 32 * n_tasks
    alg_bytes = bytes_in + bytes_out
    avg_launch_s = (kernel_ms / max(launches, 1)) / 1e3          # the whole launch chain of one batch
    avg_first_s = (first_ms / max(launches, 1)) / 1e3            # its first kernel: the one that sees every task
    ref_cells = np.array([(len(r[0]) + 1) * (L + 1) for r in refs], dtype=np.int64)      # cells of the reference's full matrix
    if all_refs:
        cells = int(n * ref_cells.sum())
    elif wl["ref_ids"] is not None:
        cells = int(ref_cells[wl["ref_ids"].astype(np.int64)].sum())
    else:
        cells = int(n * ref_cells[0])

    def task_ref(t):
        return (t % k) if all_refs else (int(wl["ref_ids"][t]) if wl["ref_ids"] is not None else 0)

    checks = {"all_status_ok": ok_status}
    if rank == 0 and args.check > 0:
        import oracle
        from oracle import cpu_baseline as cb
        # (1) a random sample against the C restatement
        rng = np.random.default_rng(12345)
        idx = rng.integers(0, n_tasks, args.check)
        t_idx = torch.from_numpy(idx).to(dev)
        a_r = d_aln_read[t_idx].cpu().numpy()
        a_f = d_aln_ref[t_idx].cpu().numpy()
        parity = True
        for j, t in enumerate(idx):
            r = task_ref(int(t))
            rd = reads[int(t) // k if all_refs else int(t)].tobytes().decode()
            st, s1, s2, mt, ln = oracle.global_align_raw(rd, refs[r][0], m, refs[r][1], GO, GE)
            T = int(rec["aln_len"][t])
            if st != 0 or T != ln or a_r[j, :T].tobytes().decode() != s1 or a_f[j, :T].tobytes().decode() != s2 or int(rec["matches"][t]) != mt:
                parity = False
                break
        checks["oracle_sample_identical"] = parity
        checks["oracle_sample"] = args.check
        # (2) EVERY alignment the CPU-baseline legs computed with the reference itself (strings by digest, the three window counts)
        if cpu_legs:
            compared, identical = _compare_with_reference(cpu_legs, d_aln_read, d_aln_ref, rec, k, all_refs)
            checks["reference_compared_n"] = compared
            checks["reference_identical_n"] = identical
            checks["reference_identical"] = bool(compared == identical)
    # (3) size-independent properties on EVERY alignment of the full-size batch
    if args.check > 0:
        checks["full_batch_properties_hold"] = job.properties_hold()
        eq_ = job.count_tensor_equals_without_hints()
        if eq_ is not None:
            checks["count_tensor_equals_without_hints"] = eq_
            h0_ = job.hint_sets[0].view(-1, 4)[:, 0]
            checks["hinted_tasks"] = int((h0_ < 0).sum().item())                       # (bit 31: a main-diagonal hint, c2_align_partition_kernel)
            checks["hinted_gapped_tasks"] = int(((h0_ >> 30) == 1).sum().item())        # (bit 30: a gapped hint, c2_group_epilogue)
    # (4) the launch chain's certificates, exhaustively: the same batch through the full-plane kernel alone
    if rank == 0 and args.check > 0 and not args.no_full_plane_check and args.kernel == "auto":
        equal_n, tf = job.chain_equals_full_plane()
        checks["chain_equals_full_plane_n"] = equal_n
        checks["chain_equals_full_plane"] = bool(equal_n == n_tasks)
        checks["full_plane_pass_s"] = tf

    info = ctx.launch_info(L)
    band = ctx.band_info(L)
    # (kernel names as rocprofv3 prints them: the packed kernels' second template argument says whether their sums are 32-bit adds)
    add32 = False
    if hasattr(ctx, "chain_info"):
        add32 = "packed fill with 32-bit adds" in ctx.chain_info(L, k)[0]
    pkv = ", true>" if add32 else ", false>"
    chain_names = {"auto": ["c2_align_diagp_kernel<8" + pkv, "c2_align_diagx_kernel<2>" if os.environ.get("C2_NO_PACKED_TIER2") else "c2_align_diagp_kernel<4" + pkv,
                            "c2_align_diag_kernel" if (os.environ.get("C2_NO_PACKED_TIER2") or os.environ.get("C2_NO_PACKED_TIER3")) else "c2_align_diagp_kernel<2" + pkv],
                   "diag4": ["c2_align_diagx_kernel<4>", "c2_align_diagx_kernel<2>", "c2_align_diag_kernel"],
                   "diag2": ["c2_align_diagx_kernel<2>", "c2_align_diag_kernel"], "diag1": ["c2_align_diag_kernel"]}
    if len(tiers) == 4 and args.kernel == "auto":                  # (behind the partition: the 40-diagonal tier between the first two)
        chain_names["auto"].insert(1, "c2_align_diagp_kernel<6" + pkv)
    if os.environ.get("C2_NO_PACKED_FILL"):
        chain_names["auto"] = chain_names["diag4"]
    chain = ((chain_names.get(args.kernel, ["c2_align_diag_kernel"])[-len(tiers):] if band["band_lanes"] < 0 else
              ["c2_align_classify_kernel<%d, true>" % info["rows_per_lane"]] if band["band_lanes"] > 0 else [])
             + ["c2_align_classify_kernel<%d, false>" % info["rows_per_lane"]])
    dominant = chain[0]
    # the score-only stage in front of the first band tier (c2_align_partition_kernel + c2_align_diags_kernel<8>): the tasks it finished never
    # reach the dominant kernel
    score_stage = None
    if score_info[0]:
        ran_, took_, fin_ = score_info
        if ran_:
            score_stage = {"kernels": ["c2_align_partition_kernel", "c2_align_diags_kernel<%s" % ("8" if os.environ.get("C2_SCORE_TIER_NA") == "8" else "16") + pkv], "tasks": took_, "finished": fin_,
                           "note": "reads as long as the amplicon whose last 32 columns differ from it in at most 6 places go through the packed fill without pointer bits; "
                                   "it finishes those whose optimal alignment is the main diagonal (gap-free predicate + certificate) and hands the rest "
                                   "to the first band tier; `partition`: tasks per class of c2_align_partition_kernel (score-only launch, 14-diagonal launch "
                                   "[opt-in], first / second / third band tier: by the diagonal the middle of the read lies on, a task goes straight to the "
                                   "tier whose band holds its path), so tasks_left_after_each_banded_launch[t] is the length of the list the launch behind "
                                   "tier t reads: what tier t left plus what the partition put there.  Class-0 reads on the amplicon's main diagonal with at most two "
                                   "differing bases never reach this launch: the partition writes their rows and records itself where the scoring proves the diagonal "
                                   "(config.finished_by_partition; c2_main_diagonal_certificate)"}
    # algorithmic bytes of the dominant kernel's launch: the reads and offsets of its tasks in; strings + record out for the tasks it finishes
    # (with the partition: the first tier sees its own class and what the launches in front could not finish; the list it leaves also holds the
    #  tasks the partition sent straight to later tiers)
    in_first = n_tasks - (score_stage["finished"] if score_stage else 0)
    left_first = tiers[0] if tiers else 0
    if part_info and part_info["ran"]:
        cls_, fin_p = part_info["classes"], part_info["finished"]
        # (class 0 counts the reads equal to their reference too: the partition finishes those itself, no launch sees them)
        in_first = cls_[2] + (cls_[0] - part_info.get("finished_by_partition", 0) - fin_p[0]) + (cls_[1] - fin_p[1])
        # (the list behind the first tier also holds what the partition sent straight to the launch that reads it: the 40-diagonal tier's class when
        #  the chain has that tier -- four band tiers --, else the 62-diagonal tier's, else the 128-diagonal one's)
        left_first = max(0, left_first - cls_[{4: 3, 3: 4, 2: 5}.get(len(tiers), 3)])
    done_first = in_first - left_first
    alg_first = int(bytes_in * (in_first / float(n_tasks))) + int(bytes_out * (done_first / float(n_tasks)))
    achieved_gbs = alg_first / avg_first_s / 1e9 if avg_first_s > 0 else 0.0
    # HBM bytes and instruction counts per alignment from the committed PMC passes of the newest round's build (separate rocprofv3
    # --pmc runs of this same script over 2 M reads; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 streaming
    # reads); null when the profile file is absent or the run is not the default one
    traffic = traffic_src = pmc_rel = None
    valu_per_aln = salu_per_aln = None
    pmc_path = next((q for q in (os.path.join(ROOT, "profiles", r_, "pmc_summary_default.json") for r_ in PROFILE_ROUNDS) if os.path.exists(q)), None)
    if pmc_path and args.config == 3 and L == 250 and args.kernel == "auto":
        with open(pmc_path) as fh:
            pj = json.load(fh)
        pmc = pj["kernels"].get(dominant)
        pmc_reads = float(pj.get("reads", 2.0e6))                    # (files of rounds 2-4 have no "reads" key: their passes ran over 2 M reads)
        pmc_rel = os.path.relpath(pmc_path, ROOT)
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 / pmc_reads * n      # bytes per launch of n alignments
            traffic_src = ("%s: (2*FETCH_SIZE + WRITE_SIZE) KB of %s over %d reads, scaled to the reads of one launch; includes the "
                           "kernel's pointer-word scratch plane" % (pmc_rel, dominant, int(pmc_reads)))
        if pmc and "SQ_INSTS_VALU" in pmc:
            valu_per_aln = pmc["SQ_INSTS_VALU"] / pmc_reads
            salu_per_aln = pmc.get("SQ_INSTS_SALU", 0.0) / pmc_reads
    tallies = job.tallies()
    selection = dict(zip(C.SELECT_STATS, job.d_selstats.cpu().numpy().tolist())) if all_refs else None
    valu = {"cells_per_s": cells / avg_launch_s if avg_launch_s > 0 else 0.0,
            "gcups": cells / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0,
            "note": "gcups = full-matrix cell updates the reference would perform per second of launch-chain time (the banded kernels "
                    "compute fewer); the fractions are of the dominant kernel: wave-instructions it issues per second against (a) what a "
                    "SIMD was measured to issue for this opcode mix and (b) the SIMD-32 two-cycle rate of MI355X_MICROARCH.md",
            "kernel": dominant,
            "wave_instr_per_alignment": valu_per_aln, "salu_instr_per_alignment": salu_per_aln,
            "cycles_per_instr_measured": VALU_MEASURED_CYCLES_PER_INSTR, "cycles_per_instr_full_rate_class": VALU_FULL_RATE_CYCLES_PER_INSTR,
            "cycles_per_instr_source": VALU_MEASURED_SOURCE,
            "peak_wave_instr_per_s_simd32": VALU_SIMD32_WAVE_INSTR_PER_S,
            "peak_lane_ops_per_s": VALU_SIMD32_WAVE_INSTR_PER_S * 64}
    if valu_per_aln and avg_first_s > 0:
        rate = valu_per_aln * n_tasks / avg_first_s
        valu["wave_instr_per_s"] = rate
        valu["frac_of_measured_issue"] = rate / (N_SIMD * CLOCK_HZ / VALU_MEASURED_CYCLES_PER_INSTR)
        valu["frac_of_simd32_peak"] = rate / VALU_SIMD32_WAVE_INSTR_PER_S

    # ---------- after the headline: the int32 chain on the same batch, the other BASELINE shapes, FASTQ -> tensors ----------
    int32_chain = other_configs = e2e = robustness = host_batch = None
    extras_done = threading.Event()
    extras_note = [None]

    def emit(file=None):
        total_reads = world * n * args.steps
        out = {
            "metric": "aligned+classified reads/sec (whole node), %d bp reads vs %d bp amplicon" % (L, L),
            "value": total_reads / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "collective_backend": (backend if world > 1 else None), "rccl_version": rccl_version,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # the arithmetic type of the dominant kernel's DP cells: int16 pairs in the packed kernels (proven range, c2_pk_eligible), int32 otherwise
            "dtype": ("int16x2 packed (host-proven range; int32 fallback)" if dominant.startswith("c2_align_diagp") else "int32"),
            "packed_fill_sums": ("v_add_u32 under a per-anti-diagonal bias (c2_pk_add32_ok)" if add32 else "v_pk_add_i16") if dominant.startswith("c2_align_diagp") else None,
            "dtype_note": "exact integer DP, not a precision trade: the packed kernels hold two alignments per 32-bit lane as int16 pairs only for "
                          "references whose DP values the host proves to fit (c2_pk_eligible); everything else runs the int32 kernels; the "
                          "results are bit-identical either way (checks.chain_equals_full_plane_n covers every alignment of the batch); "
                          "int32_chain is the same step with the 32-bit kernels only",
            "data": "synthetic",
            "config": {"workload": wl["text"] + ", EDNAFULL, gap_open -20, gap_extend -2, gap_incentive 1 at the cut",
                       "baseline_config": args.config,
                       "reads_per_gpu_per_step": n, "alignments_per_gpu_per_step": n_tasks, "read_len": L, "amplicon_len": Lmax, "n_amplicons": k,
                       "unique_read_fraction": unique_fraction, "unique_read_fraction_sample": n_u,
                       "rows_per_lane": info["rows_per_lane"], "lds_bytes_per_workgroup": info["lds_bytes"],
                       "workgroups_per_cu": info["workgroups_per_cu"], "compute_units": info["compute_units"],
                       "kernel_chain": (score_stage["kernels"] if score_stage else []) + chain,
                       "score_only_stage_tasks": None if not score_stage else score_stage["tasks"],
                       "score_only_stage_finished": None if not score_stage else score_stage["finished"],
                       # (reads on their amplicon's main diagonal -- byte-for-byte copies, one or two differing bases: c2_align_partition_kernel writes their rows
                       #  and records itself where the diagonal is provably the best path, c2_main_diagonal_certificate -- no launch fills a matrix for them)
                       "finished_by_partition": None if not (part_info and part_info.get("ran")) else part_info.get("finished_by_partition"),
                       "tasks_left_after_each_banded_launch": tiers,
                       "pointer_band_lanes": band["band_lanes"], "full_plane_fallback_tasks": band["fallback_tasks_last_launch"],
                       # (scalars the driver's record keeps: the all-int32 chain on the same batch, how much of the batch the packed kernels finished,
                       #  the checks against the reference's compiled code, the end-to-end legs)
                       "int32_chain_reads_per_s": None if not int32_chain else int32_chain["reads_per_s"],
                       "int32_chain_ms_per_step": None if not int32_chain else int32_chain["ms_per_step"],
                       "int32_chain_records_equal": None if not int32_chain else int32_chain["records_equal_the_packed_chain"],
                       "packed_int16_share": packed_share,
                       "chain_equals_full_plane_n": checks.get("chain_equals_full_plane_n"),
                       "reference_compared_n": checks.get("reference_compared_n"), "reference_identical_n": checks.get("reference_identical_n"),
                       "other_configs_reference_identical": None if not other_configs else {
                           c_: "%s/%s" % (e_.get("reference_identical_n"), e_.get("reference_compared_n")) for c_, e_ in other_configs.items() if isinstance(e_, dict) and "reference_compared_n" in e_},
                       "robust_fanc_shaped_reads_per_s": None if not robustness else (robustness.get("fanc_shaped") or {}).get("reads_per_s"),
                       "robust_lengths_200_to_L_reads_per_s": None if not robustness else (robustness.get("lengths_200_to_L") or {}).get("reads_per_s"),
                       "robust_unrelated_10_percent_reads_per_s": None if not robustness else (robustness.get("unrelated_10_percent") or {}).get("reads_per_s"),
                       "robust_full_plane_floor_reads_per_s": None if not robustness else (robustness.get("full_plane_floor") or {}).get("reads_per_s"),
                       "robust_worst_case": None if not robustness or "worst_case" not in robustness else "%s: %.1f M reads/s" % (
                           robustness["worst_case"]["leg"], robustness["worst_case"]["reads_per_s"] / 1e6),
                       "other_configs_reads_per_s": None if not other_configs else {c_: e_.get("reads_per_s") for c_, e_ in other_configs.items() if isinstance(e_, dict) and "reads_per_s" in e_},
                       "e2e_fastq_to_tensors_reads_per_s": (None if not e2e else e2e["plain"]["reads_per_s"] if "plain" in e2e else
                                                            (e2e.get("sharded") or {}).get("reads_per_s")),
                       "e2e_sharded_shard_bytes_per_rank": None if not e2e or "sharded" not in e2e or not e2e["sharded"].get("per_rank") else
                                                           [None if sh is None else sh.get("shard_bytes") for sh in e2e["sharded"]["per_rank"]],
                       "host_batch_pcie_inclusive_reads_per_s": None if not host_batch else host_batch.get("reads_per_s"),
                       "e2e_gzip_single_member_reads_per_s": None if not e2e or "gzip" not in e2e else e2e["gzip"].get("reads_per_s"),
                       "e2e_frac_of_link_peak": None if not e2e or "plain" not in e2e else e2e["plain"].get("frac_of_link_peak"),
                       "e2e_fastq_to_all_tables_seconds": None if not e2e or "with_all_tables" not in e2e else e2e["with_all_tables"]["seconds"],
                       "e2e_fastq_to_all_tables_reads_per_s": None if not e2e or "with_all_tables" not in e2e else e2e["with_all_tables"]["reads_per_s"]},
            "alignments_per_s": world * n_tasks * args.steps / dt,
            "step_breakdown_ms": {"align_chain": tm["align_ms"], "select_best": tm["select_ms"], "count_vectors_and_all_reduce": tm["count_ms"],
                                  "note": "rank 0, HIP events on the stream of each phase, mean over the timed steps" +
                                          ("" if not args.overlap_count else "; the count pass of batch k runs on a second stream while batch k+1 is aligned "
                                           "(two output buffer sets), so the phases overlap and do not add up to ms_per_step")},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_file": pmc_rel if traffic is not None else None,
                         "kernel": dominant, "avg_launch_ms": 1e3 * avg_first_s, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_first,
                         "algorithmic_bytes_per_read": alg_first / n_tasks,
                         "chain_avg_ms": 1e3 * avg_launch_s, "chain_algorithmic_bytes": alg_bytes,
                         "note": "integer DP: VALU-issue-bound by construction, HBM fraction is small (SURVEY 8d); see valu and profiles/*/README.md"},
            "valu": valu,
            "score_only_stage": score_stage,
            "partition": part_info,
            "int32_chain": int32_chain,
            "other_configs": other_configs,
            "robustness": robustness,
            "e2e": e2e,
            "host_batch_pcie_inclusive": host_batch,
            "dedup_on": dedup_on,
            "cpu_baseline": cpu_baseline,
            "checks": checks,
            "counts": [{"amplicon": r, "reads_aligned_all_gpus": tl["counts_total"], "modified": tl["counts_modified"],
                        "unmodified": tl["counts_unmodified"], "with_insertion": tl["counts_insertion"],
                        "with_deletion": tl["counts_deletion"], "with_substitution": tl["counts_substitution"]} for r, tl in list(enumerate(tallies))[:4]],
            "per_rank_reads_aligned": per_rank_aligned,
            "host": {"cpus": ncpu, "data_generation_s": t_gen},
        }
        if selection is not None:
            out["selection"] = selection
        if cpu_baseline:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu_baseline["value"]
        # a side leg that failed or did not finish leaves its error in its own entry; this flag is what a script should look at (the exit code
        # stays 0: the headline was measured, and N ranks must all end the same way -- ADVICE r04)
        def _has_error(x):
            return isinstance(x, dict) and ("error" in x or any(_has_error(v) for v in x.values()))
        out["side_legs_ok"] = bool(extras_note[0] is None and not any(_has_error(leg) for leg in (int32_chain, other_configs, robustness, e2e, host_batch)))
        if extras_note[0]:
            out["side_legs_note"] = extras_note[0]
        # Everything above goes to bench_detail.json (legs, notes, stage seconds); stdout gets ONE short line (< 4 KB, no prose) with what the
        # contract names -- a 31 KB line was more than the driver's record could parse (BENCH_r05.json: "parsed": null).
        detail_path = os.environ.get("C2_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail_path, "w") as fh:
                json.dump(out, fh)
                fh.write("\n")
        except OSError as e:
            detail_path = "not written: %r" % (e,)
        print(short_line(out, detail_path), file=file or sys.stdout)
        (file or sys.stdout).flush()

    if extras and world > 1:
        # The side legs below hold collectives.  Whatever happens in them on any rank -- an exception on one rank leaves the others waiting in
        # a collective -- must not cost the headline that was already measured: every rank runs a watchdog; when the legs have not finished
        # within C2_BENCH_EXTRAS_TIMEOUT seconds, rank 0 prints the line with what there is and every rank ends with exit code 0.
        limit = float(os.environ.get("C2_BENCH_EXTRAS_TIMEOUT", "600"))

        def watchdog():
            if extras_done.wait(limit):
                return
            if rank == 0:
                extras_note[0] = "the side legs did not finish within %.0f s on every rank (%s): the line carries what was there" % (limit, extras_note[0] or "no error on rank 0")
                try:
                    emit(sys.__stdout__)
                finally:
                    os._exit(0)
            time.sleep(5.0)                                           # (rank 0 prints first)
            os._exit(0)
        threading.Thread(target=watchdog, name="c2-bench-extras-watchdog", daemon=True).start()

    def side_leg_failed(e):
        """world > 1: this rank cannot go on with the side legs (the others may be waiting for it in a collective): say why and wait for the watchdog"""
        sys.stderr.write("bench.py rank %d: a side leg failed: %r\n" % (rank, e))
        sys.stderr.flush()
        extras_note[0] = repr(e)
        threading.Event().wait()
    try:
        if extras:
            # the same reads, buffers and step with the 32-bit kernels only (the reference's DP is C int, pyx:142-147): all ranks take part
            ctx.set_kernel_mode("diag4")
            job.kernel = "diag4"
            t32 = job.timed(1, args.extra_steps)
            rec32_same = bool(torch.equal(job.outputs[2].cpu(), torch.from_numpy(rec.view(np.uint8).reshape(-1, 32))))
            int32_chain = {"reads_per_s": t32["reads_per_s"], "ms_per_step": 1e3 * t32["dt"] / args.extra_steps, "steps": args.extra_steps,
                           "kernel_chain": chain_names["diag4"] + [chain[-1]], "dtype": "int32",
                           "tasks_left_after_each_banded_launch": ctx.tier_info(), "align_chain_ms": t32["align_ms"],
                           "records_equal_the_packed_chain": rec32_same,
                           "note": "c2_set_kernel_mode(diag4): the same batch, buffers and step as the headline with every DP cell in int32"}
            ctx.set_kernel_mode(args.kernel)
            job.kernel = args.kernel
        job.free()
        del job, d_aln_read, d_aln_ref, d_records
        torch.cuda.empty_cache()
        if extras:
            def run_leg(wlc, Lc, ref_leg):
                """one more workload through the default chain: 1 warm-up + --extra-steps timed steps, the tier shares, chain = full plane on every
                task, the reference-compiled slice"""
                jc = Job(ctx, wlc, Lc, m, dev, world, kernel="auto")
                tc = jc.timed(1, args.extra_steps)
                tiers_c = ctx.tier_info()
                part_c = ctx.partition_info() if hasattr(ctx, "partition_info") else None
                rec_c = jc.outputs[2].cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
                entry = {"workload": wlc["text"], "reads_per_gpu_per_step": jc.n, "alignments_per_gpu_per_step": jc.n_tasks, "n_amplicons": jc.k,
                         "steps": args.extra_steps, "ms_per_step": 1e3 * tc["dt"] / args.extra_steps, "reads_per_s": tc["reads_per_s"],
                         "alignments_per_s": tc["alignments_per_s"],
                         "step_breakdown_ms": {"align_chain": tc["align_ms"], "select_best": tc["select_ms"], "count_vectors_and_all_reduce": tc["count_ms"]},
                         "tasks_left_after_each_banded_launch": tiers_c, "all_status_ok": bool((rec_c["status"] == 0).all())}
                if part_c and part_c.get("ran"):
                    # which launch sees a task FIRST (c2_align_partition_kernel's classes) and what each tier hands on: the tier shares of this input
                    cls_ = part_c["classes"]
                    entry["first_launch_share"] = {"score_only": cls_[0] / float(jc.n_tasks), "band_32_diagonals": (cls_[1] + cls_[2]) / float(jc.n_tasks),
                                                   "band_40_diagonals": cls_[3] / float(jc.n_tasks), "band_62_diagonals": cls_[4] / float(jc.n_tasks),
                                                   "band_128_diagonals": cls_[5] / float(jc.n_tasks), "full_matrix": cls_[6] / float(jc.n_tasks)}
                    entry["score_only_finished_share"] = part_c["finished"][0] / float(jc.n_tasks)
                    entry["finished_by_partition_share"] = part_c.get("finished_by_partition", 0) / float(jc.n_tasks)      # main-diagonal reads (copies of the reference, one or two differing bases): finished by the partition, no fill
                entry["full_plane_launch_share"] = (tiers_c[-1] / float(jc.n_tasks)) if tiers_c else None
                if ref_leg is not None:
                    leg_c, kind_c = ref_leg
                    if "error" in leg_c:
                        entry["reference_check_error"] = leg_c["error"]
                    else:
                        cmp_n, same_n = _compare_with_reference([leg_c], jc.outputs[0], jc.outputs[1], rec_c, jc.k, jc.all_refs)
                        entry["reference_compared_n"], entry["reference_identical_n"] = cmp_n, same_n
                        entry["reference_identical"] = bool(cmp_n == same_n)
                        entry["reference_check"] = {"kind": kind_c, "procs": leg_c["procs"], "seconds": leg_c["seconds"],
                                                    "note": "the first reads of this workload aligned by the reference's compiled code (oracle/_ref) on the "
                                                            "host before the timed region; strings by digest + the best alignment's three window counts"}
                del rec_c
                if rank == 0 and args.check > 0 and not args.no_full_plane_check:
                    eq, tf = jc.chain_equals_full_plane()
                    entry["chain_equals_full_plane_n"] = eq
                    entry["chain_equals_full_plane"] = bool(eq == jc.n_tasks)
                    entry["full_plane_pass_s"] = tf
                eqh_ = jc.count_tensor_equals_without_hints() if (rank == 0 and args.check > 0) else None
                if eqh_ is not None:
                    entry["count_tensor_equals_without_hints"] = eqh_     # (the step counted its hinted tasks from their hint words: the same tensor over the rows)
                tl = jc.tallies()
                entry["reads_aligned_all_gpus"] = int(sum(t_["counts_total"] for t_ in tl))
                entry["modified"] = int(sum(t_["counts_modified"] for t_ in tl))
                if jc.all_refs:
                    entry["selection"] = dict(zip(C.SELECT_STATS, jc.d_selstats.cpu().numpy().tolist()))
                jc.free()
                del jc
                return entry

            other_configs = {"data_generation_s": t_gen_other}
            for cfg, (Lc, wlc) in sorted(other_wl.items()):
                try:
                    entry = run_leg(wlc, Lc, other_ref.get(cfg))
                except Exception as e:                                   # (a side leg reports its failure; with several ranks the others may be waiting in a
                    entry = {"error": repr(e)}                           #  collective: this rank waits for the watchdog, which still prints the headline)
                    if world > 1:
                        other_configs["config%d" % cfg] = entry
                        side_leg_failed(e)
                other_configs["config%d" % cfg] = entry
            # the headline's read budget on inputs that are not the generator's best case
            robustness = {"data_generation_s": t_gen_robust,
                          "full_plane_floor": None if "full_plane_pass_s" not in checks else {
                              "reads_per_s": n_tasks / checks["full_plane_pass_s"], "seconds": checks["full_plane_pass_s"],
                              "note": "the headline batch through c2_align_classify_kernel alone (--kernel full: every cell of every matrix, no band, no certificate): "
                                      "what ANY input of this size costs at most per launch chain"}}
            for name, wlc in robust_wl.items():
                try:
                    entry = run_leg(wlc, wlc["max_len"], robust_ref.get(name))
                except Exception as e:
                    entry = {"error": repr(e)}
                    if world > 1:
                        robustness[name] = entry
                        side_leg_failed(e)
                robustness[name] = entry
            rates = {k_: v_["reads_per_s"] for k_, v_ in robustness.items() if k_ != "full_plane_floor" and isinstance(v_, dict) and "reads_per_s" in v_}
            if rates:
                worst = min(rates, key=rates.get)
                robustness["worst_case"] = {"leg": worst, "reads_per_s": rates[worst]}
            del robust_wl
            del other_wl
            if world > 1:
                # the sharded FASTQ leg: rank 0 wrote the file (same node: /dev/shm), every rank ingests its byte range of it
                box = [None if (e2e_files is None or "skipped" in e2e_files) else (e2e_files["plain"], e2e_files["reads"])]
                dist.broadcast_object_list(box, src=0)
                if box[0] is not None:
                    try:
                        e2e = {"sharded": _e2e_sharded_leg(box[0][0], box[0][1], ctx, L, m, dev)}
                    except Exception as e:
                        e2e = {"sharded": {"error": repr(e)}}
                        side_leg_failed(e)
                    dist.barrier()
                    if rank == 0:
                        shutil.rmtree(e2e_files["dir"], ignore_errors=True)
                elif e2e_files is not None:
                    e2e = e2e_files
            elif e2e_files is not None:
                if "skipped" in e2e_files:
                    e2e = e2e_files
                else:
                    try:
                        e2e = _e2e_leg(e2e_files, ctx, L, m)
                    except Exception as e:
                        e2e = {"error": repr(e)}
                    finally:
                        shutil.rmtree(e2e_files["dir"], ignore_errors=True)
            if world == 1 and not all_refs and wl["ref_ids"] is None:
                try:
                    host_batch = _host_batch_leg(ctx, reads, L, m, n_reads=min(2_000_000, args.extra_reads or 2_000_000))
                except Exception as e:
                    host_batch = {"error": repr(e)}

    except Exception as e:                                       # (one rank alone in trouble: the others wait in a collective; see the watchdog)
        if world == 1:
            raise
        side_leg_failed(e)
    extras_done.set()
    if rank == 0:
        emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
