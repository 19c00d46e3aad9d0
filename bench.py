#!/usr/bin/env python3
"""bench.py -- aligned+classified reads/sec of the align + classify hot path on MI355X.

A "step" is one pass of the hot path (c2_align_classify_kernel: NW fill + traceback + fused
classification) over one batch of synthetic reads that is already resident in HBM.
N = 1 workload: BASELINE.json configs[2], "10M synthetic 250 bp reads vs one 250 bp amplicon"
(the configuration the metric is quoted on).  N > 1: every rank runs the same amount of work on its own
shard of the read stream (weak scaling, no data-path collective; the per-amplicon count reduction is the
only exchange and is included in the step when it is enabled).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run
with one rank per GPU.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz int32 lane-ops/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--len", type=int, default=250, dest="L", help="read and amplicon length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--workers", type=int, default=0, help="processes generating the synthetic reads (0 = auto; 1 = no fork, for profiler runs)")
    ap.add_argument("--band", type=int, default=-1, help="pointer-plane band: -1 auto, 0 off, n lanes each side")
    ap.add_argument("--band-wgs", type=int, default=0, help="target workgroups per CU for the automatic band")
    ap.add_argument("--kernel", choices=["auto", "band", "full", "diag1", "diag2"], default="auto",
                    help="kernel chain (auto = diagonal-band kernels with certificate, 4 -> 2 -> 1 alignments per wavefront)")
    ap.add_argument("--check", type=int, default=300, help="reads compared with the oracle after the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    L, n = args.L, args.reads

    from crispresso2_amd import synth
    amp, gap_inc, include = synth.amplicon_setup(L)
    matrix_path = os.path.join(ROOT, "crispresso2_amd", "EDNAFULL")

    # ---------- everything that fork()s happens before this process touches HIP ----------
    ncpu = os.cpu_count() or 1
    workers = args.workers if args.workers > 0 else max(1, min(32, ncpu // max(world, 1)))
    blocks_per_rank = (n + synth.BLOCK - 1) // synth.BLOCK
    t0 = time.perf_counter()
    reads = synth.make_reads(L, n, first_block=rank * blocks_per_rank, workers=workers)
    t_gen = time.perf_counter() - t0
    n_u = min(n, 1_000_000)
    unique_fraction = float(len(np.unique(reads[:n_u].view([("r", "V%d" % L)]))) / n_u)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline as cb
        cpu_baseline = cb.run(reads, amp, gap_inc, include, matrix_path, -20, -2, cores=ncpu, target_seconds=args.cpu_seconds)

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from crispresso2_amd import CRISPResso2Align as A, _native
    from crispresso2_amd.batch import BatchAligner
    m = A.read_matrix(matrix_path)
    ctx = _native.Context(local_rank)
    ctx.set_band(args.band, args.band_wgs)
    ctx.set_kernel_mode(args.kernel)
    al = BatchAligner([amp], [gap_inc], [include], m, -20, -2, ctx=ctx)
    stride = al.stride_for(L)

    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_offsets = (torch.arange(n + 1, dtype=torch.int64, device=dev) * L)
    d_aln_read = torch.empty((n, stride), dtype=torch.uint8, device=dev)
    d_aln_ref = torch.empty((n, stride), dtype=torch.uint8, device=dev)
    d_records = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # per-amplicon count tensor (CRISPRessoCORE.py:3865-4115 on the device) -- the only thing the GPUs exchange
    from crispresso2_amd import counts as C
    layout = C.CountLayout(1, L, L)
    d_counts = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)
    min_matches = C.min_matches_table([60.0], 2 * L)          # --default_min_aln_score 60

    def step():
        al.align_device(n, d_reads.data_ptr(), d_offsets.data_ptr(), d_aln_read.data_ptr(), d_aln_ref.data_ptr(),
                        d_records.data_ptr(), stride, L, stream=stream)
        d_counts.zero_()
        C.accumulate_device(ctx, layout, n, d_aln_read.data_ptr(), d_aln_ref.data_ptr(), stride, d_records.data_ptr(),
                            d_counts.data_ptr(), min_matches=min_matches, stream=stream)
        C.all_reduce(d_counts)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms, first_ms, launches = ctx.timing_read_split()
    ctx.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---------- after the timed region: algorithmic bytes of one launch, parity spot check ----------
    rec = d_records.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    ok_status = bool((rec["status"] == 0).all())
    aln_cols = int(rec["aln_len"].astype(np.int64).sum())
    bytes_in = n * L + (n + 1) * 8
    bytes_out = 2 * aln_cols + 32 * n
    alg_bytes = bytes_in + bytes_out
    avg_launch_s = (kernel_ms / max(launches, 1)) / 1e3          # the whole launch chain of one batch
    avg_first_s = (first_ms / max(launches, 1)) / 1e3            # its first kernel: the one that sees every task
    cells = n * (L + 1) * (L + 1)
    parity = None
    if rank == 0 and args.check > 0:
        import oracle
        rng = np.random.default_rng(12345)
        idx = rng.integers(0, n, args.check)
        a_r = d_aln_read[torch.from_numpy(idx).to(dev)].cpu().numpy()
        a_f = d_aln_ref[torch.from_numpy(idx).to(dev)].cpu().numpy()
        parity = True
        for j, k in enumerate(idx):
            st, s1, s2, mt, ln = oracle.global_align_raw(reads[k].tobytes().decode(), amp, m, gap_inc, -20, -2)
            T = int(rec["aln_len"][k])
            if st != 0 or T != ln or a_r[j, :T].tobytes().decode() != s1 or a_f[j, :T].tobytes().decode() != s2 or int(rec["matches"][k]) != mt:
                parity = False
                break
    # size-independent properties on EVERY alignment of the full-size batch (device-side, in slices): no double-gap column;
    # removing the gaps gives back the read and the amplicon (every base exactly once, in order); `matches` and
    # `all_deletion_bases` of the record agree with the strings
    props = None
    if args.check > 0:
        props = True
        d_amp = torch.from_numpy(np.frombuffer(amp.encode(), dtype=np.uint8).copy()).to(dev)
        d_reads2 = d_reads.view(n, L)
        recs_t = d_records.view(torch.int16)                      # aln_len, matches are the first two uint16 (< 32768 here)
        cols = torch.arange(stride, device=dev)[None, :]
        dash = ord("-")
        SL = 500_000
        for a0 in range(0, n, SL):
            a1 = min(n, a0 + SL)
            T_ = recs_t[a0:a1, 0].to(torch.int64)
            mt_ = recs_t[a0:a1, 1].to(torch.int64)
            delb_ = recs_t[a0:a1, 9].to(torch.int64)
            R_, F_ = d_aln_read[a0:a1], d_aln_ref[a0:a1]
            valid = cols < T_[:, None]
            rgap = (R_ == dash) & valid
            fgap = (F_ == dash) & valid
            ok = not bool((rgap & fgap).any())
            keep_r, keep_f = valid & ~rgap, valid & ~fgap
            ok = ok and bool((keep_r.sum(1) == L).all()) and bool((keep_f.sum(1) == L).all())
            if ok:
                ok = bool(torch.equal(R_[keep_r].view(a1 - a0, L), d_reads2[a0:a1]))
                ok = ok and bool(torch.equal(F_[keep_f].view(a1 - a0, L), d_amp[None, :].expand(a1 - a0, L)))
                ok = ok and bool(torch.equal(((R_ == F_) & keep_r & keep_f).sum(1), mt_)) and bool(torch.equal(rgap.sum(1), delb_))
            props = props and ok
            del valid, rgap, fgap, keep_r, keep_f
    info = ctx.launch_info(L)
    band = ctx.band_info(L)
    tiers = ctx.tier_info()
    chain_names = {"auto": ["c2_align_diagx_kernel<4>", "c2_align_diagx_kernel<2>", "c2_align_diag_kernel"],
                   "diag2": ["c2_align_diagx_kernel<2>", "c2_align_diag_kernel"], "diag1": ["c2_align_diag_kernel"]}
    # HBM bytes per alignment from the committed PMC passes (separate rocprofv3 --pmc runs of this same script; FETCH_SIZE
    # doubled as MI355X_MICROARCH.md prescribes for gfx950 streaming reads); null when the profile file is absent
    chain = ((chain_names.get(args.kernel, ["c2_align_diag_kernel"])[-len(tiers):] if band["band_lanes"] < 0 else
              ["c2_align_classify_kernel<%d, true>" % info["rows_per_lane"]] if band["band_lanes"] > 0 else [])
             + ["c2_align_classify_kernel<%d, false>" % info["rows_per_lane"]])
    dominant = chain[0]
    # algorithmic bytes of the dominant kernel's launch: every read and offset in; strings + record out for the tasks it finishes
    done_first = n - (tiers[0] if tiers else 0)
    alg_first = bytes_in + int(bytes_out * (done_first / float(n)))
    achieved_gbs = alg_first / avg_first_s / 1e9 if avg_first_s > 0 else 0.0
    traffic = traffic_src = None
    pmc_path = os.path.join(ROOT, "profiles", "r01", "pmc_summary_default.json")
    if os.path.exists(pmc_path) and L == 250 and args.kernel == "auto":
        with open(pmc_path) as fh:
            pmc = json.load(fh)["kernels"].get(dominant)
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 / 2.0e6 * n      # bytes per launch of n alignments
            traffic_src = ("profiles/r01/pmc_summary_default.json: (2*FETCH_SIZE + WRITE_SIZE) KB of %s over 2,000,000 reads, scaled to "
                           "the reads of one launch; includes the kernel's pointer-word scratch plane (written once, read back once)" % dominant)
    tallies = layout.unpack(d_counts.cpu().numpy(), 0, L)

    if rank == 0:
        total_reads = world * n * args.steps
        out = {
            "metric": "aligned+classified reads/sec (whole node), %d bp reads vs %d bp amplicon" % (L, L),
            "value": total_reads / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "%s synthetic %d bp reads vs one %d bp amplicon per GPU (BASELINE.json configs[2] shape), "
                                   "every read aligned (dedup off), EDNAFULL, gap_open -20, gap_extend -2, gap_incentive 1 at the cut"
                                   % ("{:,}".format(n), L, L),
                       "reads_per_gpu_per_step": n, "read_len": L, "amplicon_len": L, "unique_read_fraction": unique_fraction, "unique_read_fraction_sample": n_u,
                       "rows_per_lane": info["rows_per_lane"], "lds_bytes_per_workgroup": info["lds_bytes"],
                       "workgroups_per_cu": info["workgroups_per_cu"], "compute_units": info["compute_units"],
                       "kernel_chain": chain,
                       "tasks_left_after_each_banded_launch": tiers,
                       "pointer_band_lanes": band["band_lanes"], "full_plane_fallback_tasks": band["fallback_tasks_last_launch"]},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dominant, "avg_launch_ms": 1e3 * avg_first_s, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_first,
                         "algorithmic_bytes_per_read": alg_first / n,
                         "chain_avg_ms": 1e3 * avg_launch_s, "chain_algorithmic_bytes": alg_bytes,
                         "note": "integer DP: VALU-issue-bound by construction, HBM fraction is small (SURVEY 8d); see valu and profiles/r01/README.md"},
            "valu": {"cells_per_s": cells / avg_launch_s if avg_launch_s > 0 else 0.0,
                     "gcups": cells / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0,
                     "note": "full-matrix cell updates the reference would perform per second of launch-chain time (the banded kernels compute fewer)",
                     "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS},
            "cpu_baseline": cpu_baseline,
            "checks": {"all_status_ok": ok_status, "oracle_sample_identical": parity, "oracle_sample": args.check,
                       "full_batch_properties_hold": props},
            "counts": {"reads_aligned_all_gpus": tallies["counts_total"], "modified": tallies["counts_modified"],
                       "unmodified": tallies["counts_unmodified"], "with_insertion": tallies["counts_insertion"],
                       "with_deletion": tallies["counts_deletion"], "with_substitution": tallies["counts_substitution"]},
            "host": {"cpus": ncpu, "data_generation_s": t_gen},
        }
        if cpu_baseline:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu_baseline["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
