"""CPU baseline leg of bench.py -- TEST/MEASUREMENT INFRASTRUCTURE ONLY.

Times the reference's own Cython hot path (oracle/_ref, kind "reference") -- or, if that is not built, the C restatement
(kind "port") -- on this host's cores: process pools over chunks of reads, each worker calling global_align per (read,
candidate amplicon) and find_indels_substitutions on the best alignment, nothing else in the loop (BASELINE.md plan (A),
the figure most favourable to the reference).  Measured: ONE process, then a sweep over pool sizes
(8, 16, 32, ... up to four times the CPUs this process may use: its affinity mask cut down by the cgroup's cpu.max quota), then the best
pool size once more for a minute; that long leg's rate is the baseline and the curve is reported with it.

Every alignment the legs compute is also kept as a 64-bit digest of its two aligned strings (blake2b; ~1 us next to the
~0.8-5 ms the alignment takes in the pool) together with the classifier's three counts, so bench.py can compare ALL of them
with what the device produced for the same reads (checks.reference_identical_n) instead of throwing them away."""
import hashlib
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_state = {}


def digest(s1, s2):
    """64-bit digest of one alignment's two strings (bytes); bench.py applies the same function to the device's output."""
    return int.from_bytes(hashlib.blake2b(s1 + b"|" + s2, digest_size=8).digest(), "little")


def _init(refs, matrix_path, go, ge):
    """refs: list of (sequence, gap_incentive list, include list)"""
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    import oracle
    ref = oracle.ref()
    if ref is not None:
        A, R = ref
        m = A.read_matrix(matrix_path)
        _state["kind"] = "reference"
        _state["align"] = lambda rd, r: A.global_align(rd, seqs[r], matrix=m, gap_incentive=gs[r], gap_open=go, gap_extend=ge)
        _state["classify"] = lambda s1, s2, r: R.find_indels_substitutions(s1, s2, incs[r])
    else:
        from crispresso2_amd import CRISPResso2Align as PA      # read_matrix only (file parsing)
        m = PA.read_matrix(matrix_path)
        _state["kind"] = "port"
        _state["align"] = lambda rd, r: oracle.global_align(rd, seqs[r], m, gs[r], go, ge)
        _state["classify"] = lambda s1, s2, r: oracle.find_indels_substitutions(s1, s2, incs[r])
    seqs = [x[0] for x in refs]
    gs = [np.asarray(x[1], dtype=np.int64) for x in refs]
    incs = [np.asarray(x[2]) for x in refs]
    _state["n_refs"] = len(refs)


def _work(chunk):
    """chunk: (bytes of concatenated fixed-width rows -- a read, padded with NUL bytes if it is shorter --, row width, ref ids bytes (uint16) or None, all_refs)
    -> (reads done, modified reads, digests uint64 [tasks], counts int32 [reads, 3], best ref per read uint16)"""
    blob, L, rid_bytes, all_refs = chunk
    n = len(blob) // L
    k = _state["n_refs"] if all_refs else 1
    rids = np.frombuffer(rid_bytes, dtype=np.uint16) if rid_bytes is not None else None
    align, classify = _state["align"], _state["classify"]
    dig = np.zeros(n * k, dtype=np.uint64)
    cnt = np.zeros((n, 3), dtype=np.int32)
    best_ref = np.zeros(n, dtype=np.uint16)
    mod = 0
    for i in range(n):
        rd = blob[i * L:(i + 1) * L].rstrip(b"\0").decode()          # (ragged workloads: rows padded with NUL bytes)
        if all_refs:
            # get_new_variant_object's loop over the candidate amplicons (CRISPRessoCORE.py:653-707), forward strand
            best, bs = None, -1.0
            for r in range(k):
                s1, s2, sc = align(rd, r)
                dig[i * k + r] = digest(s1.encode(), s2.encode())
                if sc > bs:
                    bs, best = sc, (s1, s2, r)
            s1, s2, r = best
        else:
            r = int(rids[i]) if rids is not None else 0
            s1, s2, _ = align(rd, r)
            dig[i] = digest(s1.encode(), s2.encode())
        p = classify(s1, s2, r)
        cnt[i] = (p["insertion_n"], p["deletion_n"], p["substitution_n"])
        best_ref[i] = r
        if p["insertion_n"] or p["deletion_n"] or p["substitution_n"]:
            mod += 1
    return n, mod, dig, cnt, best_ref


def _kind(_):
    return _state["kind"]


def _leg(ctx, procs, refs, matrix_path, go, ge, reads_u8, ref_ids, all_refs, start, seconds, max_reads, rate_hint=None):
    """One pool size on reads [start, start + sample) -> (dict, reads consumed, kind).  The sample is sized from a calibration on a few
    reads per worker, or from rate_hint (reads/s this pool size was measured at before: the calibration's cold start underestimates; a quarter more reads than that rate needs, so that the leg lasts at least `seconds` also when the long run is faster than the short one)."""
    n, L = reads_u8.shape
    n = min(n, start + max_reads)
    with ctx.Pool(procs, initializer=_init, initargs=(refs, matrix_path, go, ge)) as pool:
        kind = pool.map(_kind, range(procs))[0]

        def chunk_of(a, b):
            return (reads_u8[a:b].tobytes(), L, None if ref_ids is None else np.ascontiguousarray(ref_ids[a:b], dtype=np.uint16).tobytes(), all_refs)
        # calibrate on a few reads per worker (kept: they are compared with the device too)
        per = 8 if procs > 1 else 32
        cal = min(n - start, per * procs)
        t0 = time.perf_counter()
        done0 = pool.map(_work, [chunk_of(start + a, min(start + a + per, start + cal)) for a in range(0, cal, per)])
        per_read = (time.perf_counter() - t0) * procs / max(cal, 1)
        a0 = start + cal
        sample = int(min(n - a0, max(procs * 16, seconds * rate_hint * 1.25 if rate_hint else seconds * procs / max(per_read, 1e-6))))
        chunk = max(8, sample // (procs * 8))
        bounds = [(a0 + a, min(a0 + a + chunk, a0 + sample)) for a in range(0, sample, chunk)]
        t0 = time.perf_counter()
        done = pool.map(_work, [chunk_of(a, b) for a, b in bounds])
        dt = time.perf_counter() - t0
    nd = sum(d[0] for d in done)
    allc = done0 + done
    out = {"procs": procs, "reads": nd, "seconds": dt, "reads_per_s": nd / dt if dt > 0 else 0.0,
           "modified_in_sample": int(sum(d[1] for d in done))}
    res = {"first_read": start, "n_reads": cal + nd,
           "digests": np.concatenate([d[2] for d in allc]) if allc else np.zeros(0, dtype=np.uint64),
           "counts": np.concatenate([d[3] for d in allc]) if allc else np.zeros((0, 3), dtype=np.int32),
           "best_ref": np.concatenate([d[4] for d in allc]) if allc else np.zeros(0, dtype=np.uint16)}
    return out, res, kind


def usable_cpus():
    """(CPUs this process may run on, the cgroup quota in CPUs or None): the affinity mask, and cpu.max of cgroup v2 / cfs_quota of v1"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()[:2]
            if q != "max":
                quota = int(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                p = int(fh.read())
            if q > 0:
                quota = q / float(p)
        except (OSError, ValueError):
            pass
    return n, quota


def reference_slice(reads_u8, refs, matrix_path, go, ge, ref_ids=None, all_refs=False, procs=8):
    """Every alignment of reads_u8 (a slice of a workload) by the reference's compiled code on a pool of `procs` processes -- not a
    timing, a checker: -> one result dict (first_read 0, n_reads, digests, counts, best_ref) for bench.py's digest comparison, + kind"""
    import multiprocessing as mp
    refs = [(s, list(map(int, g)), list(map(int, inc))) for s, g, inc in refs]
    n, L = reads_u8.shape
    ctx = mp.get_context("fork")
    with ctx.Pool(procs, initializer=_init, initargs=(refs, matrix_path, go, ge)) as pool:
        kind = pool.map(_kind, range(procs))[0]
        chunk = max(8, n // (procs * 8))
        jobs = [(reads_u8[a:min(n, a + chunk)].tobytes(), L,
                 None if ref_ids is None else np.ascontiguousarray(ref_ids[a:min(n, a + chunk)], dtype=np.uint16).tobytes(), all_refs) for a in range(0, n, chunk)]
        t0 = time.perf_counter()
        done = pool.map(_work, jobs)
        dt = time.perf_counter() - t0
    return {"first_read": 0, "n_reads": int(sum(d[0] for d in done)), "digests": np.concatenate([d[2] for d in done]),
            "counts": np.concatenate([d[3] for d in done]), "best_ref": np.concatenate([d[4] for d in done]), "seconds": dt, "procs": procs}, kind


def run(reads_u8, refs, matrix_path, go, ge, ref_ids=None, all_refs=False, cores=None, target_seconds=20.0, sweep=None, long_seconds=60.0):
    """reads_u8: uint8 [n, L]; refs: list of (sequence, gap_incentive, include_idxs); ref_ids: uint16 [n] or None;
    all_refs: every read against every reference.  Every leg works on its own consecutive slice of the reads, sized for its
    share of ~target_seconds; then the best pool size runs once more for ~long_seconds (0: no such leg) and gives the reported value.
    -> (dict for the bench line, list of per-leg results with the digests)"""
    import multiprocessing as mp
    host_cpus = cores or os.cpu_count() or 1
    affinity, quota = usable_cpus()
    usable = max(1, int(min(affinity, quota) if quota else affinity))
    refs = [(s, list(map(int, g)), list(map(int, inc))) for s, g, inc in refs]
    if sweep is None:
        sweep = [p for p in (8, 16, 32, 64, 128, 256, 512) if p <= min(host_cpus, 4 * usable)]
        if usable not in sweep and usable > 1:
            sweep.append(usable)
        sweep.sort()
    legs = [1] + [p for p in sweep if p > 1]
    ctx = mp.get_context("fork")      # bench.py calls this BEFORE it touches HIP, so fork is safe
    share = target_seconds / (len(legs) + 0.5)
    curve, results, start, kind = [], [], 0, None
    n, L = reads_u8.shape
    reserve = n // 2 if long_seconds > 0 else 0                      # (reads kept for the long leg)
    for q, p in enumerate(legs):
        if start >= n - reserve:
            break
        leg, res, kind = _leg(ctx, p, refs, matrix_path, go, ge, reads_u8[:n - reserve], ref_ids, all_refs, start, share * (0.5 if p == 1 else 1.0),
                              max(1, (n - reserve - start) // (len(legs) - q)))
        curve.append(leg)
        results.append(res)
        start += res["n_reads"]
    multi = [c for c in curve if c["procs"] > 1] or curve
    best = max(multi, key=lambda c: c["reads_per_s"])
    long_leg = None
    if long_seconds > 0 and start < n:
        long_leg, res, kind = _leg(ctx, best["procs"], refs, matrix_path, go, ge, reads_u8, ref_ids, all_refs, start, long_seconds, n - start,
                                   rate_hint=best["reads_per_s"])
        results.append(res)
        start += res["n_reads"]
    rep = long_leg or best
    one = next((c for c in curve if c["procs"] == 1), None)
    k = len(refs) if all_refs else 1
    out = {"value": rep["reads_per_s"], "unit": "reads/s", "cores": min(best["procs"], usable), "kind": kind,
           "best_procs": best["procs"], "host_cpus": host_cpus, "cpus_in_affinity_mask": affinity, "cgroup_cpu_quota": quota,
           "cores_note": "cores = min(pool size, CPUs this process may use: the affinity mask cut down by the cgroup's cpu.max quota)",
           "one_proc_reads_per_s": one["reads_per_s"] if one else None,
           "curve": [{"procs": c["procs"], "reads_per_s": c["reads_per_s"], "reads": c["reads"], "seconds": c["seconds"]} for c in curve],
           "long_leg": None if long_leg is None else {"procs": long_leg["procs"], "reads": long_leg["reads"], "seconds": long_leg["seconds"],
                                                      "reads_per_s": long_leg["reads_per_s"]},
           "alignments_per_read": k,
           "sample": "reads %d..%d of the benchmark's reads (%d bp, %d candidate amplicon%s per read), one consecutive slice per "
                     "pool size, then the best size (%d) for %.0f s: %d reads; global_align per (read, amplicon) + find_indels_substitutions "
                     "on the best alignment" % (0, start, L, k, "" if k == 1 else "s", best["procs"], rep["seconds"], rep["reads"]),
           "sample_short": "%d reads of the workload in %.0f s on %d processes: global_align + find_indels_substitutions per read" % (
               rep["reads"], rep["seconds"], best["procs"]),
           "modified_in_sample": rep["modified_in_sample"]}
    return out, results
