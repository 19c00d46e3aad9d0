"""CPU baseline leg of bench.py -- TEST/MEASUREMENT INFRASTRUCTURE ONLY.

Times the reference's own Cython hot path (oracle/_ref, kind "reference") -- or, if that is not
built, the C restatement (kind "port") -- on this host's cores: a process pool over chunks of reads,
each worker calling global_align + find_indels_substitutions per read, nothing else in the loop
(BASELINE.md plan (A), the figure most favourable to the reference)."""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_state = {}


def _init(amplicon, gap_incentive, include, matrix_path, go, ge):
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    import numpy as np
    import oracle
    ref = oracle.ref()
    if ref is not None:
        A, R = ref
        m = A.read_matrix(matrix_path)
        _state["kind"] = "reference"
        _state["align"] = lambda rd: A.global_align(rd, amplicon, matrix=m, gap_incentive=g, gap_open=go, gap_extend=ge)
        _state["classify"] = lambda s1, s2: R.find_indels_substitutions(s1, s2, inc)
    else:
        from crispresso2_amd import CRISPResso2Align as PA      # read_matrix only (file parsing)
        m = PA.read_matrix(matrix_path)
        _state["kind"] = "port"
        _state["align"] = lambda rd: oracle.global_align(rd, amplicon, m, g, go, ge)
        _state["classify"] = lambda s1, s2: oracle.find_indels_substitutions(s1, s2, inc)
    g = np.asarray(gap_incentive, dtype=np.int64)
    inc = np.asarray(include)


def _work(chunk):
    """chunk: (bytes of concatenated fixed-length reads, read length) -> (reads done, modified reads)"""
    blob, L = chunk
    n = len(blob) // L
    mod = 0
    align, classify = _state["align"], _state["classify"]
    for k in range(n):
        s1, s2, _ = align(blob[k * L:(k + 1) * L].decode())
        p = classify(s1, s2)
        if p["insertion_n"] or p["deletion_n"] or p["substitution_n"]:
            mod += 1
    return n, mod


def _kind(_):
    return _state["kind"]


def run(reads_u8, amplicon, gap_incentive, include, matrix_path, go, ge, cores=None, target_seconds=15.0):
    """reads_u8: uint8 [n, L].  Uses a bounded prefix sized for ~target_seconds of work.  -> dict"""
    import multiprocessing as mp
    cores = cores or os.cpu_count() or 1
    n, L = reads_u8.shape
    ctx = mp.get_context("fork")      # bench.py calls this BEFORE it touches HIP, so fork is safe
    with ctx.Pool(cores, initializer=_init, initargs=(amplicon, list(map(int, gap_incentive)), list(map(int, include)),
                                                      matrix_path, go, ge)) as pool:
        kind = pool.map(_kind, range(cores))[0]
        # calibrate on a few reads per worker
        cal = min(n, 64 * cores)
        t0 = time.perf_counter()
        pool.map(_work, [(reads_u8[k:k + 64].tobytes(), L) for k in range(0, cal, 64)])
        per_read = (time.perf_counter() - t0) * cores / max(cal, 1)
        sample = int(min(n, max(cores * 256, target_seconds * cores / max(per_read, 1e-6))))
        chunk = max(64, sample // (cores * 8))
        chunks = [(reads_u8[k:min(k + chunk, sample)].tobytes(), L) for k in range(0, sample, chunk)]
        t0 = time.perf_counter()
        done = pool.map(_work, chunks)
        dt = time.perf_counter() - t0
    nd = sum(d[0] for d in done)
    return {"value": nd / dt, "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": "first %d of the benchmark's reads (%d bp vs %d bp amplicon), %.1f s wall on %d processes, "
                      "global_align + find_indels_substitutions per read" % (nd, L, len(amplicon), dt, cores),
            "modified_in_sample": sum(d[1] for d in done)}
