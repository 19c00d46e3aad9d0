"""TEST INFRASTRUCTURE ONLY -- runs bench.py's own main() on a machine without a GPU: torch's "cuda" device becomes the CPU, and
the four device entry points bench.py uses (BatchAligner.align_device, counts.accumulate_device, counts.select_best_device,
the Context's bookkeeping calls) are redirected to the wave emulator (tests/emu: the same HIP kernel source compiled for the
host).  Used to check the script's plumbing -- workload builders, the CPU-baseline legs and the exhaustive comparison of
their digests with the "device" output, the chain-vs-full-plane check, the JSON line -- before GPU minutes are spent on it."""
import contextlib
import ctypes
import io
import os
import json
import sys

import numpy as np

import emu_driver as E
from pipeline_on_emulator import EmulatedAligner, _view, _accumulate, _select


class _Ctx:
    """stands in for _native.Context"""
    def __init__(self, device=0):
        self.mode = "auto"
        self.tiers = []

    def set_band(self, *a): pass
    def set_kernel_mode(self, mode): self.mode = mode
    def timing_enable(self, on): pass
    def timing_read_split(self): return 3.0, 2.0, 1
    def tier_info(self): return [7, 3, 1]
    def tier_info_ex(self): return [7, 3, 1], [1, 0, 0]
    def launch_info(self, L): return dict(rows_per_lane=4, passes=1, lds_bytes=0, workgroups_per_cu=1, compute_units=1)
    def band_info(self, L): return dict(band_lanes=-1, fallback_tasks_last_launch=0)


class _Aligner(EmulatedAligner):
    """the launch chain the context's kernel mode asks for: 'auto' = 4 -> 2 -> 1 -> full plane, 'full' = full plane only"""
    def __init__(self, *a, ctx=None, **kw):
        super().__init__(*a, **kw)
        self.ctx = ctx

    def align_device(self, n_reads, d_reads, d_offsets, d_aln_read, d_aln_ref, d_records, aln_stride, max_read_len,
                     d_ref_ids=None, d_strands=None, all_refs=False, stream=None, legacy=False, min_read_len=0, d_hints=None):
        n, k = int(n_reads), len(self.seqs)
        off = _view(d_offsets, 8 * (n + 1)).view(np.int64)
        arena = _view(d_reads, max(int(off[-1]), 1)).tobytes()
        reads = [arena[int(off[i]):int(off[i + 1])].decode() for i in range(n)]
        ntasks = n * k if all_refs else n
        rids = None if d_ref_ids is None else _view(d_ref_ids, 2 * n).view(np.int16).astype(np.uint16)
        st = {"want_hints": bool(d_hints)}
        _, rec = E.align_batch(reads, self.seqs, self.g, self.inc, self.m, self.go, self.ge, ref_ids=rids, all_refs=all_refs,
                               band_lanes=-87 if self.ctx.mode == "auto" else 0, stats=st)
        if d_hints:
            _view(d_hints, 16 * ntasks).view(np.uint32)[:] = st["hints"].reshape(-1)
        o1, o2 = st["raw"]
        w = min(o1.shape[1], aln_stride)
        a = _view(d_aln_read, ntasks * aln_stride).reshape(ntasks, aln_stride)
        f = _view(d_aln_ref, ntasks * aln_stride).reshape(ntasks, aln_stride)
        a[:, :w] = o1[:, :w]
        f[:, :w] = o2[:, :w]
        _view(d_records, 32 * ntasks)[:] = rec.view(np.uint8).reshape(-1)


class _Event:
    def __init__(self, enable_timing=False): pass
    def record(self, stream=None): pass
    def elapsed_time(self, other): return 1.0


def run_bench(argv):
    """bench.main() with `argv` on the emulator -> the parsed JSON line"""
    import torch
    import bench
    from crispresso2_amd import _native, batch, counts as C
    made = []

    def make_aligner(*a, **kw):
        made.append(_Aligner(*a, **kw))
        return made[-1]

    class _Stream:
        cuda_stream = 0
        def __init__(self, *a, **k): pass
        def wait_event(self, e): pass
        def __enter__(self): return self
        def __exit__(self, *exc): return False
    real_device = torch.device
    saved = (torch.device, torch.cuda.current_stream, torch.cuda.synchronize, torch.cuda.set_device, torch.cuda.Event,
             batch.BatchAligner, C.accumulate_device, C.select_best_device, _native.Context, sys.argv, sys.stdout,
             torch.cuda.Stream, torch.cuda.stream, torch.cuda.empty_cache)
    def apply():
        torch.cuda.Stream = _Stream
        torch.cuda.stream = lambda st: st
        torch.cuda.empty_cache = lambda: None
        torch.device = lambda *a, **k: real_device("cpu")
        torch.cuda.current_stream = lambda *a, **k: _Stream()
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.set_device = lambda *a, **k: None
        torch.cuda.Event = _Event
        batch.BatchAligner = make_aligner
        C.accumulate_device = _accumulate(made)
        C.select_best_device = _select
        _native.Context = _Ctx
    sys.argv = ["bench.py"] + list(argv)
    import tempfile
    own_detail = "C2_BENCH_DETAIL" not in os.environ
    if own_detail:                                                   # (the full record goes to a file, the short line to stdout: see bench.emit)
        fd, detail_path = tempfile.mkstemp(prefix="c2_bench_detail_", suffix=".json")
        os.close(fd)
        os.environ["C2_BENCH_DETAIL"] = detail_path
    detail_path = os.environ["C2_BENCH_DETAIL"]
    buf = io.StringIO()
    sys.stdout = buf
    from pipeline_on_emulator import emulated_device
    try:
        # (the FASTQ -> tensors leg runs pipeline.quantify_fastq: its device calls go to the emulator too; ONE list of aligners, so
        # that the count pass of either route sees the aligner of the batch it counts)
        with emulated_device(made):
            apply()
            bench.main()
    finally:
        (torch.device, torch.cuda.current_stream, torch.cuda.synchronize, torch.cuda.set_device, torch.cuda.Event,
         batch.BatchAligner, C.accumulate_device, C.select_best_device, _native.Context, sys.argv, sys.stdout,
         torch.cuda.Stream, torch.cuda.stream, torch.cuda.empty_cache) = saved
    if own_detail:
        os.environ.pop("C2_BENCH_DETAIL", None)
    lines = [x for x in buf.getvalue().splitlines() if x.startswith("{")]
    if int(os.environ.get("RANK", "0")) != 0:                        # only rank 0 prints
        assert not lines, buf.getvalue()
        return None
    assert len(lines) == 1, buf.getvalue()
    # the ONE stdout line is short and self-contained; the full record (what these tests read) is the detail file
    assert len(lines[0]) < bench.SHORT_LINE_LIMIT, len(lines[0])
    short = json.loads(lines[0])
    with open(detail_path) as fh:
        out = json.load(fh)
    if own_detail:
        os.unlink(detail_path)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert short[key] == out[key] or (isinstance(out[key], float) and abs(short[key] - out[key]) <= 1e-4 * abs(out[key])), key
    out["_short"] = short
    out["_line_bytes"] = len(lines[0])
    return out
