"""TEST INFRASTRUCTURE ONLY -- runs crispresso2_amd.pipeline's host logic on a machine without a GPU by redirecting its three
device entry points to the wave emulator (tests/emu: the same HIP kernel source compiled for the host):

    torch "cuda" tensors        -> CPU tensors (their data_ptr()s are host addresses the emulator can write to)
    BatchAligner.align_device   -> emu_driver.align_batch over the bytes behind those addresses (default launch chain)
    counts.accumulate_device    -> emu_driver.count_vectors, added into the tensor behind d_counts

Everything else -- native ingest, strand plans, strand / best-amplicon selection, reverse-complement merge, weights, the
first-amplicon view, statistics, allele rows -- is the product's own code.  Used by tests only; the product never imports this."""
import contextlib
import ctypes

import numpy as np

import emu_driver as E


def _view(addr, nbytes):
    return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(addr))


class EmulatedAligner:
    def __init__(self, seqs, gap_incentives, include_idxs, matrix, gap_open, gap_extend, ctx=None, device=None):
        self.seqs, self.g, self.inc = list(seqs), [np.asarray(x, dtype=np.int64) for x in gap_incentives], [list(x) for x in include_idxs]
        self.m, self.go, self.ge = matrix, gap_open, gap_extend
        self.max_ref_len = max(len(s) for s in seqs)

    def stride_for(self, max_read_len):
        return (self.max_ref_len + int(max_read_len) + 15) // 16 * 16

    def align_device(self, n_reads, d_reads, d_offsets, d_aln_read, d_aln_ref, d_records, aln_stride, max_read_len,
                     d_ref_ids=None, d_strands=None, all_refs=False, stream=None):
        n, k = int(n_reads), len(self.seqs)
        off = _view(d_offsets, 8 * (n + 1)).view(np.int64)
        arena = _view(d_reads, max(int(off[-1]), 1)).tobytes()
        reads = [arena[int(off[i]):int(off[i + 1])].decode() for i in range(n)]
        ntasks = n * k if all_refs else n
        strands = None if d_strands is None else _view(d_strands, ntasks).copy()
        rids = None if d_ref_ids is None else _view(d_ref_ids, 2 * n).view(np.int16).astype(np.uint16)
        st = {}
        _, rec = E.align_batch(reads, self.seqs, self.g, self.inc, self.m, self.go, self.ge, ref_ids=rids, strands=strands,
                               all_refs=all_refs, band_lanes=-7, stats=st)
        o1, o2 = st["raw"]
        w = min(o1.shape[1], aln_stride)
        assert int(rec["aln_len"].max()) <= w
        a = _view(d_aln_read, ntasks * aln_stride).reshape(ntasks, aln_stride)
        f = _view(d_aln_ref, ntasks * aln_stride).reshape(ntasks, aln_stride)
        a[:, :w] = o1[:, :w]
        f[:, :w] = o2[:, :w]
        _view(d_records, 32 * ntasks)[:] = rec.view(np.uint8).reshape(-1)


def _accumulate(aligners):
    def accumulate_device(ctx, layout, n_tasks, d_aln_read, d_aln_ref, aln_stride, d_records, d_counts, d_weights=None,
                          min_matches=None, flags=0, stream=None):
        al = aligners[-1]
        a = _view(d_aln_read, n_tasks * aln_stride).reshape(n_tasks, aln_stride)
        f = _view(d_aln_ref, n_tasks * aln_stride).reshape(n_tasks, aln_stride)
        rec = _view(d_records, 32 * n_tasks).view(E.REC_DTYPE).reshape(-1)
        w = None if not d_weights else _view(d_weights, 4 * n_tasks).view(np.uint32).copy()
        counts, lay = E.count_vectors(a, f, rec, al.seqs, al.inc, layout.hl - layout.lmax - 2,
                                      weights=w, min_matches=min_matches, flags=flags)
        assert lay.shape() == layout.shape(), (lay.shape(), layout.shape())
        n64 = int(np.prod(layout.shape()))
        _view(d_counts, 8 * n64).view(np.int64)[:] += counts.reshape(-1)
    return accumulate_device


@contextlib.contextmanager
def emulated_device():
    """Inside the block pipeline.quantify_* run on the emulator; restored afterwards."""
    import torch
    from crispresso2_amd import pipeline, counts as C, _native
    made = []

    def make_aligner(*a, **kw):
        made.append(EmulatedAligner(*a, **kw))
        return made[-1]

    class _Stream:
        cuda_stream = 0
    saved = (torch.device, torch.cuda.current_stream, torch.cuda.synchronize, pipeline.BatchAligner, C.accumulate_device, _native.default_context)
    real_device = torch.device
    torch.device = lambda *a, **k: real_device("cpu")
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.synchronize = lambda *a, **k: None
    pipeline.BatchAligner = make_aligner
    C.accumulate_device = _accumulate(made)
    _native.default_context = lambda *a, **k: object()
    try:
        yield
    finally:
        torch.device, torch.cuda.current_stream, torch.cuda.synchronize, pipeline.BatchAligner, C.accumulate_device, _native.default_context = saved
