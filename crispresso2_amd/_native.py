"""ctypes binding of libcrispresso2_amd.so (the C ABI declared in include/crispresso2_amd.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, the calls
raise.  Build the library with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C crispresso2_amd/csrc`.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C2_AMD_LIB") or os.path.join(_HERE, "lib", "libcrispresso2_amd.so")   # (C2_AMD_LIB: another build of the same library, for A/B runs)

# every symbol include/crispresso2_amd.h declares
SYMBOLS = [
    "c2_device_count", "c2_create", "c2_destroy", "c2_last_error", "c2_abi_version",
    "c2_set_scoring", "c2_set_refs",
    "c2_align_classify_batch_device", "c2_align_classify_batch_host", "c2_synchronize",
    "c2_timing_enable", "c2_timing_read", "c2_launch_info",
    "c2_global_align", "c2_find_indels_substitutions", "c2_calculate_homology",
    "c2_selftest", "c2_selftest_rows", "c2_phase_profile", "c2_set_band", "c2_band_info", "c2_set_kernel_mode", "c2_tier_info", "c2_tier_info_ex", "c2_chain_info", "c2_timing_read_split", "c2_count_vectors_device", "c2_count_vectors_hinted_device", "c2_select_best_device",
    "c2_comm_unique_id", "c2_comm_init", "c2_reduce_counts", "c2_comm_destroy",
    "c2_classify_lists_batch", "c2_lists_total", "c2_lists_index", "c2_lists_values", "c2_lists_counts", "c2_lists_free",
    "c2_fastq_unique", "c2_fastq_unique_filtered", "c2_fastq_n_unique", "c2_fastq_n_reads", "c2_fastq_nonempty_lines", "c2_fastq_arena_bytes", "c2_fastq_arena", "c2_fastq_offsets",
    "c2_fastq_counts", "c2_fastq_free", "c2_fastq_stream_open", "c2_fastq_stream_next", "c2_fastq_stream_arena", "c2_fastq_stream_offsets", "c2_fastq_stream_text_bytes", "c2_fastq_stream_n_reads", "c2_fastq_stream_nonempty_lines", "c2_fastq_stream_nonempty_lines_input", "c2_fastq_stream_counts", "c2_fastq_stream_rc_partners", "c2_fastq_stream_close", "c2_fastq_last_error", "c2_strand_plan", "c2_strand_plan_device", "c2_merge_reverse_complements", "c2_rc_partners", "c2_merge_counts_with_partners", "c2_gather_reads",
    "c2_score_stage_info", "c2_partition_info", "c2_partition_finished", "c2_bgzf_open", "c2_bgzf_n_blocks", "c2_bgzf_text_offsets", "c2_bgzf_inflate", "c2_bgzf_close", "c2_gzseg_open", "c2_gz_inflate_parallel", "c2_gz_parallel_last",
    "c2_consensus_pairs_batch", "c2_consensus_pairs_device", "c2_classify_records_device",
    "c2_fq_count_device", "c2_fq_lines_device", "c2_fq_dedup_device", "c2_fq_gather_device", "c2_fq_rc_partner_device",
    "c2_fq_lines4_device", "c2_fq_pair_lengths_device", "c2_fq_pair_write_device",
    "c2_fastq_unique_paired", "c2_fastq_paired_occurrences", "c2_fastq_aux_bytes", "c2_fastq_aux", "c2_fastq_aux_offsets", "c2_fastq_stream_text",
    "c2_allele_table_build", "c2_allele_table_rows", "c2_allele_table_write", "c2_allele_table_write_zip", "c2_allele_table_fetch", "c2_allele_table_around_cut_write",
    "c2_allele_table_free", "c2_format_float_repr",
]

REC_DTYPE = np.dtype([
    ("aln_len", "<u2"), ("matches", "<u2"), ("insertion_n", "<u2"), ("deletion_n", "<u2"), ("substitution_n", "<u2"),
    ("all_insertion_events", "<u2"), ("win_insertion_events", "<u2"), ("all_deletion_events", "<u2"),
    ("win_deletion_events", "<u2"), ("all_deletion_bases", "<u2"), ("all_substitutions", "<u2"),
    ("irregular_ends", "u1"), ("status", "u1"), ("strand", "u1"), ("reserved0", "u1"),
    ("ref_id", "<u2"), ("reserved2", "<u4")])
assert REC_DTYPE.itemsize == 32

STATUS_EMPTY, STATUS_OOB_CHAR, STATUS_SENTINEL_PATH, STATUS_UNINIT_PTR, STATUS_RC_CHAR, STATUS_TOO_LONG = 1, 2, 4, 8, 16, 32
E_OVERFLOW = -6
ABI_VERSION = 3
LIST_COUNT = 15


class Batch(ctypes.Structure):
    """struct c2_batch"""
    _fields_ = [
        ("n_reads", ctypes.c_uint64),
        ("reads", ctypes.c_void_p), ("offsets", ctypes.c_void_p), ("ref_ids", ctypes.c_void_p), ("strands", ctypes.c_void_p),
        ("all_refs", ctypes.c_int32), ("max_read_len", ctypes.c_int32),
        ("aln_read", ctypes.c_void_p), ("aln_ref", ctypes.c_void_p),
        ("aln_stride", ctypes.c_uint32), ("flags", ctypes.c_uint32),
        ("records", ctypes.c_void_p),
        ("min_read_len", ctypes.c_int32),
        ("diag_hints", ctypes.c_void_p),
    ]


class NativeError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def load():
    """Load the shared library (no device call).  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeError("crispresso2_amd: %s is missing -- build the HIP extension first "
                                  "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
            # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and can no longer see the GPU once another
            # copy (the /opt/rocm one this library would pull in by itself) has initialised the device.  Importing torch
            # first makes the dynamic loader hand torch's copy to this library as well.  Without torch installed the
            # library simply uses /opt/rocm's.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            lib = ctypes.CDLL(LIB_PATH)
            if lib.c2_abi_version() != ABI_VERSION:                   # (struct layouts below are this version's: include/crispresso2_amd.h)
                raise NativeError("crispresso2_amd: %s has C ABI version %d, this binding was written for %d -- rebuild the library"
                                  % (LIB_PATH, lib.c2_abi_version(), ABI_VERSION))
            lib.c2_last_error.restype = ctypes.c_char_p
            lib.c2_last_error.argtypes = [ctypes.c_void_p]
            lib.c2_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
            lib.c2_destroy.argtypes = [ctypes.c_void_p]
            lib.c2_destroy.restype = None
            lib.c2_lists_total.restype = ctypes.c_uint64
            lib.c2_lists_total.argtypes = [ctypes.c_void_p]
            for fn in (lib.c2_lists_index, lib.c2_lists_values, lib.c2_lists_counts):
                fn.restype = ctypes.c_void_p
                fn.argtypes = [ctypes.c_void_p]
            lib.c2_lists_free.restype = None
            lib.c2_lists_free.argtypes = [ctypes.c_void_p]
            for fn in (lib.c2_fastq_n_unique, lib.c2_fastq_n_reads, lib.c2_fastq_nonempty_lines, lib.c2_fastq_arena_bytes, lib.c2_fastq_aux_bytes):
                fn.restype = ctypes.c_uint64
                fn.argtypes = [ctypes.c_void_p]
            lib.c2_fastq_unique_paired.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
            lib.c2_fastq_paired_occurrences.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p,
                                                        ctypes.POINTER(ctypes.c_void_p)]
            for fn in (lib.c2_fastq_arena, lib.c2_fastq_offsets, lib.c2_fastq_counts, lib.c2_fastq_aux, lib.c2_fastq_aux_offsets):
                fn.restype = ctypes.c_void_p
                fn.argtypes = [ctypes.c_void_p]
            lib.c2_fastq_free.restype = None
            lib.c2_fastq_free.argtypes = [ctypes.c_void_p]
            for fn in (lib.c2_fastq_stream_arena, lib.c2_fastq_stream_offsets, lib.c2_fastq_stream_text):
                fn.restype = ctypes.c_void_p
                fn.argtypes = [ctypes.c_void_p]
            for fn in (lib.c2_fastq_stream_text_bytes, lib.c2_fastq_stream_n_reads, lib.c2_fastq_stream_nonempty_lines,
                       lib.c2_fastq_stream_nonempty_lines_input):
                fn.restype = ctypes.c_uint64
                fn.argtypes = [ctypes.c_void_p]
            lib.c2_bgzf_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
            lib.c2_bgzf_n_blocks.restype = ctypes.c_uint64
            lib.c2_bgzf_n_blocks.argtypes = [ctypes.c_void_p]
            lib.c2_bgzf_text_offsets.restype = ctypes.c_void_p
            lib.c2_bgzf_text_offsets.argtypes = [ctypes.c_void_p]
            lib.c2_bgzf_inflate.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32]
            lib.c2_bgzf_close.restype = None
            lib.c2_bgzf_close.argtypes = [ctypes.c_void_p]
            lib.c2_gzseg_open.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
            lib.c2_gz_parallel_last.restype = None
            lib.c2_gz_parallel_last.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
            lib.c2_gz_inflate_parallel.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                                   ctypes.c_int32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
            lib.c2_fastq_stream_close.restype = None
            lib.c2_fastq_stream_close.argtypes = [ctypes.c_void_p]
            lib.c2_fastq_stream_open.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
            lib.c2_fastq_stream_next.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int32)]
            lib.c2_fastq_stream_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
            lib.c2_fastq_stream_rc_partners.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
            lib.c2_fastq_last_error.restype = ctypes.c_char_p
            lib.c2_fastq_unique.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
            lib.c2_fastq_unique_filtered.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
            _lib = lib
    return _lib


class Context:
    """One c2_ctx bound to one GPU."""

    def __init__(self, device=0):
        if in_forked_child():
            raise NativeError("crispresso2_amd: this process was fork()ed from one that had already opened the GPU; the HIP runtime it "
                              "inherited cannot be used or re-initialised here (the per-call API goes through crispresso2_amd._helper "
                              "instead; anything else needs a spawned process)")
        self.lib = load()
        self.handle = ctypes.c_void_p()
        rc = self.lib.c2_create(int(device), ctypes.byref(self.handle))
        if rc != 0:
            msg = self.lib.c2_last_error(None)
            raise NativeError("crispresso2_amd: cannot create a GPU context on device %d: %s (rc=%d); "
                              "this package has no CPU fallback" % (device, msg.decode() if msg else "?", rc))
        self.device = int(device)
        self.pid = os.getpid()
        note_gpu_opened()

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.c2_last_error(self.handle)
            raise NativeError("%s failed: %s (rc=%d)" % (what, msg.decode() if msg else "?", rc))

    def close(self):
        if self.handle:
            if getattr(self, "pid", os.getpid()) == os.getpid():   # (a forked child must not tear down the parent's streams and buffers)
                self.lib.c2_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scoring / refs
    def set_scoring(self, matrix, gap_open, gap_extend):
        m = np.ascontiguousarray(matrix, dtype=np.int64)
        if m.ndim != 2 or m.shape[0] != m.shape[1]:
            raise ValueError("score matrix must be square")
        self.check(self.lib.c2_set_scoring(self.handle, m.ctypes.data_as(ctypes.c_void_p), int(m.shape[0]),
                                           int(gap_open), int(gap_extend)), "c2_set_scoring")

    def set_refs(self, seqs, gap_incentives, include_idxs):
        n = len(seqs)
        bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in seqs]
        arr = (ctypes.c_char_p * n)(*bs)
        lens = np.array([len(b) for b in bs], dtype=np.int32)
        g = [np.ascontiguousarray(x, dtype=np.int64) for x in gap_incentives]
        for k in range(n):
            if g[k].shape[0] != lens[k] + 1:
                raise ValueError("gap_incentive of reference %d must have len(ref)+1 entries" % k)
        gp = (ctypes.c_void_p * n)(*[x.ctypes.data for x in g])
        inc = [np.ascontiguousarray(np.asarray(list(x), dtype=np.int64).astype(np.int32)) for x in include_idxs]
        ip = (ctypes.c_void_p * n)(*[x.ctypes.data if x.size else None for x in inc])
        ninc = np.array([x.size for x in inc], dtype=np.int32)
        self.check(self.lib.c2_set_refs(self.handle, n, arr, lens.ctypes.data_as(ctypes.c_void_p), gp, ip,
                                        ninc.ctypes.data_as(ctypes.c_void_p)), "c2_set_refs")

    # ---- batch
    def align_classify_device(self, batch, stream=None):
        self.check(self.lib.c2_align_classify_batch_device(self.handle, ctypes.byref(batch),
                                                           ctypes.c_void_p(stream or 0)), "c2_align_classify_batch_device")

    def align_classify_host(self, batch):
        self.check(self.lib.c2_align_classify_batch_host(self.handle, ctypes.byref(batch)), "c2_align_classify_batch_host")

    def synchronize(self, stream=None):
        self.check(self.lib.c2_synchronize(self.handle, ctypes.c_void_p(stream or 0)), "c2_synchronize")

    def timing_enable(self, on=True):
        self.check(self.lib.c2_timing_enable(self.handle, int(bool(on))), "c2_timing_enable")

    def timing_read(self, reset=True):
        ms = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        self.check(self.lib.c2_timing_read(self.handle, ctypes.byref(ms), ctypes.byref(n), int(bool(reset))), "c2_timing_read")
        return ms.value, n.value

    def timing_read_split(self, reset=True):
        """-> (ms of the whole launch chains, ms of their first kernels, batches)"""
        ms, first = ctypes.c_double(0), ctypes.c_double(0)
        n = ctypes.c_int64(0)
        self.check(self.lib.c2_timing_read_split(self.handle, ctypes.byref(ms), ctypes.byref(first), ctypes.byref(n), int(bool(reset))),
                   "c2_timing_read_split")
        return ms.value, first.value, n.value

    def set_band(self, band_lanes=-1, target_workgroups_per_cu=0):
        self.check(self.lib.c2_set_band(self.handle, int(band_lanes), int(target_workgroups_per_cu)), "c2_set_band")

    def set_kernel_mode(self, mode):
        """'auto' (diagonal-band tiers: 8 alignments per wavefront in int16 pairs where the reference admits it, else 4; then 2; then
        1), 'diag4' (tiers 4 -> 2 -> 1, 32-bit cells only), 'diag2' (2 -> 1), 'diag1' (single-alignment diagonal-band kernel),
        'band' (banded row-strip), 'full' (full-plane row-strip)"""
        code = {'auto': 0, 'band': 1, 'full': 2, 'diag1': 3, 'diag2': 4, 'diag4': 5}.get(mode, mode)
        self.check(self.lib.c2_set_kernel_mode(self.handle, int(code)), "c2_set_kernel_mode")

    def band_info(self, max_read_len):
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        self.check(self.lib.c2_band_info(self.handle, int(max_read_len), ctypes.byref(a), ctypes.byref(b)), "c2_band_info")
        return {"band_lanes": a.value, "fallback_tasks_last_launch": b.value}

    def classify_lists_batch(self, aln_read, aln_ref, lens, set_ids, include_sets, legacy=False):
        """c2_classify_lists_batch: aln_read / aln_ref uint8 [n, stride] (host), lens int32 [n], set_ids uint16 [n] or None,
        include_sets: list of integer sequences.  -> (index int64 [n*15+1], values int32, counts int64 [n, 3])"""
        a1 = np.ascontiguousarray(aln_read, dtype=np.uint8)
        a2 = np.ascontiguousarray(aln_ref, dtype=np.uint8)
        n, stride = a1.shape
        ln = np.ascontiguousarray(lens, dtype=np.int32)
        ids = None if set_ids is None else np.ascontiguousarray(set_ids, dtype=np.uint16)
        sets = [np.asarray(list(x), dtype=np.int64).astype(np.int32) for x in include_sets]
        off = np.zeros(len(sets) + 1, dtype=np.int64)
        off[1:] = np.cumsum([x.size for x in sets])
        flat = np.ascontiguousarray(np.concatenate(sets) if sets and off[-1] else np.zeros(1, dtype=np.int32), dtype=np.int32)
        h = ctypes.c_void_p()
        self.check(self.lib.c2_classify_lists_batch(
            self.handle, ctypes.c_uint64(n), a1.ctypes.data_as(ctypes.c_void_p), a2.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_uint32(stride), ln.ctypes.data_as(ctypes.c_void_p),
            None if ids is None else ids.ctypes.data_as(ctypes.c_void_p), flat.ctypes.data_as(ctypes.c_void_p),
            off.ctypes.data_as(ctypes.c_void_p), len(sets), int(bool(legacy)), ctypes.byref(h)), "c2_classify_lists_batch")
        try:
            tot = int(self.lib.c2_lists_total(h))
            index = np.ctypeslib.as_array(ctypes.cast(self.lib.c2_lists_index(h), ctypes.POINTER(ctypes.c_int64)), (n * LIST_COUNT + 1,)).copy()
            values = (np.ctypeslib.as_array(ctypes.cast(self.lib.c2_lists_values(h), ctypes.POINTER(ctypes.c_int32)), (tot,)).copy()
                      if tot else np.zeros(0, dtype=np.int32))
            counts = (np.ctypeslib.as_array(ctypes.cast(self.lib.c2_lists_counts(h), ctypes.POINTER(ctypes.c_int64)), (n, 3)).copy()
                      if n else np.zeros((0, 3), dtype=np.int64))
        finally:
            self.lib.c2_lists_free(h)
        return index, values, counts

    # ---- multi-GPU exchange through the C ABI (RCCL); torch.distributed is only used to hand the communicator id around
    def comm_init(self, rank=None, world=None):
        """Collective: every rank calls it.  The 128-byte id comes from rank 0 (c2_comm_unique_id) and travels through
        torch.distributed's object broadcast when a process group exists (world > 1)."""
        rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        buf = (ctypes.c_uint8 * 128)()
        if rank == 0:
            self.check(self.lib.c2_comm_unique_id(buf), "c2_comm_unique_id")
        if world > 1:
            import torch.distributed as dist
            box = [bytes(buf)]
            dist.broadcast_object_list(box, src=0)
            buf = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        self.check(self.lib.c2_comm_init(self.handle, rank, world, buf), "c2_comm_init")

    def reduce_counts(self, d_counts, n_elements, stream=None):
        """in-place sum over the ranks of n_elements int64 at the device address d_counts (c2_reduce_counts)"""
        self.check(self.lib.c2_reduce_counts(self.handle, ctypes.c_void_p(d_counts), ctypes.c_uint64(n_elements), ctypes.c_void_p(stream or 0)),
                   "c2_reduce_counts")

    def tier_info(self):
        """Banded launches of the most recent batch and the number of tasks each left for the next one."""
        n = ctypes.c_int32(0)
        left = (ctypes.c_int32 * 4)()
        self.check(self.lib.c2_tier_info(self.handle, ctypes.byref(n), left), "c2_tier_info")
        return [int(left[k]) for k in range(n.value)]

    CHAIN_KERNELS = ["c2_align_diagp_kernel<8>", "c2_align_diagx_kernel<4>", "c2_align_diagp_kernel<4>", "c2_align_diagx_kernel<2>",
                     "c2_align_diagp_kernel<2>", "c2_align_diag_kernel", "banded row-strip", "full plane in HBM scratch", "packed fill with 32-bit adds",
                     "score-only stage (c2_align_partition_kernel + c2_align_diags_kernel<16>) in front of the first band tier"]

    def chain_info(self, max_read_len, n_refs):
        """-> (names of the kernels in the launch chain for reads up to max_read_len, [packed fill admits reference r])"""
        kern = ctypes.c_uint32(0)
        ok = (ctypes.c_uint8 * max(n_refs, 1))()
        self.check(self.lib.c2_chain_info(self.handle, int(max_read_len), ctypes.byref(kern), ok), "c2_chain_info")
        return [nm for b, nm in enumerate(self.CHAIN_KERNELS) if kern.value >> b & 1], [bool(ok[r]) for r in range(n_refs)]

    def score_stage_info(self):
        """-> (ran, tasks taken, tasks finished) of the score-only stage of the most recent batch (c2_score_stage_info)"""
        ran, t, f = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self.check(self.lib.c2_score_stage_info(self.handle, ctypes.byref(ran), ctypes.byref(t), ctypes.byref(f)), "c2_score_stage_info")
        return bool(ran.value), int(t.value), int(f.value)

    def partition_info(self):
        """-> {"ran", "p16", "classes": tasks per class [score-only, 14 diagonals (opt-in), 32 diagonals, 40, 62, 126/128, full matrix],
        "finished": [score-only, 14 diagonals], "finished_by_partition": class-0 reads on the main diagonal (a copy of the reference, or one / two differing bases)
        that the partition finished itself} of the most recent batch (c2_partition_info, c2_partition_finished)"""
        ran = ctypes.c_int32(0)
        cls, fin = (ctypes.c_int64 * 7)(), (ctypes.c_int64 * 2)()
        self.check(self.lib.c2_partition_info(self.handle, ctypes.byref(ran), cls, fin), "c2_partition_info")
        exact = ctypes.c_int64(0)
        self.check(self.lib.c2_partition_finished(self.handle, ctypes.byref(exact)), "c2_partition_finished")
        return {"ran": bool(ran.value & 1), "p16": bool(ran.value & 2), "classes": [int(x) for x in cls], "finished": [int(x) for x in fin],
                "finished_by_partition": int(exact.value)}

    def tier_info_ex(self):
        """-> (left_over, unpaired) per band tier of the most recent batch (c2_tier_info_ex)."""
        n = ctypes.c_int32(0)
        left, un = (ctypes.c_int32 * 8)(), (ctypes.c_int32 * 8)()
        self.check(self.lib.c2_tier_info_ex(self.handle, ctypes.byref(n), left, un), "c2_tier_info_ex")
        return [int(left[k]) for k in range(n.value)], [int(un[k]) for k in range(n.value)]

    def phase_profile(self, enable):
        """-> the four per-phase cycle counters accumulated so far (then cleared); sets the mode."""
        out = (ctypes.c_uint64 * 4)()
        self.check(self.lib.c2_phase_profile(self.handle, int(bool(enable)), out), "c2_phase_profile")
        return [int(x) for x in out]

    def launch_info(self, max_read_len):
        v = [ctypes.c_int32(0) for _ in range(5)]
        self.check(self.lib.c2_launch_info(self.handle, int(max_read_len), *[ctypes.byref(x) for x in v]), "c2_launch_info")
        return dict(zip(("rows_per_lane", "passes", "lds_bytes", "workgroups_per_cu", "compute_units"), [x.value for x in v]))


def gz_parallel_last():
    """c2_gz_parallel_last: what the one-member-on-all-threads route did for the last .gz file opened on this thread"""
    st = (ctypes.c_uint64 * 8)()
    load().c2_gz_parallel_last(st)
    return dict(segments=int(st[0]), block_starts_found=int(st[1]), text_bytes=int(st[2]), fell_back=bool(st[3]),
                seconds=dict(search=st[4] / 1e6, first_pass=st[5] / 1e6, windows=st[6] / 1e6, second_pass=st[7] / 1e6))


def fastq_unique(path, min_single_bp_quality=0, min_average_read_quality=0, min_bp_quality_or_N=0, stats=None):
    """c2_fastq_unique (host code, needs no GPU): -> (arena uint8, offsets uint64 [n_unique+1], counts uint32 [n_unique], n_reads)
    -- the unique sequences of the FASTQ in first-seen order, packed as the align kernels take them.  With any of the three
    quality options (> 0) the reference's read filter (filterFastqs.py) runs fused in front, n_reads is the number of reads
    that passed, and `stats` (a dict) receives N_READS_INPUT as the reference counts it."""
    lib = load()
    h = ctypes.c_void_p()
    if min_single_bp_quality > 0 or min_average_read_quality > 0 or min_bp_quality_or_N > 0:
        lines = ctypes.c_uint64(0)
        rc = lib.c2_fastq_unique_filtered(os.fsencode(path), int(min_single_bp_quality), int(min_average_read_quality),
                                          int(min_bp_quality_or_N), ctypes.byref(h), ctypes.byref(lines))
        if rc == 0 and stats is not None:
            stats["N_READS_INPUT"] = int(float(lines.value) / 4.0)          # get_n_reads_fastq, CRISPRessoShared.py:746-747
    else:
        rc = lib.c2_fastq_unique(os.fsencode(path), ctypes.byref(h))
    if rc != 0:
        raise NativeError("c2_fastq_unique: %s" % lib.c2_fastq_last_error().decode())
    try:
        n = int(lib.c2_fastq_n_unique(h))
        nb = int(lib.c2_fastq_arena_bytes(h))
        arena = (np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_arena(h), ctypes.POINTER(ctypes.c_uint8)), (nb,)).copy()
                 if nb else np.zeros(0, dtype=np.uint8))
        offsets = np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_offsets(h), ctypes.POINTER(ctypes.c_uint64)), (n + 1,)).copy()
        counts = (np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_counts(h), ctypes.POINTER(ctypes.c_uint32)), (n,)).copy()
                  if n else np.zeros(0, dtype=np.uint32))
        total = int(lib.c2_fastq_n_reads(h))
        if stats is not None:
            _line_stats(stats, int(lib.c2_fastq_nonempty_lines(h)))
    finally:
        lib.c2_fastq_free(h)
    return arena, offsets, counts, total


def _line_stats(stats, parsed_nonempty_lines):
    """N_READS_AFTER_PREPROCESSING = get_n_reads_fastq of the text that was parsed (CRISPRessoCORE.py:3723-3727: the filtered file,
    or the input itself -- then N_READS_INPUT is the same number, :3539)."""
    n = int(float(parsed_nonempty_lines) / 4.0)
    stats["N_READS_AFTER_PREPROCESSING"] = n
    stats.setdefault("N_READS_INPUT", n)


class BgzfFile:
    """c2_bgzf_*: a BGZF file whose members are inflated range by range into memory the caller names (fastq_device uploads the text
    while it is inflated).  NativeError "not a BGZF file" for anything else."""

    def __init__(self, path):
        lib = load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        if lib.c2_bgzf_open(os.fsencode(path), ctypes.byref(self._h)) != 0:
            raise NativeError("c2_bgzf_open: %s" % lib.c2_fastq_last_error().decode())
        self.n_blocks = int(lib.c2_bgzf_n_blocks(self._h))
        ptr = lib.c2_bgzf_text_offsets(self._h)
        self.text_offsets = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), (self.n_blocks + 1,)).copy()
        self.text_bytes = int(self.text_offsets[-1])

    def inflate(self, b0, b1, dst_address, cap, threads=0):
        """members [b0, b1) -> memory at dst_address (ctypes drops the GIL; `threads` native threads)"""
        if self._lib.c2_bgzf_inflate(self._h, ctypes.c_uint64(b0), ctypes.c_uint64(b1), ctypes.c_void_p(dst_address), ctypes.c_uint64(cap),
                                     int(threads)) != 0:
            raise NativeError("c2_bgzf_inflate: %s" % self._lib.c2_fastq_last_error().decode())

    def close(self):
        if self._h:
            self._lib.c2_bgzf_close(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class GzSegFile(BgzfFile):
    """c2_gzseg_open: ONE ordinary gzip member cut into segments at deflate block starts (c2_gz_parallel.h) -- the same interface as a BGZF
    file's members (n_blocks, text_offsets, inflate(b0, b1, ...)), so fastq_device inflates it range by range straight into the upload buffers.
    NativeError "not applicable ..." for anything that is not one clean member of a few megabytes and more."""

    def __init__(self, path, threads=0, chunk_bytes=0):
        lib = load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        if lib.c2_gzseg_open(os.fsencode(path), int(threads), ctypes.c_uint64(chunk_bytes), ctypes.byref(self._h)) != 0:
            raise NativeError("c2_gzseg_open: %s" % lib.c2_fastq_last_error().decode())
        self.n_blocks = int(lib.c2_bgzf_n_blocks(self._h))
        ptr = lib.c2_bgzf_text_offsets(self._h)
        self.text_offsets = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), (self.n_blocks + 1,)).copy()
        self.text_bytes = int(self.text_offsets[-1])


class FastqStream:
    """c2_fastq_stream_*: the ingest of FastqUnique chunk by chunk.  next() parses one more chunk (all host threads; ctypes drops
    the GIL) and returns (unique reads so far, done); everything below that mark is final -- `arena` (one view over the whole
    reservation: the pointer never moves), offsets_slice(a, b) (copied out: the native array may move at the next call), first-seen
    order.  counts() are the multiplicities so far (final once done).  Same statistics as FastqUnique when done."""

    def __init__(self, path, min_single_bp_quality=0, min_average_read_quality=0, min_bp_quality_or_N=0):
        lib = load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        rc = lib.c2_fastq_stream_open(os.fsencode(path), int(min_single_bp_quality), int(min_average_read_quality), int(min_bp_quality_or_N),
                                      ctypes.byref(self._h))
        if rc != 0:
            raise NativeError("c2_fastq_stream_open: %s" % lib.c2_fastq_last_error().decode())
        self.filtered = min_single_bp_quality > 0 or min_average_read_quality > 0 or min_bp_quality_or_N > 0
        self.text_bytes = int(lib.c2_fastq_stream_text_bytes(self._h))
        ptr = lib.c2_fastq_stream_arena(self._h)
        self.arena = (np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), (self.text_bytes + 1,))
                      if ptr else np.zeros(1, dtype=np.uint8))
        self.n_unique, self.arena_bytes, self.done = 0, 0, self.text_bytes == 0

    def text(self):
        """the text next() would parse, as a read-only view of native memory (inflated / filtered / mapped input), or None (a plain file
        that is read chunk by chunk); valid until close()"""
        ptr = self._lib.c2_fastq_stream_text(self._h)
        if not ptr or not self.text_bytes:
            return None
        a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), (self.text_bytes,))
        a.flags.writeable = False
        return a

    def lines_input(self):
        """non-empty lines of the text in front of the read filter (filtered streams)"""
        return int(self._lib.c2_fastq_stream_nonempty_lines_input(self._h))

    def next(self):
        nu, ab, dn = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int32(0)
        rc = self._lib.c2_fastq_stream_next(self._h, ctypes.byref(nu), ctypes.byref(ab), ctypes.byref(dn))
        if rc != 0:
            raise NativeError("c2_fastq_stream_next: %s" % self._lib.c2_fastq_last_error().decode())
        self.n_unique, self.arena_bytes, self.done = int(nu.value), int(ab.value), bool(dn.value)
        return self.n_unique, self.done

    def offsets_slice(self, a, b):
        """offsets[a : b + 1] of the unique reads a .. b - 1 (a copy)"""
        ptr = self._lib.c2_fastq_stream_offsets(self._h)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), (self.n_unique + 1,))[a:b + 1].copy()

    def counts(self):
        out = np.zeros(self.n_unique, dtype=np.uint32)
        rc = self._lib.c2_fastq_stream_counts(self._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(self.n_unique))
        if rc != 0:
            raise NativeError("c2_fastq_stream_counts: %s" % self._lib.c2_fastq_last_error().decode())
        return out

    def rc_partners(self):
        """-> int64 [n_unique]: index of the unique read equal to reverse_complement(read i), or -1 (rc_partners(), from the ingest's own
        table; the stream must be done).  ctypes drops the GIL: a host thread can run it while the device works."""
        out = np.empty(self.n_unique, dtype=np.int64)
        rc = self._lib.c2_fastq_stream_rc_partners(self._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(self.n_unique))
        if rc != 0:
            raise NativeError("c2_fastq_stream_rc_partners: %s" % self._lib.c2_fastq_last_error().decode())
        return out

    @property
    def n_reads(self):
        return int(self._lib.c2_fastq_stream_n_reads(self._h))

    def line_stats(self, stats):
        if self.filtered:
            stats["N_READS_INPUT"] = int(float(self._lib.c2_fastq_stream_nonempty_lines_input(self._h)) / 4.0)
        _line_stats(stats, int(self._lib.c2_fastq_stream_nonempty_lines(self._h)))

    def close(self):
        if self._h:
            big = self.arena_bytes > (32 << 20)
            self.arena = None
            h, self._h = self._h, ctypes.c_void_p()
            if big and not os.environ.get("C2_SYNC_FREE"):
                import threading
                threading.Thread(target=self._lib.c2_fastq_stream_close, args=(h,), name="c2-fastq-free", daemon=False).start()
            else:
                self._lib.c2_fastq_stream_close(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastqUnique:
    """c2_fastq_unique without the copies: arena / offsets / counts are VIEWS of the native handle's memory (211 MB for 845 k
    unique 250-bp reads), valid until close().  Same arguments and statistics as fastq_unique()."""

    def __init__(self, path, min_single_bp_quality=0, min_average_read_quality=0, min_bp_quality_or_N=0, stats=None):
        lib = load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        if min_single_bp_quality > 0 or min_average_read_quality > 0 or min_bp_quality_or_N > 0:
            lines = ctypes.c_uint64(0)
            rc = lib.c2_fastq_unique_filtered(os.fsencode(path), int(min_single_bp_quality), int(min_average_read_quality),
                                              int(min_bp_quality_or_N), ctypes.byref(self._h), ctypes.byref(lines))
            if rc == 0 and stats is not None:
                stats["N_READS_INPUT"] = int(float(lines.value) / 4.0)          # get_n_reads_fastq, CRISPRessoShared.py:746-747
        else:
            rc = lib.c2_fastq_unique(os.fsencode(path), ctypes.byref(self._h))
        if rc != 0:
            raise NativeError("c2_fastq_unique: %s" % lib.c2_fastq_last_error().decode())
        h = self._h
        n = int(lib.c2_fastq_n_unique(h))
        nb = int(lib.c2_fastq_arena_bytes(h))
        self.arena = (np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_arena(h), ctypes.POINTER(ctypes.c_uint8)), (nb,))
                      if nb else np.zeros(0, dtype=np.uint8))
        self.offsets = np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_offsets(h), ctypes.POINTER(ctypes.c_uint64)), (n + 1,))
        self.counts = (np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_counts(h), ctypes.POINTER(ctypes.c_uint32)), (n,))
                       if n else np.zeros(0, dtype=np.uint32))
        self.n_reads = int(lib.c2_fastq_n_reads(h))
        if stats is not None:
            _line_stats(stats, int(lib.c2_fastq_nonempty_lines(h)))

    def close(self):
        """Releases the native memory.  Large arenas are handed to a helper thread (unmapping a few hundred megabytes takes tens
        of milliseconds of kernel time that nobody has to wait for; ctypes drops the GIL for the call)."""
        if self._h:
            big = self.arena is not None and self.arena.size > (32 << 20)
            self.arena = self.offsets = self.counts = None
            h, self._h = self._h, ctypes.c_void_p()
            if big and not os.environ.get("C2_SYNC_FREE"):
                import threading
                threading.Thread(target=self._lib.c2_fastq_free, args=(h,), name="c2-fastq-free", daemon=False).start()
            else:
                self._lib.c2_fastq_free(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _fastq_strings(lib, h, n, data_fn, bytes_fn, offsets_fn):
    nb = int(bytes_fn(h))
    if n == 0:
        return []
    buf = ctypes.string_at(data_fn(h), nb) if nb else b""
    off = np.ctypeslib.as_array(ctypes.cast(offsets_fn(h), ctypes.POINTER(ctypes.c_uint64)), (n + 1,))
    return [buf[int(off[k]):int(off[k + 1])].decode('utf-8') for k in range(n)]


def _fastq_arrays(lib, h, n):
    """the arena / offsets and aux / aux offsets of a c2_fastq handle as numpy views"""
    def view(data_fn, bytes_fn, offsets_fn):
        nb = int(bytes_fn(h))
        a = np.ctypeslib.as_array(ctypes.cast(data_fn(h), ctypes.POINTER(ctypes.c_uint8)), (nb,)) if nb else np.zeros(0, dtype=np.uint8)
        o = np.ctypeslib.as_array(ctypes.cast(offsets_fn(h), ctypes.POINTER(ctypes.c_uint64)), (n + 1,)) if n else np.zeros(1, dtype=np.uint64)
        return a, o
    ka, ko = view(lib.c2_fastq_arena, lib.c2_fastq_arena_bytes, lib.c2_fastq_offsets)
    qa, qo = view(lib.c2_fastq_aux, lib.c2_fastq_aux_bytes, lib.c2_fastq_aux_offsets)
    return ka, ko, qa, qo


class PairedFastq:
    """c2_fastq_unique_paired (host code, needs no GPU): the unique read pairs of two FASTQ files read in lockstep.
    keys[k] = seq1 + '+' + reverse_complement(seq2), counts[k] copies, quals[k] = qual1 + ' ' + qual2[::-1] of the first
    occurrence, all in first-seen order; n_pairs = records read.  occurrences(selected) re-reads the files."""

    def __init__(self, path1, path2):
        lib = load()
        self._lib, self._paths = lib, (os.fsencode(path1), os.fsencode(path2))
        self._h = ctypes.c_void_p()
        rc = lib.c2_fastq_unique_paired(self._paths[0], self._paths[1], ctypes.byref(self._h))
        if rc != 0:
            msg = lib.c2_fastq_last_error().decode()
            if msg.startswith("KeyError"):
                raise KeyError(msg)
            raise NativeError("c2_fastq_unique_paired: %s" % msg)
        h = self._h
        n = int(lib.c2_fastq_n_unique(h))
        self.n_unique = n
        self._keys = self._quals = None                               # (Python strings only for a caller that asks for them)
        self.counts = (np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_counts(h), ctypes.POINTER(ctypes.c_uint32)), (n,)).copy()
                       if n else np.zeros(0, dtype=np.uint32))
        self.n_pairs = int(lib.c2_fastq_n_reads(h))

    @property
    def keys(self):
        if self._keys is None:
            self._keys = _fastq_strings(self._lib, self._h, self.n_unique, self._lib.c2_fastq_arena, self._lib.c2_fastq_arena_bytes, self._lib.c2_fastq_offsets)
        return self._keys

    @property
    def quals(self):
        if self._quals is None:
            self._quals = _fastq_strings(self._lib, self._h, self.n_unique, self._lib.c2_fastq_aux, self._lib.c2_fastq_aux_bytes, self._lib.c2_fastq_aux_offsets)
        return self._quals

    def arrays(self):
        """-> (key bytes uint8, key offsets uint64 [n + 1], quality bytes uint8, quality offsets uint64 [n + 1]): views of the native arenas
        (valid until close()) -- the keys and quality pairs without a Python string per pair"""
        return _fastq_arrays(self._lib, self._h, self.n_unique)

    def occurrences(self, selected, as_arrays=False):
        """selected: bool per key -> (key index per occurrence, quality pair per occurrence), in file order.
        as_arrays: the quality pairs as (bytes uint8, offsets uint64 [m + 1]) instead of strings."""
        lib = self._lib
        sel = np.ascontiguousarray(selected, dtype=np.uint8)
        if sel.shape != (self.n_unique,):
            raise ValueError("one flag per key")
        out = ctypes.c_void_p()
        rc = lib.c2_fastq_paired_occurrences(self._paths[0], self._paths[1], self._h, sel.ctypes.data_as(ctypes.c_void_p), ctypes.byref(out))
        if rc != 0:
            raise NativeError("c2_fastq_paired_occurrences: %s" % lib.c2_fastq_last_error().decode())
        try:
            m = int(lib.c2_fastq_n_unique(out))
            idx = (np.ctypeslib.as_array(ctypes.cast(lib.c2_fastq_counts(out), ctypes.POINTER(ctypes.c_uint32)), (m,)).copy()
                   if m else np.zeros(0, dtype=np.uint32))
            if as_arrays:
                _, _, qa, qo = _fastq_arrays(lib, out, m)
                quals = (qa.copy(), qo.copy())
            else:
                quals = _fastq_strings(lib, out, m, lib.c2_fastq_aux, lib.c2_fastq_aux_bytes, lib.c2_fastq_aux_offsets)
        finally:
            lib.c2_fastq_free(out)
        return idx, quals

    def close(self):
        if self._h:
            self._lib.c2_fastq_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def strand_plan(arena, offsets, fw_seeds, rc_seeds, seed_min):
    """c2_strand_plan for one reference -> uint8 [n] (0 forward, 1 reverse complement, 2 both)."""
    lib = load()
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.zeros(n, dtype=np.uint8)
    m = len(fw_seeds)
    fw = (ctypes.c_char_p * max(m, 1))(*[s.encode() for s in fw_seeds])
    rc = (ctypes.c_char_p * max(m, 1))(*[s.encode() for s in rc_seeds])
    rcode = lib.c2_strand_plan(arena.ctypes.data_as(ctypes.c_void_p) if arena.size else None, offsets.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_uint64(n), fw, rc, int(m), int(seed_min), out.ctypes.data_as(ctypes.c_void_p))
    if rcode != 0:
        raise NativeError("c2_strand_plan: %s" % lib.c2_fastq_last_error().decode())
    return out


def merge_reverse_complements(arena, offsets, aligned, counts):
    """c2_merge_reverse_complements, in place on counts (int64 [n])."""
    lib = load()
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    al = np.ascontiguousarray(aligned, dtype=np.uint8)
    assert counts.dtype == np.int64 and counts.flags["C_CONTIGUOUS"]
    rcode = lib.c2_merge_reverse_complements(arena.ctypes.data_as(ctypes.c_void_p) if arena.size else None,
                                             offsets.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(offsets) - 1),
                                             al.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p))
    if rcode != 0:
        raise NativeError("c2_merge_reverse_complements: %s" % lib.c2_fastq_last_error().decode())
    return counts


def rc_partners(arena, offsets):
    """c2_rc_partners -> int64 [n]: index of the read equal to reverse_complement(read i), or -1.  (ctypes releases the GIL for the
    call: pipeline.quantify_unique runs it on a thread while the device aligns.)"""
    lib = load()
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.empty(n, dtype=np.int64)
    rcode = lib.c2_rc_partners(arena.ctypes.data_as(ctypes.c_void_p) if arena.size else None, offsets.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_uint64(n), out.ctypes.data_as(ctypes.c_void_p))
    if rcode != 0:
        raise NativeError("c2_rc_partners: %s" % lib.c2_fastq_last_error().decode())
    return out


def gather_reads(arena, offsets, idx):
    """c2_gather_reads -> (uint8 arena of the reads idx[...] back to back, uint64 offsets [len(idx) + 1])"""
    lib = load()
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    m = len(idx)
    off = np.zeros(m + 1, dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    if lib.c2_gather_reads(P(arena) if arena.size else None, P(offsets), P(idx), ctypes.c_uint64(m), None, P(off)) != 0:
        raise NativeError("c2_gather_reads: %s" % lib.c2_fastq_last_error().decode())
    out = np.empty(max(int(off[-1]), 1), dtype=np.uint8)
    if lib.c2_gather_reads(P(arena) if arena.size else None, P(offsets), P(idx), ctypes.c_uint64(m), P(out), P(off)) != 0:
        raise NativeError("c2_gather_reads: %s" % lib.c2_fastq_last_error().decode())
    return out[:int(off[-1])], off


def merge_counts_with_partners(aligned, partner, counts):
    """c2_merge_counts_with_partners, in place on counts (int64 [n])."""
    lib = load()
    al = np.ascontiguousarray(aligned, dtype=np.uint8)
    pt = np.ascontiguousarray(partner, dtype=np.int64)
    assert counts.dtype == np.int64 and counts.flags["C_CONTIGUOUS"] and len(pt) == len(counts) == len(al)
    rcode = lib.c2_merge_counts_with_partners(ctypes.c_uint64(len(counts)), al.ctypes.data_as(ctypes.c_void_p),
                                              pt.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p))
    if rcode != 0:
        raise NativeError("c2_merge_counts_with_partners: %s" % lib.c2_fastq_last_error().decode())
    return counts


_default_ctx = None


def default_context():
    """Process-wide context on the current device (LOCAL_RANK, else 0) for the per-call API.  In a process that was fork()ed
    after the GPU had been opened (the reference's `-p N` workers, CRISPRessoCORE.py:1870-1898: main() has aligned its guides
    before it forks) this raises -- the shim modules send their per-call work to `forked_child_helper()` there."""
    global _default_ctx
    if in_forked_child():
        raise NativeError("crispresso2_amd: no GPU context in a fork()ed child of a process that had opened the GPU")
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


# ---- fork(): HIP does not survive it.  The pid that first opened the GPU is remembered; any other pid that finds the mark is a forked
# child (a spawned process imports this module afresh and has no mark).
_gpu_pid = [None]
_helper = [None, None]                                                 # (pid that owns it, _Helper)


def note_gpu_opened():
    if _gpu_pid[0] is None:
        _gpu_pid[0] = os.getpid()


def in_forked_child():
    return _gpu_pid[0] is not None and _gpu_pid[0] != os.getpid()


class _Helper:
    """A SPAWNED python process with its own HIP runtime that runs the per-call entry points for a forked child
    (`python -m crispresso2_amd._helper <read fd> <write fd>`; length-prefixed pickles either way)."""

    def __init__(self):
        import subprocess
        import sys
        c2h_r, c2h_w = os.pipe()
        h2c_r, h2c_w = os.pipe()
        env = dict(os.environ, C2_PRIME_FROM_ARGV="0")
        env.pop("C2_PRIME_FASTQ", None)
        root = os.path.dirname(_HERE)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        self.proc = subprocess.Popen([sys.executable, "-m", "crispresso2_amd._helper", str(c2h_r), str(h2c_w)], pass_fds=(c2h_r, h2c_w),
                                     env=env, stdin=subprocess.DEVNULL)
        os.close(c2h_r)
        os.close(h2c_w)
        self.w = os.fdopen(c2h_w, "wb", buffering=0)
        self.r = os.fdopen(h2c_r, "rb", buffering=0)
        self.calls = 0

    def call(self, name, *args):
        import pickle
        import struct
        blob = pickle.dumps((name, args), protocol=pickle.HIGHEST_PROTOCOL)
        self.w.write(struct.pack("<Q", len(blob)) + blob)
        head = _read_exactly(self.r, 8)
        if head is None:
            raise NativeError("crispresso2_amd: the helper process of this forked child ended (exit code %s)" % self.proc.poll())
        kind, value = pickle.loads(_read_exactly(self.r, struct.unpack("<Q", head)[0]))
        self.calls += 1
        if kind == "err":
            raise value
        return value

    def close(self):
        try:
            self.w.close()
            self.r.close()
            self.proc.wait(timeout=30)
        except Exception:
            pass


def _read_exactly(fh, n):
    out = b""
    while len(out) < n:
        part = fh.read(n - len(out))
        if not part:
            return None
        out += part
    return out


def forked_child_helper():
    """the helper of THIS process (a grandchild forked from a forked child starts its own)"""
    if _helper[0] != os.getpid():
        _helper[0], _helper[1] = os.getpid(), _Helper()
        import atexit
        atexit.register(_helper[1].close)
    return _helper[1]
