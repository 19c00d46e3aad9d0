"""Per-amplicon count tensor: layout, device accumulation and multi-GPU reduction.

The tensor holds, per reference amplicon, what the reference's "Quantifying indels/substitutions" loop builds in Python
dicts of numpy vectors (CRISPRessoCORE.py:3865-3901 initialised, :3964-4115 filled) plus process_fastq's aln_stats
(:1974-1979).  c2_count_vectors_kernel fills it on the GPU from the aligned strings that are already in HBM; across GPUs it
is summed with ONE all-reduce (RCCL over xGMI) -- the only exchange step of the sharded path (reads are independent).
"""
import ctypes

import numpy as np

N_VECTORS, N_SCALARS, N_HISTS = 20, 24, 4
VECTORS = ["all_insertion_count_vectors", "all_insertion_left_count_vectors", "all_deletion_count_vectors",
           "all_substitution_count_vectors", "insertion_count_vectors", "deletion_count_vectors",
           "substitution_count_vectors",
           "all_substitution_base_vectors_A", "all_substitution_base_vectors_C", "all_substitution_base_vectors_G",
           "all_substitution_base_vectors_T", "all_substitution_base_vectors_N",
           "all_base_count_vectors_A", "all_base_count_vectors_C", "all_base_count_vectors_G", "all_base_count_vectors_T",
           "all_base_count_vectors_N", "all_base_count_vectors_-",
           "insertion_length_vectors", "deletion_length_vectors"]
SCALARS = ["counts_total", "counts_modified", "counts_unmodified", "counts_discarded", "counts_insertion", "counts_deletion",
           "counts_substitution", "counts_only_insertion", "counts_only_deletion", "counts_only_substitution",
           "counts_insertion_and_deletion", "counts_insertion_and_substitution", "counts_deletion_and_substitution",
           "counts_insertion_and_deletion_and_substitution",
           "N_GLOBAL_SUBS", "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS",
           "alignments_counted"]
HISTS = ["inserted_n", "deleted_n", "substituted_n", "effective_len"]
FLAG_IGNORE_SUBSTITUTIONS, FLAG_IGNORE_INSERTIONS, FLAG_IGNORE_DELETIONS, FLAG_DISCARD_INDEL_READS = 1, 2, 4, 8
FLAG_ALL_REFS_LAYOUT = 16        # the tasks are one all-references batch (task = read * n_refs + reference)
FLAG_LEGACY_CLASSIFIER = 32      # --use_legacy_insertion_quantification: positions of find_indels_substitutions_legacy
assert len(VECTORS) == N_VECTORS and len(HISTS) == N_HISTS


class CountLayout:
    def __init__(self, n_refs, lmax, max_read_len):
        self.n_refs, self.lmax = int(n_refs), int(lmax)
        self.hl = self.lmax + int(max_read_len) + 2
        self.vl = self.lmax + 1
        self.per_ref = N_VECTORS * self.vl + N_SCALARS + N_HISTS * self.hl

    def shape(self):
        return (self.n_refs, self.per_ref)

    def scalar_offset(self, name):
        """index of a scalar counter inside one reference's row of the tensor"""
        return N_VECTORS * self.vl + SCALARS.index(name)

    def unpack(self, counts, ref, ref_len=None):
        """counts: int64 array [n_refs, per_ref] (numpy).  -> dict name -> vector (length ref_len) / int / histogram dict"""
        row = np.asarray(counts)[ref]
        L = self.lmax if ref_len is None else int(ref_len)
        out = {}
        for k, name in enumerate(VECTORS):
            out[name] = row[k * self.vl:k * self.vl + L].copy()
        base = N_VECTORS * self.vl
        for k, name in enumerate(SCALARS):
            out[name] = int(row[base + k])
        base += N_SCALARS
        for k, name in enumerate(HISTS):
            h = row[base + k * self.hl:base + (k + 1) * self.hl]
            out[name] = {int(i): int(h[i]) for i in np.nonzero(h)[0]}
        return out


def min_matches_table(min_aln_scores, max_t):
    """uint16 [n_refs, max_t+1]: for every alignment length T, the smallest `matches` with
    round(100*matches/float(T), 3) > min_aln_score -- the reference's own expression (CRISPResso2Align.pyx:433-434,
    CRISPRessoCORE.py:697), evaluated here so the device only compares integers."""
    tab = np.zeros((len(min_aln_scores), max_t + 1), dtype=np.uint16)
    for r, thr in enumerate(min_aln_scores):
        for T in range(1, max_t + 1):
            lo, hi = 0, T + 1                      # the score is monotone in matches
            while lo < hi:
                mid = (lo + hi) // 2
                if round(100 * mid / float(T), 3) > thr:
                    hi = mid
                else:
                    lo = mid + 1
            tab[r, T] = min(lo, 65535)
        tab[r, 0] = 65535
    return tab


def accumulate_device(ctx, layout, n_tasks, d_aln_read, d_aln_ref, aln_stride, d_records, d_counts, d_weights=None,
                      min_matches=None, flags=0, stream=None, d_hints=None):
    """Enqueue c2_count_vectors_kernel.  d_* are device addresses; min_matches is a host uint16 table or None.
    d_hints: the hint words (four per task) the align call of the same batch wrote (BatchAligner.align_device(..., d_hints=)): a task with a valid hint is counted from
    its hint alone (c2_count_hinted_kernel) -- same tensor, the rows of those tasks are not read back."""
    mm = None
    max_t = 0
    if min_matches is not None:
        mm = np.ascontiguousarray(min_matches, dtype=np.uint16)
        max_t = mm.shape[1] - 1
    rc = ctx.lib.c2_count_vectors_hinted_device(
        ctx.handle, ctypes.c_uint64(n_tasks), ctypes.c_void_p(d_aln_read), ctypes.c_void_p(d_aln_ref),
        ctypes.c_uint32(aln_stride), ctypes.c_void_p(d_records), ctypes.c_void_p(d_weights or 0), ctypes.c_void_p(d_hints or 0),
        mm.ctypes.data_as(ctypes.c_void_p) if mm is not None else None, int(max_t), int(flags), int(layout.hl),
        ctypes.c_void_p(d_counts), ctypes.c_void_p(stream or 0))
    ctx.check(rc, "c2_count_vectors_hinted_device")


def all_reduce_max(value, device):
    """max of an integer over all ranks (the ranks agree on the tensor layout with it); the value itself for one process"""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())
    return int(value)


def all_reduce(counts_tensor):
    """Sum the per-GPU count tensors over all ranks (RCCL all-reduce; a no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts_tensor, op=dist.ReduceOp.SUM)
    return counts_tensor


# ---- best-reference selection on the device (c2_select_best_device) ----
SELECT_STATS = ["N_COMPUTED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_ALN", "N_CACHED_NOTALN", "N_GLOBAL_SUBS", "N_SUBS_OUTSIDE_WINDOW",
                "N_MODS_IN_WINDOW", "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "n_bad_status", "a_bad_status"]
SELECT_DROP_AMBIGUOUS, SELECT_FIRST, SELECT_EXPAND = 0, 1, 2
SELECT_MAX_ALN_LEN = 7999


def min_mscore_table(min_aln_scores):
    """uint32 [n_refs]: the smallest integer k with k / 1000.0 > refs[name]['min_aln_score'].  The reference compares the
    Python float round(100*matches/float(len), 3) with min_aln_score (CRISPRessoCORE.py:697); that float is the double nearest
    to k/1000 for an integer k, so the comparison is one of integers once this threshold is known."""
    out = np.zeros(len(min_aln_scores), dtype=np.uint32)
    for r, thr in enumerate(min_aln_scores):
        lo, hi = 0, 100001                       # scores lie in [0, 100]
        while lo < hi:
            mid = (lo + hi) // 2
            if mid / 1000.0 > thr:
                hi = mid
            else:
                lo = mid + 1
        out[r] = lo
    return out


def select_mode(args):
    if getattr(args, 'assign_ambiguous_alignments_to_first_reference', False):
        return SELECT_FIRST
    if getattr(args, 'expand_ambiguous_alignments', False):
        return SELECT_EXPAND
    return SELECT_DROP_AMBIGUOUS


def select_best_device(ctx, n_reads, n_refs, d_records, min_mscore, mode, max_aln_len, d_records2=None, d_slot2=None,
                       d_raw_counts=None, d_counts=None, d_member=None, d_use2=None, d_flags=None, d_weights=None,
                       d_weights2=None, d_stats=None, stream=None):
    """Enqueue c2_select_best_kernel.  d_* are device addresses (ints) or None; min_mscore: host uint32 [n_refs]."""
    mm = np.ascontiguousarray(min_mscore, dtype=np.uint32)
    if mm.shape != (n_refs,):
        raise ValueError("one threshold per reference")
    P = lambda x: ctypes.c_void_p(x or 0)
    rc = ctx.lib.c2_select_best_device(
        ctx.handle, ctypes.c_uint64(n_reads), int(n_refs), P(d_records), P(d_records2), P(d_slot2),
        mm.ctypes.data_as(ctypes.c_void_p), P(d_raw_counts), P(d_counts), int(mode), int(max_aln_len),
        P(d_member), P(d_use2), P(d_flags), P(d_weights), P(d_weights2), P(d_stats), P(stream))
    ctx.check(rc, "c2_select_best_device")


def seed_tables(refs, ref_names, seed_count):
    """The seeds of c2_strand_plan_device: (blob uint8, off int32 [k, 2, S], len int32 [k, 2, S], n_seeds int32 [k], S) from
    refs[name]['fw_seeds'] / ['rc_seeds'] (the first min(seed_count, len) of each reference take part, CRISPRessoCORE.py:656-661)."""
    k = len(ref_names)
    ns = np.array([min(int(seed_count), len(refs[name]['fw_seeds'])) for name in ref_names], dtype=np.int32)
    S = max(1, int(ns.max()) if k else 1)
    off = np.zeros((k, 2, S), dtype=np.int32)
    ln = np.zeros((k, 2, S), dtype=np.int32)
    parts, pos = [], 0
    for r, name in enumerate(ref_names):
        for st, key in enumerate(('fw_seeds', 'rc_seeds')):
            for q in range(int(ns[r])):
                b = refs[name][key][q].encode()
                off[r, st, q], ln[r, st, q] = pos, len(b)
                parts.append(b)
                pos += len(b)
    blob = np.frombuffer(b"".join(parts), dtype=np.uint8).copy() if pos else np.zeros(0, dtype=np.uint8)
    return blob, off, ln, ns, S


def strand_plan_device(ctx, n_reads, d_reads, d_offsets, max_read_len, refs, ref_names, seed_count, seed_min, d_plan, stream=None):
    """c2_strand_plan_device: the seed test of get_new_variant_object for reads that are on the device -> d_plan uint8 [n, k]."""
    blob, off, ln, ns, S = seed_tables(refs, ref_names, seed_count)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else None
    ctx.check(ctx.lib.c2_strand_plan_device(ctx.handle, ctypes.c_uint64(int(n_reads)), ctypes.c_void_p(d_reads), ctypes.c_void_p(d_offsets),
                                            int(max_read_len), len(ref_names), int(S), P(ns), P(blob), int(blob.size), P(off), P(ln),
                                            int(seed_min), ctypes.c_void_p(d_plan), ctypes.c_void_p(stream)), "c2_strand_plan_device")
