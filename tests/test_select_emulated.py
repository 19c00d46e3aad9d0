"""The device-side strand / best-reference choice (c2_select_best_kernel) on the wave emulator against the reference's loop
restated in oracle/aggregate.select_best, and the integer form of the score (c2_mscore) against the reference's own float
expression round(100*matches/float(len), 3) (CRISPResso2Align.pyx:433-434)."""
import ctypes

import numpy as np
import pytest

import emu_driver as E
from crispresso2_amd import counts as C
from oracle import aggregate as AG


def test_mscore_is_the_rounded_score_times_1000():
    lib = E.lib()
    lib.emu_mscore.restype = ctypes.c_uint32
    # every (matches, len) pair of alignments up to 700 columns, then the lengths where exact ties occur (multiples of 64)
    # and a sample of long ones
    lens = list(range(1, 701)) + [64 * k for k in range(11, 125)] + [320 * k for k in range(1, 25)] + [1600, 3200, 4800, 6400, 7999, 7936]
    rng = np.random.default_rng(5)
    for T in lens:
        ms = range(T + 1) if T <= 700 else sorted(set(rng.integers(0, T + 1, 300).tolist() + [0, T, T // 2, T // 64, 3 * T // 64]))
        for m in ms:
            want = round(100 * m / float(T), 3)
            got = lib.emu_mscore(m, T)
            assert got / 1000.0 == want, (m, T, got, want)


def test_min_mscore_table_matches_the_float_comparison():
    for thr in (60.0, 0.0, 59.9995, 80, 99.9994, 100.0, 12.3456, -1.0):
        k = int(C.min_mscore_table([thr])[0])
        assert (k / 1000.0 > thr) and (k == 0 or not ((k - 1) / 1000.0 > thr))


def _run_kernel(rec1, rec2, slot2, thr, raw, cnt, mode):
    n, k = rec1.shape
    lib = E.lib()
    words = (k + 63) // 64                                    # per read: bit r % 64 of word r / 64 = reference r
    member = np.zeros((n, words), dtype=np.uint64); use2 = np.zeros((n, words), dtype=np.uint64); flags = np.zeros(n, dtype=np.uint8)
    w1 = np.full(n * k, 77, dtype=np.uint32)
    w2 = np.full(max(len(rec2), 1), 77, dtype=np.uint32)
    stats = np.zeros(len(C.SELECT_STATS), dtype=np.uint64)
    mm = C.min_mscore_table(thr)
    P = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    r1 = np.ascontiguousarray(rec1.reshape(-1))
    rc = lib.emu_select_best(ctypes.c_uint64(n), k, P(r1), P(rec2) if len(rec2) else None, P(slot2) if len(rec2) else None, P(mm),
                             P(raw), P(cnt), mode, P(member), P(use2), P(flags), P(w1), P(w2), P(stats))
    assert rc == 0
    return member, use2, flags, w1.reshape(n, k), w2, stats


@pytest.mark.parametrize("mode", [C.SELECT_DROP_AMBIGUOUS, C.SELECT_FIRST, C.SELECT_EXPAND])
@pytest.mark.parametrize("k", [1, 3, 7, 64, 70, 130])
def test_select_kernel_equals_the_reference_loop(mode, k):
    rng = np.random.default_rng(100 * k + mode)
    n = 700 if k < 64 else 300                                # three workgroups of 256, the last one ragged (k >= 64: masks of 1-3 words)
    rec1 = np.zeros((n, k), dtype=E.REC_DTYPE)
    T = rng.integers(200, 520, (n, k))
    # few distinct scores, so that ties between references and between strands are common
    m = (T * rng.choice([0.0, 0.3, 0.59, 0.6, 0.61, 0.9, 1.0], (n, k))).astype(np.int64)
    same = rng.random((n, k)) < 0.4
    T[:, 1:] = np.where(same[:, 1:], T[:, :1], T[:, 1:]); m[:, 1:] = np.where(same[:, 1:], m[:, :1], m[:, 1:])
    rec1["aln_len"], rec1["matches"] = T, m
    for f in ("insertion_n", "deletion_n", "substitution_n", "all_insertion_events", "all_deletion_bases", "all_substitutions"):
        rec1[f] = rng.integers(0, 9, (n, k))
    rec1["all_substitutions"] += rec1["substitution_n"]
    rec1["all_insertion_events"] += 9                         # keeps total_mods - mods_in_window from going negative too often; both signs occur
    rec1["irregular_ends"] = rng.integers(0, 2, (n, k))
    both = rng.random((n, k)) < 0.3
    bi, br = np.nonzero(both)
    rec2 = np.zeros(len(bi), dtype=E.REC_DTYPE)
    rec2["aln_len"] = T[bi, br]
    rec2["matches"] = np.clip(m[bi, br] + rng.integers(-20, 21, len(bi)) * (rng.random(len(bi)) < 0.6), 0, T[bi, br])
    for f in ("insertion_n", "deletion_n", "substitution_n", "all_insertion_events", "all_deletion_bases", "all_substitutions", "irregular_ends"):
        rec2[f] = rec1[f][bi, br] + 1
    slot2 = np.full((n, k), -1, dtype=np.int32)
    slot2[bi, br] = np.arange(len(bi))
    thr = ([60.0, 59.0, 0.0, 61.5, 60.0, 99.0, 60.0] * 19)[:k]
    raw = rng.integers(1, 50, n).astype(np.uint32)
    cnt = (raw + rng.integers(0, 5, n)).astype(np.uint32)
    member, use2, flags, w1, w2, stats = _run_kernel(rec1, rec2, slot2, thr, raw, cnt, mode)

    want_stats = dict.fromkeys(C.SELECT_STATS, 0)
    sc = lambda r: round(100 * int(r["matches"]) / float(int(r["aln_len"])), 3)
    for i in range(n):
        s_fw = [sc(rec1[i, r]) for r in range(k)]
        s_rc = [sc(rec2[slot2[i, r]]) if slot2[i, r] >= 0 else None for r in range(k)]
        best, use_rc, aligned, counted, ambiguous = AG.select_best(s_fw, s_rc, thr, assign_first=mode == C.SELECT_FIRST, expand=mode == C.SELECT_EXPAND)
        assert sum(int(x) << (64 * q) for q, x in enumerate(member[i])) == sum(1 << r for r in best), i
        assert sum(int(x) << (64 * q) for q, x in enumerate(use2[i])) == sum(1 << r for r in range(k) if use_rc[r]), i
        assert int(flags[i]) == (1 if aligned else 0) | (2 if ambiguous else 0), i
        for r in range(k):
            assert int(w1[i, r]) == (int(cnt[i]) if (r in counted and not use_rc[r]) else 0), (i, r)
            if slot2[i, r] >= 0:
                assert int(w2[slot2[i, r]]) == (int(cnt[i]) if (r in counted and use_rc[r]) else 0), (i, r)
        c = int(raw[i])
        if aligned:
            last = best[-1]
            p = rec2[slot2[i, last]] if use_rc[last] else rec1[i, last]
            sub_all, sub_win = int(p["all_substitutions"]), int(p["substitution_n"])
            total = int(p["all_insertion_events"]) + int(p["all_deletion_bases"]) + sub_all
            in_win = sub_win + int(p["deletion_n"]) + int(p["insertion_n"])
            want_stats["N_COMPUTED_ALN"] += 1; want_stats["N_CACHED_ALN"] += c - 1
            want_stats["N_GLOBAL_SUBS"] += sub_all * c; want_stats["N_SUBS_OUTSIDE_WINDOW"] += (sub_all - sub_win) * c
            want_stats["N_MODS_IN_WINDOW"] += in_win * c; want_stats["N_MODS_OUTSIDE_WINDOW"] += (total - in_win) * c
            want_stats["N_READS_IRREGULAR_ENDS"] += int(p["irregular_ends"]) * c
        else:
            want_stats["N_COMPUTED_NOTALN"] += 1; want_stats["N_CACHED_NOTALN"] += c - 1
    got = dict(zip(C.SELECT_STATS, stats.view(np.int64).tolist()))
    assert got == want_stats


def test_select_kernel_reports_records_with_a_status():
    rec1 = np.zeros((5, 2), dtype=E.REC_DTYPE)
    rec1["aln_len"], rec1["matches"] = 100, 90
    rec1["status"][3, 1] = 16
    *_, stats = _run_kernel(rec1, np.zeros(0, dtype=E.REC_DTYPE), np.zeros((5, 2), dtype=np.int32), [60.0, 60.0], None, None, 0)
    assert int(stats[C.SELECT_STATS.index("n_bad_status")]) == 1 and int(stats[C.SELECT_STATS.index("a_bad_status")]) == 16


def _seed_case(seed_count, seed_len, seed_min):
    """reads of mixed strands / lengths (shorter than a seed too), references with seed lists as the reference's setup builds them"""
    from crispresso2_amd import refs as RF, synth
    from types import SimpleNamespace
    amp, _g, inc = synth.amplicon_setup(250)
    amp2 = synth.make_variant(amp, "pe")[:-31]
    args = SimpleNamespace(aln_seed_count=seed_count, aln_seed_len=seed_len, aln_seed_min=seed_min)
    made = max(seed_count, 2)                                        # (the seeds a reference has; args.aln_seed_count says how many take part)
    refs = {"A": RF.make_ref("A", amp, [125], inc, min_aln_score=60, aln_seed_count=made, aln_seed_len=seed_len),
            "B": RF.make_ref("B", amp2, [110], [109, 110], min_aln_score=60, aln_seed_count=made, aln_seed_len=seed_len)}
    rng = np.random.default_rng(seed_count * 100 + seed_len)
    reads = []
    for r in synth.make_reads(250, 150):
        s = r.tobytes().decode()
        kind = int(rng.integers(0, 6))
        if kind == 0:
            s = RF.reverse_complement(s)
        elif kind == 1:
            s = s[:int(rng.integers(1, 40))]                        # shorter than most seeds' positions, some shorter than a seed
        elif kind == 2:
            s = "".join(rng.choice(list("ACGT"), int(rng.integers(5, 260))))
        elif kind == 3:
            s = s[:100] + RF.reverse_complement(s[100:])             # seeds of both strands
        reads.append(s)
    return args, refs, ["A", "B"], reads


@pytest.mark.parametrize("bytewise", [False, True])
@pytest.mark.parametrize("seed_count,seed_len,seed_min", [(5, 10, 2), (1, 10, 0), (9, 7, 4), (3, 40, 1), (0, 10, 0), (4, 32, 1), (4, 5, 0)])
def test_strand_plan_kernel_equals_the_host_seed_test(seed_count, seed_len, seed_min, bytewise, monkeypatch):
    """c2_strand_plan_kernel (one wavefront per read, all references) on the emulator = the host's threaded c2_strand_plan = the
    statements of get_new_variant_object (CRISPRessoCORE.py:656-687) in variants._strand_plan."""
    # (bytewise: the kernel's byte-by-byte path, which seeds of more than 32 bytes take anyway; else the LDS seed table with four-byte
    #  window compares -- seed lengths 5, 7, 10, 32 are 2, 2, 3, 8 dwords with and without a partial last one)
    if bytewise:
        monkeypatch.setenv("C2_EMU_STRAND_BYTEWISE", "1")
    from crispresso2_amd import _native, counts as C, variants
    args, refs, names, reads = _seed_case(seed_count, seed_len, seed_min)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy()
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    blob, soff, slen, ns, S = C.seed_tables(refs, names, args.aln_seed_count)
    plan = np.full((len(reads), len(names)), 9, dtype=np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else None
    rc = E.lib().emu_strand_plan(ctypes.c_uint64(len(reads)), P(arena), P(off), int(max(len(r) for r in reads)), len(names), int(S), P(ns), P(blob),
                                 int(blob.size), P(soff), P(slen), int(seed_min), P(plan))
    assert rc == 0
    for r, name in enumerate(names):
        m = min(args.aln_seed_count, len(refs[name]["fw_seeds"]))
        host = _native.strand_plan(arena, off, refs[name]["fw_seeds"][:m], refs[name]["rc_seeds"][:m], args.aln_seed_min)
        assert np.array_equal(plan[:, r], host), name
        exp = np.array([variants._strand_plan(args, s, refs[name]) for s in reads], dtype=np.uint8)
        assert np.array_equal(plan[:, r], exp), name
    if seed_count == 5:
        assert set(np.unique(plan)) == {0, 1, 2}
