"""Parity tests proper: the HIP path, called through the C ABI, against the golden vectors and the
CPU oracle on the same seeded inputs.  Integer / byte work: the bar is bit-exact.  Run on an MI355X
with `pytest -m gpu`."""
import ctypes
import os

import numpy as np
import pytest

from helpers import PAYLOAD_FIELDS, load_golden, matrices, payload_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mats():
    return matrices()


@pytest.fixture(scope="module")
def ctx():
    from crispresso2_amd import _native
    return _native.default_context()       # raises loudly if the HIP extension or the GPU is missing


def check_record(rec, payload, s1, s2):
    assert rec["insertion_n"] == payload["insertion_n"]
    assert rec["deletion_n"] == payload["deletion_n"]
    assert rec["substitution_n"] == payload["substitution_n"]
    assert rec["all_insertion_events"] == len(payload["all_insertion_left_positions"])
    assert rec["win_insertion_events"] == len(payload["insertion_sizes"])
    assert rec["all_deletion_events"] == len(payload["all_deletion_coordinates"])
    assert rec["win_deletion_events"] == len(payload["deletion_coordinates"])
    assert rec["all_deletion_bases"] == len(payload["all_deletion_positions"])
    assert rec["all_substitutions"] == len(payload["all_substitution_positions"])
    irregular = (s1[0] == "-" or s2[0] == "-" or s1[0] != s2[0]) or (s1[-1] == "-" or s2[-1] == "-" or s1[-1] != s2[-1])
    assert bool(rec["irregular_ends"]) == irregular


def check_selftest(out):
    assert out[0] == -7
    assert (out[1:64] == 3 * np.arange(0, 63) + 1).all()
    assert (out[64:128] == 85).all()
    assert out[128] == 22 + 1 and (out[129:192] == 22).all()
    # lane 31 switched off in EXEC: it keeps its value, and the lanes that would read it keep `old`
    shr = 3 * np.arange(-1, 63) + 1
    shr[0] = -7; shr[31] = -99; shr[32] = -7
    assert (out[192:256] == shr).all(), out[192:256]
    shl = 3 * np.arange(1, 65) + 1
    shl[63] = -7; shl[31] = -99; shl[30] = -7
    assert (out[256:320] == shl).all(), out[256:320]
    # bound_ctrl set + folded into an add: a lane without a source reads 0
    shrz = 3 * np.arange(-1, 63) + 1 + 1000
    shrz[0] = 1000; shrz[31] = -99; shrz[32] = 1000
    assert (out[320:384] == shrz).all(), out[320:384]
    shlz = 3 * np.arange(1, 65) + 1 + 1000
    shlz[63] = 1000; shlz[31] = -99; shlz[30] = 1000
    assert (out[384:448] == shlz).all(), out[384:448]


def test_cross_lane_primitives(ctx):
    """DPP wave_shr:1 keeps `old` in lane 0 and shifts lane n-1 -> n; readlane; 64-bit ballot."""
    out = np.zeros(448, dtype=np.int32)
    ctx.check(ctx.lib.c2_selftest(ctx.handle, out.ctypes.data_as(ctypes.c_void_p)), "c2_selftest")
    check_selftest(out)
    # row forms (hand-off inside the 16-lane groups of c2_align_diagp_kernel<8>): a row's first / last lane reads 0
    rows = np.zeros(128, dtype=np.int32)
    ctx.check(ctx.lib.c2_selftest_rows(ctx.handle, rows.ctypes.data_as(ctypes.c_void_p)), "c2_selftest_rows")
    lane = np.arange(64)
    assert (rows[:64] == np.where(lane % 16 == 0, 1000, 3 * (lane - 1) + 1 + 1000)).all(), rows[:64]
    assert (rows[64:] == np.where(lane % 16 == 15, 1000, 3 * (lane + 1) + 1 + 1000)).all(), rows[64:]


def run_batch_vectors(vecs, mats, ctx):
    from crispresso2_amd.batch import BatchAligner
    import oracle
    groups = {}
    for v in vecs:
        key = (v["seqi"], tuple(v["gap_incentive"]), v["matrix"], v["gap_open"], v["gap_extend"], tuple(v.get("include", [])))
        groups.setdefault(key, []).append(v)
    n = 0
    for (seqi, g, mat, go, ge, inc), vs in groups.items():
        al = BatchAligner([seqi], [np.array(g, dtype=np.int64)], [list(inc)], mats[mat], go, ge, ctx=ctx)
        res = al.align([v["seqj"] for v in vs])
        sc = res.scores
        for k, v in enumerate(vs):
            r = res.records[k]
            assert r["status"] == 0, (v, r)
            s1, s2 = res.strings(k)
            assert [s1, s2] == v["out"][:2], v
            assert sc[k] == v["out"][2]
            payload = v.get("payload") or oracle.find_indels_substitutions(s1, s2, list(inc))
            check_record(r, payload, s1, s2)
            n += 1
    return n


def test_batch_reference_unit_test_answers(mats, ctx):
    kats = [k for k in load_golden("ref_unit_kats.json") if k["fn"] == "global_align"]
    assert run_batch_vectors(kats, mats, ctx) == len(kats)


def test_batch_fuzz_vectors(mats, ctx):
    vecs = load_golden("fuzz_align.json")
    assert run_batch_vectors(vecs, mats, ctx) == len(vecs)


def test_batch_realistic_vectors(mats, ctx):
    vecs = load_golden("realistic.json")
    assert run_batch_vectors(vecs, mats, ctx) == len(vecs)


def test_per_call_api_matches_reference_signature_and_answers(mats, ctx):
    """crispresso2_amd.CRISPResso2Align / CRISPRessoCOREResources called the way CRISPRessoCORE.py:667-724 calls them."""
    from crispresso2_amd import CRISPResso2Align, CRISPRessoCOREResources
    for k in load_golden("ref_unit_kats.json"):
        if k["fn"] == "global_align":
            out = CRISPResso2Align.global_align(k["seqj"], k["seqi"], matrix=mats[k["matrix"]],
                                                gap_incentive=np.array(k["gap_incentive"], dtype=int),
                                                gap_open=k["gap_open"], gap_extend=k["gap_extend"])
            assert list(out) == k["out"], k["ref_test"]
            assert isinstance(out[0], str) and isinstance(out[2], float)
        elif k["fn"] == "find_indels_substitutions":
            p = CRISPRessoCOREResources.find_indels_substitutions(k["read_al"], k["ref_al"], k["include"])
            assert isinstance(p, CRISPRessoCOREResources.ResultsSlotsDict)
            assert payload_diff(p, k["out"]) == [], k["ref_test"]
        else:
            p = CRISPRessoCOREResources.find_indels_substitutions_legacy(k["read_al"], k["ref_al"], k["include"])
            assert isinstance(p, dict)
            assert payload_diff(p, k["out"]) == [], k["ref_test"]
    for v in load_golden("realistic.json")[::9]:
        s1, s2, sc = CRISPResso2Align.global_align(v["seqj"], v["seqi"], matrix=mats["EDNAFULL"],
                                                   gap_incentive=np.array(v["gap_incentive"], dtype=int),
                                                   gap_open=-20, gap_extend=-2)
        assert [s1, s2, sc] == v["out"]
        p = CRISPRessoCOREResources.find_indels_substitutions(s1, s2, np.array(v["include"]))
        assert payload_diff(p, v["payload"]) == []
        assert isinstance(p["all_substitution_values"], np.ndarray)


def test_per_call_error_behaviour(mats, ctx, capsys):
    from crispresso2_amd import CRISPResso2Align
    m = mats["EDNAFULL"]
    # pyx:124-126: prints an error and returns the int 0
    assert CRISPResso2Align.global_align("ACGT", "ACGT", matrix=m, gap_incentive=np.zeros(3, dtype=int)) == 0
    assert "Mismatch in gap_incentive length" in capsys.readouterr().out
    with pytest.raises(TypeError):
        CRISPResso2Align.global_align(b"ACGT", "ACGT", matrix=m, gap_incentive=np.zeros(5, dtype=int))
    with pytest.raises(ValueError):
        CRISPResso2Align.global_align("ACGT", "ACGT", matrix=m, gap_incentive=np.zeros(5, dtype=np.int32))
    with pytest.raises(Exception):       # a reference character outside the matrix: out of bounds in the reference, refused here
        CRISPResso2Align.global_align("ACGT", "ACGa", matrix=m, gap_incentive=np.zeros(5, dtype=int))
    with pytest.raises(Exception):       # 'a' against 'Y' (the matrix's last row): past the end of the flat buffer
        CRISPResso2Align.global_align("ACGa", "ACGY", matrix=m, gap_incentive=np.zeros(5, dtype=int))
    # 'a' against A/C/G/T is an element of the buffer (pyx:212, bounds checking off): the reference's answer
    import oracle
    assert CRISPResso2Align.global_align("ACGa", "ACGT", matrix=m, gap_incentive=np.zeros(5, dtype=int)) == \
        tuple(oracle.global_align("ACGa", "ACGT", m, np.zeros(5, dtype=np.int64)))


def test_classify_lists_fuzz_vectors(ctx):
    from crispresso2_amd import CRISPRessoCOREResources as R
    n_legacy = 0
    for v in load_golden("fuzz_classify.json"):
        legacy = v["fn"].endswith("legacy")
        p = (R.find_indels_substitutions_legacy if legacy else R.find_indels_substitutions)(v["read_al"], v["ref_al"], v["include"])
        assert payload_diff(p, v["out"]) == [], v
        n_legacy += legacy
    assert n_legacy >= 100


def test_classify_lists_batch_equals_per_call_answers(ctx):
    """c2_classify_lists_batch (one lane per alignment, count pass + write pass) on all the reference-generated classifier
    vectors at once -- every vector with its own include set, negative-coordinate quirk cases included."""
    from crispresso2_amd import CRISPRessoCOREResources as R
    vecs = load_golden("fuzz_classify.json")
    for legacy in (False, True):
        vs = [v for v in vecs if v["fn"].endswith("legacy") == legacy]
        assert len(vs) >= 100
        out = R.find_indels_substitutions_batch([(v["read_al"], v["ref_al"]) for v in vs], [v["include"] for v in vs],
                                                set_ids=np.arange(len(vs), dtype=np.uint16), legacy=legacy)
        assert len(out) == len(vs)
        for p, v in zip(out, vs):
            assert payload_diff(p, v["out"]) == [], v
    assert R.find_indels_substitutions_batch([], [[1]]) == []


def test_calculate_homology(ctx):
    from crispresso2_amd import CRISPRessoCOREResources as R
    import oracle
    for a, b in [(b"ACGTACGT", b"ACGTTCGT"), (b"AAAA", b"TTTT"), (b"ACG", b"ACG"), (b"ACGTACGTAC" * 7, b"ACGAACGTAC" * 7)]:
        assert R.calculate_homology(a, b) == oracle.calculate_homology(a, b)
    # 200 random byte-string pairs (1 .. 700 bytes, related and unrelated) against the reference's own compiled calculate_homology
    # (oracle/_ref; the C restatement where that is not built) -- the float32 accumulation included
    compiled = oracle.ref()
    want = compiled[1].calculate_homology if compiled is not None else oracle.calculate_homology
    rng = np.random.default_rng(808)
    for _ in range(200):
        n = int(rng.integers(1, 700))
        a = rng.choice(np.frombuffer(b"ACGTN-", dtype=np.uint8), n)
        b = a.copy()
        flip = rng.random(n) < rng.choice([0.0, 0.01, 0.3, 1.0])
        b[flip] = rng.choice(np.frombuffer(b"ACGTN-", dtype=np.uint8), int(flip.sum()))
        assert R.calculate_homology(a.tobytes(), b.tobytes()) == want(a.tobytes(), b.tobytes())


def test_long_references_multipass_vs_oracle(mats, ctx):
    """References longer than 256 rows take several systolic passes through the LDS boundary row."""
    from crispresso2_amd.batch import BatchAligner
    import oracle
    rng = np.random.default_rng(2718)
    m = mats["EDNAFULL"]
    for L in (257, 300, 400, 513):
        ref = "".join(rng.choice(list("ACGT"), L))
        g = np.zeros(L + 1, dtype=np.int64)
        g[L // 2 + 1] = 1
        inc = list(range(L // 2 - 10, L // 2 + 10))
        reads = []
        for _ in range(12):
            s = list(ref)
            p = int(rng.integers(5, L - 70))
            k = int(rng.integers(0, 3))
            if k == 0:
                del s[p:p + int(rng.integers(1, 60))]
            elif k == 1:
                s[p:p] = list(rng.choice(list("ACGT"), int(rng.integers(1, 20))))
            s[int(rng.integers(0, len(s)))] = "N"
            reads.append("".join(s)[: min(len(s), 330)])
        reads.append("".join(rng.choice(list("ACGT"), 250)))
        al = BatchAligner([ref], [g], [inc], m, -20, -2, ctx=ctx)
        res = al.align(reads)
        for k, rd in enumerate(reads):
            st, s1, s2, mt, ln = oracle.global_align_raw(rd, ref, m, g, -20, -2)
            assert st == 0 and res.records["status"][k] == 0
            assert res.strings(k) == (s1, s2)
            assert (int(res.records["matches"][k]), int(res.records["aln_len"][k])) == (mt, ln)
            check_record(res.records[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)


@pytest.mark.parametrize("li,lj", [(600, 600), (1000, 300), (300, 1000), (40, 2500), (2000, 2000)])
def test_alignments_larger_than_the_lds_pointer_plane_vs_oracle(mats, ctx, li, lj):
    """600 x 600 / 1000 x 300 and beyond: the full pointer plane exceeds the 160 KB of LDS, so the chain's last launch keeps it in
    per-workgroup HBM scratch (r01 answered C2_E_TOO_LARGE here).  Bit-exact against the oracle, whole chain and full-plane only."""
    from crispresso2_amd.batch import BatchAligner
    import oracle
    rng = np.random.default_rng(li * 7 + lj)
    m = mats["EDNAFULL"]
    ref = "".join(rng.choice(list("ACGT"), li))
    g = np.zeros(li + 1, dtype=np.int64)
    g[li // 2] = 1
    inc = list(range(li // 2 - 10, li // 2 + 10))
    reads = []
    for _ in range(40):
        base = (ref * (lj // li + 1))[:lj] if lj >= li else ref[int(rng.integers(0, li - lj)):][:lj]
        s = list(base)
        for _e in range(int(rng.integers(0, 5))):
            p = int(rng.integers(5, len(s) - 5))
            k = int(rng.integers(0, 3))
            if k == 0:
                s[p] = "ACGT"[int(rng.integers(0, 4))]
            elif k == 1:
                del s[p:p + int(rng.integers(1, 150))]
            else:
                s[p:p] = list(rng.choice(list("ACGT"), int(rng.integers(1, 30))))
        reads.append("".join(s)[:max(lj, 20)])
    reads.append("".join(rng.choice(list("ACGT"), lj)))
    al = BatchAligner([ref], [g], [inc], m, -20, -2, ctx=ctx)
    for mode in ("auto", "full"):
        ctx.set_kernel_mode(mode)
        try:
            res = al.align(reads)
        finally:
            ctx.set_kernel_mode("auto")
        for k, rd in enumerate(reads):
            st, s1, s2, mt, ln = oracle.global_align_raw(rd, ref, m, g, -20, -2)
            assert st == 0 and res.records["status"][k] == 0, (mode, k)
            assert res.strings(k) == (s1, s2), (mode, k)
            assert (int(res.records["matches"][k]), int(res.records["aln_len"][k])) == (mt, ln)
            check_record(res.records[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)


def test_hbm_pointer_plane_forced_on_amplicon_sized_batches(mats, ctx):
    """C2_FORCE_HBM_PLANE routes the full-plane launch of ordinary batches through the HBM-plane instance: same bytes out."""
    os.environ["C2_FORCE_HBM_PLANE"] = "1"
    try:
        ctx.set_kernel_mode("full")
        try:
            for name in ("realistic.json", "fuzz_align.json"):
                vecs = load_golden(name)
                assert run_batch_vectors(vecs, mats, ctx) == len(vecs)
        finally:
            ctx.set_kernel_mode("auto")
        vecs = load_golden("realistic.json")
        assert run_batch_vectors(vecs, mats, ctx) == len(vecs)
    finally:
        del os.environ["C2_FORCE_HBM_PLANE"]


def test_multi_reference_strands_and_pooled_ids_vs_oracle(mats, ctx):
    """all_refs (every read x every reference, CRISPRessoCORE.py:653), per-read amplicon ids (Pooled), reverse complement."""
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    amp, g, inc = synth.amplicon_setup(250)
    hdr, pe = synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")
    refs = [amp, hdr, pe]
    gis = [np.zeros(len(r) + 1, dtype=np.int64) for r in refs]
    for x in gis:
        x[126] = 1
    incs = [[125, 126]] * 3
    reads_u8 = synth.make_reads(250, 60)
    reads = [r.tobytes().decode() for r in reads_u8]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    al = BatchAligner(refs, gis, incs, m, -20, -2, ctx=ctx)
    res = al.align(reads, all_refs=True)
    assert len(res) == 180
    for t in range(0, 180, 7):
        rd, rf = reads[t // 3], t % 3
        st, s1, s2, mt, ln = oracle.global_align_raw(rd, refs[rf], m, gis[rf], -20, -2)
        assert res.strings(t) == (s1, s2) and res.records["ref_id"][t] == rf
        check_record(res.records[t], oracle.find_indels_substitutions(s1, s2, incs[rf]), s1, s2)
    rids = np.arange(60) % 3
    strands = (np.arange(60) // 3) % 2
    rc_reads = ["".join(comp[c] for c in reversed(r)) if s else r for r, s in zip(reads, strands)]
    res = al.align(rc_reads, ref_ids=rids, strands=strands)
    for k in range(60):
        st, s1, s2, mt, ln = oracle.global_align_raw(reads[k], refs[rids[k]], m, gis[rids[k]], -20, -2)
        assert res.strings(k) == (s1, s2)
        assert res.records["strand"][k] == strands[k] and res.records["ref_id"][k] == rids[k]


def test_host_batch_pipelined_through_pinned_staging_equals_the_one_shot_path(mats, ctx):
    """c2_align_classify_batch_host cuts large batches into chunks (pinned staging both ways, copies overlapped with the launch
    chains).  Forced here on small ragged batches with odd chunk sizes -- single reference, per-read amplicon ids + strands, and
    an all-references batch: the same bytes as the one-shot path, and a 300 k-read batch at the default sizes against it."""
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    m = mats["EDNAFULL"]
    amp, g, inc = synth.amplicon_setup(250)
    hdr, pe = synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")
    refs = [amp, hdr, pe[:-17]]
    gis = [np.zeros(len(r) + 1, dtype=np.int64) for r in refs]
    for x in gis:
        x[126] = 1
    incs = [[125, 126]] * 3
    rng = np.random.default_rng(99)
    reads_u8 = synth.make_reads(250, 700)
    reads = [r.tobytes().decode()[:int(rng.integers(180, 251))] for r in reads_u8]          # ragged
    rids = rng.integers(0, 3, len(reads))
    strands = rng.integers(0, 2, len(reads))
    al = BatchAligner(refs, gis, incs, m, -20, -2, ctx=ctx)

    def run(**kw):
        r = al.align(reads, **kw)
        return r.aln_read.copy(), r.aln_ref.copy(), r.records.copy()

    def same(a, b):
        T = a[2]["aln_len"].astype(np.int64)
        cols = np.arange(a[0].shape[1])[None, :] < T[:, None]
        return ((a[0] == b[0]) | ~cols).all() and ((a[1] == b[1]) | ~cols).all() and (a[2].tobytes() == b[2].tobytes())

    os.environ["C2_HOST_PIPE_MIN_TASKS"] = str(1 << 40)
    try:
        base = [run(), run(ref_ids=rids, strands=strands), run(all_refs=True)]
    finally:
        del os.environ["C2_HOST_PIPE_MIN_TASKS"]
    for chunk in (37, 256, 699, 5000):
        os.environ["C2_HOST_PIPE_MIN_TASKS"] = "1"
        os.environ["C2_HOST_PIPE_CHUNK_TASKS"] = str(chunk)
        try:
            got = [run(), run(ref_ids=rids, strands=strands), run(all_refs=True)]
        finally:
            del os.environ["C2_HOST_PIPE_MIN_TASKS"], os.environ["C2_HOST_PIPE_CHUNK_TASKS"]
        for a, b_ in zip(got, base):
            assert (a[2]["status"] == 0).all() and same(a, b_), chunk
    # default sizes
    n = 300_000
    big = synth.make_reads(250, n)
    offsets = np.arange(n + 1, dtype=np.uint64) * 250
    al1 = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    r1 = al1.align((big.reshape(-1), offsets))
    os.environ["C2_HOST_PIPE_MIN_TASKS"] = str(1 << 40)
    try:
        r0 = al1.align((big.reshape(-1), offsets))
    finally:
        del os.environ["C2_HOST_PIPE_MIN_TASKS"]
    assert same((r1.aln_read, r1.aln_ref, r1.records), (r0.aln_read, r0.aln_ref, r0.records))


def test_banded_pointer_plane_equals_full_plane(mats, ctx):
    """The banded first launch + full-plane fallback must give byte-identical outputs to the full-plane kernel alone,
    for a narrow band (many fallbacks), the automatic band, and with the band off."""
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    m = mats["EDNAFULL"]
    L, n = 250, 30_000
    amp, g, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, n)
    rng = np.random.default_rng(7)
    reads[::97] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, reads[::97].shape)]   # unrelated reads wander off-diagonal
    offsets = np.arange(n + 1, dtype=np.uint64) * L
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    outs = {}
    try:
        ctx.set_kernel_mode("band")
        for band in (0, 2, 6, -1):
            ctx.set_band(band)
            res = al.align((reads.reshape(-1), offsets))
            info = ctx.band_info(L)
            outs[band] = (res, info)
            assert (res.records["status"] == 0).all()
        ctx.set_band(-1)
        # diagonal-band kernels + certificate: 1 alignment per wavefront; tiers 2 -> 1; tiers 4 -> 2 -> 1 (the default)
        tiers = {}
        for mode in ("diag1", "diag2", "auto"):
            ctx.set_kernel_mode(mode)
            res = al.align((reads.reshape(-1), offsets))
            outs[mode] = (res, ctx.band_info(L))
            tiers[mode] = ctx.tier_info()
            assert (res.records["status"] == 0).all()
    finally:
        ctx.set_band(-1)
        ctx.set_kernel_mode("auto")
    base = outs[0][0]
    assert outs[0][1]["band_lanes"] == 0
    for band in (2, 6, -1, "diag1", "diag2", "auto"):
        res, info = outs[band]
        assert info["band_lanes"] == -1 if isinstance(band, str) else info["band_lanes"] > 0
        assert np.array_equal(res.records, base.records)
        assert np.array_equal(res.aln_read, base.aln_read) and np.array_equal(res.aln_ref, base.aln_ref)
    assert outs[2][1]["fallback_tasks_last_launch"] > outs[6][1]["fallback_tasks_last_launch"] > 0
    assert 0 < outs["diag1"][1]["fallback_tasks_last_launch"] < n // 20
    # every tier certifies most of what it gets and hands the rest down; the last banded tier is the same in all chains
    assert len(tiers["diag1"]) == 1 and len(tiers["diag2"]) == 2 and len(tiers["auto"]) == 4      # (32 / 40 / 62 / 128 diagonals)
    assert n // 2 > tiers["auto"][0] > tiers["auto"][1] >= tiers["auto"][2] >= tiers["auto"][3] > 0
    # (round 5: in the default chain the partition sends the reads that match the reference nowhere straight to the LAST list, so the list behind
    #  its second tier is shorter than the 32-bit chain's; what reaches the full-matrix launch is the same)
    assert tiers["auto"][3:] == tiers["diag2"][1:] and tiers["auto"][2] <= tiers["diag2"][0] and tiers["diag2"][1:] == tiers["diag1"]


def test_count_vectors_device_vs_reference_aggregation(mats, ctx):
    """c2_count_vectors_kernel on the GPU (fed by the align kernel's outputs in HBM) against oracle/aggregate.py, the CPU
    restatement of CRISPRessoCORE.py:3964-4115; weights, the min_aln_score gate and the ignore_* / discard flags."""
    import torch
    from crispresso2_amd import synth, counts as C
    from crispresso2_amd.batch import BatchAligner
    import oracle
    from oracle import aggregate
    m = mats["EDNAFULL"]
    L, n = 250, 3000
    amp, g, _ = synth.amplicon_setup(L)
    inc = list(range(L // 2 - 10, L // 2 + 10))
    reads = synth.make_reads(L, n)
    rng = np.random.default_rng(3)
    reads[::50] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, reads[::50].shape)]     # unrelated: fail the score gate
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L, stream=s)
    torch.cuda.synchronize()
    from crispresso2_amd import _native
    records = rec.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    a1, a2 = o1.cpu().numpy(), o2.cpu().numpy()
    w = rng.integers(0, 40, n).astype(np.uint32)
    d_w = torch.from_numpy(w.astype(np.int32)).to(dev)
    mm = C.min_matches_table([60.0], 2 * L)
    scores = [round(100 * int(r["matches"]) / float(int(r["aln_len"])), 3) for r in records]
    assert min(scores) <= 60 < max(scores)
    payloads = []
    for k in range(n):
        T = int(records["aln_len"][k])
        s1, s2 = a1[k, :T].tobytes().decode(), a2[k, :T].tobytes().decode()
        p = oracle.find_indels_substitutions(s1, s2, inc)
        p["aln_seq"], p["aln_ref"] = s1, s2
        payloads.append(p)
    lay = C.CountLayout(1, L, L)
    for flags in (0, C.FLAG_IGNORE_SUBSTITUTIONS, C.FLAG_DISCARD_INDEL_READS):
        d_counts = torch.zeros(lay.shape(), dtype=torch.int64, device=dev)
        C.accumulate_device(ctx, lay, n, o1.data_ptr(), o2.data_ptr(), stride, rec.data_ptr(), d_counts.data_ptr(),
                            d_weights=d_w.data_ptr(), min_matches=mm, flags=flags, stream=s)
        torch.cuda.synchronize()
        got = lay.unpack(d_counts.cpu().numpy(), 0, L)
        items = [(p, int(c)) for p, c, sc in zip(payloads, w, scores) if c > 0 and sc > 60.0]
        exp = aggregate.aggregate(items, L, ignore_substitutions=bool(flags & 1), discard_indel_reads=bool(flags & 8))
        for k, v in exp.items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(got[k], v), k
            else:
                assert got[k] == v, (k, got[k], v)


def test_status_bits(mats, ctx):
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    al = BatchAligner(["ACGT"], [np.zeros(5, dtype=np.int64)], [[1, 2]], m, -20, -2, ctx=ctx)
    # reads b"ACG\x85", "ACGT", "" and "ACRT" (the last one reverse-complemented) as a packed arena
    res = al.align((np.frombuffer(b"ACG\x85ACGTACRT", dtype=np.uint8).copy(), np.array([0, 4, 8, 8, 12], dtype=np.uint64)), strands=[0, 0, 0, 1])
    assert res.records["status"][0] & 2          # a byte >= 128 indexes the matrix backwards in the reference (signed char): refused
    assert res.records["status"][1] == 0
    assert res.records["status"][2] & 1
    assert res.records["status"][3] & 16
    # a read character beyond the 90 x 90 matrix reads the flat element ci * 90 + cj in the reference (pyx:212, bounds checking off):
    # defined while the largest reference character keeps it inside the buffer, refused otherwise; a reference character beyond
    # the matrix is always out of bounds
    g = np.zeros(5, dtype=np.int64)
    al = BatchAligner(["ACGT", "ACGY", "ACGa"], [g, g, g], [[1, 2]] * 3, m, -20, -2, ctx=ctx)
    res = al.align(["ACGa", "ACGa", "ACGT", "None"], ref_ids=[0, 1, 2, 0])
    assert res.records["status"][0] == 0 and res.strings(0) == tuple(oracle.global_align("ACGa", "ACGT", m, g, -20, -2)[:2])
    assert res.records["status"][1] & 2 and res.records["status"][2] & 2
    assert res.records["status"][3] == 0 and res.strings(3) == tuple(oracle.global_align("None", "ACGT", m, g, -20, -2)[:2])
    al = BatchAligner(["G"], [np.array([1, 0], dtype=np.int64)], [[0]], m, -1, -1, ctx=ctx)
    res = al.align(["TT"])
    assert oracle.global_align_raw("TT", "G", m, np.array([1, 0], dtype=np.int64), -1, -1)[0] != 0
    assert res.records["status"][0] & (4 | 8)


@pytest.mark.parametrize("L,n", [(150, 200_000), (250, 200_000)])
def test_full_length_batches_properties_and_sampled_oracle(mats, ctx, L, n):
    """BASELINE configs 2/3 shape at a size the GPU does in well under a second: size-independent properties on
    every alignment (ungapped strings reproduce read and reference; counts are consistent; two launches are
    identical), plus a seeded sample checked in full against the oracle."""
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    amp, g, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, n)
    offsets = np.arange(n + 1, dtype=np.uint64) * L
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    res = al.align((reads.reshape(-1), offsets))
    res2 = al.align((reads.reshape(-1), offsets))
    assert (res.records["status"] == 0).all()
    assert np.array_equal(res.records, res2.records) and np.array_equal(res.aln_read, res2.aln_read) and np.array_equal(res.aln_ref, res2.aln_ref)
    T = res.records["aln_len"].astype(np.int64)
    cols = np.arange(res.aln_read.shape[1])[None, :]
    valid = cols < T[:, None]
    rgap = (res.aln_read == ord("-")) & valid
    fgap = (res.aln_ref == ord("-")) & valid
    assert not (rgap & fgap).any()                                   # no double-gap column
    assert ((valid & ~rgap).sum(1) == L).all()                       # every read base appears exactly once
    assert ((valid & ~fgap).sum(1) == L).all()                       # every reference base appears exactly once
    amp_u8 = np.frombuffer(amp.encode(), dtype=np.uint8)
    # ungapped aligned strings are the inputs again (stable selection keeps order)
    assert np.array_equal(res.aln_read[valid & ~rgap].reshape(n, L), reads)
    assert np.array_equal(res.aln_ref[valid & ~fgap].reshape(n, L), np.broadcast_to(amp_u8, (n, L)))
    both = valid & ~rgap & ~fgap
    assert np.array_equal((both & (res.aln_read == res.aln_ref)).sum(1), res.records["matches"].astype(np.int64))
    assert np.array_equal(rgap.sum(1), res.records["all_deletion_bases"].astype(np.int64))
    rng = np.random.default_rng(L)
    for k in rng.integers(0, n, 300):
        rd = reads[k].tobytes().decode()
        st, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, -20, -2)
        assert st == 0 and res.strings(k) == (s1, s2) and int(res.records["matches"][k]) == mt
        check_record(res.records[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)


def _variant_equal(got, exp):
    """got: dict from crispresso2_amd.variants; exp: JSON form of the reference's dict."""
    from helpers import norm, PAYLOAD_FIELDS
    assert set(got.keys()) == set(exp.keys()), (sorted(got.keys()), sorted(exp.keys()))
    for k, e in exp.items():
        g = got[k]
        if k.startswith("variant_"):
            gd = g.__dict__ if not isinstance(g, dict) else g
            assert set(gd.keys()) == set(e.keys()), (k, sorted(gd.keys()), sorted(e.keys()))
            for f, ev in e.items():
                assert norm(gd[f]) == ev, (k, f, norm(gd[f]), ev)
        else:
            assert norm(g) == e, (k, norm(g), e)


def test_get_new_variant_objects_vs_reference_function(mats, ctx):
    """crispresso2_amd.variants.get_new_variant_objects against the reference's own get_new_variant_object
    (CRISPRessoCORE.py:627-798) on the reads of tests/FANC.Cas9.fastq: strand choice by seeds, reverse complements,
    two references with ambiguous reads, unaligned reads, legacy quantification and ignore flags."""
    import types
    from crispresso2_amd import refs as RF, variants as V
    cases = load_golden("variants.json.gz")
    n_amb = n_unal = n_rc = 0
    for case in cases:
        args = types.SimpleNamespace(**case["args"])
        refs, names = {}, []
        for r in case["refs"]:
            refs[r["name"]] = RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])
            assert refs[r["name"]]["fw_seeds"] == r["fw_seeds"] and refs[r["name"]]["rc_seeds"] == r["rc_seeds"]
            names.append(r["name"])
        got = V.get_new_variant_objects(args, case["reads"], refs, names, mats["EDNAFULL"], None, ctx=ctx)
        assert len(got) == len(case["variants"])
        for g, e in zip(got, case["variants"]):
            _variant_equal(g, e)
            n_unal += e["best_match_score"] <= 0
            n_amb += len(e.get("aln_ref_names", [])) > 1
            n_rc += any(e.get("variant_" + nm, {}).get("aln_strand") == "-" for nm in names)
    assert n_amb > 0 and n_unal > 0 and n_rc > 0


def test_process_fastq_equivalent(mats, ctx, tmp_path):
    """FASTQ -> unique-read dict -> one batch -> variantCache + aln_stats, against the same bookkeeping done per read."""
    import types
    from crispresso2_amd import refs as RF, variants as V
    case = load_golden("variants.json.gz")[0]
    args = types.SimpleNamespace(**case["args"])
    r = case["refs"][0]
    refs = {r["name"]: RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])}
    reads = case["reads"][:60]
    fq = tmp_path / "x.fastq"
    with open(fq, "w") as fh:
        for k, s in enumerate(reads + reads[:20] + reads[:5]):
            fh.write("@r%d\n%s\n+\n%s\n" % (k, s, "I" * len(s)))
    cache, not_aligned, st = V.process_fastq(str(fq), args, refs, [r["name"]], mats["EDNAFULL"], ctx=ctx)
    assert st["N_TOT_READS"] == 85
    assert st["N_COMPUTED_ALN"] + st["N_COMPUTED_NOTALN"] == len(set(reads))
    assert st["N_CACHED_ALN"] + st["N_CACHED_NOTALN"] == 85 - len(set(reads))
    exp = {s: v for s, v in zip(case["reads"], case["variants"])}
    for s, v in cache.items():
        assert v["count"] == (reads + reads[:20] + reads[:5]).count(s)
        e = dict(exp[s]); e["count"] = v["count"]
        _variant_equal(v, e)


def test_pooled_96_amplicons_counts_per_amplicon(mats, ctx):
    """BASELINE config 5 shape at test size: 96 amplicons, every read tagged with its amplicon id (CRISPRessoPooled
    semantics: each read is aligned to ITS amplicon only), per-amplicon count tensor vs the reference aggregation."""
    import torch
    from crispresso2_amd import synth, counts as C, _native
    from crispresso2_amd.batch import BatchAligner
    import oracle
    from oracle import aggregate
    m = mats["EDNAFULL"]
    L, n_amp, per = 250, 96, 40
    setups = [synth.amplicon_setup(L, 1000 + a) for a in range(n_amp)]
    reads = np.concatenate([synth.make_reads(L, per, amplicon_id=1000 + a, amplicon=setups[a][0]) for a in range(n_amp)])
    rids = np.repeat(np.arange(n_amp, dtype=np.uint16), per)
    perm = np.random.default_rng(0).permutation(len(reads))            # interleave amplicons: ids travel with the reads
    reads, rids = reads[perm], rids[perm]
    n = len(reads)
    al = BatchAligner([s[0] for s in setups], [s[1] for s in setups], [s[2] for s in setups], m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    d_rid = torch.from_numpy(rids.astype(np.int16)).to(dev)
    o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L,
                    d_ref_ids=d_rid.data_ptr(), stream=s)
    lay = C.CountLayout(n_amp, L, L)
    d_counts = torch.zeros(lay.shape(), dtype=torch.int64, device=dev)
    C.accumulate_device(ctx, lay, n, o1.data_ptr(), o2.data_ptr(), stride, rec.data_ptr(), d_counts.data_ptr(), stream=s)
    torch.cuda.synchronize()
    records = rec.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    assert (records["status"] == 0).all() and np.array_equal(records["ref_id"], rids)
    a1, a2, cnt = o1.cpu().numpy(), o2.cpu().numpy(), d_counts.cpu().numpy()
    for a in (0, 17, 95):
        amp, g, inc = setups[a]
        items = []
        for k in np.nonzero(rids == a)[0]:
            st, s1, s2, mt, ln = oracle.global_align_raw(reads[k].tobytes().decode(), amp, m, g, -20, -2)
            T = int(records["aln_len"][k])
            assert (a1[k, :T].tobytes().decode(), a2[k, :T].tobytes().decode()) == (s1, s2)
            p = oracle.find_indels_substitutions(s1, s2, inc)
            p["aln_seq"], p["aln_ref"] = s1, s2
            items.append((p, 1))
        exp = aggregate.aggregate(items, L)
        got = lay.unpack(cnt, a, L)
        for k, v in exp.items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(got[k], v), (a, k)
            else:
                assert got[k] == v, (a, k, got[k], v)
    assert sum(lay.unpack(cnt, a, L)["counts_total"] for a in range(n_amp)) == n


def test_three_candidate_references_best_reference_selection(mats, ctx):
    """BASELINE config 4 shape at test size: every read against 3 candidate amplicons (wild type, HDR, prime edit);
    best reference by the reference's rules (strictly higher score above min_aln_score; ties are ambiguous)."""
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    L, n = 250, 300
    amp, g, inc = synth.amplicon_setup(L)
    refs = [amp, synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")]
    gis = []
    for r in refs:
        x = np.zeros(len(r) + 1, dtype=np.int64)
        x[L // 2 + 1] = 1
        gis.append(x)
    reads = np.concatenate([synth.make_reads(L, n // 3, amplicon=r)[:, :L] if len(r) >= L else synth.make_reads(L, n // 3, amplicon=r)
                            for r in (refs[0], refs[1][:L], refs[2])])
    rd = [x.tobytes().decode() for x in reads]
    al = BatchAligner(refs, gis, [inc] * 3, m, -20, -2, ctx=ctx)
    res = al.align(rd, all_refs=True)
    sc = res.scores.reshape(len(rd), 3)
    best_gpu = []
    for k in range(len(rd)):
        exp = [oracle.global_align(rd[k], refs[r], m, gis[r], -20, -2) for r in range(3)]
        for r in range(3):
            assert res.strings(3 * k + r) == (exp[r][0], exp[r][1]) and sc[k, r] == exp[r][2]
        best, names = -1, []
        for r in range(3):                       # CRISPRessoCORE.py:697-707
            if exp[r][2] > best and exp[r][2] > 60:
                best, names = exp[r][2], [r]
            elif exp[r][2] == best:
                names.append(r)
        best_gpu.append(names)
    assert any(len(x) == 1 and x[0] == 1 for x in best_gpu) and any(len(x) == 1 and x[0] == 2 for x in best_gpu)
    # ... and c2_select_best_kernel on the records that are on the device: its member masks, aligned / ambiguous flags against the names above
    import torch
    from crispresso2_amd import counts as C
    dev = torch.device("cuda", 0)
    n_r = len(rd)
    d_rec = torch.from_numpy(res.records.view(np.uint8).reshape(-1, 32).copy()).to(dev)
    d_member = torch.zeros((n_r, 1), dtype=torch.int64, device=dev)
    d_use2 = torch.zeros((n_r, 1), dtype=torch.int64, device=dev)
    d_flags = torch.zeros(n_r, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(len(C.SELECT_STATS), dtype=torch.int64, device=dev)
    C.select_best_device(ctx, n_r, 3, d_rec.data_ptr(), C.min_mscore_table([60, 60, 60]), C.SELECT_DROP_AMBIGUOUS, int(res.records["aln_len"].max()),
                         d_member=d_member.data_ptr(), d_use2=d_use2.data_ptr(), d_flags=d_flags.data_ptr(), d_stats=d_stats.data_ptr())
    torch.cuda.synchronize()
    member = d_member.cpu().numpy().reshape(-1)
    fl = d_flags.cpu().numpy()
    for k in range(n_r):
        assert [r for r in range(3) if (int(member[k]) >> r) & 1] == best_gpu[k], (k, member[k], best_gpu[k])
        assert bool(fl[k] & 1) == (len(best_gpu[k]) > 0) and bool(fl[k] & 2) == (len(best_gpu[k]) > 1), k
    st = dict(zip(C.SELECT_STATS, d_stats.cpu().numpy().tolist()))
    assert st["N_COMPUTED_ALN"] == sum(1 for x in best_gpu if x) and st["n_bad_status"] == 0


def _default_args(**over):
    from types import SimpleNamespace
    a = SimpleNamespace(aln_seed_count=5, aln_seed_len=10, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                        use_legacy_insertion_quantification=False, ignore_deletions=False, ignore_insertions=False,
                        ignore_substitutions=False, assign_ambiguous_alignments_to_first_reference=False,
                        expand_ambiguous_alignments=False, prime_editing_pegRNA_scaffold_seq="", discard_indel_reads=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_whole_run_fanc_fastq_equals_the_reference_result_tables(mats, ctx, tmp_path):
    """The reference's own end-to-end test (tests/Makefile: CRISPResso -r1 FANC.Cas9.fastq -a ... -g ...): FASTQ file ->
    native ingest -> device alignments -> best-reference / strand rules -> count kernel, against the result tables the
    reference repository keeps for that run (quantification of editing frequency + nucleotide frequency table)."""
    import gzip
    import json
    from crispresso2_amd import pipeline, refs as RF
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fanc_run.json.gz"), "rt") as fh:
        g = json.load(fh)
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(g["fastq"])
    cut = g["cut_point"]
    ref = RF.make_ref("Reference", g["amplicon"], [cut], [cut, cut + 1], min_aln_score=60)
    res = pipeline.quantify_fastq(str(fq), {"Reference": ref}, ["Reference"], mats["EDNAFULL"], _default_args(), ctx=ctx)
    q, c = g["quantification"], res.per_ref["Reference"]
    assert res.stats["N_READS_INPUT"] == int(q["Reads_in_input"]) == res.stats["N_TOT_READS"]
    assert res.stats["N_TOTAL"] == int(q["Reads_aligned_all_amplicons"])
    got = {"Reads_aligned": c["counts_total"], "Unmodified": c["counts_unmodified"], "Modified": c["counts_modified"],
           "Discarded": c["counts_discarded"], "Insertions": c["counts_insertion"], "Deletions": c["counts_deletion"],
           "Substitutions": c["counts_substitution"], "Only Insertions": c["counts_only_insertion"],
           "Only Deletions": c["counts_only_deletion"], "Only Substitutions": c["counts_only_substitution"],
           "Insertions and Deletions": c["counts_insertion_and_deletion"],
           "Insertions and Substitutions": c["counts_insertion_and_substitution"],
           "Deletions and Substitutions": c["counts_deletion_and_substitution"],
           "Insertions Deletions and Substitutions": c["counts_insertion_and_deletion_and_substitution"]}
    assert got == {k_: int(q[k_]) for k_ in got}
    assert round(100.0 * c["counts_unmodified"] / c["counts_total"], 8) == float(q["Unmodified%"])
    assert list(g["amplicon"]) == g["nucleotide_frequency_reference_row"]
    for base in "ACGTN-":
        assert [float(x) for x in c["all_base_count_vectors_" + base]] == g["nucleotide_frequency"][base], base
    # the files themselves: byte for byte what the reference repository keeps for this run
    from crispresso2_amd import tables
    out = tmp_path / "CRISPResso_on_FANC.Cas9"
    names = tables.write_tables(res, {"Reference": ref}, ["Reference"], str(out))
    for fn, text in g["expected_files"].items():
        assert fn in names
        assert (out / fn).read_text() == text, fn
    mod = (out / "Modification_count_vectors.txt").read_text().split("\n")
    assert mod[0].split("\t")[1:] == list(g["amplicon"]) and mod[6].split("\t")[:3] == ["Total", "235", "235"]


def test_pipeline_equals_per_read_path_plus_reference_aggregation(mats, ctx):
    """pipeline.quantify_unique (device-resident; 32 bytes per alignment come back) against the per-read dict path
    (variants.get_new_variant_objects, itself pinned to the reference's function) fed through oracle/aggregate.py, with
    two similar references (ambiguous reads), reverse-complemented reads and their forward twins (rc merge), the
    ambiguity flags and ignore_substitutions."""
    import gzip
    import json
    from crispresso2_amd import pipeline, variants, refs as RF
    from oracle import aggregate
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "variants.json.gz"), "rt") as fh:
        cases = json.load(fh)
    base = [c for c in cases if c["label"] == "FANC+HDR"][0]
    reads = list(dict.fromkeys(base["reads"]))
    reads += [RF.reverse_complement(r) for r in reads[:25] if RF.reverse_complement(r) not in reads]      # rc twins of counted reads
    fanc = base["refs"][0]["sequence"]
    reads += [fanc[:80] + fanc[100:], fanc[:78] + fanc[101:]]          # the two references differ only inside 88..95: equal scores
    rng = np.random.default_rng(9)
    mult = rng.integers(1, 40, len(reads)).astype(np.uint32)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    for over in ({}, {"assign_ambiguous_alignments_to_first_reference": True}, {"expand_ambiguous_alignments": True},
                 {"ignore_substitutions": True}, {"discard_indel_reads": True}):
        args = _default_args(**over)
        refs = {r["name"]: RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], min_aln_score=r["min_aln_score"])
                for r in base["refs"]}
        names = [r["name"] for r in base["refs"]]
        res = pipeline.quantify_unique(arena, offsets, mult, refs, names, mats["EDNAFULL"], args, ctx=ctx)
        vs = variants.get_new_variant_objects(args, reads, refs, names, mats["EDNAFULL"], ctx=ctx)
        # the aggregation loop's bookkeeping (CRISPRessoCORE.py:3964-4000) on the per-read dicts
        cache = {r: dict(v, count=int(c)) for r, v, c in zip(reads, vs, mult) if v["best_match_score"] > 0}
        items = {nm: [] for nm in names}
        n_total = 0
        for r in list(cache):
            v = cache[r]
            if v["count"] == 0:
                continue
            rc = RF.reverse_complement(r)
            if rc in cache and cache[rc]["count"] > 0:
                v["count"] += cache[rc]["count"]
                cache[rc]["count"] = 0
            n_total += v["count"]
            if v["class_name"] == "AMBIGUOUS":
                continue
            for nm in v["aln_ref_names"]:
                items[nm].append((v["variant_" + nm], v["count"]))
        assert res.stats["N_TOTAL"] == n_total
        # the allele table rows (get_allele_row, :3926-3959; AMBIGUOUS_ / DISCARDED_ labels :3987-4000; sort and %Reads :4298-4303)
        exp_rows = []
        for r in cache:
            v = cache[r]
            if v["count"] == 0:
                continue

            def row(label, p):
                return (p["aln_seq"], p["aln_ref"], label, p["classification"], int(p["deletion_n"]), int(p["insertion_n"]),
                        int(p["substitution_n"]), v["count"], v["count"] / n_total * 100)
            if v["class_name"] == "AMBIGUOUS":
                exp_rows.append(row("AMBIGUOUS_" + v["aln_ref_names"][0], v["variant_" + v["aln_ref_names"][0]]))
                continue
            for nm in v["aln_ref_names"]:
                p = v["variant_" + nm]
                if args.discard_indel_reads and (p["deletion_n"] > 0 or p["insertion_n"] > 0):
                    exp_rows.append(row("DISCARDED_" + v["aln_ref_names"][0], p))
                else:
                    exp_rows.append(row(nm, p))
        exp_rows.sort(key=lambda t: (-t[7], t[0], t[1]))
        assert res.alleles() == exp_rows
        if not over:
            assert sum(v.get("class_name") == "AMBIGUOUS" for v in vs) >= 2 and res.stats["N_AMBIGUOUS"] > 0
        for nm, ref in zip(names, base["refs"]):
            exp = aggregate.aggregate(items[nm], len(ref["sequence"]), ignore_substitutions=args.ignore_substitutions,
                                      ignore_insertions=args.ignore_insertions, ignore_deletions=args.ignore_deletions,
                                      discard_indel_reads=args.discard_indel_reads)
            got = res.per_ref[nm]
            for k_, v_ in exp.items():
                if isinstance(v_, np.ndarray):
                    assert np.array_equal(got[k_][:len(ref["sequence"])], v_), (over, nm, k_)
                else:
                    assert got[k_] == v_, (over, nm, k_, got[k_], v_)


def test_paired_consensus_and_variants_vs_reference_functions(mats, ctx):
    """crispresso2_amd.paired against the reference's get_consensus_alignment_from_pairs (every call its own unit test makes
    + pairs cut from the FANC reads) and get_new_variant_object_from_paired (one / two references, both strands, unrelated
    reads, ambiguity flags, legacy quantification) -- goldens recorded from the reference (make_golden.py --paired)."""
    import types
    from crispresso2_amd import paired, refs as RF
    g = load_golden("paired.json.gz")
    calls = g["unit"] + g["fuzz"]
    assert len(g["unit"]) >= 15 and len(g["fuzz"]) >= 200
    ok = [c for c in calls if "raises" not in c]
    got = paired.consensus_batch([tuple(c["args"]) for c in ok], ctx=ctx)
    for c, o in zip(ok, got):
        assert list(o) == c["out"], (c, o)
    assert any(not c["out"][4] for c in ok) and any(c["out"][4] for c in ok)         # caching_is_ok both ways
    for c in calls:
        if "raises" in c:
            with pytest.raises(IndexError):
                paired.get_consensus_alignment_from_pairs(*c["args"])
    one = ok[0]
    assert list(paired.get_consensus_alignment_from_pairs(*one["args"])) == one["out"]
    n_amb = n_unal = 0
    for case in g["variants"]:
        args = types.SimpleNamespace(**case["args"])
        refs, names = {}, []
        for r in case["refs"]:
            refs[r["name"]] = RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])
            names.append(r["name"])
        vs = paired.get_new_variant_objects_from_paired(args, [tuple(p) for p in case["pairs"]], refs, names, mats["EDNAFULL"], None, ctx=ctx)
        assert len(vs) == len(case["variants"])
        for v, e in zip(vs, case["variants"]):
            _variant_equal(v, e)
            n_amb += len(e.get("aln_ref_names", [])) > 1 or e.get("class_name") == "AMBIGUOUS"
            n_unal += e["best_match_score"] <= 0
    assert n_amb >= 3 and n_unal >= 3


def test_variant_files_and_annotated_fastq_equal_the_reference_text(mats, ctx, tmp_path):
    """SURVEY 8(f)-4 end to end on the device: FASTQ -> unique reads -> device alignments + classifier -> the variants_<k>.tsv
    files of the reference's two-worker run and its --fastq_output file, byte for byte; both routes' statistics."""
    import gzip
    import types
    from crispresso2_amd import refs as RF, variants as V
    gold = load_golden("variant_io.json.gz")
    fq = tmp_path / "in.fastq"
    fq.write_text(gold["fastq"])
    for case in gold["cases"]:
        args = types.SimpleNamespace(**case["args"])
        refs, names = {}, []
        for r in case["refs"]:
            refs[r["name"]] = RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])
            names.append(r["name"])
        out = tmp_path / "out.fastq.gz"
        cache, not_aligned, st = V.process_fastq_write_out(str(fq), str(out), args, refs, names, mats["EDNAFULL"], ctx=ctx)
        with gzip.open(out, "rt") as fh:
            assert fh.read() == case["annotated"]
        assert st == case["single"]["aln_stats"]
        assert list(not_aligned) == case["single"]["not_aligned"] and list(cache) == case["single"]["aligned"]
        assert [cache[k]["count"] for k in cache] == case["single"]["counts"]
        # the sharded route, both "ranks" run here one after the other (no process group: the barrier is skipped)
        d = tmp_path / case["label"].replace(" ", "_").replace("+", "_")
        d.mkdir()
        assert V.process_fastq_sharded(str(fq), args, refs, names, mats["EDNAFULL"], str(d), ctx=ctx, rank=1, world=2) is None
        cache, not_aligned, st = V.process_fastq_sharded(str(fq), args, refs, names, mats["EDNAFULL"], str(d), ctx=ctx, rank=0, world=2)
        for k in range(2):
            assert (d / ("variants_%d.tsv" % k)).read_text() == case["tsv"][k]
        assert st == case["multi"]["aln_stats"]
        assert list(not_aligned) == case["multi"]["not_aligned"] and list(cache) == case["multi"]["aligned"]


def test_paired_fastq_files_equal_the_reference_run(mats, ctx, tmp_path):
    """process_paired_fastq on the device: two FASTQ files -> native paired ingest -> alignments, consensus kernel and
    classifier on the GPU -> the reference's variants_<k>.tsv files byte for byte, its per-occurrence second pass, final
    variantCache (keys, order, counts, every dict as JSON text) and statistics (make_golden.py --paired-fastq)."""
    import json
    import types
    from crispresso2_amd import refs as RF, paired as P, variant_io as IO
    gold = load_golden("paired_fastq.json.gz")
    p1, p2 = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    p1.write_text(gold["fastq1"])
    p2.write_text(gold["fastq2"])
    for case in gold["cases"]:
        args = types.SimpleNamespace(**case["args"])
        refs, names = {}, []
        for r in case["refs"]:
            refs[r["name"]] = RF.make_ref(r["name"], r["sequence"], r["cut_points"], r["include_idxs"], r["min_aln_score"])
            names.append(r["name"])
        d = tmp_path / ("v_" + case["label"].replace(" ", "_").replace("+", "_"))
        d.mkdir()
        assert P.process_paired_fastq(str(p1), str(p2), args, refs, names, mats["EDNAFULL"], ctx=ctx, variants_dir=str(d), rank=1, world=2) is None
        cache, not_aln, st = P.process_paired_fastq(str(p1), str(p2), args, refs, names, mats["EDNAFULL"], ctx=ctx, variants_dir=str(d),
                                                    rank=0, world=2)
        for k in range(2):
            assert (d / ("variants_%d.tsv" % k)).read_text() == case["tsv"][k]
        exp = case["result"]
        assert st == exp["aln_stats"]
        assert list(not_aln) == exp["not_aligned"] and list(cache) == exp["aligned"]
        assert [cache[k]["count"] for k in cache] == exp["counts"]
        assert [json.dumps(cache[k], cls=IO.CRISPRessoJSONEncoder) for k in cache] == exp["variants"]
        # one process, nothing on disk: same result
        cache1, not_aln1, st1 = P.process_paired_fastq(str(p1), str(p2), args, refs, names, mats["EDNAFULL"], ctx=ctx)
        assert st1 == st and list(cache1) == list(cache) and [cache1[k]["count"] for k in cache1] == exp["counts"]


def test_strand_plan_kernel_equals_the_host_seed_test(ctx):
    """c2_strand_plan_device (reads on the device, all references, one wavefront per read) = the host's c2_strand_plan on reads of both
    strands, chimeras, unrelated and very short reads, for several seed counts / lengths / thresholds."""
    import torch
    from types import SimpleNamespace
    from crispresso2_amd import _native, counts as C, refs as RF, synth
    amp, _g, inc = synth.amplicon_setup(250)
    amp2 = synth.make_variant(amp, "pe")[:-31]
    rng = np.random.default_rng(4)
    reads = []
    for r in synth.make_reads(250, 3000):
        s = r.tobytes().decode()
        kind = int(rng.integers(0, 6))
        if kind == 0:
            s = RF.reverse_complement(s)
        elif kind == 1:
            s = s[:int(rng.integers(1, 40))]
        elif kind == 2:
            s = "".join(rng.choice(list("ACGT"), int(rng.integers(5, 300))))
        elif kind == 3:
            s = s[:100] + RF.reverse_complement(s[100:])
        reads.append(s)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy()
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    dev = torch.device("cuda", 0)
    d_reads = torch.from_numpy(arena).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    for seed_count, seed_len, seed_min in ((5, 10, 2), (1, 10, 0), (9, 7, 4), (3, 40, 1), (0, 10, 0)):
        made = max(seed_count, 2)
        refs = {"A": RF.make_ref("A", amp, [125], inc, aln_seed_count=made, aln_seed_len=seed_len),
                "B": RF.make_ref("B", amp2, [110], [109, 110], aln_seed_count=made, aln_seed_len=seed_len)}
        names = ["A", "B"]
        d_plan = torch.full((len(reads) * 2,), 9, dtype=torch.uint8, device=dev)
        C.strand_plan_device(ctx, len(reads), d_reads.data_ptr(), d_off.data_ptr(), max(len(r) for r in reads), refs, names, seed_count, seed_min,
                             d_plan.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
        plan = d_plan.cpu().numpy().reshape(-1, 2)
        for r, name in enumerate(names):
            m = min(seed_count, len(refs[name]["fw_seeds"]))
            host = _native.strand_plan(arena, off, refs[name]["fw_seeds"][:m], refs[name]["rc_seeds"][:m], seed_min)
            assert np.array_equal(plan[:, r], host), (seed_count, seed_len, seed_min, name)
        if seed_count == 5:
            assert set(np.unique(plan)) == {0, 1, 2}


def test_count_reduce_through_the_c_abi_rccl(ctx):
    """c2_comm_unique_id / c2_comm_init / c2_reduce_counts on the one GPU of this box: a communicator of one rank, the all-reduce
    leaves the tensor as it is (the N-rank sum is covered by the gloo tests and by bench.py --gpus N through torch.distributed)."""
    import torch
    t = torch.arange(5000, dtype=torch.int64, device="cuda") * 3 - 7
    want = t.clone()
    ctx.comm_init(rank=0, world=1)
    ctx.reduce_counts(t.data_ptr(), t.numel(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    ctx.check(ctx.lib.c2_comm_destroy(ctx.handle), "c2_comm_destroy")


def test_consensus_host_call_in_several_pipelined_chunks_and_the_device_entry(mats, ctx):
    """c2_consensus_pairs_batch over more than two chunks of 65,536 pairs (pinned staging on three streams, results fetched as
    lengths + the written part of the rows): every pair's result equals what the same pair gives in a small call; and
    c2_consensus_pairs_device (every array already in HBM) writes the same rows and lengths."""
    import ctypes
    import torch
    from crispresso2_amd import paired
    g = load_golden("paired.json.gz")
    ok = [tuple(c["args"]) for c in g["unit"] + g["fuzz"] if "raises" not in c]
    want = paired.consensus_batch(ok, ctx=ctx)
    reps = 150_000 // len(ok) + 1
    rng = np.random.default_rng(5)
    order = rng.permutation(len(ok) * reps) % len(ok)
    got = paired.consensus_batch([ok[k] for k in order], ctx=ctx)
    assert len(got) == len(order) > 2 * 65536
    for k, o in zip(order[::97], got[::97]):
        assert o == want[k]
    assert all(o == want[k] for k, o in zip(order[-300:], got[-300:])) and all(o == want[k] for k, o in zip(order[65530:65545], got[65530:65545]))
    # device entry on the first 3000 pairs of that order
    items = [ok[k] for k in order[:3000]]
    n = len(items)
    n1 = np.array([len(it[1]) for it in items], dtype=np.int32); n2 = np.array([len(it[5]) for it in items], dtype=np.int32)
    lq1 = np.array([len(it[3]) for it in items], dtype=np.int32); lq2 = np.array([len(it[7]) for it in items], dtype=np.int32)
    stride = max(16, (int(max(n1.max(), n2.max())) + 15) // 16 * 16)
    qstride = max(16, (int(max(lq1.max(), lq2.max())) + 15) // 16 * 16)
    ostride = 2 * stride
    rows = paired._rows
    host = [rows([it[0][:len(it[1])] for it in items], stride), rows([it[1] for it in items], stride), rows([it[4][:len(it[5])] for it in items], stride),
            rows([it[5] for it in items], stride), n1, n2, rows([it[3] for it in items], qstride), rows([it[7] for it in items], qstride), lq1, lq2,
            np.array([1 if it[2] >= it[6] else 0 for it in items], dtype=np.uint8)]
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in host]
    d_oa = torch.zeros((n, ostride), dtype=torch.uint8, device=dev); d_or = torch.zeros_like(d_oa); d_oq = torch.zeros_like(d_oa)
    d_info = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ctx.check(ctx.lib.c2_consensus_pairs_device(ctx.handle, ctypes.c_uint64(n), P(d[0]), P(d[1]), P(d[2]), P(d[3]), ctypes.c_uint32(stride), P(d[4]), P(d[5]),
                                                P(d[6]), P(d[7]), ctypes.c_uint32(qstride), P(d[8]), P(d[9]), P(d[10]), P(d_oa), P(d_or), P(d_oq),
                                                ctypes.c_uint32(ostride), P(d_info), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "c2_consensus_pairs_device")
    torch.cuda.synchronize()
    info, oa, orf, oq = d_info.cpu().numpy(), d_oa.cpu().numpy(), d_or.cpu().numpy(), d_oq.cpu().numpy()
    for k in range(n):
        ln, lq, hom, fl = (int(x) for x in info[k])
        assert not fl & 2
        assert (oa[k, :ln].tobytes().decode(), oq[k, :lq].tobytes().decode(), orf[k, :ln].tobytes().decode(),
                round(float(100 * hom / float(ln)), 3), bool(fl & 1)) == want[order[k]], k


@pytest.mark.parametrize("L,force", [(250, True), (3000, False)])
def test_count_route_with_the_accumulator_block_in_hbm(mats, ctx, L, force, monkeypatch):
    """The count route beyond the LDS block (VERDICT r02 "limits the reference does not have": C2_E_TOO_LARGE above ~1,650 bp): a 3 kb
    amplicon takes c2_count_vectors_hbm_kernel (the workgroup's int32 block in HBM scratch); the 250-bp case forces the same kernel
    (C2_COUNT_HBM_BLOCK) and must give the LDS kernel's tensor.  Against oracle/aggregate.py on every read."""
    import torch
    from crispresso2_amd import synth, counts as C, _native
    from crispresso2_amd.batch import BatchAligner
    import oracle
    from oracle import aggregate
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(31 + L)
    n = 1500 if L == 250 else 160
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = list(range(L // 2 - 10, L // 2 + 10))
    reads = []
    for t in range(n):
        s_ = list(amp)
        kind = t % 4
        if kind == 1:
            d = int(rng.integers(1, 30)); p = L // 2 - int(rng.integers(0, d + 1)); del s_[p:p + d]
        elif kind == 2:
            p = L // 2; s_[p:p] = list(rng.choice(list("ACGT"), int(rng.integers(1, 12))))
        for q in np.nonzero(rng.random(len(s_)) < 0.004)[0]:
            s_[q] = "ACGTN"[int(rng.integers(0, 5))]
        reads.append("".join(s_))
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    res = al.align(reads)
    assert (res.records["status"] == 0).all()
    dev = torch.device("cuda", 0)
    o1, o2 = torch.from_numpy(res.aln_read).to(dev), torch.from_numpy(res.aln_ref).to(dev)
    rec = torch.from_numpy(res.records.view(np.uint8).reshape(-1, 32)).to(dev)
    w = rng.integers(0, 40, n).astype(np.uint32)
    d_w = torch.from_numpy(w.astype(np.int32)).to(dev)
    payloads = []
    for k in range(n):
        s1, s2 = res.strings(k)
        p = oracle.find_indels_substitutions(s1, s2, inc)
        p["aln_seq"], p["aln_ref"] = s1, s2
        payloads.append(p)
    max_lj = max(len(r) for r in reads)
    lay = C.CountLayout(1, L, max_lj)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        d_counts = torch.zeros(lay.shape(), dtype=torch.int64, device=dev)
        C.accumulate_device(ctx, lay, n, o1.data_ptr(), o2.data_ptr(), res.aln_read.shape[1], rec.data_ptr(), d_counts.data_ptr(),
                            d_weights=d_w.data_ptr(), stream=s)
        torch.cuda.synchronize()
        return d_counts.cpu().numpy()
    if force:
        plain = run()
        monkeypatch.setenv("C2_COUNT_HBM_BLOCK", "1")
    counts = run()
    if force:
        assert np.array_equal(counts, plain)
    got = lay.unpack(counts, 0, L)
    exp = aggregate.aggregate([(p, int(c)) for p, c in zip(payloads, w) if c > 0], L)
    for k_, v in exp.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(got[k_][:L], v), k_
        else:
            assert got[k_] == v, (k_, got[k_], v)
    assert got["counts_total"] == int(w.sum()) and got["counts_deletion"] > 0 and got["counts_insertion"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("read_len,expect_skipped", [(150, 3), (215, 2), (240, 0)])
def test_band_tiers_no_read_of_the_batch_can_use_are_not_launched(mats, ctx, read_len, expect_skipped):
    """c2_batch.min_read_len: mates of 150 bp against a 250-bp amplicon cannot use the 32- and 62-diagonal tiers (diagonal 0 and diagonal
    len(ref) - len(read) = 100 do not fit one band): with the hint those tiers are not launched -- the same strings and records as without
    it, and as the oracle's on a sample -- and fewer tiers in c2_tier_info (round 5: the chain behind the partition has four band tiers, 32 / 40 / 62 /
    128 diagonals: three of them go); 215 bp: the first tier goes, and with it the partition and the 40-diagonal tier that only runs behind it; 240 bp: none."""
    import torch
    from crispresso2_amd import synth, _native
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    L, n = 250, 4000
    amp, g, _ = synth.amplicon_setup(L)
    inc = list(range(L // 2 - 10, L // 2 + 10))
    full = synth.make_reads(L, n)
    start = (L - read_len) // 2
    reads = np.ascontiguousarray(full[:, start:start + read_len])
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    stride = al.stride_for(read_len)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * read_len
    s = torch.cuda.current_stream().cuda_stream
    outs, tiers = [], []
    for hint in (0, read_len):
        o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
        al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, read_len, stream=s, min_read_len=hint)
        torch.cuda.synchronize()
        outs.append((o1.cpu().numpy(), o2.cpu().numpy(), rec.cpu().numpy()))
        tiers.append(len(ctx.tier_info()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert tiers[0] - tiers[1] == expect_skipped, tiers
    records = outs[1][2].view(_native.REC_DTYPE).reshape(-1)
    assert (records["status"] == 0).all()
    for i in range(0, n, 97):
        T = int(records["aln_len"][i])
        s1, s2, _ = oracle.global_align(reads[i].tobytes().decode(), amp, m, g, -20, -2)
        assert outs[1][0][i, :T].tobytes().decode() == s1 and outs[1][1][i, :T].tobytes().decode() == s2, i


@pytest.mark.gpu
def test_score_only_stage_changes_no_result(mats, ctx, monkeypatch):
    """The stage in front of the first band tier (c2_align_partition_kernel + c2_align_diags_kernel<8>: the packed fill without pointer bits over
    the reads predicted to align along the main diagonal) against the same batch with the stage switched off, and against the oracle: reads equal
    to the amplicon, with substitutions only, with indels in the middle (predicted right: never in the stage), and the WRONG predictions -- an
    indel inside the last 32 columns, compensated so that the read keeps the amplicon's length, and a substitution-rich tail on a gap-free read."""
    import torch
    from crispresso2_amd import synth, _native
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    L = 250
    amp, g, _ = synth.amplicon_setup(L)
    inc = list(range(L // 2 - 10, L // 2 + 10))
    rng = np.random.default_rng(11)
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}

    def subs(s, ps):
        s = list(s)
        for p_ in ps:
            s[p_] = other[s[p_]]
        return "".join(s)
    reads = [amp, subs(amp, [3, 100]), subs(amp, [249]), subs(amp, range(222, 250, 3))]             # gap-free; the last one with a busy tail
    for cut in (30, 125, 200, 236, 243, 246, 248):
        for d in (1, 2, 3, 7):
            reads.append((amp[:cut] + amp[cut + d:] + "ACGTACGTAC"[:d])[:L])                             # deletion, the length kept by bases behind the end
            reads.append((amp[:cut] + "TGCATGCATG"[:d] + amp[cut:])[:L])                                 # insertion, the end cut off
    reads += [synth.make_reads(L, 400)[i].tobytes().decode() for i in range(400)]
    n = len(reads)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(arena.copy()).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    s = torch.cuda.current_stream().cuda_stream
    outs, infos = [], []
    for off in (False, True):
        if off:
            monkeypatch.setenv("C2_NO_SCORE_TIER", "1")
        o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
        al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L, stream=s)
        torch.cuda.synchronize()
        infos.append(ctx.score_stage_info() + (ctx.partition_info()["finished_by_partition"],))
        outs.append((o1.cpu().numpy(), o2.cpu().numpy(), rec.cpu().numpy()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    ran, took, finished, by_partition = infos[0]
    assert ran and not infos[1][0]
    # (the partition itself finishes the main-diagonal reads with at most two differing bases; the score-only launch takes the rest of class 0
    #  and cannot finish all of it: the wrong predictions)
    assert by_partition > 100 and 10 < finished < took and took + by_partition < n, (took, finished, by_partition)
    records = outs[0][2].view(_native.REC_DTYPE).reshape(-1)
    assert (records["status"] == 0).all()
    for i in range(n):
        T = int(records["aln_len"][i])
        s1, s2, _ = oracle.global_align(reads[i], amp, m, g, -20, -2)
        assert outs[0][0][i, :T].tobytes().decode() == s1 and outs[0][1][i, :T].tobytes().decode() == s2, (i, reads[i])


@pytest.mark.gpu
@pytest.mark.parametrize("L", [250, 150])
def test_partition_routing_changes_no_result(mats, ctx, monkeypatch, L):
    """c2_align_partition_kernel's classes on the device: the benchmark's kinds of reads (deletions of 2 .. 40 bases whose place in the read is
    taken by bases behind the amplicon's end, insertions, shorter reads) plus synthetic ones -- with the routing (long indels straight to the tier
    whose band holds them), without it (C2_NO_ROUTE=1), with the opt-in 14-diagonal launch (C2_P16_TIER=1): the same bytes every time, and the
    oracle's alignment for every read; c2_partition_info says that every class the setting allows saw tasks."""
    import torch
    from crispresso2_amd import synth, _native
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    amp, g, _ = synth.amplicon_setup(L)
    inc = list(range(L // 2 - 10, L // 2 + 10))
    rng = np.random.default_rng(77 + L)
    cut = L // 2
    reads = []
    for k in range(640):
        t = list(amp)
        for _ in range(int(rng.integers(0, 3))):
            t[int(rng.integers(0, L))] = str(rng.choice(list("ACGT")))
        t = "".join(t)
        kind = k % 8
        d = [0, 2, 5, 9, 20, 40, 3, 25][kind]
        if kind < 6:
            t = (t[:cut - d // 2] + t[cut - d // 2 + d:] + "".join(rng.choice(list("ACGT"), d)))[:L]
        elif kind == 6:
            t = (t[:cut] + "".join(rng.choice(list("ACGT"), d)) + t[cut:])[:L]
        else:
            t = t[:cut - 10] + t[cut - 10 + d:]
        reads.append(t)
    reads += [synth.make_reads(L, 384)[i].tobytes().decode() for i in range(384)]
    n = len(reads)
    arena = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(arena.copy()).to(dev)
    d_off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    outs, infos = [], []
    for env in ({}, {"C2_NO_ROUTE": "1"}, {"C2_P16_TIER": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
        al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L, stream=s)
        torch.cuda.synchronize()
        infos.append(ctx.partition_info())
        outs.append((o1.cpu().numpy(), o2.cpu().numpy(), rec.cpu().numpy()))
        for k_ in env:
            monkeypatch.delenv(k_)
    for other in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], other))
    routed, unrouted, with16 = infos
    assert routed["ran"] and not routed["p16"] and sum(routed["classes"]) == n
    # (classes: 0 score-only, 1 the 14-diagonal launch, 2 / 3 / 4 / 5 the band tiers of 32 / 40 / 62 / 128 diagonals, 6 the full-matrix launch)
    assert routed["classes"][0] > 100 and routed["classes"][1] == 0 and routed["classes"][3] + routed["classes"][4] >= 60 and (routed["classes"][5] >= 40 or L == 150), routed
    assert sum(unrouted["classes"][3:6]) == 0 and unrouted["classes"][2] > routed["classes"][2], unrouted
    # (round 6: a band is chosen only if its launch can be expected to CERTIFY the alignment -- fewer tasks take the 14-diagonal launch, nearly all of them finish there)
    assert with16["p16"] and with16["classes"][1] >= 100 and with16["finished"][1] >= with16["classes"][1] * 3 // 4, with16
    records = outs[0][2].view(_native.REC_DTYPE).reshape(-1)
    assert (records["status"] == 0).all()
    for i in range(n):
        T = int(records["aln_len"][i])
        s1, s2, _ = oracle.global_align(reads[i], amp, m, g, -20, -2)
        assert outs[0][0][i, :T].tobytes().decode() == s1 and outs[0][1][i, :T].tobytes().decode() == s2, (i, reads[i])


@pytest.mark.gpu
def test_ragged_and_unrelated_reads_and_the_full_matrix_launch_with_its_plane_in_hbm(mats, ctx, monkeypatch):
    """Round 5, inputs that are not the generator's best case: 6,000 reads cut to lengths U[200, 250], every tenth one replaced by a random sequence,
    against the 250-bp amplicon through the default chain of a BATCH (>= 4,096 tasks: the full-matrix launch keeps its pointer plane in HBM scratch,
    four wavefronts per SIMD instead of one).  The partition orders the chunks by read length and sends the unrelated reads straight to the last
    list (class 6).  A sample of every kind against the oracle; and every byte equal to the same batch with the three knobs turned the other way
    (plane in LDS, task order, no direct route) -- results never depend on them."""
    import torch
    from crispresso2_amd import synth
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = mats["EDNAFULL"]
    L, n = 250, 6000
    amp, g, inc = synth.amplicon_setup(L)
    rng = np.random.default_rng(2025)
    base = synth.make_reads(L, n)
    reads = []
    for k in range(n):
        s = base[k].tobytes().decode()[:int(rng.integers(200, L + 1))]
        if k % 10 == 7:
            s = "".join(rng.choice(list("ACGT"), len(s)))
        reads.append(s)
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    res = al.align(reads)
    part = ctx.partition_info()
    tiers = ctx.tier_info()
    assert part["ran"] and sum(part["classes"]) == n and 500 <= part["classes"][6] <= 600, part
    assert tiers[-1] >= part["classes"][6]                             # they sit in the list the full-matrix launch reads
    assert (res.records["status"] == 0).all()
    for k in list(range(0, n, 37)) + list(range(7, n, 310)):
        st, s1, s2, mt, ln = oracle.global_align_raw(reads[k], amp, m, g, -20, -2)
        assert st == 0 and res.strings(k) == (s1, s2) and int(res.records["matches"][k]) == mt, k
        p = oracle.find_indels_substitutions(s1, s2, inc)
        r = res.records[k]
        assert (r["insertion_n"], r["deletion_n"], r["substitution_n"]) == (p["insertion_n"], p["deletion_n"], p["substitution_n"]), k
    for knob in ("C2_FULL_PLANE_IN_LDS", "C2_NO_LENGTH_ORDER", "C2_NO_DIRECT_FULL"):
        monkeypatch.setenv(knob, "1")
        other = al.align(reads)
        monkeypatch.delenv(knob)
        assert np.array_equal(other.records, res.records) and np.array_equal(other.aln_read, res.aln_read) and np.array_equal(other.aln_ref, res.aln_ref), knob


def test_gpu_hinted_count_of_an_all_references_batch(mats, ctx, monkeypatch):
    """Every read against every amplicon (config 4's shape: task = read * n_refs + reference) with weights as the selection leaves them -- none for most
    tasks, a read's multiplicity for its best amplicon: the count tensor with the hint words equals the one without them and the one with the list of
    left-over tasks switched off; three amplicons of two lengths."""
    import torch
    from crispresso2_amd import synth, _native, counts as C
    from crispresso2_amd.batch import BatchAligner
    m = mats["EDNAFULL"]
    L = 250
    amp, g, inc = synth.amplicon_setup(L)
    refs = [amp, synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")]
    gis = []
    for r in refs:
        x = np.zeros(len(r) + 1, dtype=np.int64); x[L // 2 + 1] = 1
        gis.append(x)
    incs = [inc] * 3
    n = 40000
    blocks = [synth.make_reads(L, n // 4 if src else n // 2, amplicon_id=100 + src, amplicon=refs[src][:L]) for src in range(3)]
    reads = np.concatenate(blocks)
    rng = np.random.default_rng(12)
    reads = np.ascontiguousarray(reads[rng.permutation(len(reads))])
    n = len(reads)
    k = 3
    nt = n * k
    al = BatchAligner(refs, gis, incs, m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    Lmax = max(len(r) for r in refs)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    o1 = torch.zeros((nt, stride), dtype=torch.uint8, device=dev)
    o2 = torch.zeros((nt, stride), dtype=torch.uint8, device=dev)
    rec = torch.zeros((nt, 32), dtype=torch.uint8, device=dev)
    hints = torch.zeros((nt * 4,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L, all_refs=True, stream=s, d_hints=hints.data_ptr())
    torch.cuda.synchronize()
    records = rec.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    assert (records["status"] == 0).all() and (records["ref_id"] == np.tile(np.arange(k), n)).all()
    h0 = hints.cpu().numpy().view(np.uint32).reshape(nt, 4)[:, 0]
    assert ((h0 >> 30) != 0).sum() > nt // 4
    # weights as the selection leaves them: the best amplicon (most matches per column) gets the read's multiplicity, the others nothing
    score = records["matches"].astype(np.float64) / np.maximum(records["aln_len"], 1)
    best = score.reshape(n, k).argmax(axis=1)
    w = np.zeros((n, k), dtype=np.uint32)
    w[np.arange(n), best] = rng.integers(1, 30, n).astype(np.uint32)
    w[7, best[7]] = 5000; w[9, best[9]] = 1024
    d_w = torch.from_numpy(w.reshape(-1).view(np.int32)).to(dev)
    layout = C.CountLayout(k, Lmax, L + 16)

    def count(use_hints):
        t = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)
        C.accumulate_device(ctx, layout, nt, o1.data_ptr(), o2.data_ptr(), stride, rec.data_ptr(), t.data_ptr(), d_weights=d_w.data_ptr(),
                            flags=C.FLAG_ALL_REFS_LAYOUT, stream=s, d_hints=hints.data_ptr() if use_hints else None)
        torch.cuda.synchronize()
        return t.cpu().numpy()
    plain = count(False)
    assert plain.sum() > 0 and all(plain[r].sum() > 0 for r in range(k))
    assert np.array_equal(plain, count(True))
    monkeypatch.setenv("C2_NO_COUNT_REST_LIST", "1")
    assert np.array_equal(plain, count(True))


@pytest.mark.gpu
def test_gpu_hinted_count_with_several_references(mats, ctx, monkeypatch):
    """CRISPRessoPooled's shape on the hardware: 7 amplicons of different lengths, 84,000 reads tagged with their amplicon (interleaved, and one amplicon
    without a read).  The hinted kernel runs per reference over the grouped order and leaves a list per reference that c2_rest_compact_kernel closes up;
    the count tensor equals the one without hints and the one with the list switched off (C2_NO_COUNT_REST_LIST=1), with and without weights."""
    import torch
    from crispresso2_amd import synth, _native, counts as C
    from crispresso2_amd.batch import BatchAligner
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(777)
    lens = [250, 200, 223, 250, 180, 240, 250]
    amps, gs, incs = [], [], []
    for k, L in enumerate(lens):
        a = "".join(rng.choice(list("ACGT"), L))
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
        amps.append(a); gs.append(g); incs.append(list(range(L // 2 - 5, L // 2 + 6)))
    n = 84000
    rids = rng.integers(0, len(lens), n).astype(np.uint16)
    rids[rids == 4] = 5                                             # (amplicon 4 has no read)
    chunks, offs = [], [0]
    for k in range(n):
        a = np.frombuffer(amps[rids[k]].encode(), dtype=np.uint8).copy()
        kind = k % 6
        if kind in (1, 2):                                          # one / two substitutions
            for _ in range(kind):
                a[int(rng.integers(0, len(a)))] = ord("ACGTN"[int(rng.integers(0, 5))])
        elif kind == 3:                                             # a deletion at the cut, fixed length
            d = int(rng.integers(1, 20)); c = len(a) // 2
            a = np.concatenate([a[:c], a[c + d:], rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), d)])
        elif kind == 4:                                             # an insertion at the cut, fixed length
            d = int(rng.integers(1, 10)); c = len(a) // 2
            a = np.concatenate([a[:c], rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), d), a[c:]])[:len(a)]
        chunks.append(a); offs.append(offs[-1] + len(a))
    arena = np.concatenate(chunks)
    al = BatchAligner(amps, gs, incs, m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    Lmax = max(lens)
    stride = al.stride_for(Lmax)
    d_reads = torch.from_numpy(arena).to(dev)
    d_off = torch.from_numpy(np.array(offs, dtype=np.int64)).to(dev)
    d_rid = torch.from_numpy(rids.view(np.int16)).to(dev)
    o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    hints = torch.zeros((n * 4,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, Lmax, d_ref_ids=d_rid.data_ptr(),
                    stream=s, d_hints=hints.data_ptr())
    torch.cuda.synchronize()
    records = rec.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    assert (records["status"] == 0).all() and (records["ref_id"] == rids).all()
    h0 = hints.cpu().numpy().view(np.uint32).reshape(n, 4)[:, 0]
    assert ((h0 >> 30) != 0).sum() > n // 2
    layout = C.CountLayout(len(lens), Lmax, Lmax + 16)
    w = rng.integers(0, 30, n).astype(np.uint32)
    w[5] = 70000; w[11] = 1024; w[17] = 0x90000000
    d_w = torch.from_numpy(w.view(np.int32)).to(dev)

    def count(use_hints, weights):
        t = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)
        C.accumulate_device(ctx, layout, n, o1.data_ptr(), o2.data_ptr(), stride, rec.data_ptr(), t.data_ptr(), d_weights=d_w.data_ptr() if weights else None,
                            stream=s, d_hints=hints.data_ptr() if use_hints else None)
        torch.cuda.synchronize()
        return t.cpu().numpy()
    for weights in (False, True):
        plain = count(False, weights)
        assert plain.sum() > 0
        assert np.array_equal(plain, count(True, weights)), weights
        monkeypatch.setenv("C2_NO_COUNT_REST_LIST", "1")
        assert np.array_equal(plain, count(True, weights)), weights
        monkeypatch.delenv("C2_NO_COUNT_REST_LIST")


@pytest.mark.parametrize("L", [250, 150])
def test_gpu_hinted_count_equals_the_count_over_the_rows(mats, ctx, L):
    """Round 6: c2_batch.diag_hints + c2_count_vectors_hinted_device on the hardware.  60,000 synthetic reads (a third of them copies of the amplicon or one /
    two bases away: finished by the partition, hinted) with weights (0, small, above the hinted kernel's LDS limit, above 2^31): the count tensor with
    the hints equals the tensor without them entry by entry -- with no gate and with a gate two differing bases fail, with and without
    --ignore_substitutions --, also after the rows of the hinted tasks were overwritten (they are not read); every hint restates its record."""
    import torch
    from crispresso2_amd import synth, _native, counts as C
    from crispresso2_amd.batch import BatchAligner
    m = mats["EDNAFULL"]
    n = 60000
    amp, g, inc = synth.amplicon_setup(L)
    reads = synth.make_reads(L, n)
    al = BatchAligner([amp], [g], [inc], m, -20, -2, ctx=ctx)
    dev = torch.device("cuda", 0)
    stride = al.stride_for(L)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    hints = torch.full((n * 4,), 0x5a5a5a5a, dtype=torch.int32, device=dev)      # four words per task
    s = torch.cuda.current_stream().cuda_stream
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, L, stream=s, d_hints=hints.data_ptr())
    torch.cuda.synchronize()
    h4 = hints.cpu().numpy().view(np.uint32).reshape(n, 4)
    h = h4[:, 0]
    records = rec.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
    valid = (h >> 31) == 1
    gapped = ((h >> 30) & 3) == 1
    assert valid.sum() == ctx.partition_info()["finished_by_partition"] > n // 4 and (h4[~valid & ~gapped] == 0).all() and gapped.sum() > n // 8
    kk = ((h >> 24) & 3)[valid]
    assert (records["aln_len"][valid] == L).all() and (records["matches"][valid].astype(np.int64) + kk == L).all()
    amp_u8 = np.frombuffer(amp.encode(), dtype=np.uint8)
    for t in np.nonzero(valid)[0][:2000]:                            # positions and read bases of the differing columns
        diff = np.nonzero(reads[t] != amp_u8)[0]
        assert len(diff) == int((h[t] >> 24) & 3)
        for e, c in enumerate(diff):
            assert int((h[t] >> (12 * e)) & 0x1ff) == c and ord("ACTG???N"[int((h[t] >> (12 * e + 9)) & 7)]) == reads[t][c]
    rng = np.random.default_rng(5)
    w = rng.integers(0, 40, n).astype(np.uint32)
    first = np.nonzero(valid)[0]
    w[first[0]] = 70000; w[first[1]] = 0x90000000; w[first[2]] = 65536; w[first[3]] = 65535
    gfirst = np.nonzero(gapped)[0]
    w[gfirst[0]] = 1024; w[gfirst[1]] = 5000; w[gfirst[2]] = 1023; w[gfirst[3]] = 0x90000000     # (at / above the hinted kernel's weight limit: the column walk's)
    d_w = torch.from_numpy(w.view(np.int32)).to(dev)
    layout = C.CountLayout(1, L, L)

    def count(a, f, use_hints, mm, flags):
        t = torch.zeros(layout.shape(), dtype=torch.int64, device=dev)
        C.accumulate_device(ctx, layout, n, a.data_ptr(), f.data_ptr(), stride, rec.data_ptr(), t.data_ptr(), d_weights=d_w.data_ptr(), min_matches=mm,
                            flags=flags, stream=s, d_hints=hints.data_ptr() if use_hints else None)
        torch.cuda.synchronize()
        return t.cpu().numpy()
    w1, w2 = o1.clone(), o2.clone()
    vt = torch.from_numpy(valid | (gapped & (w < 1024))).to(dev)          # (a gapped hint is used below the hinted kernel's weight limit)
    w1[vt] = 0x58; w2[vt] = 0x59
    for mm in (None, C.min_matches_table([99.3], L + L)):
        for flags in (0, C.FLAG_IGNORE_SUBSTITUTIONS, C.FLAG_DISCARD_INDEL_READS):
            plain = count(o1, o2, False, mm, flags)
            assert plain.sum() > 0
            assert np.array_equal(plain, count(o1, o2, True, mm, flags)), (L, mm is None, flags)
            assert np.array_equal(plain, count(w1, w2, True, mm, flags)), (L, mm is None, flags)
