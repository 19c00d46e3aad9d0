"""The HIP kernel SOURCE (crispresso2_amd/csrc/c2_kernels.hip), compiled unchanged for the host by
the wave emulator in tests/emu/, against the golden vectors and the oracle (CPU only).

This checks the kernel's logic -- systolic schedule, tie rules, pointer packing, wave-parallel
traceback, fused classification, multi-pass references -- in the GPU-less container.  It is a test
harness, not a product path; the `-m gpu` tests are the parity tests proper.
"""
import os

import numpy as np
import pytest

import emu_driver as E
import oracle
from helpers import adversarial_case, load_golden, matrices


@pytest.fixture(scope="module")
def mats():
    E.build()
    return matrices()


def check_record(rec, payload, s1, s2):
    """summary record vs the reference's payload + the derived fields of CRISPRessoCORE.py:726-744"""
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


def run_vectors(vecs, mats, force_R=0, batch=64, no_packed=False, band_lanes=0, stats=None):
    """Group vectors that share (ref, gap_incentive, matrix, params) into one emulated launch each."""
    groups = {}
    for v in vecs:
        key = (v["seqi"], tuple(v["gap_incentive"]), v["matrix"], v["gap_open"], v["gap_extend"], tuple(v.get("include", [])))
        groups.setdefault(key, []).append(v)
    n = 0
    for (seqi, g, mat, go, ge, inc), vs in groups.items():
        for b in range(0, len(vs), batch):
            chunk = vs[b:b + batch]
            res, rec = E.align_batch([v["seqj"] for v in chunk], [seqi], [list(g)], [list(inc)], mats[mat], go, ge,
                                     force_R=force_R, grid=min(len(chunk), 3), no_packed=no_packed,
                                     band_lanes=band_lanes, stats=stats)
            for v, (s1, s2), r in zip(chunk, res, rec):
                assert r["status"] == 0, (v, r)
                assert [s1, s2] == v["out"][:2], v
                assert round(100 * int(r["matches"]) / float(int(r["aln_len"])), 3) == v["out"][2]
                payload = v.get("payload") or oracle.find_indels_substitutions(s1, s2, list(inc))
                check_record(r, payload, s1, s2)
                n += 1
    return n


def test_emulated_kernel_reference_unit_test_answers(mats):
    kats = [k for k in load_golden("ref_unit_kats.json") if k["fn"] == "global_align"]
    assert run_vectors(kats, mats) == len(kats)


def test_emulated_kernel_fuzz_vectors(mats):
    vecs = load_golden("fuzz_align.json")
    assert run_vectors(vecs, mats) == len(vecs)


def test_emulated_kernel_realistic_vectors(mats):
    vecs = load_golden("realistic.json")
    assert run_vectors(vecs, mats) == len(vecs)


def test_emulated_kernel_lds_score_table_path(mats):
    """EDNAFULL normally takes the packed-nibble score rows; force the general LDS score-table path too."""
    vecs = load_golden("realistic.json")[::4]
    assert run_vectors(vecs, mats, no_packed=True) == len(vecs)
    assert run_vectors(vecs[:10], mats, no_packed=True, force_R=2) == 10


@pytest.mark.parametrize("band_lanes", [1, 3, 8])
def test_emulated_kernel_banded_pointer_plane_with_fallback(mats, band_lanes):
    """Banded launch (only lanes near the main diagonal keep their pointer words) + full-plane launch over the tasks
    whose traceback left the band: the union must be identical to the full kernel, and both launches must be used."""
    st = {}
    vecs = load_golden("realistic.json")
    assert run_vectors(vecs, mats, band_lanes=band_lanes, stats=st) == len(vecs)
    small = load_golden("fuzz_align.json")[::5]
    assert run_vectors(small, mats, band_lanes=band_lanes, stats=st) == len(small)
    assert 0 < st["fallback"] < st["tasks"], st


def test_emulated_diagonal_band_kernel_with_certificate_and_fallback(mats):
    """c2_align_diag_kernel (lanes = diagonals, 128-diagonal band, proof of optimality after the fill) + the full-plane
    kernel for the tasks whose certificate fails: identical to the reference on every vector; both paths must be used."""
    st = {}
    vecs = load_golden("realistic.json")
    assert run_vectors(vecs, mats, band_lanes=-1, stats=st) == len(vecs)
    certified_realistic = st["tasks"] - st["fallback"]
    assert certified_realistic > len(vecs) // 2, st          # most amplicon reads are certified inside the band
    vecs = load_golden("fuzz_align.json")
    assert run_vectors(vecs, mats, band_lanes=-1, stats=st) == len(vecs)
    kats = [k for k in load_golden("ref_unit_kats.json") if k["fn"] == "global_align"]
    assert run_vectors(kats, mats, band_lanes=-1, stats=st) == len(kats)
    assert 0 < st["fallback"] < st["tasks"], st


@pytest.mark.parametrize("mode", [-2, -4, -5, -7, -75, -8, -84, -87, -82])
def test_emulated_multi_alignment_diagonal_kernels(mats, mode):
    """c2_align_diagx_kernel: 2 (-2) or 4 (-4) alignments per wavefront, lane groups isolated by an EXEC-disabled lane,
    pointer words in a global scratch plane; -7 is the host library's whole chain 4 -> 2 -> 1 -> full-plane kernel.
    Identical to the reference on every vector; the narrow bands must certify a good share of the amplicon reads."""
    st = {}
    vecs = load_golden("realistic.json")
    assert run_vectors(vecs, mats, band_lanes=mode, stats=st) == len(vecs)
    assert st["tasks"] - st["fallback"] > len(vecs) // 4, st
    vecs = load_golden("fuzz_align.json")[::3]
    assert run_vectors(vecs, mats, band_lanes=mode, stats=st) == len(vecs)
    kats = [k for k in load_golden("ref_unit_kats.json") if k["fn"] == "global_align"]
    assert run_vectors(kats, mats, band_lanes=mode, stats=st) == len(kats)
    assert 0 < st["fallback"] < st["tasks"], st


@pytest.mark.parametrize("mode", [-1, -2, -4, -5, -7, -75, -8, -84, -87, -82])
def test_emulated_diagonal_band_kernel_unequal_lengths_rc_and_multi_ref(mats, mode):
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(77)
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in (180, 223, 140)]
    gis = [np.zeros(len(r) + 1, dtype=np.int64) for r in refs]
    for g in gis:
        g[len(g) // 2] = 1
    incs = [list(range(len(r) // 2 - 5, len(r) // 2 + 5)) for r in refs]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    reads, rids, strands, truth = [], [], [], []
    for k in range(36):
        r = k % 3
        s = list(refs[r])
        p = int(rng.integers(10, len(s) - 40))
        if k % 4 == 0:
            del s[p:p + int(rng.integers(1, 30))]
        elif k % 4 == 1:
            s[p:p] = list(rng.choice(list("ACGT"), int(rng.integers(1, 12))))
        elif k % 4 == 2:
            s = s[int(rng.integers(0, 25)):] + list(rng.choice(list("ACGT"), int(rng.integers(0, 30))))
        s[int(rng.integers(0, len(s)))] = "N"
        fw = "".join(s)
        rc = k % 2
        reads.append("".join(comp[c] for c in reversed(fw)) if rc else fw)
        rids.append(r); strands.append(rc); truth.append(fw)
    st = {}
    res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, ref_ids=rids, strands=strands, band_lanes=mode, stats=st)
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        exp = oracle.global_align_raw(truth[k], refs[rids[k]], m, gis[rids[k]], -20, -2)
        assert r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], k
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[rids[k]]), s1, s2)
    assert st["fallback"] < st["tasks"]


@pytest.mark.parametrize("chain", [-7, -87])
@pytest.mark.parametrize("go,ge,scale", [(-20, -2, 1), (-20, -4, 3), (-6, -2, 1)])
def test_emulated_chain_adversarial_gap_incentives(mats, go, ge, scale, chain):
    """The out-of-band bound of the diagonal kernels (c2_outside_band_bound) prices steps down at gap_extend and steps right
    by runs; these references put the incentives where that reasoning has exceptions (last row, row 0, blocks of rows, values
    up to 3) and the reads make the cheap gapped paths optimal.  Whole chain 4 -> 2 -> 1 -> full plane vs the oracle."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(1000 + scale - go)
    refs, gis, incs, reads, rids = adversarial_case(rng, 96)
    gis = [g * scale for g in gis]
    st = {}
    res, rec = E.align_batch(reads, refs, gis, incs, m, go, ge, ref_ids=rids, band_lanes=chain, grid=3, stats=st)
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        exp = oracle.global_align_raw(reads[k], refs[rids[k]], m, gis[rids[k]], go, ge)
        assert exp[0] == 0 and r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], (k, rids[k])
    assert 0 < st["fallback"] < st["tasks"], st


@pytest.mark.parametrize("R", [1, 2, 3])
def test_emulated_kernel_multipass(mats, R):
    """Force fewer rows per lane so that 150..250-row references need 2..4 passes through the LDS boundary row."""
    vecs = load_golden("realistic.json")
    vecs = vecs[:12] + vecs[45:57] + vecs[-12:]
    assert run_vectors(vecs, mats, force_R=R) == len(vecs)
    small = load_golden("fuzz_align.json")[::7]
    assert run_vectors(small, mats, force_R=R) == len(small)


def test_emulated_kernel_reverse_complement_and_ref_ids(mats):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    vecs = load_golden("realistic.json")
    a, b = vecs[0], vecs[50]           # 150-bp and 250-bp amplicons
    reads, rids, strands, exp = [], [], [], []
    for k, v in enumerate(vecs[:10] + vecs[45:55]):
        rid = 0 if v["seqi"] == a["seqi"] else 1
        rc = k % 2
        rd = v["seqj"]
        reads.append("".join(comp[c] for c in reversed(rd)) if rc else rd)
        rids.append(rid); strands.append(rc); exp.append(v)
    res, rec = E.align_batch(reads, [a["seqi"], b["seqi"]], [a["gap_incentive"], b["gap_incentive"]],
                             [a["include"], b["include"]], mats["EDNAFULL"], -20, -2, ref_ids=rids, strands=strands)
    for v, (s1, s2), r, rid, rc in zip(exp, res, rec, rids, strands):
        assert r["status"] == 0 and r["ref_id"] == rid and r["strand"] == rc
        assert [s1, s2] == v["out"][:2]
        check_record(r, v["payload"], s1, s2)


def test_emulated_kernel_all_refs_mode(mats):
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(5)
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in (40, 55, 47)]
    gis = [np.zeros(len(r) + 1, dtype=np.int64) for r in refs]
    for g in gis:
        g[len(g) // 2] = 1
    incs = [[len(r) // 2 - 1, len(r) // 2] for r in refs]
    reads = [refs[k % 3][:20] + "ACG" + refs[k % 3][22:] for k in range(7)]
    res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, all_refs=True)
    assert len(res) == 21
    for t, ((s1, s2), r) in enumerate(zip(res, rec)):
        rd, rf = reads[t // 3], t % 3
        exp = oracle.global_align_raw(rd, refs[rf], m, gis[rf], -20, -2)
        assert (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:]
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[rf]), s1, s2)


def test_emulated_kernel_status_bits(mats):
    m = mats["EDNAFULL"]
    g = np.zeros(5, dtype=np.int64)
    res, rec = E.align_batch(["ACG\u00e9"[:3] + "T", "ACGT", "AC-T"], ["ACGT"], [g], [[1, 2]], m, -20, -2, strands=[0, 0, 1])
    assert rec["status"][0] == 0
    assert rec["status"][1] == 0
    assert rec["status"][2] == 0         # '-' is legal input to reverse_complement
    # a read character beyond the 90 x 90 matrix reads the flat element ci * 90 + cj in the reference (bounds checking off):
    # defined while the largest reference character keeps it inside the buffer ('T': 84 * 90 + 97 < 8100), refused otherwise
    # ('Y' = 89, the last row); a REFERENCE character beyond the matrix is always out of bounds
    res, rec = E.align_batch(["ACGa", "ACGa", "ACGT"], ["ACGT", "ACGY", "ACGa"], [g, g, g], [[1, 2]] * 3, m, -20, -2, ref_ids=[0, 1, 2])
    assert rec["status"][0] == 0 and res[0] == oracle.global_align("ACGa", "ACGT", m, g, -20, -2)[:2]
    assert rec["status"][1] & 2 and rec["status"][2] & 2
    assert oracle.global_align_raw("ACGa", "ACGY", m, g, -20, -2)[0] & oracle.ERR_OOB_CHAR
    assert oracle.global_align_raw("ACGT", "ACGa", m, g, -20, -2)[0] & oracle.ERR_OOB_CHAR
    res, rec = E.align_batch(["ACRT"], ["ACGT"], [g], [[1, 2]], m, -20, -2, strands=[1])
    assert rec["status"][0] & 16         # 'R' -> KeyError in CRISPRessoShared.reverse_complement
    # a path that leaves the reference's defined domain: the oracle flags it, so must the kernel
    st = oracle.global_align_raw("TT", "G", m, np.array([1, 0], dtype=np.int64), -1, -1)[0]
    res, rec = E.align_batch(["TT"], ["G"], [np.array([1, 0], dtype=np.int64)], [[0]], m, -1, -1)
    assert st != 0 and rec["status"][0] & (4 | 8)


def test_emulated_classify_lists(mats):
    from helpers import PAYLOAD_FIELDS
    order = ["ref_positions", "all_insertion_positions", "all_insertion_left_positions", "insertion_positions",
             "insertion_coordinates", "insertion_sizes", "all_deletion_positions", "all_deletion_coordinates",
             "deletion_positions", "deletion_coordinates", "deletion_sizes", "all_substitution_positions",
             "all_substitution_values", "substitution_positions", "substitution_values"]
    vecs = load_golden("fuzz_classify.json") + [k for k in load_golden("ref_unit_kats.json") if k["fn"].startswith("find_indels")]
    n_legacy = 0
    # the batched kernel (one lane per alignment, count pass + write pass) must give the per-call kernel's lists
    batched = {}
    for legacy in (False, True):
        vs = [(k, v) for k, v in enumerate(vecs) if v["fn"].endswith("legacy") == legacy][:150]
        outs = E.classify_lists_batch([(v["read_al"], v["ref_al"]) for _, v in vs], [v["include"] for _, v in vs], legacy=legacy)
        for (k, _), o in zip(vs, outs):
            batched[k] = o
    assert len(batched) >= 200
    for kv, v in enumerate(vecs):
        legacy = v["fn"].endswith("legacy")
        lists, counts = E.classify_lists(v["read_al"], v["ref_al"], v["include"], legacy=legacy)
        if kv in batched:
            assert batched[kv] == (lists, counts), v
        got = dict(zip(order, lists))
        for f in ("insertion_coordinates", "all_deletion_coordinates", "deletion_coordinates"):
            got[f] = [[got[f][k], got[f][k + 1]] for k in range(0, len(got[f]), 2)]
        for f in ("all_substitution_values", "substitution_values"):
            got[f] = [chr(c) for c in got[f]]
        got["insertion_n"], got["deletion_n"], got["substitution_n"] = counts
        for f in PAYLOAD_FIELDS:
            assert got[f] == v["out"][f], (f, v)
        n_legacy += legacy
    assert n_legacy >= 100


def test_emulated_paired_consensus_kernel():
    """c2_consensus_pairs_kernel against the reference's get_consensus_alignment_from_pairs: every call of the reference's own
    unit test and 260 pairs cut from the FANC reads (goldens recorded from the reference, make_golden.py --paired)."""
    g = load_golden("paired.json.gz")
    calls = [c for c in g["unit"] + g["fuzz"] if "raises" not in c]
    assert len(calls) >= 250
    outs = E.consensus_pairs([tuple(c["args"]) for c in calls])
    for c, (aln, qual, ref, hom, caching, err) in zip(calls, outs):
        assert not err
        assert [aln, qual, ref, round(float(100 * hom / float(len(ref))), 3), caching] == c["out"], c
    # a quality string that is too short is an IndexError in the reference: flagged, not read out of bounds
    bad = E.consensus_pairs([("ACGT", "ACGT", 100.0, "II", "ACGT", "ACGT", 100.0, "IIII")])[0]
    assert bad[5]


def test_emulated_chain_on_the_reads_of_the_reference_params_run(mats):
    """The reads of the reference's CRISPResso_on_params test (real FANC amplicon reads, 250 bp against 223 / 2xx bp
    amplicons whose gap incentive is set at three cut points, quantification window of 17 scattered positions) against BOTH
    amplicons through the default launch chain (4 -> 2 -> 1 alignments per wavefront, full plane last): every string, score
    and classifier count equals the oracle's."""
    g = load_golden("params_run.json.gz")
    m = mats["EDNAFULL"]
    refs = [r["sequence"] for r in g["refs"]]
    gis = [np.array(r["gap_incentive"], dtype=np.int64) for r in g["refs"]]
    incs = [r["include_idxs"] for r in g["refs"]]
    lines = g["fastq_after_quality_filter"].split("\n")
    reads = list(dict.fromkeys(lines[k] for k in range(1, len(lines) - 1, 4)))
    st = {}
    res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, all_refs=True, band_lanes=-7, stats=st)
    assert len(res) == 2 * len(reads)
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        rd, ri = reads[k // 2], k % 2
        exp = oracle.global_align_raw(rd, refs[ri], m, gis[ri], -20, -2)
        assert r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], k
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[ri]), s1, s2)
    assert st["fallback"] < st["tasks"]


@pytest.mark.parametrize("chain", [-7, -87])
@pytest.mark.parametrize("go,ge,seed", [(-20, -2, 11), (-10, -3, 12), (-5, -1, 13)])
def test_emulated_chain_soak_mixed_batches(mats, go, ge, seed, chain):
    """The GPU soak's generator (tests/test_gpu_soak.py: three references of different lengths, both strands, long indels,
    truncated / unrelated reads, N and IUPAC symbols, one or two cut sites) through the emulated default chain: EVERY
    alignment and record against the oracle, on the CPU."""
    from test_gpu_soak import COMP, mutate
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(seed)
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in (int(rng.integers(60, 120)), int(rng.integers(180, 260)), int(rng.integers(120, 200)))]
    gis, incs = [], []
    for r in refs:
        g = np.zeros(len(r) + 1, dtype=np.int64)
        g[len(r) // 2 + 1] = 1
        if seed % 2 == 0:
            g[len(r) // 3] = 1
        gis.append(g)
        incs.append(list(range(len(r) // 2 - 3, len(r) // 2 + 3)))
    n = 700
    rids = rng.integers(0, 3, n).astype(np.uint16)
    truth = [mutate(rng, refs[r]) for r in rids]
    strands = np.array([1 if (rng.random() < 0.3 and set(t) <= set("ACGTN")) else 0 for t in truth], dtype=np.uint8)
    reads = [("".join(COMP[c] for c in reversed(t)) if st else t) for t, st in zip(truth, strands)]
    st = {}
    res, rec = E.align_batch(reads, refs, gis, incs, m, go, ge, ref_ids=rids, strands=strands, band_lanes=chain, stats=st)
    n_undefined = 0
    for k in range(n):
        status, s1, s2, mt, ln = oracle.global_align_raw(truth[k], refs[rids[k]], m, gis[rids[k]], go, ge)
        r = rec[k]
        if status != 0:
            assert r["status"] != 0, k
            n_undefined += 1
            continue
        assert r["status"] == 0 and res[k] == (s1, s2) and int(r["matches"]) == mt and int(r["aln_len"]) == ln, k
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[rids[k]]), s1, s2)
    assert n_undefined < n // 10
    if (go, ge) == (-5, -1):
        assert st["fallback"] == st["tasks"]                        # max(go, ge) + incentive = 0: the diagonal kernels do not apply
    else:
        assert 0 < st["fallback"] < st["tasks"]


def test_gap_free_predicate_kept_in_registers_every_length_residue(mats):
    """c2_align_diagx_kernel decides "gap-free" from the two low pointer bits of the main-diagonal cells, collected in the
    lane's registers during the fill and captured at the cell (L, L) (c2_gapfree) -- no pointer word is read back for such
    reads.  Square alignments of every length residue mod 8 (the capture masks the word in the making by the position of the
    last cell inside its group), without gaps (must take the shortcut and give the reference's strings), with a gap or a
    terminal overhang close to either end (must NOT), against the oracle."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(424242)
    for na in (-4, -2):
        reads, refs, gis, incs, rids = [], [], [], [], []
        for L in range(33, 58):
            ref = "".join(rng.choice(list("ACGT"), L))
            g = np.zeros(L + 1, dtype=np.int64)
            g[L // 2 + 1] = 1
            variants = [ref]                                                        # identical
            r = list(ref); r[int(rng.integers(0, L))] = "N"; variants.append("".join(r))
            r = list(ref); r[L - 1] = {"A": "C", "C": "G", "G": "T", "T": "A"}[r[L - 1]]; variants.append("".join(r))   # last cell a mismatch
            r = list(ref); r[0] = {"A": "C", "C": "G", "G": "T", "T": "A"}[r[0]]; variants.append("".join(r))
            variants.append(ref[:L - 2] + ref[L - 1] + "A")                         # one base deleted near the end, padded: same length, gapped
            variants.append("T" + ref[:L - 1])                                      # shifted by one: gaps at both ends
            variants.append(ref[1:] + "G")
            variants.append(ref[:3] + ref[5:] + "CA")                               # deletion near the start
            for v in variants:
                assert len(v) == L
                reads.append(v); rids.append(len(refs))
            refs.append(ref); gis.append(g); incs.append([L // 2, L // 2 + 1])
        res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, ref_ids=rids, band_lanes=na, grid=3)
        n_gapfree = 0
        for k, rd in enumerate(reads):
            st, s1, s2, mt, ln = oracle.global_align_raw(rd, refs[rids[k]], m, gis[rids[k]], -20, -2)
            assert st == 0 and rec["status"][k] == 0 and res[k] == (s1, s2) and int(rec["matches"][k]) == mt and int(rec["aln_len"][k]) == ln, (na, k, rd)
            p = oracle.find_indels_substitutions(s1, s2, incs[rids[k]])
            assert (rec["insertion_n"][k], rec["deletion_n"][k], rec["substitution_n"][k]) == (p["insertion_n"], p["deletion_n"], p["substitution_n"])
            n_gapfree += "-" not in s1 and "-" not in s2
        assert n_gapfree >= 4 * 25


def test_packed_kernel_pairs_singles_and_mismatched_neighbours(mats):
    """c2_align_diagp_kernel: two alignments per lane group in int16 halves.  Lane groups whose two tasks share reference and
    read length run as a pair; a second task of another length or another reference is handed to the next launch; a group
    with one task runs it in both halves; references the int16 range does not admit (here: one too short for the sentinel
    argument, one with an IUPAC symbol) never enter the packed fill.  Every alignment and record against the oracle, through
    the packed kernel alone (-8: what it leaves goes to the full-plane kernel) and through the default chain (-87)."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(8088)
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in (200, 231, 90)]
    refs.append(refs[0][:100] + "R" + refs[0][101:])                       # IUPAC symbol: code >= 8
    gis, incs = [], []
    for r in refs:
        g = np.zeros(len(r) + 1, dtype=np.int64)
        g[len(r) // 2 + 1] = 1
        gis.append(g)
        incs.append([len(r) // 2, len(r) // 2 + 1])
    reads, rids = [], []
    for k in range(333):                                                    # (odd: the last lane group holds one task)
        r = int(rng.choice([0, 0, 0, 1, 1, 2, 3]))
        t = list(refs[r].replace("R", "A"))
        for _ in range(int(rng.integers(0, 4))):
            t[int(rng.integers(0, len(t)))] = str(rng.choice(list("ACGTN")))
        t = "".join(t)
        kind = rng.random()
        if kind < 0.25:
            d = int(rng.integers(1, 14)); p0 = int(rng.integers(5, len(t) - 20)); t = t[:p0] + t[p0 + d:] + "".join(rng.choice(list("ACGT"), d))   # deletion, length kept
        elif kind < 0.4:
            d = int(rng.integers(1, 10)); p0 = int(rng.integers(5, len(t) - 20)); t = (t[:p0] + "".join(rng.choice(list("ACGT"), d)) + t[p0:])[:len(t)]   # insertion, length kept
        elif kind < 0.6:
            t = t[:len(t) - int(rng.integers(1, 9))]                         # shorter read: another length in the neighbouring half
        reads.append(t); rids.append(r)
    for chain in (-8, -84, -87):
        st = {}
        res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, ref_ids=rids, band_lanes=chain, grid=3, stats=st)
        for k, rd in enumerate(reads):
            status, s1, s2, mt, ln = oracle.global_align_raw(rd, refs[rids[k]], m, gis[rids[k]], -20, -2)
            assert status == 0 and rec["status"][k] == 0 and res[k] == (s1, s2) and int(rec["matches"][k]) == mt and int(rec["aln_len"][k]) == ln, (chain, k)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, incs[rids[k]]), s1, s2)
        assert 0 < st["fallback"] < st["tasks"], st
        if chain == -8:
            # neighbours of another reference / length go to the 32-bit kernel of the band.  Round 5: the partition orders a ragged chunk's slots by
            # read length first, so fewer are left over than in task order (C2_NO_LENGTH_ORDER=1) -- same alignments either way
            os.environ["C2_NO_LENGTH_ORDER"] = "1"
            try:
                st_t = {}
                res_t, rec_t = E.align_batch(reads, refs, gis, incs, m, -20, -2, ref_ids=rids, band_lanes=chain, grid=3, stats=st_t)
            finally:
                del os.environ["C2_NO_LENGTH_ORDER"]
            assert res_t == res and rec_t.tobytes() == rec.tobytes()
            assert st_t["unpaired"] > 40 and st["unpaired"] < st_t["unpaired"], (st, st_t)
            in_task_order = st_t["unpaired"]
    # the same reads in pairs of equal (reference, length): hardly anything is left unpaired
    order = sorted(range(len(reads)), key=lambda k: (rids[k], len(reads[k])))
    st2 = {}
    E.align_batch([reads[k] for k in order], refs, gis, incs, m, -20, -2, ref_ids=[rids[k] for k in order], band_lanes=-8, grid=3, stats=st2)
    assert st2["unpaired"] < 40 and st2["unpaired"] < in_task_order // 3, (st2, in_task_order)


# ---- pointer plane in HBM: alignments whose full plane does not fit a CU's LDS --------------------------------------------------
def _edited(rng, ref, n_events):
    s = list(ref)
    for _ in range(n_events):
        p = int(rng.integers(5, len(s) - 5))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            s[p] = "ACGT"[int(rng.integers(0, 4))]
        elif kind == 1:
            del s[p:p + int(rng.integers(1, 12))]
        else:
            s[p:p] = list("".join(rng.choice(list("ACGT"), int(rng.integers(1, 12)))))
    return "".join(s)


@pytest.mark.parametrize("R", [0, 1, 2])
def test_emulated_kernel_pointer_plane_in_hbm_small_cases(mats, R, monkeypatch):
    """The HBM-plane instance of the row-strip kernel (plane row = step of the sweep) forced on sizes that would fit LDS:
    single- and multi-pass references, packed and LDS score tables, identical to the reference on every vector."""
    monkeypatch.setenv("C2_EMU_HBM_PLANE", "1")
    vecs = load_golden("realistic.json")
    vecs = vecs[:10] + vecs[45:55] + vecs[-10:]
    assert run_vectors(vecs, mats, force_R=R) == len(vecs)
    small = load_golden("fuzz_align.json")[::5]
    assert run_vectors(small, mats, force_R=R, no_packed=(R == 2)) == len(small)
    kats = [k for k in load_golden("ref_unit_kats.json") if k["fn"] == "global_align"]
    assert run_vectors(kats, mats, force_R=R, band_lanes=-87 if R == 0 else 0) == len(kats)


@pytest.mark.parametrize("li,lj", [(600, 600), (1000, 300), (300, 1000), (40, 2500)])
def test_emulated_kernel_alignments_larger_than_the_lds_plane(mats, li, lj):
    """600 x 600 needs 237 KB of pointer words, 1000 x 300 196 KB: more than the 160 KB of LDS.  The chain's last launch keeps
    them in HBM scratch instead (r01: C2_E_TOO_LARGE); the diagonal tiers in front of it are unaffected.  Edited copies (the band
    certifies them), unrelated reads and reads with a long deletion (the full-plane launch has to finish them)."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(li * 7 + lj)
    ref = "".join(rng.choice(list("ACGT"), li))
    g = np.zeros(li + 1, dtype=np.int64)
    g[li // 2] = 1
    inc = list(range(li // 2 - 10, li // 2 + 10))
    reads = []
    if lj == li:
        reads += [_edited(rng, ref, 3) for _ in range(3)]
        reads.append(ref[:200] + ref[420:])                                  # a 220-base deletion: outside every band
        reads.append("".join(rng.choice(list("ACGT"), lj)))                  # unrelated
    else:
        lo = int(rng.integers(0, max(1, li - lj))) if li > lj else 0
        reads.append((ref[lo:lo + lj] if li > lj else ref + "".join(rng.choice(list("ACGT"), lj - li)))[:lj])
        reads.append("".join(rng.choice(list("ACGT"), lj)))
        reads.append(_edited(rng, (ref * (lj // li + 1))[:lj], 4))
    for chain in (0, -87):
        st = {}
        res, rec = E.align_batch(reads, [ref], [g], [inc], m, -20, -2, band_lanes=chain, grid=3, stats=st)
        for k, ((s1, s2), r) in enumerate(zip(res, rec)):
            exp = oracle.global_align_raw(reads[k], ref, m, g, -20, -2)
            assert exp[0] == 0 and r["status"] == 0, (k, r)
            assert (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], (chain, k)
            check_record(r, oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
        if chain and lj == li:
            # the long deletion and the unrelated read are beyond the first tier: handed on by it, or (round 5) sent past it by the partition --
            # the unrelated read straight to the full-matrix launch (class 6) -- and (round 6) the 220-base deletion with it: no band launch holds it
            assert 0 < st["fallback"] + sum(st["classes"][3:]) < st["tasks"] and st["classes"][6] == 2, st


# ---- the two variants of the packed fill: sums as v_pk_add_i16, or as plain 32-bit adds under a per-anti-diagonal bias ----------------
@pytest.mark.parametrize("mode", [-8, -84, -87, -82])
def test_packed_fill_with_packed_adds_is_still_exact(mats, mode, monkeypatch):
    """The default for the usual amplicons is now the 32-bit-add variant (c2_pk_add32_ok); references beyond its range keep the
    packed adds (C2_EMU_NO_ADD32 forces them here): the same vectors through that variant."""
    monkeypatch.setenv("C2_EMU_NO_ADD32", "1")
    st = {}
    vecs = load_golden("realistic.json")
    assert run_vectors(vecs, mats, band_lanes=mode, stats=st) == len(vecs)
    assert st["pk_beta"] == 0 and st["tasks"] - st["fallback"] > len(vecs) // 4, st
    kats = [k for k in load_golden("ref_unit_kats.json") if k["fn"] == "global_align"]
    assert run_vectors(kats, mats, band_lanes=mode, stats=st) == len(kats)


def _extreme_equal_length_reads(rng, ref, n):
    from test_gpu_soak import _equal_length_reads
    return _equal_length_reads(rng, ref, n)


@pytest.mark.parametrize("go,ge,gval", [(-20, -2, 1), (-8, -3, 2), (-30, -30, 5)])
def test_packed_fill_32bit_adds_at_the_limit_of_their_range(mats, go, ge, gval):
    """c2_pk_add32_ok admits a reference while bias + hi + beta * (anti-diagonals) <= 32000.  At the longest admitted length -- reads of
    that length, so that they pair: the reference itself (the largest values), every base mismatched (the smallest), all N, long
    indels -- every alignment against the oracle; one base longer must fall back to the packed adds (same results)."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(4242 - go)

    def probe(L):
        ref = "ACGT" * (L // 4) + "ACGT"[:L % 4]
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = gval
        st = {}
        E.align_batch([ref, ref], [ref], [g], [[L // 2]], m, go, ge, band_lanes=-8, stats=st)
        return st["pk_beta"]
    lo, hi = 200, 900                                                   # (below ~150 bp the reference's finite sentinel is in reach: no packed fill at all)
    assert probe(lo) > 0
    while lo < hi:
        mid = (lo + hi + 1) // 2
        lo, hi = (mid, hi) if probe(mid) > 0 else (lo, mid - 1)
    limit = lo
    assert 150 < limit < 900, limit
    for L, want in ((limit, True), (limit + 1, False)):
        ref = "".join(rng.choice(list("ACGT"), L))
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = gval; g[0] = gval; g[L] = gval if gval < -ge else 0
        inc = list(range(L // 2 - 3, L // 2 + 3))
        reads = _extreme_equal_length_reads(rng, ref, 40)
        st = {}
        res, rec = E.align_batch(reads, [ref], [g], [inc], m, go, ge, band_lanes=-87, stats=st)
        assert (st["pk_beta"] > 0) == want, (L, st)
        if want:
            assert st["pk_beta"] >= -go and st["pk_bias"] < 16384
        for k, rd in enumerate(reads):
            status, s1, s2, mt, ln = oracle.global_align_raw(rd, ref, m, g, go, ge)
            assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (L, k)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)


def _reads_for_the_partition(rng, amp, n):
    """fixed-length reads like the benchmark's (a deletion at the cut pulls the rest of the amplicon forward and random bases fill the end;
    an insertion pushes it out) and variable-length ones (the read simply ends where the amplicon does), a few substitutions in each"""
    L, cut = len(amp), len(amp) // 2
    reads, kinds = [], []
    for k in range(n):
        t = list(amp)
        for _ in range(int(rng.integers(0, 3))):
            t[int(rng.integers(0, L))] = str(rng.choice(list("ACGT")))
        t = "".join(t)
        kind = k % 8
        d = [0, 2, 5, 9, 20, 40, 3, 25][kind]
        if kind < 6:                                   # deletion of d bases, fixed length
            t = (t[:cut - d // 2] + t[cut - d // 2 + d:] + "".join(rng.choice(list("ACGT"), d)))[:L]
        elif kind == 6:                                # insertion of d bases, fixed length
            t = (t[:cut] + "".join(rng.choice(list("ACGT"), d)) + t[cut:])[:L]
        else:                                          # deletion, the read is shorter
            t = t[:cut - 10] + t[cut - 10 + d:]
        reads.append(t); kinds.append(kind)
    return reads, kinds


@pytest.mark.parametrize("L", [250, 150])
def test_partition_routes_tasks_to_the_launch_whose_band_holds_their_path(mats, L, monkeypatch):
    """c2_align_partition_kernel's classes and the launches behind them (the host library's wiring, mirrored by the emulator harness): reads without
    an indel go through the score-only launch, long indels straight to the second / third tier, and -- with C2_P16_TIER=1, the library's opt-in --
    short indels through the 14-diagonal launch (sixteen alignments per wavefront, pointer words kept).  Whatever the class, every alignment is
    the oracle's; the same reads with the routing switched off give the same results."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(4242 + L)
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = [L // 2, L // 2 + 1]
    reads, kinds = _reads_for_the_partition(rng, amp, 96)
    want = []
    for k, rd in enumerate(reads):
        status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, -20, -2)
        assert status == 0
        want.append((s1, s2, mt))
    for p16 in (False, True):
        if p16:
            monkeypatch.setenv("C2_P16_TIER", "1")
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
        for k, (s1, s2, mt) in enumerate(want):
            assert rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (L, k, kinds[k], p16)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
        cls = st["classes"]
        assert sum(cls) == len(reads), cls
        # every kind of launch saw tasks (150 bp: the 40-base deletion leaves the probe's window in the filler -- nothing found, first tier) ...
        # (classes: 0 score-only, 1 the 14-diagonal launch, 2 / 3 / 4 / 5 the band tiers of 32 / 40 / 62 / 128 diagonals, 6 the full-matrix launch)
        assert cls[0] >= 10 and cls[3] + cls[4] >= 10 and (cls[5] >= 8 or L == 150), cls
        if p16:
            # ... and the 14-diagonal launch finished most of its own (round 6: a band is chosen only if its launch can be expected to CERTIFY the alignment,
            # c2_part_probe -- the 2-base deletions qualify, the 5-base ones and the 3-base insertions with their trailing run do not)
            assert cls[1] >= 10 and st["p16_finished"] >= cls[1] * 3 // 4, st
        else:
            assert cls[1] == 0 and cls[2] >= 20, cls
    monkeypatch.setenv("C2_NO_ROUTE", "1")
    st2 = {}
    res2, rec2 = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st2)
    assert res2 == res and np.array_equal(rec2, rec)
    assert st2["classes"][3] == 0 and st2["classes"][4] == 0 and st2["classes"][5] == 0 and st2["classes"][1] > 0, st2


@pytest.mark.parametrize("L", [250, 223, 150])
def test_partition_probe_walks_words_as_it_walked_bytes(mats, L, monkeypatch):
    """c2_part_window over c2_dev_ref.seq2 (16 two-bit codes per word, a word ahead in a register) finds the windows the byte walk finds: the same
    classes for the same reads -- shifts to either side, up to the walk's reach, reads shorter and longer than the reference, windows at both ends."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(9100 + L)
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = [L // 2, L // 2 + 1]
    reads = []
    for k in range(160):
        t = list(amp)
        for _ in range(int(rng.integers(0, 3))):
            t[int(rng.integers(0, L))] = str(rng.choice(list("ACGTN")))
        t = "".join(t)
        cut = int(rng.integers(20, L - 20))
        d = int(rng.integers(1, 70))
        kind = k % 5
        if kind == 0:
            t = (t[:cut] + t[cut + d:] + "".join(rng.choice(list("ACGT"), d)))[:L]            # deletion, fixed length
        elif kind == 1:
            t = (t[:cut] + "".join(rng.choice(list("ACGT"), d)) + t[cut:])[:L]                # insertion, fixed length
        elif kind == 2:
            t = t[:cut] + t[cut + d:]                                                          # deletion, shorter read
        elif kind == 3:
            t = "".join(rng.choice(list("ACGT"), d % 9)) + t + "".join(rng.choice(list("ACGT"), d % 31))   # overhangs
        else:
            t = t[d % 40:]                                                                     # the read starts inside the reference
        if len(t) >= 40:
            reads.append(t)
    st, st2 = {}, {}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    monkeypatch.setenv("C2_EMU_NO_SEQ2", "1")
    res2, rec2 = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st2)
    assert st["classes"] == st2["classes"] and sum(st["classes"][3:]) >= 20, (st["classes"], st2["classes"])
    assert res2 == res and np.array_equal(rec2, rec)
    for k in range(0, len(reads), 7):
        status, s1, s2, mt, ln = oracle.global_align_raw(reads[k], amp, m, g, -20, -2)
        assert status == 0 and res[k] == (s1, s2), (L, k)


@pytest.mark.parametrize("L", [300, 700])
def test_traceback_runs_of_m_longer_than_one_pass_of_the_word_probe(mats, L):
    """c2_traceback reads a run of state M off whole pointer words, four cells per lane: 250-odd cells per pass.  References of 300 and 700 bases
    with reads that differ from them in a substitution or two, one short indel, an indel at either end, or nothing at all: runs of M of up to 700
    cells (several passes, the last one partial), runs that end in the first word of a pass, at a word's last cell, on the matrix edge."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(9000 + L)
    ref = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = list(range(L // 2 - 5, L // 2 + 5))
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    reads = [ref, ref[1:], ref[:-1], ref[2:] + "AC", "GT" + ref[:-2]]
    for cut in (1, 2, 3, 4, 7, 8, 9, 250, 251, 252, 253, 254, 255, 256, 257, L - 9, L - 8, L - 4, L - 3, L - 2, L - 1):
        reads.append(ref[:cut] + ref[cut + 1:])                                     # one base deleted at every kind of place in a word / a pass
        reads.append(ref[:cut] + "T" + ref[cut:])                                   # ... inserted
        reads.append(ref[:cut] + other[ref[cut]] + ref[cut + 1:])                   # ... substituted
    for _ in range(12):
        t = list(ref)
        for _ in range(int(rng.integers(0, 4))):
            q = int(rng.integers(0, L)); t[q] = other[t[q]]
        t = "".join(t)
        a = int(rng.integers(5, L - 40)); d = int(rng.integers(1, 12))
        reads.append(t[:a] + t[a + d:] if rng.random() < 0.5 else t[:a] + "".join(rng.choice(list("ACGT"), d)) + t[a:])
    reads.sort(key=len)                                                             # (a packed kernel pairs neighbours of one length: the words read here are its)
    st = {}
    res, rec = E.align_batch(reads, [ref], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    assert st["unpaired"] <= 12, st["unpaired"]
    for k, rd in enumerate(reads):
        status, s1, s2, mt, ln = oracle.global_align_raw(rd, ref, m, g, -20, -2)
        assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (L, k, len(rd))
        check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)


@pytest.mark.parametrize("go,ge", [(-20, -2), (-5, -3)])
@pytest.mark.parametrize("env", [{}, {"C2_ROUTE_MARGIN": "0"}, {"C2_ROUTE_MARGIN": "12"}, {"C2_ROUTE_PROBE_MISMATCH": "32"}, {"C2_SCORE_TIER_MAX_MISMATCH": "32"},
                                 {"C2_P16_TIER": "1", "C2_ROUTE_MARGIN": "0"}])
def test_partition_settings_never_change_a_result(mats, go, ge, env, monkeypatch):
    """The partition only says which launch sees a task FIRST; every launch verifies what it finishes.  So the most careless settings -- no margin
    (tasks routed to bands their certificate will refuse), a probe that accepts any window, a tail check that sends every equal-length read to the
    score-only launch -- must give the oracle's alignments all the same: three references (ref_ids), both strands, reads from 40 to 260 bases (the
    probe needs 96), Ns, indels of up to 45 bases, two gap scorings."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(515)
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in (250, 200, 120)]
    gis = [np.zeros(len(r) + 1, dtype=np.int64) for r in refs]
    for g in gis:
        g[len(g) // 2 + 1] = 1
    incs = [list(range(len(r) // 2 - 5, len(r) // 2 + 5)) for r in refs]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    reads, rids, strands, truth = [], [], [], []
    for k in range(72):
        r = k % 3
        s = list(refs[r])
        L = len(s)
        kind = (k // 3) % 6
        if kind == 1:
            d = int(rng.integers(1, 46)); a = L // 2 - d // 2
            s = (s[:a] + s[a + d:] + list(rng.choice(list("ACGT"), d)))[:L]            # deletion, length kept
        elif kind == 2:
            d = int(rng.integers(1, 16)); s = (s[:L // 2] + list(rng.choice(list("ACGT"), d)) + s[L // 2:])[:L]
        elif kind == 3:
            d = int(rng.integers(1, 46)); a = int(rng.integers(5, L - 50)); s = s[:a] + s[a + d:]   # deletion, the read is shorter
        elif kind == 4:
            s = s[int(rng.integers(0, 30)):int(rng.integers(L - 30, L))]                # a fragment
        elif kind == 5:
            s = s[:int(rng.integers(40, 96))]                                           # too short for the probe
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, len(s)))] = str(rng.choice(list("ACGTN")))
        fw = "".join(s)
        rc = (k // 18) % 2
        reads.append("".join(comp[c] for c in reversed(fw)) if rc else fw)
        rids.append(r); strands.append(rc); truth.append(fw)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    st = {}
    res, rec = E.align_batch(reads, refs, gis, incs, m, go, ge, ref_ids=rids, strands=strands, band_lanes=-87, stats=st)
    assert sum(st["classes"]) == len(reads)
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        exp = oracle.global_align_raw(truth[k], refs[rids[k]], m, gis[rids[k]], go, ge)
        assert r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], (k, env)
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[rids[k]]), s1, s2)


@pytest.mark.parametrize("layout", ["tagged_sorted", "tagged_interleaved", "all_refs"])
def test_main_diagonal_reads_of_several_references_are_finished_by_the_partition(mats, layout):
    """Round 6: with several references the partition keeps the reference's side of the main-diagonal shortcut per lane group and makes it again when a group
    meets another reference (c2_align_partition_kernel: fetchM / ensure).  Four amplicons of different lengths (one of them a length no block edge likes),
    reads = copies and one / two / three substitutions, tagged with their amplicon in runs, interleaved, or every read against every amplicon: the
    partition finishes reads of every amplicon, and every alignment and record is the oracle's."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(606)
    lens = [250, 203, 160, 97]
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in lens]
    gis, incs = [], []
    for r in refs:
        g = np.zeros(len(r) + 1, dtype=np.int64); g[len(r) // 2 + 1] = 1
        gis.append(g); incs.append([len(r) // 2, len(r) // 2 + 1])
    reads, rids = [], []
    for k in range(400):
        r = (k // 100) if layout == "tagged_sorted" else int(rng.integers(0, 4))
        t = list(refs[r])
        for _ in range(k % 4):
            t[int(rng.integers(0, len(t)))] = str(rng.choice(list("ACGTN")))
        reads.append("".join(t)); rids.append(r)
    st = {}
    if layout == "all_refs":
        reads = reads[:150]
        res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, all_refs=True, band_lanes=-87, stats=st)
        pairs = [(reads[k // 4], k % 4) for k in range(4 * len(reads))]
    else:
        res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, ref_ids=np.array(rids, dtype=np.uint16), band_lanes=-87, stats=st)
        pairs = list(zip(reads, rids))
    finished_of = [0, 0, 0, 0]
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        rd, ri = pairs[k]
        exp = oracle.global_align_raw(rd, refs[ri], m, gis[ri], -20, -2)
        assert r["status"] == 0 and int(r["ref_id"]) == ri and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], (layout, k)
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[ri]), s1, s2)
    assert st["exact_copies"] >= (100 if layout != "all_refs" else 60), (layout, st["exact_copies"], st["classes"])


@pytest.mark.parametrize("seed", [0, 1])
def test_main_diagonal_shortcut_at_every_block_edge_length_one_and_three_references(mats, seed):
    """References of 100 .. 256 bases -- among them every length at a 32-byte block's edge (128 / 129 / 159 / 160 / 161 / ... / 255 / 256) -- alone and three
    to a batch (reads tagged), reads with 0 .. 3 changed bases (N among them) and a few 3-base deletions: the partition finishes most of them from the
    compared registers, and every alignment is the oracle's."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(9900 + seed)
    edge = [128, 129, 159, 160, 161, 191, 192, 193, 224, 225, 255, 256]
    finished = 0
    for trial in range(12):
        nrefs = 1 if trial % 2 == 0 else 3
        lens = [int(rng.integers(100, 257)) for _ in range(nrefs)]
        lens[0] = edge[(6 * seed + trial // 2) % len(edge)]
        refs = ["".join(rng.choice(list("ACGT"), L)) for L in lens]
        gis, incs = [], []
        for r in refs:
            g = np.zeros(len(r) + 1, dtype=np.int64); g[len(r) // 2 + 1] = 1
            gis.append(g); incs.append([len(r) // 2 - 1, len(r) // 2])
        reads, rids = [], []
        for k in range(96):
            r = int(rng.integers(0, nrefs)); t = list(refs[r])
            for _ in range(k % 4):
                t[int(rng.integers(0, len(t)))] = str(rng.choice(list("ACGTN")))
            if k % 11 == 0:
                t = t[:len(t) // 2] + t[len(t) // 2 + 3:]
            reads.append("".join(t)); rids.append(r)
        st = {}
        kw = dict(ref_ids=np.array(rids, dtype=np.uint16)) if nrefs > 1 else {}
        res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, band_lanes=-87, stats=st, **kw)
        for k in range(len(reads)):
            exp = oracle.global_align_raw(reads[k], refs[rids[k]], m, gis[rids[k]], -20, -2)
            assert rec[k]["status"] == 0 and (res[k][0], res[k][1], int(rec[k]["matches"]), int(rec[k]["aln_len"])) == exp[1:], (seed, trial, k, lens)
        finished += st.get("exact_copies", 0)
    assert finished >= 12 * 96 // 2, finished


def test_tasks_of_a_wavefront_are_paired_by_key_before_staging(mats, monkeypatch):
    """Round 6: the lists behind the first band launch are in the order of their atomics, so neighbours there seldom share reference AND read length -- the
    condition for two alignments to share a lane group.  c2_align_diagp_kernel puts the tasks a wavefront holds in the order of their keys first (whole
    pairs in front).  Three amplicons, reads with deletions long enough for the wider tiers and of three lengths: fewer tasks are left unpaired than with
    C2_NO_PAIR_SORT=1, the routing is the same, and every alignment is the oracle's either way."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(3)
    L = 160
    refs = ["".join(rng.choice(list("ACGT"), L)) for _ in range(3)]
    gis, incs = [], []
    for r in refs:
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
        gis.append(g); incs.append([L // 2, L // 2 + 1])
    reads, rids = [], []
    for k in range(600):
        r = int(rng.integers(0, 3)); t = refs[r]
        d = int(rng.integers(12, 40)); c = L // 2
        t = (t[:c] + t[c + d:] + "".join(rng.choice(list("ACGT"), d)))[:L - int(rng.integers(0, 3))]
        reads.append(t); rids.append(r)
    rids = np.array(rids, dtype=np.uint16)
    out = {}
    for knob in ("", "1"):
        if knob:
            monkeypatch.setenv("C2_NO_PAIR_SORT", knob)
        st = {}
        res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, ref_ids=rids, band_lanes=-87, stats=st)
        out[knob] = (res, rec, st)
    (res, rec, st), (res0, rec0, st0) = out[""], out["1"]
    assert res == res0 and np.array_equal(rec, rec0) and st["classes"] == st0["classes"]
    assert st["unpaired"] * 2 <= st0["unpaired"] and st0["unpaired"] >= 30, (st["unpaired"], st0["unpaired"])
    for k in range(0, len(reads), 3):
        exp = oracle.global_align_raw(reads[k], refs[rids[k]], m, gis[rids[k]], -20, -2)
        assert rec[k]["status"] == 0 and (res[k][0], res[k][1], int(rec[k]["matches"]), int(rec[k]["aln_len"])) == exp[1:], k


def test_all_references_batch_goes_through_the_partition_and_pairs_by_reference(mats):
    """Round 5 (BASELINE config 4): an all-references batch of several references -- task = read * n_refs + reference --
    gets the partition and the score-only stage too.  c2_align_partition_kernel walks a chunk reference-major, so the neighbours in every list
    are two reads against the SAME reference and the packed kernels pair them (in task order nothing would pair).  Three candidate amplicons of
    config 4's kind (wild type, a 3-base insertion + substitutions, a 12-base replacement), reads derived from each: every alignment and record
    equals the oracle's; the main-diagonal candidates took the score-only launch; almost nothing was left unpaired; and the chain without the
    partition for such batches (C2_NO_ALLREFS_PARTITION=1: round 4's pair-order walk) gives the same bytes."""
    from crispresso2_amd import synth
    m = mats["EDNAFULL"]
    L = 160
    amp, g, inc = synth.amplicon_setup(L)
    refs = [amp, synth.make_variant(amp, "hdr"), synth.make_variant(amp, "pe")]
    gis = []
    for r in refs:
        x = np.zeros(len(r) + 1, dtype=np.int64)
        x[L // 2 + 1] = 1
        gis.append(x)
    incs = [inc] * 3
    reads = []
    for src in range(3):
        blk = synth.make_reads(L, 70, amplicon_id=100 + src, amplicon=refs[src][:L])
        reads += [b.tobytes().decode() for b in blk]
    rng = np.random.default_rng(5)
    reads = [reads[int(i)] for i in rng.permutation(len(reads))]
    st = {}
    res, rec = E.align_batch(reads, refs, gis, incs, m, -20, -2, all_refs=True, band_lanes=-87, stats=st)
    assert len(res) == 3 * len(reads)
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        rd, ri = reads[k // 3], k % 3
        exp = oracle.global_align_raw(rd, refs[ri], m, gis[ri], -20, -2)
        assert r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], k
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[ri]), s1, s2)
    assert sum(st["classes"]) == 3 * len(reads), st["classes"]       # the partition ran over every task ...
    assert st["classes"][0] >= 60, st["classes"]                      # ... reads against the amplicon they derive from are main-diagonal candidates (not those against the 12-base replacement: the look at the cut site)
    assert st["unpaired"] <= 24, st["unpaired"]                       # a pair breaks only where a list passes from one reference to the next
    os.environ["C2_NO_ALLREFS_PARTITION"] = "1"
    try:
        st2 = {}
        res2, rec2 = E.align_batch(reads, refs, gis, incs, m, -20, -2, all_refs=True, band_lanes=-87, stats=st2)
    finally:
        del os.environ["C2_NO_ALLREFS_PARTITION"]
    assert sum(st2["classes"]) == 0 and res2 == res and rec2.tobytes() == rec.tobytes()


def test_ragged_and_unrelated_reads_through_the_partition(mats):
    """Round 5 (inputs that are not the generator's best case).  Reads cut to lengths U[120, 160], a tenth of them replaced by
    random sequences, against a 160-bp amplicon through the default chain: the partition orders every chunk's slots by read length (else no two
    neighbours could share a lane group of the packed kernels) and sends the reads that match the amplicon nowhere straight to the full-matrix
    launch (class 6).  Every alignment and record equals the oracle's, and the two knobs (C2_NO_LENGTH_ORDER, C2_NO_DIRECT_FULL) change no byte --
    only how many tasks the int16 kernels could pair and how many launches the unrelated reads passed through."""
    from crispresso2_amd import synth
    m = mats["EDNAFULL"]
    L = 160
    amp, g, inc = synth.amplicon_setup(L)
    rng = np.random.default_rng(77)
    base = synth.make_reads(L, 260)
    reads = []
    for k in range(len(base)):
        s = base[k].tobytes().decode()[:int(rng.integers(120, L + 1))]
        if k % 10 == 3:
            s = "".join(rng.choice(list("ACGT"), len(s)))
        reads.append(s)
    st = {}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    for k, ((s1, s2), r) in enumerate(zip(res, rec)):
        exp = oracle.global_align_raw(reads[k], amp, m, g, -20, -2)
        assert exp[0] == 0 and r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == exp[1:], k
        check_record(r, oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
    assert sum(st["classes"]) == len(reads) and 20 <= st["classes"][6] <= 26, st["classes"]      # the 26 random reads (a few find a window by chance)
    variants = {}
    for knob in ("C2_NO_LENGTH_ORDER", "C2_NO_DIRECT_FULL"):
        os.environ[knob] = "1"
        try:
            sv = {}
            rv, recv = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=sv)
        finally:
            del os.environ[knob]
        assert rv == res and recv.tobytes() == rec.tobytes(), knob
        variants[knob] = sv
    assert variants["C2_NO_LENGTH_ORDER"]["unpaired"] > 3 * max(st["unpaired"], 1), (st["unpaired"], variants["C2_NO_LENGTH_ORDER"]["unpaired"])
    assert variants["C2_NO_DIRECT_FULL"]["classes"][6] == 0


def test_forty_diagonal_tier_six_alignments_per_wavefront(mats):
    """Round 5: c2_align_diagp_kernel<6> -- three lane groups of 21 lanes (20 live: 40 diagonals), two alignments per group, the pointer plane's groups
    padded to 24 words.  Reads shaped like the reference's own test data (tests/FANC.Cas9.fastq resampled by crispresso2_amd.synth: a 4-base overhang in
    front of the 223-bp amplicon, 23+ bases of flank behind it -- 28 diagonals the 32-diagonal tier cannot certify --, lengths 248-250, deletions at
    the cut, some unrelated reads): through the kernel alone (-86: what it cannot finish goes to the full plane) and through the default chain (-87: the
    partition sends most of them to this tier first).  Every alignment and record equals the oracle's; with the tier left out (C2_NO_TIER40) the bytes
    are the same."""
    from crispresso2_amd import synth
    m = mats["EDNAFULL"]
    amp, g, inc = synth.fanc_setup()
    R, lens = synth.make_fanc_reads(420, first_block=3)
    reads = [R[k, :lens[k]].tobytes().decode() for k in range(len(lens))]
    want = [oracle.global_align_raw(rd, amp, m, g, -20, -2) for rd in reads]
    outs = {}
    for chain in (-86, -87):
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=chain, grid=3, stats=st)
        for k, ((s1, s2), r) in enumerate(zip(res, rec)):
            assert want[k][0] == 0 and r["status"] == 0 and (s1, s2, int(r["matches"]), int(r["aln_len"])) == want[k][1:], (chain, k)
            check_record(r, oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
        outs[chain] = (res, rec, st)
    assert outs[-86][2]["fallback"] < 0.5 * len(reads)                # the 40-diagonal kernel alone finishes most of these reads
    cls = outs[-87][2]["classes"]
    assert sum(cls) == len(reads) and cls[3] > 0.6 * len(reads) and cls[0] == 0, cls      # class 3 = this tier; no read is as long as the amplicon
    os.environ["C2_NO_TIER40"] = "1"
    try:
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, grid=3, stats=st)
    finally:
        del os.environ["C2_NO_TIER40"]
    assert st["classes"][3] == 0 and res == outs[-87][0] and rec.tobytes() == outs[-87][1].tobytes()


@pytest.mark.parametrize("L", [250, 151, 203, 256])
def test_main_diagonal_reads_are_finished_by_the_partition(mats, L, monkeypatch):
    """A class-0 read on its reference's main diagonal -- a byte-for-byte copy, or one / two differing bases of A C G T N -- needs no fill where the
    scoring proves the diagonal unbeatable (c2_main_diagonal_certificate).  The partition compares such candidates 16 bytes at a time, counts the
    equal bytes of the diagonals +-1 / +-2, and writes the rows and the record itself; three differing bases, an IUPAC code or a lower-case base in
    the read, an indel, another length go through the launches as before.  All results are the oracle's, and identical with the shortcut switched
    off or restricted to copies (the first byte, the last byte and the partial last dword are among the places that differ)."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(7100 + L)
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = [L // 2 - 1, L // 2, L // 2 + 1]
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    sub = lambda s_, q, c=None: s_[:q] + (c or other[s_[q]]) + s_[q + 1:]
    reads = [amp] * 9
    one = []
    for q in (0, 1, 3, 4, 15, 16, 17, L // 2 - 1, L // 2, L // 2 + 1, L - 17, L - 16, L - 5, L - 4, L - 3, L - 2, L - 1):
        one += [sub(amp, q), sub(amp, q, "N")]
    two = [sub(sub(amp, 0), L - 1), sub(sub(amp, 7), 8), sub(sub(amp, L // 2), L // 2 + 1, "N"), sub(sub(amp, L - 2), L - 1), sub(sub(amp, 30, "N"), 90, "N"),
           sub(sub(amp, L - 16), L - 15), sub(sub(amp, 15), 16)]
    more = [sub(sub(sub(amp, 5), 50), 100), sub(amp, 40, "R"), sub(amp, 41, amp[41].lower()), sub(sub(amp, 9), 60, "Y"),
            amp[:L // 2] + amp[L // 2 + 3:] + "ACG", amp[:L // 2] + "TT" + amp[L // 2:-2], amp[:-1], amp + "A", amp[1:]]
    reads += one + two + more + [amp] * 3
    want = []
    for rd in reads:
        status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, -20, -2)
        want.append((status, s1, s2, mt))
    outs = {}
    for name, env in (("on", {}), ("copies", {"C2_DIAG_CERT_KMAX": "0"}), ("off", {"C2_NO_EXACT_COPIES": "1"})):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
        for k_ in env:
            monkeypatch.delenv(k_)
        for k, (status, s1, s2, mt) in enumerate(want):
            if status != 0:
                assert rec[k]["status"] != 0, (name, k)
                continue
            assert rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (name, L, k)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
        assert sum(st["classes"]) == len(reads), st
        outs[name] = (res, rec, st)
    assert outs["on"][2]["exact_copies"] == 12 + len(one) + len(two), outs["on"][2]
    assert outs["copies"][2]["exact_copies"] == 12 and outs["off"][2]["exact_copies"] == 0
    assert outs["on"][2]["classes"] == outs["off"][2]["classes"]
    for name in ("copies", "off"):
        assert outs[name][0] == outs["on"][0] and outs[name][1].tobytes() == outs["on"][1].tobytes(), name
        o1, o2 = outs["on"][2]["raw"]
        p1, p2 = outs[name][2]["raw"]
        for k in range(len(reads)):                                # the rows as the launches leave them, padding of the last dword included
            n4 = (int(outs["on"][1][k]["aln_len"]) + 3) // 4 * 4
            assert o1[k, :n4].tobytes() == p1[k, :n4].tobytes() and o2[k, :n4].tobytes() == p2[k, :n4].tobytes(), (name, k)


def test_main_diagonal_shortcut_only_where_the_scoring_proves_it(mats, monkeypatch):
    """the certificate is per reference: two N's in the reference (EDNAFULL scores N -1 against itself: the diagonal falls 12 short of 5 L, the bound
    is 7 short) or a matrix whose diagonal is not uniform leave the shortcut off; one N still passes (5 L - 6 against 5 L - 7; with two differing
    bases 5 L - 24 against the 5 L - 25 of a path with a gap opened inside the matrix)"""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(7300)
    L = 160
    amp = "".join(rng.choice(list("ACGT"), L))
    with_n = amp[:30] + "N" + amp[31:]
    with_nn = with_n[:90] + "N" + with_n[91:]
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = [L // 2, L // 2 + 1]
    for ref, expect in ((amp, 7), (with_n, 7), (with_nn, 0)):
        reads = [ref] * 6 + [ref[:50] + "A" + ref[50:-1], ref[:10] + ("C" if ref[10] != "C" else "G") + ref[11:]]
        st = {}
        res, rec = E.align_batch(reads, [ref], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
        for k, rd in enumerate(reads):
            status, s1, s2, mt, ln = oracle.global_align_raw(rd, ref, m, g, -20, -2)
            assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (ref == amp, k)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
        assert st["exact_copies"] == expect, st
    uneven = m.copy()
    uneven[ord("C"), ord("C")] = 3                                   # C pairs with C for less than the other bases pair with themselves
    reads = [amp] * 5 + [amp[:40] + amp[41:] + "T"]
    st = {}
    res, rec = E.align_batch(reads, [amp], [g], [inc], uneven, -20, -2, band_lanes=-87, stats=st)
    for k, rd in enumerate(reads):
        status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, uneven, g, -20, -2)
        assert status == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, k
    assert st["exact_copies"] == 0, st                               # (some forty C's at 3 instead of 5: the diagonal is 80 below 5 L, the bound 7 below)


@pytest.mark.parametrize("kind", ["homopolymer", "marker_in_homopolymer", "dinucleotide", "period3", "period5", "half_repeat", "gentle_scoring"])
def test_main_diagonal_shortcut_refuses_what_shifts_onto_itself(mats, kind):
    """References that shift onto themselves -- a homopolymer, tandem repeats of period 2 / 3 / 5, a unique half followed by a repeat -- are where a
    path along ANOTHER diagonal (a leading and a trailing gap run, no opening inside the matrix) comes close to the main diagonal or beats it: reads
    with one or two differing bases there must keep their fill.  Whatever the partition decides, every alignment is the oracle's (the oracle decides
    ties and near-ties by the reference's own comparisons).  gentle_scoring: a gap_open of -6, where two mismatches already cost more than a gap."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(hash(kind) % 10000)
    L = 200
    go, ge = -20, -2
    if kind == "homopolymer":
        amp = "A" * L
    elif kind == "marker_in_homopolymer":
        amp = "A" * 100 + "C" + "A" * (L - 101)
    elif kind == "dinucleotide":
        amp = "AC" * (L // 2)
    elif kind == "period3":
        amp = ("ACG" * L)[:L]
    elif kind == "period5":
        amp = ("ACGTT" * L)[:L]
    elif kind == "half_repeat":
        amp = "".join(rng.choice(list("ACGT"), L // 2)) + ("GA" * L)[:L - L // 2]
    else:
        amp = "".join(rng.choice(list("ACGT"), L))
        go, ge = -6, -2
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = [L // 2, L // 2 + 1]
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    sub = lambda s_, q, c=None: s_[:q] + (c or other[s_[q]]) + s_[q + 1:]
    reads = [amp] * 4
    for q in (0, 1, 2, 50, 99, 100, 101, 150, L - 3, L - 2, L - 1):
        reads += [sub(amp, q), sub(amp, q, "N"), sub(amp, q, amp[(q + 1) % L]), sub(amp, q, amp[q - 1])]
    for q, r in ((0, 1), (0, L - 1), (1, 2), (98, 102), (L - 2, L - 1), (3, 7), (60, 61)):
        reads += [sub(sub(amp, q), r), sub(sub(amp, q, amp[(q + 1) % L]), r, amp[r - 1])]
    reads += [amp[1:] + amp[0], amp[-1] + amp[:-1], amp[2:] + amp[:2]]                  # the reference rotated: ANOTHER diagonal is the perfect one
    if kind == "marker_in_homopolymer":
        # the marker moved by one or two: two differing bases on the main diagonal, none on the diagonal next to it -- the shifted path WINS
        # (5 (L - 1) - 4 against 5 (L - 2) - 8), and the partition must see that its equal bytes are over the limit
        for d in (-2, -1, 1, 2):
            t = list(amp); t[100] = "A"; t[100 + d] = "C"
            reads.append("".join(t))
    st = {}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, go, ge, band_lanes=-87, stats=st)
    for k, rd in enumerate(reads):
        status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, go, ge)
        assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (kind, k, rd)
        check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
    if kind == "marker_in_homopolymer":
        moved = len(reads) - 4
        assert all("-" in res[k][0] and "-" in res[k][1] for k in (moved + 1, moved + 2)), "the marker moved by one aligns along the next diagonal, with end gaps"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_main_diagonal_shortcut_on_random_low_complexity_references(mats, seed):
    """Randomised: references made of a repeat unit of period 1 .. 6 with a few point defects (where another diagonal fits as well as the main one
    except at the defects), reads = the reference with 0 .. 2 bases changed -- at random places, or so that a defect MOVES by one or two (the read
    then matches a shifted reference better than the reference itself).  The oracle decides every case, near-ties and ties included."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(991 + seed)
    for trial in range(12):
        L = int(rng.integers(160, 257))
        period = int(rng.integers(1, 7))
        unit = "".join(rng.choice(list("ACGT"), period))
        ref = list((unit * L)[:L])
        defects = sorted(set(int(x) for x in rng.integers(5, L - 5, int(rng.integers(0, 4)))))
        for q in defects:
            ref[q] = rng.choice([c for c in "ACGT" if c != ref[q]])
        amp = "".join(ref)
        go, ge, cut_incentive = [(-20, -2, 1), (-10, -1, 0), (-6, -2, 1), (-30, -3, 2)][trial % 4]
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = cut_incentive
        inc = [L // 2, L // 2 + 1]
        reads = [amp]
        for _ in range(150):
            t = list(amp)
            for _ in range(int(rng.integers(1, 3))):
                q = int(rng.integers(0, L))
                t[q] = rng.choice(list("ACGTN"))
            reads.append("".join(t))
        for q in defects:                                            # a defect moved: restore the unit's base, put the defect's base next door
            for d in (-2, -1, 1, 2):
                t = list(amp)
                t[q] = (unit * L)[q]
                t[q + d] = amp[q]
                reads.append("".join(t))
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, go, ge, band_lanes=-87, stats=st)
        for k, rd in enumerate(reads):
            status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, go, ge)
            assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (seed, trial, period, defects, go, ge, k, rd)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)


@pytest.mark.parametrize("scheme", [(1, -1, -1, -1, -2, -1, 0), (1, -1, 0, 0, -3, -1, 0), (2, -3, -1, -1, -5, -2, 1), (5, -4, -2, -1, -8, -1, 0),
                                    (3, -2, -2, -1, -4, -4, 2), (7, -8, -3, 0, -8, -3, 2), (1, -2, -1, 1, -2, -2, 1)])
def test_main_diagonal_shortcut_under_small_scores_where_ties_are_near(scheme):
    """make_matrix scorings with small numbers (match 1, mismatch -1, gap_open -2 ...): the main diagonal's lead over the next best path is a point
    or two, ties are common, and the certificate's strict inequalities decide.  Low-complexity references, reads with 0 .. 2 changed bases, moved
    defects.  Whatever the partition finishes itself or hands on, every alignment is the oracle's."""
    from crispresso2_amd import CRISPResso2Align as A
    match, mismatch, n_mis, n_match, go, ge, cut_incentive = scheme
    m = A.make_matrix(match_score=match, mismatch_score=mismatch, n_mismatch_score=n_mis, n_match_score=n_match)
    rng = np.random.default_rng(abs(hash(scheme)) % 100000)
    finished = 0
    for trial in range(6):
        L = int(rng.integers(200, 257))
        period = int(rng.integers(1, 5)) if trial % 2 else 0
        if period:
            unit = "".join(rng.choice(list("ACGT"), period))
            ref = list((unit * L)[:L])
            for q in rng.integers(5, L - 5, int(rng.integers(0, 3))):
                ref[int(q)] = rng.choice([c for c in "ACGT" if c != ref[int(q)]])
        else:
            ref = list(rng.choice(list("ACGT"), L))
        amp = "".join(ref)
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = cut_incentive
        inc = [L // 2, L // 2 + 1]
        reads = [amp] * 2
        for _ in range(90):
            t = list(amp)
            for _ in range(int(rng.integers(1, 3))):
                q = int(rng.integers(0, L))
                t[q] = rng.choice(list("ACGTN"))
            reads.append("".join(t))
        for q in range(8, L - 8, 37):                                # a base moved by one or two (two differing bases that a shifted diagonal explains)
            for d in (-2, -1, 1, 2):
                t = list(amp); t[q], t[q + d] = amp[q + d], amp[q]
                reads.append("".join(t))
        st = {}
        res, rec = E.align_batch(reads, [amp], [g], [inc], m, go, ge, band_lanes=-87, stats=st)
        finished += st["exact_copies"]
        for k, rd in enumerate(reads):
            status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, go, ge)
            assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (scheme, trial, period, k, rd)
            check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
    print("finished by the partition:", finished)


@pytest.mark.parametrize("L", [151, 203, 250, 256])
def test_main_diagonal_differences_at_block_edges(mats, L):
    """The partition compares a main-diagonal candidate 32 bytes per lane, 16 bytes per load, eight lanes per candidate, with an overlapping last
    block: one and two differing bases on every byte next to a 16-byte and a 32-byte boundary (both sides), the first and the last byte, pairs that
    straddle a boundary or sit in different lanes of the group -- every alignment is the oracle's and each of these reads is finished by the partition."""
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(7700 + L)
    amp = "".join(rng.choice(list("ACGT"), L))
    g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = 1
    inc = [L // 2 - 1, L // 2, L // 2 + 1]
    other = {"A": "C", "C": "G", "G": "T", "T": "A"}
    sub = lambda s_, q, c=None: s_[:q] + (c or other[s_[q]]) + s_[q + 1:]
    edges = sorted(set(q for b in range(0, L + 16, 16) for q in (b - 1, b, b + 1) if 0 <= q < L) | {0, L - 1, L - 2, L - 15, L - 16, L - 17, L - 31, L - 32, L - 33})
    reads = [amp]
    reads += [sub(amp, q) for q in edges] + [sub(amp, q, "N") for q in edges[::3]]
    pairs = [(q, q + 1) for q in edges if q + 1 < L] + [(q, min(L - 1, q + 32)) for q in edges[::2] if q + 32 != q and min(L - 1, q + 32) != q] + [(0, L - 1), (15, L - 16), (31, L - 32)]
    reads += [sub(sub(amp, a), b) for a, b in pairs if a != b]
    st = {}
    res, rec = E.align_batch(reads, [amp], [g], [inc], m, -20, -2, band_lanes=-87, stats=st)
    for k, rd in enumerate(reads):
        status, s1, s2, mt, ln = oracle.global_align_raw(rd, amp, m, g, -20, -2)
        assert status == 0 and rec[k]["status"] == 0 and res[k] == (s1, s2) and int(rec[k]["matches"]) == mt, (L, k, rd)
        check_record(rec[k], oracle.find_indels_substitutions(s1, s2, inc), s1, s2)
    assert st["exact_copies"] == len(reads), (st["exact_copies"], len(reads))      # a random amplicon: every one of them is certified
