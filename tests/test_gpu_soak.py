"""Randomised soak of the whole launch chain against the oracle: every alignment of a few tens of thousands, over mixed
read / reference lengths, three references per batch, both strands, several gap-parameter sets (incl. ones for which the
diagonal kernels do not apply), reads with N and IUPAC symbols, long indels, unrelated reads.  Everything the batch returns
(strings, matches, every record field) is compared, not a sample."""
import os

import numpy as np
import pytest

from helpers import adversarial_case, matrices
from test_gpu_parity import check_record

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from crispresso2_amd import _native
    return _native.default_context()       # raises loudly if the HIP extension or the GPU is missing

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def mutate(rng, s, p_sub=0.02):
    s = list(s)
    kind = int(rng.integers(0, 10))
    if kind < 3 and len(s) > 30:                                   # deletion
        p = int(rng.integers(5, len(s) - 10)); del s[p:p + int(rng.integers(1, 70))]
    elif kind < 5:                                                 # insertion
        p = int(rng.integers(1, len(s))); s[p:p] = list(rng.choice(list("ACGT"), int(rng.integers(1, 40))))
    elif kind == 5 and len(s) > 40:                                # both
        p = int(rng.integers(5, len(s) // 2)); del s[p:p + int(rng.integers(1, 12))]
        p = int(rng.integers(len(s) // 2, len(s) - 2)); s[p:p] = list(rng.choice(list("ACGT"), int(rng.integers(1, 12))))
    elif kind == 6:                                                # truncated / extended ends
        s = s[int(rng.integers(0, 30)):len(s) - int(rng.integers(0, 30))] + list(rng.choice(list("ACGT"), int(rng.integers(0, 25))))
    elif kind == 7:                                                # unrelated read
        s = list(rng.choice(list("ACGT"), int(rng.integers(20, 300))))
    for k in np.nonzero(rng.random(len(s)) < p_sub)[0]:
        s[k] = "ACGTN"[int(rng.integers(0, 5))]
    if rng.random() < 0.02 and s:
        s[int(rng.integers(0, len(s)))] = "RYKMSW"[int(rng.integers(0, 6))]      # IUPAC: no packed score row for this read
    return "".join(s) if s else "A"


@pytest.mark.parametrize("go,ge,seed", [(-20, -2, 1), (-10, -3, 2), (-5, -1, 3), (-1, -1, 4)])
def test_soak_every_alignment_vs_oracle(ctx, go, ge, seed):
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(seed)
    refs = ["".join(rng.choice(list("ACGT"), L)) for L in (int(rng.integers(60, 120)), int(rng.integers(180, 260)), int(rng.integers(120, 200)))]
    gis, incs = [], []
    for r in refs:
        g = np.zeros(len(r) + 1, dtype=np.int64)
        g[len(r) // 2 + 1] = 1
        if seed % 2 == 0:
            g[len(r) // 3] = 1                                     # a second cut site
        gis.append(g)
        incs.append(list(range(len(r) // 2 - 3, len(r) // 2 + 3)))
    n = 6000 if (go, ge) != (-1, -1) else 1500
    n = int(os.environ.get("C2_SOAK_N", n))                        # a longer soak on demand
    rids = rng.integers(0, 3, n).astype(np.uint16)
    strands = (rng.random(n) < 0.3).astype(np.uint8)
    truth = [mutate(rng, refs[r]) for r in rids]
    reads = [("".join(COMP[c] for c in reversed(t)) if st and set(t) <= set("ACGTN") else t) for t, st in zip(truth, strands)]
    strands = np.array([1 if (st and set(t) <= set("ACGTN")) else 0 for t, st in zip(truth, strands)], dtype=np.uint8)
    al = BatchAligner(refs, gis, incs, m, go, ge, ctx=ctx)
    res = al.align(reads, ref_ids=rids, strands=strands)
    n_undefined = 0
    for k in range(n):
        st, s1, s2, mt, ln = oracle.global_align_raw(truth[k], refs[rids[k]], m, gis[rids[k]], go, ge)
        r = res.records[k]
        if st != 0:                                                # the reference's undefined corner: both must flag it
            assert r["status"] != 0, k
            n_undefined += 1
            continue
        assert r["status"] == 0 and res.strings(k) == (s1, s2) and int(r["matches"]) == mt and int(r["aln_len"]) == ln, k
        check_record(r, oracle.find_indels_substitutions(s1, s2, incs[rids[k]]), s1, s2)
    assert n_undefined < n // 10
    tiers = ctx.tier_info()
    if (go, ge) == (-20, -2):
        assert len(tiers) == 3 and tiers[0] > tiers[2] > 0          # all four kernels did part of the batch


@pytest.mark.parametrize("go,ge,scale", [(-20, -2, 1), (-20, -4, 3), (-6, -2, 1), (-20, -7, 6), (-2, -3, 1)])
def test_soak_adversarial_gap_incentives(ctx, go, ge, scale):
    """Gap incentives where the diagonal kernels' out-of-band bound has exceptions (last row, row 0, blocks of rows, large
    values; gap_open > gap_extend) with reads that make the cheap gapped paths optimal: every alignment vs the oracle."""
    from crispresso2_amd.batch import BatchAligner
    import oracle
    m = matrices()["EDNAFULL"]
    rng = np.random.default_rng(500 + scale - go)
    n = int(os.environ.get("C2_SOAK_N", 4000))
    refs, gis, incs, reads, rids = adversarial_case(rng, n, L=int(rng.integers(100, 230)))
    gis = [g * scale for g in gis]
    al = BatchAligner(refs, gis, incs, m, go, ge, ctx=ctx)
    res = al.align(reads, ref_ids=np.array(rids, dtype=np.uint16))
    bad = 0
    for k in range(n):
        st, s1, s2, mt, ln = oracle.global_align_raw(reads[k], refs[rids[k]], m, gis[rids[k]], go, ge)
        r = res.records[k]
        if st != 0:
            assert r["status"] != 0, k
            bad += 1
            continue
        assert r["status"] == 0 and res.strings(k) == (s1, s2) and int(r["matches"]) == mt and int(r["aln_len"]) == ln, (k, rids[k])
    assert bad < n // 10
