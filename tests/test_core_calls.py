"""Every call the reference's own main() made into its two Cython modules while running its two end-to-end tests
(tests/golden/core_calls.json.gz, recorded by make_golden.py --core-calls with the reference's OWN modules: 671 distinct
global_align calls -- quantification-window and flexiguide alignments, incl. the default flexiguide "None" whose
characters lie beyond the score matrix, the hot loop of process_fastq, BLOSUM62 amino-acid alignments -- and 385
find_indels_substitutions calls), answered by the product's drop-in modules through their public API, the way
CRISPRessoCORE.py calls them.  CPU: device calls on the wave emulator.  GPU: the same replay on the MI355X (the reference's
sources cannot travel to the GPU box; this fixture is what travels)."""
import os
import sys

import numpy as np
import pytest

from helpers import load_golden, matrices, payload_diff

HERE = os.path.dirname(os.path.abspath(__file__))


def replay(A, R, mats):
    d = load_golden("core_calls.json.gz")
    n_align = n_classify = 0
    for k in d["calls"]:
        if k["fn"] == "global_align":
            g = np.zeros(k["gi_len"], dtype=int)
            for i, v in k["gi_nonzero"]:
                g[i] = v
            out = A.global_align(k["seqj"], k["seqi"], matrix=mats[k["matrix"]], gap_incentive=g, gap_open=k["gap_open"],
                                 gap_extend=k["gap_extend"])
            assert list(out) == k["out"], (k["seqj"], k["seqi"][:30])
            n_align += 1
        else:
            p = getattr(R, k["fn"])(k["read_al"], k["ref_al"], np.array(k["include"]))
            assert payload_diff(p, k["out"]) == [], (k["read_al"], k["ref_al"])
            n_classify += 1
    assert n_align == 671 and n_classify == 385
    return d["runs"]


def test_reference_main_calls_replayed_on_the_emulator():
    sys.path.insert(0, HERE)
    import dropin_inject as D
    from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R, _native
    saved = _native.default_context
    ctx = D._EmuContext()
    _native.default_context = lambda *a, **k: ctx
    try:
        runs = replay(A, R, matrices())
    finally:
        _native.default_context = saved
    assert set(runs) == {"CRISPResso_on_FANC.Cas9", "CRISPResso_on_params"}


@pytest.mark.gpu
def test_reference_main_calls_replayed_on_the_gpu():
    from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R
    replay(A, R, matrices())


def _fanc_fastq(tmp_path):
    fq = tmp_path / "FANC.Cas9.fastq"
    fq.write_text(load_golden("fanc_run.json.gz")["fastq"])
    return str(fq)


def _primed_replay(A, R, tmp_path):
    """the same replay with the run's FASTQ registered (crispresso2_amd.prime): identical answers, most of them from batches"""
    from crispresso2_amd import prime
    prime.register_reads(_fanc_fastq(tmp_path))
    try:
        replay(A, R, matrices())
        st = dict(prime.stats)
    finally:
        prime.clear()
    # two runs x (one or two amplicons) x (forward, sometimes reverse-complement) batches; the hot loops' calls were look-ups
    assert 2 <= st["batches"] <= 8 and 1 <= st["classify_batches"] <= 6, st
    assert st["align_hits"] > 350 and st["classify_hits"] > 250, st
    assert st["per_call_align"] == 671 - st["align_hits"] and st["per_call_classify"] == 385 - st["classify_hits"], st
    return st


def test_reference_main_calls_replayed_from_primed_batches_on_the_emulator(tmp_path):
    sys.path.insert(0, HERE)
    import dropin_inject as D
    from pipeline_on_emulator import EmulatedAligner
    from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R, _native, batch
    saved = (_native.default_context, batch.BatchAligner)
    ctx = D._EmuContext()
    _native.default_context = lambda *a, **k: ctx
    batch.BatchAligner = EmulatedAligner
    try:
        _primed_replay(A, R, tmp_path)
    finally:
        _native.default_context, batch.BatchAligner = saved


@pytest.mark.gpu
def test_reference_main_calls_replayed_from_primed_batches_on_the_gpu(tmp_path):
    from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R
    _primed_replay(A, R, tmp_path)
