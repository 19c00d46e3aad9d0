"""The `-p N` route of the unchanged caller ON THE HARDWARE (first item; the twin of
tests/test_dropin_reference_core.py::test_reference_main_with_p2_forks_workers_that_answer_from_the_parents_batches, which runs the
reference's unmodified main() over the wave emulator where /root/reference exists).

The GPU box has no reference, so the caller here is a stand-in with the reference's shape (CRISPRessoCORE.py:1735-1898): a function
NAMED `process_fastq` whose locals are `variantCache`, `args`, `refs`, `ref_names`, `aln_matrix`, which has used the device before
it fork()s (main() aligns its guides first, :3002-3015) and whose forked workers call the two drop-in modules once per read, as
`variant_file_generator_process` / `get_new_variant_object` do (:1198-1242, :627-798).  Checked: the workers' answers equal the
oracle's, none of them touched the inherited HIP runtime (their only ways out are the inherited memo and a spawned helper), and the
parent's context still works afterwards."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r'''
import json, multiprocessing as mp, os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
from types import SimpleNamespace
import numpy as np
if os.environ.get("C2_DROPIN_DEVICE") == "emulator":
    import dropin_inject
    dropin_inject.emulator_context()
from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R, refs as RF, synth, prime, _native

OUT = %(out)r
L, n = 180, %(n)d
amp, g, inc = synth.amplicon_setup(L)
reads = [r.tobytes().decode() for r in synth.make_reads(L, n)]
reads += [RF.reverse_complement(r) for r in reads[:n // 4]]            # some reads from the other strand
reads = list(dict.fromkeys(reads))


def get_new_variant_object(args, fastq_seq, refs, ref_names, aln_matrix):
    """the per-read calls of CRISPRessoCORE.py:655-724 (seed test, one or two alignments, the classifier on the better one)"""
    out = {}
    for name in ref_names:
        ref = refs[name]
        fw = sum(1 for s in ref['fw_seeds'][:args.aln_seed_count] if s in fastq_seq)
        rc = sum(1 for s in ref['rc_seeds'][:args.aln_seed_count] if s in fastq_seq)
        kw = dict(matrix=aln_matrix, gap_incentive=ref['gap_incentive'], gap_open=args.needleman_wunsch_gap_open, gap_extend=args.needleman_wunsch_gap_extend)
        cands = []
        if not (fw == 0 and rc > args.aln_seed_min):
            cands.append(A.global_align(fastq_seq, ref['sequence'], **kw))
        if not (fw > args.aln_seed_min and rc == 0):
            cands.append(A.global_align(RF.reverse_complement(fastq_seq), ref['sequence'], **kw))
        s1, s2, score = cands[0] if len(cands) == 1 or not cands[1][2] > cands[0][2] else cands[1]
        p = R.find_indels_substitutions(s1, s2, ref['include_idxs'])
        out[name] = [s1, s2, score, p['insertion_n'], p['deletion_n'], p['substitution_n'], list(p['all_substitution_positions']),
                     [list(c) for c in p['deletion_coordinates']], [list(c) for c in p['insertion_coordinates']], list(p['ref_positions'])[:8]]
    return out


def variant_file_generator_process(seq_list, args, refs, ref_names, aln_matrix, process_id):
    rows = [[s, get_new_variant_object(args, s, refs, ref_names, aln_matrix)] for s in seq_list]
    helper = _native._helper[1].calls if _native._helper[0] == os.getpid() else 0
    with open(os.path.join(OUT, "worker_%%d.json" %% process_id), "w") as fh:
        json.dump({"rows": rows, "stats": dict(prime.stats), "helper_calls": helper, "forked": _native.in_forked_child()}, fh)


def set_up_alignment(refs, ref_names):
    """what main() does before it reaches the read loop: alignments of its own on the device (guides, amplicons against each other, :3002-3015)"""
    m = A.read_matrix(os.path.join(os.path.dirname(A.__file__), "EDNAFULL"))
    return A.global_align(refs[ref_names[0]]['sequence'][40:60], refs[ref_names[0]]['sequence'], matrix=m,
                          gap_incentive=refs[ref_names[0]]['gap_incentive'], gap_open=-20, gap_extend=-2)


def process_fastq(variantCache, ref_names, refs, args, n_processes):
    aln_matrix = A.read_matrix(os.path.join(os.path.dirname(A.__file__), "EDNAFULL"))
    keys = list(variantCache.keys())
    b = [len(keys) * k // n_processes for k in range(n_processes + 1)]
    procs = []
    for i in range(n_processes):
        p = mp.get_context("fork").Process(target=variant_file_generator_process, args=(keys[b[i]:b[i + 1]], args, refs, ref_names, aln_matrix, i))
        p.start()
        procs.append(p)
    for p in procs:
        p.join(%(join)d)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    return codes


args = SimpleNamespace(aln_seed_count=5, aln_seed_min=2, needleman_wunsch_gap_open=-20, needleman_wunsch_gap_extend=-2,
                       use_legacy_insertion_quantification=False, crispresso_merge=False)
refs = {"Reference": RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)}
setup = set_up_alignment(refs, ["Reference"])
codes = process_fastq({r: 1 for r in reads}, ["Reference"], refs, args, 2)
assert set_up_alignment(refs, ["Reference"]) == setup                   # the parent's context after the forks: still usable
print("FORK_PARENT", json.dumps({"codes": codes, "stats": prime.counters(), "reads": len(reads)}))
'''


def _run(tmp_path, n, env_extra, join=600):
    out = str(tmp_path)
    env = dict(os.environ)
    for k in ("C2_PRIME_FROM_ARGV", "C2_PRIME_FASTQ", "C2_PRIME_FROM_FRAMES"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, tests=HERE, out=out, n=n, join=join)], capture_output=True, text=True,
                       cwd=out, env=env, timeout=join + 600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    parent = json.loads([x for x in p.stdout.splitlines() if x.startswith("FORK_PARENT")][-1].split(" ", 1)[1])
    workers = []
    for i in range(2):
        with open(os.path.join(out, "worker_%d.json" % i)) as fh:
            workers.append(json.load(fh))
    return parent, workers


def _expected(rows):
    """the same per-read flow through the ORACLE (test infrastructure: the C restatement of the reference, oracle/c2_oracle.c)"""
    import oracle
    from crispresso2_amd import CRISPResso2Align as A, refs as RF, synth
    L = 180
    amp, g, inc = synth.amplicon_setup(L)
    ref = RF.make_ref("Reference", amp, [L // 2], inc, min_aln_score=60)
    m = A.read_matrix(os.path.join(os.path.dirname(A.__file__), "EDNAFULL"))
    out = []
    for s, _ in rows:
        fw = sum(1 for x in ref['fw_seeds'][:5] if x in s)
        rc = sum(1 for x in ref['rc_seeds'][:5] if x in s)
        cands = []
        if not (fw == 0 and rc > 2):
            cands.append(tuple(oracle.global_align(s, amp, m, ref['gap_incentive'], -20, -2)))
        if not (fw > 2 and rc == 0):
            cands.append(tuple(oracle.global_align(RF.reverse_complement(s), amp, m, ref['gap_incentive'], -20, -2)))
        s1, s2, score = cands[0] if len(cands) == 1 or not cands[1][2] > cands[0][2] else cands[1]
        p = oracle.find_indels_substitutions(s1, s2, inc)
        out.append([s, {"Reference": [s1, s2, score, p['insertion_n'], p['deletion_n'], p['substitution_n'], list(p['all_substitution_positions']),
                                      [list(c) for c in p['deletion_coordinates']], [list(c) for c in p['insertion_coordinates']], list(p['ref_positions'])[:8]]}])
    return out


def _check(parent, workers, primed):
    assert parent["codes"] == [0, 0], parent
    n_rows = 0
    for w in workers:
        assert w["forked"] is True
        assert json.loads(json.dumps(_expected(w["rows"]))) == w["rows"]
        n_rows += len(w["rows"])
        if primed:
            assert w["helper_calls"] == 0 and w["stats"]["align_hits"] >= len(w["rows"]) and w["stats"]["classify_hits"] == len(w["rows"]), w["stats"]
            assert w["stats"]["batches"] == parent["stats"]["batches"]
        else:
            assert w["stats"]["align_hits"] == 0 and w["helper_calls"] >= 2 * len(w["rows"]), (w["stats"], w["helper_calls"])
    assert n_rows == parent["reads"]
    if primed:
        st = parent["stats"]
        assert st["from_frames"] == 1 and st["before_fork"] == 1 and 1 <= st["batches"] <= 2 and st["classify_batches"] == 1, st


@pytest.mark.gpu
def test_forked_workers_answer_from_the_batches_the_parent_primed_before_the_fork(tmp_path):
    parent, workers = _run(tmp_path, 600, {})
    _check(parent, workers, primed=True)


@pytest.mark.gpu
def test_forked_workers_without_priming_go_through_a_spawned_helper_never_the_inherited_context(tmp_path):
    parent, workers = _run(tmp_path, 60, {"C2_PRIME_FROM_FRAMES": "0", "C2_PRIME_FROM_ARGV": "0"})
    _check(parent, workers, primed=False)


def test_fork_route_stand_in_on_the_emulator(tmp_path):
    """the same two scenarios with the wave emulator in the GPU's place (a forked child treats it as it must treat HIP)"""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    parent, workers = _run(tmp_path / "a", 48, {"C2_DROPIN_DEVICE": "emulator"})
    _check(parent, workers, primed=True)
    parent, workers = _run(tmp_path / "b", 12, {"C2_DROPIN_DEVICE": "emulator", "C2_PRIME_FROM_FRAMES": "0", "C2_PRIME_FROM_ARGV": "0"})
    _check(parent, workers, primed=False)
