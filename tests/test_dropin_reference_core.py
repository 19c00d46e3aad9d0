"""north_star: "keeping the CRISPResso2Align / CRISPRessoCOREResources Cython module API so CRISPRessoCORE.py calls it
unchanged" -- executed.  The reference's UNMODIFIED CRISPRessoCORE.main() runs its own two end-to-end tests
(tests/Makefile:20 CRISPResso_on_FANC.Cas9, :23 CRISPResso_on_params) with the product's two modules installed under the
reference's module names (tests/dropin_inject.py = INTEGRATION.md section 1), and the files the reference repository keeps as
expected results must come out byte for byte; then the reference's own unit-test files for the two modules (and for
CRISPRessoCORE's consensus / variant functions, which call them) are collected unchanged and run against the shim.
Device calls go to the wave emulator here; tests/test_core_calls.py replays every call those two runs make (recorded with the
reference's own modules) through the product's modules on the GPU (-m gpu) and on the emulator.
Skipped when /root/reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("C2_REFERENCE_DIR", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "CRISPResso2")), reason="reference sources not present")

FANC_AMPLICON = ("CGGATGTTCCAATCAGTACGCAGAGAGTCGCCGTCTCCAAGGTGAAAGCGGAAGTAGGGCCTTCGCGCACCTCATGGAATCCCTTCTGCAGCACCTGGATCGCTTTTCCGAGCTTCTGGCGGTCTCAAG"
                 "CACTACCTACGTCAGCACCTGGGACCCCGCCACCGTGCGCCGGGCCTTGCAGTGGGCGCGCTACCTGCGCCACATCCATCGGCGCTTTGGTCGG")
HDR = ("CGGCCGGATGTTCCAATCAGTACGCAGAGAGTCGCCGTCTCCAAGGTGAAAGCTGAAGTAGGGCCTTCGCGCACCTCATGGAATCCCTTCTGCAGCTTTTCCGAGCTTCTGGCGGTCTCAAGCACTACCTACG"
       "TCAGCACCTGGGACCCCGCCACCGTGCGCCGGGCCTTGCAGTGGGCGCGCTACCTGCGCCACATCCATCGGCGCTTTGGTCGG")
RUNS = {
    "CRISPResso_on_FANC.Cas9": (["-a", FANC_AMPLICON, "-g", "GGAATCCCTTCTGCAGCACC"],
                                {"CRISPResso_quantification_of_editing_frequency.txt": "CRISPResso_quantification_of_editing_frequency.txt",
                                 "Nucleotide_frequency_table.txt": "Nucleotide_frequency_table.txt"}),
    "CRISPResso_on_params": (["-a", FANC_AMPLICON, "-g", "GGAATCCCTTCTGCAGCACC", "-e", HDR, "-c", "GGGCCTTCGCGCACCTCATGGAATCCCTTCTGCAGCACCTGGATCGCTTTT",
                              "--dump", "-qwc", "20-30_45-50", "-q", "30", "--default_min_aln_score", "80", "-an", "FANC", "-n", "params",
                              "--base_editor_output", "-fg", "AGCCTTGCAGTGGGCGCGCTA,CCCACTGAAGGCCC", "--dsODN", "GCTAGATTTCCCAAGAAGA", "-gn", "hi",
                              "-fgn", "dear"],
                             {"CRISPResso_quantification_of_editing_frequency.txt": "CRISPResso_quantification_of_editing_frequency.txt",
                              "FANC.Nucleotide_frequency_table.txt": "FANC.Nucleotide_frequency_table.txt"}),
}

RUNNER = r'''
import os, sys
sys.path.insert(0, %(tests)r)
import dropin_inject as D
A, R = D.inject()
core = D.reference_core()
# the reference resolved its two imports to the product's modules
assert core.CRISPResso2Align is A and core.CRISPRessoCOREResources is R, "CRISPRessoCORE did not import the shim"
from CRISPResso2 import CRISPRessoShared
assert CRISPRessoShared.CRISPRessoCOREResources is R
assert A.__name__ == "crispresso2_amd.CRISPResso2Align" and R.__name__ == "crispresso2_amd.CRISPRessoCOREResources"
calls = {"align": 0, "classify": 0}
ga, fi = A.global_align, R.find_indels_substitutions
def ga_(*a, **k):
    calls["align"] += 1
    return ga(*a, **k)
def fi_(*a, **k):
    calls["classify"] += 1
    return fi(*a, **k)
A.global_align, R.find_indels_substitutions = ga_, fi_
D.run_core_main(["CRISPResso", "-r1", %(fastq)r, "-o", %(out)r, "--suppress_plots", "--suppress_report"] + %(extra)r)
print("DROPIN_CALLS", calls["align"], calls["classify"])
from crispresso2_amd import prime
print("DROPIN_PRIME", " ".join("%%s=%%d" %% kv for kv in sorted(prime.stats.items())))
'''


@pytest.mark.parametrize("name", sorted(RUNS))
def test_reference_main_runs_unchanged_over_the_shim(name, tmp_path):
    extra, expected = RUNS[name]
    code = RUNNER % dict(tests=HERE, fastq=os.path.join(REF, "tests", "FANC.Cas9.fastq"), out=str(tmp_path), extra=extra)
    env = dict(os.environ, C2_DROPIN_DEVICE="emulator", C2_PRIME_FROM_ARGV="0", C2_PRIME_FROM_FRAMES="0")        # (every call per call: the priming is off for this one)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=3000)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("DROPIN_CALLS")][-1].split()
    assert int(line[1]) > 200 and int(line[2]) > 150, line           # the hot loop really went through the shim
    assert " batches=0 " in [x for x in p.stdout.splitlines() if x.startswith("DROPIN_PRIME")][-1] + " "
    outdir = os.path.join(str(tmp_path), name)
    for made, kept in expected.items():
        with open(os.path.join(outdir, made)) as fh:
            got = fh.read()
        with open(os.path.join(REF, "tests", "expectedResults", name, kept)) as fh:
            want = fh.read()
        assert got == want, made


@pytest.mark.parametrize("test_file", ["test_CRISPResso2Align.py", "test_CRISPRessoCOREResources.py", "test_CRISPRessoCORE.py"])
def test_reference_unit_tests_collected_unchanged_pass_against_the_shim(test_file, tmp_path):
    env = dict(os.environ, C2_DROPIN_DEVICE="emulator", PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=HERE + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "pytest", "-p", "dropin_inject", "-q", "-x", "-p", "no:cacheprovider",
                        "--rootdir", str(tmp_path), "-c", os.devnull, os.path.join(REF, "tests", "unit_tests", test_file)],
                       capture_output=True, text=True, cwd=REF, env=env, timeout=3000)   # (the files open "./CRISPResso2/EDNAFULL"; nothing is written there)
    tail = p.stdout[-2500:] + p.stderr[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and " failed" not in p.stdout, tail


@pytest.mark.parametrize("source", ["frames", "argv"])
@pytest.mark.parametrize("name", sorted(RUNS))
def test_reference_main_unchanged_gets_its_alignments_from_one_batch_when_primed(name, source, tmp_path):
    """the same unmodified main() with NOTHING in the environment (crispresso2_amd.prime watches the
    command line by default and reads the -r1 of the reference's own): after the first misses of the hot loop ALL unique reads of the FASTQ are aligned in one
    device batch per amplicon (and classified in one), the rest of the run's >200 calls are look-ups -- the files are the same and
    only a handful of per-call launches remain."""
    extra, expected = RUNS[name]
    code = RUNNER % dict(tests=HERE, fastq=os.path.join(REF, "tests", "FANC.Cas9.fastq"), out=str(tmp_path), extra=extra)
    env = dict(os.environ, C2_DROPIN_DEVICE="emulator", C2_PRIME_REPORT="1")
    env.pop("C2_PRIME_FROM_ARGV", None)
    env.pop("C2_PRIME_FASTQ", None)
    env.pop("C2_PRIME_FROM_FRAMES", None)
    if source == "argv":                                              # (round 3's route: the -r1 of the command line, lazily per amplicon; round 5's
        env["C2_PRIME_FROM_FRAMES"] = "0"                             # default finds process_fastq's variantCache in the caller's frames first)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=3000)
    assert "crispresso2_amd.prime: " in p.stderr and "align_hits=" in p.stderr              # (the counters, published at exit)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("DROPIN_CALLS")][-1].split()
    n_align, n_classify = int(line[1]), int(line[2])
    assert n_align > 200 and n_classify > 150, line
    st = dict(kv.split("=") for kv in [x for x in p.stdout.splitlines() if x.startswith("DROPIN_PRIME")][-1].split()[1:])
    st = {k: int(v) for k, v in st.items()}
    n_amplicons = 2 if "params" in name else 1
    assert 1 <= st["batches"] <= 2 * n_amplicons and 1 <= st["classify_batches"] <= 2 * n_amplicons, st
    assert st["from_frames"] == (1 if source == "frames" else 0), st
    assert st["align_hits"] > (350 if "params" in name else 150) and st["classify_hits"] > 100, st
    # what is left per call: the run's set-up alignments (guides, amplicons against each other) and -- with a coding sequence (-c, the
    # params run) -- the exon analysis of the aggregation loop, which aligns SLICES of aligned reads against the exon with gap
    # penalties -1 / -1 (CRISPRessoCORE.py:4100-4171: strings no FASTQ holds).  Every hot-loop call was a look-up.
    assert st["per_call_align"] == n_align - st["align_hits"] and st["per_call_classify"] == n_classify - st["classify_hits"], st
    assert st["per_call_align"] < (200 if "params" in name else 40) and st["per_call_classify"] < 8, st
    outdir = os.path.join(str(tmp_path), name)
    for made, kept in expected.items():
        with open(os.path.join(outdir, made)) as fh:
            got = fh.read()
        with open(os.path.join(REF, "tests", "expectedResults", name, kept)) as fh:
            want = fh.read()
        assert got == want, made


# ---------------------------------------------------------------- the unchanged caller's other routes: -p N (fork) and pairs
# (first item: CRISPRessoCORE.py:1870-1898 forks its workers AFTER main() has aligned its guides on the device; a forked
# child cannot touch the HIP runtime it inherits.  crispresso2_amd.prime primes what the workers will ask for BEFORE the fork, from the
# frames of the reference's own read loop; a call that misses in a child goes to a spawned helper.)
FASTP_STUB = r'''#!%(python)s
"""stand-in for fastp with every filter switched off (the reference's default options for --crispresso_merge,
CRISPRessoCORE.py:3672-3677): reads pass through unchanged; TEST INFRASTRUCTURE ONLY"""
import gzip, sys
a = sys.argv[1:]
if "--version" in a:
    sys.stderr.write("fastp 0.23.4\n")
    sys.exit(0)
get = lambda f: a[a.index(f) + 1] if f in a else None
for src, dst in ((get("-i"), get("--out1") or get("-o")), (get("-I"), get("--out2"))):
    if src and dst:
        with (gzip.open(src, "rb") if src.endswith(".gz") else open(src, "rb")) as fi, (gzip.open(dst, "wb", 1) if dst.endswith(".gz") else open(dst, "wb")) as fo:
            fo.write(fi.read())
for f in ("--json", "--html"):
    if get(f):
        open(get(f), "w").write("{}")
'''

RUNNER_WORKERS = r'''
import json, os, sys
sys.path.insert(0, %(tests)r)
OUT = %(out)r
if %(shim)r:
    import dropin_inject as D
    A, R = D.inject()
    core = D.reference_core()
    assert core.CRISPResso2Align is A and core.CRISPRessoCOREResources is R, "CRISPRessoCORE did not import the shim"
    from crispresso2_amd import prime, _native
    orig = core.variant_file_generator_process
    def worker(*a, **k):                                              # (instrumentation only: the worker's counters, which die with it)
        try:
            return orig(*a, **k)
        finally:
            with open(os.path.join(OUT, "prime_stats_%%d.json" %% os.getpid()), "w") as fh:
                json.dump(dict(prime.stats, helper_calls=(_native._helper[1].calls if _native._helper[0] == os.getpid() else 0)), fh)
    core.variant_file_generator_process = worker
    D.run_core_main(%(argv)r)
    print("DROPIN_PARENT", json.dumps(prime.counters()))
else:
    # the same command with the reference's OWN compiled modules (oracle/_ref): what the files must be
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(tests)r, "golden"))
    import importlib, types
    import importlib.metadata as md
    from oracle._ref.c2ref import CRISPResso2Align as A, CRISPRessoCOREResources as R
    pkg = types.ModuleType("CRISPResso2")
    pkg.__path__ = [os.path.join(%(ref)r, "CRISPResso2")]
    sys.modules["CRISPResso2"] = pkg
    sys.modules["CRISPResso2.CRISPResso2Align"] = A
    sys.modules["CRISPResso2.CRISPRessoCOREResources"] = R
    pkg.CRISPResso2Align, pkg.CRISPRessoCOREResources = A, R
    sb = types.ModuleType("seaborn")
    sb.set_context = sb.set = sb.set_style = sb.set_theme = lambda *a, **k: None
    sb.matrix = types.SimpleNamespace(_HeatMapper=object)
    sb.utils = types.SimpleNamespace()
    sys.modules["seaborn"] = sb
    orig_v = md.version
    md.version = lambda name: "2.3.4" if name.lower().startswith("crispresso") else orig_v(name)
    core = importlib.import_module("CRISPResso2.CRISPRessoCORE")
    sys.argv = %(argv)r
    try:
        core.main()
    except SystemExit as e:
        assert e.code in (0, None), e.code
'''


def _run_workers(tmp_path, argv, shim, env_extra=None, sub="shim"):
    out = os.path.join(str(tmp_path), sub)
    os.makedirs(out, exist_ok=True)
    bindir = os.path.join(str(tmp_path), "bin")
    os.makedirs(bindir, exist_ok=True)
    stub = os.path.join(bindir, "fastp")
    if not os.path.exists(stub):
        with open(stub, "w") as fh:
            fh.write(FASTP_STUB % dict(python=sys.executable))
        os.chmod(stub, 0o755)
    argv = [a if a != "@OUT@" else out for a in argv]
    code = RUNNER_WORKERS % dict(tests=HERE, root=ROOT, ref=REF, out=out, shim=shim, argv=argv)
    env = dict(os.environ, C2_DROPIN_DEVICE="emulator", PATH=bindir + os.pathsep + os.environ.get("PATH", ""))
    for k in ("C2_PRIME_FROM_ARGV", "C2_PRIME_FASTQ", "C2_PRIME_FROM_FRAMES"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=out, env=env, timeout=3000)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    import glob
    import json
    workers = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(out, "prime_stats_*.json")))]
    parent = [json.loads(x.split(" ", 1)[1]) for x in p.stdout.splitlines() if x.startswith("DROPIN_PARENT")]
    return out, (parent[-1] if parent else None), workers


def _result_files(folder):
    """every result table of a run (not the logs, the run info with its timestamps and paths, or the zip with its member dates)"""
    out = {}
    for dirpath, _, files in os.walk(folder):
        for f in files:
            if f.endswith(".txt") and "RUNNING_LOG" not in f:
                with open(os.path.join(dirpath, f), "rb") as fh:
                    out[os.path.relpath(os.path.join(dirpath, f), folder)] = fh.read()
    return out


def test_reference_main_with_p2_forks_workers_that_answer_from_the_parents_batches(tmp_path):
    """`CRISPResso -r1 FANC.Cas9.fastq ... -p 2`, main() unmodified: process_fastq forks two workers (CRISPRessoCORE.py:1870-1898).  Before the
    first fork the parent has aligned and classified every unique read in one batch per strand; the workers' >200 calls are look-ups in
    the memory they inherited -- no helper process, no launch -- and the files are the reference's expected results."""
    name = "CRISPResso_on_FANC.Cas9"
    argv = ["CRISPResso", "-r1", os.path.join(REF, "tests", "FANC.Cas9.fastq"), "-o", "@OUT@", "--suppress_plots", "--suppress_report"] + RUNS[name][0] + ["-p", "2"]
    out, parent, workers = _run_workers(tmp_path, argv, shim=True)
    assert parent["from_frames"] == 1 and parent["before_fork"] == 1 and 1 <= parent["batches"] <= 2 and parent["classify_batches"] == 1, parent
    assert len(workers) == 2
    for w in workers:
        assert w["helper_calls"] == 0 and w["align_hits"] > 100 and w["classify_hits"] > 90, w
        assert w["per_call_align"] == parent["per_call_align"] and w["per_call_classify"] == parent["per_call_classify"], (w, parent)   # (inherited: the parent's set-up calls)
        assert w["batches"] == parent["batches"]                     # nothing was launched in a worker
    for made, kept in RUNS[name][1].items():
        with open(os.path.join(out, name, made)) as fh, open(os.path.join(REF, "tests", "expectedResults", name, kept)) as fk:
            assert fh.read() == fk.read(), made


def test_forked_workers_that_were_not_primed_are_served_by_a_spawned_helper(tmp_path):
    """The same run with the priming switched off: every call of the forked workers misses, and none of them may touch the context the
    parent opened.  Each worker starts ONE helper process (its own device runtime) and gets its answers from there; same files."""
    name = "CRISPResso_on_FANC.Cas9"
    argv = ["CRISPResso", "-r1", os.path.join(REF, "tests", "FANC.Cas9.fastq"), "-o", "@OUT@", "--suppress_plots", "--suppress_report"] + RUNS[name][0] + ["-p", "2"]
    out, parent, workers = _run_workers(tmp_path, argv, shim=True, env_extra={"C2_PRIME_FROM_ARGV": "0", "C2_PRIME_FROM_FRAMES": "0"})
    assert parent["batches"] == 0 and parent["from_frames"] == 0, parent
    assert len(workers) == 2
    for w in workers:
        assert w["align_hits"] == 0 and w["helper_calls"] > 190, w
        assert w["helper_calls"] == (w["per_call_align"] - parent["per_call_align"]) + (w["per_call_classify"] - parent["per_call_classify"]), (w, parent)
    for made, kept in RUNS[name][1].items():
        with open(os.path.join(out, name, made)) as fh, open(os.path.join(REF, "tests", "expectedResults", name, kept)) as fk:
            assert fh.read() == fk.read(), made


@pytest.mark.parametrize("procs", ["1", "2"])
def test_reference_main_on_read_pairs_unchanged_over_the_shim(procs, tmp_path):
    """`CRISPResso -r1 R1.fastq -r2 R2.fastq --crispresso_merge [-p 2]`, main() unmodified (process_paired_fastq, CRISPRessoCORE.py:1245-1733:
    2-4 global_align calls per pair and amplicon, the classifier on the pair's consensus).  The pairs are tests/golden/paired_fastq.json.gz's
    (exact duplicates, duplicates with other qualities, pairs from the other strand, unrelated pairs).  The expected files come from the same
    command with the reference's own compiled modules; over the shim the hot loop's calls are look-ups -- both reads of every pair aligned in
    one batch per strand, the consensus alignments formed and classified on the device before the loop (or the fork) starts."""
    import gzip
    import json
    with gzip.open(os.path.join(HERE, "golden", "paired_fastq.json.gz"), "rt") as fh:
        gold = json.load(fh)
    r1, r2 = os.path.join(str(tmp_path), "r1.fastq"), os.path.join(str(tmp_path), "r2.fastq")
    with open(r1, "w") as fh:
        fh.write(gold["fastq1"])
    with open(r2, "w") as fh:
        fh.write(gold["fastq2"])
    argv = ["CRISPResso", "-r1", r1, "-r2", r2, "--crispresso_merge", "-o", "@OUT@", "--suppress_plots", "--suppress_report",
            "-a", FANC_AMPLICON, "-g", "GGAATCCCTTCTGCAGCACC", "-p", procs]
    # (one amplicon: with a second one the reference's own main() stops at CRISPRessoCORE.py:4243, which unpacks four of the five fields
    # that the paired route's ref_aln_details carry)
    want_dir, _, _ = _run_workers(tmp_path, argv, shim=False, sub="reference")
    got_dir, parent, workers = _run_workers(tmp_path, argv, shim=True)
    want, got = _result_files(want_dir), _result_files(got_dir)
    assert len(want) >= 10 and sorted(want) == sorted(got), (sorted(want), sorted(got))
    for f in want:
        assert want[f] == got[f], f
    assert parent["from_frames"] >= 1 and parent["consensus_batches"] >= 1 and 1 <= parent["batches"] <= 2, parent
    if procs == "2":
        assert parent["before_fork"] == 1 and len(workers) == 2
        for w in workers:
            # every call of the workers' loops was a look-up (a per-call launch in a forked worker can only be a helper call); what the
            # counters inherited is the parent's set-up alignments
            assert w["helper_calls"] == 0 and w["align_hits"] > 100 and w["classify_hits"] > 40, w
            assert w["per_call_align"] == parent["per_call_align"] and w["per_call_classify"] == 0 and w["align_misses"] == w["per_call_align"], (w, parent)
        # the parent's second pass (:1450-1513: pairs seen more than once whose consensus chose a base by quality, re-done per occurrence
        # with that occurrence's qualities): alignments from the memo, the classifier per call where the consensus differs from the first occurrence's
        assert parent["align_hits"] > 100 and parent["per_call_classify"] < 40, parent
    else:
        assert parent["align_hits"] > 300 and parent["classify_hits"] > 100, parent
        # per call: the run's set-up alignments, and the pairs the parent re-aligns per occurrence with that occurrence's own qualities
        # (:1450-1513) when their consensus differs from the first occurrence's
        assert parent["per_call_classify"] < 40, parent
