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
    env = dict(os.environ, C2_DROPIN_DEVICE="emulator", C2_PRIME_FROM_ARGV="0")        # (every call per call: the priming is off for this one)
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


@pytest.mark.parametrize("name", sorted(RUNS))
def test_reference_main_unchanged_gets_its_alignments_from_one_batch_when_primed(name, tmp_path):
    """the same unmodified main() with NOTHING in the environment (crispresso2_amd.prime watches the
    command line by default and reads the -r1 of the reference's own): after the first misses of the hot loop ALL unique reads of the FASTQ are aligned in one
    device batch per amplicon (and classified in one), the rest of the run's >200 calls are look-ups -- the files are the same and
    only a handful of per-call launches remain."""
    extra, expected = RUNS[name]
    code = RUNNER % dict(tests=HERE, fastq=os.path.join(REF, "tests", "FANC.Cas9.fastq"), out=str(tmp_path), extra=extra)
    env = dict(os.environ, C2_DROPIN_DEVICE="emulator", C2_PRIME_REPORT="1")
    env.pop("C2_PRIME_FROM_ARGV", None)
    env.pop("C2_PRIME_FASTQ", None)
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
