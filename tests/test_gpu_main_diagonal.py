"""The main-diagonal shortcut of c2_align_partition_kernel ON THE HARDWARE.

58 % of the headline batch is finished by the partition without a DP: a class-0 read with at most two differing bases gets its rows and record
from the compared registers where c2_main_diagonal_certificate + the equal-byte counts of the diagonals +-1 / +-2 prove the diagonal unbeatable.
What it must not change are the reference's tie rules (CRISPResso2Align.pyx:349-358, 361-421).  The adversarial suite of
tests/test_kernel_emulated.py (tandem repeats, moved markers in homopolymers, small scorings where ties are a point away, N / IUPAC / lower
case, block edges of the 16-byte loads) ran on the CPU wave emulator only; the kernel's 8-lane group sums, its __ballot / __shfl position
exchange and the overlapping last block are exactly what differs between emulator and silicon.  Here the SAME test functions run with the
emulator's launch replaced by the product's (BatchAligner.align_device through the C ABI), and each asserts that the partition did finish reads.
Plus a soak: 200+ low-complexity references x 2,000+ reads, every alignment against the oracle and against the same batch with the shortcut off."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import test_kernel_emulated as K
from helpers import matrices

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mats():
    return matrices()


@pytest.fixture(scope="module")
def ctx():
    from crispresso2_amd import _native
    return _native.default_context()       # raises loudly if the HIP extension or the GPU is missing


def device_batch(ctx, reads, refs, gap_incentives, includes, matrix, gap_open, gap_extend):
    """one batch through the product's launch chain, rows zeroed first -> (aln_read, aln_ref uint8 [n, stride], records, partition_info)"""
    import torch
    from crispresso2_amd import _native
    from crispresso2_amd.batch import BatchAligner, pack_reads
    al = BatchAligner(list(refs), list(gap_incentives), [list(x) for x in includes], matrix, gap_open, gap_extend, ctx=ctx)
    arena, off = pack_reads(reads)
    n = len(reads)
    max_lj = int((off[1:] - off[:-1]).max())
    stride = al.stride_for(max_lj)
    dev = torch.device("cuda", 0)
    d_reads = torch.from_numpy(arena.copy()).to(dev)
    d_off = torch.from_numpy(off.view(np.int64).copy()).to(dev)
    o1 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    o2 = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    rec = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    al.align_device(n, d_reads.data_ptr(), d_off.data_ptr(), o1.data_ptr(), o2.data_ptr(), rec.data_ptr(), stride, max_lj,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    info = ctx.partition_info()
    return o1.cpu().numpy(), o2.cpu().numpy(), rec.cpu().numpy().view(_native.REC_DTYPE).reshape(-1), info


class OnDevice:
    """stands where tests/emu_driver.align_batch stands in the emulated tests: same arguments, same return value, same `stats` keys"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.finished = 0
        self.batches = 0

    def __call__(self, reads, refs, gap_incentives, includes, matrix, gap_open, gap_extend, ref_ids=None, strands=None, all_refs=False,
                 force_R=0, grid=0, no_packed=False, band_lanes=0, stats=None):
        assert band_lanes == -87 and ref_ids is None and strands is None and not all_refs and not no_packed     # (the default chain, partition in front)
        o1, o2, rec, info = device_batch(self.ctx, reads, refs, gap_incentives, includes, matrix, gap_open, gap_extend)
        self.batches += 1
        self.finished += info["finished_by_partition"]
        if stats is not None:
            stats["classes"] = [a + b for a, b in zip(stats.get("classes", [0] * 7), info["classes"])]
            stats["exact_copies"] = stats.get("exact_copies", 0) + info["finished_by_partition"]
            stats["tasks"] = stats.get("tasks", 0) + len(reads)
            stats["raw"] = (o1, o2)
        out = [(o1[k, :int(rec["aln_len"][k])].tobytes().decode(), o2[k, :int(rec["aln_len"][k])].tobytes().decode()) for k in range(len(reads))]
        return out, rec


@pytest.fixture
def on_device(ctx, monkeypatch):
    dev = OnDevice(ctx)
    monkeypatch.setattr(K.E, "align_batch", dev)
    return dev


@pytest.mark.parametrize("L", [250, 151, 203, 256])
def test_gpu_main_diagonal_reads_are_finished_by_the_partition(mats, on_device, monkeypatch, L):
    K.test_main_diagonal_reads_are_finished_by_the_partition(mats, L, monkeypatch)
    assert on_device.batches == 3 and on_device.finished > 0


@pytest.mark.parametrize("L", [151, 203, 250, 256])
def test_gpu_main_diagonal_differences_at_block_edges(mats, on_device, L):
    K.test_main_diagonal_differences_at_block_edges(mats, L)
    assert on_device.finished > 0


def test_gpu_main_diagonal_shortcut_only_where_the_scoring_proves_it(mats, on_device, monkeypatch):
    K.test_main_diagonal_shortcut_only_where_the_scoring_proves_it(mats, monkeypatch)
    assert on_device.batches == 4 and on_device.finished == 14          # (7 + 7 + 0 + 0: the function asserts each)


@pytest.mark.parametrize("kind", ["homopolymer", "marker_in_homopolymer", "dinucleotide", "period3", "period5", "half_repeat", "gentle_scoring"])
def test_gpu_main_diagonal_shortcut_refuses_what_shifts_onto_itself(mats, on_device, kind):
    K.test_main_diagonal_shortcut_refuses_what_shifts_onto_itself(mats, kind)
    # (even a homopolymer has reads the certificate allows: the emulator run of the same function finishes 48 .. 62 of them per kind)
    assert on_device.batches == 1 and on_device.finished > 0, kind


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gpu_main_diagonal_shortcut_on_random_low_complexity_references(mats, on_device, seed):
    K.test_main_diagonal_shortcut_on_random_low_complexity_references(mats, seed)
    assert on_device.batches == 12 and on_device.finished > 0


@pytest.mark.parametrize("scheme", [(1, -1, -1, -1, -2, -1, 0), (1, -1, 0, 0, -3, -1, 0), (2, -3, -1, -1, -5, -2, 1), (5, -4, -2, -1, -8, -1, 0),
                                    (3, -2, -2, -1, -4, -4, 2), (7, -8, -3, 0, -8, -3, 2), (1, -2, -1, 1, -2, -2, 1)])
def test_gpu_main_diagonal_shortcut_under_small_scores_where_ties_are_near(on_device, scheme):
    K.test_main_diagonal_shortcut_under_small_scores_where_ties_are_near(scheme)
    assert on_device.batches == 6 and on_device.finished > 0, scheme


def low_complexity_case(rng, n_reads):
    """a reference made of a repeat unit of period 1 .. 6 with a few point defects + reads with 0 .. 2 changed or MOVED bases"""
    L = int(rng.integers(150, 257))
    period = int(rng.integers(1, 7))
    unit = "".join(rng.choice(list("ACGT"), period))
    if period > 1 and len(set(unit)) == 1:
        unit = unit[:-1] + ("C" if unit[0] != "C" else "G")
    ref = list((unit * L)[:L])
    # (defects: few in a short-period repeat -- the hard cases --, and in one reference out of three so many that no diagonal but the main one fits)
    n_def = int(rng.integers(0, 4)) if rng.random() < 0.67 else int(rng.integers(L // 8, L // 3))
    defects = sorted(set(int(x) for x in rng.integers(3, L - 3, n_def))) if n_def else []
    for q in defects:
        ref[q] = rng.choice([c for c in "ACGT" if c != ref[q]])
    amp = "".join(ref)
    reads = [amp]
    pos = rng.integers(0, L, (n_reads, 2))
    kinds = rng.integers(0, 8, n_reads)
    letters = rng.choice(list("ACGTN"), (n_reads, 2), p=[0.24, 0.24, 0.24, 0.24, 0.04])
    for k in range(n_reads - 1):
        t = list(amp)
        kind = int(kinds[k])
        q, r = int(pos[k, 0]), int(pos[k, 1])
        if kind == 0:
            pass                                                    # a copy
        elif kind <= 2:
            t[q] = letters[k, 0]
        elif kind <= 5:
            t[q] = letters[k, 0]; t[r] = letters[k, 1]
        elif kind == 6 and defects:                                 # a defect moved by one or two: the shifted diagonal explains the read better
            q = defects[r % len(defects)]
            d = (-2, -1, 1, 2)[k & 3]
            if 0 <= q + d < L:
                t[q] = (unit * L)[q]; t[q + d] = amp[q]
        else:                                                       # two neighbours swapped
            if q + 1 < L:
                t[q], t[q + 1] = t[q + 1], t[q]
        reads.append("".join(t))
    return amp, reads


SOAK_TRIALS, SOAK_READS = 52, 2000


@pytest.mark.parametrize("part", [0, 1, 2, 3])
def test_gpu_main_diagonal_soak(mats, ctx, monkeypatch, part):
    """4 x 52 = 208 low-complexity references x 2,000 reads: every alignment of the default chain (partition shortcut ON) equals the oracle's and equals,
    byte for byte (rows up to the last dword the launches write, records), the same batch with C2_NO_EXACT_COPIES=1; four scorings in turn."""
    import oracle
    m = mats["EDNAFULL"]
    rng = np.random.default_rng(66000 + part)
    finished = total = 0
    pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))       # (the oracle is a ctypes call: the GIL is released inside it)
    for trial in range(SOAK_TRIALS):
        amp, reads = low_complexity_case(rng, SOAK_READS)
        L = len(amp)
        go, ge, cut_incentive = [(-20, -2, 1), (-20, -2, 1), (-10, -1, 0), (-30, -3, 2)][trial % 4]
        g = np.zeros(L + 1, dtype=np.int64); g[L // 2 + 1] = cut_incentive
        inc = [L // 2 - 1, L // 2, L // 2 + 1]
        o1, o2, rec, info = device_batch(ctx, reads, [amp], [g], [inc], m, go, ge)
        monkeypatch.setenv("C2_NO_EXACT_COPIES", "1")
        p1, p2, prec, pinfo = device_batch(ctx, reads, [amp], [g], [inc], m, go, ge)
        monkeypatch.delenv("C2_NO_EXACT_COPIES")
        assert pinfo["finished_by_partition"] == 0 and info["classes"] == pinfo["classes"]
        assert rec.tobytes() == prec.tobytes(), (part, trial)
        T = rec["aln_len"].astype(np.int64)
        n4 = (T + 3) // 4 * 4
        live = np.arange(o1.shape[1])[None, :] < n4[:, None]
        assert np.array_equal(o1 * live, p1 * live) and np.array_equal(o2 * live, p2 * live), (part, trial)
        want = list(pool.map(lambda rd: oracle.global_align_raw(rd, amp, m, g, go, ge), reads))
        for k, (status, s1, s2, mt, ln) in enumerate(want):
            assert status == 0 and rec[k]["status"] == 0 and int(T[k]) == ln and int(rec[k]["matches"]) == mt, (part, trial, k, reads[k], amp)
            assert o1[k, :ln].tobytes().decode() == s1 and o2[k, :ln].tobytes().decode() == s2, (part, trial, k, reads[k], amp)
        for k in range(0, len(reads), 97):                               # (the fused classification of a sample; the records of ALL reads equal the shortcut-off batch's)
            K.check_record(rec[k], oracle.find_indels_substitutions(want[k][1], want[k][2], inc), want[k][1], want[k][2])
        finished += info["finished_by_partition"]
        total += len(reads)
    pool.shutdown()
    print("soak part %d: %d of %d reads finished by the partition" % (part, finished, total))
    assert finished > 0.1 * total, (finished, total)
