"""The GPU parity tests of the per-read route and of the count pipeline, run on the CPU: the test FUNCTIONS of tests/test_gpu_parity.py /
test_variant_io.py are called as they are, with the product's device entry points redirected to the wave emulator
(tests/pipeline_on_emulator.py).  What runs on the MI355X at round end therefore already ran here against the same goldens --
same product code above the three device calls, same kernel source below them."""
import pytest

from helpers import matrices
from pipeline_on_emulator import EmulatedContext, emulated_device


@pytest.fixture(scope="module")
def mats():
    return matrices()


def test_get_new_variant_objects_and_process_fastq_on_the_emulator(mats, tmp_path):
    import test_gpu_parity as G
    with emulated_device():
        G.test_get_new_variant_objects_vs_reference_function(mats, EmulatedContext())
        G.test_process_fastq_equivalent(mats, EmulatedContext(), tmp_path)


def test_variant_files_annotated_fastq_and_sam_on_the_emulator(mats, tmp_path, monkeypatch):
    import test_gpu_parity as G
    import test_variant_io as VIO
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    with emulated_device():
        G.test_variant_files_and_annotated_fastq_equal_the_reference_text(mats, EmulatedContext(), tmp_path / "a")
        VIO.test_bam_output_sam_text_from_the_device_route(tmp_path / "b")


def test_count_pipeline_gpu_tests_on_the_emulator(mats, tmp_path):
    import test_gpu_parity as G
    with emulated_device():
        G.test_whole_run_fanc_fastq_equals_the_reference_result_tables(mats, EmulatedContext(), tmp_path)
        G.test_pipeline_equals_per_read_path_plus_reference_aggregation(mats, EmulatedContext())


def test_whole_run_gpu_tests_on_the_emulator(tmp_path):
    import test_whole_run_tables as W
    for k, fn in enumerate((W.test_tables_from_the_device_pipeline_equal_every_file_of_the_reference_run,
                            W.test_params_run_tables_from_the_device_pipeline,
                            W.test_params_run_from_the_unfiltered_fastq_with_the_fused_read_filter)):
        d = tmp_path / str(k)
        d.mkdir()
        with emulated_device():
            fn(d)


def test_paired_read_gpu_tests_on_the_emulator(mats, tmp_path):
    """consensus kernel + paired per-pair dicts (reference unit-test calls, FANC pairs, recorded dicts) and the paired FASTQ route
    (TSV files, second pass, final cache) -- tests/test_gpu_parity.py's functions on the emulator."""
    import test_gpu_parity as G
    with emulated_device():
        G.test_paired_consensus_and_variants_vs_reference_functions(mats, EmulatedContext())
        G.test_paired_fastq_files_equal_the_reference_run(mats, EmulatedContext(), tmp_path)


def test_main_diagonal_gpu_twins_on_the_emulator(mats, monkeypatch):
    """tests/test_gpu_main_diagonal.py's own plumbing (the stand-in for the emulator's launch, its counters, the soak's generator and comparisons) with the
    product's launch replaced by the emulator's: what runs on the MI355X at round end ran here first, at a reduced size."""
    import emu_driver as E
    import test_gpu_main_diagonal as G
    real = E.align_batch

    def emulated_batch(ctx, reads, refs, gap_incentives, includes, matrix, gap_open, gap_extend):
        st = {}
        res, rec = real(reads, refs, gap_incentives, includes, matrix, gap_open, gap_extend, band_lanes=-87, stats=st)
        import os
        return st["raw"][0], st["raw"][1], rec, {"finished_by_partition": st["exact_copies"], "classes": st["classes"]}
    monkeypatch.setattr(G, "device_batch", emulated_batch)
    monkeypatch.setattr(G, "SOAK_TRIALS", 8)
    monkeypatch.setattr(G, "SOAK_READS", 250)
    G.test_gpu_main_diagonal_soak(mats, None, monkeypatch, 0)
    seen = {}
    for kind in ("homopolymer", "marker_in_homopolymer", "dinucleotide", "period3", "period5", "half_repeat", "gentle_scoring"):
        dev = G.OnDevice(None)
        monkeypatch.setattr(E, "align_batch", dev)
        G.test_gpu_main_diagonal_shortcut_refuses_what_shifts_onto_itself(mats, dev, kind)
        seen[kind] = dev.finished
    for scheme in [(1, -1, -1, -1, -2, -1, 0), (1, -1, 0, 0, -3, -1, 0), (2, -3, -1, -1, -5, -2, 1), (5, -4, -2, -1, -8, -1, 0),
                   (3, -2, -2, -1, -4, -4, 2), (7, -8, -3, 0, -8, -3, 2), (1, -2, -1, 1, -2, -2, 1)]:
        dev = G.OnDevice(None)
        monkeypatch.setattr(E, "align_batch", dev)
        G.test_gpu_main_diagonal_shortcut_under_small_scores_where_ties_are_near(dev, scheme)
        seen[scheme] = dev.finished
    dev = G.OnDevice(None)
    monkeypatch.setattr(E, "align_batch", dev)
    G.test_gpu_main_diagonal_shortcut_only_where_the_scoring_proves_it(mats, dev, monkeypatch)
    dev = G.OnDevice(None)
    monkeypatch.setattr(E, "align_batch", dev)
    G.test_gpu_main_diagonal_differences_at_block_edges(mats, dev, 203)
    print("finished by the partition per case:", seen)
