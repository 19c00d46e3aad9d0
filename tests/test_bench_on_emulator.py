"""bench.py's plumbing on the wave emulator (tests/bench_on_emulator.py): every workload shape builds, the CPU-baseline legs
run the reference (oracle/_ref) and ALL their alignments are compared with the emulated device's, the chain-vs-full-plane
check covers every task, and the line carries the fields the contract names."""
import pytest

import bench_on_emulator as BE


@pytest.mark.parametrize("config,reads,steps,extra", [(3, 260, 1, []), (2, 200, 3, []), (4, 150, 2, ["--overlap-count"]), (5, 192, 1, [])])
def test_bench_line_on_the_emulator(config, reads, steps, extra):
    out = BE.run_bench(["--config", str(config), "--reads", str(reads), "--steps", str(steps), "--warmup", "0", "--workers", "1",
                        "--cpu-seconds", "1.5", "--check", "40"] + extra)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "checks", "valu"):
        assert key in out, key
    assert out["config"]["baseline_config"] == config and out["config"]["reads_per_gpu_per_step"] == reads
    ck = out["checks"]
    assert ck["all_status_ok"] and ck["oracle_sample_identical"] and ck["full_batch_properties_hold"]
    k = out["config"]["n_amplicons"] if config == 4 else 1
    assert ck["chain_equals_full_plane"] and ck["chain_equals_full_plane_n"] == reads * k
    assert ck["reference_identical"] and ck["reference_compared_n"] == ck["reference_identical_n"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["one_proc_reads_per_s"] > 0 and len(cb["curve"]) >= 2
    assert sum(c["reads_aligned_all_gpus"] for c in out["counts"]) > 0
