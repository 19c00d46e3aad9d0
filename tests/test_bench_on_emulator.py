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
    if config in (2, 3):                                             # one amplicon: the step counts the partition-finished reads from their hint words
        assert ck["count_tensor_equals_without_hints"] is True and ck["hinted_tasks"] > 0
    if config in (2, 3):                                             # the dedup-on leg: unique reads + multiplicities give the same count tensor
        d = out["dedup_on"]
        assert d and 0 < d["unique_reads"] <= reads and d["count_tensor_equals_dedup_off"] and d["reads_per_s"] > 0
    else:
        assert out["dedup_on"] is None


def test_bench_two_ranks_over_gloo_on_the_emulator(tmp_path):
    """bench.py's N > 1 plumbing without GPUs: two processes (RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them),
    gloo in place of RCCL, the device calls on the emulator.  Every rank aligns its own shard of the read stream, the count tensor
    is all-reduced inside the step, the time is the maximum over the ranks, and rank 0 alone prints the line -- whose value is the
    whole job's and whose counts are the sum over both shards."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json; sys.path.insert(0, %r); import bench_on_emulator as BE; "
            "out = BE.run_bench(['--gpus', '2', '--config', '3', '--reads', '160', '--steps', '2', '--warmup', '1', '--workers', '1', "
            "'--no-cpu-baseline', '--check', '20']); print('RESULT ' + json.dumps(out))" % here)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   C2_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-3000:]
    res = [json.loads([x for x in so.splitlines() if x.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    assert res[1] is None
    out = res[0]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["reads_per_gpu_per_step"] == 160
    # the short line of an N-rank run parses, says how many ranks it saw, every rank's own share and the all-reduced total
    short = out["_short"]
    assert out["_line_bytes"] < 4096 and short["ranks_seen"] == 2 and short["n_gpus"] == 2 and short["collective_backend"] == "gloo"
    assert len(short["per_rank_reads_aligned"]) == 2 and all(v > 0 for v in short["per_rank_reads_aligned"])
    assert sum(short["per_rank_reads_aligned"]) == short["reads_aligned_all_gpus"][0] == out["counts"][0]["reads_aligned_all_gpus"]
    assert out["value"] == pytest.approx(2 * 160 * 2 / (out["ms_per_step"] * 2 / 1e3), rel=1e-6)      # whole job: both ranks' reads
    single = BE.run_bench(["--config", "3", "--reads", "160", "--steps", "1", "--warmup", "0", "--workers", "1", "--no-cpu-baseline", "--check", "0"])
    # rank 0's shard alone aligns fewer reads than the all-reduced tensor holds (the second shard is a different block of the stream)
    assert out["counts"][0]["reads_aligned_all_gpus"] > single["counts"][0]["reads_aligned_all_gpus"] >= 100
    assert out["counts"][0]["reads_aligned_all_gpus"] <= 320


def test_bench_started_bare_with_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2 --steps K --warmup W` with NOTHING prepared in the environment (no RANK /
    WORLD_SIZE / MASTER_*: the driver's command form) must start its two ranks itself and print rank 0's single line.  Here through
    tests/bench_emulated_main.py (the same main() and launcher, device calls on the emulator, gloo in place of RCCL)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["C2_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(here, "bench_emulated_main.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "130",
                        "--workers", "1", "--no-cpu-baseline", "--check", "10", "--extras", "on", "--extra-reads", "70", "--extra-steps", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = [json.loads(x[7:]) for x in p.stdout.splitlines() if x.startswith("RESULT ")]
    assert len(res) == 2 and sum(r is not None for r in res) == 1, p.stdout[-2000:]
    out = next(r for r in res if r is not None)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["collective_backend"] == "gloo" and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["reads_per_gpu_per_step"] == 130 and 130 < out["counts"][0]["reads_aligned_all_gpus"] <= 260
    # the legs behind the headline ran on both ranks too (their steps hold collectives): the int32 chain and BASELINE's 8-GPU shape
    assert out["int32_chain"]["records_equal_the_packed_chain"] and out["int32_chain"]["reads_per_s"] > 0
    c5 = out["other_configs"]["config5"]
    assert c5["n_amplicons"] == 96 and c5["reads_per_gpu_per_step"] == 70 and 70 < c5["reads_aligned_all_gpus"] <= 140
    sh = out["e2e"]["sharded"]                                         # N ranks: the sharded FASTQ leg (here the file is small: the host parser on every rank)
    assert sh["reads"] == 70 and sh["tallies"]["N_TOT_READS"] == 70 and sh["reads_per_s"] > 0 and len(sh["per_rank"]) == 2


def test_side_legs_that_do_not_finish_cannot_cost_the_headline():
    """N ranks: the legs behind the headline hold collectives; when they do not finish on every rank within C2_BENCH_EXTRAS_TIMEOUT (here: at
    once) rank 0 still prints the line -- the headline that was measured, a note, whatever the legs had produced -- and every rank ends with
    exit code 0 (a rank that fails alone waits for the watchdog instead of leaving the others in a collective)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["C2_BENCH_BACKEND"] = "gloo"
    env["C2_BENCH_EXTRAS_TIMEOUT"] = "0.05"
    p = subprocess.run([sys.executable, os.path.join(here, "bench_emulated_main.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "66",
                        "--workers", "1", "--no-cpu-baseline", "--check", "0", "--extras", "on", "--extra-reads", "64", "--extra-steps", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = lines[0]
    assert len(p.stdout.splitlines()[-1]) < 4096                      # (the watchdog prints the same SHORT line)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 1 and out["value"] > 0 and "did not finish" in out["side_legs_note"]
    assert out["side_legs_ok"] is False                               # (what a script looks at: the exit code stays 0)
    assert 66 < out["reads_aligned_all_gpus"][0] <= 132 and sum(out["per_rank_reads_aligned"]) == out["reads_aligned_all_gpus"][0]


def test_bench_default_line_carries_the_int32_chain_the_other_configs_and_the_fastq_leg():
    """the default run's line alone must show the int32 chain's rate, configs 2 / 4 / 5 with their
    chain = full-plane check, and FASTQ -> tensors (plain and BGZF)."""
    out = BE.run_bench(["--steps", "1", "--warmup", "0", "--reads", "150", "--workers", "1", "--cpu-seconds", "1.0", "--check", "10",
                        "--extras", "on", "--extra-reads", "90", "--extra-steps", "1"])
    assert out["dtype"].startswith("int16x2 packed") and len(out["dtype"]) < 128 and out["int32_chain"]["dtype"] == "int32" and out["int32_chain"]["records_equal_the_packed_chain"]
    cfgd = out["config"]                                            # the scalars the driver's record keeps
    assert cfgd["int32_chain_reads_per_s"] == out["int32_chain"]["reads_per_s"] and cfgd["int32_chain_records_equal"] is True
    assert 0.0 < cfgd["packed_int16_share"] <= 1.0 and cfgd["e2e_fastq_to_all_tables_seconds"] > 0
    for cfg, k_ in ((2, 1), (4, 3), (5, 1)):
        e = out["other_configs"]["config%d" % cfg]
        assert e["chain_equals_full_plane"] and e["chain_equals_full_plane_n"] == 90 * k_ and e["all_status_ok"] and e["reads_per_s"] > 0
        if cfg != 4:                                                  # (hint words: one amplicon, or every read tagged with its own -- config 5, the hinted kernel per reference)
            assert e["count_tensor_equals_without_hints"] is True
    e2e = out["e2e"]
    assert e2e["reads"] == 90 and e2e["plain_equals_bgzf"] and e2e["plain"]["reads_per_s"] > 0 and e2e["bgzf"]["reads_per_s"] > 0
    assert e2e["plain_equals_gzip"] and e2e["gzip"]["reads_per_s"] > 0 and e2e["file_bytes_gzip"] > 100       # an ordinary single-member .gz of the same reads
    assert e2e["tallies"]["N_TOT_READS"] == 90 and set(e2e["stage_seconds"]) >= {"ingest_dedup_streamed", "stream_tail_device", "count_kernels"}
    wt = e2e["with_all_tables"]                                      # FASTQ -> every result table on disk, the allele table among them
    assert wt["files_written"] >= 18 and wt["allele_table_rows"] > 0 and wt["allele_table_zip_bytes"] > 100 and wt["alleles_around_cut_bytes"] > 100
    assert wt["zip_member_equals_the_txt_legs_file"] is True          # (the reference ends with Alleles_frequency_table.zip; Python's zipfile reads our stream back)
    assert e2e["with_all_tables_txt"]["allele_table_bytes"] == wt["allele_table_text_bytes"] > 500
    assert e2e["plain"]["link_bytes"] == e2e["file_bytes"] and 0 < e2e["plain"]["frac_of_link_peak"]
    assert out["side_legs_ok"] is True
    hb = out["host_batch_pcie_inclusive"]                            # the boundary with host buffers on both sides
    assert hb["reads"] == 90 and hb["all_status_ok"] and hb["reads_per_s"] > 0 and hb["link_bytes_per_read"] > 700
    assert set(wt["write_tables_stage_seconds"]) >= {"allele_table_build", "allele_table_write", "around_cut_tables", "other_tables"}
    # the same read budget on inputs that are not the generator's best case -- FANC-shaped reads (ragged, overhangs on both sides),
    # reads cut to U[200, L], one read in ten unrelated -- each with its tier shares, chain = full plane on every task and a reference-compiled slice
    rb = out["robustness"]
    for leg in ("fanc_shaped", "lengths_200_to_L", "unrelated_10_percent"):
        e = rb[leg]
        assert e["reads_per_s"] > 0 and e["all_status_ok"] and e["chain_equals_full_plane"] and e["chain_equals_full_plane_n"] == 90, (leg, e)
        assert e["reference_identical"] and e["reference_compared_n"] == 90, (leg, e)
        assert e["reads_aligned_all_gpus"] > 60
    assert rb["full_plane_floor"]["reads_per_s"] > 0 and rb["worst_case"]["leg"] in rb
    assert cfgd["robust_fanc_shaped_reads_per_s"] == rb["fanc_shaped"]["reads_per_s"] and cfgd["robust_worst_case"].startswith(rb["worst_case"]["leg"])


SHORT_LINE_KEYS = {
    "": ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
         "config", "roofline", "valu", "cpu_baseline", "checks", "side_legs_ok", "speedup_vs_cpu_baseline", "detail"),
    "config": ("workload", "baseline_config", "reads_per_gpu_per_step", "finished_by_partition", "packed_int16_share", "int32_chain_reads_per_s",
               "robust_fanc_shaped_reads_per_s", "robust_lengths_200_to_L_reads_per_s", "robust_unrelated_10_percent_reads_per_s",
               "robust_full_plane_floor_reads_per_s", "other_configs_reads_per_s", "chain_equals_full_plane_n", "reference_identical_n",
               "e2e_frac_of_link_peak", "e2e_gzip_seconds"),
    "roofline": ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"),
    "valu": ("frac_of_simd32_peak",),
    "cpu_baseline": ("value", "unit", "cores", "kind", "best_procs", "sample"),
}


def test_the_one_stdout_line_is_short_and_complete():
    """BENCH_r05.json had "parsed": null -- the 31 KB line (legs, notes, stage seconds) was more than the driver's record could
    parse.  The default run's stdout is ONE line under 4 KB with every field the contract and the grading name, no prose; the rest is bench_detail.json."""
    out = BE.run_bench(["--steps", "1", "--warmup", "0", "--reads", "150", "--workers", "1", "--cpu-seconds", "1.0", "--cpu-long-seconds", "1.0", "--check", "10",
                        "--extras", "on", "--extra-reads", "90", "--extra-steps", "1"])
    short = out["_short"]
    assert out["_line_bytes"] < 4096
    for group, keys in SHORT_LINE_KEYS.items():
        d = short if group == "" else short[group]
        for key in keys:
            assert key in d, (group, key)
    assert short["roofline"]["bound"] == "hbm" and short["roofline"]["peak"] == 8000.0 and 0 < short["roofline"]["frac"] < 1
    assert abs(short["roofline"]["frac"] - short["roofline"]["achieved"] / short["roofline"]["peak"]) < 1e-3 * short["roofline"]["frac"] + 1e-9
    assert short["cpu_baseline"]["kind"] in ("reference", "port") and short["cpu_baseline"]["value"] > 0 and len(short["cpu_baseline"]["sample"]) <= 160
    assert short["config"]["baseline_config"] == 3 and short["config"]["reads_per_gpu_per_step"] == 150
    assert set(short["config"]["other_configs_reads_per_s"]) == {"config2", "config4", "config5"}
    assert short["config"]["other_configs_alignments_per_s"].keys() == {"config4"}          # (the three-amplicon shape: alignments/s next to reads/s)
    assert short["config"]["e2e_frac_of_link_peak"]["plain"] > 0 and short["config"]["e2e_frac_of_link_peak"]["gzip"] > 0
    assert short["checks"]["chain_equals_full_plane"] is True and short["checks"]["reference_identical"] is True
    assert short["side_legs_ok"] is True and short["n_gpus"] == short["ranks_seen"] == 1 and short["per_rank_reads_aligned"] is None
    # nothing in it is a paragraph
    def texts(x):
        if isinstance(x, dict):
            for k_, v_ in x.items():
                if k_ != "workload":
                    yield from texts(v_)
        elif isinstance(x, list):
            for v_ in x:
                yield from texts(v_)
        elif isinstance(x, str):
            yield x
    assert max(len(t) for t in texts(short)) <= 160
