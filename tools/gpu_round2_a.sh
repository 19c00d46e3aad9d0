#!/bin/bash
# Round 2, first GPU session: GPU test suite, smoke, bench.py on every BASELINE config, profiles of the default workload.
# Run on the GPU box:  gpurun -- bash tools/gpu_round2_a.sh      (outputs under gpurun_out/r02a/)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02a
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > "$OUT/gpu_tests.txt" 2>&1
tail -3 "$OUT/gpu_tests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -1 "$OUT/smoke.txt"
for cfg in 3 2 4 5; do
    ( time timeout 900 python bench.py --config $cfg ) > "$OUT/bench_config$cfg.json" 2> "$OUT/bench_config$cfg.err"
    tail -c 600 "$OUT/bench_config$cfg.json"; echo; tail -4 "$OUT/bench_config$cfg.err"
done
timeout 1200 bash tools/profile_round.sh default > "$OUT/profile_round.log" 2>&1
cp -r "$ROOT/gpurun_out/prof_default/kernel_stats_default.csv" "$ROOT/gpurun_out/prof_default/pmc_summary_default.json" "$OUT/" 2>/dev/null
head -12 "$OUT/kernel_stats_default.csv"
