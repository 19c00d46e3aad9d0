#!/bin/bash
# Round 3: profiles for profiles/r03 (kernel trace + PMC passes of the current build, counter calibration), the consensus call, GPU tests
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03h
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
timeout 600 python tools/consensus_rate.py > "$OUT/consensus_rate_500k.txt" 2> "$OUT/consensus.err"; tail -1 "$OUT/consensus_rate_500k.txt"; tail -2 "$OUT/consensus.err"
bash tools/profile_round.sh default > "$OUT/profile.log" 2>&1
cp "$ROOT/gpurun_out/prof_default/pmc_summary_default.json" "$OUT/" 2>/dev/null
cp "$ROOT/gpurun_out/prof_default/kernel_stats_default.csv" "$OUT/" 2>/dev/null
find "$ROOT/gpurun_out/prof_default/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/rocprofv3_kernel_stats_default.csv"
cat "$OUT/kernel_stats_default.csv"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib/fetch" -o pmc -- python "$ROOT/tools/pmc_calibrate.py" > "$OUT/calib_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib/write" -o pmc -- python "$ROOT/tools/pmc_calibrate.py" > "$OUT/calib_write.log" 2>&1
cd "$ROOT"
python tools/pmc_calibrate_summary.py "$OUT/calib" | tee "$OUT/pmc_calibration.json"
