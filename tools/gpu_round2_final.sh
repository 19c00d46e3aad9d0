#!/bin/bash
# Round 2, final GPU session: the numbers and profiles that go into profiles/r02/ (see its README.md)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02final
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
bash tools/profile_round.sh default > "$OUT/profile.log" 2>&1
mkdir -p "$ROOT/profiles/r02"
cp "$ROOT/gpurun_out/prof_default/pmc_summary_default.json" "$ROOT/profiles/r02/pmc_summary_default.json"
cp "$ROOT/gpurun_out/prof_default/kernel_stats_default.csv" "$ROOT/profiles/r02/kernel_stats_default.csv" 2>/dev/null
cd "$ROOT"
( time timeout 1200 python bench.py ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 1500 "$OUT/bench_default.json" | head -c 400; echo
for cfg in 2 4 5; do
    ( time timeout 1200 python bench.py --config $cfg --cpu-seconds 8 ) > "$OUT/bench_config$cfg.json" 2> "$OUT/bench_config$cfg.err"
done
for cfg in default config2 config4 config5; do
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_$cfg.json') if x.startswith('{')][-1])
print('$cfg', round(d['value']/1e6,1), 'M reads/s', round(d['alignments_per_s']/1e6,1), 'M aln/s', d['step_breakdown_ms']['align_chain'], d['config']['tasks_left_after_each_banded_launch'], round(d['roofline']['avg_launch_ms'],2), d['checks'], (d.get('cpu_baseline') or {}).get('value'))"
done
timeout 600 python bench.py --no-cpu-baseline --check 0 --overlap-count > "$OUT/bench_overlap_count.json" 2> "$OUT/bench_overlap_count.err"
timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"
timeout 600 python tools/e2e_rate.py --reads 2000000 --gz --repeat 2 > "$OUT/e2e_rate_2M_gz.json" 2> "$OUT/e2e_gz.err"; tail -1 "$OUT/e2e_rate_2M_gz.json"
timeout 600 python tools/host_path_rate.py > "$OUT/host_path_rate_2M.json" 2> "$OUT/host_path.err"; tail -1 "$OUT/host_path_rate_2M.json"
