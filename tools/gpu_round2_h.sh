#!/bin/bash
# Round 2, GPU session: HBM pointer plane + legacy count route + count kernel dword walk
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02h
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
for cfg in 3; do
    ( time timeout 900 python bench.py --config $cfg --no-cpu-baseline ) > "$OUT/bench_config$cfg.json" 2> "$OUT/bench_config$cfg.err"
    python -c "
import json
d=json.loads([x for x in open('$OUT/bench_config$cfg.json') if x.startswith('{')][-1])
print('config$cfg', round(d['value']/1e6,1), 'M reads/s', round(d['alignments_per_s']/1e6,1), 'M aln/s', d['step_breakdown_ms'], d['config']['tasks_left_after_each_banded_launch'], round(d['roofline']['avg_launch_ms'],2), d['checks'])"
done
timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"
