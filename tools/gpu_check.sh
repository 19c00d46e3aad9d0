#!/bin/bash
# quick GPU check of a build: parity + soak tests, then bench.py (no CPU leg) and the end-to-end rate
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/check
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/gpu_tests.txt" 2>&1; grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -3
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_1.json" 2> "$OUT/bench_1.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_1.json') if x.startswith('{')][-1])
print(round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms']['align_chain'], d['step_breakdown_ms']['count_vectors_and_all_reduce'], round(d['roofline']['avg_launch_ms'],2), d['config']['tasks_left_after_each_banded_launch'], d['checks'])"
timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"
