#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02m
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "count or whole or pipeline" > "$OUT/gpu_tests_counts.txt" 2>&1; tail -2 "$OUT/gpu_tests_counts.txt"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --check 0 > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_$i.json') if x.startswith('{')][-1])
print(round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms'])"
done
