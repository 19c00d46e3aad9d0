#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02n
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -m gpu -q -x > "$OUT/gpu_tests.txt" 2>&1; tail -2 "$OUT/gpu_tests.txt"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_$i.json') if x.startswith('{')][-1])
print(round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms']['align_chain'], d['step_breakdown_ms']['count_vectors_and_all_reduce'], round(d['roofline']['avg_launch_ms'],2), d['checks'])"
done
