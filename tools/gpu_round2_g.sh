#!/bin/bash
# Round 2, GPU session: packed second tier; every BASELINE config with all checks
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02g
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -3
timeout 900 python tools/ab/variants.py --rounds 2 x4,C2_NO_PACKED_FILL=1 p8_x2,C2_NO_PACKED_TIER2=1 p8_p4 > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
for cfg in 3 2 4 5; do
    ( time timeout 900 python bench.py --config $cfg --no-cpu-baseline ) > "$OUT/bench_config$cfg.json" 2> "$OUT/bench_config$cfg.err"
    python -c "
import json
d=json.loads([x for x in open('$OUT/bench_config$cfg.json') if x.startswith('{')][-1])
print('config$cfg', round(d['value']/1e6,1), 'M reads/s', round(d['alignments_per_s']/1e6,1), 'M aln/s', d['step_breakdown_ms'], d['config']['tasks_left_after_each_banded_launch'], round(d['roofline']['avg_launch_ms'],2), d['checks'])"
done
