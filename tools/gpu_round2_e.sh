#!/bin/bash
# Round 2, fifth GPU session: the packed (int16, 8 per wavefront) first tier
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02e
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -3
timeout 900 python tools/ab/variants.py --rounds 2 base=tools/ab/lib_base.so x4,C2_NO_PACKED_FILL=1 packed > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
( time timeout 900 python bench.py --no-cpu-baseline ) > "$OUT/bench_config3.json" 2> "$OUT/bench_config3.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_config3.json') if x.startswith('{')][-1])
print('config3', d['value']/1e6, 'M reads/s', d['step_breakdown_ms'], d['config']['tasks_left_after_each_banded_launch'], d['roofline']['avg_launch_ms'], d['checks'])"
tail -3 "$OUT/bench_config3.err"
