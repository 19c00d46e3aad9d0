#!/bin/bash
# Round 3: the 32-bit-add variant of the packed fill on the GPU -- all GPU tests, then the headline with and without it (same box)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03g
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
for v in add32 pkadd; do
    if [ $v = pkadd ]; then export C2_NO_PK_ADD32=1; else unset C2_NO_PK_ADD32; fi
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-dedup-leg > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
    python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_$v.json') if x.startswith('{')][-1])
    print('$v', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms']['align_chain'], round(d['roofline']['avg_launch_ms'],2), d['config']['tasks_left_after_each_banded_launch'], d['checks'].get('chain_equals_full_plane_n'), d['checks'].get('oracle_sample_identical'))
except Exception as ex:
    print('$v parse failed', ex)
PY
done
unset C2_NO_PK_ADD32
for cfg in 2 4 5; do
    timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bench_config$cfg.json" 2> "$OUT/bench_config$cfg.err"
    python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_config$cfg.json') if x.startswith('{')][-1])
    print('config$cfg', round(d['value']/1e6,1), 'M reads/s', round(d['alignments_per_s']/1e6,1), 'M aln/s', d['step_breakdown_ms']['align_chain'], d['config']['tasks_left_after_each_banded_launch'], d['checks'].get('chain_equals_full_plane'))
except Exception as ex:
    print('config$cfg parse failed', ex)
PY
done
