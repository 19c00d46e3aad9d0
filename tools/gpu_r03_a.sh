#!/bin/bash
# Round 3, first GPU session: the GPU tests (incl. the packed-fill soak at its limits), the driver's bench command with the new legs,
# and the multi-rank plumbing of bench.py on this ONE GPU (two gloo ranks sharing it; RCCL needs one GPU per rank).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03a
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_default.json') if x.startswith('{')][-1])
    print('headline', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms'], d['config']['tasks_left_after_each_banded_launch'], d['checks'])
    print('int32', d['int32_chain'])
    for k,v in (d['other_configs'] or {}).items():
        if isinstance(v, dict): print(k, {q: v.get(q) for q in ('reads_per_s','alignments_per_s','ms_per_step','chain_equals_full_plane','tasks_left_after_each_banded_launch','error')})
    e=d['e2e']; print('e2e', {q: (e or {}).get(q) for q in ('reads','reads_per_s','stage_seconds','plain_equals_bgzf','error','skipped')}, (e or {}).get('bgzf'))
except Exception as ex:
    print('bench parse failed', ex)
PY
tail -5 "$OUT/bench_default.err"
( time C2_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --reads 2000000 --cpu-seconds 2 --extras on --extra-reads 1000000 ) > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> "$OUT/bench_2ranks_gloo_one_gpu.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_2ranks_gloo_one_gpu.json') if x.startswith('{')][-1])
    print('2 ranks (gloo, one GPU):', d['n_gpus'], d['ranks_seen'], d['collective_backend'], round(d['value']/1e6,1), d['counts'], {k:(v.get('reads_per_s') if isinstance(v,dict) else v) for k,v in d['other_configs'].items()})
except Exception as ex:
    print('2-rank parse failed', ex)
PY
tail -5 "$OUT/bench_2ranks_gloo_one_gpu.err"
