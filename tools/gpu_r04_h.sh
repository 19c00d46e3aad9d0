#!/bin/bash
# Round 4, GPU session: all GPU tests, paired rate after the device gathers, sharded FASTQ leg on 2 gloo ranks sharing the GPU (device route forced: the file is small)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04h
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1800 python -m pytest tests -m gpu -q ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -8
( time timeout 600 python tools/paired_rate.py 2000000 ) > "$OUT/paired_rate_2M.jsonl" 2> "$OUT/paired_rate.err"
cat "$OUT/paired_rate_2M.jsonl"; tail -3 "$OUT/paired_rate.err"
( time C2_BENCH_BACKEND=gloo C2_FQ_INGEST=device timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --reads 4000000 --no-cpu-baseline --extras on --extra-reads 4000000 ) > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> "$OUT/bench_2ranks.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_2ranks_gloo_one_gpu.json') if x.startswith('{')][-1])
    print('2 ranks (gloo, one GPU):', d['n_gpus'], d['ranks_seen'], d['collective_backend'], round(d['value']/1e6,1), d['counts'][0]['reads_aligned_all_gpus'])
    print('sharded e2e:', json.dumps(d['e2e']))
except Exception as ex:
    print('2-rank parse failed', ex)
PY
tail -3 "$OUT/bench_2ranks.err"
