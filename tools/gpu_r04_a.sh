#!/bin/bash
# Round 4, first GPU session: the GPU tests, the default bench line, FASTQ -> all tables with a kernel trace.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04a
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests/test_gpu_alleles.py tests/test_whole_run_tables.py -m gpu -q -x ) > "$OUT/gpu_tests_alleles.txt" 2>&1
tail -5 "$OUT/gpu_tests_alleles.txt"
( time timeout 600 python tools/e2e_tables.py 10000000 ) > "$OUT/e2e_tables_10M.jsonl" 2> "$OUT/e2e_tables.err"
cat "$OUT/e2e_tables_10M.jsonl"; tail -3 "$OUT/e2e_tables.err"
( cd /tmp && export TMPDIR=/tmp && C2_WORKERS=1 C2_REPS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_tables" -o trace -- python "$ROOT/tools/e2e_tables.py" 10000000 > "$OUT/trace_tables.log" 2>&1 )
find "$OUT/trace_tables" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/e2e_tables_kernel_stats.csv"
head -30 "$OUT/e2e_tables_kernel_stats.csv"
( time timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_alleles.py ) > "$OUT/gpu_tests.txt" 2>&1
tail -5 "$OUT/gpu_tests.txt"
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_default_10M.json" 2> "$OUT/bench_default.err"
tail -3 "$OUT/bench_default.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_default_10M.json') if x.startswith('{')][-1])
    print('headline', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms'], d['checks'])
    print('config', {k: v for k, v in d['config'].items() if k.startswith(('int32', 'packed', 'e2e', 'other', 'reference', 'chain'))})
    print('with_all_tables', d['e2e'].get('with_all_tables'))
    print('cpu', {k: d['cpu_baseline'][k] for k in ('value','cores','kind','one_proc_reads_per_s','best_procs','cgroup_cpu_quota','long_leg')}, d['cpu_baseline']['curve'])
except Exception as ex:
    print('bench parse failed', ex)
PY
