#!/bin/bash
# Round 3, final GPU session: the numbers and profiles that go into profiles/r03/ (see its README.md)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03final
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_default_10M.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_default_10M.json') if x.startswith('{')][-1])
    print('headline', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms'], d['config']['tasks_left_after_each_banded_launch'], d['checks'])
    print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','traffic','avg_launch_ms','kernel')}, 'valu', {k: d['valu'].get(k) for k in ('wave_instr_per_alignment','frac_of_measured_issue','frac_of_simd32_peak','gcups')})
    print('int32', {k: d['int32_chain'][k] for k in ('reads_per_s','ms_per_step','records_equal_the_packed_chain')})
    for k,v in (d['other_configs'] or {}).items():
        if isinstance(v, dict): print(k, {q: v.get(q) for q in ('reads_per_s','alignments_per_s','ms_per_step','chain_equals_full_plane','tasks_left_after_each_banded_launch','error')})
    e=d['e2e']; print('e2e', {q: (e or {}).get(q) for q in ('reads','reads_per_s','stage_seconds','plain_equals_bgzf','error','skipped')}); print('e2e bgzf', (e or {}).get('bgzf'))
    print('dedup_on', d['dedup_on']); print('cpu', {k: d['cpu_baseline'][k] for k in ('value','cores','kind','one_proc_reads_per_s')})
except Exception as ex:
    print('bench parse failed', ex)
PY
tail -3 "$OUT/bench_default.err"
bash tools/profile_round.sh default > "$OUT/profile.log" 2>&1
cp "$ROOT/gpurun_out/prof_default/pmc_summary_default.json" "$OUT/" 2>/dev/null
cp "$ROOT/gpurun_out/prof_default/kernel_stats_default.csv" "$OUT/" 2>/dev/null
find "$ROOT/gpurun_out/prof_default/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/rocprofv3_kernel_stats_default.csv"
head -12 "$OUT/kernel_stats_default.csv"
( time C2_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --reads 2000000 --no-cpu-baseline --extras on --extra-reads 1000000 ) > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> "$OUT/bench_2ranks_gloo_one_gpu.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_2ranks_gloo_one_gpu.json') if x.startswith('{')][-1])
    print('2 ranks (gloo, one GPU):', d['n_gpus'], d['ranks_seen'], d['collective_backend'], round(d['value']/1e6,1), d['counts'][0]['reads_aligned_all_gpus'])
except Exception as ex:
    print('2-rank parse failed', ex)
PY
timeout 600 python tools/shim_call_rate.py > "$OUT/shim_call_rate_20k.json" 2>/dev/null; tail -1 "$OUT/shim_call_rate_20k.json"
