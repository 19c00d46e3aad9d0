#!/bin/bash
# A round's closing GPU session (run through gpurun): the numbers and profiles that go into profiles/<round>/ (its README names them).
#   tools/gpu_session.sh <round, e.g. r04> [quick]
# writes gpurun_out/<round>final/: all GPU tests; the driver's bench command; rocprofv3 kernel trace + PMC passes (tools/profile_round.sh);
# FASTQ -> all result tables (tools/e2e_tables.py); paired FASTQ rate; two gloo ranks sharing the GPU with the sharded FASTQ leg; the
# unchanged caller's call rate; the end-to-end .gz leg alone.  `quick`: tests + bench only.
set -u
ROUND=${1:-r05}
QUICK=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${ROUND}final
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1800 python -m pytest tests -m gpu -q ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
if [ -z "$QUICK" ]; then
  # the profile FIRST: bench.py reads profiles/<round>/pmc_summary_default.json for roofline.traffic and valu.* -- the line below and the PMC file then come
  # from one session at one commit (the kept bench copy predated the kept PMC file).  Copy the same file into profiles/<round>/ when committing.
  bash tools/profile_round.sh default > "$OUT/profile.log" 2>&1
  cp "$ROOT/gpurun_out/prof_default/pmc_summary_default.json" "$OUT/" 2>/dev/null
  cp "$ROOT/gpurun_out/prof_default/kernel_stats_default.csv" "$OUT/" 2>/dev/null
  find "$ROOT/gpurun_out/prof_default/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/rocprofv3_kernel_stats_default.csv"
  mkdir -p "$ROOT/profiles/$ROUND" && cp "$OUT/pmc_summary_default.json" "$ROOT/profiles/$ROUND/" 2>/dev/null
  head -14 "$OUT/kernel_stats_default.csv"
fi
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_default_10M.json" 2> "$OUT/bench_default.err"
cp "$ROOT/bench_detail.json" "$OUT/bench_default_10M_detail.json" 2>/dev/null
python - <<PY
import json
try:
    line=[x for x in open('$OUT/bench_default_10M.json') if x.startswith('{')][-1]
    d=json.loads(line)
    print('line bytes', len(line), 'headline', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms'], d['checks'])
    print('config', d['config'])
    print('roofline', d['roofline']); print('valu', d['valu']); print('cpu', d['cpu_baseline'])
except Exception as ex:
    print('bench parse failed', ex)
PY
tail -3 "$OUT/bench_default.err"
[ -n "$QUICK" ] && exit 0
( time timeout 900 python tools/e2e_tables.py 10000000 ) > "$OUT/e2e_tables_10M.jsonl" 2> "$OUT/e2e_tables.err"; tail -2 "$OUT/e2e_tables_10M.jsonl"
( time timeout 900 python tools/paired_rate.py 2000000 ) > "$OUT/paired_rate_2M.jsonl" 2> "$OUT/paired_rate.err"; tail -3 "$OUT/paired_rate_2M.jsonl"
( time C2_BENCH_BACKEND=gloo C2_FQ_INGEST=device C2_BENCH_DETAIL=$OUT/bench_2ranks_detail.json timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --reads 4000000 --no-cpu-baseline --extras on --extra-reads 4000000 ) \
    > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> "$OUT/bench_2ranks.err"
python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_2ranks_gloo_one_gpu.json') if x.startswith('{')][-1])
    print('2 ranks (gloo, one GPU):', d['n_gpus'], d['ranks_seen'], d['collective_backend'], round(d['value']/1e6,1), d['per_rank_reads_aligned'], d['reads_aligned_all_gpus'])
except Exception as ex:
    print('2-rank parse failed', ex)
PY
timeout 600 python tools/shim_call_rate.py --procs 16 > "$OUT/shim_call_rate_20k.json" 2>/dev/null; tail -1 "$OUT/shim_call_rate_20k.json" | cut -c1-1500
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.txt" 2>&1; tail -1 "$OUT/smoke.txt"
( timeout 300 python tools/gz_leg.py --reads 8000000 ) > "$OUT/gz_leg_8M.json" 2> "$OUT/gz_leg.err"; tail -c 600 "$OUT/gz_leg_8M.json"
