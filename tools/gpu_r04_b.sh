#!/bin/bash
# Round 4, A/B of the count kernel's staging (C2_CNT_STAGE alignments per wavefront copied to LDS by global_load_lds before they are walked):
# every variant of crispresso2_amd/lib/variants/ on tools/count_kernel_split.py (per kind of alignment) and on bench.py's default step.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04f
mkdir -p "$OUT"
cd "$ROOT"
for v in r03 ev_s1o8 ev_s1o6 ev_s2o6 ev_s1o5; do
  LIB=$ROOT/crispresso2_amd/lib/variants/lib_$v.so
  echo "== $v" | tee -a "$OUT/count_split.txt"
  C2_AMD_LIB=$LIB timeout 300 python tools/count_kernel_split.py 4000000 2>&1 | grep -E '"kind"' | tee -a "$OUT/count_split.txt"
done
for v in ev_s1o8 ev_s1o6 ev_s2o6 ev_s1o5; do
  LIB=$ROOT/crispresso2_amd/lib/variants/lib_$v.so
  C2_AMD_LIB=$LIB timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 300 --no-dedup-leg --no-extras --no-full-plane-check > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
  python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_$v.json') if x.startswith('{')][-1])
    print('$v', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms']['count_vectors_and_all_reduce'], d['checks'].get('oracle_sample_identical'), d['counts'][0])
except Exception as ex:
    print('$v bench parse failed', ex)
PY
done
