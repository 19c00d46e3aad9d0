#!/bin/bash
# A/B of library builds on one box: tools/ab/quick_lib.sh <name> ...   (crispresso2_amd/lib/variants/lib_<name>.so; "shipped" = the regular build)
# per build: bench.py without checks (headline, chain ms, first-tier ms); then the shipped build once more with every check of the default line
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ab_lib
for name in "$@"; do
  lib=$PWD/crispresso2_amd/lib/variants/lib_$name.so
  [ "$name" = "shipped" ] && lib=$PWD/crispresso2_amd/lib/libcrispresso2_amd.so
  for rep in 1 2; do
  C2_AMD_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-dedup-leg --workers 16 --check 0 > gpurun_out/ab_lib/bench_$name.json 2> gpurun_out/ab_lib/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads([x for x in open('gpurun_out/ab_lib/bench_$name.json') if x.startswith('{')][-1])
    print('$name', round(d['value']/1e6,1), 'M reads/s chain', round(d['step_breakdown_ms']['align_chain'],2), 'first', round(d['roofline']['avg_launch_ms'],2), 'count', round(d['step_breakdown_ms']['count_vectors_and_all_reduce'],2))
except Exception as ex:
    print('$name bench parse failed', ex)
PY
  done
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-dedup-leg --workers 16 > gpurun_out/ab_lib/bench_shipped_checked.json 2> gpurun_out/ab_lib/bench_shipped_checked.err
python - <<PY
import json
try:
    d=json.loads([x for x in open('gpurun_out/ab_lib/bench_shipped_checked.json') if x.startswith('{')][-1])
    print('shipped, checked:', round(d['value']/1e6,1), d['checks'])
except Exception as ex:
    print('checked bench parse failed', ex)
PY
