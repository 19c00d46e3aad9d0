#!/bin/bash
# A/B of environment settings with the shipped build: tools/ab/quick_env.sh "<name>:<ENV=1 ...>" ...   (bench.py without checks, twice each)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ab_env
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  for rep in 1 2; do
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-dedup-leg --workers 16 --check 0 > gpurun_out/ab_env/bench_$name.json 2> gpurun_out/ab_env/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads([x for x in open('gpurun_out/ab_env/bench_$name.json') if x.startswith('{')][-1])
    print('$name', round(d['value']/1e6,1), 'M reads/s chain', round(d['step_breakdown_ms']['align_chain'],2), 'first', round(d['roofline']['avg_launch_ms'],2), d.get('partition'))
except Exception as ex:
    print('$name bench parse failed', ex)
PY
  done
done
