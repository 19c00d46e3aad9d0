set -u
cd /root/repo
mkdir -p gpurun_out/ab_route2
for spec in "m3:C2_ROUTE_MARGIN=3" "m4:C2_ROUTE_MARGIN=4" "m5:C2_ROUTE_MARGIN=5" "mm4:C2_SCORE_TIER_MAX_MISMATCH=4" "mm10:C2_SCORE_TIER_MAX_MISMATCH=10"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-dedup-leg --workers 16 --check 0 > gpurun_out/ab_route2/bench_$name.json 2> gpurun_out/ab_route2/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads([x for x in open('gpurun_out/ab_route2/bench_$name.json') if x.startswith('{')][-1])
    print('$name', round(d['value']/1e6,1), 'M reads/s chain', round(d['step_breakdown_ms']['align_chain'],2), 'first', round(d['roofline']['avg_launch_ms'],2), d.get('partition'), 'left', d['config']['tasks_left_after_each_banded_launch'])
except Exception as ex:
    print('$name bench parse failed', ex)
PY
done
timeout 200 python tools/phase_profile.py --reads 2000000 2>&1 | tail -1 | tee gpurun_out/ab_route2/phase_profile.json
