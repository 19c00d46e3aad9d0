#!/bin/bash
# A/B of the partition's routing on one box (run through gpurun): tools/ab_route.sh "<name>:<ENV=1 ENV2=3 ...>" ...
# per variant: bench.py --no-extras (headline, chain ms, partition classes, all checks of the default line); then the kernel trace of the shipped setting
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ab_route
mkdir -p "$OUT"
cd "$ROOT"
for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    echo "== $name ($envs)"
    env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-dedup-leg --workers 16 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
    python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_$name.json') if x.startswith('{')][-1])
    print('$name', round(d['value']/1e6,1), 'M reads/s chain', round(d['step_breakdown_ms']['align_chain'],2), 'first', round(d['roofline']['avg_launch_ms'],2), d.get('partition'),
          'left', d['config']['tasks_left_after_each_banded_launch'], d['checks'].get('chain_equals_full_plane'), d['checks'].get('chain_equals_full_plane_n'), d['checks'].get('full_batch_properties_hold'))
except Exception as ex:
    print('$name bench parse failed', ex)
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --check 0 --workers 16 --no-extras > "$OUT/prof.log" 2>&1
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -14 "$OUT/kernel_stats.csv" | cut -c1-160
find "$OUT/prof" -type f ! -name "*stats.csv" -delete 2>/dev/null
