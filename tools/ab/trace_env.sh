#!/bin/bash
# per-kernel times of the headline chain under environment settings: tools/ab/trace_env.sh "<name>:<ENV=1 ...>" ...
# rocprofv3 --kernel-trace --stats of bench.py --no-extras (10 M reads, 3 steps) -> gpurun_out/ab_trace/<name>_kernel_stats.csv; prints the c2_ kernels' lines and the partition's classes
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$ROOT/gpurun_out/ab_trace"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  OUT=$ROOT/gpurun_out/ab_trace/$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  env $envs C2_BENCH_DETAIL=$OUT/detail.json timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
      python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --check 0 --no-dedup-leg --workers 1 --no-extras ${BENCH_ARGS:-} > "$OUT/bench.log" 2>&1
  f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$ROOT/gpurun_out/ab_trace/${name}_kernel_stats.csv"
  python - "$ROOT/gpurun_out/ab_trace/${name}_kernel_stats.csv" "$OUT/detail.json" "$name" <<'PY'
import csv, json, sys
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        if r["Name"].startswith("c2_") or "c2_" in r["Name"][:40]:
            print(sys.argv[3], "%-60s calls %4s  avg %9.3f ms  total %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
except Exception as ex:
    print("no kernel stats:", ex)
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[3], "value %.1f M reads/s" % (d["value"] / 1e6), "chain ms", d["step_breakdown_ms"], "partition", d.get("partition"), "tiers", d["config"]["tasks_left_after_each_banded_launch"], "score stage", d["config"]["score_only_stage_tasks"], d["config"]["score_only_stage_finished"])
except Exception as ex:
    print("no detail:", ex)
PY
  rm -rf "$OUT/trace"
done
