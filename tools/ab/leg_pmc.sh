#!/bin/bash
# instruction / cycle counters and kernel times of ONE robustness leg (tools/robust_rate.py) under environment settings:
#   tools/ab/leg_pmc.sh <leg> <reads> "<name>:<ENV=1 ...>" ...      ("base:" = no setting)
# -> gpurun_out/leg_pmc/<leg>_<name>.json (counter sums per kernel) + <leg>_<name>_kernel_stats.csv; prints every c2_align kernel's line
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LEG=$1; N=$2; shift 2
mkdir -p "$ROOT/gpurun_out/leg_pmc"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  OUT=$ROOT/gpurun_out/leg_pmc/${LEG}_$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
      python "$ROOT/tools/robust_rate.py" --reads $N --legs $LEG --no-check --steps 3 > "$OUT/rate.json" 2> "$OUT/rate.err"
  f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$ROOT/gpurun_out/leg_pmc/${LEG}_${name}_kernel_stats.csv"
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    env $envs timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -o pmc -- \
        python "$ROOT/tools/robust_rate.py" --reads $N --legs $LEG --no-check --steps 1 > "$OUT/pmc_$i.log" 2>&1
    i=$((i + 1))
  done
  python - "$OUT" "${LEG}_$name" "$ROOT/gpurun_out/leg_pmc/${LEG}_${name}_kernel_stats.csv" <<'PY'
import csv, glob, json, os, re, sys
out, name, stats = sys.argv[1], sys.argv[2], sys.argv[3]
pmc = {}
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace(".kd", "").strip()
            if k.startswith("c2_"):
                e = pmc.setdefault(k, {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
json.dump(pmc, open(os.path.join(os.path.dirname(out), name + ".json"), "w"), indent=1, sort_keys=True)
# (robust_rate.py launches the chain 1 warm-up + `steps` times: the counters are sums over 2 chains, the stats over 4)
for k, v in sorted(pmc.items()):
    if "c2_align" in k:
        print(name, k, {c: int(x) for c, x in sorted(v.items())})
try:
    for r in csv.DictReader(open(stats)):
        if "c2_" in r["Name"][:40]:
            print(name, "%-56s calls %3s avg %8.3f ms" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e6))
except Exception as ex:
    print("no kernel stats:", ex)
try:
    print(name, open(os.path.join(out, "rate.json")).read().strip().splitlines()[-1][:600])
except Exception as ex:
    print("no rate:", ex)
PY
  rm -rf "$OUT"/pmc_*/ "$OUT/trace"
done
