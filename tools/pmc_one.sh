#!/bin/bash
# one --pmc pass over a 2 M-read bench launch:  tools/pmc_one.sh <tag> "<counters>"  -> prints the per-kernel sums
set -u
TAG=$1; SET=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_one_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc" -o pmc -- \
    python "$ROOT/bench.py" --no-cpu-baseline --check 0 --workers 1 --reads 2000000 --steps 1 --warmup 0 > "$OUT/pmc.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
out = sys.argv[1]
pmc = {}
for path in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace(".kd", "").strip()
            if k.startswith("c2_"):
                e = pmc.setdefault(k, {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
json.dump(pmc, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in pmc.items() if "diagx" in k}, indent=1, sort_keys=True))
PY
tail -3 "$OUT/pmc.log"
