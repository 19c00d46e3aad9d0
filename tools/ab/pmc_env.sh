#!/bin/bash
# instruction / cycle counters of the shipped build under environment settings: tools/ab/pmc_env.sh "<name>:<ENV=1 ...>" ...  ("base:" = no setting)
# two --pmc passes per setting over one launch chain of 2 M headline reads -> gpurun_out/ab_pmc_env/<name>.json; prints the first band tier's line
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$ROOT/gpurun_out/ab_pmc_env"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  OUT=$ROOT/gpurun_out/ab_pmc_env/$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    env $envs timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -o pmc -- \
        python "$ROOT/bench.py" --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline --check 0 --no-dedup-leg --workers 1 --no-extras ${BENCH_ARGS:-} > "$OUT/pmc_$i.log" 2>&1
    i=$((i + 1))
  done
  python - "$OUT" "$name" "${PMC_KERNEL:-diagp_kernel<8}" <<'PY'
import csv, glob, json, os, re, sys
out, name, want = sys.argv[1], sys.argv[2], sys.argv[3]
pmc = {}
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace(".kd", "").strip()
            if k.startswith("c2_"):
                e = pmc.setdefault(k, {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
json.dump(pmc, open(os.path.join(os.path.dirname(out), name + ".json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(pmc.items()):
    if want in k:
        print(name, k, {c: int(x) for c, x in sorted(v.items())})
PY
  rm -rf "$OUT"/pmc_*/
done
