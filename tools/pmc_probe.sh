#!/bin/bash
# Issue / wait / fetch counters of the align kernels over one bench.py launch of 2 M reads (separate --pmc passes):
#   tools/pmc_probe.sh <tag>   -> gpurun_out/pmc_probe_<tag>/summary.json
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_probe_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --check 0 --workers 1 --reads 2000000 --steps 1 --warmup 0"
i=0
for set in "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU"; do
    timeout 200 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -o pmc -- \
        python "$ROOT/bench.py" $COMMON "$@" > "$OUT/pmc_$i.log" 2>&1
    i=$((i + 1))
done
python - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
out = sys.argv[1]
pmc = {}
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace(".kd", "").strip()
            if k.startswith("c2_"):
                e = pmc.setdefault(k, {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
json.dump(pmc, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
for k in sorted(pmc):
    if "diagx_kernel<4>" in k:
        print(k); print(json.dumps(pmc[k], indent=1, sort_keys=True))
PY
