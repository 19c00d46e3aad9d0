#!/bin/bash
# Round 2, fourth GPU session: vectorized read staging / gap-free emit (tree build) and the alignbit pointer-bit variant
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02d
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed" "$OUT/gpu_tests.txt" | tail -2
C2_AMD_LIB=$ROOT/tools/ab/lib_alignbit.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 python tools/ab/variants.py --rounds 2 base=tools/ab/lib_base.so new alignbit=tools/ab/lib_alignbit.so > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
cd /tmp && export TMPDIR=/tmp
pmc() {
    tag=$1; shift
    env "$@" timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
        python "$ROOT/bench.py" --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline --check 0 --workers 1 > "$OUT/pmc_$tag.log" 2>&1
    python - "$OUT/pmc_$tag" "$tag" <<'PY'
import csv, glob, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
pmc = {}
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace(".kd", "").strip()
        if "diagx_kernel<4>" in k:
            pmc[row["Counter_Name"]] = pmc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print(tag, {k: round(v / 2e6, 1) for k, v in sorted(pmc.items())})
PY
}
pmc whole C2_X=1
pmc noepi C2_DEBUG_SKIP_EPILOGUE=1
cd "$ROOT"
( time timeout 900 python bench.py --config 4 --no-cpu-baseline ) > "$OUT/bench_config4.json" 2> "$OUT/bench_config4.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_config4.json') if x.startswith('{')][-1])
print('config4', d['alignments_per_s']/1e6, 'M aln/s', d['step_breakdown_ms'], d['checks'])"
