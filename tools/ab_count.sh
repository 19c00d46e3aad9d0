#!/bin/bash
# A/B of count-kernel builds on one box: tools/ab_count.sh "<name>:<flags>" ...   (run through gpurun; builds must exist: tools/build_variant.sh)
# per build: the kernel by kind of alignment (tools/count_kernel_split.py) and bench.py --no-extras (count ms, all checks)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ab_count
mkdir -p "$OUT"
cd "$ROOT"
for name in "$@"; do
    lib=$ROOT/crispresso2_amd/lib/variants/lib_$name.so
    [ "$name" = "shipped" ] && lib=$ROOT/crispresso2_amd/lib/libcrispresso2_amd.so
    echo "== $name"
    C2_AMD_LIB=$lib timeout 300 python tools/count_kernel_split.py 4000000 2> "$OUT/split_$name.err" | tee "$OUT/split_$name.jsonl"
    C2_AMD_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-dedup-leg --workers 16 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
    python - <<PY
import json
try:
    d=json.loads([x for x in open('$OUT/bench_$name.json') if x.startswith('{')][-1])
    print('$name', round(d['value']/1e6,1), 'M reads/s count', d['step_breakdown_ms']['count_vectors_and_all_reduce'], d['checks'].get('chain_equals_full_plane'), d['checks'].get('full_batch_properties_hold'), d['counts'][0])
except Exception as ex:
    print('$name bench parse failed', ex)
PY
done
