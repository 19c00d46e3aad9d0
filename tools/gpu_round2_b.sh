#!/bin/bash
# Round 2, second GPU session: full GPU test suite; same-box variants of the dominant kernel (gap-free predicate vs the build
# before it; measurement knobs that switch off the epilogue / half of the fill); config 4 with the new count / select kernels.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02b
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/gpu_tests.txt" 2>&1
tail -5 "$OUT/gpu_tests.txt"
timeout 900 python tools/ab/variants.py --rounds 2 base=tools/ab/lib_base.so new new_noepi,C2_DEBUG_SKIP_EPILOGUE=1 \
    new_noepi_halffill,C2_DEBUG_SKIP_EPILOGUE=1,C2_DEBUG_HALF_FILL=1 new_nostrings,C2_DEBUG_SKIP_STRINGS=1 > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
( time timeout 900 python bench.py --config 4 --no-cpu-baseline ) > "$OUT/bench_config4.json" 2> "$OUT/bench_config4.err"
python -c "
import json,sys
d=json.loads([x for x in open('$OUT/bench_config4.json') if x.startswith('{')][-1])
print('config4', d['alignments_per_s']/1e6, 'M aln/s', d['step_breakdown_ms'], d['checks'])"
