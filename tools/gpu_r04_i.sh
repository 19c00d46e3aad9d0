#!/bin/bash
# Round 4, GPU session: paired FASTQ with the pair keys built on the device -- its tests and its rate
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04i
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 900 python -m pytest tests/test_paired_device.py tests/test_gpu_fastq_device.py -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -8
( time timeout 900 python tools/paired_rate.py 2000000 ) > "$OUT/paired_rate_2M.jsonl" 2> "$OUT/paired_rate.err"
cat "$OUT/paired_rate_2M.jsonl"; tail -3 "$OUT/paired_rate.err"
