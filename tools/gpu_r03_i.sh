#!/bin/bash
# Round 3: the device ingest -- GPU tests of the c2_fq_* kernels, then FASTQ -> count tensors on either route
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03i
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 900 python -m pytest tests/test_gpu_fastq_device.py -m gpu -q -x ) > "$OUT/gpu_tests_fastq_device.txt" 2>&1
tail -15 "$OUT/gpu_tests_fastq_device.txt"
timeout 900 python tools/e2e_device_ingest.py > "$OUT/e2e_device_ingest.jsonl" 2> "$OUT/e2e_device_ingest.err"
cat "$OUT/e2e_device_ingest.jsonl"; tail -5 "$OUT/e2e_device_ingest.err"
