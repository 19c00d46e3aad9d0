#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03k
mkdir -p "$OUT"
cd "$ROOT"
C2_FQ_TRACE=1 timeout 600 python tools/device_ingest_only.py 10000000 4 > "$OUT/device_ingest_trace.txt" 2>&1
C2_FQ_TRACE=1 C2_CHUNK_MB=256 timeout 600 python tools/device_ingest_only.py 10000000 4 >> "$OUT/device_ingest_trace.txt" 2>&1
grep seconds "$OUT/device_ingest_trace.txt"
