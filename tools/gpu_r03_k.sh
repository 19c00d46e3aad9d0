#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03k
mkdir -p "$OUT"
cd "$ROOT"
C2_FQ_TRACE=1 C2_CHUNK_MB=256 timeout 600 python tools/device_ingest_only.py 10000000 6 2>&1 | grep -v amdgpu.ids | tail -5 > "$OUT/device_ingest_trace.txt"
cat "$OUT/device_ingest_trace.txt" | cut -c1-500
