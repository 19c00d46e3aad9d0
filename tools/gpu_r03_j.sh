#!/bin/bash
# Round 3: kernel trace of the device ingest alone
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03j
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C2_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o ingest -- python "$ROOT/tools/device_ingest_only.py" 10000000 3 > "$OUT/device_ingest_only.txt" 2>&1
tail -5 "$OUT/device_ingest_only.txt"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/ingest_kernel_stats.csv"; head -25 "$OUT/ingest_kernel_stats.csv" | cut -c1-200
rm -rf "$OUT/trace"
