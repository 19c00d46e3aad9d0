#!/bin/bash
# Round 3, second GPU session: GPU tests again (soak bound fixed, primed replay), shim call rates, ingest phase trace at 10 M reads
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03b
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -8
timeout 600 python tools/shim_call_rate.py > "$OUT/shim_call_rate.json" 2> "$OUT/shim_call_rate.err"; tail -1 "$OUT/shim_call_rate.json"; tail -3 "$OUT/shim_call_rate.err"
timeout 600 python tools/ingest_trace.py > "$OUT/ingest_trace.txt" 2>&1; cat "$OUT/ingest_trace.txt"
nproc; free -g | head -2; df -h /dev/shm | tail -1
