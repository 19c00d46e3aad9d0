#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02o
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/consensus_trace" -o trace -- python "$ROOT/tools/consensus_rate.py" > "$OUT/consensus_rate.json" 2> "$OUT/consensus.err"
tail -1 "$OUT/consensus_rate.json"
grep -i "consensus" "$OUT"/consensus_trace/*/*kernel_stats.csv "$OUT"/consensus_trace/*kernel_stats.csv 2>/dev/null | head -3
