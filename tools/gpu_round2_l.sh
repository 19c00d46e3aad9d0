#!/bin/bash
# Round 2, GPU session: full GPU suite on the current build; host path pipelined vs one-shot; e2e
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02l
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > "$OUT/gpu_tests.txt" 2>&1
grep -E "passed|failed|rror" "$OUT/gpu_tests.txt" | tail -5
timeout 600 python tools/host_path_rate.py > "$OUT/host_path_rate_2M.json" 2> "$OUT/host_path.err"; tail -1 "$OUT/host_path_rate_2M.json"
C2_FASTQ_TRACE=1 timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"; grep c2_fastq "$OUT/e2e.err" | tail -3
timeout 600 python tools/e2e_rate.py --reads 2000000 --gz > "$OUT/e2e_rate_2M_gz.json" 2> "$OUT/e2e_gz.err"; tail -1 "$OUT/e2e_rate_2M_gz.json"
