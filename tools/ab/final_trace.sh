#!/bin/bash
# the kernel trace of the shipped build (tools/profile_round.sh's first step alone) and the GPU tests of the launch chain
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/final_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
    python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --check 0 --no-dedup-leg --workers 1 --no-extras > "$OUT/trace_bench.log" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/rocprofv3_kernel_stats.csv" && head -12 "$f" | cut -c1-150
find "$OUT/trace" -type f -name "*kernel_trace.csv" -delete
cd "$ROOT"
( time timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -q -m gpu -x ) > "$OUT/gpu_tests_chain.txt" 2>&1
tail -4 "$OUT/gpu_tests_chain.txt"
