#!/bin/bash
# Round 2, GPU session: e2e after huge-page arena; occupancy experiment (timing only) and phase ablation of the packed kernel
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02k
mkdir -p "$OUT"
cd "$ROOT"
cat /sys/kernel/mm/transparent_hugepage/enabled > "$OUT/thp.txt"
C2_FASTQ_TRACE=1 timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"; grep c2_fastq "$OUT/e2e.err" | tail -2
timeout 1200 python tools/ab/variants.py --rounds 2 --reads 5000000 --steps 3 base lb4=tools/ab/lib_lb4.so occ16=tools/ab/lib_occ16.so noepi,C2_DEBUG_SKIP_EPILOGUE=1 noepi_half,C2_DEBUG_SKIP_EPILOGUE=1,C2_DEBUG_HALF_FILL=1 > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
