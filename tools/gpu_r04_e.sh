#!/bin/bash
# ablation of the count kernel: where does the time of an alignment with a gap go?
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04e
mkdir -p "$OUT"
cd "$ROOT"
for v in ab0 ab1 ab2 ab3; do
  LIB=$ROOT/crispresso2_amd/lib/variants/lib_$v.so
  echo "== $v" | tee -a "$OUT/count_split.txt"
  C2_AMD_LIB=$LIB timeout 300 python tools/count_kernel_split.py 4000000 2>&1 | grep -E '"kind"' | grep -v "gap-free, [15] " | tee -a "$OUT/count_split.txt"
done
