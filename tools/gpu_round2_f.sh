#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02f; mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python tools/ab/variants.py --rounds 1 x4,C2_NO_PACKED_FILL=1 x4_pad5k,C2_NO_PACKED_FILL=1,C2_DEBUG_X_LDS_PAD=5000 x4_pad13k,C2_NO_PACKED_FILL=1,C2_DEBUG_X_LDS_PAD=13000 packed > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
timeout 600 python tools/ab/variants.py --rounds 1 --config 2 --reads 1000000 x4,C2_NO_PACKED_FILL=1 packed > "$OUT/variants_config2.txt" 2>&1
cat "$OUT/variants_config2.txt"
