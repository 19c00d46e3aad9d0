#!/bin/bash
# Round 2, GPU session: e2e phase split, occupancy sensitivity of the packed kernel, PMC profile of the default chain
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02i
mkdir -p "$OUT"
cd "$ROOT"
nproc > "$OUT/nproc.txt"; lscpu | head -20 >> "$OUT/nproc.txt"
C2_FASTQ_TRACE=1 timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"; grep c2_fastq "$OUT/e2e.err" | tail -2
for pad in 0 3456 7552; do
    C2_DEBUG_PK_LDS_PAD=$pad timeout 600 python bench.py --no-cpu-baseline --check 0 --steps 3 --warmup 1 > "$OUT/bench_pkpad_$pad.json" 2> "$OUT/bench_pkpad_$pad.err"
    python -c "
import json
d=json.loads([x for x in open('$OUT/bench_pkpad_$pad.json') if x.startswith('{')][-1])
print('pkpad $pad', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms']['align_chain'], round(d['roofline']['avg_launch_ms'],2))"
done
bash tools/profile_round.sh default > "$OUT/profile.log" 2>&1
tail -5 "$OUT/profile.log"
cat "$ROOT/gpurun_out/prof_default/pmc_summary_default.json" | head -80
