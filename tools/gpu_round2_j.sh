#!/bin/bash
# Round 2, GPU session: count pass overlapped with the next batch's launch chain (bench.py default) vs --serial; e2e after the ingest changes
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02j
mkdir -p "$OUT"
cd "$ROOT"
C2_FASTQ_TRACE=1 timeout 600 python tools/e2e_rate.py --reads 2000000 > "$OUT/e2e_rate_2M.json" 2> "$OUT/e2e.err"; tail -1 "$OUT/e2e_rate_2M.json"; grep c2_fastq "$OUT/e2e.err" | tail -2
for mode in "" "--serial"; do
  for steps in 3 8; do
    timeout 600 python bench.py --no-cpu-baseline --check 0 --steps $steps --warmup 1 $mode > "$OUT/bench_s${steps}_${mode#--}.json" 2> "$OUT/bench_s${steps}_${mode#--}.err"
    python -c "
import json
d=json.loads([x for x in open('$OUT/bench_s${steps}_${mode#--}.json') if x.startswith('{')][-1])
print('mode [$mode] steps $steps', round(d['value']/1e6,1), 'M reads/s', d['step_breakdown_ms']['align_chain'], d['step_breakdown_ms']['count_vectors_and_all_reduce'], round(d['roofline']['avg_launch_ms'],2), d['ms_per_step'])"
  done
done
timeout 900 python bench.py --config 4 --no-cpu-baseline --check 0 --steps 3 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_c4.json') if x.startswith('{')][-1])
print('config4', round(d['alignments_per_s']/1e6,1), 'M aln/s', d['step_breakdown_ms'])"
timeout 900 python bench.py --config 4 --no-cpu-baseline --check 0 --steps 3 --warmup 1 --serial > "$OUT/bench_c4s.json" 2> "$OUT/bench_c4s.err"
python -c "
import json
d=json.loads([x for x in open('$OUT/bench_c4s.json') if x.startswith('{')][-1])
print('config4 serial', round(d['alignments_per_s']/1e6,1), 'M aln/s', d['step_breakdown_ms'])"
