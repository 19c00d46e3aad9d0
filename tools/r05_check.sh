#!/bin/bash
# round 5 working session: GPU tests (all), config 4 with / without the partition (3 M reads, no profiler), the headline quick
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05e}; mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -q ) > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt
for mode in part nopart; do
  if [ $mode = nopart ]; then export C2_NO_ALLREFS_PARTITION=1; else unset C2_NO_ALLREFS_PARTITION; fi
  timeout 300 python bench.py --config 4 --reads 3000000 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-dedup-leg --check 50 > $OUT/cfg4_$mode.json 2> $OUT/cfg4_$mode.err
  python -c "
import json; d=json.loads([x for x in open('$OUT/cfg4_$mode.json') if x.startswith('{')][-1]); print('$mode', d['alignments_per_s']/1e6, d['step_breakdown_ms']['align_chain'], d['partition'] and d['partition']['classes'], d['config']['tasks_left_after_each_banded_launch'], d['checks'].get('chain_equals_full_plane'))"
done
unset C2_NO_ALLREFS_PARTITION
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-dedup-leg --check 50 > $OUT/headline.json 2> $OUT/headline.err
python -c "
import json; d=json.loads([x for x in open('$OUT/headline.json') if x.startswith('{')][-1]); print('headline', d['value']/1e6, d['step_breakdown_ms'], d['checks'].get('chain_equals_full_plane'))"
