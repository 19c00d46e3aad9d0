#!/bin/bash
# config 4 (all-references batch) with and without the partition for such batches: rocprofv3 kernel trace of 3 M reads x 3 amplicons.
# (--workers 1: bench.py must not fork its data-generation pool under the profiler; every step under its own timeout)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05d
for mode in part nopart; do
  if [ $mode = nopart ]; then export C2_NO_ALLREFS_PARTITION=1; else unset C2_NO_ALLREFS_PARTITION; fi
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o p -- python bench.py --config 4 --reads 3000000 --steps 3 --warmup 1 --workers 1 --no-cpu-baseline --no-extras --check 0 --no-dedup-leg > gpurun_out/r05d/bench_cfg4_$mode.json 2> gpurun_out/r05d/bench_cfg4_$mode.err
  f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05d/kernel_stats_cfg4_$mode.csv
  echo == $mode; head -14 gpurun_out/r05d/kernel_stats_cfg4_$mode.csv | cut -c1-160
  python -c "
import json; d=json.loads([x for x in open('gpurun_out/r05d/bench_cfg4_$mode.json') if x.startswith('{')][-1]); print(d['alignments_per_s']/1e6, d['step_breakdown_ms'], d['partition'], d['config']['tasks_left_after_each_banded_launch'])"
done
