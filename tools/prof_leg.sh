#!/bin/bash
# rocprofv3 kernel trace of one robustness leg (tools/robust_rate.py): tools/prof_leg.sh <leg> [reads]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
LEG=${1:-fanc_shaped}; N=${2:-3000000}
mkdir -p gpurun_out/r05_legs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$LEG -o p -- python tools/robust_rate.py --reads $N --legs $LEG --no-check --steps 3 > gpurun_out/r05_legs/$LEG.json 2> gpurun_out/r05_legs/$LEG.err
f=$(find /tmp/prof_$LEG -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_legs/kernel_stats_$LEG.csv
head -14 gpurun_out/r05_legs/kernel_stats_$LEG.csv | cut -c1-150; tail -1 gpurun_out/r05_legs/$LEG.json | cut -c1-300
