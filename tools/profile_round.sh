#!/bin/bash
# Profiles of bench.py for profiles/rNN/ (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag> [extra bench.py arguments]
# writes gpurun_out/prof_<tag>/: trace/ (rocprofv3 --kernel-trace --stats of the default workload) and pmc_*/ (separate
# --pmc passes over one launch of 2 M reads: FETCH_SIZE; WRITE_SIZE; SQ instruction counts; SQ cycle counters; LDS),
# then tools/pmc_summary.py condenses them into kernel_stats_<tag>.csv and pmc_summary_<tag>.json.
# --workers 1: bench.py must not fork() its data-generation pool under the profiler (the children hang in its exit handler).
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PMC_READS=${C2_PMC_READS:-2000000}; export C2_PMC_READS=$PMC_READS
COMMON="--no-cpu-baseline --check 0 --no-dedup-leg --workers 1 --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
    python "$ROOT/bench.py" --steps 3 --warmup 1 $COMMON "$@" > "$OUT/trace_bench.log" 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -o pmc -- \
        python "$ROOT/bench.py" --reads $PMC_READS --steps 1 --warmup 0 $COMMON "$@" > "$OUT/pmc_$i.log" 2>&1
    i=$((i + 1))
done
python "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG"
