#!/bin/bash
# A/B of two builds of the library on the SAME GPU box: the tree's build vs tools/ab/libcrispresso2_amd_old.so (built by hand
# from an earlier revision).  Prints ms per step, the dominant kernel's average launch and the chain time, alternating runs.
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/old && mkdir -p /tmp/old && cp -r $ROOT/bench.py $ROOT/crispresso2_amd $ROOT/oracle $ROOT/BASELINE.json /tmp/old/ 2>/dev/null
cp $ROOT/tools/ab/libcrispresso2_amd_old.so /tmp/old/crispresso2_amd/lib/libcrispresso2_amd.so
ARGS="--reads ${READS:-2000000} --steps 5 --warmup 1 --no-cpu-baseline --check 0"
show() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[2], 'ms/step %.3f  first kernel %.3f ms  chain %.3f ms  reads/s %.1f M' % (d['ms_per_step'], r['avg_launch_ms'], r['chain_avg_ms'], d['value']/1e6))" $1 $2; }
for i in 1 2 3; do
  (cd $ROOT && python bench.py $ARGS > /tmp/new_$i.json 2>/tmp/new_$i.err) ; show /tmp/new_$i.json new
  (cd /tmp/old && python bench.py $ARGS > /tmp/old_$i.json 2>/tmp/old_$i.err) ; show /tmp/old_$i.json old
done
