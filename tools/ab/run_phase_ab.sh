#!/bin/bash
# phase cycle accounting (tools/phase_profile.py) of the tree's build and of tools/ab/libcrispresso2_amd_old.so on the same box
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/old && mkdir -p /tmp/old/tools && cp -r $ROOT/crispresso2_amd /tmp/old/ && cp $ROOT/tools/phase_profile.py /tmp/old/tools/
cp $ROOT/tools/ab/libcrispresso2_amd_old.so /tmp/old/crispresso2_amd/lib/libcrispresso2_amd.so
for k in auto diag4only; do
  [ $k = diag4only ] && continue
  echo "new: $(cd $ROOT && python tools/phase_profile.py --reads 2000000 2>&1 | tail -1)"
  echo "old: $(cd /tmp/old && python tools/phase_profile.py --reads 2000000 2>&1 | tail -1)"
done
