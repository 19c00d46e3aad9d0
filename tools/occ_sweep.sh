#!/bin/bash
# occupancy experiment: pad the diagonal-band kernel's LDS so that fewer workgroups share a CU; prints reads/s per setting
for pad in "$@"; do
  C2_DEBUG_DIAG_LDS_PAD=$pad timeout 100 python bench.py --steps 3 --warmup 1 --workers 8 --no-cpu-baseline --reads 2000000 2>&1 | tail -1 > /tmp/occ.json
  python - "$pad" <<'PY'
import sys, json
d = json.loads(open('/tmp/occ.json').read())
print(sys.argv[1], round(d["value"] / 1e6, 2), "M reads/s", d["ms_per_step"], {k: v for k, v in d["config"].items() if "wg" in k or "workgroup" in k or "ms" in k})
PY
done
