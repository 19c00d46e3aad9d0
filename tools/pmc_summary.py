#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of tools/profile_round.sh: per-kernel launch statistics of the traced run and
per-kernel counter sums of the --pmc passes (one launch over 2 M reads each).
usage: pmc_summary.py <prof dir> <tag>  ->  <prof dir>/kernel_stats_<tag>.csv, <prof dir>/pmc_summary_<tag>.json"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    name = re.sub(r"^void ", "", name)
    return name.replace(".kd", "")


def main():
    out, tag = sys.argv[1], sys.argv[2]
    stats = {}
    for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                s = stats.setdefault(k, [0, 0, 1 << 62, 0])
                s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
    tot = sum(s[1] for s in stats.values()) or 1
    with open(os.path.join(out, "kernel_stats_%s.csv" % tag), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "total_ms", "avg_ms", "min_ms", "max_ms", "percent"])
        for k, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, s[0], "%.3f" % (s[1] / 1e6), "%.3f" % (s[1] / s[0] / 1e6), "%.3f" % (s[2] / 1e6), "%.3f" % (s[3] / 1e6),
                        "%.2f" % (100.0 * s[1] / tot)])
    pmc = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if not k.startswith("c2_"):
                        continue
                    e = pmc.setdefault(k, {})
                    e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    with open(os.path.join(out, "pmc_summary_%s.json" % tag), "w") as fh:
        reads = int(os.environ.get("C2_PMC_READS", "2000000"))       # (tools/profile_round.sh exports what it passed to bench.py --reads)
        json.dump({"note": "counter sums per kernel over ONE bench.py launch of %d reads (separate --pmc passes)" % reads, "reads": reads, "kernels": pmc},
                  fh, indent=1, sort_keys=True)
    print(open(os.path.join(out, "kernel_stats_%s.csv" % tag)).read())
    print(json.dumps(pmc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
