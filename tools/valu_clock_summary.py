#!/usr/bin/env python3
"""tools/valu_microbench4.hip under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE`: per row the event-timed cycles at the nominal
2.4 GHz (the program's own output) next to the TRUE cycles per wave64 instruction from GRBM_GUI_ACTIVE / (waves per SIMD x ITER x 64)
and the clock GRBM_GUI_ACTIVE / kernel duration.  usage: valu_clock_summary.py <microbench stdout> <rocprof output dir>"""
import csv, glob, os, re, sys
txt, d = sys.argv[1], sys.argv[2]
rows = {}
for line in open(txt):
    m = re.match(r"(.+?)\s+([0-9.]+)\s+([0-9.]+)\s+kind (\d+)", line)
    if m:
        rows[int(m.group(4))] = (m.group(1).strip(), float(m.group(2)), float(m.group(3)))
ITER = 4096
XCDS = 8                                    # rocprofv3 reports GRBM_GUI_ACTIVE summed over the chip's 8 XCDs (each has its own GRBM)
disp = {}                                   # (kind, grid) -> (duration ns, GRBM_GUI_ACTIVE)
ctr = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
ktr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in ktr:
    for r in csv.DictReader(open(f)):
        dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0))
per = {}
for f in ctr:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        did = r.get("Dispatch_Id")
        name = r.get("Kernel_Name", "")
        m = re.search(r"k<(\d+)>", name)
        if not m:
            continue
        grid = int(r.get("Grid_Size", 0) or 0)
        per.setdefault((int(m.group(1)), grid), []).append((float(r["Counter_Value"]), dur.get(did, (0,))[0]))
print("cycles per wave64 instruction per SIMD; event-timed columns assume 2.4 GHz, GRBM columns are counted cycles (GRBM_GUI_ACTIVE / 8 XCDs; the second launch of each pair)")
print("%-34s %9s %9s %11s %11s %9s" % ("instruction", "3w@2.4GHz", "8w@2.4GHz", "3w GRBM", "8w GRBM", "clock GHz"))
for k in sorted(rows):
    name, c3, c8 = rows[k]
    out = []
    clk = []
    for wps in (3, 8):
        grid = 256 * 4 * wps * 64
        v = per.get((k, grid)) or per.get((k, 256 * 4 * wps))
        if v:
            val, ns = v[-1]
            out.append("%11.2f" % (val / XCDS / (wps * ITER * 64.0)))
            if ns:
                clk.append(val / XCDS / ns)
        else:
            out.append("%11s" % "-")
    print("%-34s %9.2f %9.2f %s %s %9s" % (name, c3, c8, out[0], out[1], ("%.2f" % (sum(clk) / len(clk))) if clk else "-"))
