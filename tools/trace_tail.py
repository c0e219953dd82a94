#!/usr/bin/env python3
"""Kernel sequence of the LAST iteration of a rocprofv3 --kernel-trace csv: name (shortened), duration, gap to the previous
kernel's end, grid / workgroup size.   python tools/trace_tail.py <kernel_trace.csv> <launches per iteration | marker kernel>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
arg = sys.argv[2]
if arg.isdigit():
    rows = rows[-int(arg):]
else:  # from the last-but-one occurrence of the marker kernel to the last one
    idx = [i for i, r in enumerate(rows) if arg in r["Kernel_Name"]]
    rows = rows[idx[-2]:idx[-1]]
prev_end = None
tot = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("dm::", "")
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    print(f"{name[:60]:60s} {(e - s) / 1e3:9.2f} us  gap {gap:7.2f}  wgs {grid // max(wg, 1):6d} x {wg}")
    prev_end = e
    tot += e - s
print(f"{len(rows)} launches, kernel time {tot / 1e6:.3f} ms, span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6:.3f} ms")
