#!/usr/bin/env python3
"""Per-frame kernel timeline from a rocprofv3 --kernel-trace CSV: for one steady-state frame (anchored at a kernel name), every kernel
with its queue, start offset, duration and the gap to the previous kernel's end on the same queue.
usage: python3 tools/timeline.py <kernel_trace.csv> [anchor-kernel-substring] [frame-index-from-end]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "equalize"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]["Start_Timestamp"])
last_end = {}
print(f"frame of {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us to the next frame's first kernel")
for r in rows[a:b]:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = r["Kernel_Name"].split("(")[0].replace("plv::", "").replace("void ", "")[:44]
    print(f"q{q:>2} +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:7.1f}  {name}  grid {r['Grid_Size_X']}/{r['Workgroup_Size_X']}")
