#!/usr/bin/env python3
"""Kernel timeline of single camera steps from a rocprofv3 --kernel-trace CSV (start / end / duration in us from the frame's first
kernel, queue, kernel name).   usage: python tools/frame_timeline.py <kernel_trace.csv> [frame offsets from the middle ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "hist_kernel" in r["Kernel_Name"]]
offs = [int(a) for a in sys.argv[2:]] or [0]
print(f"{len(rows)} dispatches, {len(first)} frames")
for off in offs:
    m = len(first) // 2 + off
    if m + 1 >= len(first):
        continue
    a, b = first[m], first[m + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"--- frame {m}")
    for r in rows[a:b]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:8.1f} {e:8.1f} {e - s:6.1f}  q{r['Queue_Id']:>2}  {r['Kernel_Name'][:56]}")
