#!/usr/bin/env python3
"""Device-side gaps of the camera step from a rocprofv3 --kernel-trace CSV: per frame, the idle time between the end of the flow
(ransac_select_kernel) and the start of the fused point launch (jacobian_nullspace_kernel), and between the point update's commit
and the fused line launch.   usage: python tools/gap_from_trace.py <kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    for k in ("ransac_select_kernel", "line_jacobian_nullspace_kernel", "jacobian_nullspace_kernel", "ekf_commit_kernel", "lk_kernel", "hist_kernel"):
        if k in n:
            return k
    return None
ev = [(s, e, short(n)) for s, e, n in rows if short(n)]
gaps_a, gaps_b, frame = [], [], []
last_sel = last_commit = last_hist = None
for s, e, k in ev:
    if k == "hist_kernel":
        last_hist = s
    elif k == "ransac_select_kernel":
        last_sel = e
    elif k == "jacobian_nullspace_kernel" and last_sel is not None:
        gaps_a.append((s - last_sel) * 1e-3)
        last_sel = None
    elif k == "ekf_commit_kernel":
        last_commit = e
    elif k == "line_jacobian_nullspace_kernel":
        if last_commit is not None:
            gaps_b.append((s - last_commit) * 1e-3)
            last_commit = None
        if last_hist is not None:
            frame.append((e - last_hist) * 1e-3)
for name, g in (("flow end -> fused point launch starts", gaps_a), ("point commit end -> fused line launch starts", gaps_b), ("hist start -> line launch end", frame)):
    g = np.array(g[20:]) if len(g) > 40 else np.array(g)
    print(f"{name}: n {len(g)}  median {np.median(g):.1f} us  mean {g.mean():.1f}  p10 {np.percentile(g, 10):.1f}  p90 {np.percentile(g, 90):.1f}")
