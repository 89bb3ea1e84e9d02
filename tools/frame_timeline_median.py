#!/usr/bin/env python3
"""Median kernel timeline of a camera frame from a rocprofv3 --kernel-trace CSV: for every kernel (by name and by its occurrence inside the
frame) the median start and end, in us from the start of the frame's first kernel (hist_kernel), over the frames of the trace that hold
the usual set of kernels; the between-frame kernels (IMU propagation, wheel update, cloning) are left out.
usage: python tools/frame_timeline_median.py <kernel_trace.csv> [frames to skip at the start, default 80]"""
import csv
import statistics as st
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
BETWEEN = ("propagate_kernel", "ekf_prop_", "wheel", "cov_clone", "cov_marginalize", "copyBuffer", "fillBuffer")


def short(n):
    n = n.split("(")[0]
    n = n.replace("plv::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return n.strip()


first = [i for i, r in enumerate(rows) if "hist_kernel" in r["Kernel_Name"]]
frames = []
for a, b in zip(first[skip:-1], first[skip + 1:]):
    t0 = int(rows[a]["Start_Timestamp"])
    seen = defaultdict(int)
    f = {}
    for r in rows[a:b]:
        n = short(r["Kernel_Name"])
        if any(k in n for k in BETWEEN):
            continue
        key = (n, seen[n])
        seen[n] += 1
        f[key] = ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"])
    frames.append(f)
keys = defaultdict(int)
for f in frames:
    for k in f:
        keys[k] += 1
print(f"{len(frames)} frames; kernels present in at least a third of them; us from the start of hist_kernel: median start, median end, median duration, share of frames, queue")
out = []
for k, c in keys.items():
    if c * 3 < len(frames):
        continue
    s = st.median(f[k][0] for f in frames if k in f)
    e = st.median(f[k][1] for f in frames if k in f)
    d = st.median(f[k][1] - f[k][0] for f in frames if k in f)
    q = st.mode(f[k][2] for f in frames if k in f)
    out.append((s, e, d, c / len(frames), q, k))
for s, e, d, share, q, k in sorted(out):
    print(f"{s:8.1f} {e:8.1f} {d:6.1f}  {share:4.0%}  q{q:>2}  {k[0]}{'' if k[1] == 0 else ' #%d' % (k[1] + 1)}")
