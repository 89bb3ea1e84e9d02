#!/usr/bin/env python3
"""Splits the per-frame trace of `PLV_BENCH_FRAMES=1 bench.py --alternate-knobs a,b` (stderr) by knob: medians of the frame time and of
the times inside the frame (flow wait, point update, line update; worker: maps wait, detection, feed)."""
import ast, sys, statistics as st
txt = open(sys.argv[1]).read()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
line = [l for l in txt.splitlines() if l.startswith("[frames] (ms,")][-1]
rows = ast.literal_eval(line.split("per step:", 1)[1].strip())
names = ("ms", "feat", "pool", "tri", "acc", "lk_it", "segs", "flow_wait", "points", "lines", "w_maps", "w_extract", "w_feed")
for k in range(n):
    sel = rows[k::n]
    print("knob #%d (%d frames): " % (k, len(sel)) + "  ".join("%s %.1f" % (names[i], st.median(r[i] for r in sel) * (1e3 if i == 0 else 1)) for i in (0, 7, 8, 9, 10, 11, 12)))
