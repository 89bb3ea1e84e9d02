"""Per-kernel time of the line front-end (config C: 752x480 with ~120 rendered edges)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import __graft_entry__ as ge  # noqa: E402
import synth  # noqa: E402

pkg = ge.load_pkg()
W, H = 752, 480
ctx = pkg.Context(pkg.default_config(W, H))
canvas = synth.texture_canvas(W, H, seed=11, blobs=200, lines=120)
frames = [synth.render_frame(canvas, W, H, tx=3.0 * i, ty=-2.0 * i, rot_deg=0.2 * i) for i in range(4)]
vps = ctx.vanishing_points(np.eye(3), synth.EUROC_K8)
for i, f in enumerate(frames):
    ctx.tracker_feed(1.0 + 0.05 * i, f)
    ctx.line_tracker_feed(1.0 + 0.05 * i, vps)
N = 20
for on_device in (False, True):
    ctx.line_walk_mode(on_device)
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for i in range(N):
        lines = ctx.detect_lines(0)
    t1 = time.perf_counter()
    print(f"walk on {'device' if on_device else 'host'}: {len(lines)} lines; plv_detect_lines wall {1e3 * (t1 - t0) / N:.3f} ms")
    for name, (cnt, ms) in sorted(ctx.prof_table().items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:24s} {1e3 * ms / N:9.1f} us/frame  ({cnt / N:.1f} launches)")
