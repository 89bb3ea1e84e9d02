"""Quick device-side timing of the two halves of the per-frame path (HIP events per kernel class)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import synth  # noqa: E402

pkg = ge.load_pkg()
w, h = 752, 480
ctx = pkg.Context(pkg.default_config(w, h))
canvas = synth.texture_canvas(w, h, seed=42)
f0 = synth.render_frame(canvas, w, h)
f1 = synth.render_frame(canvas, w, h, tx=4.2, ty=-3.1, rot_deg=0.3, scale=1.002)
ctx.image_stage(0, f0)
ctx.image_stage(1, f1)
pts = synth.grid_points(w, h, 250, seed=5, border=16)
n, k = 113, 98
P = synth.spd_cov(n)
cols = synth.col_map(n, k)
rows, Hf, Hx, res = synth.msckf_batch(F=70, M=15, k=k, seed=1, ragged=False)
ctx.cov_upload(P)
ctx.cov_checkpoint()
ctx.feat_batch_upload(rows, Hf, Hx, res, cols)


def step(i):
    ctx.feed_staged(i & 1)
    out = ctx.perform_matching(pts, pts)
    ctx.cov_rollback()
    rc, dx, acc, nr = ctx.msckf_update_resident(n, 2.25)
    return out, acc


ctx.feed_staged(1)
for i in range(10):
    step(i)
ctx.prof_enable(True)
ctx.prof_reset()
N = 50
t0 = time.perf_counter()
for i in range(N):
    out, acc = step(i)
t1 = time.perf_counter()
ctx.prof_enable(False)
print(f"wall per frame with profiling events: {(t1 - t0) / N * 1e3:.3f} ms; accepted {int(acc.sum())}/70; "
      f"tracked {int(out[1].sum())}/250; lk iters/pt {out[4] / 250:.1f}")
tot = 0
for name, (cnt, ms) in sorted(ctx.prof_table().items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:24s} launches/frame {cnt / N:5.1f}  us/frame {ms / N * 1e3:9.2f}  us/launch {ms / max(cnt,1) * 1e3:8.2f}")
    tot += ms
print(f"  sum of kernel time per frame: {tot / N * 1e3:.1f} us")
t0 = time.perf_counter()
for i in range(N):
    step(i)
t1 = time.perf_counter()
print(f"wall per frame without profiling: {(t1 - t0) / N * 1e3:.3f} ms")
