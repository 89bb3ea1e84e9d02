#!/usr/bin/env python3
"""Times the pieces a config-C frame adds to config B: line detection, point-line assignment / matching, line update."""
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
W, H = 752, 480
ctx = pkg.Context(pkg.default_config(W, H))
canvas = synth.texture_canvas(W, H, seed=42, lines=120)
frames = [synth.render_frame(canvas, W, H), synth.render_frame(canvas, W, H, tx=4.2, ty=-3.1, rot_deg=0.3, scale=1.002)]
ctx.image_stage(0, frames[0]); ctx.image_stage(1, frames[1])
ctx.feed_staged(0); ctx.feed_staged(1)
pts = synth.grid_points(W, H, 250, seed=5, border=16)
ids = np.arange(1, 251, dtype=np.uint64)


def timeit(f, n=200):
    for _ in range(10):
        f()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, r


for mode in (0, 1):
    ctx.line_walk_mode(mode)
    us, lines = timeit(lambda: ctx.detect_lines(1))
    print(f"detect_lines walk_mode={mode}: {us:.1f} us, {len(lines)} lines")
ctx.line_walk_mode(0)
us, a = timeit(lambda: ctx.assign_points_to_lines(lines, pts, ids))
print(f"assign_points_to_lines: {us:.1f} us, kept {len(a[0]) if isinstance(a, tuple) else a}")
sc = synth.vio_scene(n_clones=15, F=4, calib_int=True, seed=3)
ls = synth.line_scene(sc, L=80, M=15, noise_px=0.4)
st, _ = synth.scene_views(pkg, sc)
lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"])
cols = ctx.line_jacobian_columns(st, lt)
n = sc["n_state"]
P = synth.spd_cov(n)
ctx.cov_upload(P); ctx.cov_checkpoint()


def upd():
    ctx.cov_rollback()
    ctx.build_line_jacobians_resident(st, lt, cols, 32)
    return ctx.msckf_update_resident(n, 2.25, res_norm_gate=0.0)


us, r = timeit(upd)
print(f"line update (80 x 15, k = {len(cols)}): {us:.1f} us, accepted {int(r[2].sum())}, rows {r[3]}")
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(20):
    ctx.detect_lines(1); upd()
for k, v in sorted(ctx.prof_table().items(), key=lambda kv: -kv[1][1] if isinstance(kv[1], tuple) else 0)[:25]:
    print(k, v)
