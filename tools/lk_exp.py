#!/usr/bin/env python3
"""lk_kernel on two consecutive frames of a bench workload, launch time per variant (measurement aid).

Renders a few frames of the workload's drive, tracks the library's own detections through them (so that the points are tracks of a
few frames' age, like the bench's), then times plv_lk_track of the last pair under every value of the experimental variant knob
(bits 21-27 of plv_debug_knobs: template argument of lk_kernel) and checks that every variant returns the same bits.

usage: python tools/lk_exp.py [workload] [variants, comma separated]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench

wl_name = sys.argv[1] if len(sys.argv) > 1 else "C"
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 0]   # 2: the loop of rounds 2-4, 0: the lean iteration (default)
wl = bench.WORKLOADS[wl_name]
N_FRAMES = 8
stream = bench.build_stream(wl, N_FRAMES + 30, 8)      # (forks: before the GPU is touched)
import __graft_entry__ as ge
pkg = ge.load_pkg()
cfg = pkg.default_config(wl["w"], wl["h"])
cfg.num_features = wl["num_features"]
ctx = pkg.Context(cfg)
imgs = stream["imgs"][25:25 + N_FRAMES]
pts, ids, cid = np.zeros((0, 2), np.float32), np.zeros(0, np.uint64), 1
ctx.feed_image(imgs[0])
for k in range(1, N_FRAMES):
    pts, ids, cid = ctx.perform_detection(0, pts, ids, cid)[:3]
    ctx.feed_image(imgs[k])
    p1, mask = ctx.perform_matching(pts, pts)[:2]
    last_pair = (pts.copy(), p1.copy())
    keep = mask.astype(bool) & (p1[:, 0] >= 0) & (p1[:, 1] >= 0) & (p1[:, 0] < wl["w"]) & (p1[:, 1] < wl["h"])
    pts, ids = p1[keep], ids[keep]
p0 = last_pair[0]
print(f"workload {wl_name}: {len(p0)} points")
ref = None
SHIFT = 21
for rep in range(2):
    for v in variants:
        pkg.debug_knobs(v << SHIFT)
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(40):
            out = ctx.lk_track(p0, p0)
        t = ctx.prof_table()["lk_kernel"]
        ctx.prof_enable(False)
        same = True
        if ref is None:
            ref = out
        else:
            same = all(np.array_equal(a, b) for a, b in zip(ref, out))
        it = out[2]
        print(f"variant {v}: lk_kernel {t[1] / t[0] * 1e3:6.1f} us   iterations/pt mean {it.mean():.1f} max {it.max()}  p90 {np.percentile(it, 90):.0f}"
              f"  lost {int((out[1] == 0).sum())}  identical to variant {variants[0]}: {same}")
pkg.debug_knobs(0)
ctx.close()

# launch time against the iteration cap and the number of levels (what a level's set-up costs, what an iteration costs)
print("default: lk_max_iters", cfg.lk_max_iters, "pyr_levels", cfg.pyr_levels, "win", cfg.win_size)
for iters, lv in ((cfg.lk_max_iters, cfg.pyr_levels), (1, cfg.pyr_levels), (2, cfg.pyr_levels), (4, cfg.pyr_levels), (8, cfg.pyr_levels), (cfg.lk_max_iters, 1), (1, 1), (cfg.lk_max_iters, 3), (1, 3)):
    c2 = pkg.default_config(wl["w"], wl["h"])
    c2.num_features = wl["num_features"]
    c2.lk_max_iters, c2.pyr_levels = iters, lv
    x = pkg.Context(c2)
    x.feed_image(imgs[-2])
    x.feed_image(imgs[-1])
    x.prof_enable(True)
    x.prof_reset()
    for _ in range(40):
        out = x.lk_track(p0, p0)
    t = x.prof_table()["lk_kernel"]
    print(f"max_iters {iters:2d} levels {lv}: lk_kernel {t[1] / t[0] * 1e3:6.1f} us, iterations/pt mean {out[2].mean():.1f} max {out[2].max()}")
    x.close()

# diagnostic builds: cycles of one phase summed over a point's iterations (8: position -> products, 9: wave sums + barrier,
# 10: partials -> step, 11: a level's set-up), per point in place of the iteration count
ctx2 = pkg.Context(cfg)
ctx2.feed_image(imgs[-2])
ctx2.feed_image(imgs[-1])
pkg.debug_knobs(0)
its = ctx2.lk_track(p0, p0)[2].astype(np.float64)
for v, name, per in ((17, "position -> products", its), (33, "wave sums + write + barrier", its), (49, "partials -> step", its), (65, "level set-up", 5.0)):
    pkg.debug_knobs(v << SHIFT)
    cyc = ctx2.lk_track(p0, p0)[2].astype(np.float64)
    q = cyc / per
    print(f"phase {name:30s}: cycles per {'iteration' if v < 65 else 'level'}: median {np.median(q):7.0f}  p10 {np.percentile(q, 10):7.0f}  p90 {np.percentile(q, 90):7.0f}")
pkg.debug_knobs(0)
