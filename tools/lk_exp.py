#!/usr/bin/env python3
"""lk_kernel on two consecutive frames of a bench workload, launch time per variant (measurement aid).

Renders a few frames of the workload's drive, tracks the library's own detections through them (so that the points are tracks of a
few frames' age, like the bench's), then times plv_lk_track of the last pair under the variant knobs (plv_debug_knobs: 0 = the default,
lk_kernel<1>; 4 = PLV_KNOB_LK_AHEAD >> 21, lk_ahead_kernel (round 6 experiment); 1 = PLV_KNOB_LK_LEGACY_LOOP >> 21, the loop of rounds
2-4) and checks that every variant returns the same bits.

usage: python tools/lk_exp.py [workload] [variants, comma separated: knob masks shifted right by 21]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench

wl_name = sys.argv[1] if len(sys.argv) > 1 else "C"
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4, 1]   # (knob mask >> 21) 0: lk_kernel<1> (default), 4: lk_ahead_kernel, 1: the loop of rounds 2-4
wl = bench.WORKLOADS[wl_name]
N_FRAMES = 8
stream = bench.build_stream(wl, N_FRAMES + 30, 8, os.environ.get("PLV_STREAM_CACHE"))      # (forks: before the GPU is touched; a profiled run loads the cache)
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
for v in variants:
    pkg.debug_knobs(v << SHIFT)
    for iters, lv in ((cfg.lk_max_iters, cfg.pyr_levels), (1, cfg.pyr_levels), (2, cfg.pyr_levels), (8, cfg.pyr_levels), (cfg.lk_max_iters, 1), (1, 1)):
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
        print(f"variant {v}: max_iters {iters:2d} levels {lv}: lk_kernel {t[1] / t[0] * 1e3:6.1f} us, iterations/pt mean {out[2].mean():.1f} max {out[2].max()}")
        x.close()
pkg.debug_knobs(0)
