import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, __graft_entry__ as ge, bench, synth
pkg = ge.load_pkg()
frames, pts, P, scene = bench.build_inputs()
for iters, lv in ((30, 5), (1, 5), (30, 0), (1, 0), (2, 5), (3, 5)):
    cfg = pkg.default_config(bench.W, bench.H)
    cfg.lk_max_iters, cfg.pyr_levels = iters, lv
    ctx = pkg.Context(cfg)
    ctx.feed_image(frames[0]); ctx.feed_image(frames[1])
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(50):
        out = ctx.lk_track(pts, pts)
    t = ctx.prof_table()["lk_kernel"]
    print(f"max_iters {iters:2d} max_level {lv}: lk_kernel {t[1] / t[0] * 1e3:6.1f} us, iterations/pt {out[2].sum() / len(pts):.1f}" if len(out) > 2 else out)
    ctx.close()
