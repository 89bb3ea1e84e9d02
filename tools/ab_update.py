#!/usr/bin/env python3
"""A/B of the update routes inside ONE process, alternating per iteration (run-to-run and box-to-box drift is larger than the
difference): SURVEY 8(d)'s update sizing (F = 70 features x 15 observations, then L = 80 lines x 15) through
plv_build_jacobians_resident + plv_msckf_update_resident, once per compression mode given on the command line.
usage: python3 tools/ab_update.py [iterations] [modes, e.g. 0,1]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("plviwo_amd", os.path.join(ROOT, "pl-viwo_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
sys.modules["plviwo_amd"] = pkg
spec.loader.exec_module(pkg)
import bench_chain as bc  # noqa: E402
import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
modes = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
ctx = pkg.Context(pkg.default_config(752, 480))
scene = synth.vio_scene(n_clones=15, F=bc.F_FEATS, M=bc.M_OBS, seed=3, noise_px=0.4)
st, tr = synth.scene_views(pkg, scene)
cols = ctx.jacobian_columns(st, tr)
ls = synth.line_scene(scene, L=bc.N_LINES, M=bc.M_OBS, noise_px=0.4)
lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"])
cols_l = ctx.line_jacobian_columns(st, lt)
n = scene["n_state"]
ctx.cov_upload(synth.spd_cov(n))
ctx.cov_checkpoint()
tp = {m: [] for m in modes}
tl = {m: [] for m in modes}
for it in range(iters + 10):
    for m in modes:
        ctx.update_compression_mode(m)
        ctx.cov_rollback()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.build_jacobians_resident(st, tr, cols, 2 * bc.M_OBS)
        rc, dx, a, nr = ctx.msckf_update_resident(n, bc.SIGMA2)
        t1 = time.perf_counter()
        ctx.build_line_jacobians_resident(st, lt, cols_l, bc.LINE_LD)
        rc2, dx2, a2, nr2 = ctx.msckf_update_resident(n, bc.SIGMA2, res_norm_gate=0.0)
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0
        if it >= 10:
            tp[m].append((t1 - t0) * 1e6)
            tl[m].append((t2 - t1) * 1e6)
for m in modes:
    print(f"mode {m}: points update mean {np.mean(tp[m]):7.1f} us  p50 {np.percentile(tp[m], 50):7.1f}   lines update mean {np.mean(tl[m]):7.1f} us  "
          f"p50 {np.percentile(tl[m], 50):7.1f}   ({iters} iterations, accepted {int(a.sum())} / {int(a2.sum())})")
ctx.close()
