#!/usr/bin/env python3
"""(debug) the line entries of one update in the HIP run and in the CPU-oracle run of a replay: ids, verdicts, gate values, Pluecker lines.
    python tools/debug/line_case.py --seconds 37 --update 1597"""
import argparse, importlib, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, oracle_context as oc, synth_dataset as sd
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=37.0)
ap.add_argument("--update", type=int, default=1597)
a = ap.parse_args()
sd.set_camera(752, 480)
pkg = ge.load_pkg()
options, rp = importlib.import_module("plviwo_amd.options"), importlib.import_module("plviwo_amd.replay")
d = tempfile.mkdtemp(prefix="plv_line_")
sd.make_dataset(d, a.seconds, cam_hz=15, style="avenue", workers=min(16, os.cpu_count() or 1))
runs = {}
for name, kw in (("hip", {}), ("cpu", dict(context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer))):
    op = options.load_options(sd.write_config(os.path.join(d, "config"), d, os.path.join(d, f"t_{name}.txt"), clone_freq=15, n_pts=360, max_msckf=70, calib_int=True, sigma_px=1.5))
    op.est.cam.use_lines = True
    dec = []
    rp.replay(op, decisions=dec, **kw)
    runs[name] = dec
np.set_printoptions(precision=12, linewidth=220)
rh, rc = runs["hip"][a.update], runs["cpu"][a.update]
print("update", a.update, rh[0], "frame", rh[1], "pool", rh[3], rc[3], "batch", len(rh[4]), len(rc[4]), "ids equal", np.array_equal(rh[4], rc[4]))
for q, lid in enumerate(rh[4]):
    j = list(rc[4]).index(lid) if lid in rc[4] else -1
    if j < 0:
        print(int(lid), "hip only")
        continue
    gh, gc = rh[7][1][q], rc[7][1][j]
    lh, lc = rh[9][q], rc[9][j]
    flag = "   <--" if rh[5][q] != rc[5][j] else ""
    print(int(lid), "accepted", int(rh[5][q]), int(rc[5][j]), "chi2 %.6g / %.6g  thr %.4g  |r| %.6g / %.6g  line diff %.3g%s" % (gh[0], gc[0], gh[1], gh[2], gc[2], np.abs(lh - lc).max(), flag))
    if flag:
        print("    hip line", lh); print("    cpu line", lc)
