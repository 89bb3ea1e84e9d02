#!/usr/bin/env python3
"""Where do the two FRONT-ENDS part on a drive?  (test infrastructure: it runs the oracle; needs a GPU)

The trackers of the HIP library and of the CPU oracle are bit-identical on identical inputs, but with the intrinsics calibrated
online their inputs are not identical: TrackKLT undistorts with the state's current intrinsics before its fundamental-matrix RANSAC
(REF: TrackKLT.cpp perform_matching; UpdaterCamera.cpp sets the calibration every frame).  This tool replays a rendered drive through
the driver over both, records the tracker's id list and the filter's intrinsics after every camera frame, and prints the first
frame whose id lists differ with the intrinsics of both filters at that frame (the one feature of the 300 s record).

    python tools/debug/track_split.py [--seconds 300] [--size 752x480] [--hz 15] [--points 360]
"""
import argparse
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import oracle_context as oc  # noqa: E402
import synth_dataset as sd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--size", default="752x480")
    ap.add_argument("--hz", type=float, default=15.0)
    ap.add_argument("--points", type=int, default=360)
    ap.add_argument("--style", default="avenue")
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    sd.set_camera(W, H)
    ge.load_pkg()
    options, rp, system = (importlib.import_module("plviwo_amd." + m) for m in ("options", "replay", "system"))
    d = tempfile.mkdtemp(prefix="plv_tsplit_")
    sd.make_dataset(d, a.seconds, cam_hz=a.hz, style=a.style, workers=min(16, os.cpu_count() or 1))
    runs = {}
    for name, kw in (("hip", {}), ("cpu", dict(context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer))):
        op = options.load_options(sd.write_config(os.path.join(d, "config"), d, os.path.join(d, f"traj_{name}.txt"), clone_freq=int(a.hz), n_pts=a.points,
                                                  max_msckf=70, calib_int=True, sigma_px=1.5))
        op.est.cam.use_lines = True
        rec = []
        init = system.SystemManager.__init__

        def init2(self, *args, _init=init, _rec=rec, **kws):
            _init(self, *args, **kws)
            frame = self.ctx.camera_frame

            def camera_frame(st, t, *fa, _frame=frame, _self=self, **fk):
                intr_before = np.array(_self.state.cam_intr.v, float) if _self.state.cam_intr is not None else np.zeros(8)
                out = _frame(st, t, *fa, **fk)
                pts, ids = _self.ctx.tracker_last()
                _rec.append((t, np.array(ids, dtype=np.uint64), np.array(pts, dtype=np.float32).reshape(-1, 2), intr_before))
                return out
            self.ctx.camera_frame = camera_frame
        system.SystemManager.__init__ = init2
        try:
            rp.replay(op, **kw)
        finally:
            system.SystemManager.__init__ = init
        runs[name] = rec
    h, c = runs["hip"], runs["cpu"]
    print(f"{len(h)} / {len(c)} camera frames recorded")
    worst = 0.0
    for k, ((ta, ia, pa, ka), (tb, ib, pb, kb)) in enumerate(zip(h, c)):
        worst = max(worst, float(np.abs(ka - kb).max()))
        if np.array_equal(ia, ib) and np.array_equal(pa, pb):
            continue
        sa, sb = set(int(v) for v in ia), set(int(v) for v in ib)
        print(f"first frame whose tracker output differs: #{k}, t = {ta:.4f}: {len(ia)} / {len(ib)} points; ids in the library's list only "
              f"{sorted(sa - sb)[:8]}, in the oracle's only {sorted(sb - sa)[:8]}; common points with different positions "
              f"{sum(1 for i, p in zip(ia, pa) if int(i) in sb and not np.array_equal(p, pb[list(ib).index(i)]))}")
        np.set_printoptions(precision=12, linewidth=200)
        print("  intrinsics the frame was tracked with, library:", ka)
        print("  intrinsics the frame was tracked with, oracle: ", kb)
        print("  difference:", ka - kb, " (largest over the frames before: %.3g)" % worst)
        if k:
            same_before = np.array_equal(h[k - 1][1], c[k - 1][1]) and np.array_equal(h[k - 1][2], c[k - 1][2])
            print("  the frame before: id lists and positions", "identical" if same_before else "DIFFERENT")
        break
    else:
        print("the trackers' id lists and positions are identical in every frame; largest difference of the intrinsics %.3g" % worst)


if __name__ == "__main__":
    main()
