#!/usr/bin/env python3
"""Where do the HIP library and the CPU oracle part on a drive?  (test infrastructure: it runs the oracle; needs a GPU)

Replays a rendered synthetic drive through the replay driver twice — over the HIP library and over the compiled CPU frame — with the
driver's decision trace and state probes on, and prints
  * the state and covariance difference in front of every camera update (the two filters agree to 1e-11 until something happens),
  * the first update whose dx differs by more than --dx-tol of its largest entry, with the update before it,
  * the largest relative difference of a recorded test value per point update (tests/decision_trace.py value_drift).
This is the tool behind DESIGN §10.4: it found the two window poses at one instant, the factor form's eps x lambda^2 and the
refinement's termination tests.

    python tools/debug/decision_drift.py [--seconds 4] [--size 1280x720] [--hz 20] [--points 780] [--mode 1]
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
import decision_trace as dt  # noqa: E402
import oracle_context as oc  # noqa: E402
import synth_dataset as sd  # noqa: E402


Trace = dt.ProbedTrace


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("--points", type=int, default=780)
    ap.add_argument("--style", default="avenue")
    ap.add_argument("--mode", type=int, default=None, help="plv_update_compression_mode of the HIP run (1 = Householder route)")
    ap.add_argument("--dx-tol", type=float, default=1e-6)
    ap.add_argument("--pivots-from", type=int, default=None, help="print the smallest pivot in front of the 24 camera updates from this frame on, "
                    "and after every step (updates, propagation + cloning + marginalisation) of the first --pivot-frames of them")
    ap.add_argument("--pivot-frames", type=int, default=3)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    sd.set_camera(W, H)
    pkg = ge.load_pkg()
    options, rp, system = (importlib.import_module("plviwo_amd." + m) for m in ("options", "replay", "system"))
    d = tempfile.mkdtemp(prefix="plv_drift_")
    sd.make_dataset(d, a.seconds, cam_hz=a.hz, style=a.style, workers=min(16, os.cpu_count() or 1))
    runs = {}
    for name, kw in (("hip", {}), ("cpu", dict(context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer))):
        op = options.load_options(sd.write_config(os.path.join(d, "config"), d, os.path.join(d, f"traj_{name}.txt"), clone_freq=int(a.hz), n_pts=a.points,
                                                  max_msckf=70, calib_int=True, sigma_px=1.5))
        op.est.cam.use_lines = True
        tr = Trace()
        if a.pivots_from is not None:
            tr.probe_cov = (a.pivots_from, a.pivots_from + a.pivot_frames)
        init = system.SystemManager.__init__
        if name == "hip" and a.mode is not None:
            def init2(self, *args, _init=init, **kws):
                _init(self, *args, **kws)
                self.ctx.update_compression_mode(a.mode)
            system.SystemManager.__init__ = init2
        try:
            rp.replay(op, decisions=tr, **kw)
        finally:
            system.SystemManager.__init__ = init
        runs[name] = tr
    h, c = runs["hip"], runs["cpu"]
    print("state / covariance difference in front of the camera updates (frame: largest state difference @ component, covariance relative):")
    for (fa, xa, Pa), (fb, xb, Pb) in zip(h.states_pre, c.states_pre):
        if len(xa) != len(xb):
            print(f"  frame {fa}: the state vectors differ in length ({len(xa)} / {len(xb)})")
            break
        dx = np.abs(xa - xb)
        print("  %d: %.2g @ %d, %.2g;" % (fa, dx.max(), int(np.argmax(dx)), np.abs(Pa - Pb).max() / np.abs(Pb).max()), end="")
    print()

    min_pivot = dt.min_unit_pivot
    print("smallest unit-diagonal Cholesky pivot of the covariance in front of every 10th camera update (frame: hip / cpu):")
    for (fa, xa, Pa), (fb, xb, Pb) in list(zip(h.states_pre, c.states_pre))[::10]:
        print("  %d: %.2g / %.2g;" % (fa, min_pivot(Pa), min_pivot(Pb)), end="")
    print()
    if a.pivots_from is not None:
        print(f"... and in front of every update from frame {a.pivots_from} on:")
        for (fa, xa, Pa), (fb, xb, Pb) in zip(h.states_pre, c.states_pre):
            if a.pivots_from <= fa < a.pivots_from + 24:
                print("  %d: %.3g / %.3g;" % (fa, min_pivot(Pa), min_pivot(Pb)), end="")
        print()
        def pivots_in_order(P, order):
            d = np.sqrt(np.abs(np.diag(P)))
            A = (P / np.outer(d, d))[np.ix_(order, order)].copy()
            out = []
            for j in range(len(A)):
                piv = A[j, j]
                out.append(piv)
                if piv <= 2e-13:
                    A[j + 1:, j] = 0.0
                    continue
                A[j + 1:, j] /= piv
                A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j + 1:, j]) * piv
            return np.array(out)
        for (fa, xa, Pa), (fb, xb, Pb) in zip(h.states_pre, c.states_pre):
            if fa == a.pivots_from + 1:
                n = len(Pa)
                order = list(range(15, n)) + list(range(0, 6))       # the update's column order: calibration, clones, the IMU pose
                np.set_printoptions(precision=2, linewidth=220)
                print("  pivots of the prior block in the update's column order in front of frame %d, numpy fp64 (dead ones eliminated as the library does):" % fa)
                print("   hip P:", pivots_in_order(Pa, order))
                print("   cpu P:", pivots_in_order(Pb, order))
                print("   asymmetry of the hip P: %.3g, of the cpu P: %.3g (largest |P - P^T| / largest |P|)" % (np.abs(Pa - Pa.T).max() / np.abs(Pa).max(), np.abs(Pb - Pb.T).max() / np.abs(Pb).max()))
        for (la, fa, ta, Pa), (lb, fb, tb, Pb) in zip(h.cov_probes, c.cov_probes):
            print("  frame %d, t = %.4f, %s: %.3g / %.3g%s" % (fa, ta, la, min_pivot(Pa), min_pivot(Pb), "" if (la, fa) == (lb, fb) else "   (cpu: %s of frame %d)" % (lb, fb)))
    for k, (ra, rb) in enumerate(zip(h, c)):
        da, db = ra[8], rb[8]
        m = max(np.abs(da).max(), np.abs(db).max(), 1e-300)
        rel = np.abs(da - db).max() / m
        if rel > a.dx_tol or ra[0] != rb[0] or int(ra[5].sum()) != int(rb[5].sum()):
            print(f"first update whose dx differs by more than {a.dx_tol:g}: #{k} ({ra[0]}, frame {ra[1]}): accepted {int(ra[5].sum())} / {int(rb[5].sum())}, "
                  f"largest |dx| {m:.3g}, difference {np.abs(da - db).max():.3g} ({rel:.3g} relative)")
            if k:
                pa, pb = h[k - 1], c[k - 1]
                print(f"   the update before: {pa[0]} of frame {pa[1]}, dx difference {np.abs(pa[8] - pb[8]).max():.3g}")
            break
    else:
        print(f"no update's dx differs by more than {a.dx_tol:g} of its largest entry")
    drift = dt.value_drift(h, c, dt.thresholds(op))
    print("value drift (update, largest relative difference of a recorded test value, which, feature):")
    print("  first:", drift[:5])
    print("  every 20th:", drift[::20])
    print("decisions:", {k: v for k, v in dt.summary(h, c, thr=dt.thresholds(op)).items() if k in ("updates", "updates_with_identical_decisions", "first_divergence", "tie_check")})


if __name__ == "__main__":
    main()
