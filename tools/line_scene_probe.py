#!/usr/bin/env python3
"""CPU-only probe of the line half on a rendered scene (no GPU): renders `frames` camera frames of a scene / mount of
tests/synth_dataset.py, runs them through the CPU oracle's frame (oracle/frame_oracle.cpp behind tests/oracle_context.py) and prints,
per frame on average: tracked points, segments kept by AssignPointToLines, line pool, triangulated, accepted — and the line classes.
usage: python tools/line_scene_probe.py [scene] [pitch] [yaw] [frames] [n_pts]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "boulevard"
    pitch = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
    yaw = float(sys.argv[3]) if len(sys.argv) > 3 else 90.0
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 120
    n_pts = int(sys.argv[5]) if len(sys.argv) > 5 else 440
    import synth_dataset as sd
    if os.environ.get("PROBE_BOULEVARD"):      # e.g. PROBE_BOULEVARD="angle=0,wall_r=6.5": overrides of the boulevard texture's geometry
        for kv in os.environ["PROBE_BOULEVARD"].split(","):
            k, v = kv.split("=")
            sd.BOULEVARD[k] = float(v)
    wl = dict(w=752, h=480, hz=15, points=250, num_features=n_pts, lines=True, cfg="probe", scene=scene, mount=(pitch, yaw))
    t0 = time.time()
    stream = bench.build_stream(wl, bench.PROLOGUE + frames, min(8, os.cpu_count() or 1))
    print(f"rendered {len(stream['imgs'])} frames in {time.time() - t0:.1f} s", flush=True)
    import __graft_entry__ as ge
    ge.load_pkg()
    import oracle_context as oc
    system = importlib.import_module("plviwo_amd.system")
    dec = []
    sm = system.SystemManager(bench.load_options(wl), context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer, decisions=dec)
    pl = bench.Player(stream, sm, staged=False)
    tracked, kept = [], []
    cls_hist, len_hist, pts_hist = np.zeros(4), np.zeros(40), np.zeros(40)
    base = None
    for f in range(bench.PROLOGUE + frames):
        nf = pl.next_frame()
        if nf is None:
            break
        pl.camera(*nf)
        if f == bench.PROLOGUE - 1:
            base = dict(sm.stats)
        if f >= bench.PROLOGUE:
            tracked.append(len(sm.ctx.tracker_last()[1]))
            kept.append(len(sm.ctx.line_tracker_last()[1]))
            if f % 5 == 0:   # the live line tracks: class, length, points on the line
                for lid in sm.ctx.line_tracker_last()[1]:
                    tr = sm.ctx.frame.line_db_track(int(lid))
                    if tr is None:
                        continue
                    cls_hist[int(tr[3])] += 1
                    len_hist[min(39, len(tr[0]))] += 1
                    pts_hist[min(39, int(tr[4]))] += 1
    st = {k: sm.stats[k] - base.get(k, 0) for k in sm.stats}
    n = max(1, len(tracked))
    print(f"scene {scene} mount {pitch}/{yaw}: per frame over {n} frames: tracked {np.mean(tracked):.1f}  kept {np.mean(kept):.1f}  "
          f"line pool {st['line_pool'] / n:.1f}  triangulated {st['lines_triangulated'] / n:.1f}  accepted {st['lines_accepted'] / n:.2f}  "
          f"line updates {st['line_updates']}  msckf {st['cam_features'] / n:.1f} / accepted {st['cam_accepted'] / n:.1f}  not_psd {st['not_psd']}")
    print("live line tracks: class D = 0/1/2/3 shares", np.round(cls_hist / max(1, cls_hist.sum()), 3), " observations per track 1/2/3/4/5+:",
          np.round(np.array([len_hist[1], len_hist[2], len_hist[3], len_hist[4], len_hist[5:].sum()]) / max(1, len_hist.sum()), 3),
          " points per track 0/1/2/3+:", np.round(np.array([pts_hist[0], pts_hist[1], pts_hist[2], pts_hist[3:].sum()]) / max(1, pts_hist.sum()), 3))
    # gate values of the triangulated lines (decision records: ("lines", frame, t, n_pool, ids, accepted, status, vals, dx, line_FinG))
    chi, thr = [], []
    for d in dec:
        if d[0] == "lines" and d[7] is not None:
            ids, vals = d[7]
            v = np.asarray(vals).reshape(-1, 3)
            m = ~np.isnan(v[:, 0])
            chi += list(v[m, 0])
            thr += list(v[m, 1])
    if chi:
        chi, thr = np.array(chi), np.array(thr)
        r = chi / thr
        print(f"gate: {len(chi)} lines reached it; chi2 / threshold quantiles 10/50/90 %: {np.percentile(r, 10):.3g} {np.percentile(r, 50):.3g} {np.percentile(r, 90):.3g}; "
              f"passed {np.mean(r < 1) * 100:.1f} %")
    sm.close()


if __name__ == "__main__":
    main()
