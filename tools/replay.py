#!/usr/bin/env python3
"""Replays a dataset directory through the filter on the GPU and scores the trajectory (run_bag + ov_eval in one command).

    python tools/replay.py CONFIG.yaml [--dataset DIR] [--gt FILE] [--align posyaw] [--out result.json]
    python tools/replay.py --synthetic 12 [--out result.json]        # renders tests/synth_dataset.py first (CPU), then replays it
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?")
    ap.add_argument("--dataset")
    ap.add_argument("--gt")
    ap.add_argument("--align", default="posyaw")
    ap.add_argument("--out")
    ap.add_argument("--synthetic", type=float, default=0.0, help="seconds of the synthetic dataset to render and replay")
    ap.add_argument("--no-wheel", action="store_true")
    ap.add_argument("--no-lines", action="store_true")
    ap.add_argument("--keep", help="directory to keep the synthetic dataset in")
    a = ap.parse_args()
    pkg = ge.load_pkg()
    options = importlib.import_module("plviwo_amd.options")
    rp = importlib.import_module("plviwo_amd.replay")
    tmp = None
    if a.synthetic > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import synth_dataset as sd
        tmp = a.keep or tempfile.mkdtemp(prefix="plv_synth_")
        t0 = time.time()
        sd.make_dataset(tmp, a.synthetic, log=lambda s: print("  [render]", s, flush=True))
        a.config = sd.write_config(os.path.join(tmp, "config"), tmp, os.path.join(tmp, "out", "traj.txt"), use_wheel=not a.no_wheel)
        a.gt = os.path.join(tmp, "gt.txt")
        print(f"synthetic dataset in {tmp} ({time.time() - t0:.1f} s)", flush=True)
    op = options.load_options(a.config)
    if a.dataset:
        op.sys.path_bag = a.dataset
    if a.no_lines:
        op.est.cam.use_lines = False
    if not op.sys.save_trajectory:
        op.sys.save_trajectory, op.sys.path_trajectory = True, os.path.join(tempfile.mkdtemp(prefix="plv_out_"), "traj.txt")
    t0 = time.time()
    stats, times, poses = rp.replay(op, progress=lambda s, t: print(f"  t={t:8.3f}  clones {s.stats['clones']}  cam accepted "
                                                                    f"{s.stats['cam_accepted']}/{s.stats['cam_features']}  wheel {s.stats['wheel_accepted']}", flush=True))
    res = dict(config=a.config, wall_s=round(time.time() - t0, 2), stats=stats, poses_logged=len(times), trajectory=op.sys.path_trajectory)
    if a.gt and len(times) > 2:
        ctx = pkg.Context(pkg.default_config(752, 480))
        et, ep = pkg.traj_load(op.sys.path_trajectory)[:2]
        gt_t, gt_p = pkg.traj_load(a.gt)[:2]
        ei, gi = pkg.traj_associate(et, gt_t)
        r = ctx.traj_ate(ep[ei], gt_p[gi], a.align)
        res["ate"] = dict(method=a.align, n=len(ei), pos=r["pos"], ori=r["ori"], length_m=pkg.traj_length(ep))
        ctx.close()
    print(json.dumps(res, indent=1, default=float))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1, default=float)


if __name__ == "__main__":
    main()
