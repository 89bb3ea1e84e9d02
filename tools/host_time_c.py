"""Host-side wall time of each call of the workload-C bench step (pipelined schedule, two contexts)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import synth  # noqa: E402


def main():
    pkg = ge.load_pkg()
    cfg = pkg.default_config(bench.W, bench.H)
    ctx, uctx = pkg.Context(cfg), pkg.Context(cfg)
    frames, pts, P, scene = bench.build_inputs(True)
    st, tr = synth.scene_views(pkg, scene)
    cols = ctx.jacobian_columns(st, tr)
    ls = synth.line_scene(scene, L=bench.N_LINES, M=bench.M_OBS, noise_px=0.4)
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"])
    cols_l = ctx.line_jacobian_columns(st, lt)
    vps = ctx.vanishing_points(scene["R_ItoC"], scene["K8"])
    ids = np.arange(1, bench.N_PTS + 1, dtype=np.uint64)
    ctx.image_stage(0, frames[0]); ctx.image_stage(1, frames[1])
    uctx.cov_upload(P); uctx.cov_checkpoint()
    ctx.feed_staged(0)
    names = ["U rollback", "U build_jacobians", "U update_launch (points)", "F feed_staged", "F line_detect_launch", "F matching_launch",
             "F line_detect_finish", "F matching_wait", "U update_wait (points)", "U build_line_jacobians", "U update_launch (lines)",
             "F line_tracker_feed_points", "U update_wait (lines)"]
    acc = np.zeros(len(names))
    N = 300
    for i in range(N + 20):
        ts = [time.perf_counter()]
        def tick():
            ts.append(time.perf_counter())
        uctx.cov_rollback(); tick()
        uctx.build_jacobians_resident(st, tr, cols, 2 * bench.M_OBS); tick()
        uctx.msckf_update_resident_launch(bench.SIGMA2); tick()
        ctx.feed_staged((i + 1) & 1); tick()
        ctx.line_detect_launch(0); tick()
        ctx.perform_matching_launch(pts, pts); tick()
        ctx.line_detect_finish(0); tick()
        out = ctx.perform_matching_wait(); tick()
        uctx.msckf_update_resident_wait(bench.N_STATE); tick()
        uctx.build_line_jacobians_resident(st, lt, cols_l, bench.LINE_LD); tick()
        uctx.msckf_update_resident_launch(bench.SIGMA2, res_norm_gate=0.0); tick()
        ctx.line_tracker_feed_points(float(i), vps, out[0], ids); tick()
        uctx.msckf_update_resident_wait(bench.N_STATE); tick()
        if i % 15 == 14:
            ctx.line_db_remove(ctx.line_db_ids())
        if i >= 20:
            acc += np.diff(ts)
    for n, v in zip(names, acc / N * 1e6):
        print(f"{n:32s} {v:8.1f} us")
    print(f"{'total':32s} {acc.sum() / N * 1e6:8.1f} us")


if __name__ == "__main__":
    main()
