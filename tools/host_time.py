"""Host-side wall time of each call of the bench step (sequential schedule), to separate launch / binding overhead from GPU time."""
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
    ctx = pkg.Context(cfg)
    frames, pts, P, scene = bench.build_inputs()
    st, tr = synth.scene_views(pkg, scene)
    cols = ctx.jacobian_columns(st, tr)
    ctx.image_stage(0, frames[0])
    ctx.image_stage(1, frames[1])
    ctx.cov_upload(P)
    ctx.cov_checkpoint()
    ctx.feed_staged(0)
    names = ["feed_staged", "perform_matching", "build_jacobians_resident", "cov_rollback", "update_launch", "update_wait", "sync"]
    acc = np.zeros(len(names))
    N = 300
    for i in range(N + 20):
        ts = [time.perf_counter()]
        ctx.feed_staged((i + 1) & 1); ts.append(time.perf_counter())
        ctx.perform_matching(pts, pts); ts.append(time.perf_counter())
        ctx.build_jacobians_resident(st, tr, cols, 2 * bench.M_OBS); ts.append(time.perf_counter())
        ctx.cov_rollback(); ts.append(time.perf_counter())
        ctx.msckf_update_resident_launch(bench.SIGMA2); ts.append(time.perf_counter())
        ctx.msckf_update_resident_wait(bench.N_STATE); ts.append(time.perf_counter())
        ctx.synchronize(); ts.append(time.perf_counter())
        if i >= 20:
            acc += np.diff(ts)
    for n, v in zip(names, acc / N * 1e6):
        print(f"{n:28s} {v:8.1f} us")
    print(f"{'total':28s} {acc.sum() / N * 1e6:8.1f} us")
    # the same with the GPU drained after every call: pure enqueue cost shows up as (call) and GPU time as (sync after)
    acc2 = np.zeros((len(names) - 1, 2))
    calls = [lambda i: ctx.feed_staged((i + 1) & 1), lambda i: ctx.perform_matching(pts, pts),
             lambda i: ctx.build_jacobians_resident(st, tr, cols, 2 * bench.M_OBS), lambda i: ctx.cov_rollback(),
             lambda i: ctx.msckf_update_resident_launch(bench.SIGMA2), lambda i: ctx.msckf_update_resident_wait(bench.N_STATE)]
    for i in range(N):
        for j, c in enumerate(calls):
            t0 = time.perf_counter(); c(i); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
            acc2[j] += (t1 - t0, t2 - t1)
    print("drained after every call:  call / sync-after")
    for n, v in zip(names, acc2 / N * 1e6):
        print(f"{n:28s} {v[0]:8.1f} {v[1]:8.1f} us")


if __name__ == "__main__":
    main()
