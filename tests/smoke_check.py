"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
import synth  # noqa: E402


def run():
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    orc = oracle_lib.load()
    ctx = pkg.Context()
    n, k = 113, 98
    P = synth.spd_cov(n)
    cols = synth.col_map(n, k)
    rows, Hf, Hx, res = synth.msckf_batch(F=20, M=15, k=k, seed=1)
    rc0, P0, dx0, acc0, _ = orc.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table())
    rc1, P1, dx1, acc1, _ = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25)
    assert rc0 == 0 and rc1 == 0, (rc0, rc1)
    assert np.array_equal(acc0, acc1)
    err = np.max(np.abs(P1 - P0)) / np.max(np.abs(P0))
    assert err < 1e-8, err
    assert np.max(np.abs(dx1 - dx0)) / np.max(np.abs(dx0)) < 1e-8
    print(f"smoke: msckf update parity ok (rel err P {err:.2e}, accepted {int(acc1.sum())}/{len(rows)})")
    # one tracked frame: hist-eq + pyramid + LK + undistort + RANSAC, bit-exact LK vs the oracle
    fo = oracle_lib.load_front()
    W, H = 752, 480
    canvas = synth.texture_canvas(W, H, seed=42, blobs=150)
    f0, f1 = synth.render_frame(canvas, W, H), synth.render_frame(canvas, W, H, tx=2.5, ty=-1.5, rot_deg=0.2)
    pts0 = synth.grid_points(W, H, 120)
    ctx.feed_image(f0)
    ctx.feed_image(f1)
    pts1, mask, n0, n1, iters = ctx.perform_matching(pts0, pts0)
    p0, p1 = fo.pyramid(fo.equalize_hist(f0)), fo.pyramid(fo.equalize_hist(f1))
    ref1, st, it = fo.lk_track(p0, p1, pts0, pts0)
    assert np.array_equal(pts1[st > 0], ref1[st > 0]) and it == iters, (it, iters)
    assert mask.sum() > 80
    print(f"smoke: tracked frame ok (LK bit-exact on {int((st > 0).sum())} points, {int(mask.sum())} RANSAC inliers)")
    ctx.close()
