"""GPU parity tests of K4/K5 through plv_perform_detection: HIP vs the CPU oracle.  FAST scores, NMS,
per-cell top-k and the id assignment are integer-exact; the sub-pixel refinement adds its normal equations up in raster order
on both sides, so the positions are compared bit for bit."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def _setup(pkg, w, h, seed, **cfgkw):
    cfg = pkg.default_config(w, h)
    for k, v in cfgkw.items():
        setattr(cfg, k, v)
    c = pkg.Context(cfg)
    canvas = synth.texture_canvas(w, h, seed=seed)
    img = synth.render_frame(canvas, w, h)
    c.feed_image(img)
    eq = oracle_lib.load_front().equalize_hist(img)
    return c, cfg, eq


def _match(a, b, tol=0):
    assert len(a[0]) == len(b[0]), (len(a[0]), len(b[0]))
    assert np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.max(np.abs(a[0] - b[0])) <= tol, np.max(np.abs(a[0] - b[0]))


@pytest.mark.parametrize("w,h,kw", [(752, 480, {}), (1280, 720, dict(num_features=500)),
                                    (752, 480, dict(grid_x=15, grid_y=15, num_features=1500, min_px_dist=15, fast_threshold=30)),
                                    (640, 400, dict(num_features=20, grid_x=8, grid_y=5))])
def test_initial_detection_parity(pkg, w, h, kw):
    c, cfg, eq = _setup(pkg, w, h, 42, **kw)
    do = oracle_lib.load_detect()
    e = (np.zeros((0, 2), np.float32), np.zeros(0, np.uint64))
    a = do.perform_detection(eq, None, e[0], e[1], 0, cfg.num_features, cfg.grid_x, cfg.grid_y, cfg.min_px_dist, cfg.fast_threshold)
    b = c.perform_detection(0, e[0], e[1], 0)
    assert len(b[0]) > 10
    _match(a, b)
    c.close()


def test_topup_with_existing_points_and_mask(pkg):
    w, h = 752, 480
    c, cfg, eq = _setup(pkg, w, h, 7)
    do = oracle_lib.load_detect()
    rng = np.random.default_rng(3)
    pts = np.column_stack([rng.uniform(0, w, 90), rng.uniform(0, h, 90)]).astype(np.float32)
    pts[:3] = [[4.0, 4.0], [w - 3.0, 50.0], [100.0, h - 2.0]]  # edge points are dropped
    ids = np.arange(100, 190, dtype=np.uint64)
    mask = np.zeros((h, w), np.uint8)
    mask[300:, 500:] = 255
    a = do.perform_detection(eq, mask, pts, ids, 500, cfg.num_features, cfg.grid_x, cfg.grid_y, cfg.min_px_dist, cfg.fast_threshold)
    b = c.perform_detection(0, pts, ids, 500, mask=mask)
    _match(a, b)
    assert b[2] > 500 and not ((b[0][:, 0] >= 500) & (b[0][:, 1] >= 300) & (b[1] > 500)).any()
    # enough features already: clean-up only, the id counter does not move
    full = b
    a2 = do.perform_detection(eq, None, full[0], full[1], full[2], len(full[0]), 5, 5, 10, 20)
    cfg2 = pkg.default_config(w, h)
    cfg2.num_features = len(full[0])
    c2 = pkg.Context(cfg2)
    c2.feed_image(np.zeros((h, w), np.uint8))  # any image: no extraction happens
    b2 = c2.perform_detection(0, full[0], full[1], full[2])
    _match(a2, b2, tol=0)
    c.close()
    c2.close()


def test_detection_needs_a_pyramid(pkg):
    c = pkg.Context(pkg.default_config(320, 240))
    with pytest.raises(pkg.PlvError):
        c.perform_detection(0, np.zeros((0, 2), np.float32), np.zeros(0, np.uint64), 0)
    c.close()
