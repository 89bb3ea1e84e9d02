"""GPU test of the tracker frame logic (TrackKLT::feed_monocular mirror) + feature database over a
synthetic 752x480 stream, against the same frame logic composed from the CPU oracle pieces."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


class OracleTracker:
    """TrackKLT::feed_monocular restated on top of the oracle (REF: TrackKLT.cpp:96-200)."""

    def __init__(self, cfg, K8):
        self.cfg, self.K8 = cfg, K8
        self.fo, self.do = oracle_lib.load_front(), oracle_lib.load_detect()
        self.pts = np.zeros((0, 2), np.float32)
        self.ids = np.zeros(0, np.uint64)
        self.currid = 0
        self.prev = None
        self.db = {}

    def _detect(self, eq, pts, ids):
        c = self.cfg
        p, i, self.currid = self.do.perform_detection(eq, None, pts, ids, self.currid, c.num_features, c.grid_x, c.grid_y,
                                                      c.min_px_dist, c.fast_threshold)
        return p, i

    def feed(self, t, img):
        eq = self.fo.equalize_hist(img)
        pyr = self.fo.pyramid(eq)
        if len(self.ids) == 0:
            self.pts, self.ids = self._detect(eq, self.pts, self.ids)
            self.prev = (eq, pyr)
            return
        pts_old, ids_old = self._detect(self.prev[0], self.pts, self.ids)
        rc, pts_new, mask, n0, n1 = self.fo.perform_matching(self.prev[1], pyr, pts_old, pts_old, self.K8)
        h, w = img.shape
        good, gid = [], []
        for i in range(len(pts_old)):
            x, y = pts_new[i]
            if x < 0 or y < 0 or int(x) >= w or int(y) >= h or not mask[i]:
                continue
            good.append(pts_new[i])
            gid.append(ids_old[i])
            self.db.setdefault(int(ids_old[i]), []).append((t, x, y, n1[i, 0], n1[i, 1]))
        self.pts = np.array(good, np.float32).reshape(-1, 2)
        self.ids = np.array(gid, np.uint64)
        self.prev = (eq, pyr)


def test_tracker_stream_matches_oracle(pkg):
    w, h = 752, 480
    canvas = synth.texture_canvas(w, h, seed=42)
    rng = np.random.default_rng(5)
    cfg = pkg.default_config(w, h)
    ctx = pkg.Context(cfg)
    ot = OracleTracker(cfg, np.array(list(cfg.intrinsics)))
    tx = ty = rot = 0.0
    for f in range(7):
        img = synth.render_frame(canvas, w, h, tx=tx, ty=ty, rot_deg=rot, scale=1.0 + 0.002 * f)
        t = 10.0 + 0.05 * f
        ctx.tracker_feed(t, img)
        ot.feed(t, img)
        pts, ids = ctx.tracker_last()
        # the whole front-end is bit-reproducible (integer image work, exact LK sums, ordered sub-pixel sums, deterministic RANSAC):
        # identical id lists and positions frame after frame
        assert np.array_equal(ids, ot.ids), (f, len(ids), len(ot.ids))
        assert np.array_equal(pts, np.asarray(ot.pts, dtype=np.float32).reshape(-1, 2)), f
        assert len(ids) >= 150
        tx += rng.uniform(-6, 6)
        ty += rng.uniform(-6, 6)
        rot += rng.uniform(-0.5, 0.5)
    # database: same track lengths for the common ids, same observations
    assert ctx.db_size() >= 200
    some = np.array(sorted(ot.db))[:50].astype(np.uint64)
    ptr, tt, uv, uvn = ctx.db_export(some)
    agree = 0
    for k, fid in enumerate(some):
        mine = list(zip(tt[ptr[k]:ptr[k + 1]], uv[ptr[k]:ptr[k + 1], 0]))
        ref = [(o[0], o[1]) for o in ot.db[int(fid)]]
        if len(mine) == len(ref) and all(m[0] == r[0] and m[1] == r[1] for m, r in zip(mine, ref)):
            agree += 1
    assert agree == len(some)
    # selection helpers
    newest = 10.0 + 0.05 * 6
    lost = ctx.db_select(0, newest)  # not seen in the newest frame
    assert all(int(i) not in {int(j) for j in ids} for i in lost)
    old = ctx.db_select(1, 10.0 + 0.05 * 2)
    assert len(old) > 0
    ctx.db_cleanup_measurements(10.0 + 0.05 * 3)
    ptr2, tt2, _, _ = ctx.db_export(old[:10])
    assert (tt2 >= 10.0 + 0.05 * 3 - 1e-12).all()
    n0 = ctx.db_size()
    ctx.db_remove(old[:5])
    assert ctx.db_size() <= n0
    ctx.close()


def test_downsampled_feed_matches_oracle(pkg):
    """OptionsCamera::downsample: pyrDown(Size(cols / 2.0, rows / 2.0)) of image and mask, then the usual path
    (REF: UpdaterCamera.cpp:85-98).  Odd source sizes halve downwards."""
    fo = oracle_lib.load_front()
    sw, sh = 1505, 961  # -> 752 x 480
    canvas = synth.texture_canvas(sw, sh, seed=3, blobs=500)
    cfg = pkg.default_config(sw // 2, sh // 2)
    ctx = pkg.Context(cfg)
    img0 = synth.render_frame(canvas, sw, sh)
    small = ctx.downsample(img0)
    assert small.shape == (480, 752) and np.array_equal(small, fo.downsample(img0))
    assert np.array_equal(ctx.downsample(img0[:100, :64]), fo.downsample(img0[:100, :64]))  # a single 16x16 tile column
    # the fed pyramid is the pyramid of the equalised halved image
    ctx.feed_image_downsampled(img0)
    ref = fo.pyramid(fo.equalize_hist(fo.downsample(img0)))
    for l in range(ref.levels):
        assert np.array_equal(ctx.pyramid_level(0, l), ref.level(l)[0]), l
    # tracker stream on the halved images == tracker stream fed the oracle's halved images; the mask is halved too
    ot = OracleTracker(cfg, np.array(list(cfg.intrinsics)))
    ctx2 = pkg.Context(cfg)
    mask = np.zeros((sh, sw), np.uint8)
    mask[:, : sw // 4] = 255
    small_mask = fo.downsample(mask)
    for f in range(3):
        img = synth.render_frame(canvas, sw, sh, tx=3.0 * f, ty=-2.0 * f)
        ctx.tracker_feed_downsampled(20.0 + 0.1 * f, img, mask)
        ctx2.tracker_feed(20.0 + 0.1 * f, fo.downsample(img), small_mask)
    p1, i1 = ctx.tracker_last()
    p2, i2 = ctx2.tracker_last()
    assert len(i1) > 100 and np.array_equal(i1, i2) and np.array_equal(p1, p2)
    assert (small_mask[p1[:, 1].astype(int), p1[:, 0].astype(int)] <= 127).all()
    ctx.close()
    ctx2.close()


def test_kaist_sized_tracker(pkg):
    """The shipped KAIST camera settings (n_pts 1500, 15x15 grid, FAST 20, min_px_dist 10; config_camera.yaml) on a 1280x720 stream."""
    w, h = 1280, 720
    canvas = synth.texture_canvas(w, h, seed=12, blobs=3000)
    cfg = pkg.default_config(w, h)
    cfg.num_features, cfg.grid_x, cfg.grid_y, cfg.min_px_dist, cfg.fast_threshold = 1500, 15, 15, 10, 20
    ctx = pkg.Context(cfg)
    ot = OracleTracker(cfg, np.array(list(cfg.intrinsics)))
    for f in range(3):
        img = synth.render_frame(canvas, w, h, tx=4.0 * f, ty=-3.0 * f, rot_deg=0.2 * f)
        ctx.tracker_feed(30.0 + 0.05 * f, img)
        ot.feed(30.0 + 0.05 * f, img)
    pts, ids = ctx.tracker_last()
    assert len(ids) > 900
    assert np.array_equal(ids, ot.ids)
    assert np.array_equal(pts, np.asarray(ot.pts, dtype=np.float32).reshape(-1, 2))
    ctx.close()
