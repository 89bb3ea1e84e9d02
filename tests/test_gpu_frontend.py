"""GPU parity tests (through the C-ABI) of the point front-end: HIP vs the CPU oracle.
Integer image work (equalize, pyramid) is compared bit-exactly.  LK positions are compared
bit-exactly too: the 2x2 normal-equation sums are exact integers in both implementations, and the
float tail is the same IEEE operation sequence (both built with -ffp-contract=off)."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fo():
    return oracle_lib.load_front()


def _ctx(pkg, w, h, **kw):
    cfg = pkg.default_config(w, h)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return pkg.Context(cfg)


@pytest.fixture(scope="module")
def seq():
    """752x480 synthetic stream (SURVEY §8(d) cfg 2): 3 frames under small similarity warps."""
    w, h = 752, 480
    canvas = synth.texture_canvas(w, h, seed=42)
    warps = [dict(tx=0, ty=0, rot_deg=0, scale=1.0), dict(tx=4.2, ty=-3.1, rot_deg=0.3, scale=1.002),
             dict(tx=-1.7, ty=5.4, rot_deg=-0.2, scale=0.997)]
    return w, h, [synth.render_frame(canvas, w, h, **wp) for wp in warps], warps


@pytest.mark.parametrize("w,h", [(752, 480), (1280, 720), (97, 65), (320, 241)])
def test_equalize_and_pyramid_bit_exact(pkg, fo, w, h):
    rng = np.random.default_rng(w)
    img = np.clip(rng.normal(110, 40, (h, w)), 5, 250).astype(np.uint8)
    c = _ctx(pkg, w, h)
    c.feed_image(img)
    eq = fo.equalize_hist(img)
    op = fo.pyramid(eq)
    assert c.pyramid_levels(0) == op.levels
    for l in range(op.levels):
        assert np.array_equal(c.pyramid_level(0, l), op.level(l)[0]), f"level {l}"
    # histogram_method NONE: pyramid of the raw image; constant image: copy
    c2 = _ctx(pkg, w, h, histogram_method=0)
    c2.feed_image(img)
    assert np.array_equal(c2.pyramid_level(0, 1), fo.pyramid(img).level(1)[0])
    const = np.full((h, w), 93, np.uint8)
    c.feed_image(const)
    assert np.array_equal(c.pyramid_level(0, 0), const)
    assert np.array_equal(c.pyramid_level(1, 0), eq)  # previous current became last
    c.close()
    c2.close()


def test_feed_rejects_wrong_size(pkg):
    c = _ctx(pkg, 320, 240)
    with pytest.raises(pkg.PlvError):
        c.feed_image(np.zeros((100, 100), np.uint8))
    with pytest.raises(pkg.PlvError):  # matching before two images were fed
        c.lk_track(np.zeros((12, 2), np.float32), np.zeros((12, 2), np.float32))
    c.close()


@pytest.mark.parametrize("win", [15, 17, 21, 9])
def test_lk_bit_exact_vs_oracle(pkg, fo, seq, win):
    """plv_config.win_size (REF: TrackKLT.h:143-144, the winSize of calcOpticalFlowPyrLK at TrackKLT.cpp:857-858): 15 is the
    reference's value; 17 and 21 (the north star's patch) give every lane of the point's workgroup a second window pixel; the
    pyramid's level count follows the window (buildOpticalFlowPyramid stops where a level is not larger than it)."""
    w, h, frames, warps = seq
    c = _ctx(pkg, w, h, win_size=win)
    c.feed_image(frames[0])
    c.feed_image(frames[1])
    p0, p1 = fo.pyramid(fo.equalize_hist(frames[0]), win=win), fo.pyramid(fo.equalize_hist(frames[1]), win=win)
    pts0 = synth.grid_points(w, h, 250, seed=3, border=12)
    # include border / out-of-image / flat cases
    pts0[:6] = [[2.5, 3.5], [w - 2.0, h - 3.0], [w + 30.0, 50.0], [-30.0, -30.0], [0.0, 0.0], [w - 1.0, h - 1.0]]
    a1, ast, ait = fo.lk_track(p0, p1, pts0, pts0, win=win)
    b1, bst, bit = c.lk_track(pts0, pts0)
    assert np.array_equal(ast, bst)
    assert int(bit.sum()) == ait
    assert np.array_equal(a1, b1), f"max |diff| {np.max(np.abs(a1 - b1))}"
    truth = synth.warp_points(pts0.astype(np.float64), w, h, **warps[1])
    ok = bst.astype(bool)
    assert ok.mean() > 0.9 and np.median(np.linalg.norm(b1[ok] - truth[ok], axis=1)) < 0.06
    c.close()


@pytest.mark.parametrize("win", [15, 9])
def test_lk_ahead_variant_is_the_same_flow(pkg, fo, seq, win):
    """lk_ahead_kernel (PLV_KNOB_LK_AHEAD = 1 << 23; round 6's experiment: the template side of every pyramid level before the first
    iteration, search tiles requested a level ahead — measured no faster, kept behind the knob) returns the bits of lk_kernel<1>,
    border / out-of-image / flat points and large initial errors (tile re-staged, requested tile missed) included."""
    w, h, frames, warps = seq
    c = _ctx(pkg, w, h, win_size=win)
    c.feed_image(frames[0])
    c.feed_image(frames[1])
    pts0 = synth.grid_points(w, h, 250, seed=3, border=12)
    pts0[:6] = [[2.5, 3.5], [w - 2.0, h - 3.0], [w + 30.0, 50.0], [-30.0, -30.0], [0.0, 0.0], [w - 1.0, h - 1.0]]
    init = pts0.copy()
    init[10:60] += np.random.default_rng(1).uniform(-25, 25, (50, 2)).astype(np.float32)
    ref = c.lk_track(pts0, init)
    prev = pkg.debug_knobs(1 << 23)
    try:
        got = c.lk_track(pts0, init)
    finally:
        pkg.debug_knobs(prev)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    c.close()


@pytest.mark.parametrize("win", [15, 21])
def test_lk_large_motion_tile_restage(pkg, fo, win):
    """Initial guesses far from the truth force the 32x32 search tile to be re-staged (a 21 x 21 window leaves it 10 pixels of slack)."""
    w, h = 640, 400
    canvas = synth.texture_canvas(w, h, seed=9)
    f0 = synth.render_frame(canvas, w, h)
    f1 = synth.render_frame(canvas, w, h, tx=14.0, ty=-11.0)
    c = _ctx(pkg, w, h, histogram_method=0, win_size=win)
    c.feed_image(f0)
    c.feed_image(f1)
    p0, p1 = fo.pyramid(f0, win=win), fo.pyramid(f1, win=win)
    pts0 = synth.grid_points(w, h, 100, seed=1, border=40)
    guess = pts0 + np.float32([30.0, 25.0])
    a1, ast, _ = fo.lk_track(p0, p1, pts0, guess, win=win)
    b1, bst, _ = c.lk_track(pts0, guess)
    assert np.array_equal(ast, bst) and np.array_equal(a1, b1)
    c.close()


@pytest.mark.parametrize("w,h,win", [(752, 480, 15), (97, 65, 15), (160, 120, 9)])
def test_lk_border_sweep_bit_exact(pkg, fo, w, h, win):
    """The flow next to the image border (round 5's lean iteration tests the window position against the search tile's usable range
    clipped to the image's, and tells "left the image" from "needs another tile" only behind that test): points in a band along all
    four borders, initial guesses pushed towards and across the border, a shift that carries windows out of the image during the
    iterations, tiles that straddle the border or are larger than a pyramid level.  Positions, status and iteration counts are the
    oracle's, bit for bit."""
    canvas = synth.texture_canvas(w, h, seed=21)
    f0 = synth.render_frame(canvas, w, h)
    f1 = synth.render_frame(canvas, w, h, tx=6.5, ty=-5.25, rot_deg=0.4)
    c = _ctx(pkg, w, h, histogram_method=0, win_size=win)
    c.feed_image(f0)
    c.feed_image(f1)
    p0, p1 = fo.pyramid(f0, win=win), fo.pyramid(f1, win=win)
    rng = np.random.default_rng(5)
    pts, guess = [], []
    for k in range(240):
        side, d, s = k % 4, rng.uniform(-3.0, 22.0), rng.uniform(0.0, 1.0)
        x, y = [(d, s * h), (w - 1 - d, s * h), (s * w, d), (s * w, h - 1 - d)][side]
        pts.append((x, y))
        push = rng.uniform(-26.0, 26.0, size=2)
        guess.append((x + push[0], y + push[1]))
    pts, guess = np.float32(pts), np.float32(guess)
    a1, ast, ait = fo.lk_track(p0, p1, pts, guess, win=win)
    b1, bst, bit = c.lk_track(pts, guess)
    assert np.array_equal(ast, bst), np.flatnonzero(ast != bst)[:10]
    assert int(bit.sum()) == ait
    assert np.array_equal(a1, b1), f"max |diff| {np.max(np.abs(a1 - b1))}"
    assert 0 < int(bst.sum()) < len(pts)      # (both outcomes occur: tracked and left the image)
    c.close()


def test_undistort_bit_exact(pkg, fo):
    c = _ctx(pkg, 752, 480)
    rng = np.random.default_rng(0)
    uv = np.column_stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)]).astype(np.float32)
    K = np.array(list(c.cfg.intrinsics))
    assert np.array_equal(c.undistort(uv), fo.undistort(K, uv))
    c.close()


def _two_view(n, seed, outliers):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 12, n)])
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    X2 = X @ R.T + np.array([0.3, 0.05, 0.1])
    m1, m2 = X[:, :2] / X[:, 2:], X2[:, :2] / X2[:, 2:]
    m1 = m1 + rng.normal(0, 0.3 / 458, m1.shape)
    m2 = m2 + rng.normal(0, 0.3 / 458, m2.shape)
    bad = rng.choice(n, outliers, replace=False)
    m2[bad] += rng.uniform(-0.2, 0.2, (outliers, 2))
    return m1.astype(np.float32), m2.astype(np.float32)


@pytest.mark.parametrize("n,outl,seed", [(250, 50, 2), (500, 200, 3), (40, 5, 4), (7, 0, 5), (12, 0, 6)])
def test_ransac_matches_oracle(pkg, fo, n, outl, seed):
    c = _ctx(pkg, 752, 480)
    m1, m2 = _two_view(n, seed, outl)
    thr = 2.0 / 458.654
    a, ag, ai = fo.ransac(m1, m2, thr, 0.999, 1000, seed=0)
    b, bg, bi = c.ransac(m1, m2, thr, seed=0)
    # same hypothesis order, same Gauss-Jordan pivot order, cubic roots from + - * / sqrt only on both sides: identical masks
    assert np.array_equal(a, b), ((a != b).sum(), ag, bg)
    assert ag == bg and ai == bi
    c.close()


def test_ransac_too_few_points(pkg):
    c = _ctx(pkg, 752, 480)
    m1, m2 = _two_view(5, 1, 0)
    mask, good, it = c.ransac(m1, m2, 0.01)
    assert not mask.any() and good == 0
    c.close()


def test_perform_matching_vs_oracle(pkg, fo, seq):
    w, h, frames, _ = seq
    c = _ctx(pkg, w, h)
    p = [fo.pyramid(fo.equalize_hist(f)) for f in frames]
    c.feed_image(frames[0])
    K = np.array(list(c.cfg.intrinsics))
    pts = synth.grid_points(w, h, 250, seed=5, border=16)
    for i in (1, 2):
        c.feed_image(frames[i])
        rc, a1, am, an0, an1 = fo.perform_matching(p[i - 1], p[i], pts, pts, K)
        b1, bm, bn0, bn1, its = c.perform_matching(pts, pts)
        assert rc == 0
        assert np.array_equal(a1, b1) and np.array_equal(an0, bn0) and np.array_equal(an1, bn1)
        assert np.array_equal(am, bm)
        assert bm.mean() > 0.8 and its > 0
        pts = b1[bm.astype(bool)]
    # fewer than 10 points: all-zero mask, not an error (REF: TrackKLT.cpp:848-852)
    b1, bm, _, _, _ = c.perform_matching(pts[:9], pts[:9])
    assert not bm.any()
    c.close()


def test_clahe_bit_exact(pkg):
    """cv::createCLAHE(10, 8x8) (TrackKLT.cpp:60-64): CLAHE'd level 0 and the pyramid built from it."""
    import oracle_lib
    fo = oracle_lib.load_front()
    cfg = pkg.default_config(752, 480)
    cfg.histogram_method = 2  # PLV_HIST_CLAHE
    c = pkg.Context(cfg)
    rng = np.random.default_rng(8)
    canvas = synth.texture_canvas(752, 480, seed=5, blobs=150)
    imgs = [synth.render_frame(canvas, 752, 480), (100 + 10 * rng.normal(size=(480, 752))).clip(0, 255).astype(np.uint8),
            np.full((480, 752), 31, dtype=np.uint8)]
    for img in imgs:
        c.feed_image(img)
        ref = fo.clahe(img)
        assert np.array_equal(c.pyramid_level(0, 0), ref)
        pyr = fo.pyramid(ref)
        assert np.array_equal(c.pyramid_level(0, 2), pyr.level(2)[0])


def test_front_end_at_config_d(pkg):
    """BASELINE configs[3]: 1280x720, 500 points (6 pyramid levels): LK / undistort bit-exact, lines identical."""
    import oracle_lib
    fo, lo = oracle_lib.load_front(), oracle_lib.load_line()
    W, H = 1280, 720
    cfg = pkg.default_config(W, H)
    cfg.num_features = 500
    c = pkg.Context(cfg)
    canvas = synth.texture_canvas(W, H, seed=21, blobs=400, lines=60)
    f0 = synth.render_frame(canvas, W, H)
    f1 = synth.render_frame(canvas, W, H, tx=4.0, ty=-3.0, rot_deg=0.3, scale=1.002)
    pts0 = synth.grid_points(W, H, 500)
    c.feed_image(f0)
    c.feed_image(f1)
    assert c.pyramid_levels(0) == 6
    pts1, mask, n0, n1, iters = c.perform_matching(pts0, pts0)
    p0, p1 = fo.pyramid(fo.equalize_hist(f0)), fo.pyramid(fo.equalize_hist(f1))
    assert p0.levels == 6
    ref1, st, it = fo.lk_track(p0, p1, pts0, pts0)
    assert it == iters and np.array_equal(pts1[st > 0], ref1[st > 0])
    assert mask.sum() > 400
    K8 = np.array(list(cfg.intrinsics))
    assert np.array_equal(n1, fo.undistort(K8, pts1))
    ref_lines = lo.detect_lines(c.pyramid_level(0, 0))
    got = c.detect_lines(0)
    assert len(got) == len(ref_lines) > 10 and np.array_equal(got, ref_lines)
    # the device walk at this size does not fit LDS (640 x 360 bytes): it runs from global memory
    c.line_walk_mode(True)
    got2 = c.detect_lines(0)
    assert np.array_equal(got2, got)


def test_perform_matching_launch_wait_equals_sync(pkg):
    W, H = 752, 480
    canvas = synth.texture_canvas(W, H, seed=11, blobs=200)
    f0, f1 = synth.render_frame(canvas, W, H), synth.render_frame(canvas, W, H, tx=3.0, ty=1.0, rot_deg=0.3)
    pts0 = synth.grid_points(W, H, 200)
    a, b = pkg.Context(pkg.default_config(W, H)), pkg.Context(pkg.default_config(W, H))
    for c in (a, b):
        c.feed_image(f0)
        c.feed_image(f1)
    ref = a.perform_matching(pts0, pts0)
    b.perform_matching_launch(pts0, pts0)
    got = b.perform_matching_wait()
    for x, y in zip(ref[:4], got[:4]):
        assert np.array_equal(x, y)
    assert ref[4] == got[4]
    with pytest.raises(pkg.PlvError):
        b.perform_matching_wait()          # nothing pending
    b.perform_matching_launch(pts0[:5], pts0[:5])   # fewer than ten points: an all-zero mask, no launch
    assert not b.perform_matching_wait()[1].any()
    a.close()
    b.close()


@pytest.mark.parametrize("w,h", [(333, 251), (752, 480), (1280, 720), (97, 131), (160, 64)])
def test_pyramid_levels_bit_exact_any_size(pkg, w, h):
    """Every level of the optical-flow pyramid against the oracle, including odd sizes (reflect-101 borders inside the two-level
    kernel) and sizes where only some levels exist."""
    fo = oracle_lib.load_front()
    rng = np.random.default_rng(w * h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ctx = pkg.Context(pkg.default_config(w, h))
    ctx.feed_image(img)
    ref = fo.pyramid(fo.equalize_hist(img))
    assert ctx.pyramid_levels(0) == ref.levels
    for l in range(ref.levels):
        got = ctx.pyramid_level(0, l)
        assert got.shape == ref.level(l)[0].shape and np.array_equal(got, ref.level(l)[0]), (w, h, l)
    ctx.close()
