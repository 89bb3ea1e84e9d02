"""CPU tests: the front-end oracle (OpenCV-contract restatement) against independent numpy/scipy
formulas and against known synthetic warps (SURVEY.md §8(c) golden-vector plan v, vi, vii)."""
import numpy as np
import pytest

import oracle_lib
import synth


@pytest.fixture(scope="module")
def fo():
    return oracle_lib.load_front()


@pytest.fixture(scope="module")
def frames():
    w, h = 320, 240
    canvas = synth.texture_canvas(w, h, seed=42)
    f0 = synth.render_frame(canvas, w, h)
    warp = dict(tx=3.3, ty=-2.6, rot_deg=0.4, scale=1.003)
    f1 = synth.render_frame(canvas, w, h, **warp)
    return w, h, f0, f1, warp


def test_equalize_hist_matches_definition(fo):
    rng = np.random.default_rng(0)
    img = np.clip(rng.normal(120, 30, (97, 131)), 10, 240).astype(np.uint8)
    out = fo.equalize_hist(img)
    hist = np.bincount(img.ravel(), minlength=256)
    i0 = np.nonzero(hist)[0][0]
    scale = np.float32(255.0) / np.float32(img.size - hist[i0])
    cdf = np.cumsum(hist) - hist[i0]
    lut = np.clip(np.rint((cdf.astype(np.float32) * scale).astype(np.float32)), 0, 255).astype(np.uint8)
    lut[:i0 + 1] = 0
    assert np.array_equal(out, lut[img])
    const = np.full((8, 8), 77, np.uint8)
    assert np.array_equal(fo.equalize_hist(const), const)


def test_pyramid_levels_and_pyrdown(fo):
    rng = np.random.default_rng(1)
    for (w, h, nlev) in [(752, 480, 5), (1280, 720, 6), (1280, 560, 6), (97, 65, 3)]:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        pyr = fo.pyramid(img)
        assert pyr.levels == nlev, (w, h, pyr.levels)
    img = rng.integers(0, 256, (65, 97), dtype=np.uint8)
    pyr = fo.pyramid(img)
    l1, _ = pyr.level(1)
    # independent separable [1 4 6 4 1] with reflect-101, (sum+128)>>8
    k = np.array([1, 4, 6, 4, 1])
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")
    full = np.zeros((65, 97), dtype=np.int64)
    for dy in range(5):
        for dx in range(5):
            full += k[dy] * k[dx] * pad[dy:dy + 65, dx:dx + 97]
    exp = ((full + 128) >> 8)[::2, ::2]
    assert l1.shape == (33, 49) and np.array_equal(l1, exp.astype(np.uint8))


def test_scharr_matches_definition(fo):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    _, der = fo.pyramid(img, win=15, max_level=0).level(0)
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    dx = 3 * (p[:-2, 2:] - p[:-2, :-2]) + 10 * (p[1:-1, 2:] - p[1:-1, :-2]) + 3 * (p[2:, 2:] - p[2:, :-2])
    dy = 3 * (p[2:, :-2] - p[:-2, :-2]) + 10 * (p[2:, 1:-1] - p[:-2, 1:-1]) + 3 * (p[2:, 2:] - p[:-2, 2:])
    assert np.array_equal(der[..., 0], dx) and np.array_equal(der[..., 1], dy)


def test_lk_recovers_known_warp(fo, frames):
    w, h, f0, f1, warp = frames
    p0, p1 = fo.pyramid(f0), fo.pyramid(f1)
    pts0 = synth.grid_points(w, h, 120, seed=3, border=30)
    pts1, st, iters = fo.lk_track(p0, p1, pts0, pts0)
    truth = synth.warp_points(pts0.astype(np.float64), w, h, **warp)
    ok = st.astype(bool)
    assert ok.mean() > 0.9
    err = np.linalg.norm(pts1[ok] - truth[ok], axis=1)
    assert np.median(err) < 0.05, np.median(err)
    assert np.percentile(err, 90) < 0.25
    assert 0 < iters < 120 * 5 * 30


def test_lk_identity_and_status(fo, frames):
    w, h, f0, _, _ = frames
    p0 = fo.pyramid(f0)
    pts0 = synth.grid_points(w, h, 50, seed=4)
    pts1, st, _ = fo.lk_track(p0, p0, pts0, pts0)
    assert st.all() and np.max(np.abs(pts1 - pts0)) < 0.02
    # a point far outside the image: status 0 at level 0 (OOB test)
    far = np.array([[w + 40.0, h + 40.0], [-40.0, 10.0]], dtype=np.float32)
    _, st2, _ = fo.lk_track(p0, p0, far, far)
    assert not st2.any()
    # flat image: min-eigenvalue test fails
    flat = fo.pyramid(np.full((h, w), 128, np.uint8))
    _, st3, _ = fo.lk_track(flat, flat, pts0[:5], pts0[:5])
    assert not st3.any()


def test_lk_thread_invariance(fo, frames):
    w, h, f0, f1, _ = frames
    p0, p1 = fo.pyramid(f0), fo.pyramid(f1)
    pts0 = synth.grid_points(w, h, 64, seed=5)
    a = fo.lk_track(p0, p1, pts0, pts0, nthreads=1)
    b = fo.lk_track(p0, p1, pts0, pts0, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_undistort_round_trip(fo):
    K = synth.EUROC_K8
    rng = np.random.default_rng(6)
    xy = rng.uniform(-0.5, 0.5, (200, 2))
    r2 = (xy ** 2).sum(1)
    rad = 1 + K[4] * r2 + K[5] * r2 ** 2
    xd = xy[:, 0] * rad + 2 * K[6] * xy[:, 0] * xy[:, 1] + K[7] * (r2 + 2 * xy[:, 0] ** 2)
    yd = xy[:, 1] * rad + K[6] * (r2 + 2 * xy[:, 1] ** 2) + 2 * K[7] * xy[:, 0] * xy[:, 1]
    uv = np.column_stack([K[0] * xd + K[2], K[1] * yd + K[3]]).astype(np.float32)
    back = fo.undistort(K, uv)
    # 5 fixed-point iterations (OpenCV 4.2) on EuRoC-strength distortion
    assert np.max(np.abs(back - xy)) < 2e-3
    near = r2 < 0.05
    assert np.max(np.abs(back[near] - xy[near])) < 2e-5


def _two_view(n, seed, outliers):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 12, n)])
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = np.array([0.3, 0.05, 0.1])
    X2 = X @ R.T + t
    m1 = X[:, :2] / X[:, 2:]
    m2 = X2[:, :2] / X2[:, 2:]
    m1 += rng.normal(0, 0.3 / 458, m1.shape)
    m2 += rng.normal(0, 0.3 / 458, m2.shape)
    bad = rng.choice(n, outliers, replace=False)
    m2[bad] += rng.uniform(-0.2, 0.2, (outliers, 2))
    truth = np.ones(n, bool)
    truth[bad] = False
    return m1.astype(np.float32), m2.astype(np.float32), truth


def test_seven_point_satisfies_constraints(fo):
    m1, m2, _ = _two_view(7, 1, 0)
    Fs = fo.run7point(m1, m2, np.arange(7))
    assert 1 <= len(Fs) <= 3
    for F in Fs:
        x1 = np.column_stack([m1, np.ones(7)]).astype(np.float64)
        x2 = np.column_stack([m2, np.ones(7)]).astype(np.float64)
        resid = np.einsum("ni,ij,nj->n", x2, F, x1)
        assert np.max(np.abs(resid)) < 1e-9 * max(1.0, np.abs(F).max())
        assert abs(np.linalg.det(F)) < 1e-9 * max(1.0, np.abs(F).max() ** 3)


def test_ransac_inlier_iou(fo):
    m1, m2, truth = _two_view(250, 2, 50)
    mask, good, iters = fo.ransac(m1, m2, 2.0 / 458.654, 0.999, 1000, seed=0)
    mask = mask.astype(bool)
    iou = (mask & truth).sum() / (mask | truth).sum()
    assert iou >= 0.95, iou
    assert good == mask.sum() and 1 <= iters < 1000
    mask2, _, _ = fo.ransac(m1, m2, 2.0 / 458.654, 0.999, 1000, seed=0)
    assert np.array_equal(mask2.astype(bool), mask)
    m, g, _ = fo.ransac(m1[:5], m2[:5], 0.01)
    assert g == 0 and not m.any()


def test_perform_matching_small_set_is_all_zero(fo, frames):
    w, h, f0, f1, _ = frames
    p0, p1 = fo.pyramid(f0), fo.pyramid(f1)
    pts = synth.grid_points(w, h, 9, seed=7)
    rc, pts1, mask, _, _ = fo.perform_matching(p0, p1, pts, pts, synth.EUROC_K8)
    assert rc == 1 and not mask.any()


def test_clahe_oracle_properties():
    import oracle_lib
    fo = oracle_lib.load_front()
    rng = np.random.default_rng(5)
    # low-contrast texture: CLAHE stretches it, stays in range, is deterministic
    img = (120 + 6 * rng.normal(size=(480, 752))).clip(0, 255).astype(np.uint8)
    out = fo.clahe(img)
    assert out.shape == img.shape and out.dtype == np.uint8
    assert out.std() > 2.0 * img.std()
    assert np.array_equal(out, fo.clahe(img))
    # a tile-constant image: every tile LUT maps its single grey level to round(255 * clipped cdf) and the
    # blend of equal LUT values is that value; with clip = 10 * area / 256 the cdf at the level is
    # (level + 1) * batch + clip + residual bins below it, the same in every tile -> constant output
    flat = np.full((480, 752), 77, dtype=np.uint8)
    o2 = fo.clahe(flat)
    assert (o2 == o2[0, 0]).all()
    # non-divisible sizes go through the reflect-101 padding branch
    odd = rng.integers(0, 256, (100, 150), dtype=np.uint8)
    assert fo.clahe(odd).shape == (100, 150)
