"""CPU tests: the detection oracle (FAST-9/16 score + NMS, cornerSubPix, perform_detection bookkeeping)
against brute-force numpy definitions and known synthetic corners."""
import numpy as np
import pytest

import oracle_lib
import synth

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


@pytest.fixture(scope="module")
def do():
    return oracle_lib.load_detect()


def _is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    for sign in (1, -1):
        flags = [(sign * (p - v)) > t for p in ring]
        ext = flags + flags[:8]
        run = 0
        for f in ext:
            run = run + 1 if f else 0
            if run >= 9:
                return True
    return False


def test_fast_score_is_the_largest_surviving_threshold(do):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 40), dtype=np.uint8)
    img[10:30, 10:30] = (img[10:30, 10:30] // 4) + 180  # a bright block: strong corners
    xy, resp = do.fast_roi(img, 0, 0, 40, 40, 20)
    assert len(xy) > 0
    for (x, y), r in zip(xy.astype(int), resp.astype(int)):
        assert _is_corner(img, x, y, r) and not _is_corner(img, x, y, r + 1)
        assert r >= 20
    # every reported point is a corner at the threshold and a strict 3x3 maximum of the score; the
    # 3-px ROI border is skipped
    assert xy.min() >= 3 and xy[:, 0].max() < 37 and xy[:, 1].max() < 37


def test_fast_on_roi_skips_cell_borders(do):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    full, _ = do.fast_roi(img, 0, 0, 80, 60, 15)
    roi, _ = do.fast_roi(img, 20, 10, 40, 30, 15)
    assert len(roi) > 0
    assert roi[:, 0].min() >= 3 and roi[:, 0].max() < 37 and roi[:, 1].min() >= 3 and roi[:, 1].max() < 27
    full_set = {(int(x), int(y)) for x, y in full}
    inner = [(int(x) + 20, int(y) + 10) for x, y in roi if 4 <= x < 36 and 4 <= y < 26]
    assert all(p in full_set for p in inner)  # away from the ROI edge NMS sees the same neighbourhood


def test_corner_subpix_finds_a_synthetic_corner(do):
    yy, xx = np.mgrid[0:64, 0:64].astype(np.float64)
    cx, cy = 30.37, 33.81
    from scipy import ndimage as ndi
    img = ((xx > cx) ^ (yy > cy)).astype(np.float64)  # checkerboard corner
    big = np.kron(img, np.ones((1, 1)))
    img8 = np.clip(ndi.gaussian_filter(((np.mgrid[0:640, 0:640][1] / 10.0 > cx) ^ (np.mgrid[0:640, 0:640][0] / 10.0 > cy))
                                       .astype(np.float64), 8.0)[5::10, 5::10] * 200 + 20, 0, 255).astype(np.uint8)
    out = do.corner_subpix(img8, np.array([[31.0, 33.0]], np.float32))
    assert np.hypot(out[0, 0] - (cx - 0.5), out[0, 1] - (cy - 0.5)) < 0.2
    # flat image: singular system -> the point is left alone
    flat = np.full((64, 64), 90, np.uint8)
    assert np.array_equal(do.corner_subpix(flat, np.array([[20.0, 20.0]], np.float32)), np.array([[20.0, 20.0]], np.float32))


def test_perform_detection_bookkeeping(do):
    w, h = 752, 480
    canvas = synth.texture_canvas(w, h, seed=42)
    img = oracle_lib.load_front().equalize_hist(synth.render_frame(canvas, w, h))
    # initial detection
    pts, ids, cid = do.perform_detection(img, None, np.zeros((0, 2), np.float32), np.zeros(0, np.uint64), 0, 250, 5, 5, 10, 20)
    assert 150 <= len(pts) <= 25 * 11
    assert np.array_equal(ids, np.arange(1, len(pts) + 1, dtype=np.uint64)) and cid == len(pts)
    cells = (pts[:, 0] / 10).astype(int) + 1000 * (pts[:, 1] / 10).astype(int)
    assert len(set(cells)) == len(cells)  # at most one point per min_px_dist cell
    # second call: enough features -> only the clean-up (edge points dropped), no top-up
    pts2 = np.vstack([pts, [[3.0, 3.0]], [pts[0] + 1.0]]).astype(np.float32)
    ids2 = np.concatenate([ids, [9001, 9002]]).astype(np.uint64)
    p3, i3, cid3 = do.perform_detection(img, None, pts2, ids2, cid, len(pts), 5, 5, 10, 20)
    inside = (pts[:, 0].astype(int) >= 10) & (pts[:, 0].astype(int) < w - 10) & (pts[:, 1].astype(int) >= 10) & (pts[:, 1].astype(int) < h - 10)
    assert 9001 not in i3 and 9002 not in i3 and cid3 == cid and len(p3) == inside.sum()  # REF :415-420 edge = 10
    # masked half: nothing new on the masked side
    mask = np.zeros((h, w), np.uint8)
    mask[:, : w // 2] = 255
    p4, _, _ = do.perform_detection(img, mask, np.zeros((0, 2), np.float32), np.zeros(0, np.uint64), 0, 250, 5, 5, 10, 20)
    assert len(p4) > 20 and (p4[:, 0] >= w // 2 - 6).all()
