"""CPU checks of the line front-end oracle (a9-a13): Canny / FastLineDetector restatement and the
TrackLSD logic around it (oracle/line_oracle.cpp)."""
import numpy as np
import pytest

import oracle_lib
import synth


@pytest.fixture(scope="module")
def lo():
    return oracle_lib.load_line()


def edge_image(w=376, h=240, segs=((40, 60, 300, 90), (100, 200, 330, 40), (60, 30, 60, 210))):
    """dark background, each segment the boundary of a bright half-plane strip: clean straight edges"""
    img = np.full((h, w), 40, dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for x1, y1, x2, y2 in segs:
        d = np.array([x2 - x1, y2 - y1], dtype=np.float32)
        L = np.hypot(*d)
        d /= L
        t = (xx - x1) * d[0] + (yy - y1) * d[1]
        s = -(xx - x1) * d[1] + (yy - y1) * d[0]
        img[(t >= 0) & (t <= L) & (s >= 0) & (s < 6)] = 200
    return img.astype(np.uint8)


def test_resize_half_is_rounded_block_mean(lo):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    half = lo.resize_half(img)
    ref = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert (half == ref).all()


def test_canny_marks_thin_edges(lo):
    img = edge_image()
    e = lo.canny(img)
    assert set(np.unique(e)) <= {0, 255}
    n = (e > 0).sum()
    assert 1200 < n < 2600          # ~2 x (262 + 281 + 180) edge pixels + strip ends, one pixel thin
    # flat regions carry no edges
    assert e[215:235, 250:370].sum() == 0
    # thinness: no 2x2 block fully set
    b = (e[:-1, :-1] > 0) & (e[1:, :-1] > 0) & (e[:-1, 1:] > 0) & (e[1:, 1:] > 0)
    assert b.sum() == 0


def test_fld_recovers_rendered_segments(lo):
    truth = ((40, 60, 300, 90), (100, 200, 330, 40), (60, 30, 60, 210))
    img = edge_image(segs=truth)
    segs, edges = lo.fld(img, want_edges=True)
    assert edges[:6, :6].sum() == 0
    assert 6 <= len(segs) <= 16      # two long sides per strip (+ possibly split pieces / short ends)
    for x1, y1, x2, y2 in truth:
        d = np.array([x2 - x1, y2 - y1], dtype=float)
        L = np.hypot(*d)
        d /= L
        # some detected segment is parallel to the strip and covers most of its length
        best = 0.0
        for s in segs:
            v = np.array([s[2] - s[0], s[3] - s[1]], dtype=float)
            ln = np.hypot(*v)
            if abs(v @ d) / ln < 0.999:
                continue
            off = abs(-(s[0] - x1) * d[1] + (s[1] - y1) * d[0])
            if off < 8:
                best = max(best, ln)
        assert best > 0.5 * L   # strips cross each other: edges are interrupted at the crossings
    # every segment is at least the length threshold
    assert (np.hypot(segs[:, 2] - segs[:, 0], segs[:, 3] - segs[:, 1]) >= 20).all()


def test_detect_lines_scales_and_filters(lo):
    img = np.kron(edge_image(), np.ones((2, 2), dtype=np.uint8))  # 752 x 480, block-constant: half-res == original
    lines = lo.detect_lines(img)
    segs = lo.fld(edge_image())
    keep = ((segs[:, 2] - segs[:, 0]) * 2) ** 2 + ((segs[:, 3] - segs[:, 1]) * 2) ** 2 > 1600
    assert np.array_equal(lines, segs[keep] * 2)


def test_point_line_distance_cases(lo):
    line = [10.0, 10.0, 30.0, 10.0]
    assert lo.point_line_distance(line, 20, 14) == pytest.approx(4.0)
    assert lo.point_line_distance(line, 4, 18) == pytest.approx(10.0)     # before the start: distance to it
    assert lo.point_line_distance(line, 33, 14) == pytest.approx(5.0)     # beyond the end


def test_assign_points_uses_reference_bbox_quirk(lo):
    # REF reads (x1, y1, x2, y2) as (lx1, lx2, ly1, ly2): the x range tested is [min(x1,y1), max(x1,y1)] and
    # the y range [min(x2,y2), max(x2,y2)]
    lines = np.array([[100, 200, 300, 250], [50, 60, 400, 80]], dtype=np.float32)
    pts = np.array([[150, 260], [150, 212.5], [55, 61]], dtype=np.float32)  # only pt 0 passes line 0's (quirky) box
    ids = np.array([7, 8, 9], dtype=np.uint64)
    r = lo.assign_points_to_lines(lines, pts, ids)
    # point 0: x in [100,200], y in [250,300] but 47 px from the segment -> rejected by distance;
    # point 1 lies ON line 0 but fails the quirky y range; point 2 is on line 1 but x range is [50,60], y in [80,400] fails
    assert len(r["kept"]) == 0
    pts2 = np.array([[150, 255]], dtype=np.float32)
    lines2 = np.array([[100, 200, 300, 250], [140, 160, 252, 258]], dtype=np.float32)
    r2 = lo.assign_points_to_lines(lines2, pts2, np.array([5], dtype=np.uint64))
    # line 1: x range [140,160], y range [252,258]; the point is ~94 px from the segment's start -> rejected
    assert len(r2["kept"]) == 0
    lines3 = np.array([[150, 160, 152, 258]], dtype=np.float32)   # near-vertical, x in [150,160], y in [152,258]
    r3 = lo.assign_points_to_lines(lines3, np.array([[151.5, 200]], dtype=np.float32), np.array([5], dtype=np.uint64))
    assert list(r3["kept"]) == [0] and list(r3["rel_id"]) == [5] and r3["rel_dist"][0] < 1.0


def test_line_match_rules(lo):
    new = np.array([[0, 0, 100, 0], [0, 50, 100, 50]], dtype=np.float32)
    last = np.array([[2, 1, 98, 1], [0, 80, 100, 80]], dtype=np.float32)
    # new 0 shares two points with last 0 -> match; new 1 shares one point with last 1 but is 30 px away -> none
    m = lo.line_match(new, [0, 2, 3], [1, 2, 3], last, [0, 2, 3], [1, 2, 3])
    assert list(m) == [0, -1]
    # one shared point and the last line's midpoint within 6 px of the new segment -> match
    m = lo.line_match(new[:1], [0, 1], [1], last[:1], [0, 1], [1])
    assert list(m) == [0]


def test_vanishing_points_and_classification(lo):
    K8 = synth.EUROC_K8
    R = np.eye(3)
    vps = lo.vanishing_points(R, K8)
    # e_z -> normalised (0,0) -> principal point, y scaled by 1000 (REF quirk)
    assert vps[2, 0] == pytest.approx(K8[2], abs=1e-3) and vps[2, 1] == pytest.approx(1000 * K8[3], rel=1e-6)
    # a segment pointing at vps[0] with a small slope is class 1
    vx = vps[0]
    mid = np.array([200.0, 240.0])
    d = (vx - mid) / np.linalg.norm(vx - mid)
    line = np.concatenate([mid - 30 * d, mid + 30 * d]).astype(np.float32)
    assert lo.line_classification(line, vps) in (1, 2)
    assert lo.line_classification(np.array([10, 10, 12, 300], dtype=np.float32), vps) == 0
