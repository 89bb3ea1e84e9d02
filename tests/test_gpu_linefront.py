"""GPU parity of the line front-end (a9-a14) against the oracle, through the C-ABI."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu

W, H = 752, 480


@pytest.fixture(scope="module")
def lo():
    return oracle_lib.load_line()


@pytest.fixture(scope="module")
def frames():
    canvas = synth.texture_canvas(W, H, seed=11, blobs=200, lines=120)
    return [synth.render_frame(canvas, W, H, tx=3.0 * i, ty=-2.0 * i, rot_deg=0.2 * i) for i in range(3)]


def eq_level0(ctx, img):
    ctx.feed_image(img)
    return ctx.pyramid_level(0, 0)


@pytest.mark.parametrize("walk_on_device", [False, True])
def test_detect_lines_parity(ctx, lo, frames, walk_on_device):
    ctx.line_walk_mode(walk_on_device)
    try:
        for img in frames[:2]:
            eq = eq_level0(ctx, img)
            ref = lo.detect_lines(eq)
            got = ctx.detect_lines(0)
            assert len(ref) > 20
            assert len(got) == len(ref)
            # same segments in the same order; the default (host walk + fit) agrees bit for bit, the device kernels to the float
            # rounding of the fit's atan2 / cos / sin
            assert np.abs(got - ref).max() < 2e-3 if walk_on_device else np.array_equal(got, ref)
    finally:
        ctx.line_walk_mode(False)


def test_detect_lines_edge_cases(ctx, lo):
    flat = np.full((H, W), 90, dtype=np.uint8)
    eq_level0(ctx, flat)
    assert len(ctx.detect_lines(0)) == 0
    # a single bright bar: two long edges; a 30 px stub next to it is dropped by the 40 px filter
    img = np.full((H, W), 30, dtype=np.uint8)
    img[200:212, 100:600] = 220
    img[300:306, 100:130] = 220
    eq = eq_level0(ctx, img)
    ref, got = lo.detect_lines(eq), ctx.detect_lines(0)
    assert len(got) == len(ref) >= 2
    assert np.array_equal(got, ref)
    assert (np.hypot(got[:, 2] - got[:, 0], got[:, 3] - got[:, 1]) > 40).all()


def test_point_line_logic_parity(ctx, lo, frames):
    eq = eq_level0(ctx, frames[0])
    lines = ctx.detect_lines(0)
    rng = np.random.default_rng(2)
    # points: half of them sampled on detected segments (+ noise), half anywhere
    k = rng.integers(0, len(lines), 150)
    t = rng.uniform(0, 1, 150)[:, None]
    on = lines[k, :2] * (1 - t) + lines[k, 2:] * t + rng.normal(0, 2.0, (150, 2))
    pts = np.vstack([on, np.column_stack([rng.uniform(0, W, 100), rng.uniform(0, H, 100)])]).astype(np.float32)
    ids = rng.permutation(1000)[:250].astype(np.uint64)
    a, b = ctx.assign_points_to_lines(lines, pts, ids), lo.assign_points_to_lines(lines, pts, ids)
    for key in ("kept", "rel_ptr", "rel_id", "pos_ptr"):
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(a["rel_dist"], b["rel_dist"]) and np.array_equal(a["pos"], b["pos"])
    assert len(a["kept"]) > 3
    # matching against a perturbed copy of the same lines with mostly the same point ids
    last = lines[a["kept"]] + rng.normal(0, 1.0, (len(a["kept"]), 4)).astype(np.float32)
    keep = rng.uniform(size=len(a["rel_id"])) < 0.8
    rid_last = np.where(keep, a["rel_id"], a["rel_id"] + 5000)
    m1 = ctx.line_match(lines[a["kept"]], a["rel_ptr"], a["rel_id"], last, a["rel_ptr"], rid_last)
    m2 = lo.line_match(lines[a["kept"]], a["rel_ptr"], a["rel_id"], last, a["rel_ptr"], rid_last)
    assert np.array_equal(m1, m2) and (m1 >= 0).sum() > 2
    # several lines of the last frame share points with one new line (two perturbed copies of the kept lines, one of them far away, ids
    # kept / replaced independently): the reference's walk lets a LATER line of the last frame overwrite an earlier match, and accepts a
    # single shared point only with the midpoint test — the indexed matching of round 5 has to arrive at the same line
    nk = len(a["kept"])
    last2 = np.vstack([last, lines[a["kept"]] + rng.normal(0, 8.0, (nk, 4)).astype(np.float32), last + np.float32(40.0)])
    n_rel = np.diff(a["rel_ptr"])
    rptr2 = np.concatenate([[0], np.cumsum(np.tile(n_rel, 3))]).astype(a["rel_ptr"].dtype)
    rid2 = np.concatenate([np.where(rng.uniform(size=len(a["rel_id"])) < p_keep, a["rel_id"], a["rel_id"] + 7000 * (q + 1))
                           for q, p_keep in enumerate((0.6, 0.5, 0.25))]).astype(a["rel_id"].dtype)
    for q in range(3):  # (the relations of a line are a std::map's keys: ascending)
        for l in range(nk):
            lo_i, hi_i = rptr2[q * nk + l], rptr2[q * nk + l + 1]
            rid2[lo_i:hi_i] = np.sort(rid2[lo_i:hi_i])
    m3 = ctx.line_match(lines[a["kept"]], a["rel_ptr"], a["rel_id"], last2, rptr2, rid2)
    m4 = lo.line_match(lines[a["kept"]], a["rel_ptr"], a["rel_id"], last2, rptr2, rid2)
    assert np.array_equal(m3, m4)
    assert len(set((m3[m3 >= 0] // nk).tolist())) >= 2     # (matches in more than one of the copies: the later line wins where both share points)
    vps = ctx.vanishing_points(synth._exp_so3(np.array([0.3, -0.2, 0.1])), synth.EUROC_K8)
    assert np.array_equal(vps, lo.vanishing_points(synth._exp_so3(np.array([0.3, -0.2, 0.1])), synth.EUROC_K8))
    cls = [ctx.line_classification(l, vps) for l in lines]
    assert cls == [lo.line_classification(l, vps) for l in lines]


def test_line_tracker_stream(pkg, lo, frames):
    """TrackLSD::feed_monocular over three frames against a composition of oracle pieces."""
    _line_tracker_stream(pkg, lo, frames, W, H)


def test_line_tracker_stream_1280x720(pkg, lo):
    """BASELINE configs[3]: the same stream test on 1280 x 720 frames with 500 points."""
    w, h = 1280, 720
    canvas = synth.texture_canvas(w, h, seed=13, blobs=400, lines=260)
    fr = [synth.render_frame(canvas, w, h, tx=3.0 * i, ty=-2.0 * i, rot_deg=0.15 * i) for i in range(3)]
    _line_tracker_stream(pkg, lo, fr, w, h, num_features=500)


def _line_tracker_stream(pkg, lo, frames, W, H, num_features=None):
    cfg = pkg.default_config(W, H)
    if num_features:
        cfg.num_features = num_features
    ctx = pkg.Context(cfg)
    K8 = np.array(list(cfg.intrinsics))
    vps = lo.vanishing_points(np.eye(3), K8)
    fo = oracle_lib.load_front()
    last = None  # (lines, ids, rel_ptr, rel_id)
    currid = 1
    db = {}
    for i, img in enumerate(frames):
        t = 10.0 + 0.05 * i
        ctx.tracker_feed(t, img)
        ctx.line_tracker_feed(t, vps)
        pts, pids = ctx.tracker_last()
        # oracle composition on the same equalised image and the same tracked points
        eq = ctx.pyramid_level(0, 0)
        lines = lo.detect_lines(eq)
        ids = np.arange(currid + 1, currid + 1 + len(lines), dtype=np.uint64)
        currid += len(lines)
        a = lo.assign_points_to_lines(lines, pts, pids)
        fl, fid = lines[a["kept"]], ids[a["kept"]].copy()
        if last is not None and len(last[0]) > 0:
            m = lo.line_match(fl, a["rel_ptr"], a["rel_id"], last[0], last[2], last[3])
            for q in range(len(fl)):
                if m[q] >= 0:
                    fid[q] = last[1][m[q]]
            for q in range(len(fl)):
                db.setdefault(int(fid[q]), []).append((t, fl[q]))
        last = (fl, fid, a["rel_ptr"], a["rel_id"])
        gl, gid = ctx.line_tracker_last()
        assert np.array_equal(gid, fid)
        assert np.array_equal(gl, fl)
    assert ctx.line_db_size() == len(db) > 0
    ids = ctx.line_db_ids()
    ex = ctx.line_db_export(ids)
    assert sorted(db.keys()) == [int(v) for v in ids]
    for j, lid in enumerate(ids):
        obs = db[int(lid)]
        assert ex["obs_ptr"][j + 1] - ex["obs_ptr"][j] == len(obs)
        o0 = ex["obs_ptr"][j]
        for q, (t, l) in enumerate(obs):
            assert ex["obs_time"][o0 + q] == t
            assert np.array_equal(ex["seg_uv"][o0 + q], l)
            un = fo.undistort(K8, ex["seg_uv"][o0 + q].reshape(2, 2)).ravel()
            assert np.abs(ex["seg_uvn"][o0 + q] - un).max() < 1e-6
    tracked_twice = [k for k, v in db.items() if len(v) >= 2]
    assert len(tracked_twice) >= 1
    del ctx


def test_line_tracker_with_the_callers_points(pkg, lo, frames):
    """plv_line_tracker_feed_points (an adapter that keeps its own point tracker): the same stream, the points handed in instead of
    read from the ctx's tracker -> the same lines, ids and track store as plv_line_tracker_feed."""
    vps = lo.vanishing_points(np.eye(3), synth.EUROC_K8)
    a, b = pkg.Context(pkg.default_config(W, H)), pkg.Context(pkg.default_config(W, H))
    for i, img in enumerate(frames):
        t = 10.0 + 0.05 * i
        a.tracker_feed(t, img)
        a.line_tracker_feed(t, vps)
        pts, pids = a.tracker_last()
        b.feed_image(img)                       # equalise + pyramid only: no point tracker state in this ctx
        b.line_tracker_feed_points(t, vps, pts, pids)
        la, ia = a.line_tracker_last()
        lb, ib = b.line_tracker_last()
        assert np.array_equal(ia, ib) and np.array_equal(la, lb)
    assert a.line_db_size() == b.line_db_size() > 0
    ids = a.line_db_ids()
    ea, eb = a.line_db_export(ids), b.line_db_export(ids)
    for k in ("obs_ptr", "obs_time", "seg_uv", "seg_uvn", "D", "pts_ptr", "pt_ids"):
        assert np.array_equal(ea[k], eb[k]), k
    # no points: every line is dropped by the assignment (TrackLSD.cpp:784-789)
    b.line_tracker_feed_points(11.0, vps, np.zeros((0, 2), dtype=np.float32), np.zeros(0, dtype=np.uint64))
    assert len(b.line_tracker_last()[1]) == 0
    a.close(), b.close()


def test_detection_ahead_of_time(pkg, lo, frames):
    """plv_line_detect_launch / _finish: the detector split around other work on the stream gives the same segments, the line
    tracker takes a finished detection of the same frame and ignores one of an older frame."""
    vps = lo.vanishing_points(np.eye(3), synth.EUROC_K8)
    a, b = pkg.Context(pkg.default_config(W, H)), pkg.Context(pkg.default_config(W, H))
    pts = synth.grid_points(W, H, 200, seed=2)
    ids = np.arange(1, 201, dtype=np.uint64)
    for c in (a, b):
        c.feed_image(frames[0])
    for i, img in enumerate(frames[1:]):
        a.feed_image(img)
        b.feed_image(img)
        ref = a.detect_lines(0)
        b.line_detect_launch(0)
        b.perform_matching_launch(pts, pts)        # device work queued behind the edge maps
        b.line_detect_finish(0)
        out = b.perform_matching_wait()
        assert np.array_equal(b.detect_lines(0), ref)
        b.line_detect_launch(0)                    # launch without finish: the next detection waits for it
        assert np.array_equal(b.detect_lines(0), ref)
        b.line_detect_launch(1)                    # asked for the other image: a detection of image 0 starts from scratch
        assert np.array_equal(b.detect_lines(0), ref)
        a.line_tracker_feed_points(float(i), vps, out[0], ids)
        b.line_detect_launch(0)
        b.line_detect_finish(0)
        b.line_tracker_feed_points(float(i), vps, out[0], ids)
        assert np.array_equal(a.line_tracker_last()[0], b.line_tracker_last()[0])
    # a finished detection that belongs to an older frame is not used
    b.line_detect_finish(0)
    b.feed_image(frames[0])
    a.feed_image(frames[0])
    a.line_tracker_feed_points(9.0, vps, pts, ids)
    b.line_tracker_feed_points(9.0, vps, pts, ids)
    assert np.array_equal(a.line_tracker_last()[0], b.line_tracker_last()[0]) and np.array_equal(a.line_tracker_last()[1], b.line_tracker_last()[1])
    a.close(), b.close()


@pytest.mark.parametrize("scene,mount", [("boulevard", (16.0, 90.0)), ("avenue", (12.0, 0.0))])
def test_component_split_detection_matches_the_oracle(pkg, lo, scene, mount):
    """Round 5: the detection the frame uses (plv_line_detect_launch: edge map -> 8-connected components labelled on the device,
    ccl_merge / ccl_flatten kernels -> host stage split by components over the library's threads) on frames of the bench scenes — the
    'boulevard' facade with its long zigzag components and the 'avenue' corridor with one component of thousands of pixels — against
    the oracle's sequential FastLineDetector: the same segments in the same order, bit for bit; and with 0, 1 and 7 helper threads."""
    import synth_dataset as sd
    sd.set_camera(W, H)
    sd.set_mount(*mount)
    try:
        sim = sd.simulate(seconds=1.0, cam_hz=15, style=scene)
        imgs = sd.render_frames(sim["cam_times"][3:6], scene, 1)
    finally:
        sd.set_mount()
    ctx = pkg.Context(pkg.default_config(W, H))
    spin, fit = pkg.line_worker_config()
    try:
        for threads in (7, 1, 0):
            pkg.line_worker_config(-1, threads)
            for img in imgs:
                ctx.feed_image(img)
                ref = lo.detect_lines(ctx.pyramid_level(0, 0))
                ctx.line_detect_launch(0)
                got = ctx.detect_lines(0)       # (joins the launched detection of this frame: the labelled, component-split one)
                assert len(ref) > 100 and got.shape == ref.shape and np.array_equal(got, ref), (scene, threads, len(ref), len(got))
        # (round 6b) helper threads that fall asleep at random, at a job's pick-up or a part's start: the worker closes jobs without
        # them, runs their parts a second time, takes their slots of the feed's jobs — the same segments, the same kept lines
        pkg.line_worker_config(-1, 7)
        pkg.debug_knobs(1 << 28)
        for rep in range(4):
            for img in imgs:
                ctx.feed_image(img)
                ref = lo.detect_lines(ctx.pyramid_level(0, 0))
                ctx.line_detect_launch(0)
                got = ctx.detect_lines(0)
                assert got.shape == ref.shape and np.array_equal(got, ref), (scene, "naps", rep)
    finally:
        pkg.debug_knobs(0)
        pkg.line_worker_config(spin, fit)
        ctx.close()


def _strips_image(w, h, seed):
    """slanted two-tone strips and boxes over a ramp: a few hundred edge components of all sizes, long and short chains"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 60.0 + 40.0 * xx / w + 30.0 * yy / h
    for _ in range(40):
        a = rng.uniform(0, np.pi)
        c, s_ = np.cos(a), np.sin(a)
        u = c * (xx - rng.uniform(0, w)) + s_ * (yy - rng.uniform(0, h))
        v = -s_ * (xx - rng.uniform(0, w)) + c * (yy - rng.uniform(0, h))
        m = (np.abs(u) < rng.uniform(3, 0.15 * w)) & (np.abs(v) < rng.uniform(10, 0.4 * w))
        img[m] = rng.uniform(20, 235)
    img += rng.normal(0, 1.5, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(320, 240), (700, 500), (1280, 720), (354, 198)])
def test_component_split_detection_at_other_sizes(pkg, lo, w, h):
    """The labelled, component-split detection (round 6b: parts staged and seeded from the per-part pixel lists ccl_flatten_kernel
    leaves per run of 256 pixels) at sizes where a run of 256 pixels spans several rows of the half-resolution map (320 x 240, 354 x
    198: fewer than 256 columns), where the last run is partial (700 x 500, 354 x 198) and at configs[3]'s size — against the oracle's
    sequential FastLineDetector, bit for bit; with the pixel lists off (knob 1 << 26: parts from the label image) the same."""
    ctx = pkg.Context(pkg.default_config(w, h))
    try:
        for seed, knobs in ((1, 0), (2, 0), (3, 1 << 26)):
            pkg.debug_knobs(knobs)
            img = _strips_image(w, h, seed)
            ctx.feed_image(img)
            ref = lo.detect_lines(ctx.pyramid_level(0, 0))
            ctx.line_detect_launch(0)
            got = ctx.detect_lines(0)
            assert len(ref) > 10 and got.shape == ref.shape and np.array_equal(got, ref), (w, h, seed, knobs, len(ref), len(got))
    finally:
        pkg.debug_knobs(0)
        ctx.close()
