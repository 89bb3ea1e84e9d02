"""UpdaterCamera::try_update (point half) as one call against a composition of oracle pieces."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def test_camera_update_points(pkg, oracle):
    jo, fo = oracle_lib.load_jac(pkg), oracle_lib.load_front()
    sc = synth.vio_scene(F=90, M=15, noise_px=0.4, seed=9)
    t, K8 = sc["t"], sc["K8"]
    n = sc["n_state"]
    st, _ = synth.scene_views(pkg, sc)
    ctx = pkg.Context(pkg.default_config(752, 480))
    rng = np.random.default_rng(0)
    tracks = {}
    for f in range(90):
        a, b = sc["obs_ptr"][f], sc["obs_ptr"][f + 1]
        uv = sc["obs_uv"][a:b].astype(np.float32)
        tracks[f + 1] = (sc["obs_time"][a:b].copy(), uv, fo.undistort(K8, uv))
    # a feature that is still being tracked and has nothing old: must stay in the database untouched
    a, b = sc["obs_ptr"][3], sc["obs_ptr"][4]
    recent = slice(b - 3, b)
    tracks[500] = (sc["obs_time"][recent].copy(), sc["obs_uv"][recent].astype(np.float32),
                   fo.undistort(K8, sc["obs_uv"][recent].astype(np.float32)))
    # a lost feature with one observation left inside the window: taken and dropped
    tracks[501] = (np.array([t[2]]), np.array([[100.0, 100.0]], dtype=np.float32), np.zeros((1, 2), dtype=np.float32))
    # an outlier track (consistent geometry but a 12 px bias): triangulates, fails the 3 px consistency check
    bad = 7
    tb, ub, nb = tracks[bad]
    ub = ub + rng.normal(0, 9.0, ub.shape).astype(np.float32)
    tracks[bad] = (tb, ub, fo.undistort(K8, ub))
    for fid, (tt, uv, uvn) in tracks.items():
        ctx.db_append_measurements(fid, tt, uv, uvn)
    P = synth.spd_cov(n, seed=4) * 1e-4
    ctx.cov_upload(P)
    MAX, MOBS = 40, 15
    # the default condition-number gate (1e4) rejects most landmarks of this short-baseline scene: open it
    TRI = dict(max_cond=1e7, max_dist=100.0, max_baseline=1e3)
    out = ctx.camera_update_points(st, n, MAX, MOBS, t_prev_frame=t[-2], state_time=t[-1], window_full=True, **TRI)

    # ---- the same decisions with the oracle
    pool = [fid for fid, (tt, _, _) in sorted(tracks.items()) if (tt < t[1]).any() or not (tt > t[-2]).any()]
    assert 500 not in pool and 501 in pool
    pool = [fid for fid in pool if len(tracks[fid][0]) >= 2]
    pool.sort(key=lambda fid: -len(tracks[fid][0]))  # stable: ties keep ascending id
    ptr = np.concatenate([[0], np.cumsum([len(tracks[f][0]) for f in pool])]).astype(np.int32)
    tr_all = pkg.Tracks(ptr, np.concatenate([tracks[f][0] for f in pool]), np.concatenate([tracks[f][1] for f in pool]),
                        np.zeros((len(pool), 3)), obs_uvn=np.concatenate([tracks[f][2] for f in pool]))
    p_o, ok_o, err_o = jo.triangulate_batch(st, tr_all, **TRI)
    sel = []
    for q, fid in enumerate(pool):
        if len(sel) >= MAX:
            break
        if ok_o[q] and err_o[q] < 3.0:
            sel.append(q)
    assert pool.index(bad) not in sel and len(sel) == MAX  # enough candidates to hit the cap
    ids_o = np.array([pool[q] for q in sel], dtype=np.uint64)
    sptr = np.concatenate([[0], np.cumsum([len(tracks[pool[q]][0]) for q in sel])]).astype(np.int32)
    tr_sel = pkg.Tracks(sptr, np.concatenate([tracks[pool[q]][0] for q in sel]), np.concatenate([tracks[pool[q]][1] for q in sel]),
                        p_o[sel])
    cols = jo.columns(st, tr_sel)
    rows, Hf, Hx, res = jo.build_jacobians(st, tr_sel, cols, 2 * MOBS)
    rc_o, P_o, dx_o, acc_o, nrows_o = oracle.msckf_update(P, rows, Hf, Hx, res, cols, st.c.sigma_pix ** 2, synth.q95_table())

    assert out["status"] == rc_o == 0
    assert out["n_pool"] == len([f for f, (tt, _, _) in tracks.items() if (tt < t[1]).any() or not (tt > t[-2]).any()])
    assert np.array_equal(out["ids"], ids_o)
    assert np.array_equal(out["accepted"], acc_o) and out["n_rows"] == nrows_o and acc_o.sum() > 20
    assert np.abs(out["p_FinG"] - p_o[sel]).max() < 1e-6
    assert np.abs(out["dx"] - dx_o).max() <= 1e-7 * max(1.0, np.abs(dx_o).max())
    assert np.abs(ctx.cov_download(n) - P_o).max() <= 1e-8 * np.abs(P).max()
    # ---- database afterwards: consumed features are gone, everything else is back (minus observations older
    # than the oldest clone, none here), the tracked feature never left
    used = {int(i) for i, a in zip(ids_o, acc_o) if a}
    expect = {fid for fid in tracks if fid not in used and fid != 501}
    got = set()
    ids_left = ctx.db_select(1, 1e18)  # everything with an observation older than +inf
    got = {int(i) for i in ids_left}
    assert got == expect
    assert ctx.db_size() == len(expect)


def test_camera_update_lines(pkg, oracle):
    jo = oracle_lib.load_jac(pkg)
    sc = synth.vio_scene(F=4, calib_int=True, dt_clone=0.5, seed=2)   # wide baseline: plane pairs pass the 8 deg gate
    ls = synth.line_scene(sc, L=50, noise_px=0.3, depth=(4.0, 14.0), seed=6)
    t = sc["t"]
    n = sc["n_state"]
    st, _ = synth.scene_views(pkg, sc)
    ctx = pkg.Context(pkg.default_config(752, 480))
    rng = np.random.default_rng(3)
    D = rng.integers(0, 4, 50)
    pts = rng.normal(size=(50, 3)) * 4 + np.array([0, 0, 8.0])
    tracks = {}
    for l in range(50):
        a, b = ls["obs_ptr"][l], ls["obs_ptr"][l + 1]
        pid = [1000 + l, 2000 + l]
        tracks[l + 2] = (ls["obs_time"][a:b].copy(), ls["seg_uv"][a:b].copy(), ls["seg_uvn"][a:b].copy(), int(D[l]), pid)
        ctx.line_db_append_measurements(l + 2, *tracks[l + 2][:3], D=int(D[l]), point_ids=pid)
        if l % 3 == 1:  # a triangulated point on every third line (its second id)
            ctx.point_used_insert(2000 + l, pts[l], t[-1])
    P = synth.spd_cov(n, seed=4) * 1e-4
    ctx.cov_upload(P)
    MOBS = 15
    out = ctx.camera_update_lines(st, n, MOBS, t_prev_frame=t[-2], state_time=t[-1], window_full=True)

    pool = [k for k in sorted(tracks) if (tracks[k][0] < t[1]).any() or not (tracks[k][0] > t[-2]).any()]
    assert 20 < len(pool) < 50  # shorter, still tracked lines stay in the database
    order = sorted(pool, key=lambda k: -len(tracks[k][0]))
    ptr = np.concatenate([[0], np.cumsum([len(tracks[k][0]) for k in order])]).astype(np.int32)
    has = np.array([1 if (k - 2) % 3 == 1 else 0 for k in order], dtype=np.uint8)
    lt_all = pkg.LineTracks(ptr, np.concatenate([tracks[k][0] for k in order]), np.concatenate([tracks[k][1] for k in order]),
                            seg_uvn=np.concatenate([tracks[k][2] for k in order]), D=[tracks[k][3] for k in order],
                            anchor_pt=np.array([pts[k - 2] for k in order]), has_pt=has)
    lg_o, ok_o = jo.triangulate_lines(st, lt_all)
    sel = [q for q in range(len(order)) if ok_o[q]]
    assert 20 < len(sel) and ((has > 0) & (np.array([tracks[k][3] for k in order]) > 0) & (ok_o > 0)).sum() > 3
    sptr = np.concatenate([[0], np.cumsum([len(tracks[order[q]][0]) for q in sel])]).astype(np.int32)
    lt = pkg.LineTracks(sptr, np.concatenate([tracks[order[q]][0] for q in sel]), np.concatenate([tracks[order[q]][1] for q in sel]),
                        line_FinG=lg_o[sel])
    cols = jo.line_columns(st, lt)
    rows, Hf, Hx, res = jo.build_line_jacobians(st, lt, cols, 2 * MOBS)
    rc_o, P_o, dx_o, acc_o, nrows_o = oracle.msckf_update(P, rows, Hf, Hx, res, cols, st.c.sigma_pix ** 2, synth.q95_table(),
                                                          res_norm_gate=0.0)
    assert out["status"] == rc_o == 0 and out["n_pool"] == len(pool)
    assert np.array_equal(out["ids"], np.array([order[q] for q in sel], dtype=np.uint64))
    assert np.abs(out["line_FinG"] - lg_o[sel]).max() <= 1e-9 * max(1.0, np.abs(lg_o).max())
    # (the reference averages direction and moment with different normalisations, so most triangulated lines carry a
    # scale error and fail the chi2 gate: few are accepted, identically on both sides)
    assert np.array_equal(out["accepted"], acc_o) and acc_o.sum() >= 1
    assert np.abs(out["dx"] - dx_o).max() <= 1e-6 * max(1.0, np.abs(dx_o).max())
    # Pluecker moments of metre-scale lines times pixel-scale intrinsics: |H| ~ 1e5, S is far worse conditioned than
    # in the point update
    assert np.abs(ctx.cov_download(n) - P_o).max() <= 1e-6 * np.abs(P).max()
    used = {order[q] for q, a in zip(sel, acc_o) if a}
    assert {int(i) for i in ctx.line_db_ids()} == set(tracks) - used


def test_update_calls_on_empty_databases(pkg):
    sc = synth.vio_scene(F=4, calib_int=True)
    st, _ = synth.scene_views(pkg, sc)
    n = sc["n_state"]
    ctx = pkg.Context(pkg.default_config(752, 480))
    P = synth.spd_cov(n, seed=1) * 1e-4
    ctx.cov_upload(P)
    t = sc["t"]
    out = ctx.camera_update_points(st, n, 40, 15, t_prev_frame=t[-2], state_time=t[-1])
    assert out["n_pool"] == 0 and out["n_msckf"] == 0 and not out["dx"].any() and out["status"] == 0
    out = ctx.camera_update_lines(st, n, 15, t_prev_frame=t[-2], state_time=t[-1])
    assert out["n_pool"] == 0 and out["n_lines"] == 0 and not out["dx"].any()
    assert np.array_equal(ctx.cov_download(n), P)
    # one feature with a single observation: taken, dropped, nothing updated (REF CamHelper.cpp:766-771)
    ctx.db_append_measurements(9, [t[0]], [[10.0, 10.0]], [[0.0, 0.0]])
    out = ctx.camera_update_points(st, n, 40, 15, t_prev_frame=t[-2], state_time=t[-1])
    assert out["n_pool"] == 1 and out["n_msckf"] == 0 and ctx.db_size() == 0


def test_camera_update_points_slam_branch(pkg, oracle):
    """max_slam > 0: get_features' three-way split (REF CamHelper.cpp:621-628,685-693) and the landmark flow after it."""
    jo, fo = oracle_lib.load_jac(pkg), oracle_lib.load_front()
    sc = synth.vio_scene(F=60, M=15, noise_px=0.4, seed=11)
    t, K8, n = sc["t"], sc["K8"], sc["n_state"]
    st, _ = synth.scene_views(pkg, sc)
    ctx = pkg.Context(pkg.default_config(752, 480))
    tracks = {}
    for f in range(60):
        a, b = sc["obs_ptr"][f], sc["obs_ptr"][f + 1]
        if f % 4 == 3:
            a = b - 6  # short tracks: never SLAM-init candidates (init_min_meas = 10)
        uv = sc["obs_uv"][a:b].astype(np.float32)
        tracks[f + 1] = (sc["obs_time"][a:b].copy(), uv, fo.undistort(K8, uv))
    # landmark 300 lives in the state and is still tracked, only recent observations (not in the pool);
    # landmark 301 lives in the state but has no track left; landmark 6 is a full track (the reference leaves it in
    # the database, so the pool sees it too)
    a, b = sc["obs_ptr"][5], sc["obs_ptr"][6]
    recent = slice(b - 2, b)
    uvr = sc["obs_uv"][recent].astype(np.float32)
    tracks[300] = (sc["obs_time"][recent].copy(), uvr, fo.undistort(K8, uvr))
    for fid, (tt, uv, uvn) in tracks.items():
        ctx.db_append_measurements(fid, tt, uv, uvn)
    slam_ids = [300, 301, 6]
    assert np.array_equal(ctx.slam_marg_flags(slam_ids), [0, 1, 0])
    assert np.array_equal(ctx.slam_marg_flags(slam_ids, [0, 0, 2]), [0, 1, 1])
    P = synth.spd_cov(n, seed=4) * 1e-4
    ctx.cov_upload(P)
    MAX, MOBS, MAX_SLAM, MIN_INIT = 25, 15, 7, 10
    TRI = dict(max_cond=1e7, max_dist=100.0, max_baseline=1e3)
    out = ctx.camera_update_points(st, n, MAX, MOBS, t_prev_frame=t[-2], state_time=t[-1], window_full=True, max_slam=MAX_SLAM,
                                   slam_ids=slam_ids, init_min_meas=MIN_INIT, **TRI)
    sl, ini = ctx.camera_update_list(0), ctx.camera_update_list(1)

    # ---- SLAM list: the landmarks with a live track, in state order, observations with bounding clones only
    assert np.array_equal(sl["ids"], [300, 6]) and out["n_slam"] == 2
    valid = lambda tt: np.array([t[0] - 0.01 <= x <= t[-1] for x in tt])  # has_bounding_poses at cam_dt = 0
    for q, fid in enumerate(sl["ids"]):
        a, b = sl["obs_ptr"][q], sl["obs_ptr"][q + 1]
        m = valid(tracks[int(fid)][0])
        assert np.array_equal(sl["obs_time"][a:b], tracks[int(fid)][0][m])
        assert np.array_equal(sl["obs_uv"][a:b], tracks[int(fid)][1][m])
    # ---- the oracle's version of the selection loop
    pool = [fid for fid, (tt, _, _) in sorted(tracks.items()) if (tt < t[1]).any() or not (tt > t[-2]).any()]
    assert 300 not in pool and 6 in pool
    pool.sort(key=lambda fid: -len(tracks[fid][0]))
    ptr = np.concatenate([[0], np.cumsum([len(tracks[f][0]) for f in pool])]).astype(np.int32)
    tr_all = pkg.Tracks(ptr, np.concatenate([tracks[f][0] for f in pool]), np.concatenate([tracks[f][1] for f in pool]),
                        np.zeros((len(pool), 3)), obs_uvn=np.concatenate([tracks[f][2] for f in pool]))
    p_o, ok_o, err_o = jo.triangulate_batch(st, tr_all, **TRI)
    sel, init = [], []
    for q, fid in enumerate(pool):
        if len(sel) >= MAX:
            break
        if not (ok_o[q] and err_o[q] < 3.0):
            continue
        if valid(tracks[fid][0]).sum() >= MIN_INIT and len(slam_ids) + len(init) < MAX_SLAM:
            init.append(q)
            continue
        sel.append(q)
    assert len(init) == MAX_SLAM - len(slam_ids) == out["n_init"] and len(sel) == MAX
    assert np.array_equal(ini["ids"], [pool[q] for q in init])
    assert np.abs(ini["p_FinG"] - p_o[init]).max() < 1e-6
    assert np.array_equal(out["ids"], np.array([pool[q] for q in sel], dtype=np.uint64))
    assert not set(ini["ids"]) & set(out["ids"])
    # ---- the MSCKF update ran on the remaining features only
    sptr = np.concatenate([[0], np.cumsum([len(tracks[pool[q]][0]) for q in sel])]).astype(np.int32)
    tr_sel = pkg.Tracks(sptr, np.concatenate([tracks[pool[q]][0] for q in sel]), np.concatenate([tracks[pool[q]][1] for q in sel]),
                        p_o[sel])
    cols = jo.columns(st, tr_sel)
    rows, Hf, Hx, res = jo.build_jacobians(st, tr_sel, cols, 2 * MOBS)
    rc_o, P_o, dx_o, acc_o, nrows_o = oracle.msckf_update(P, rows, Hf, Hx, res, cols, st.c.sigma_pix ** 2, synth.q95_table())
    assert out["status"] == rc_o == 0 and np.array_equal(out["accepted"], acc_o)
    assert np.abs(out["dx"] - dx_o).max() <= 1e-7 * max(1.0, np.abs(dx_o).max())
    assert np.abs(ctx.cov_download(n) - P_o).max() <= 1e-8 * np.abs(P).max()
    # ---- database: SLAM-init candidates and accepted MSCKF features are gone, landmark 300 never left
    gone = {int(i) for i, a in zip(out["ids"], acc_o) if a} | {int(i) for i in ini["ids"]}
    assert {int(i) for i in ctx.db_select(1, 1e18)} == set(tracks) - gone
    # ---- UpdaterCamera::slam_init on the first candidate (state unchanged here: dx is the caller's to apply)
    a, b = ini["obs_ptr"][0], ini["obs_ptr"][1]
    tr1 = pkg.Tracks(np.array([0, b - a], dtype=np.int32), ini["obs_time"][a:b], ini["obs_uv"][a:b], ini["p_FinG"][:1])
    c1 = ctx.jacobian_columns(st, tr1)
    assert np.array_equal(c1, jo.columns(st, tr1))
    r1, Hf1, Hx1, res1 = ctx.build_jacobians(st, tr1, c1, 2 * MOBS)
    m = int(r1[0])
    ok_o, P2_o, dxi_o, dx2_o = oracle.slam_initialize(P_o, Hf1[0, :, :m].T, Hx1[0, :, :m].T, res1[0, :m], c1, synth.q95_table(), chi2_mult=1.0)
    ok, dxi, dx2 = ctx.slam_initialize(n, Hf1[0, :, :m].T, Hx1[0, :, :m].T, res1[0, :m], c1, chi2_mult=1.0)
    assert ok == ok_o
    if ok:
        assert np.abs(ctx.cov_download(n + 3) - P2_o).max() <= 1e-8 * np.abs(P2_o).max()
        assert np.abs(dxi - dxi_o).max() <= 1e-8 * max(1.0, np.abs(dxi_o).max())
    else:  # REF UpdaterCamera.cpp:363-364 a failed candidate returns to the database
        ctx.db_append_measurements(int(ini["ids"][0]), ini["obs_time"][a:b], ini["obs_uv"][a:b], ini["obs_uvn"][a:b])
        assert int(ini["ids"][0]) in {int(i) for i in ctx.db_select(1, 1e18)}


def test_camera_update_points_with_cpi_poses(pkg, oracle):
    """use_imu_res in the one-call updates (plv_update_options::cpi): poses of every observation from the CPI table (validity,
    triangulation, residual), against the same composition of oracle pieces; observations the table cannot serve return to the
    database."""
    jo, fo = oracle_lib.load_jac(pkg), oracle_lib.load_front()
    sc = synth.vio_scene(n_clones=8, F=60, M=8, noise_px=0.4, seed=12, obs_offset=0.0125)
    cp = synth.cpi_scene(sc)
    t, K8, n = sc["t"], sc["K8"], sc["n_state"]
    st, _ = synth.scene_views(pkg, sc)
    # observations sit 12.5 ms after the clones: between two records of the table (create_new_cpi_linear)
    tab = pkg.CpiTable(cp["t"], cp["clone_t"], cp["R"], cp["alpha"], cp["v"], gravity=cp["gravity"])
    ctx = pkg.Context(pkg.default_config(752, 480))
    tracks = {}
    for f in range(60):
        a, b = sc["obs_ptr"][f], sc["obs_ptr"][f + 1]
        uv = sc["obs_uv"][a:b].astype(np.float32)
        tracks[f + 1] = [sc["obs_time"][a:b].copy(), uv, fo.undistort(K8, uv)]
    # an observation beyond the last record of the table: nothing to interpolate towards (create_new_cpi_integrate territory)
    late = float(cp["t"][-1] + 0.004)
    tracks[5][0][-1] = late
    for fid, (tt, uv, uvn) in tracks.items():
        ctx.db_append_measurements(fid, tt, uv, uvn)
    P = synth.spd_cov(n, seed=4) * 1e-4
    ctx.cov_upload(P)
    MAX, MOBS = 30, 10
    TRI = dict(max_cond=1e7, max_dist=100.0, max_baseline=1e3)
    state_time = late + 0.1
    out = ctx.camera_update_points(st, n, MAX, MOBS, t_prev_frame=t[-2], state_time=state_time, window_full=True, cpi=tab, **TRI)

    # ---- oracle composition
    pool = [fid for fid, (tt, _, _) in sorted(tracks.items()) if (tt < t[1]).any() or not (tt > t[-2]).any()]
    pool.sort(key=lambda fid: -len(tracks[fid][0]))
    tq = np.concatenate([tracks[f][0] for f in pool])
    Rq, pq, okq = jo.cpi_poses(st, tab, tq)
    assert (okq == 0).sum() == 1 and not okq[np.flatnonzero(tq == late)[0]]
    ptr = np.concatenate([[0], np.cumsum([len(tracks[f][0]) for f in pool])])
    kept = {}
    for j, f in enumerate(pool):
        m = okq[ptr[j]:ptr[j + 1]].astype(bool)
        kept[f] = (tracks[f][0][m], tracks[f][1][m], tracks[f][2][m], Rq[ptr[j]:ptr[j + 1]][m], pq[ptr[j]:ptr[j + 1]][m])
    kptr = np.concatenate([[0], np.cumsum([len(kept[f][0]) for f in pool])]).astype(np.int32)
    cat = lambda i: np.concatenate([kept[f][i] for f in pool])
    tr_all = pkg.Tracks(kptr, cat(0), cat(1), np.zeros((len(pool), 3)), obs_uvn=cat(2), res_R=cat(3), res_p=cat(4))
    p_o, ok_o, err_o = jo.triangulate_batch(st, tr_all, **TRI)
    sel = [q for q in range(len(pool)) if ok_o[q] and err_o[q] < 3.0][:MAX]
    sptr = np.concatenate([[0], np.cumsum([len(kept[pool[q]][0]) for q in sel])]).astype(np.int32)
    scat = lambda i: np.concatenate([kept[pool[q]][i] for q in sel])
    tr_sel = pkg.Tracks(sptr, scat(0), scat(1), p_o[sel], res_R=scat(3), res_p=scat(4))
    cols = jo.columns(st, tr_sel)
    rows, Hf, Hx, res = jo.build_jacobians(st, tr_sel, cols, 2 * MOBS)
    rc_o, P_o, dx_o, acc_o, nrows_o = oracle.msckf_update(P, rows, Hf, Hx, res, cols, st.c.sigma_pix ** 2, synth.q95_table())
    assert out["status"] == rc_o == 0 and len(sel) >= 20
    assert np.array_equal(out["ids"], np.array([pool[q] for q in sel], dtype=np.uint64))
    assert np.array_equal(out["accepted"], acc_o) and acc_o.sum() >= 15 and out["n_rows"] == nrows_o
    assert np.abs(out["p_FinG"] - p_o[sel]).max() < 1e-6
    assert np.abs(out["dx"] - dx_o).max() <= 1e-7 * max(1.0, np.abs(dx_o).max())
    assert np.abs(ctx.cov_download(n) - P_o).max() <= 1e-8 * np.abs(P).max()
    ptr_l, tt_l, _, _ = ctx.db_export(np.array([5], dtype=np.uint64))
    assert late in list(tt_l[ptr_l[0]:ptr_l[1]])          # the observation without a pose is back in the database
    ctx.close()
