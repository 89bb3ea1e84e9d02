"""The drop-in call for UpdaterCamera::try_update (plv_camera_update_points: feature database -> get_features -> msckf_update ->
cleanup) driven over a simulated sequence, against a Python mirror of the same bookkeeping that computes with the oracle.
Frame by frame: same features chosen, same gate decisions, same correction, same database afterwards."""
import numpy as np
import pytest

import oracle_lib
import synth
import vio_sequence as vs

pytestmark = pytest.mark.gpu

MAX_MSCKF, MAX_OBS, CHI2_MULT = 30, 12, 1.0
TRI = dict(max_cond=1e6, max_dist=60.0, max_baseline=1e3)


class MirrorUpdater:
    """plv_camera_update_points restated on a dict database (REF: CamHelper.cpp:613-738, UpdaterCamera.cpp:197-294)."""

    def __init__(self, pkg):
        self.pkg = pkg
        self.db = {}  # id -> [t list, uv list, uvn list]
        self.used = {}  # point_used: id -> (p_FinG, newest observation time) of triangulated features (REF CamHelper.cpp:677,697)
        self.o, self.jo = oracle_lib.load(), oracle_lib.load_jac(pkg)
        self.q95 = synth.q95_table()

    def append(self, fid, t, uv, uvn):
        e = self.db.setdefault(fid, [[], [], []])
        e[0].append(t), e[1].append(uv), e[2].append(uvn)

    @staticmethod
    def _bounding(ct, t, dt_exp=0.01):
        if len(ct) < 4 or t < ct[0] - dt_exp or t > ct[-1] + dt_exp or t > ct[-1]:
            return False
        return any(ct[i] - dt_exp <= t <= ct[i + 1] + dt_exp for i in range(len(ct) - 1))

    def update(self, st, ct, P, t_prev, state_time, window_full, sigma_pix):
        pkg = self.pkg
        n = P.shape[0]
        unused = {}

        def give(fid, t, uv, uvn):
            e = unused.setdefault(fid, [[], [], []])
            e[0].append(t), e[1].append(uv), e[2].append(uvn)

        take = sorted(fid for fid, e in self.db.items() if any(t < ct[1] for t in e[0]) or not any(t > t_prev for t in e[0]))
        pool = [(fid, self.db.pop(fid)) for fid in take]
        n_pool = len(pool)
        kept = []
        for fid, e in pool:
            k = [[], [], []]
            for t, uv, uvn in zip(*e):
                if t > state_time + 0.01:
                    give(fid, t, uv, uvn)
                elif t < ct[0] - 0.01:
                    continue
                else:
                    k[0].append(t), k[1].append(uv), k[2].append(uvn)
            if len(k[0]) >= 2:
                kept.append((fid, k))
        kept.sort(key=lambda x: -len(x[1][0]))   # stable
        out = dict(n_pool=n_pool, ids=[], accepted=[], dx=np.zeros(n), P=P)
        if kept:
            ptr = np.concatenate([[0], np.cumsum([len(k[0]) for _, k in kept])]).astype(np.int32)
            tr_all = pkg.Tracks(ptr, np.concatenate([k[0] for _, k in kept]), np.concatenate([k[1] for _, k in kept]),
                                np.zeros((len(kept), 3)), obs_uvn=np.concatenate([k[2] for _, k in kept]))
            pf, ok, err = self.jo.triangulate_batch(st, tr_all, **TRI)
            sel, t_first = [], {}
            for q, (fid, k) in enumerate(kept):
                if len(sel) >= MAX_MSCKF:
                    for x in zip(*k):
                        give(fid, *x)
                    continue
                valid = sum(self._bounding(ct, t) for t in k[0])
                if valid >= 2 and ok[q]:
                    self.used[fid] = (pf[q].copy(), k[0][-1])
                if valid < 2 or not ok[q] or not (err[q] < 3.0):
                    for x in zip(*k):
                        give(fid, *x)
                    continue
                if valid > MAX_OBS:   # batch capacity: the last MAX_OBS usable observations
                    t_first[q] = valid - MAX_OBS
                sel.append(q)
            if sel:
                tt, uvs, counts = [], [], []
                for q in sel:
                    fid, k = kept[q]
                    c = seen = 0
                    for t, uv, uvn in zip(*k):
                        if not self._bounding(ct, t):
                            give(fid, t, uv, uvn)
                            continue
                        seen += 1
                        if seen <= t_first.get(q, 0):
                            continue
                        tt.append(t), uvs.append(uv)
                        c += 1
                    counts.append(c)
                sptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
                tr = pkg.Tracks(sptr, np.array(tt), np.array(uvs, dtype=np.float32), pf[sel])
                cols = self.jo.columns(st, tr)
                rows, Hf, Hx, res = self.jo.build_jacobians(st, tr, cols, 2 * MAX_OBS)
                rc, P2, dx, acc, _ = self.o.msckf_update(P, rows, Hf, Hx, res, cols, sigma_pix ** 2, self.q95, chi2_mult=CHI2_MULT)
                assert rc == 0
                out.update(ids=[kept[q][0] for q in sel], accepted=list(acc), dx=dx, P=P2, p_FinG=pf[sel])
                for q, a in zip(sel, acc):
                    if not a:
                        fid, k = kept[q]
                        for t, uv, uvn in zip(*k):
                            if self._bounding(ct, t):
                                give(fid, t, uv, uvn)
        for fid, e in unused.items():
            d = self.db.setdefault(fid, [[], [], []])
            for j in range(3):
                d[j].extend(e[j])
        if window_full:
            for fid in list(self.db):
                e = self.db[fid]
                keep = [i for i, t in enumerate(e[0]) if not t < ct[0]]
                self.db[fid] = [[e[j][i] for i in keep] for j in range(3)]
                if not keep:
                    del self.db[fid]
        return out


def test_camera_update_points_over_a_sequence(pkg):
    world = vs.make_world(3.0, seed=5)
    sw, swb, sa, sab = world["noise"]
    nz = pkg.imu_noise(sw, swb, sa, sab, tuple(vs.G))
    po = oracle_lib.load_prop(pkg)
    ctx = pkg.Context(pkg.default_config(752, 480))
    mir = MirrorUpdater(pkg)
    R0, p0 = vs.trajectory(0.0)
    v0 = (vs.trajectory(1e-5)[1] - vs.trajectory(-1e-5)[1]) / 2e-5
    imu = pkg.PlvImuState.make(vs.rot_2_quat(R0), p0, v0)
    K8 = world["K8"].copy()
    P = np.zeros((23, 23))
    P[np.arange(23), np.arange(23)] = [1e-6] * 6 + [1e-4] * 6 + [1e-2] * 3 + [1.0] * 4 + [1e-4] * 4
    ctx.cov_upload(P)
    clones, t_state, t_prev, n_upd, n_acc = [], 0.0, -1.0, 0, 0
    MAXC = 9
    for tk, lms, uvs in world["frames"]:
        if tk > t_state:
            ok, st_, sw_, sa_ = pkg.select_imu_readings(world["t_imu"], world["wm"], world["am"], t_state, tk)
            imu_o = imu.copy()
            ctx.propagate(imu, nz, st_, sw_, sa_, P.shape[0])
            P = po.propagate(imu_o, nz, st_, sw_, sa_, P=P)[3]
            assert np.abs(imu.vec() - imu_o.vec()).max() < 1e-11
            t_state = tk
        ctx.cov_clone(P.shape[0], 0, 6)
        P = po.cov_clone(P, 0, 6)
        R, p = vs.quat_2_rot(np.array(imu.q)), np.array(imu.p)
        clones.append(dict(t=tk, R=R.copy(), p=p.copy(), Rf=R.copy(), pf=p.copy()))
        for lm, uv in zip(lms[::3], uvs[::3]):        # a third of the landmarks keeps the test short
            uvn = np.array(vs._undistort(K8, uv), dtype=np.float32)
            ctx.db_append_measurements(int(lm) + 1, [tk], [uv], [uvn])
            mir.append(int(lm) + 1, tk, uv, uvn)
        n = P.shape[0]
        if len(clones) >= 4:
            ct = [c["t"] for c in clones]
            st = pkg.StateView(ct, [c["R"] for c in clones], [c["p"] for c in clones], 23 + 6 * np.arange(len(clones)), world["R_ItoC"],
                               world["p_IinC"], K8, clone_R_fej=[c["Rf"] for c in clones], clone_p_fej=[c["pf"] for c in clones],
                               intrinsic_state_id=15, sigma_pix=vs.SIGMA_PIX)
            full = len(clones) > MAXC
            out = ctx.camera_update_points(st, n, MAX_MSCKF, MAX_OBS, t_prev_frame=t_prev, state_time=tk, window_full=full, **TRI)
            ref = mir.update(st, ct, P, t_prev, tk, full, vs.SIGMA_PIX)
            assert out["status"] == 0 and out["n_pool"] == ref["n_pool"], tk
            assert list(out["ids"]) == ref["ids"] and list(out["accepted"]) == ref["accepted"], tk
            assert np.abs(out["dx"] - ref["dx"]).max() <= 1e-7 * max(1.0, np.abs(ref["dx"]).max()), tk
            P = ref["P"]
            assert np.abs(ctx.cov_download(n) - P).max() <= 1e-8 * np.abs(P).max()
            ctx.cov_upload(P)   # keep the two filters on the same covariance bits: the comparison is per frame
            assert ctx.db_size() == len(mir.db), tk
            left = {int(i) for i in ctx.db_select(1, 1e18)}
            assert left == set(mir.db)
            if ref["ids"]:
                n_upd += 1
                n_acc += int(sum(ref["accepted"]))
                dx = ref["dx"]
                q = vs.quat_left_update(np.array(imu.q), dx[0:3])
                for i in range(4):
                    imu.q[i] = q[i]
                for i in range(3):
                    imu.p[i] += dx[3 + i]
                    imu.v[i] += dx[6 + i]
                    imu.bg[i] += dx[9 + i]
                    imu.ba[i] += dx[12 + i]
                K8 = K8 + dx[15:23]
                for ci, c in enumerate(clones):
                    d = dx[23 + 6 * ci:29 + 6 * ci]
                    c["R"] = vs.quat_2_rot(vs.quat_left_update(vs.rot_2_quat(c["R"]), d[:3]))
                    c["p"] = c["p"] + d[3:]
        if len(clones) > MAXC:
            ctx.cov_marginalize(23, 6)
            P = mir.o.cov_marginalize(P, 23, 6)
            clones.pop(0)
        t_prev = tk
    # spot check of the stored observations of what is left
    some = np.array(sorted(mir.db))[:20].astype(np.uint64)
    ptr, tt, uv, _ = ctx.db_export(some)
    for j, fid in enumerate(some):
        assert list(tt[ptr[j]:ptr[j + 1]]) == mir.db[int(fid)][0]
    assert n_upd >= 15 and n_acc >= 100
    Rt, pt = vs.trajectory(t_state)
    assert np.linalg.norm(np.array(imu.p) - pt) < 0.1
    ctx.close()
