"""The slice of plviwo_amd.Context that pl-viwo_amd/system.py drives (IMU + one camera, points and lines, wheel), served by the CPU oracle:
the same SystemManager then runs the whole filter on the CPU and its trajectory is the "CPU reference" of the replay tests.
Test infrastructure.  OracleContext = the compiled CPU frame (oracle/frame_oracle.cpp: tracker, databases, try_update in C++);
PyMirrorContext = the same bookkeeping in Python over the oracle's numeric pieces (compositions as in test_gpu_tracker.py and
test_gpu_dropin_sequence.py), kept as the cross-check of the compiled one (tests/test_frame_oracle.py)."""
import numpy as np

import oracle_lib
import synth
from test_gpu_dropin_sequence import MirrorUpdater


class PyMirrorContext:
    def __init__(self, cfg):
        import __graft_entry__ as ge
        self.pkg = ge.load_pkg()
        self.cfg = cfg
        self.o, self.fo, self.do = oracle_lib.load(), oracle_lib.load_front(), oracle_lib.load_detect()
        self.po, self.jo = oracle_lib.load_prop(self.pkg), oracle_lib.load_jac(self.pkg)
        self.q95 = synth.q95_table()
        self.P = None
        self.K8 = np.array(list(cfg.intrinsics))
        self.mir = MirrorUpdater(self.pkg)       # owns the feature database: id -> [t, uv, uvn]
        self.pts, self.ids, self.currid, self.prev = np.zeros((0, 2), np.float32), np.zeros(0, np.uint64), 0, None
        # TrackLSD state + LineFeatureDatabase: id -> dict(t, uv, uvn, points, D)
        self.lo = oracle_lib.load_line()
        self.line_last, self.line_currid, self.ldb = None, 1, {}

    def close(self):
        pass

    # ---- covariance
    def cov_upload(self, P):
        self.P = np.array(P, dtype=np.float64, order="F")

    def cov_download(self, n):
        assert self.P.shape[0] == n
        return self.P.copy()

    def cov_clone(self, n, src_id, size=6):
        self.P = self.po.cov_clone(self.P, src_id, size)

    def cov_marginalize(self, idx, size):
        self.P = self.o.cov_marginalize(self.P, idx, size)

    def propagate(self, imu, noise, t, wm, am, n, acc=None, imu_id=0, want_records=True):
        Phi, Qd, rec, self.P = self.po.propagate(imu, noise, t, wm, am, P=self.P, acc=acc, imu_id=imu_id)
        return Phi, Qd, rec

    def set_camera_intrinsics(self, K8):
        self.K8 = np.array(K8, dtype=np.float64)

    # ---- TrackKLT::feed_monocular (REF: TrackKLT.cpp:96-200) + FeatureDatabase::update_feature
    def _detect(self, eq, pts, ids, mask):
        c = self.cfg
        p, i, self.currid = self.do.perform_detection(eq, mask, pts, ids, self.currid, c.num_features, c.grid_x, c.grid_y, c.min_px_dist,
                                                      c.fast_threshold)
        return p, i

    def tracker_feed(self, t, img, mask=None):
        eq = self.fo.equalize_hist(img)
        pyr = self.fo.pyramid(eq)
        if len(self.ids) == 0:
            self.pts, self.ids = self._detect(eq, self.pts, self.ids, mask)   # REF :110-123: detections only, no database entry
            self.prev = (eq, pyr)
            return
        pts_old, ids_old = self._detect(self.prev[0], self.pts, self.ids, mask)
        rc, pts_new, ok, n0, n1 = self.fo.perform_matching(self.prev[1], pyr, pts_old, pts_old, self.K8, nthreads=getattr(self, "lk_threads", 1))
        h, w = img.shape
        good, gid = [], []
        for i in range(len(pts_old)):
            x, y = pts_new[i]
            if x < 0 or y < 0 or int(x) >= w or int(y) >= h or not ok[i]:
                continue
            if mask is not None and mask[int(y), int(x)] > 127:
                continue
            good.append(pts_new[i])
            gid.append(ids_old[i])
            self.mir.append(int(ids_old[i]), t, pts_new[i].copy(), n1[i].copy())
        self.pts, self.ids = np.array(good, np.float32).reshape(-1, 2), np.array(gid, np.uint64)
        self.prev = (eq, pyr)

    def db_cleanup_measurements(self, t):
        for fid in list(self.mir.db):
            e = self.mir.db[fid]
            keep = [i for i, x in enumerate(e[0]) if not x < t]
            if keep:
                self.mir.db[fid] = [[e[j][i] for i in keep] for j in range(3)]
            else:
                del self.mir.db[fid]

    # ---- UpdaterCamera::try_update, point half
    def camera_update_points(self, st, n, max_msckf, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, min_dist=0.1,
                             max_dist=60.0, max_cond=1e4, max_baseline=40.0, refine=True, max_slam=0, slam_ids=(), init_min_meas=10,
                             cpi=None):
        assert max_slam == 0 and cpi is None
        import test_gpu_dropin_sequence as m
        m.MAX_MSCKF, m.MAX_OBS = max_msckf, max_obs
        m.TRI = dict(min_dist=min_dist, max_dist=max_dist, max_cond=max_cond, max_baseline=max_baseline, refine=refine)
        ct = [float(x) for x in st.t]
        m.CHI2_MULT = chi2_mult
        ref = self.mir.update(st, ct, self.P, t_prev_frame, state_time, window_full, st.c.sigma_pix)
        self.P = np.array(ref["P"], order="F")
        acc = np.array(ref["accepted"], dtype=np.uint8)
        return dict(dx=ref["dx"], n_pool=ref["n_pool"], n_msckf=len(ref["ids"]), n_accepted=int(acc.sum()), status=0, ids=ref["ids"],
                    accepted=acc, n_slam=0, n_init=0)

    # ---- TrackLSD::feed_monocular (REF: TrackLSD.cpp:70-192) on the image and the points of the last tracker_feed
    def vanishing_points(self, R_ItoC, K8):
        return self.lo.vanishing_points(R_ItoC, K8)

    def line_db_size(self):
        return len(self.ldb)

    def line_tracker_feed(self, t, vps):
        lo = self.lo
        lines = lo.detect_lines(self.prev[0])
        ids = np.arange(self.line_currid + 1, self.line_currid + 1 + len(lines), dtype=np.uint64)   # REF :233-236 ++currid
        self.line_currid += len(lines)
        a = lo.assign_points_to_lines(lines, self.pts, self.ids)
        fl, fid = lines[a["kept"]], ids[a["kept"]].copy()
        last = self.line_last
        if last is not None and len(last[0]) > 0:      # REF :100: first frame or everything lost -> no matching, no database entry
            if len(fl):
                m = lo.line_match(fl, a["rel_ptr"], a["rel_id"], last[0], last[2], last[3])
                for q in range(len(fl)):
                    if m[q] >= 0:
                        fid[q] = last[1][m[q]]
                un = self.fo.undistort(self.K8, fl.reshape(-1, 2)).reshape(-1, 4)
            for q in range(len(fl)):
                D = lo.line_classification(fl[q], vps)
                e = self.ldb.get(int(fid[q]))
                if e is None:
                    e = self.ldb[int(fid[q])] = dict(t=[], uv=[], uvn=[], points=[], D=D)   # only a new feature takes D
                e["t"].append(t), e["uv"].append(fl[q].copy()), e["uvn"].append(un[q].copy())
                e["points"].extend(int(x) for x in a["rel_id"][a["rel_ptr"][q]:a["rel_ptr"][q + 1]])
        self.line_last = (fl, fid, a["rel_ptr"], a["rel_id"])

    def camera_get_line_features(self, st, n, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, cpi=None):
        """LineHelper::get_line_features (REF: linefeat/LineHelper.cpp:19-72): pool, unusable measurements, sort, triangulation — on the
        state handed in, which in the reference is the one BEFORE the point update is applied (UpdaterCamera.cpp:148-152)."""
        assert cpi is None
        pkg, mir = self.pkg, self.mir
        ct = [float(x) for x in st.t]
        dt = float(st.c.cam_dt)
        t_oldest, t_oldest2 = ct[0], ct[1]
        unused = {}

        def give(lid, e, i):
            u = unused.get(lid)
            if u is None:
                u = unused[lid] = dict(t=[], uv=[], uvn=[], points=list(e["points"]), D=e["D"])
            u["t"].append(e["t"][i]), u["uv"].append(e["uv"][i]), u["uvn"].append(e["uvn"][i])

        take = sorted(k for k, e in self.ldb.items() if any(t < t_oldest2 - dt for t in e["t"]) or not any(t > t_prev_frame - dt for t in e["t"]))
        pool = [(k, self.ldb.pop(k)) for k in take]
        kept = []
        for lid, e in pool:
            keep = []
            for i, t in enumerate(e["t"]):
                tm = t + dt
                if tm > state_time + 0.01:
                    give(lid, e, i)
                elif tm < t_oldest - 0.01:
                    continue
                else:
                    keep.append(i)
            if len(keep) >= 2:
                kept.append((lid, dict(t=[e["t"][i] for i in keep], uv=[e["uv"][i] for i in keep], uvn=[e["uvn"][i] for i in keep],
                                       points=e["points"], D=e["D"])))
        kept.sort(key=lambda x: -len(x[1]["t"]))     # stable
        lg, ok = np.zeros((0, 6)), np.zeros(0, dtype=np.uint8)
        if kept:
            ptr = np.concatenate([[0], np.cumsum([len(e["t"]) for _, e in kept])]).astype(np.int32)
            anchor, has = np.zeros((len(kept), 3)), np.zeros(len(kept), dtype=np.uint8)
            for l, (lid, e) in enumerate(kept):
                for pid in e["points"]:               # the first triangulated point of the line (REF LineHelper.cpp:233-247)
                    if pid in mir.used:
                        anchor[l], has[l] = mir.used[pid][0], 1
                        break
            lt_all = pkg.LineTracks(ptr, np.concatenate([e["t"] for _, e in kept]), np.concatenate([e["uv"] for _, e in kept]),
                                    seg_uvn=np.concatenate([e["uvn"] for _, e in kept]), D=[e["D"] for _, e in kept], anchor_pt=anchor,
                                    has_pt=has)
            lg, ok = self.jo.triangulate_lines(st, lt_all)
        self._prep = dict(n_pool=len(pool), kept=kept, unused=unused, give=give, lg=lg, ok=ok)

    # ---- UpdaterCamera::lines_update -> cleanup_lines (REF: UpdaterCamera.cpp:371-464, LineHelper.cpp:522-553) on what
    # camera_get_line_features prepared (two-call form without it: pool and triangulation on the state handed in here)
    def camera_update_lines(self, st, n, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, cap=512, cpi=None):
        assert cpi is None
        if getattr(self, "_prep", None) is None:
            self.camera_get_line_features(st, n, max_obs, t_prev_frame, state_time, window_full, chi2_mult)
        prep, self._prep = self._prep, None
        pkg, mir = self.pkg, self.mir
        ct = [float(x) for x in st.t]
        dt = float(st.c.cam_dt)
        t_oldest = ct[0]
        bounding = lambda t: mir._bounding(ct, t + dt)
        kept, unused, give, lg, ok = prep["kept"], prep["unused"], prep["give"], prep["lg"], prep["ok"]
        out = dict(dx=np.zeros(n), n_pool=prep["n_pool"], n_lines=0, n_accepted=0, n_rows=0, status=0, ids=[], accepted=[])
        if kept:
            sel, t_first = [], {}
            for l, (lid, e) in enumerate(kept):
                valid = sum(bounding(t) for t in e["t"])
                if not ok[l] or valid < 2 or len(sel) >= cap:
                    for i in range(len(e["t"])):
                        give(lid, e, i)
                    continue
                if valid > max_obs:
                    t_first[l] = valid - max_obs
                sel.append(l)
            if sel:
                tt, uv, counts = [], [], []
                for l in sel:
                    lid, e = kept[l]
                    c = seen = 0
                    for i, t in enumerate(e["t"]):
                        if not bounding(t):
                            give(lid, e, i)
                            continue
                        seen += 1
                        if seen <= t_first.get(l, 0):
                            continue
                        tt.append(t), uv.append(e["uv"][i])
                        c += 1
                    counts.append(c)
                sptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
                lt = pkg.LineTracks(sptr, np.array(tt), np.array(uv, dtype=np.float32), line_FinG=lg[sel])
                cols = self.jo.line_columns(st, lt)
                rows, Hf, Hx, res = self.jo.build_line_jacobians(st, lt, cols, 2 * max_obs)
                rc, P2, dx, acc, nrows = self.o.msckf_update(self.P, rows, Hf, Hx, res, cols, st.c.sigma_pix ** 2, self.q95, chi2_mult=chi2_mult,
                                                             res_norm_gate=0.0)
                out.update(n_lines=len(sel), ids=[kept[l][0] for l in sel], accepted=list(acc), n_rows=int(nrows), status=int(rc),
                           line_FinG=lg[sel])
                if rc == 0:
                    self.P = np.array(P2, order="F")
                    out["dx"] = dx
                out["n_accepted"] = int(np.sum(acc))
                for l, a in zip(sel, acc):
                    if not a:
                        lid, e = kept[l]
                        for i, t in enumerate(e["t"]):
                            if bounding(t):
                                give(lid, e, i)
        for lid, u in unused.items():
            d = self.ldb.get(lid)
            if d is None:
                d = self.ldb[lid] = dict(t=[], uv=[], uvn=[], points=list(u["points"]), D=u["D"])
            d["t"].extend(u["t"]), d["uv"].extend(u["uv"]), d["uvn"].extend(u["uvn"])
        if window_full:
            for lid in list(self.ldb):
                e = self.ldb[lid]
                keep = [i for i, t in enumerate(e["t"]) if not t < t_oldest]
                if not keep:
                    del self.ldb[lid]
                elif len(keep) != len(e["t"]):
                    for key in ("t", "uv", "uvn"):
                        e[key] = [e[key][i] for i in keep]
            for pid in [pid for pid, (_, newest) in mir.used.items() if newest < t_oldest]:   # point_used->cleanup_measurements
                del mir.used[pid]
        return out

    # ---- UpdaterWheel::update from the selected samples on: gate + EKFUpdate with the full noise matrix
    def wheel_update(self, opt, ws, t, m1, m2, n):
        H, res, Cov, cols, _, _ = self.po.wheel_linear_system(opt, ws, t, m1, m2)
        Hf = np.zeros((len(res), n))
        Hf[:, cols] = H
        P = self.P
        S = Hf @ P @ Hf.T + Cov
        chi2 = res @ np.linalg.solve(S, res)
        if not chi2 < opt.chi2_mult * self.q95[len(res)]:
            return 0, 0, np.zeros(n)
        K = P @ Hf.T @ np.linalg.inv(S)
        dP = K @ Hf @ P
        if (np.diag(P) - np.diag(dP) < 0).any():
            return self.pkg.PLV_E_NOT_PSD, 0, np.zeros(n)
        Pn = P - dP
        self.P = np.asfortranarray(0.5 * (Pn + Pn.T))
        return 0, 1, K @ res


class OracleContext(PyMirrorContext):
    """The camera path through the compiled CPU frame (oracle/frame_oracle.cpp).  Covariance bookkeeping between the frames (propagation,
    cloning, marginalisation, wheel updates) as in PyMirrorContext: numpy + the oracle's pieces on self.P."""

    def __init__(self, cfg):
        PyMirrorContext.__init__(self, cfg)
        self.frame = oracle_lib.FrameOracle(self.pkg, cfg, self.q95)
        self._lk_threads = 1

    def close(self):
        self.frame.close()

    @property
    def lk_threads(self):
        return self._lk_threads

    @lk_threads.setter
    def lk_threads(self, n):
        self._lk_threads = int(n)
        self.frame.set_threads(int(n))

    def _Pf(self):
        if not (self.P.flags.f_contiguous and self.P.dtype == np.float64):
            self.P = np.asfortranarray(self.P, dtype=np.float64)
        return self.P

    def set_camera_intrinsics(self, K8):
        self.K8 = np.array(K8, dtype=np.float64)
        self.frame.set_intrinsics(self.K8)

    def tracker_feed(self, t, img, mask=None):
        self.frame.tracker_feed(t, img, mask)

    def tracker_last(self):
        return self.frame.tracker_last()

    def line_tracker_last(self):
        return self.frame.line_last()

    def db_size(self):
        return self.frame.db_size()

    def db_cleanup_measurements(self, t):
        self.frame.db_cleanup_measurements(t)

    def line_db_size(self):
        return self.frame.line_db_size()

    def line_tracker_feed(self, t, vps):
        self.frame.line_feed(t, vps)

    def camera_update_points(self, st, n, max_msckf, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, min_dist=0.1,
                             max_dist=60.0, max_cond=1e4, max_baseline=40.0, refine=True, max_slam=0, slam_ids=(), init_min_meas=10,
                             cpi=None):
        assert max_slam == 0 and cpi is None
        return self.frame.update_points(self._Pf(), st, max_msckf, max_obs, t_prev_frame, state_time, window_full, chi2_mult, min_dist, max_dist,
                                        max_cond, max_baseline, refine)

    def camera_get_line_features(self, st, n, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, cpi=None):
        """LineHelper::get_line_features on the state handed in: in the reference it runs BEFORE the point update is applied
        (UpdaterCamera.cpp:148-152); camera_update_lines then uses what it prepared."""
        assert cpi is None
        self.frame.get_line_features(st, max_obs, t_prev_frame, state_time, window_full, chi2_mult)

    def camera_update_lines(self, st, n, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, cap=512, cpi=None):
        assert cpi is None
        return self.frame.update_lines(self._Pf(), st, max_obs, t_prev_frame, state_time, window_full, chi2_mult, cap)

    def last_point_decisions(self):
        return self.frame.last_point_decisions()

    def last_line_decisions(self):
        return self.frame.last_line_decisions()

    def camera_try_update(self, st, plus, n, max_msckf, max_obs, t_prev_frame, state_time, **kw):
        return self.frame.try_update(self._Pf(), st, dict(plus=plus, n=n, max_msckf=max_msckf, max_obs=max_obs, t_prev_frame=t_prev_frame,
                                                          state_time=state_time, **kw))

    def camera_frame(self, st, timestamp, slot=None, img=None, mask=None, use_lines=False, update=None):
        assert slot is None
        return self.frame.camera_frame(self._Pf() if self.P is not None else None, st, timestamp, img, mask, use_lines, update)


class OracleIwInitializer:
    """plviwo_amd.IwInitializer's interface over oracle/init_oracle.py."""

    def __init__(self, wheel_type, intrinsics, R_ItoO, p_IinO, toff, threshold, gravity=(0.0, 0.0, 9.81), imu_gravity_aligned=False):
        import os
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import init_oracle
        self.impl = init_oracle.IWInitializer(wheel_type, intrinsics, R_ItoO, p_IinO, toff, threshold, gravity, imu_gravity_aligned)

    def initialization(self, t, wm, am, tw, m1, m2):
        return self.impl.initialization(np.asarray(t), np.asarray(wm), np.asarray(am), np.asarray(tw), np.asarray(m1), np.asarray(m2))
