"""The slice of plviwo_amd.Context that pl-viwo_amd/system.py drives (IMU + one camera, points, wheel), served by the CPU oracle:
the same SystemManager then runs the whole filter on the CPU and its trajectory is the "CPU reference" of the replay tests.
Test infrastructure; compositions of oracle pieces as in test_gpu_tracker.py (frame logic) and test_gpu_dropin_sequence.py
(try_update)."""
import numpy as np

import oracle_lib
import synth
from test_gpu_dropin_sequence import MirrorUpdater


class OracleContext:
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
        rc, pts_new, ok, n0, n1 = self.fo.perform_matching(self.prev[1], pyr, pts_old, pts_old, self.K8)
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
        assert max_slam == 0 and cpi is None and chi2_mult == 1.0
        import test_gpu_dropin_sequence as m
        m.MAX_MSCKF, m.MAX_OBS = max_msckf, max_obs
        m.TRI = dict(min_dist=min_dist, max_dist=max_dist, max_cond=max_cond, max_baseline=max_baseline, refine=refine)
        ct = [float(x) for x in st.t]
        ref = self.mir.update(st, ct, self.P, t_prev_frame, state_time, window_full, st.c.sigma_pix)
        self.P = np.array(ref["P"], order="F")
        acc = np.array(ref["accepted"], dtype=np.uint8)
        return dict(dx=ref["dx"], n_pool=ref["n_pool"], n_msckf=len(ref["ids"]), n_accepted=int(acc.sum()), status=0, ids=ref["ids"],
                    accepted=acc, n_slam=0, n_init=0)

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
