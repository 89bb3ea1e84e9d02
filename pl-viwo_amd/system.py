"""SystemManager, State, Initializer, Propagator and the camera / wheel updaters as one host-side driver over the C-ABI
(SURVEY §8(f) rank 4): everything numeric runs behind `plv_*` on the GPU, this module keeps the state vector, the clone
window and the measurement buffers, in the order the reference does.

REF: PL-VIWO/src/core/SystemManager.cpp:55-135 (feed_measurement_imu / _camera / _wheel), :172-312 (clone schedule, accelerations,
     dynamic cloning); PL-VIWO/src/state/State.cpp:30-120,228-268 (variable order, prior), StateHelper.cpp:120-172,175-232;
     PL-VIWO/src/state/Propagator.cpp:17-91,333-357; PL-VIWO/src/init/Initializer.cpp:58-180;
     PL-VIWO/src/update/cam/UpdaterCamera.cpp:77-195; PL-VIWO/src/update/wheel/UpdaterWheel.cpp:21-139.

The camera path follows the intended flow feed_measurement -> try_update (as published, SystemManager.cpp:107-127 returns
before try_update once the filter is initialised).  Scope of this driver: one camera (monocular), MSCKF points and lines
(in-state SLAM landmarks in the GLOBAL_3D representation; the shipped configuration has cam.max_slam: 0), wheel optional; `use_imu_res` takes the poses of the camera update from the CPI records of plv_propagate; GPS / LiDAR / stereo /
simulation are outside SURVEY §8.
"""
import ctypes
import os
import math
import time as _time

import numpy as np

from . import BoxPlus, PlvStateView, jpl_left_update
from . import (Context, CpiTable, IwInitializer, PlvError, PlvImuState, PlvWheelOptions, PlvWheelState, StateView, Tracks, WHEEL_TYPES, default_config,
               imu_noise, init_imu_static, next_clone_time, reset_cpi, select_imu_readings, select_wheel_data)
from .options import OptionsError


# ------------------------------------------------------------------------------------------------ JPL helpers (host side)
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def quat_2_Rot(q):
    """REF: open_vins/ov_core/src/utils/quat_ops.h:152-157"""
    q = np.asarray(q, dtype=np.float64)
    return (2 * q[3] ** 2 - 1) * np.eye(3) - 2 * q[3] * skew(q[:3]) + 2 * np.outer(q[:3], q[:3])


def quat_left_update(q, dth):
    """JPLQuat::update: q <- quatnorm([dth / 2, 1]) (x) q  (REF: open_vins/ov_core/src/types/JPLQuat.h:62-73)."""
    dq = np.array([0.5 * dth[0], 0.5 * dth[1], 0.5 * dth[2], 1.0])
    dq /= np.linalg.norm(dq)
    M = np.zeros((4, 4))
    M[:3, :3] = dq[3] * np.eye(3) - skew(dq[:3])
    M[:3, 3], M[3, :3], M[3, 3] = dq[:3], -dq[:3], dq[3]
    r = M @ np.asarray(q, dtype=np.float64)
    if r[3] < 0:
        r = -r
    return r / np.linalg.norm(r)


def quats_2_Rots(Q):
    """quat_2_Rot for a stack of quaternions [n][4] -> [n][3][3] (one numpy pass per window instead of a Python call per clone)."""
    Q = np.asarray(Q, dtype=np.float64).reshape(-1, 4)
    v, w = Q[:, :3], Q[:, 3]
    K = np.zeros((len(Q), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -v[:, 2], v[:, 1], v[:, 2], -v[:, 0], -v[:, 1], v[:, 0]
    return (2 * w * w - 1)[:, None, None] * np.eye(3) - (2 * w)[:, None, None] * K + 2 * v[:, :, None] * v[:, None, :]


def quats_left_update(Q, dth):
    """quat_left_update for stacks: Q [n][4], dth [n][3]."""
    Q = np.asarray(Q, dtype=np.float64).reshape(-1, 4)
    dq = np.concatenate([0.5 * np.asarray(dth, dtype=np.float64).reshape(-1, 3), np.ones((len(Q), 1))], axis=1)
    dq /= np.linalg.norm(dq, axis=1)[:, None]
    a, b, v, w = dq[:, :3], dq[:, 3], Q[:, :3], Q[:, 3]
    r = np.empty_like(Q)
    r[:, :3] = b[:, None] * v - np.cross(a, v) + a * w[:, None]
    r[:, 3] = -np.einsum("ij,ij->i", a, v) + b * w
    r[r[:, 3] < 0] *= -1
    return r / np.linalg.norm(r, axis=1)[:, None]


class Stat:
    """viw::STAT (REF: PL-VIWO/src/utils/Jabdongsani.cpp:8-33), float arithmetic as there."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.mean, self.var, self.cnt = np.float32(0), np.float32(0), np.float32(0)

    def add_stat(self, val):
        val = np.float32(val)
        self.cnt = np.float32(self.cnt + 1)
        if self.cnt > 1:
            self.var = np.float32((self.cnt - 2) / (self.cnt - 1) * self.var + 1.0 / self.cnt * (val - self.mean) ** 2)
        self.mean = np.float32((val + (self.cnt - 1) * self.mean) / self.cnt)


class Pose:
    """ov_type::PoseJPL: value and first estimate."""

    def __init__(self, q, p, var_id=-1):
        self.q, self.p = np.array(q, dtype=np.float64), np.array(p, dtype=np.float64)
        self.q_fej, self.p_fej = self.q.copy(), self.p.copy()
        self.id = var_id
        self._R = self._Rf = None     # rotation matrices of q / q_fej, built when first asked for, dropped when q changes

    def Rot(self):
        if self._R is None:
            self._R = quat_2_Rot(self.q)
        return self._R

    def Rot_fej(self):
        if self._Rf is None:
            self._Rf = quat_2_Rot(self.q_fej)
        return self._Rf

    def update(self, dx):
        self.q[:] = quat_left_update(self.q, dx[0:3])
        self.p[:] = self.p + dx[3:6]
        if self._R is not None:      # in place: State's prepared boxplus call and the state view hold on to the array
            self._R[...] = quat_2_Rot(self.q)

    def clone(self, var_id):
        c = Pose(self.q, self.p, var_id)
        c.q_fej, c.p_fej = self.q_fej.copy(), self.p_fej.copy()
        return c


class Vec:
    def __init__(self, v, var_id=-1):
        self.v = np.array(v, dtype=np.float64).ravel()
        self.id = var_id

    def update(self, dx):
        self.v += dx      # in place (State's prepared boxplus call holds on to the array)


class Landmark:
    """ov_type::Landmark, GLOBAL_3D: value, first estimate, covariance index (REF: open_vins/ov_core/src/types/Landmark.h)."""

    def __init__(self, featid, p, var_id):
        self.featid, self.p, self.p_fej, self.id = int(featid), np.array(p, dtype=np.float64), np.array(p, dtype=np.float64), var_id
        self.update_fail_count = 0   # read by marginalize_slam_features (UpdaterCamera.cpp:130); nothing in the reference increments it


class State:
    """viw::State: mean, variable order, clone window (the covariance itself is resident in the plv_ctx)."""

    def __init__(self, op, ctx):
        self.op, self.ctx = op, ctx
        e = op.est
        self.time, self.startup_time, self.initialized = -1.0, -1.0, False
        self.imu = PlvImuState.make([0, 0, 0, 1], [0, 0, 0], [0, 0, 0])
        self.clones = {}            # time -> Pose (real clones; the IMU pose pseudo-clone is added by clone_list())
        self._win, self._win_dirty = None, True   # array form of the window (_window_arrays)
        self.slam = {}              # feature id -> Landmark (State::cam_SLAM_features)
        self.imu_pose_in_clones = False
        self.cpis = {}              # time -> PlvCpiRecord-like dict (t, dt, clone_t, R, w, v)
        self.est_a, self.est_A = Stat(), Stat()
        cur = 15                    # State.cpp:27-32: the IMU first
        self.cam_ext = self.cam_intr = self.cam_dt = None
        prior = []                  # (id, values on the diagonal)
        if e.cam.enabled:           # set_camera_state :58-112: extrinsic, intrinsic, time offset
            c = e.cam
            self.cam_ext = Pose(c.extrinsics[0][:4], c.extrinsics[0][4:])
            self.cam_intr, self.cam_dt = Vec(c.intrinsics[0]), Vec([c.dt[0]])
            if c.do_calib_ext:
                self.cam_ext.id = cur
                prior.append((cur, [c.init_cov_ex_o] * 3 + [c.init_cov_ex_p] * 3))
                cur += 6
            if c.do_calib_int:
                self.cam_intr.id = cur
                r2 = (math.sqrt(c.init_cov_in_r) / 10.0) ** 2 if c.init_cov_in_r > 1e-5 else c.init_cov_in_r   # State.cpp:245-249
                prior.append((cur, [c.init_cov_in_k] * 2 + [c.init_cov_in_c] * 2 + [c.init_cov_in_r] * 2 + [r2] * 2))
                cur += 8
            if c.do_calib_dt:
                self.cam_dt.id = cur
                prior.append((cur, [c.init_cov_dt]))
                cur += 1
        self.wheel_ext = self.wheel_intr = self.wheel_dt = None
        if e.wheel.enabled:         # set_wheel_state :151-190: time offset, extrinsic, intrinsic
            w = e.wheel
            self.wheel_dt, self.wheel_ext, self.wheel_intr = Vec([w.dt]), Pose(w.extrinsics[:4], w.extrinsics[4:]), Vec(w.intrinsics)
            if w.do_calib_dt:
                self.wheel_dt.id = cur
                prior.append((cur, [w.init_cov_dt]))
                cur += 1
            if w.do_calib_ext:
                self.wheel_ext.id = cur
                prior.append((cur, [w.init_cov_ex_o] * 3 + [w.init_cov_ex_p] * 3))
                cur += 6
            if w.do_calib_int:
                self.wheel_intr.id = cur
                prior.append((cur, [w.init_cov_in_r] * 2 + [w.init_cov_in_b]))
                cur += 3
        self.n = cur
        # set_state_covariance :228-268 (the IMU block is overwritten at initialisation, Initializer.cpp:182-186)
        P = e.init.cov_size * np.eye(cur)
        P[6:9, 6:9] *= 2
        for i, d in prior:
            P[i:i + len(d), i:i + len(d)] = np.diag(d)
        ctx.cov_upload(P)

    # ---- clone bookkeeping
    def clone_times(self):
        t = sorted(self.clones)
        if self.imu_pose_in_clones and self.time not in self.clones:
            t.append(self.time)
        return t

    def _window_arrays(self):
        """The clone window as contiguous arrays (times, q, p, first estimates, rotation matrices, covariance ids) with one spare row for
        the IMU pose; rebuilt when the set of clones changed, updated in place by apply().  The Pose objects of the clones are views of
        their rows, so the two never disagree."""
        w = self._win
        if w is not None and not self._win_dirty:
            return w
        ts = sorted(self.clones)
        n = len(ts)
        w = dict(n=n, t=np.zeros(n + 1), q=np.zeros((n + 1, 4)), p=np.zeros((n + 1, 3)), qf=np.zeros((n + 1, 4)), pf=np.zeros((n + 1, 3)),
                 R=np.zeros((n + 1, 9)), Rf=np.zeros((n + 1, 9)), id=np.zeros(n + 1, dtype=np.int32))
        for i, t in enumerate(ts):
            c = self.clones[t]
            w["t"][i], w["id"][i] = t, c.id
            w["q"][i], w["p"][i], w["qf"][i], w["pf"][i] = c.q, c.p, c.q_fej, c.p_fej
            w["R"][i], w["Rf"][i] = c.Rot().ravel(), c.Rot_fej().ravel()
            c.q, c.p, c.q_fej, c.p_fej = w["q"][i], w["p"][i], w["qf"][i], w["pf"][i]      # views from here on
            c._R, c._Rf = w["R"][i].reshape(3, 3), w["Rf"][i].reshape(3, 3)
        self._win, self._win_dirty = w, False
        # the state view wraps the window arrays themselves and is refreshed in place (view()); the prepared boxplus call lists every
        # variable dx moves, with the view's calibration fields as mirrors, so that apply() is one C call and leaves the view current
        c = self.op.est.cam
        if self.cam_ext is not None:
            self.cam_ext.Rot()
            sv = w["sv"] = StateView(w["t"], w["R"], w["p"], w["id"], self.cam_ext.Rot(), self.cam_ext.p, self.cam_intr.v,
                                     clone_R_fej=w["Rf"], clone_p_fej=w["pf"], cam_dt=float(self.cam_dt.v[0]),
                                     extrinsic_state_id=self.cam_ext.id, intrinsic_state_id=self.cam_intr.id, dt_state_id=self.cam_dt.id,
                                     sigma_pix=c.sigma_pix, use_pol_cov=1 if self.op.est.use_pol_cov else 0,
                                     feat_rep=c.feat_rep, dt_exp=self.op.est.dt_exp, use_imu_cov=1 if self.op.est.use_imu_cov else 0,
                                     intr_err_mlt=self.op.est.intr_err.mlt)
            assert np.shares_memory(sv.R, w["R"]) and np.shares_memory(sv.p, w["p"]) and np.shares_memory(sv.t, w["t"])    # no copies
            base = ctypes.addressof(sv.c)
            m_R, m_p, m_K, m_dt = (base + getattr(PlvStateView, f).offset for f in ("R_ItoC", "p_IinC", "intrinsics", "cam_dt"))
        else:
            m_R = m_p = m_K = m_dt = None
        x = np.frombuffer(self.imu, dtype=np.float64)      # q (4), p, v, bg, ba (3 each), then the first estimates
        # (row n of the window arrays is the IMU pose when it closes the clone list, view(): kept current as well)
        ent = [("quat", 0, x[0:4], w["R"][n], None), ("vec", 3, x[4:7], None, w["p"][n]), ("vec", 6, x[7:16], None, None)]
        for var, mr, mp in ((self.cam_ext, m_R, m_p), (self.wheel_ext, None, None)):
            if var is not None and var.id >= 0:
                var.Rot()
                ent += [("quat", var.id, var.q, var._R.reshape(9), mr), ("vec", var.id + 3, var.p, None, mp)]
        for var, mv in ((self.cam_intr, m_K), (self.cam_dt, m_dt), (self.wheel_dt, None), (self.wheel_intr, None)):
            if var is not None and var.id >= 0:
                ent.append(("vec", var.id, var.v, None, mv))
        for i in range(n):
            ent += [("quat", int(w["id"][i]), w["q"][i], w["R"][i], None), ("vec", int(w["id"][i]) + 3, w["p"][i], None, None)]
        w["plus"], w["imu"] = BoxPlus(ent), self.imu
        return w

    def refresh_window(self):
        """Rebuilds the array form of the window now (after a clone was added or marginalised) rather than at its next use."""
        self._window_arrays()

    def imu_pose(self):
        p = Pose(self.imu.q, self.imu.p, 0)
        p.q_fej, p.p_fej = np.array(self.imu.q_fej), np.array(self.imu.p_fej)
        return p

    def clone_at(self, t):
        return self.clones[t] if t in self.clones else self.imu_pose()

    def clone_window(self):
        t = self.clone_times()
        return t[-1] - t[0] if t else 0.0

    def intr_cov(self):
        ie, e = self.op.est.intr_err, self.op.est
        hz, order = e.clone_freq, e.intr_order
        if not e.use_pol_cov or hz not in ie.ori_slope:
            return 0.0, 0.0
        return ie.ori_cov(hz, order, float(self.est_A.mean)), ie.pos_cov(hz, order, float(self.est_a.mean))

    def view(self):
        """The plv_state_view of the current window: real clones, then the IMU pose (id 0) when it sits in the clone list."""
        w = self._window_arrays()
        m = w["n"]
        if self.imu_pose_in_clones and self.time not in self.clones:    # the IMU pose closes the list (covariance id 0)
            x = np.frombuffer(self.imu, dtype=np.float64)   # q, p, v, bg, ba, q_fej, p_fej, v_fej
            w["t"][m], w["id"][m] = self.time, 0
            w["q"][m], w["p"][m], w["qf"][m], w["pf"][m] = x[0:4], x[4:7], x[16:20], x[20:23]
            jpl_left_update(w["q"][m:m + 1], None, w["R"][m:m + 1])
            jpl_left_update(w["qf"][m:m + 1], None, w["Rf"][m:m + 1])
            m += 1
        oc, pc = self.intr_cov()
        sv = w["sv"]
        v = sv.c
        v.n_clones = m
        sv.t = w["t"][:m]          # (what the tests and the oracle context read)
        v.intr_ori_cov, v.intr_pos_cov = oc, pc        # (the calibration fields are kept current by apply())
        return sv

    def cpi_table(self):
        """State::cpis as the plv_cpi_table of plv_cpi_poses / plv_update_options::cpi."""
        ts = sorted(self.cpis)
        r = [self.cpis[t] for t in ts]
        return CpiTable(ts, [x["clone_t"] for x in r], [x["R"] for x in r], [x["alpha"] for x in r], [x["v"] for x in r],
                        gravity=tuple(self.op.est.gravity), Q=[x.get("Q", np.zeros(36)) for x in r] if self.op.est.use_imu_cov else None)

    # ---- x <- x [+] dx for every variable (StateHelper::EKFUpdate :156-168)
    def apply(self, dx):
        w = self._window_arrays()
        if w["imu"] is not self.imu:      # the IMU state object was replaced (initialisation): lay the call out again
            self._win_dirty = True
            w = self._window_arrays()
        w["plus"].apply(dx)               # every variable, the clone window and the state view's calibration fields: one C call
        for lm in self.slam.values():
            lm.p = lm.p + dx[lm.id:lm.id + 3]
        if self.cam_intr is not None and self.op.est.cam.do_calib_int:
            self.ctx.set_camera_intrinsics(self.cam_intr.v)     # StateHelper.cpp:163-168

    # ---- StateHelper::augment_clone / marginalize_old_clone
    def augment_clone(self):
        if self.time in self.clones:
            raise RuntimeError("TRIED TO INSERT A CLONE AT THE SAME TIME AS AN EXISTING CLONE")   # StateHelper.cpp:178-181
        self.ctx.cov_clone(self.n, 0, 6)
        self.clones[self.time] = self.imu_pose().clone(self.n)
        self.n += 6
        self._win_dirty = True

    def marginalize_old_clone(self):
        while self.clone_window() > self.op.est.window_size and self.clones:
            c = self.clones.pop(min(self.clones))
            self._win_dirty = True
            self.marginalize(c.id, 6)

    def marginalize(self, var_id, size):   # StateHelper::marginalize :234-303: later variables move up
        self.ctx.cov_marginalize(var_id, size)
        self.n -= size
        for o in list(self.clones.values()) + list(self.slam.values()):
            if o.id > var_id:
                o.id -= size
        self._win_dirty = True

    def flush_old_data(self):   # State.cpp:605-628
        if not self.clones:
            return
        old_t = min(self.clones)
        for t in sorted(self.cpis):
            if old_t > self.cpis[t]["clone_t"]:
                del self.cpis[t]
            else:
                break


class SampleBuffer:
    """Time-ordered measurement buffer (Propagator::imu_data, UpdaterWheel::data_stack) as one growing array with a moving front:
    appending and dropping old samples cost nothing per message and the live part is handed to the C-ABI without a copy."""

    def __init__(self, width):
        self.a = np.zeros((1024, width))
        self.lo = self.hi = 0

    def __len__(self):
        return self.hi - self.lo

    def append(self, row):
        if self.hi == len(self.a):
            n = self.hi - self.lo
            if self.lo >= len(self.a) // 2:
                self.a[:n] = self.a[self.lo:self.hi]
            else:
                b = np.zeros((2 * len(self.a), self.a.shape[1]))
                b[:n] = self.a[self.lo:self.hi]
                self.a = b
            self.lo, self.hi = 0, n
        self.a[self.hi] = row
        self.hi += 1

    def drop_before(self, t):
        """erases the samples with time < t (column 0)"""
        k = int(np.searchsorted(self.a[self.lo:self.hi, 0], t, side="left"))
        self.lo += k
        return k

    def view(self):
        return self.a[self.lo:self.hi]

    def t(self, i):
        return float(self.a[self.hi + i if i < 0 else self.lo + i, 0])


class TimeChecker:
    """ding / dong totals per label (REF: PL-VIWO/src/utils/TimeChecker.h:56-135)."""

    def __init__(self):
        self.total, self.count, self._t0 = {}, {}, {}

    def ding(self, k):
        self._t0[k] = _time.perf_counter()

    def dong(self, k):
        self.total[k] = self.total.get(k, 0.0) + _time.perf_counter() - self._t0.pop(k)
        self.count[k] = self.count.get(k, 0) + 1


class SystemManager:
    """viw::SystemManager for IMU + one camera (+ wheel)."""

    one_call_frame = os.environ.get("PLV_ONE_CALL", "2") == "2"
    one_call_update = True      # try_update through plv_camera_try_update (False: plv_camera_update_points / _lines with the dx applied here)

    def __init__(self, op, device=0, max_obs=24, context_factory=None, iw_initializer_factory=None, decisions=None):
        """context_factory / iw_initializer_factory: stand-ins with the interface of Context / IwInitializer (the tests run the same
        driver over the CPU oracle through them); the defaults are the HIP library.  decisions: a list that receives one record per
        camera update — (kind, frame, state time, pool size, ids that were triangulated, accepted flags, status, the values behind the verdicts
        where the context reports them, dx) — for the
        comparison of two runs decision by decision (tests/decision_trace.py)."""
        self.decisions = decisions
        e = op.est
        if e.cam.enabled and e.cam.max_n != 1:
            raise OptionsError("replay driver: one camera (cam.max_n: 1, use_stereo: false)")
        if e.cam.enabled and e.cam.max_slam != 0 and e.cam.feat_rep != 0:
            raise OptionsError("replay driver: in-state landmarks (cam.max_slam > 0) are driven in the GLOBAL_3D representation only")
        if e.cam.enabled and e.cam.distortion_model[0] != "radtan":
            raise OptionsError("only the radtan camera model is built (SURVEY §8 a7)")
        if e.use_imu_cov and not e.use_pol_cov and not e.use_imu_res:
            # CamHelper.cpp:217-224 reads state->cpis.at(tm + dt), a record that only get_interpolated_pose_imu creates (use_imu_res)
            raise OptionsError("est.use_imu_cov needs est.use_imu_res (the CPI records behind the observation poses)")
        if e.init.use_gt:
            raise OptionsError("init.use_gt needs the simulator / ground-truth reader, outside SURVEY §8")
        self.op = op
        w, h = (e.cam.wh[0] if e.cam.enabled else (752, 480))
        cfg = default_config(w, h)
        if e.cam.enabled:
            c = e.cam
            cfg.num_features, cfg.fast_threshold, cfg.grid_x, cfg.grid_y, cfg.min_px_dist = c.n_pts, c.fast, c.grid_x, c.grid_y, c.min_px_dist
            cfg.histogram_method = c.histogram
            for i in range(8):
                cfg.intrinsics[i] = float(c.intrinsics[0][i])
            cfg.sigma_pix, cfg.chi2_mult = c.sigma_pix, c.chi2_mult
        clones_max = int(e.window_size * max(e.clone_freq, max(e.intr_err.available_clone_hz() or [e.clone_freq]))) + 3
        cfg.max_state_dim = max(cfg.max_state_dim, 15 + 30 + 6 * clones_max + 3 * (e.cam.max_slam if e.cam.enabled else 0))
        cfg.max_rows_per_feat = max(cfg.max_rows_per_feat, 2 * max_obs)
        cfg.device = device
        self.ctx = (context_factory or Context)(cfg)
        import os
        if os.environ.get("PLV_AHEAD") and hasattr(self.ctx, "tracker_detect_ahead"):
            self.ctx.tracker_detect_ahead(int(os.environ["PLV_AHEAD"]))
        self.max_obs = max_obs
        if decisions is not None and hasattr(self.ctx, "decision_trace"):
            self.ctx.decision_trace(True)
        self.state = State(op, self.ctx)
        self.noise = imu_noise(e.imu.sigma_w, e.imu.sigma_wb, e.imu.sigma_a, e.imu.sigma_ab, tuple(e.gravity))
        # Propagator
        self.imu = SampleBuffer(7)   # t, wm (3), am (3)
        self.cpi_acc = None
        # UpdaterCamera
        self.cam_t_hist = []
        self.use_lines = bool(e.cam.enabled and e.cam.use_lines)
        if self.use_lines and hasattr(self.ctx, "line_prefetch_mode"):
            self.ctx.line_prefetch_mode(True)     # the tracker feed detects the frame's lines while the device tracks the points
        # UpdaterWheel
        self.whl = SampleBuffer(3)   # t, m1, m2
        self.whl_last_updated = -1.0
        self.wheel_opt = None
        if e.wheel.enabled:
            wl = e.wheel
            self.wheel_opt = PlvWheelOptions(WHEEL_TYPES[wl.type], wl.noise_w, wl.noise_v, wl.noise_p, int(wl.do_calib_ext), int(wl.do_calib_dt),
                                             int(wl.do_calib_int), wl.chi2_mult)
        # Initializer (REF: Initializer.cpp:58-91)
        self.iw_init = None
        if not e.init.imu_only_init and e.wheel.enabled:
            st = self.state
            self.iw_init = (iw_initializer_factory or IwInitializer)(e.wheel.type, st.wheel_intr.v, st.wheel_ext.Rot(), st.wheel_ext.p, float(st.wheel_dt.v[0]),
                                         e.init.imu_wheel_thresh, e.gravity, e.init.imu_gravity_aligned)
        self.last_cam_delete_t = -math.inf
        self.tc = TimeChecker()
        self._lines_in_flight = False
        self.stats = dict(clones=0, cam_updates=0, cam_features=0, cam_accepted=0, line_updates=0, lines_accepted=0, wheel_updates=0, wheel_accepted=0,
                          not_psd=0, frames=0, line_pool=0, lines_triangulated=0, lines_tracked=0, slam_initialized=0, slam_updates=0,
                          slam_marginalized=0)
        self.distance = 0.0

    def close(self):
        self.ctx.close()

    # ================================================================================================ Initializer
    def _try_initialization(self):
        e = self.op.est
        v = self.imu.view()
        t, wm, am = np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1:4]), np.ascontiguousarray(v[:, 4:7])
        if self.iw_init is not None:
            w = self.whl.view()
            x = self.iw_init.initialization(t, wm, am, np.ascontiguousarray(w[:, 0]), np.ascontiguousarray(w[:, 1]), np.ascontiguousarray(w[:, 2]))
        else:
            x = init_imu_static(t, wm, am, e.init.window_time, e.init.imu_thresh, e.gravity)
        if x is None:
            self._delete_old_measurements()
            return False
        self._set_state(x)
        return True

    def _delete_old_measurements(self):   # Initializer.cpp:115-172
        W = self.op.est.init.window_time
        if not len(self.imu) or self.imu.t(-1) - self.imu.t(0) <= 3 * W:
            return
        old = self.imu.t(-1) - 3 * W
        self.imu.drop_before(old)
        if self.op.est.cam.enabled and len(self.cam_t_hist) > 1:
            cam_hz = (len(self.cam_t_hist) - 1) / (self.cam_t_hist[-1] - self.cam_t_hist[0])
            if self.last_cam_delete_t + 100.0 / cam_hz < old:
                self.ctx.db_cleanup_measurements(old)
                self.last_cam_delete_t = old
        if self.op.est.wheel.enabled:
            self.whl.drop_before(old)

    def _set_state(self, x):   # Initializer.cpp:174-220
        st = self.state
        st.imu = PlvImuState.make(x[1:5], x[5:8], x[8:11], x[11:14], x[14:17])
        P = self.ctx.cov_download(st.n)
        P[:15, :] = 0
        P[:, :15] = 0
        P[:15, :15] = self.op.est.init.cov_size * np.eye(15)
        self.ctx.cov_upload(P)
        st.time = st.startup_time = float(x[0])
        st.initialized = True

    # ================================================================================================ Propagator
    def _feed_imu(self, t, wm, am):   # Propagator.cpp:17-28
        self.imu.append((t, wm[0], wm[1], wm[2], am[0], am[1], am[2]))
        st = self.state
        if st.clones:
            self.imu.drop_before(min(st.clones) - 1)

    def _propagate(self, timestamp):   # Propagator.cpp:30-91
        st = self.state
        v = self.imu.view()
        # only the tail that can matter: from the sample before the state time on (the buffer reaches a second further back)
        k = max(0, int(np.searchsorted(v[:, 0], st.time, side="right")) - 2)
        ok, t, wm, am = select_imu_readings(np.ascontiguousarray(v[k:, 0]), np.ascontiguousarray(v[k:, 1:4]), np.ascontiguousarray(v[k:, 4:7]),
                                            st.time, timestamp)
        if not ok:
            return
        _, _, recs = self.ctx.propagate(st.imu, self.noise, t, wm, am, st.n, acc=self.cpi_acc, imu_id=0, want_records=True)
        for r in recs:
            st.cpis[r.t] = dict(t=r.t, dt=r.dt, clone_t=r.clone_t, R=np.array(r.R_I0toIk).reshape(3, 3), alpha=np.array(r.alpha),
                                w=np.array(r.w), v=np.array(r.v), Q=np.array(r.Q))
        st.time = float(timestamp)
        if self.decisions is not None and getattr(self.decisions, "probe_state", False):
            self.decisions.states_prop.append((float(timestamp), len(t), np.concatenate([np.asarray(st.imu.q, float), np.asarray(st.imu.p, float), np.asarray(st.imu.v, float)]),
                                               np.array(t, float), np.array(am, float)))

    def _reset_cpi(self, clone_t):   # Propagator.cpp:333-357
        st = self.state
        self.cpi_acc = reset_cpi(st.imu, clone_t)
        w = st.cpis[clone_t]["w"] if clone_t in st.cpis else np.zeros(3)
        st.cpis[clone_t] = dict(t=clone_t, dt=0.0, clone_t=clone_t, R=np.eye(3), alpha=np.zeros(3), w=w, v=np.array(st.imu.v), Q=np.zeros(36))

    # ================================================================================================ SystemManager
    def feed_measurement_imu(self, t, wm, am):
        """REF: SystemManager.cpp:55-105.  Returns True when a clone was created (the reference's cue to log / visualise)."""
        self._feed_imu(t, wm, am)
        st = self.state
        if not st.initialized and not self._try_initialization():
            return False
        self.tc.ding("IMU")
        st.imu_pose_in_clones = False                      # erase the IMU pose from the clone list
        clone_time = self._get_next_clone_time(t)
        if clone_time is not None:
            self._propagate(clone_time)
            st.augment_clone()
            st.marginalize_old_clone()
            self._reset_cpi(st.time)
            st.est_A.reset(), st.est_a.reset()
            st.flush_old_data()
            st.refresh_window()       # window maintenance belongs to the cloning step, not to the next camera frame
            self._cov_probe("after propagation to the clone time, cloning and marginalisation")
            self.stats["clones"] += 1
            ct = sorted(st.clones)
            if len(ct) > 1:
                self.distance += float(np.linalg.norm(st.clones[ct[-1]].p - st.clones[ct[-2]].p))
        self._propagate(t)
        if st.time not in st.clones:
            st.imu_pose_in_clones = True
        self.tc.dong("IMU")
        return clone_time is not None

    def _get_next_clone_time(self, meas_t):   # SystemManager.cpp:172-267
        st, e = self.state, self.op.est
        if not st.initialized:
            return None
        if not st.clones:
            return st.time
        self._compute_accelerations()
        freq = e.clone_freq
        if e.dynamic_cloning:
            freq = self._dynamic_cloning()
        ct = sorted(st.clones)
        sensor_t = self.cam_t_hist if e.cam.enabled else []
        sensor_dt = float(st.cam_dt.v[0]) if e.cam.enabled else 0.0
        r = next_clone_time(len(ct), st.time, meas_t, ct[-1], ct[-2] if len(ct) > 1 else -math.inf, False, freq, sensor_t, sensor_dt,
                            self.imu.t(0), self.imu.t(-1), wheel_enabled=e.wheel.enabled)
        if r is not None and e.dynamic_cloning:
            e.clone_freq, e.intr_order = freq, 3   # SystemManager.cpp:186-189,253-256 overwrite both
        return r

    def _compute_accelerations(self):   # SystemManager.cpp:269-295
        st = self.state
        if len(self.imu) < 2:
            return
        t1 = self.imu.t(-2)
        c1 = st.cpis.get(t1)
        if c1 is None or c1["clone_t"] not in st.clones:
            return   # State::have_cpi would try to create one (linear / integrated): only stored records are used here
        R_I0toG = st.clones[c1["clone_t"]].Rot().T
        R_IktoI0 = c1["R"].T
        a = R_I0toG @ R_IktoI0 @ self.imu.view()[-2, 4:7] - quat_2_Rot(st.imu.q).T @ np.array(st.imu.ba) - self.op.est.gravity
        st.est_a.add_stat(np.linalg.norm(a))
        if c1["dt"] != 0:
            st.est_A.reset()
            c0 = st.cpis.get(c1["clone_t"])
            if c0 is not None:
                st.est_A.add_stat(np.linalg.norm((R_IktoI0 @ c1["w"] - c0["w"]) / c1["dt"]))

    def _dynamic_cloning(self):   # SystemManager.cpp:297-312
        st, ie = self.state, self.op.est.intr_err
        hzs = ie.available_clone_hz()
        if not hzs:
            return self.op.est.clone_freq
        for hz in hzs:
            if hz < 4:
                continue
            if ie.ori_std(hz, 3, float(st.est_A.mean)) < ie.threshold_ori and ie.pos_std(hz, 3, float(st.est_a.mean)) < ie.threshold_pos:
                return hz
        return hzs[-1]

    # ---------------------------------------------------------------------------------------------- camera
    def camera_prepare(self, t, img, mask=None, staged_slot=None, _bookkeeping_done=False):
        """First third of feed_measurement_camera on the one-call path (plv_camera_frame): the frame-time bookkeeping and the call's
        arguments marshalled into their C structures (the state view, the update options).  Returns the record camera_run /
        camera_finish take, or None when this frame has to go the long way (feed_measurement_camera does that by itself).  bench.py
        calls the three parts separately so that its timed region is the C-ABI call alone, as a C++ caller would make it."""
        e, st = self.op.est, self.state
        if not e.cam.enabled:
            return None
        if not (self.one_call_update and self.one_call_frame and e.cam.max_slam == 0 and not e.use_imu_res and not e.cam.downsample and not st.slam
                and hasattr(self.ctx, "camera_frame")):
            return None
        if not _bookkeeping_done:
            if st.initialized:
                self.tc.ding("CAM")
            if len(self.cam_t_hist) > 100:
                self.cam_t_hist.pop(0)
            self.cam_t_hist.append(float(t))
        # feed_measurement + try_update in one library call (plv_camera_frame)
        upd = None
        args = self._try_update_args() if st.initialized else None
        if self.decisions is not None and getattr(self.decisions, "probe_state", False) and args is not None:
            self.decisions.states_pre.append((self.stats["frames"], self._state_probe(), self.ctx.cov_download(st.n) if hasattr(self.ctx, "cov_download") else None))
        sv = st.view()
        if args is not None:
            pk, kw, max_msckf = args
            upd = dict(plus=st._window_arrays()["plus"], n=st.n, max_msckf=max_msckf, max_obs=self.max_obs, lines=self.use_lines, **pk, **kw)
        kw = dict(slot=staged_slot, img=img, mask=mask, use_lines=self.use_lines, update=upd)
        if hasattr(self.ctx, "camera_frame_prepare"):
            return dict(frame=self.ctx.camera_frame_prepare(sv, t, **kw))
        return dict(frame=None, call=(sv, t, kw), out=None)      # (a context without the split: the CPU frame of tests/oracle_context.py)

    def camera_run(self, prep, sync=True):
        """plv_camera_frame + plv_ctx_synchronize on the prepared arguments"""
        if prep["frame"] is not None:
            self.ctx.camera_frame_run(prep["frame"], sync=sync)
            return
        sv, t, kw = prep["call"]
        prep["out"] = self.ctx.camera_frame(sv, t, **kw)
        if sync and hasattr(self.ctx, "synchronize"):
            self.ctx.synchronize()

    def camera_finish(self, prep):
        """Last third: return codes checked, the update's results counted"""
        st = self.state
        out, lo, n_db = self.ctx.camera_frame_collect(prep["frame"]) if prep["frame"] is not None else prep["out"]
        self._lines_in_flight = False
        self.stats["frames"] += 1
        if self.use_lines:
            self.stats["lines_tracked"] += n_db
        if out is not None:
            self._count_points(out)
            if lo is not None:
                self._count_lines(lo)
        if st.initialized:
            self.tc.dong("CAM")

    def feed_measurement_camera(self, t, img, mask=None, staged_slot=None):
        """UpdaterCamera::feed_measurement + try_update (REF: UpdaterCamera.cpp:77-195).  staged_slot: the image already sits in that
        HBM slot of the context (Context.image_stage); `img` is then not read."""
        e, st = self.op.est, self.state
        if not e.cam.enabled:
            return
        if st.initialized:
            self.tc.ding("CAM")
        if len(self.cam_t_hist) > 100:
            self.cam_t_hist.pop(0)
        self.cam_t_hist.append(float(t))
        prep = self.camera_prepare(t, img, mask, staged_slot, _bookkeeping_done=True)
        if prep is not None:
            self.tc.ding("[Time-Cam] feed measurement + try_update")
            self.camera_run(prep, sync=False)
            self.tc.dong("[Time-Cam] feed measurement + try_update")
            self.camera_finish(prep)
            return
        self.tc.ding("[Time-Cam] feed measurement: points")      # labels of UpdaterCamera.cpp:79-190
        if staged_slot is not None and not e.cam.downsample:
            self.ctx.tracker_feed_staged(t, staged_slot, mask)
        elif e.cam.downsample:
            self.ctx.tracker_feed_downsampled(t, img, mask)
        else:
            self.ctx.tracker_feed(t, img, mask)
        self.tc.dong("[Time-Cam] feed measurement: points")
        self._lines_in_flight = False
        if self.use_lines:
            self.tc.ding("[Time-Cam] feed measurement: lines")
            vps = self.ctx.vanishing_points(st.cam_ext.Rot(), st.cam_intr.v)
            if st.initialized and hasattr(self.ctx, "line_tracker_feed_async"):
                # the line tracker's host logic runs on the library's worker thread while this thread enqueues the point update; both
                # read only what the point tracker produced for this frame (feed_measurement precedes try_update in the reference)
                self.ctx.line_tracker_feed_async(t, vps)
                self._lines_in_flight = True
            else:
                self.ctx.line_tracker_feed(t, vps)
                self.stats["lines_tracked"] += self.ctx.line_db_size()
            self.tc.dong("[Time-Cam] feed measurement: lines")
        self._marginalize_slam_features()
        self.stats["frames"] += 1
        if st.initialized:
            self._camera_try_update()
            self.tc.dong("CAM")
        self._join_lines()

    def _join_lines(self):
        if self._lines_in_flight:
            self.ctx.line_tracker_feed_wait()
            self.stats["lines_tracked"] += self.ctx.line_db_size()
            self._lines_in_flight = False

    def _marginalize_slam_features(self):   # UpdaterCamera.cpp:118-137 + StateHelper::marginalize_slam :203-213
        st = self.state
        if not st.slam:
            return
        ids = list(st.slam)
        flags = self.ctx.slam_marg_flags(ids, [st.slam[i].update_fail_count for i in ids])
        for fid, f in zip(ids, flags):
            if f:
                st.marginalize(st.slam.pop(fid).id, 3)
                self.stats["slam_marginalized"] += 1

    def _try_update_args(self):
        """the options of try_update for the current window, or None when there is no interpolation polynomial yet"""
        st, e = self.state, self.op.est
        ts = st.clone_times()
        if len(ts) < e.intr_order + 1 or len(self.cam_t_hist) < 2:      # have_polynomial, CamHelper.cpp:615-616
            return None
        c, fi = e.cam, e.cam.featinit
        full = st.clone_window() > e.window_size
        kw = dict(t_prev_frame=self.cam_t_hist[-2], state_time=st.time, window_full=full, chi2_mult=c.chi2_mult)
        if e.use_imu_res:      # State::get_interpolated_pose = get_interpolated_pose_imu (State.cpp:975-977)
            kw["cpi"] = st.cpi_table()
        cam_hz = (len(self.cam_t_hist) - 1) / (self.cam_t_hist[-1] - self.cam_t_hist[0])
        pk = dict(min_dist=fi.min_dist, max_dist=fi.max_dist, max_cond=fi.max_cond_number, max_baseline=fi.max_baseline, refine=fi.refine_features,
                  init_min_meas=min(int(e.window_size) * int(cam_hz) - 1, 10))      # CamHelper.cpp:686
        return pk, kw, min(c.max_msckf, self.ctx.cfg.max_features)

    def _camera_try_update(self):
        st, e = self.state, self.op.est
        if self.decisions is not None and getattr(self.decisions, "probe_state", False):
            self.decisions.states_pre.append((self.stats["frames"], self._state_probe(), self.ctx.cov_download(st.n) if hasattr(self.ctx, "cov_download") else None))
        args = self._try_update_args()
        if args is None:
            return
        pk, kw, max_msckf = args
        c = e.cam
        if self.one_call_update and c.max_slam == 0 and "cpi" not in kw and hasattr(self.ctx, "camera_try_update"):
            # the whole of try_update in one library call: point update, dx applied, line update on the updated state, dx applied
            label = "[Time-Cam] get features + MSCKF update + LINE update" if self.use_lines else "[Time-Cam] get features + MSCKF update"
            self.tc.ding(label)
            sv = st.view()
            out, lo, n_db = self.ctx.camera_try_update(sv, st._window_arrays()["plus"], st.n, max_msckf, self.max_obs, lines=self.use_lines, **pk, **kw)
            self.tc.dong(label)
            if self._lines_in_flight:       # (joined inside the call)
                self.stats["lines_tracked"] += n_db
                self._lines_in_flight = False
            self._count_points(out)
            if lo is not None:
                self._count_lines(lo)
            return
        self.tc.ding("[Time-Cam] get features + MSCKF update")
        out = self.ctx.camera_update_points(st.view(), st.n, max_msckf, self.max_obs, max_slam=c.max_slam, slam_ids=list(st.slam), **pk, **kw)
        if self.use_lines and hasattr(self.ctx, "camera_get_line_features"):
            # LineHelper::get_line_features runs before msckf_update's correction reaches the state (UpdaterCamera.cpp:148-152): the line pool
            # is formed and triangulated on the state as it is now; camera_update_lines below linearises on the updated one
            self._join_lines()
            self.ctx.camera_get_line_features(st.view(), st.n, self.max_obs, **kw)
        if out["status"] == 0 and out["n_accepted"] > 0:
            st.apply(out["dx"])
        self.tc.dong("[Time-Cam] get features + MSCKF update")
        self._count_points(out)
        if c.max_slam > 0:
            self._slam_update_and_init(out)
        if self.use_lines:
            self.tc.ding("[Time-Cam] LINE update")
            self._join_lines()
            lo = self.ctx.camera_update_lines(st.view(), st.n, self.max_obs, **kw)
            if lo["status"] == 0 and lo["n_accepted"] > 0:
                st.apply(lo["dx"])
            self._count_lines(lo)
            self.tc.dong("[Time-Cam] LINE update")

    def _cov_probe(self, label):
        d = self.decisions
        if d is not None and getattr(d, "probe_cov", None) is not None and d.probe_cov[0] <= self.stats["frames"] <= d.probe_cov[1] and hasattr(self.ctx, "cov_download"):
            d.cov_probes.append((label, self.stats["frames"], self.state.time, self.ctx.cov_download(self.state.n)))

    def _state_probe(self):
        st = self.state
        x = [np.asarray(st.imu.q, float), np.asarray(st.imu.p, float), np.asarray(st.imu.v, float), np.asarray(st.imu.bg, float), np.asarray(st.imu.ba, float)]
        if st.cam_intr is not None:
            x.append(np.asarray(st.cam_intr.v, float))
        for t in sorted(st.clones):
            x.append(np.asarray(st.clones[t].q, float)), x.append(np.asarray(st.clones[t].p, float))
        return np.concatenate(x)

    def _count_points(self, out):
        if self.decisions is not None and getattr(self.decisions, "probe_state", False):
            self.decisions.states.append(self._state_probe())
        if self.decisions is not None:
            vals = self.ctx.last_point_decisions() if hasattr(self.ctx, "last_point_decisions") else None
            self.decisions.append(("points", self.stats["frames"], self.state.time, int(out["n_pool"]), np.array(out["ids"], dtype=np.uint64),
                                   np.array(out["accepted"], dtype=np.uint8), int(out["status"]), vals, np.array(out["dx"], dtype=float)))
        if out["status"] != 0:
            self.stats["not_psd"] += 1
        elif out["n_accepted"] > 0:
            self.stats["cam_updates"] += 1
        self._cov_probe("after the point update")
        self.stats["cam_features"] += out["n_msckf"]
        self.stats["cam_accepted"] += out["n_accepted"] if out["status"] == 0 else 0

    def _count_lines(self, lo):
        if self.decisions is not None:
            vals = self.ctx.last_line_decisions() if (hasattr(self.ctx, "last_line_decisions") and lo["n_lines"] > 0) else None
            self.decisions.append(("lines", self.stats["frames"], self.state.time, int(lo["n_pool"]), np.array(lo["ids"], dtype=np.uint64),
                                   np.array(lo["accepted"], dtype=np.uint8), int(lo["status"]), vals, np.array(lo["dx"], dtype=float),
                                   np.array(lo["line_FinG"], dtype=float) if "line_FinG" in lo else None))
        self._cov_probe("after the line update")
        self.stats["line_pool"] += lo["n_pool"]
        self.stats["lines_triangulated"] += lo["n_lines"]
        if lo["status"] != 0:
            self.stats["not_psd"] += 1
        elif lo["n_accepted"] > 0:
            self.stats["line_updates"] += 1
            self.stats["lines_accepted"] += lo["n_accepted"]

    def _landmark_system(self, t, uv, p, p_fej):
        """get_feature_jacobian_full for one feature: (rows, Hf [rows][3], Hx [rows][k], res, cols)."""
        st = self.state.view()
        tr = Tracks([0, len(t)], t, uv, [p], p_FinG_fej=[p_fej])
        cols = self.ctx.jacobian_columns(st, tr)
        rows, Hf, Hx, res = self.ctx.build_jacobians(st, tr, cols, 2 * self.max_obs)
        r = int(rows[0])
        return r, Hf[0].T[:r].copy(), Hx[0].T[:r].copy(), res[0][:r].copy(), cols

    def _slam_update_and_init(self, out):   # UpdaterCamera::slam_update :296-338, slam_init :340-369
        st, c = self.state, self.op.est.cam
        if out["n_slam"]:
            lst = self.ctx.camera_update_list(0)
            for j, fid in enumerate(lst["ids"]):
                a, b = lst["obs_ptr"][j], lst["obs_ptr"][j + 1]
                lm = st.slam.get(int(fid))
                if lm is None or b - a < 1 or b - a > self.max_obs:
                    continue
                r, Hf, Hx, res, cols = self._landmark_system(lst["obs_time"][a:b], lst["obs_uv"][a:b], lm.p, lm.p_fej)
                if r < 2:
                    continue
                H = np.hstack([Hx, Hf])
                cols_f = np.concatenate([cols, [lm.id, lm.id + 1, lm.id + 2]]).astype(np.int32)
                rc, acc, dx = self.ctx.slam_update(st.n, H, res, cols_f, c.chi2_mult)
                if rc != 0:
                    self.stats["not_psd"] += 1
                elif acc:
                    st.apply(dx)
                    self.stats["slam_updates"] += 1
        if out["n_init"]:
            lst = self.ctx.camera_update_list(1)
            for j, fid in enumerate(lst["ids"]):
                a, b = lst["obs_ptr"][j], lst["obs_ptr"][j + 1]
                t, uv, p = lst["obs_time"][a:b], lst["obs_uv"][a:b], lst["p_FinG"][j]
                if int(fid) in st.slam:
                    # A landmark of the state stays in the tracker's database (FeatureDatabase::get_feature does not remove it,
                    # CamHelper.cpp:621-628) and can come back through the pool as an initialisation candidate.  The reference
                    # initialises a second copy whose map insert then fails (UpdaterCamera.cpp:361): three orphaned columns that
                    # are never marginalised.  Here the candidate is consumed without growing the state.
                    continue
                ok = 0
                if 2 <= b - a <= self.max_obs:
                    r, Hf, Hx, res, cols = self._landmark_system(t, uv, p, p)
                    if r >= 4:
                        ok, dxi, dx = self.ctx.slam_initialize(st.n, Hf, Hx, res, cols, c.chi2_mult)
                if ok:   # StateHelper::initialize :357-439: the landmark joins the state at the end, then the EKF correction
                    st.slam[int(fid)] = Landmark(fid, p + dxi, st.n)
                    st.slam[int(fid)].p_fej = np.array(p, dtype=np.float64)
                    st.n += 3
                    st.apply(dx)
                    self.stats["slam_initialized"] += 1
                else:    # UpdaterCamera.cpp:363-364: back to the database
                    self.ctx.db_append_measurements(int(fid), t, uv, lst["obs_uvn"][a:b])

    # ---------------------------------------------------------------------------------------------- wheel
    def feed_measurement_wheel(self, t, m1, m2):
        """REF: SystemManager.cpp:129-137, UpdaterWheel.cpp:21-70."""
        if not self.op.est.wheel.enabled:
            return
        st = self.state
        if st.initialized:
            self.tc.ding("WHL")
        self.whl.append((t, m1, m2))
        if t - self.whl.t(0) > 1000:   # UpdaterWheel.cpp:24-30
            self.whl.drop_before(t - 1000)
        if st.initialized:
            self._wheel_try_update()
            self.tc.dong("WHL")

    def _wheel_try_update(self):
        st, wl = self.state, self.op.est.wheel
        ts = st.clone_times()
        if not ts:
            return
        if wl.reuse_of_information:   # UpdaterWheel.cpp:38-49
            if st.clone_window() > self.op.est.window_size:
                return
            older = [t for t in ts if t < self.whl.t(-1) + float(st.wheel_dt.v[0])]
            if older:
                self._wheel_update(ts[0], older[-1])
            return
        if self.whl_last_updated not in ts:   # :52-60
            newer = [t for t in ts if t > self.whl_last_updated]
            if not newer:
                self.whl_last_updated = ts[-1]
                return
            self.whl_last_updated = newer[0]
        for t in ts:
            if t <= self.whl_last_updated:
                continue
            if not self._wheel_update(self.whl_last_updated, t):
                break

    def _wheel_update(self, time0, time1):   # UpdaterWheel.cpp:72-139
        st = self.state
        toff = float(st.wheel_dt.v[0])
        w = self.whl.view()
        if w[0, 0] > time0 - toff:     # the reference's own test on the whole stack (UpdaterWheel.cpp:150)
            return False
        k = max(0, int(np.searchsorted(w[:, 0], time0 - toff, side="right")) - 2)   # only the tail that can matter
        ok, t, m1, m2 = select_wheel_data(np.ascontiguousarray(w[k:, 0]), np.ascontiguousarray(w[k:, 1]), np.ascontiguousarray(w[k:, 2]),
                                          time0 - toff, time1 - toff)
        if not ok:
            return False
        c0, c1 = st.clone_at(time0), st.clone_at(time1)
        z = np.zeros(3)
        r0, r1 = st.cpis.get(time0), st.cpis.get(time1)
        ws = PlvWheelState.make(st.wheel_intr.v, st.wheel_ext.Rot(), st.wheel_ext.p, c0.Rot(), c0.p, c1.Rot(), c1.p, c0.id, c1.id,
                                R0_fej=c0.Rot_fej(), p0_fej=c0.p_fej, R1_fej=c1.Rot_fej(), p1_fej=c1.p_fej,
                                w0=r0["w"] if r0 else z, v0=r0["v"] if r0 else z, w1=r1["w"] if r1 else z, v1=r1["v"] if r1 else z,
                                ext_id=st.wheel_ext.id, dt_id=st.wheel_dt.id, intr_id=st.wheel_intr.id)
        rc, acc, dx = self.ctx.wheel_update(self.wheel_opt, ws, t, m1, m2, st.n)
        if self.decisions is not None:
            self.decisions.append(("wheel", self.stats["frames"], st.time, 1, np.zeros(1, dtype=np.uint64), np.array([1 if (rc == 0 and acc) else 0], dtype=np.uint8),
                                   int(rc), None, np.array(dx, dtype=float)))
        self._cov_probe("after a wheel update")
        self.stats["wheel_updates"] += 1
        if rc != 0:
            self.stats["not_psd"] += 1
        elif acc:
            st.apply(dx)
            self.stats["wheel_accepted"] += 1
        self.whl_last_updated = time1
        return True

    # ---------------------------------------------------------------------------------------------- logging
    def imu_pose_covariance(self):
        """StateHelper::get_marginal_covariance(state, {imu->pose()}) as State_Logger logs it (6 x 6)."""
        return self.ctx.cov_download(self.state.n)[:6, :6]
