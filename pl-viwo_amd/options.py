"""YAML options of the estimator: the reference's keys, defaults and derived values, read from the same file layout (a master
`config.yaml` naming one file per section).  SURVEY §8(f) rank 4.

REF: PL-VIWO/src/options/Options.cpp:16-31, OptionsSystem.cpp:8-25, OptionsEstimator.cpp:10-70 (+ :110-139 set_values),
     OptionsCamera.cpp:10-170, OptionsWheel.cpp:8-75, OptionsInit.cpp:8-20, OptionsIMU.cpp:8-15, and the defaults of the
     matching *.h; open_vins/ov_core/src/utils/opencv_yaml_parse.h (parse_external: key -> file named in the master config ->
     section -> entry; a missing optional entry keeps the default, a missing required one is an error here as it is a
     warning + default there -- see `strict`).
GPS / LiDAR / simulation sections are outside SURVEY §8 and are not read.
"""
import os
import re
from types import SimpleNamespace

import numpy as np
import yaml

WHEEL_TYPES = ("Wheel2DAng", "Wheel2DLin", "Wheel2DCen", "Wheel3DAng", "Wheel3DLin", "Wheel3DCen")
HISTOGRAM = {"NONE": 0, "HISTOGRAM": 1, "CLAHE": 2}   # ov_core::TrackBase::HistogramMethod / PLV_HIST_*
FEAT_REP = {"GLOBAL_3D": 0, "GLOBAL_FULL_INVERSE_DEPTH": 1}


class OptionsError(ValueError):
    pass


def _rot_2_quat(R):
    """ov_core::rot_2_quat (REF: open_vins/ov_core/src/utils/quat_ops.h:88-130)."""
    T = np.trace(R)
    q = np.zeros(4)
    if R[0, 0] >= T and R[0, 0] >= R[1, 1] and R[0, 0] >= R[2, 2]:
        q[0] = np.sqrt((1 + 2 * R[0, 0] - T) / 4)
        q[1], q[2], q[3] = (R[0, 1] + R[1, 0]) / (4 * q[0]), (R[0, 2] + R[2, 0]) / (4 * q[0]), (R[1, 2] - R[2, 1]) / (4 * q[0])
    elif R[1, 1] >= T and R[1, 1] >= R[0, 0] and R[1, 1] >= R[2, 2]:
        q[1] = np.sqrt((1 + 2 * R[1, 1] - T) / 4)
        q[0], q[2], q[3] = (R[0, 1] + R[1, 0]) / (4 * q[1]), (R[1, 2] + R[2, 1]) / (4 * q[1]), (R[2, 0] - R[0, 2]) / (4 * q[1])
    elif R[2, 2] >= T and R[2, 2] >= R[0, 0] and R[2, 2] >= R[1, 1]:
        q[2] = np.sqrt((1 + 2 * R[2, 2] - T) / 4)
        q[0], q[1], q[3] = (R[0, 2] + R[2, 0]) / (4 * q[2]), (R[1, 2] + R[2, 1]) / (4 * q[2]), (R[0, 1] - R[1, 0]) / (4 * q[2])
    else:
        q[3] = np.sqrt((1 + T) / 4)
        q[0], q[1], q[2] = (R[1, 2] - R[2, 1]) / (4 * q[3]), (R[2, 0] - R[0, 2]) / (4 * q[3]), (R[0, 1] - R[1, 0]) / (4 * q[3])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def pose_from_T(T):
    """T_imu_sensor (sensor -> IMU, 4 x 4) -> the 7-vector the state keeps: q_ItoS (JPL), p_IinS
    (REF: OptionsCamera.cpp:141-146, OptionsWheel.cpp:48-54)."""
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    R_StoI = T[:3, :3]
    return np.concatenate([_rot_2_quat(R_StoI.T), -R_StoI.T @ T[:3, 3]])


class YamlParser:
    """The slice of ov_core::YamlParser the options use: a master file whose entries name the per-section files."""

    def __init__(self, config_path, strict=True):
        self.config_path = os.path.abspath(config_path)
        self.folder = os.path.dirname(self.config_path) + os.sep
        self.master = self._read(self.config_path)
        self.strict = strict
        self._files = {}

    @staticmethod
    def _read(path):
        with open(path) as f:
            text = f.read()
        # OpenCV FileStorage headers (`%YAML:1.0`) are not YAML directives
        text = re.sub(r"^%YAML[:\s][^\n]*\n", "", text)
        data = yaml.safe_load(text)
        return data if data is not None else {}

    def has_file(self, f):
        """boost::filesystem::exists(config_folder + f + ".yaml") of the reference's enable checks."""
        return os.path.exists(self.folder + f + ".yaml")

    def external(self, f):
        if f not in self._files:
            rel = self.master.get(f)
            if rel is None:
                raise OptionsError(f"{self.config_path}: no entry '{f}' naming the external file")
            path = os.path.join(self.folder, rel)
            if not os.path.exists(path):
                raise OptionsError(f"{path}: external configuration file of '{f}' not found")
            self._files[f] = self._read(path)
        return self._files[f]

    def get(self, f, section, key, default, required=True):
        node = self.external(f).get(section)
        if node is None or key not in node:
            if required and self.strict:
                raise OptionsError(f"{f}.yaml: required entry {section}.{key} is missing")
            return default
        val = node[key]
        if isinstance(default, bool):
            if isinstance(val, str):
                return val.lower() == "true"
            return bool(val)
        if isinstance(default, int) and not isinstance(default, bool):
            return int(val)
        if isinstance(default, float):
            return float(val)
        return val


def _load_system(p):
    f = "config_system"
    s = SimpleNamespace(bag_start=0.0, bag_durr=-1.0, path_bag="", path_gt="", save_timing=False, save_state=False,
                        save_trajectory=False, path_state="", path_timing="", path_trajectory="", exp_id=0, verbosity=2, save_prints=False)
    for key, req in (("verbosity", True), ("save_timing", True), ("path_timing", True), ("save_state", True), ("path_state", True),
                     ("save_trajectory", True), ("path_trajectory", True), ("save_prints", True), ("exp_id", True), ("path_bag", True),
                     ("bag_start", True), ("bag_durr", True), ("path_gt", False)):
        setattr(s, key, p.get(f, "sys", key, getattr(s, key), req))
    return s


def _load_imu(p):
    f = "config_imu"
    return SimpleNamespace(sigma_w=p.get(f, "imu", "gyro_noise", 1.6968e-04), sigma_wb=p.get(f, "imu", "gyro_bias", 1.9393e-05),
                           sigma_a=p.get(f, "imu", "accel_noise", 2.0000e-03), sigma_ab=p.get(f, "imu", "accel_bias", 3.0000e-03),
                           topic=p.get(f, "imu", "topic", ""))


def _load_init(p):
    f = "config_init"
    g = lambda k, d, r=True: p.get(f, "init", k, d, r)
    return SimpleNamespace(window_time=g("window_time", 1.0), imu_thresh=g("imu_thresh", 1.0), imu_wheel_thresh=g("imu_wheel_thresh", 0.1, False),
                           imu_only_init=g("imu_only_init", False), imu_gravity_aligned=g("imu_gravity_aligned", False, False),
                           use_gt=g("use_gt", False), use_gt_gnss=g("use_gt_gnss", False, False), use_gt_lidar=g("use_gt_lidar", False, False),
                           cov_size=g("cov_size", 1e-4), path_gt="")


def _load_camera(p):
    f = "config_camera"
    c = SimpleNamespace(enabled=False, max_n=2, time_analysis=False, topic=[], dt={}, intrinsics={}, distortion_model={}, wh={}, extrinsics={},
                        init_cov_dt=1e-4, init_cov_ex_o=1e-4, init_cov_ex_p=1e-3, init_cov_in_k=1.0, init_cov_in_c=1.0, init_cov_in_r=1e-5,
                        do_calib_ext=False, do_calib_int=False, do_calib_dt=False, use_mask={}, mask_path={}, downsample=False, n_pts=150, fast=20,
                        grid_x=5, grid_y=5, min_px_dist=10, histogram=HISTOGRAM["HISTOGRAM"], knn=0.85, max_slam=25, max_msckf=1000,
                        use_stereo=True, use_lines=True, feat_rep=FEAT_REP["GLOBAL_3D"], chi2_mult=1.0, sigma_pix=1.0, stereo_pairs={},
                        featinit=SimpleNamespace(triangulate_1d=False, refine_features=True, max_runs=5, init_lamda=1e-3, max_lamda=1e10,
                                                 min_dx=1e-6, min_dcost=1e-6, lam_mult=10.0, min_dist=0.10, max_dist=60.0, max_baseline=40.0,
                                                 max_cond_number=10000.0))
    if not p.has_file(f):   # OptionsCamera.cpp:13-16
        return c
    g = lambda k, d, r=True: p.get(f, "cam", k, d, r)
    c.enabled = g("enabled", c.enabled)
    c.time_analysis = g("time_analysis", c.time_analysis, False)
    for key in ("use_stereo", "max_n", "do_calib_ext", "do_calib_int", "do_calib_dt", "downsample", "n_pts", "fast", "grid_x", "grid_y",
                "min_px_dist", "init_cov_dt", "init_cov_ex_o", "init_cov_ex_p", "init_cov_in_k", "init_cov_in_c", "init_cov_in_r", "chi2_mult",
                "knn", "max_slam", "max_msckf"):
        setattr(c, key, g(key, getattr(c, key)))
    c.sigma_pix = g("sigma_px", c.sigma_pix)
    for key in ("triangulate_1d", "refine_features", "max_runs", "init_lamda", "max_lamda", "min_dx", "min_dcost", "lam_mult", "min_dist",
                "max_dist", "max_baseline", "max_cond_number"):
        setattr(c.featinit, key, g("fi_" + key, getattr(c.featinit, key), False))
    rep = g("feat_rep", "GLOBAL_3D")
    if rep not in FEAT_REP:   # OptionsCamera.cpp:55-59
        raise OptionsError(f"unsupported feature representation: {rep}")
    c.feat_rep = FEAT_REP[rep]
    hist = g("histogram_method", "HISTOGRAM")
    if hist not in HISTOGRAM:   # :60-71
        raise OptionsError(f"OptionsCamera: invalid feature histogram specified: {hist}. Available: NONE, HISTOGRAM, CLAHE")
    c.histogram = HISTOGRAM[hist]
    div = 2.0 if c.downsample else 1.0
    for i in range(c.max_n):   # load_i :111-170
        sec = f"cam{i}"
        gi = lambda k, d, r=True: p.get(f, sec, k, d, r)
        c.dt[i] = float(gi("timeoffset", 0.0))
        k4 = [float(x) for x in gi("intrinsics", [1.0, 1.0, 0.0, 0.0])]
        d4 = [float(x) for x in gi("distortion_coeffs", [0.0, 0.0, 0.0, 0.0])]
        intr = np.array(k4 + d4)
        intr[:4] /= div
        c.intrinsics[i] = intr
        c.distortion_model[i] = gi("distortion_model", "radtan")
        wh = [int(x) for x in gi("resolution", [1, 1])]
        c.wh[i] = [int(wh[0] / div), int(wh[1] / div)]
        c.extrinsics[i] = pose_from_T(gi("T_imu_cam", np.eye(4).tolist()))
        c.use_mask[i] = bool(gi("use_mask", False, False))
        c.topic.append(gi("topic", ""))
        if c.use_mask[i]:
            mp = os.path.join(p.folder, gi("mask", ""))
            if not os.path.exists(mp):
                raise OptionsError(f"invalid mask path: mask{i} - {mp}")
            c.mask_path[i] = mp
    if c.use_stereo:   # :79-104
        pair = [int(x) for x in g("stereo_pair", [])]
        if len(pair) % 2 != 0:
            raise OptionsError("Stero pair should be provided even number.")
        for a, b in zip(pair[0::2], pair[1::2]):
            if a < c.max_n and b < c.max_n:
                if a in c.stereo_pairs or b in c.stereo_pairs:
                    raise OptionsError("A camera is paired with more than one camera.")
                c.stereo_pairs[a], c.stereo_pairs[b] = b, a
        if not pair:
            c.use_stereo = False
    for a, b in c.stereo_pairs.items():   # one time offset per pair :107-108
        if b > a:
            c.dt[b] = c.dt[a]
        else:
            c.dt[a] = c.dt[b]
    return c


def _load_wheel(p):
    f = "config_wheel"
    w = SimpleNamespace(enabled=True, topic="", sub_topics=["front_left_wheel_joint", "front_right_wheel_joint", "rear_left_wheel_joint",
                                                            "rear_right_wheel_joint"], type="", noise_w=0.005, noise_v=0.005, noise_p=0.01,
                        init_cov_dt=1e-4, init_cov_ex_o=1e-4, init_cov_ex_p=1e-3, init_cov_in_b=1e-4, init_cov_in_r=1e-4, dt=0.0, chi2_mult=1.0,
                        extrinsics=pose_from_T(np.eye(4)), intrinsics=np.array([1.0, 1.0, 2.0]), do_calib_dt=True, do_calib_ext=True,
                        do_calib_int=True, reuse_of_information=False)
    if not p.has_file(f):   # OptionsWheel.cpp:11-14
        w.enabled = False
        return w
    g = lambda k, d, r=True: p.get(f, "wheel", k, d, r)
    for key in ("enabled", "chi2_mult", "noise_w", "noise_v", "noise_p", "do_calib_ext", "do_calib_dt", "do_calib_int", "init_cov_dt",
                "init_cov_ex_o", "init_cov_ex_p", "init_cov_in_b", "init_cov_in_r", "reuse_of_information", "topic"):
        setattr(w, key, g(key, getattr(w, key)))
    subs = g("sub_topics", "", False)
    if subs:
        w.sub_topics = subs.split(", ")
    w.dt = float(g("timeoffset", w.dt))
    w.extrinsics = pose_from_T(g("T_imu_wheel", np.eye(4).tolist()))
    w.intrinsics = np.array([float(x) for x in g("intrinsics", [1.0, 1.0, 2.0])])
    w.type = g("type", w.type)
    if w.type not in WHEEL_TYPES:   # :64-68
        raise OptionsError(f"{w.type} is not a supported type of wheel. Available: " + ", ".join(WHEEL_TYPES))
    if w.type not in ("Wheel2DAng", "Wheel3DAng"):   # :71-73
        w.do_calib_int = False
    return w


class InterpolationError:
    """OptionsEstimator::interpolation_error (REF: OptionsEstimator.h:58-107): per clone rate and polynomial order, the slope of the
    pose-interpolation error against the estimated acceleration."""

    def __init__(self):
        self.threshold_ori, self.threshold_pos, self.mlt = 0.01, 0.001, 1.0
        self.ori_slope, self.pos_slope = {}, {}

    def set_values(self, hz, ori, pos):   # OptionsEstimator.cpp:110-139
        if ori is None or pos is None or len(ori) != 5 or len(pos) != 5 or min(ori) < 0 or min(pos) < 0:
            return
        self.ori_slope[hz] = {order: float(v) for order, v in zip((1, 3, 5, 7, 9), ori)}
        self.pos_slope[hz] = {order: float(v) for order, v in zip((1, 3, 5, 7, 9), pos)}

    def available_clone_hz(self):
        return sorted(self.ori_slope)

    def ori_std(self, hz, order, est_A):
        return self.mlt * est_A * self.ori_slope[hz][order]

    def pos_std(self, hz, order, est_a):
        return self.mlt * est_a * self.pos_slope[hz][order]

    def ori_cov(self, hz, order, est_A):
        return self.ori_std(hz, order, est_A) ** 2

    def pos_cov(self, hz, order, est_a):
        return self.pos_std(hz, order, est_a) ** 2


def _load_estimator(p):
    f = "config_estimator"
    g = lambda k, d, r=True: p.get(f, "est", k, d, r)
    e = SimpleNamespace(gravity=np.array([0.0, 0.0, float(g("gravity_mag", 0.0))]), use_imu_res=g("use_imu_res", False),
                        use_imu_cov=g("use_imu_cov", False), use_pol_cov=g("use_pol_cov", False), window_size=g("window_size", 0.5),
                        clone_freq=g("clone_freq", 10), dt_exp=g("dt_extrapolation", 0.01), intr_order=g("intr_order", 3),
                        dynamic_cloning=g("dynamic_cloning", True), intr_err=InterpolationError())
    e.intr_err.mlt = float(g("intr_error_mlt", 1.0))
    ext = p.external(f)
    for hz in range(1, 40):   # :22-26
        ori = (ext.get("intr_ori") or {}).get(f"Hz_{hz}")
        pos = (ext.get("intr_pos") or {}).get(f"Hz_{hz}")
        e.intr_err.set_values(hz, ori, pos)
    e.intr_err.threshold_ori = float(g("intr_error_ori_thr", 0.01))
    e.intr_err.threshold_pos = float(g("intr_error_pos_thr", 0.001))
    mlt = float(g("intr_error_thr_mlt", 1.0))
    e.intr_err.threshold_ori *= mlt
    e.intr_err.threshold_pos *= mlt
    if e.intr_order < 1 or e.intr_order % 2 == 0:   # :33-36
        raise OptionsError(f"Estimator polynomial order should be >= 1 and odd number. Current value: {e.intr_order}")
    max_clone_size = e.window_size * e.clone_freq + 1
    if max_clone_size < e.intr_order + 1:   # :39-44
        raise OptionsError(f"Max clone size is smaller than required for polynomial interpolation ({max_clone_size} < {e.intr_order} + 1).")
    if e.dynamic_cloning and max_clone_size < 6:   # :46-50
        raise OptionsError(f"Max clone size is smaller than required for polynomial interpolation ({max_clone_size} < 6).")
    if int(e.use_imu_cov) + int(e.use_pol_cov) > 1:   # :52-57
        raise OptionsError("More than 1 cov method enabled.")
    e.init, e.imu, e.cam, e.wheel = _load_init(p), _load_imu(p), _load_camera(p), _load_wheel(p)
    return e


def load_options(config_path, strict=True):
    """Options::load_print (REF: Options.cpp:16-31): `.sys` and `.est` (with .imu .cam .wheel .init)."""
    p = YamlParser(config_path, strict=strict)
    op = SimpleNamespace(sys=_load_system(p), est=_load_estimator(p), config_folder=p.folder)
    if op.est.init.use_gt:
        op.est.init.path_gt = op.sys.path_gt
    return op
