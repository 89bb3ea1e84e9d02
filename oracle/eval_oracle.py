"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference's ov_eval/example lost its
ground-truth file, so there is no golden trajectory to pin against; the restatement is checked against hand-computed
rigid transforms instead (tests/test_oracle_eval.py).

numpy restatement of the reference's trajectory evaluation:
  ov_eval::Loader::load_data                      REF: open_vins/ov_eval/src/utils/Loader.cpp:26-90
  AlignUtils::perform_association / align_umeyama REF: open_vins/ov_eval/src/alignment/AlignUtils.cpp:26-91,101-189
  AlignTrajectory::align_*                        REF: open_vins/ov_eval/src/alignment/AlignTrajectory.cpp:26-166
  ResultTrajectory ctor / calculate_ate           REF: open_vins/ov_eval/src/calc/ResultTrajectory.cpp:26-121
  Statistics::calculate                           REF: open_vins/ov_eval/src/utils/Statistics.h:72-119
  State_Logger::save_trajectory_to_file           REF: PL-VIWO/src/utils/State_Logger.h:188-205
  quat_2_Rot / rot_2_quat / quat_multiply / log_so3  REF: open_vins/ov_core/src/utils/quat_ops.h:88-195,273-313
"""
import math

import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def quat_2_rot(q):
    q = np.asarray(q, dtype=np.float64)
    return (2 * q[3] ** 2 - 1) * np.eye(3) - 2 * q[3] * skew(q[:3]) + 2 * np.outer(q[:3], q[:3])


def rot_2_quat(R):
    T = np.trace(R)
    q = np.zeros(4)
    if R[0, 0] >= T and R[0, 0] >= R[1, 1] and R[0, 0] >= R[2, 2]:
        q[0] = math.sqrt((1 + 2 * R[0, 0] - T) / 4)
        q[1] = (1 / (4 * q[0])) * (R[0, 1] + R[1, 0])
        q[2] = (1 / (4 * q[0])) * (R[0, 2] + R[2, 0])
        q[3] = (1 / (4 * q[0])) * (R[1, 2] - R[2, 1])
    elif R[1, 1] >= T and R[1, 1] >= R[0, 0] and R[1, 1] >= R[2, 2]:
        q[1] = math.sqrt((1 + 2 * R[1, 1] - T) / 4)
        q[0] = (1 / (4 * q[1])) * (R[0, 1] + R[1, 0])
        q[2] = (1 / (4 * q[1])) * (R[1, 2] + R[2, 1])
        q[3] = (1 / (4 * q[1])) * (R[2, 0] - R[0, 2])
    elif R[2, 2] >= T and R[2, 2] >= R[0, 0] and R[2, 2] >= R[1, 1]:
        q[2] = math.sqrt((1 + 2 * R[2, 2] - T) / 4)
        q[0] = (1 / (4 * q[2])) * (R[0, 2] + R[2, 0])
        q[1] = (1 / (4 * q[2])) * (R[1, 2] + R[2, 1])
        q[3] = (1 / (4 * q[2])) * (R[0, 1] - R[1, 0])
    else:
        q[3] = math.sqrt((1 + T) / 4)
        q[0] = (1 / (4 * q[3])) * (R[1, 2] - R[2, 1])
        q[1] = (1 / (4 * q[3])) * (R[2, 0] - R[0, 2])
        q[2] = (1 / (4 * q[3])) * (R[0, 1] - R[1, 0])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_multiply(q, p):
    Qm = np.zeros((4, 4))
    Qm[:3, :3] = q[3] * np.eye(3) - skew(q[:3])
    Qm[:3, 3] = q[:3]
    Qm[3, :3] = -q[:3]
    Qm[3, 3] = q[3]
    r = Qm @ p
    if r[3] < 0:
        r = -r
    return r / np.linalg.norm(r)


def quat_inv(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def log_so3(R):
    tr = np.trace(R)
    if tr + 1.0 < 1e-10:
        if abs(R[2, 2] + 1.0) > 1e-5:
            return (math.pi / math.sqrt(2.0 + 2.0 * R[2, 2])) * np.array([R[0, 2], R[1, 2], 1.0 + R[2, 2]])
        if abs(R[1, 1] + 1.0) > 1e-5:
            return (math.pi / math.sqrt(2.0 + 2.0 * R[1, 1])) * np.array([R[0, 1], 1.0 + R[1, 1], R[2, 1]])
        return (math.pi / math.sqrt(2.0 + 2.0 * R[0, 0])) * np.array([1.0 + R[0, 0], R[1, 0], R[2, 0]])
    tr_3 = tr - 3.0
    if tr_3 < -1e-7:
        theta = math.acos((tr - 1.0) / 2.0)
        mag = theta / (2.0 * math.sin(theta))
    else:
        mag = 0.5 - tr_3 / 12.0
    return mag * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def rot_z(t):
    return np.array([[math.cos(t), -math.sin(t), 0.0], [math.sin(t), math.cos(t), 0.0], [0.0, 0.0, 1.0]])


def load_data(path):
    times, poses, cov_ori, cov_pos = [], [], [], []
    with open(path) as f:
        for line in f.read().split("\n"):
            if line.startswith("#"):
                continue
            data = [float(_atof(x)) for x in line.split(" ") if x != ""][:20]
            if len(data) >= 20:
                times.append(data[0])
                poses.append(data[1:8])
                d = data
                cov_ori.append([[d[8], d[9], d[10]], [d[9], d[11], d[12]], [d[10], d[12], d[13]]])
                cov_pos.append([[d[14], d[15], d[16]], [d[15], d[17], d[18]], [d[16], d[18], d[19]]])
            elif len(data) >= 8:
                times.append(data[0])
                poses.append(data[1:8])
    return np.array(times), np.array(poses).reshape(-1, 7), np.array(cov_ori).reshape(-1, 3, 3), np.array(cov_pos).reshape(-1, 3, 3)


def _atof(s):
    """std::atof: longest valid prefix, 0.0 when there is none."""
    import re
    m = re.match(r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf|nan)", s, re.I)
    return float(m.group(0)) if m else 0.0


def total_length(poses):
    return float(sum(np.linalg.norm(poses[i, :3] - poses[i - 1, :3]) for i in range(1, len(poses))))


def perform_association(offset, max_difference, est_times, gt_times):
    est_idx, gt_idx = [], []
    gp = 0
    for i in range(len(est_times)):
        best, best_gt = max_difference, -1
        te = est_times[i] + offset
        while gp < len(gt_times) and gt_times[gp] < te and abs(gt_times[gp] - te) > max_difference:
            gp += 1
        while gp < len(gt_times) and abs(gt_times[gp] - te) <= max_difference:
            if abs(gt_times[gp] - te) >= best:
                break
            best = abs(gt_times[gp] - te)
            best_gt = gp
            gp += 1
        if best_gt != -1:
            est_idx.append(i)
            gt_idx.append(best_gt)
    return np.array(est_idx, dtype=np.int32), np.array(gt_idx, dtype=np.int32)


def align_umeyama(data, model, known_scale, yaw_only):
    mu_M, mu_D = model.mean(axis=0), data.mean(axis=0)
    mz, dz = model - mu_M, data - mu_D
    n = float(len(model))
    Cm = (mz.T @ dz) / n
    sigma2 = float((dz * dz).sum()) / n
    U, D, Vt = np.linalg.svd(Cm)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt.T) < 0:
        S[2, 2] = -1
    if yaw_only:
        rot_C = n * Cm.T
        R = rot_z(math.atan2(rot_C[0, 1] - rot_C[1, 0], rot_C[0, 0] + rot_C[1, 1]))
    else:
        R = U @ S @ Vt
    s = 1.0 if known_scale else 1.0 / sigma2 * float(np.trace(np.diag(D) @ S))
    t = mu_M - s * R @ mu_D
    return R, t, s


def align_trajectory(est, gt, method, n_aligned=-1):
    def single(yaw):
        g_rot, e_rot = quat_2_rot(gt[0, 3:]).T, quat_2_rot(est[0, 3:]).T
        if yaw:
            CR = e_rot @ g_rot.T
            R = rot_z(math.atan2(CR[0, 1] - CR[1, 0], CR[0, 0] + CR[1, 1]))
        else:
            R = g_rot @ e_rot.T
        return R, gt[0, :3] - R @ est[0, :3], 1.0

    if method == "none":
        return np.eye(3), np.zeros(3), 1.0
    if method == "posyawsingle" or (method == "posyaw" and n_aligned == 1):
        return single(True)
    if method == "se3single" or (method == "se3" and n_aligned == 1):
        return single(False)
    if method == "posyaw":
        return align_umeyama(est[:, :3], gt[:, :3], True, True)
    if method == "se3":
        return align_umeyama(est[:, :3], gt[:, :3], True, False)
    if method == "sim3":
        return align_umeyama(est[:, :3], gt[:, :3], False, False)
    raise ValueError(method)


def statistics(values):
    v = np.sort(np.asarray(values, dtype=np.float64))
    n = len(v)
    if n == 0:
        return dict(min=0, max=0, median=0, mean=0, rmse=0, std=0, ninetynine=0)
    median = v[0] if n == 1 else (v[n // 2] if n % 2 == 1 else 0.5 * (v[n // 2 - 1] + v[n // 2]))
    mean = float(v.sum() / n) if False else float(sum(v) / n)
    rmse = math.sqrt(float(sum(x * x for x in v)) / n)
    with np.errstate(invalid="ignore", divide="ignore"):
        std = float(np.sqrt(np.float64(sum((x - mean) ** 2 for x in v)) / np.float64(n - 1)))
    return dict(min=float(v[0]), max=float(v[-1]), median=float(median), mean=mean, rmse=rmse, std=std, ninetynine=mean + 2.326 * std)


def calculate_ate(est, gt, method, n_aligned=-1):
    R, t, s = align_trajectory(est, gt, method, n_aligned)
    q_inv = quat_inv(rot_2_quat(R))
    aligned = np.zeros_like(est)
    ori, pos = np.zeros(len(est)), np.zeros(len(est))
    for i in range(len(est)):
        aligned[i, :3] = s * R @ est[i, :3] + t
        aligned[i, 3:] = quat_multiply(est[i, 3:], q_inv)
        eR = quat_2_rot(aligned[i, 3:]).T @ quat_2_rot(gt[i, 3:])
        ori[i] = 180.0 / math.pi * np.linalg.norm(log_so3(eR))
        pos[i] = np.linalg.norm(gt[i, :3] - aligned[i, :3])
    return dict(R=R, t=t, s=s, aligned=aligned, ori_err=ori, pos_err=pos, ori=statistics(ori), pos=statistics(pos))


def format_pose(t, p, q, P=None):
    s = "%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f" % (t, p[0], p[1], p[2], q[0], q[1], q[2], q[3])
    if P is not None:
        P = np.asarray(P).reshape(6, 6)
        vals = [P[0, 0], P[0, 1], P[0, 2], P[1, 1], P[1, 2], P[2, 2], P[3, 3], P[3, 4], P[3, 5], P[4, 4], P[4, 5], P[5, 5]]
        s += " " + " ".join("%.10f" % v for v in vals)
    return s + "\n"


HEADER = "# timestamp(s) tx ty tz qx qy qz qw Pr11 Pr12 Pr13 Pr22 Pr23 Pr33 Pt11 Pt12 Pt13 Pt22 Pt23 Pt33\n"
