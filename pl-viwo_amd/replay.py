"""File-based replay in place of the reference's rosbag player: a directory of images plus IMU / wheel
CSV files is merged into one time-ordered message stream and fed to the SystemManager; poses are logged in the reference's
trajectory format.

REF: PL-VIWO/src/run_bag.cpp:51-144 (message loop: IMU -> feed_measurement_imu (+ trajectory line when a clone was made), camera ->
     feed_measurement_camera, wheel -> feed_measurement_wheel; bag_start / bag_durr window), :272-340 (mono camera message ->
     CameraData with an all-zero mask unless use_mask); PL-VIWO/src/core/ROSHelper.cpp:151-216 (image / JointState conversions);
     PL-VIWO/src/utils/State_Logger.h:188-205 (trajectory line).

Dataset layout (`sys.path_bag` names the directory):
    imu.csv              t, wx, wy, wz, ax, ay, az          ('#' comment lines; t in seconds, or integer nanoseconds as EuRoC)
    wheel.csv            t, m1, m2                          (the two readings of the configured wheel type; optional)
    cam0/data.csv        t, file name                       (optional: without it the file stem is the time stamp)
    cam0/data/*          8-bit grey images: .pgm (P5), .png (grey / RGB, not interlaced) or .npy
A directory in the EuRoC MAV (ASL) layout is read as it is: mav0/imu0/data.csv, mav0/cam0/data.csv + mav0/cam0/data/*.png; a KAIST
Complex Urban raw directory (sensor_data/xsens_imu.csv, encoder.csv, image/stereo_left) through kaist.KaistDataset (open_dataset).
"""
import os
import struct
import zlib

import numpy as np

from . import traj_format, traj_header
from .system import SystemManager

IMU, WHEEL, CAM = 0, 1, 2


# ------------------------------------------------------------------------------------------------------------ images
def _read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"P5":
        raise ValueError(f"{path}: not a binary PGM")
    vals, pos = [], 2
    while len(vals) < 3:   # width, height, maxval with '#' comments
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        vals.append(int(data[pos:end]))
        pos = end
    w, h, maxval = vals
    pos += 1
    if maxval > 255:
        img = np.frombuffer(data, dtype=">u2", count=w * h, offset=pos).reshape(h, w)
        return (img >> 8).astype(np.uint8)
    return np.frombuffer(data, dtype=np.uint8, count=w * h, offset=pos).reshape(h, w).copy()


def _read_png(path):
    """8-bit grey / grey+alpha / RGB / RGBA, non-interlaced (what cv_bridge's MONO8 conversion would be handed)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    ch = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or ch is None or interlace:
        raise ValueError(f"{path}: only 8-bit non-interlaced grey / RGB PNGs are read")
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:   # 1 (sub), 3 (average), 4 (Paeth) depend on the pixel to the left: byte-serial
            cur = np.zeros_like(line)
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pr) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    if ch <= 2:
        return img[:, :, 0].copy()
    rgb = img[:, :, :3].astype(np.float64)   # OpenCV's RGB -> grey weights
    return np.clip(np.rint(0.299 * rgb[:, :, 0] + 0.587 * rgb[:, :, 1] + 0.114 * rgb[:, :, 2]), 0, 255).astype(np.uint8)


def read_image(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".pgm":
        return _read_pgm(path)
    if ext == ".png":
        return _read_png(path)
    if ext == ".npy":
        return np.ascontiguousarray(np.load(path), dtype=np.uint8)
    raise ValueError(f"{path}: unsupported image format")


# ----------------------------------------------------------------------------------------------------------- dataset
def _read_csv(path, ncol):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            parts = [x for x in line.replace(",", " ").split()]
            if len(parts) < ncol:
                raise ValueError(f"{path}: expected {ncol} columns: {line}")
            rows.append(parts[:ncol])
    return rows


def _stamp(s):
    """seconds as a decimal number, or integer nanoseconds (EuRoC)"""
    if "." not in s and "e" not in s.lower() and len(s.lstrip("-")) > 12:
        return int(s) * 1e-9
    return float(s)


class Dataset:
    """The time-ordered message list of a dataset directory (the rosbag::View of run_bag.cpp)."""

    def __init__(self, root, cam_dir="cam0", use_wheel=True, use_cam=True):
        self.root = root
        imu_path = os.path.join(root, "imu.csv")
        if not os.path.exists(imu_path) and os.path.isdir(os.path.join(root, "mav0")):   # EuRoC MAV (ASL) layout
            imu_path = os.path.join(root, "mav0", "imu0", "data.csv")
            cam_dir = os.path.join("mav0", cam_dir)
        imu = _read_csv(imu_path, 7)
        self.imu = np.array([[_stamp(r[0])] + [float(x) for x in r[1:]] for r in imu])
        self.wheel = np.zeros((0, 3))
        wp = os.path.join(root, "wheel.csv")
        if use_wheel and os.path.exists(wp):
            self.wheel = np.array([[_stamp(r[0]), float(r[1]), float(r[2])] for r in _read_csv(wp, 3)])
        self.frames = []
        cdir = os.path.join(root, cam_dir)
        if use_cam and os.path.isdir(cdir):
            listing = os.path.join(cdir, "data.csv")
            if os.path.exists(listing):
                self.frames = [(_stamp(r[0]), os.path.join(cdir, "data", r[1])) for r in _read_csv(listing, 2)]
            else:
                for name in sorted(os.listdir(os.path.join(cdir, "data"))):
                    self.frames.append((_stamp(os.path.splitext(name)[0]), os.path.join(cdir, "data", name)))
        msgs = [(t, IMU, i) for i, t in enumerate(self.imu[:, 0])] + [(t, WHEEL, i) for i, t in enumerate(self.wheel[:, 0])] + \
               [(t, CAM, i) for i, (t, _) in enumerate(self.frames)]
        msgs.sort(key=lambda m: (m[0], m[1]))
        self.msgs = msgs

    def t_begin(self):
        return self.msgs[0][0]

    def image(self, i):
        return read_image(self.frames[i][1])


def open_dataset(root, use_wheel=True, use_cam=True):
    """Dataset for this layout / EuRoC, or kaist.KaistDataset for a KAIST Complex Urban raw directory (sensor_data/xsens_imu.csv)"""
    from . import kaist
    if kaist.is_kaist_raw(root):
        return kaist.KaistDataset(root, use_wheel=use_wheel, use_cam=use_cam)
    return Dataset(root, use_wheel=use_wheel, use_cam=use_cam)


class TrajectoryLogger:
    """State_Logger::save_trajectory_to_file (REF: State_Logger.h:160-205)."""

    def __init__(self, path):
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        self.f = open(path, "w")
        self.f.write(traj_header())
        self.n = 0

    def save(self, sys):
        st = sys.state
        self.f.write(traj_format(st.time, np.array(st.imu.p), np.array(st.imu.q), sys.imu_pose_covariance()))
        self.n += 1

    def close(self):
        self.f.close()


def replay(op, dataset=None, trajectory_path=None, device=0, progress=None, max_obs=24, **system_kw):
    """run_bag's main loop.  Returns (SystemManager statistics, times, poses [n][7] = p, q)."""
    ds = dataset if dataset is not None else open_dataset(op.sys.path_bag, use_wheel=op.est.wheel.enabled, use_cam=op.est.cam.enabled)
    sys = SystemManager(op, device=device, max_obs=max_obs, **system_kw)
    path = trajectory_path if trajectory_path is not None else (op.sys.path_trajectory if op.sys.save_trajectory else None)
    log = TrajectoryLogger(path) if path else None
    t_init = ds.t_begin() + op.sys.bag_start                               # run_bag.cpp:214-216
    t_finish = math_inf if op.sys.bag_durr < 0 else t_init + op.sys.bag_durr
    mask = None
    if op.est.cam.enabled and op.est.cam.use_mask.get(0):
        mask = read_image(op.est.cam.mask_path[0])
    times, poses = [], []
    for k, (t, kind, i) in enumerate(ds.msgs):
        if t > t_finish:
            break
        if t < t_init:
            continue
        if kind == IMU:
            r = ds.imu[i]
            if sys.feed_measurement_imu(r[0], r[1:4], r[4:7]):
                st = sys.state
                times.append(st.time), poses.append(np.concatenate([np.array(st.imu.p), np.array(st.imu.q)]))
                if log:
                    log.save(sys)
        elif kind == CAM:
            sys.feed_measurement_camera(t, ds.image(i), mask)
        else:
            r = ds.wheel[i]
            sys.feed_measurement_wheel(r[0], r[1], r[2])
        if progress and k % 2000 == 0:
            progress(sys, t)
    if log:
        log.close()
    stats = dict(sys.stats)
    stats.update(distance_m=sys.distance, time_s={k: round(v, 4) for k, v in sys.tc.total.items()}, initialized=sys.state.initialized,
                 startup_time=sys.state.startup_time, end_time=sys.state.time, n_state=sys.state.n, clone_freq=op.est.clone_freq)
    sys.close()
    return stats, np.array(times), np.array(poses).reshape(-1, 7)


math_inf = float("inf")
