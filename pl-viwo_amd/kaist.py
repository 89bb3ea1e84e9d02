"""KAIST Complex Urban raw-directory reader: BASELINE configs[0] / [4] name urban26 / urban38 / urban39, which the reference plays from
a rosbag (PL-VIWO/launch/rosbag.launch, topics /imu/data_raw, /joint_states, /stereo/left/image_raw); with this reader the same
sequences replay from the dataset's own file layout the moment it is mounted — no ROS, no bag.

    <root>/sensor_data/xsens_imu.csv     t [ns], q_x, q_y, q_z, q_w, eul_x, eul_y, eul_z, gyr_x, gyr_y, gyr_z, acc_x, acc_y, acc_z, mag_x, mag_y, mag_z
    <root>/sensor_data/encoder.csv       t [ns], left count, right count
    <root>/sensor_data/stereo_stamp.csv  t [ns] of every stereo pair (optional: else the file stems)
    <root>/image/stereo_left/<t>.png     1280 x 560, 8-bit Bayer (RGGB)
    <root>/calibration/EncoderParameter.txt   resolution, wheel diameters, wheel base (optional: defaults below)

What the reference's subscribers do with the corresponding messages (REF: PL-VIWO/src/core/ROSHelper.cpp:151-216):
    Imu2Data            t, angular_velocity, linear_acceleration                 -> columns 8-10, 11-13 here
    JointState2Data     m1 = velocity[0], m2 = velocity[1]  (left / right wheel ANGULAR velocity, rad/s: Wheel3DAng,
                        PL-VIWO/config/kaist/kaist_C/config_wheel.yaml:3-26, intrinsics = radii 0.3 / 0.3, base 1.5)
                        -> 2 pi (count[i+1] - count[i]) / (resolution dt), stamped at the later sample
    Image2Data          cv_bridge::toCvShare(msg, MONO8): a Bayer image goes through cv::cvtColor(COLOR_BayerRG2GRAY)
                        -> bayer_rg_to_grey below (OpenCV's fixed-point weights R 4899, G 9617, B 1868 / 2^14, the four / two
                        neighbours of the missing colours averaged inside the same rounding; border pixels replicated)
The dataset is not in this container: the layout and the Bayer contract are restated from the dataset's documentation and OpenCV's
demosaicing source as recalled (SURVEY.md Appendix A's caveat applies); tests/test_kaist_reader.py exercises the reader on a synthetic
directory in this layout.
"""
import math
import os

import numpy as np

IMU, WHEEL, CAM = 0, 1, 2
ENCODER_DEFAULTS = dict(resolution=4096.0, left_diameter=0.623479, right_diameter=0.622806, wheel_base=1.52439)


def bayer_rg_to_grey(b):
    """cv::cvtColor(bayer, COLOR_BayerRG2GRAY) for an 8-bit RGGB mosaic (row 0: R G R G ..., row 1: G B G B ...)."""
    b = np.ascontiguousarray(b, dtype=np.uint8)
    h, w = b.shape
    p = b.astype(np.int64)
    R2Y, G2Y, B2Y, SHIFT = 4899, 9617, 1868, 14
    out = np.zeros((h, w), dtype=np.int64)
    c = p[1:-1, 1:-1]
    cross = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:]
    diag = p[:-2, :-2] + p[:-2, 2:] + p[2:, :-2] + p[2:, 2:]
    vert = p[:-2, 1:-1] + p[2:, 1:-1]
    horz = p[1:-1, :-2] + p[1:-1, 2:]
    yy, xx = np.mgrid[1:h - 1, 1:w - 1]
    red, blue = (yy % 2 == 0) & (xx % 2 == 0), (yy % 2 == 1) & (xx % 2 == 1)
    g_on_red_row, g_on_blue_row = (yy % 2 == 0) & (xx % 2 == 1), (yy % 2 == 1) & (xx % 2 == 0)
    half2, half1 = 1 << (SHIFT + 1), 1 << SHIFT
    v = np.zeros_like(c)
    v[red] = ((diag * B2Y + cross * G2Y + c * (4 * R2Y) + half2) >> (SHIFT + 2))[red]
    v[blue] = ((diag * R2Y + cross * G2Y + c * (4 * B2Y) + half2) >> (SHIFT + 2))[blue]
    v[g_on_red_row] = ((horz * R2Y + vert * B2Y + c * (2 * G2Y) + half1) >> (SHIFT + 1))[g_on_red_row]
    v[g_on_blue_row] = ((horz * B2Y + vert * R2Y + c * (2 * G2Y) + half1) >> (SHIFT + 1))[g_on_blue_row]
    out[1:-1, 1:-1] = v
    out[0, :], out[-1, :] = out[1, :], out[-2, :]
    out[:, 0], out[:, -1] = out[:, 1], out[:, -2]
    return np.clip(out, 0, 255).astype(np.uint8)


def read_encoder_parameters(path):
    p = dict(ENCODER_DEFAULTS)
    if not os.path.exists(path):
        return p
    with open(path) as f:
        for line in f:
            low = line.lower()
            if ":" not in line:
                continue
            try:
                val = float(line.split(":")[1].split()[0])
            except (IndexError, ValueError):
                continue
            if "resolution" in low:
                p["resolution"] = val
            elif "left" in low and "diameter" in low:
                p["left_diameter"] = val
            elif "right" in low and "diameter" in low:
                p["right_diameter"] = val
            elif "base" in low:
                p["wheel_base"] = val
    return p


def is_kaist_raw(root):
    return os.path.exists(os.path.join(root, "sensor_data", "xsens_imu.csv"))


def _rows(path):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield [x for x in line.replace(",", " ").split()]


class KaistDataset:
    """The message list replay.replay() walks (same interface as replay.Dataset): imu [n][7], wheel [n][3] (t, left, right wheel
    angular velocity in rad/s), frames [(t, path)], msgs sorted by time (IMU before wheel before camera at equal stamps)."""

    def __init__(self, root, use_wheel=True, use_cam=True):
        self.root = root
        imu = []
        for r in _rows(os.path.join(root, "sensor_data", "xsens_imu.csv")):
            if len(r) < 14:
                raise ValueError("xsens_imu.csv: expected 17 columns (the sequences without raw gyro / accelerometer columns cannot drive the filter)")
            imu.append([int(r[0]) * 1e-9] + [float(x) for x in r[8:11]] + [float(x) for x in r[11:14]])
        self.imu = np.array(imu).reshape(-1, 7)
        self.encoder = read_encoder_parameters(os.path.join(root, "calibration", "EncoderParameter.txt"))
        self.wheel = np.zeros((0, 3))
        ep = os.path.join(root, "sensor_data", "encoder.csv")
        if use_wheel and os.path.exists(ep):
            enc = np.array([[int(r[0]), int(r[1]), int(r[2])] for r in _rows(ep)], dtype=np.int64).reshape(-1, 3)
            if len(enc) >= 2:
                dt = np.diff(enc[:, 0]) * 1e-9
                k = 2.0 * math.pi / self.encoder["resolution"]
                ok = dt > 0
                self.wheel = np.column_stack([enc[1:, 0] * 1e-9, k * np.diff(enc[:, 1]) / np.where(ok, dt, 1.0),
                                              k * np.diff(enc[:, 2]) / np.where(ok, dt, 1.0)])[ok]
        self.frames = []
        idir = os.path.join(root, "image", "stereo_left")
        if use_cam and os.path.isdir(idir):
            sp = os.path.join(root, "sensor_data", "stereo_stamp.csv")
            stamps = [int(r[0]) for r in _rows(sp)] if os.path.exists(sp) else sorted(int(os.path.splitext(n)[0]) for n in os.listdir(idir))
            for s in stamps:
                path = os.path.join(idir, f"{s}.png")
                if os.path.exists(path):
                    self.frames.append((s * 1e-9, path))
        msgs = [(t, IMU, i) for i, t in enumerate(self.imu[:, 0])] + [(t, WHEEL, i) for i, t in enumerate(self.wheel[:, 0])] + \
               [(t, CAM, i) for i, (t, _) in enumerate(self.frames)]
        msgs.sort(key=lambda m: (m[0], m[1]))
        self.msgs = msgs

    def t_begin(self):
        return self.msgs[0][0]

    def image(self, i):
        from .replay import read_image
        return bayer_rg_to_grey(read_image(self.frames[i][1]))

    def wheel_intrinsics(self):
        """(r_l, r_r, base) of EncoderParameter.txt, the values config_wheel.yaml's `intrinsics` rounds to 0.3 / 0.3 / 1.5"""
        e = self.encoder
        return 0.5 * e["left_diameter"], 0.5 * e["right_diameter"], e["wheel_base"]
