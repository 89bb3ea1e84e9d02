"""Test infrastructure: a rendered dataset (tests/synth_dataset.py layout) rewritten in the KAIST Complex Urban raw layout that
pl-viwo_amd/kaist.py reads (BASELINE configs[0] / [4] name urban26 / urban38 / urban39, which are not in this container): Bayer RGGB
PNG frames under image/stereo_left, sensor_data/xsens_imu.csv (17 columns, ns stamps), encoder counts, EncoderParameter.txt.  What
the reference's subscribers apply to such messages (REF: PL-VIWO/src/core/ROSHelper.cpp:151-216, config/kaist/kaist_C/config_wheel.yaml:3-26)
is then exercised end to end by the replay driver: tests/test_kaist_reader.py (CPU dry run) and tests/test_gpu_kaist_replay.py."""
import importlib
import math
import os
import struct
import zlib

import numpy as np

T0_NS = 1544590798000000000   # (a 2018 stamp, the order of magnitude of the real sequences: double seconds resolve 0.24 us there)


def write_png(path, a):
    h, w = a.shape
    raw = b"".join(b"\x00" + a[i].tobytes() for i in range(h))
    ch = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + ch(b"IDAT", zlib.compress(raw, 1)) + ch(b"IEND", b""))


def convert(src_dir, dst_dir, radius_l, radius_r, base, resolution=4096, t0_ns=None):
    """src_dir: imu.csv, wheel.csv, cam0/ (replay.Dataset) -> dst_dir in the KAIST raw layout.  A grey image is written as the Bayer
    mosaic whose three colour planes all equal it (the reader's demosaicing then returns a lightly smoothed grey image: both replays of
    a parity test see the same one).  Wheel angular velocities become accumulated encoder counts (quantised: 2 pi / resolution rad)."""
    replay = importlib.import_module("plviwo_amd.replay")
    ds = replay.Dataset(src_dir)
    for d in ("sensor_data", os.path.join("image", "stereo_left"), "calibration"):
        os.makedirs(os.path.join(dst_dir, d), exist_ok=True)
    t0_ns = T0_NS if t0_ns is None else int(t0_ns)
    ns = lambda t: t0_ns + int(round(t * 1e9))
    with open(os.path.join(dst_dir, "sensor_data", "xsens_imu.csv"), "w") as f:
        for r in ds.imu:
            row = [ns(r[0]), 0, 0, 0, 1, 0, 0, 0] + [repr(float(x)) for x in r[1:4]] + [repr(float(x)) for x in r[4:7]] + [0, 0, 0]
            f.write(",".join(str(x) for x in row) + "\n")
    if len(ds.wheel):
        k = resolution / (2.0 * math.pi)
        t = ds.wheel[:, 0]
        dt = np.diff(t, prepend=t[0] - (t[1] - t[0]))
        cl = np.round(np.cumsum(ds.wheel[:, 1] * dt * k)).astype(np.int64)
        cr = np.round(np.cumsum(ds.wheel[:, 2] * dt * k)).astype(np.int64)
        with open(os.path.join(dst_dir, "sensor_data", "encoder.csv"), "w") as f:
            f.write(f"{ns(t[0] - dt[0])},0,0\n")
            for i in range(len(t)):
                f.write(f"{ns(t[i])},{cl[i]},{cr[i]}\n")
    with open(os.path.join(dst_dir, "calibration", "EncoderParameter.txt"), "w") as f:
        f.write(f"Encoder calibrated parameter\nEncoder resolution: {resolution}\nEncoder left wheel diameter: {2 * radius_l!r}\n"
                f"Encoder right wheel diameter: {2 * radius_r!r}\nEncoder wheel base: {base!r}\n")
    with open(os.path.join(dst_dir, "sensor_data", "stereo_stamp.csv"), "w") as f:
        for i, (t, _) in enumerate(ds.frames):
            s = ns(t)
            write_png(os.path.join(dst_dir, "image", "stereo_left", f"{s}.png"), np.ascontiguousarray(ds.image(i), dtype=np.uint8))
            f.write(f"{s}\n")
    return dst_dir
