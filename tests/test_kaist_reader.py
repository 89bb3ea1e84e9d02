"""KAIST Complex Urban raw layout (pl-viwo_amd/kaist.py) on a three-frame synthetic directory: the conversions the reference's ROS
subscribers apply (ROSHelper.cpp:151-216; config_wheel.yaml:3-26) and the hand-over to the replay driver.  The real sequences are not
in this container (BASELINE configs[0] / [4]); this is the dry run that makes them one mount away."""
import importlib
import math
import os
import struct
import zlib

import numpy as np


from kaist_synth import write_png as _png


def _make(root, rng):
    os.makedirs(os.path.join(root, "sensor_data"))
    os.makedirs(os.path.join(root, "image", "stereo_left"))
    os.makedirs(os.path.join(root, "calibration"))
    t0 = 1544590798000000000
    with open(os.path.join(root, "sensor_data", "xsens_imu.csv"), "w") as f:
        for i in range(40):
            row = [t0 + i * 10000000, 0, 0, 0, 1, 0, 0, 0, 0.01 * i, -0.02, 0.03, 0.1, 0.2, 9.8 + 0.001 * i, 0, 0, 0]
            f.write(",".join(str(x) for x in row) + "\n")
    with open(os.path.join(root, "sensor_data", "encoder.csv"), "w") as f:
        for i in range(21):     # 100 Hz, left 41 counts / sample, right 40
            f.write(f"{t0 + i * 10000000},{1000 + 41 * i},{2000 + 40 * i}\n")
    with open(os.path.join(root, "calibration", "EncoderParameter.txt"), "w") as f:
        f.write("Encoder calibrated parameter\nEncoder resolution: 4096\nEncoder left wheel diameter: 0.623479\n"
                "Encoder right wheel diameter: 0.622806\nEncoder wheel base: 1.52439\n")
    stamps, imgs = [], []
    with open(os.path.join(root, "sensor_data", "stereo_stamp.csv"), "w") as f:
        for i in range(3):
            s = t0 + 5000000 + i * 100000000
            img = rng.integers(0, 256, (56, 128), dtype=np.uint8)
            _png(os.path.join(root, "image", "stereo_left", f"{s}.png"), img)
            f.write(f"{s}\n")
            stamps.append(s), imgs.append(img)
    return t0, stamps, imgs


def test_kaist_raw_directory(pkg, tmp_path):
    kaist = importlib.import_module("plviwo_amd.kaist")
    replay = importlib.import_module("plviwo_amd.replay")
    rng = np.random.default_rng(1)
    root = str(tmp_path / "urban26")
    t0, stamps, imgs = _make(root, rng)
    assert kaist.is_kaist_raw(root)
    ds = replay.open_dataset(root)
    assert isinstance(ds, kaist.KaistDataset)
    # IMU: seconds, gyro columns 8-10, accelerometer 11-13
    assert ds.imu.shape == (40, 7) and abs(ds.imu[0, 0] - t0 * 1e-9) < 1e-6
    assert np.allclose(ds.imu[3, 1:4], [0.03, -0.02, 0.03]) and np.allclose(ds.imu[3, 4:7], [0.1, 0.2, 9.803])
    # encoder counts -> wheel angular velocity [rad/s], stamped at the later sample
    assert ds.wheel.shape == (20, 3)
    assert np.allclose(ds.wheel[:, 1], 2 * math.pi * 41 / 4096 / 0.01) and np.allclose(ds.wheel[:, 2], 2 * math.pi * 40 / 4096 / 0.01)
    assert abs(ds.wheel[0, 0] - (t0 + 10000000) * 1e-9) < 1e-6
    rl, rr, base = ds.wheel_intrinsics()
    assert abs(rl - 0.3117395) < 1e-9 and abs(rr - 0.311403) < 1e-9 and base == 1.52439
    # left wheel speed 41 counts per 10 ms -> 6.29 rad/s x 0.3117 m = 1.96 m/s: a car
    assert 1.9 < ds.wheel[0, 1] * rl < 2.0
    # frames and the merged message order (IMU before camera at equal or earlier stamps)
    assert len(ds.frames) == 3 and all(abs(t - s * 1e-9) < 1e-6 for (t, _), s in zip(ds.frames, stamps))
    order = [k for _, k, _ in ds.msgs]
    assert order.count(2) == 3 and order[0] == 0
    assert all(ds.msgs[i][0] <= ds.msgs[i + 1][0] for i in range(len(ds.msgs) - 1))
    # Bayer -> grey: constant colour planes give the BT.601 grey; size kept
    g = ds.image(0)
    assert g.shape == imgs[0].shape and g.dtype == np.uint8
    mosaic = np.zeros((8, 8), dtype=np.uint8)
    mosaic[0::2, 0::2], mosaic[0::2, 1::2], mosaic[1::2, 0::2], mosaic[1::2, 1::2] = 200, 100, 100, 50    # R G / G B
    want = (200 * 4899 + 100 * 9617 + 50 * 1868 + 8192) >> 14
    assert np.all(np.abs(kaist.bayer_rg_to_grey(mosaic).astype(int) - want) <= 1)
    assert np.all(kaist.bayer_rg_to_grey(np.full((6, 6), 77, dtype=np.uint8)) == 77)


def test_converter_round_trip(pkg, tmp_path):
    """tests/kaist_synth.py (the generator of the GPU suite's KAIST-layout replay): a rendered drive rewritten in the KAIST layout reads
    back as the same IMU stream, the same frame times, wheel speeds to the encoder's quantisation and frames to the demosaicing."""
    import kaist_synth
    import synth_dataset as sd
    replay = importlib.import_module("plviwo_amd.replay")
    src = str(tmp_path / "src")
    sd.make_dataset(src, seconds=1.0, cam_hz=10.0, style="street", workers=4)
    dst = kaist_synth.convert(src, str(tmp_path / "urban_synth"), sd.RL, sd.RR, sd.BASE)
    a, b = replay.Dataset(src), replay.open_dataset(dst)
    off = kaist_synth.T0_NS * 1e-9
    assert b.imu.shape == a.imu.shape and np.array_equal(b.imu[:, 1:], a.imu[:, 1:]) and np.abs(b.imu[:, 0] - off - a.imu[:, 0]).max() < 1e-6
    assert len(b.frames) == len(a.frames) and all(abs(tb - off - ta) < 1e-6 for (tb, _), (ta, _) in zip(b.frames, a.frames))
    assert b.wheel.shape == a.wheel.shape
    q = 2 * math.pi / 4096 / np.diff(a.wheel[:, 0]).min()       # one count per sample interval
    assert np.abs(b.wheel[:, 1:] - a.wheel[:, 1:]).max() <= 1.01 * q
    ia, ib = a.image(0).astype(int), b.image(0).astype(int)
    assert ia.shape == ib.shape and np.abs(ia - ib)[2:-2, 2:-2].mean() < 6.0
    rl, rr, base = b.wheel_intrinsics()
    assert (rl, rr, base) == (sd.RL, sd.RR, sd.BASE)
