"""Host-side pieces of the replay harness that need no device: image readers, the dataset's message order, time stamps."""
import importlib
import os
import struct
import zlib

import numpy as np
import pytest

import synth_dataset as sd


@pytest.fixture(scope="module")
def rp(pkg):
    return importlib.import_module("plviwo_amd.replay")


def _png(path, a, ctype=0, filt=None):
    """minimal PNG writer with a chosen filter type per row (the reader has to undo all five)"""
    h, w = a.shape[:2]
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    a = a.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft = (y % 5) if filt is None else filt
        line = a[y]
        left = np.concatenate([np.zeros(ch, dtype=np.int32), line[:-ch]])
        ul = np.concatenate([np.zeros(ch, dtype=np.int32), prev[:-ch]])
        if ft == 0:
            out = line
        elif ft == 1:
            out = line - left
        elif ft == 2:
            out = line - prev
        elif ft == 3:
            out = line - ((left + prev) >> 1)
        else:
            pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            out = line - pred
        raw += bytes([ft]) + (out & 255).astype(np.uint8).tobytes()
        prev = line

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) +
                chunk(b"IEND", b""))


def test_image_readers(rp, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    sd.write_pgm(str(tmp_path / "a.pgm"), img)
    assert np.array_equal(rp.read_image(str(tmp_path / "a.pgm")), img)
    with open(tmp_path / "c.pgm", "wb") as f:       # comments and 16-bit samples
        f.write(b"P5\n# a comment\n53 37\n# another\n65535\n" + (img.astype(np.uint16) << 8).astype(">u2").tobytes())
    assert np.array_equal(rp.read_image(str(tmp_path / "c.pgm")), img)
    np.save(tmp_path / "a.npy", img)
    assert np.array_equal(rp.read_image(str(tmp_path / "a.npy")), img)
    _png(str(tmp_path / "g.png"), img)              # all five filter types in one file
    assert np.array_equal(rp.read_image(str(tmp_path / "g.png")), img)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    _png(str(tmp_path / "c.png"), rgb, ctype=2)
    grey = np.clip(np.rint(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]), 0, 255).astype(np.uint8)
    assert np.array_equal(rp.read_image(str(tmp_path / "c.png")), grey)
    with pytest.raises(ValueError):
        rp.read_image(str(tmp_path / "a.jpg"))


def test_dataset_message_order(rp, tmp_path):
    d = str(tmp_path / "ds")
    sd.make_dataset(d, seconds=1.0, render=False)
    ds = rp.Dataset(d)
    assert len(ds.imu) == 221 and len(ds.wheel) == 55 and len(ds.frames) == 10
    t = [m[0] for m in ds.msgs]
    assert t == sorted(t) and len(ds.msgs) == 221 + 55 + 10 and ds.t_begin() == 0.0
    assert ds.frames[3][1].endswith(os.path.join("cam0", "data", "000003.pgm")) and abs(ds.frames[3][0] - (0.35 + sd.CAM_PHASE)) < 1e-9
    # messages with one time stamp: IMU before wheel before camera
    with open(os.path.join(d, "wheel.csv"), "a") as f:
        f.write("1.200000000,5.0,6.0\n")
    with open(os.path.join(d, "imu.csv"), "a") as f:
        f.write("1.200000000,0,0,0,0,0,9.81\n")
    with open(os.path.join(d, "cam0", "data.csv"), "a") as f:
        f.write("1.200000000,zzz.pgm\n")
    ds = rp.Dataset(d)
    assert [m[1] for m in ds.msgs[-3:]] == [rp.IMU, rp.WHEEL, rp.CAM]
    assert len(rp.Dataset(d, use_wheel=False).wheel) == 0 and not rp.Dataset(d, use_cam=False).frames


def test_time_stamps_in_nanoseconds(rp, tmp_path):
    """EuRoC-style files: integer nanoseconds, comma separated, a '#' header; without data.csv the file stem is the stamp."""
    d = tmp_path / "euroc"
    os.makedirs(d / "cam0" / "data")
    (d / "imu.csv").write_text("#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z\n1403636579758555392,0.1,0.2,0.3,9.0,0.1,0.2\n"
                               "1403636579763555584,0.1,0.2,0.3,9.0,0.1,0.2\n")
    sd.write_pgm(str(d / "cam0" / "data" / "1403636579763555584.pgm"), np.zeros((4, 4), dtype=np.uint8))
    ds = rp.Dataset(str(d))
    assert abs(ds.imu[0, 0] - 1403636579.758555392) < 1e-6 and abs(ds.imu[1, 0] - ds.imu[0, 0] - 0.005000192) < 1e-6
    assert len(ds.frames) == 1 and abs(ds.frames[0][0] - ds.imu[1, 0]) < 1e-9
    assert [m[1] for m in ds.msgs] == [rp.IMU, rp.IMU, rp.CAM]


def test_euroc_layout(rp, tmp_path):
    d = tmp_path / "MH_01"
    os.makedirs(d / "mav0" / "cam0" / "data")
    os.makedirs(d / "mav0" / "imu0")
    (d / "mav0" / "imu0" / "data.csv").write_text("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y,w_RS_S_z,a_RS_S_x [m s^-2],a_RS_S_y,a_RS_S_z\n"
                                                  "1403636579758555392,-0.09,0.02,0.07,8.1,-0.3,-3.9\n1403636579763555584,-0.09,0.03,0.07,8.2,-0.3,-3.9\n")
    (d / "mav0" / "cam0" / "data.csv").write_text("#timestamp [ns],filename\n1403636579763555584,1403636579763555584.png\n")
    _png(str(d / "mav0" / "cam0" / "data" / "1403636579763555584.png"), np.full((6, 8), 7, dtype=np.uint8), filt=4)
    ds = rp.Dataset(str(d))
    assert len(ds.imu) == 2 and len(ds.frames) == 1 and ds.imu[1, 4] == 8.2
    assert rp.read_image(ds.frames[0][1]).shape == (6, 8) and [m[1] for m in ds.msgs] == [rp.IMU, rp.IMU, rp.CAM]


def test_stat_matches_the_running_formulas(pkg):
    system = importlib.import_module("plviwo_amd.system")
    s = system.Stat()
    vals = [0.5, 1.5, 0.25, 2.0, 1.0]
    for v in vals:
        s.add_stat(v)
    assert abs(float(s.mean) - np.mean(vals)) < 1e-6 and s.cnt == 5
    s.reset()
    assert s.mean == 0 and s.cnt == 0
    q = system.quat_left_update(np.array([0, 0, 0, 1.0]), np.array([0.02, -0.01, 0.03]))
    R = system.quat_2_Rot(q)
    from scipy.spatial.transform import Rotation
    # R_true = exp(-[dth]x) R_est to first order (the filter's error definition)
    assert np.abs(R - Rotation.from_rotvec([-0.02, 0.01, -0.03]).as_matrix()).max() < 2e-5
