"""ctypes binding of oracle/liboracle.so — the CPU restatement used ONLY as the checker."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
FRAME_LIB = None   # bench.py's cpu_baseline: another build of the same sources for FrameOracle only (build_native: -march=native, timing)


def build_native(out_dir):
    """oracle/*.cpp compiled for the machine this runs on (-O3 -march=native, contraction allowed): used by bench.py's cpu_baseline for
    TIMING only (SURVEY 8(d) / BASELINE.md ask for -march=native); the parity tests keep liboracle.so (-ffp-contract=off).  Returns
    the path, or None when the compiler is missing or fails."""
    import glob
    out = os.path.join(out_dir, "liboracle_native.so")
    cmd = ["g++", "-O3", "-march=native", "-ffp-contract=fast", "-std=c++17", "-fPIC", "-Wno-unused-function", "-shared", "-o", out] + \
        sorted(glob.glob(os.path.join(ORACLE_DIR, "*.cpp"))) + ["-lpthread"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return out

dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)


def _dp(a):
    return a.ctypes.data_as(dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(ip) if a is not None else None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_make_givens.argtypes = [C.c_double, C.c_double, dp, dp]
        L.orc_nullspace_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp]
        L.orc_compress.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]
        L.orc_compress.restype = C.c_int
        L.orc_chi2_batch.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, ip, C.c_double, dp]
        L.orc_ekf_update.argtypes = [dp, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp]
        L.orc_ekf_update.restype = C.c_int
        L.orc_msckf_update.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp, ip,
                                       C.c_double, C.c_double, C.c_double, dp, u8p, ip, dp]
        L.orc_msckf_update.restype = C.c_int

    def make_givens(self, p, q):
        c, s = C.c_double(), C.c_double()
        self.lib.orc_make_givens(p, q, C.byref(c), C.byref(s))
        return c.value, s.value

    def nullspace_batch(self, rows, Hf, Hx, res):
        Hf = np.ascontiguousarray(Hf, dtype=np.float64).copy()
        Hx = np.ascontiguousarray(Hx, dtype=np.float64).copy()
        res = np.ascontiguousarray(res, dtype=np.float64).copy()
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        F, fdim, ld = Hf.shape
        k = Hx.shape[1]
        self.lib.orc_nullspace_batch(F, fdim, k, ld, _ip(rows), _dp(Hf), _dp(Hx), _dp(res))
        return Hf, Hx, res

    def compress(self, H, res):
        H = np.asfortranarray(H, dtype=np.float64).copy(order="F")
        res = np.ascontiguousarray(res, dtype=np.float64).copy()
        m, k = H.shape
        mo = self.lib.orc_compress(_dp(H), m, k, m, _dp(res))
        Hout = H[:mo, :].copy()
        return Hout, res[:mo].copy()

    def chi2_batch(self, P, rows, Hx, res, cols, sigma2):
        P = np.asfortranarray(P, dtype=np.float64)
        Hx = np.ascontiguousarray(Hx, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        F, k, ld = Hx.shape
        chi = np.zeros(F)
        self.lib.orc_chi2_batch(_dp(P), P.shape[0], P.shape[0], F, k, ld, _ip(rows), _dp(Hx), _dp(res), _ip(cols),
                                float(sigma2), _dp(chi))
        return chi

    def ekf_update(self, P, H, cols, res, Rdiag=None):
        P = np.asfortranarray(P, dtype=np.float64).copy(order="F")
        H = np.asfortranarray(H, dtype=np.float64)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        res = np.ascontiguousarray(res, dtype=np.float64)
        Rd = np.ascontiguousarray(Rdiag, dtype=np.float64) if Rdiag is not None else None
        n, (r, k) = P.shape[0], H.shape
        dx = np.zeros(n)
        rc = self.lib.orc_ekf_update(_dp(P), n, n, _dp(H), r, k, r, _ip(cols), _dp(res), _dp(Rd), _dp(dx))
        return rc, P, dx

    def msckf_update(self, P, rows, Hf, Hx, res, cols, sigma2, q95, chi2_mult=1.0, res_norm_gate=3.0):
        P = np.asfortranarray(P, dtype=np.float64).copy(order="F")
        Hf = np.ascontiguousarray(Hf, dtype=np.float64)
        Hx = np.ascontiguousarray(Hx, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        q95 = np.ascontiguousarray(q95, dtype=np.float64)
        F, fdim, ld = Hf.shape
        k = Hx.shape[1]
        n = P.shape[0]
        dx = np.zeros(n)
        acc = np.zeros(F, dtype=np.uint8)
        nrows = C.c_int()
        rc = self.lib.orc_msckf_update(_dp(P), n, n, F, fdim, k, ld, _ip(rows), _dp(Hf), _dp(Hx), _dp(res), _ip(cols),
                                       float(sigma2), float(chi2_mult), float(res_norm_gate), _dp(q95),
                                       acc.ctypes.data_as(u8p), C.byref(nrows), _dp(dx))
        return rc, P, dx, acc, nrows.value


def _slam_bind(cls):
    def slam_update(self, P, H, res, cols, q95, chi2_mult=1.0):
        P = np.asfortranarray(P, dtype=np.float64).copy(order="F")
        H = np.asfortranarray(H, dtype=np.float64)
        rows, k = H.shape
        n = P.shape[0]
        res = np.ascontiguousarray(res, dtype=np.float64)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        q95 = np.ascontiguousarray(q95, dtype=np.float64)
        acc = np.zeros(1, dtype=np.uint8)
        dx = np.zeros(n)
        rc = self.lib.orc_slam_update(_dp(P), n, n, _dp(H), rows, k, rows, _ip(cols), _dp(res), float(chi2_mult), _dp(q95),
                                      acc.ctypes.data_as(u8p), _dp(dx))
        return rc, P, int(acc[0]), dx

    def slam_initialize(self, P, Hf, Hx, res, cols, q95, chi2_mult=1.0):
        P = np.asfortranarray(P, dtype=np.float64)
        Hf, Hx = np.asfortranarray(Hf, dtype=np.float64), np.asfortranarray(Hx, dtype=np.float64)
        rows, k = Hx.shape
        n = P.shape[0]
        res = np.ascontiguousarray(res, dtype=np.float64)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        q95 = np.ascontiguousarray(q95, dtype=np.float64)
        P2 = np.zeros((n + 3, n + 3), order="F")
        dxi, dx = np.zeros(3), np.zeros(n + 3)
        self.lib.orc_slam_initialize.restype = C.c_int
        ok = self.lib.orc_slam_initialize(_dp(P), n, n, rows, k, rows, _dp(Hf), _dp(Hx), _dp(res), _ip(cols), C.c_double(chi2_mult),
                                          _dp(q95), _dp(P2), _dp(dxi), _dp(dx))
        return ok, P2, dxi, dx

    def cov_marginalize(self, P, idx, size):
        P = np.asfortranarray(P, dtype=np.float64)
        n = P.shape[0]
        out = np.zeros((n - size, n - size), order="F")
        self.lib.orc_cov_marginalize(_dp(P), n, int(idx), int(size), _dp(out))
        return out

    cls.slam_update, cls.slam_initialize, cls.cov_marginalize = slam_update, slam_initialize, cov_marginalize


_inst = None


def load():
    global _inst
    if _inst is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        _slam_bind(Oracle)
        _inst = Oracle(C.CDLL(LIB))
        _inst.lib.orc_slam_update.argtypes = [dp, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_int, ip, dp, C.c_double, dp, u8p, dp]
        _inst.lib.orc_slam_update.restype = C.c_int
    return _inst


# ------------------------------------------------------------------ front-end oracle
fp, u8 = C.POINTER(C.c_float), C.POINTER(C.c_uint8)


class FrontOracle:
    def clahe(self, img, clip=10.0, tiles=8):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros_like(img)
        self.lib.orc_clahe.argtypes = [u8, C.c_int, C.c_int, C.c_double, C.c_int, u8]
        self.lib.orc_clahe.restype = None
        self.lib.orc_clahe(img.ctypes.data_as(u8), w, h, float(clip), tiles, out.ctypes.data_as(u8))
        return out

    def __init__(self, lib):
        self.lib = L = lib
        L.orc_equalize_hist.argtypes = [u8, C.c_int, C.c_int, C.c_int, u8]
        L.orc_pyramid_build.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_pyramid_build.restype = C.c_void_p
        L.orc_pyramid_free.argtypes = [C.c_void_p]
        L.orc_pyr_down.argtypes = [u8, C.c_int, C.c_int, u8, C.c_int, C.c_int]
        L.orc_pyramid_levels.argtypes = [C.c_void_p]
        L.orc_pyramid_levels.restype = C.c_int
        L.orc_pyramid_level.argtypes = [C.c_void_p, C.c_int, ip, ip, u8, C.POINTER(C.c_int16)]
        L.orc_lk_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, fp, fp, u8, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.POINTER(C.c_longlong)]
        L.orc_undistort.argtypes = [dp, C.c_int, fp, fp]
        L.orc_ransac_fundamental.argtypes = [fp, fp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint32, u8, ip]
        L.orc_ransac_fundamental.restype = C.c_int
        L.orc_run7point.argtypes = [fp, fp, ip, dp]
        L.orc_run7point.restype = C.c_int
        L.orc_perform_matching.argtypes = [C.c_void_p, C.c_void_p, C.c_int, fp, fp, dp, C.c_int, C.c_int, C.c_float,
                                           C.c_double, C.c_double, C.c_int, C.c_uint32, u8, fp, fp, C.c_int]
        L.orc_perform_matching.restype = C.c_int

    def equalize_hist(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty_like(img)
        self.lib.orc_equalize_hist(img.ctypes.data_as(u8), img.shape[1], img.shape[0], img.shape[1],
                                   out.ctypes.data_as(u8))
        return out

    def pyramid(self, img, win=15, max_level=5):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        return OraclePyramid(self, self.lib.orc_pyramid_build(img.ctypes.data_as(u8), img.shape[1], img.shape[0],
                                                              img.shape[1], win, max_level))

    def downsample(self, img):
        """cv::pyrDown(img, Size(cols / 2.0, rows / 2.0))"""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros((h // 2, w // 2), dtype=np.uint8)
        self.lib.orc_pyr_down(img.ctypes.data_as(u8), w, h, out.ctypes.data_as(u8), w // 2, h // 2)
        return out

    def lk_track(self, prev, cur, pts0, pts1_init, win=15, max_iters=30, eps=0.01, nthreads=1):
        pts0 = np.ascontiguousarray(pts0, dtype=np.float32)
        pts1 = np.ascontiguousarray(pts1_init, dtype=np.float32).copy()
        n = pts0.shape[0]
        st = np.zeros(n, dtype=np.uint8)
        iters = C.c_longlong()
        self.lib.orc_lk_track(prev.h, cur.h, n, pts0.ctypes.data_as(fp), pts1.ctypes.data_as(fp), st.ctypes.data_as(u8),
                              win, max_iters, eps, nthreads, C.byref(iters))
        return pts1, st, iters.value

    def undistort(self, K8, uv):
        K8 = np.ascontiguousarray(K8, dtype=np.float64)
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        out = np.empty_like(uv)
        self.lib.orc_undistort(_dp(K8), uv.shape[0], uv.ctypes.data_as(fp), out.ctypes.data_as(fp))
        return out

    def ransac(self, m1, m2, thr, conf=0.999, max_iters=1000, seed=0):
        m1 = np.ascontiguousarray(m1, dtype=np.float32)
        m2 = np.ascontiguousarray(m2, dtype=np.float32)
        n = m1.shape[0]
        mask = np.zeros(n, dtype=np.uint8)
        it = C.c_int()
        good = self.lib.orc_ransac_fundamental(m1.ctypes.data_as(fp), m2.ctypes.data_as(fp), n, thr, conf, max_iters,
                                               seed, mask.ctypes.data_as(u8), C.byref(it))
        return mask, good, it.value

    def run7point(self, m1, m2, idx):
        m1 = np.ascontiguousarray(m1, dtype=np.float32)
        m2 = np.ascontiguousarray(m2, dtype=np.float32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F = np.zeros(27)
        n = self.lib.orc_run7point(m1.ctypes.data_as(fp), m2.ctypes.data_as(fp), _ip(idx), _dp(F))
        return F.reshape(3, 3, 3)[:n]

    def perform_matching(self, prev, cur, pts0, pts1_init, K8, win=15, max_iters=30, eps=0.01, thr_px=2.0, conf=0.999,
                         ransac_iters=1000, seed=0, nthreads=1):
        pts0 = np.ascontiguousarray(pts0, dtype=np.float32)
        pts1 = np.ascontiguousarray(pts1_init, dtype=np.float32).copy()
        K8 = np.ascontiguousarray(K8, dtype=np.float64)
        n = pts0.shape[0]
        mask = np.zeros(n, dtype=np.uint8)
        n0 = np.zeros((n, 2), dtype=np.float32)
        n1 = np.zeros((n, 2), dtype=np.float32)
        rc = self.lib.orc_perform_matching(prev.h, cur.h, n, pts0.ctypes.data_as(fp), pts1.ctypes.data_as(fp), _dp(K8),
                                           win, max_iters, eps, thr_px, conf, ransac_iters, seed,
                                           mask.ctypes.data_as(u8), n0.ctypes.data_as(fp), n1.ctypes.data_as(fp),
                                           nthreads)
        return rc, pts1, mask, n0, n1


class OraclePyramid:
    def __init__(self, fo, h):
        self.fo, self.h = fo, h
        self.levels = fo.lib.orc_pyramid_levels(h)

    def level(self, l):
        w, hh = C.c_int(), C.c_int()
        self.fo.lib.orc_pyramid_level(self.h, l, C.byref(w), C.byref(hh), None, None)
        img = np.zeros((hh.value, w.value), dtype=np.uint8)
        der = np.zeros((hh.value, w.value, 2), dtype=np.int16)
        self.fo.lib.orc_pyramid_level(self.h, l, C.byref(w), C.byref(hh), img.ctypes.data_as(u8),
                                      der.ctypes.data_as(C.POINTER(C.c_int16)))
        return img, der

    def __del__(self):
        try:
            self.fo.lib.orc_pyramid_free(self.h)
        except Exception:
            pass


class DetectOracle:
    def __init__(self, lib):
        self.lib = L = lib
        L.orc_fast_roi.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_fast_roi.restype = C.c_int
        L.orc_corner_subpix.argtypes = [u8, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_double]
        L.orc_perform_detection.argtypes = [u8, u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp,
                                            C.POINTER(C.c_uint64), C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_perform_detection.restype = C.c_int

    def _unused(self):
        pass

    def fast_roi(self, img, x0, y0, w, h, thr, cap=20000):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        xy = np.zeros((cap, 2), dtype=np.float32)
        r = np.zeros(cap, dtype=np.float32)
        n = self.lib.orc_fast_roi(img.ctypes.data_as(u8), img.shape[1], x0, y0, w, h, thr, cap, xy.ctypes.data_as(fp),
                                  r.ctypes.data_as(fp))
        return xy[:n].copy(), r[:n].copy()

    def corner_subpix(self, img, xy, win=5, max_iters=20, eps=0.001):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        xy = np.ascontiguousarray(xy, dtype=np.float32).copy()
        self.lib.orc_corner_subpix(img.ctypes.data_as(u8), img.shape[1], img.shape[0], len(xy), xy.ctypes.data_as(fp), win,
                                   max_iters, eps)
        return xy

    def perform_detection(self, img, mask, pts, ids, currid, num_features, grid_x, grid_y, min_px_dist, threshold, cap=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        mask = np.zeros((h, w), np.uint8) if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        n_in = len(pts)
        cap = cap or (n_in + 4 * num_features + 64)
        P = np.zeros((cap, 2), dtype=np.float32)
        I = np.zeros(cap, dtype=np.uint64)
        P[:n_in] = pts
        I[:n_in] = ids
        cid = C.c_uint64(currid)
        n = self.lib.orc_perform_detection(img.ctypes.data_as(u8), mask.ctypes.data_as(u8), w, h, num_features, grid_x, grid_y,
                                           min_px_dist, threshold, P.ctypes.data_as(fp),
                                           I.ctypes.data_as(C.POINTER(C.c_uint64)), n_in, cap, C.byref(cid))
        return P[:n].copy(), I[:n].copy(), cid.value


_det = None


def load_detect():
    global _det
    if _det is None:
        load()
        _det = DetectOracle(_inst.lib)
    return _det


_front = None


def load_front():
    global _front
    if _front is None:
        load()
        _front = FrontOracle(_inst.lib)
    return _front


# ------------------------------------------------------------------ Jacobian / triangulation oracle
class JacOracle:
    def __init__(self, lib, pkg):
        self.lib, self.pkg = lib, pkg
        sv, tk = C.POINTER(pkg.PlvStateView), C.POINTER(pkg.PlvTracks)
        lib.orc_jacobian_columns.argtypes = [sv, tk, ip, C.c_int, ip]
        lib.orc_jacobian_columns.restype = C.c_int
        lib.orc_interpolate.argtypes = [sv, C.c_double, C.c_int, dp, dp, dp, dp, ip]
        lib.orc_interpolate.restype = C.c_int
        lib.orc_build_jacobians.argtypes = [sv, tk, C.c_int, ip, C.c_int, ip, dp, dp, dp]
        lib.orc_build_jacobians.restype = C.c_int
        lib.orc_triangulate.argtypes = [C.c_int, dp, dp, fp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp]
        lib.orc_triangulate.restype = C.c_int
        lib.orc_triangulate_batch.argtypes = [sv, tk, C.POINTER(pkg.PlvTriOptions), dp, u8p, dp]
        lib.orc_triangulate_batch.restype = C.c_int
        lk = C.POINTER(pkg.PlvLineTracks)
        lib.orc_line_jacobian_columns.argtypes = [sv, lk, ip, C.c_int, ip]
        lib.orc_line_jacobian_columns.restype = C.c_int
        lib.orc_build_line_jacobians.argtypes = [sv, lk, C.c_int, ip, C.c_int, ip, dp, dp, dp]
        lib.orc_build_line_jacobians.restype = C.c_int
        lib.orc_triangulate_lines.argtypes = [sv, lk, dp, u8p]
        lib.orc_triangulate_lines.restype = C.c_int
        lib.orc_cpi_poses.argtypes = [sv, C.POINTER(pkg.PlvCpiTable), C.c_int, dp, dp, dp, u8p]
        lib.orc_cpi_poses.restype = C.c_int

    def cpi_poses(self, st, cpi, t_q):
        t_q = np.ascontiguousarray(t_q, dtype=np.float64)
        R, p, ok = np.zeros((len(t_q), 9)), np.zeros((len(t_q), 3)), np.zeros(len(t_q), dtype=np.uint8)
        assert self.lib.orc_cpi_poses(C.byref(st.c), C.byref(cpi.c), len(t_q), _dp(t_q), _dp(R), _dp(p), ok.ctypes.data_as(u8p)) == 0
        return R, p, ok

    def line_columns(self, st, lt, cap=512):
        cols = np.zeros(cap, dtype=np.int32)
        k = C.c_int()
        rc = self.lib.orc_line_jacobian_columns(C.byref(st.c), C.byref(lt.c), _ip(cols), cap, C.byref(k))
        assert rc == 0
        return cols[:k.value].copy()

    def build_line_jacobians(self, st, lt, cols, ld):
        L, k = lt.c.n_lines, len(cols)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        rows = np.zeros(L, dtype=np.int32)
        Hf, Hx, res = np.zeros((L, 6, ld)), np.zeros((L, k, ld)), np.zeros((L, ld))
        rc = self.lib.orc_build_line_jacobians(C.byref(st.c), C.byref(lt.c), k, _ip(cols), ld, _ip(rows), _dp(Hf), _dp(Hx),
                                               _dp(res))
        assert rc == 0, rc
        return rows, Hf, Hx, res

    def triangulate_lines(self, st, lt):
        L = lt.c.n_lines
        out, ok = np.zeros((L, 6)), np.zeros(L, dtype=np.uint8)
        rc = self.lib.orc_triangulate_lines(C.byref(st.c), C.byref(lt.c), _dp(out), ok.ctypes.data_as(u8p))
        assert rc == 0
        return out, ok

    def columns(self, st, tr, cap=512):
        cols = np.zeros(cap, dtype=np.int32)
        k = C.c_int()
        rc = self.lib.orc_jacobian_columns(C.byref(st.c), C.byref(tr.c), _ip(cols), cap, C.byref(k))
        assert rc == 0
        return cols[:k.value].copy()

    def interpolate(self, st, t, fej=True):
        R, p, H, dtj = np.zeros(9), np.zeros(3), np.zeros((4, 2, 3, 3)), np.zeros(6)
        start = C.c_int()
        rc = self.lib.orc_interpolate(C.byref(st.c), float(t), 1 if fej else 0, _dp(R), _dp(p), _dp(H), _dp(dtj),
                                      C.byref(start))
        if rc != 0:
            return None
        return R.reshape(3, 3), p, H, dtj, start.value

    def build_jacobians(self, st, tr, cols, ld):
        F, k = tr.c.n_feat, len(cols)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        rows = np.zeros(F, dtype=np.int32)
        Hf, Hx, res = np.zeros((F, 3, ld)), np.zeros((F, k, ld)), np.zeros((F, ld))
        rc = self.lib.orc_build_jacobians(C.byref(st.c), C.byref(tr.c), k, _ip(cols), ld, _ip(rows), _dp(Hf), _dp(Hx), _dp(res))
        assert rc == 0, rc
        return rows, Hf, Hx, res

    def triangulate(self, Rc, pc, uvn, min_dist=0.1, max_dist=60.0, max_cond=1e4, max_baseline=40.0, refine=True):
        Rc = np.ascontiguousarray(Rc, dtype=np.float64).reshape(-1, 9)
        pc = np.ascontiguousarray(pc, dtype=np.float64).reshape(-1, 3)
        uvn = np.ascontiguousarray(uvn, dtype=np.float32).reshape(-1, 2)
        out = np.zeros(3)
        ok = self.lib.orc_triangulate(len(Rc), _dp(Rc), _dp(pc), uvn.ctypes.data_as(fp), min_dist, max_dist, max_cond,
                                      max_baseline, 1 if refine else 0, _dp(out))
        return bool(ok), out


    def _tb(self, st, tr, min_dist=0.1, max_dist=60.0, max_cond=1e4, max_baseline=40.0, refine=True):
        opt = self.pkg.PlvTriOptions(min_dist, max_dist, max_cond, max_baseline, 1 if refine else 0)
        F = tr.c.n_feat
        p, ok, err = np.zeros((F, 3)), np.zeros(F, dtype=np.uint8), np.zeros(F)
        self.lib.orc_triangulate_batch(C.byref(st.c), C.byref(tr.c), C.byref(opt), _dp(p), ok.ctypes.data_as(u8p), _dp(err))
        return p, ok, err


JacOracle.triangulate_batch = JacOracle._tb

_jac = None


class PropOracle:
    """oracle/propagate_oracle.cpp"""

    def __init__(self, lib, pkg):
        self.lib, self.pkg = lib, pkg
        S, N, A, R = C.POINTER(pkg.PlvImuState), C.POINTER(pkg.PlvImuNoise), C.POINTER(pkg.PlvCpiAccum), C.POINTER(pkg.PlvCpiRecord)
        lib.orc_select_imu_readings.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int, dp, dp, dp, ip]
        lib.orc_select_imu_readings.restype = C.c_int
        lib.orc_reset_cpi.argtypes = [A, S, C.c_double]
        lib.orc_reset_cpi.restype = None
        lib.orc_propagate.argtypes = [S, N, C.c_int, dp, dp, dp, A, R, dp, C.c_int, C.c_int, C.c_int, dp, dp]
        lib.orc_propagate.restype = C.c_int
        lib.orc_cov_clone.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.orc_cov_clone.restype = None
        lib.orc_cpi_integrate.argtypes = [N, C.c_double, C.c_double, dp, dp, dp, dp, C.c_int, dp, dp, dp, R]
        lib.orc_cpi_integrate.restype = C.c_int
        lib.orc_select_wheel_data.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_int, dp, dp, dp, ip]
        lib.orc_select_wheel_data.restype = C.c_int
        lib.orc_wheel_linear_system.argtypes = [C.POINTER(pkg.PlvWheelOptions), C.POINTER(pkg.PlvWheelState), C.c_int, dp, dp, dp, dp, dp, dp,
                                                ip, dp, dp]
        lib.orc_wheel_linear_system.restype = C.c_int

    def select_imu_readings(self, t, wm, am, time0, time1):
        t, wm, am = (np.ascontiguousarray(x, dtype=np.float64) for x in (t, wm, am))
        cap = len(t) + 2
        ot, ow, oa = np.zeros(cap), np.zeros((cap, 3)), np.zeros((cap, 3))
        n = C.c_int()
        ok = self.lib.orc_select_imu_readings(len(t), _dp(t), _dp(wm), _dp(am), time0, time1, cap, _dp(ot), _dp(ow), _dp(oa), C.byref(n))
        return bool(ok), ot[:n.value].copy(), ow[:n.value].copy(), oa[:n.value].copy()

    def reset_cpi(self, imu, clone_t):
        acc = self.pkg.PlvCpiAccum()
        self.lib.orc_reset_cpi(C.byref(acc), C.byref(imu), clone_t)
        return acc

    def propagate(self, imu, noise, t, wm, am, P=None, acc=None, imu_id=0):
        """imu / acc updated in place; returns Phi, Qd, records, P' (col-major semantics: symmetric anyway)."""
        t, wm, am = (np.ascontiguousarray(x, dtype=np.float64) for x in (t, wm, am))
        rec = (self.pkg.PlvCpiRecord * max(len(t) - 1, 1))() if acc is not None else None
        Phi, Qd = np.zeros((15, 15)), np.zeros((15, 15))
        Pn = np.asfortranarray(P, dtype=np.float64).copy(order="F") if P is not None else None
        n = Pn.shape[0] if Pn is not None else 0
        rc = self.lib.orc_propagate(C.byref(imu), C.byref(noise), len(t), _dp(t), _dp(wm), _dp(am), C.byref(acc) if acc is not None else None,
                                    rec, _dp(Pn), n, n, imu_id, _dp(Phi), _dp(Qd))
        assert rc == 0
        return Phi, Qd, (list(rec)[:len(t) - 1] if rec is not None else []), Pn

    def select_wheel_data(self, t, m1, m2, time0, time1):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        t, m1, m2 = f(t), f(m1), f(m2)
        cap = len(t) + 4
        ot, o1, o2 = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        n = C.c_int()
        ok = self.lib.orc_select_wheel_data(len(t), _dp(t), _dp(m1), _dp(m2), time0, time1, cap, _dp(ot), _dp(o1), _dp(o2), C.byref(n))
        m = n.value if ok else 0
        return bool(ok), ot[:m].copy(), o1[:m].copy(), o2[:m].copy()

    def wheel_linear_system(self, opt, st, t, m1, m2):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        t, m1, m2 = f(t), f(m1), f(m2)
        H, res, Cov, cols = np.zeros(22 * 6), np.zeros(6), np.zeros(36), np.zeros(22, dtype=np.int32)
        R, p = np.zeros((3, 3)), np.zeros(3)
        k = self.lib.orc_wheel_linear_system(C.byref(opt), C.byref(st), len(t), _dp(t), _dp(m1), _dp(m2), _dp(H), _dp(res), _dp(Cov), _ip(cols),
                                             _dp(R), _dp(p))
        r = 3 if opt.type >= 3 else 6
        return H[:r * k].reshape(k, r).T.copy(), res[:r].copy(), Cov[:r * r].reshape(r, r).copy(), cols[:k].copy(), R, p

    def cpi_integrate(self, noise, t_given, clone_t, R_clone, v_clone, bg, ba, t, wm, am):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        t, wm, am, Rc, vc, bg, ba = f(t), f(wm), f(am), f(R_clone), f(v_clone), f(bg), f(ba)
        rec = self.pkg.PlvCpiRecord()
        ok = self.lib.orc_cpi_integrate(C.byref(noise), t_given, clone_t, _dp(Rc), _dp(vc), _dp(bg), _dp(ba), len(t), _dp(t), _dp(wm),
                                        _dp(am), C.byref(rec))
        return bool(ok), rec

    def cov_clone(self, P, src_id, size=6):
        n = P.shape[0]
        out = np.zeros((n + size, n + size), order="F")
        out[:n, :n] = P
        self.lib.orc_cov_clone(_dp(out), n, n + size, src_id, size)
        return out


_prop = None


def load_prop(pkg):
    global _prop
    if _prop is None:
        load()
        _prop = PropOracle(_inst.lib, pkg)
    return _prop


def load_jac(pkg):
    global _jac
    if _jac is None:
        load()
        _jac = JacOracle(_inst.lib, pkg)
    return _jac


# ------------------------------------------------------------------ line front-end oracle (oracle/line_oracle.cpp)
class LineOracle:
    def __init__(self, lib):
        self.lib = L = lib
        u64p = C.POINTER(C.c_uint64)
        L.orc_resize_half.argtypes = [u8, C.c_int, C.c_int, u8]
        L.orc_canny.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_int, u8]
        L.orc_fld.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, fp, C.c_int, u8]
        L.orc_fld.restype = C.c_int
        L.orc_detect_lines.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, fp, C.c_int]
        L.orc_detect_lines.restype = C.c_int
        L.orc_point_line_distance.argtypes = [fp, C.c_float, C.c_float]
        L.orc_point_line_distance.restype = C.c_float
        L.orc_assign_points_to_lines.argtypes = [fp, C.c_int, fp, u64p, C.c_int, ip, ip, u64p, dp, ip, fp]
        L.orc_assign_points_to_lines.restype = C.c_int
        L.orc_line_match.argtypes = [fp, C.c_int, ip, u64p, fp, C.c_int, ip, u64p, ip]
        L.orc_line_classification.argtypes = [fp, dp]
        L.orc_line_classification.restype = C.c_int
        L.orc_vanishing_points.argtypes = [dp, dp, dp]

    def resize_half(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros((h // 2, w // 2), dtype=np.uint8)
        self.lib.orc_resize_half(img.ctypes.data_as(u8), w, h, out.ctypes.data_as(u8))
        return out

    def canny(self, img, low=50, high=50):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros((h, w), dtype=np.uint8)
        self.lib.orc_canny(img.ctypes.data_as(u8), w, h, low, high, out.ctypes.data_as(u8))
        return out

    def fld(self, img, length=20, dist=1.414213562, c1=50, c2=50, cap=4096, want_edges=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        segs = np.zeros((cap, 4), dtype=np.float32)
        edges = np.zeros((h, w), dtype=np.uint8) if want_edges else None
        n = self.lib.orc_fld(img.ctypes.data_as(u8), w, h, length, dist, c1, c2, segs.ctypes.data_as(fp), cap,
                             edges.ctypes.data_as(u8) if want_edges else None)
        assert n <= cap
        return (segs[:n].copy(), edges) if want_edges else segs[:n].copy()

    def detect_lines(self, img, length=20, dist=1.414213562, c1=50, c2=50, min_len=40.0, cap=4096):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        lines = np.zeros((cap, 4), dtype=np.float32)
        n = self.lib.orc_detect_lines(img.ctypes.data_as(u8), w, h, length, dist, c1, c2, min_len, lines.ctypes.data_as(fp), cap)
        assert n <= cap
        return lines[:n].copy()

    def point_line_distance(self, line, x, y):
        line = np.ascontiguousarray(line, dtype=np.float32)
        return self.lib.orc_point_line_distance(line.ctypes.data_as(fp), float(x), float(y))

    def assign_points_to_lines(self, lines, pts, ids):
        lines = np.ascontiguousarray(lines, dtype=np.float32).reshape(-1, 4)
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        nl, npt = len(lines), len(pts)
        kept = np.zeros(nl, dtype=np.int32)
        rel_ptr, pos_ptr = np.zeros(nl + 1, dtype=np.int32), np.zeros(nl + 1, dtype=np.int32)
        rel_id, rel_d = np.zeros(nl * npt + 1, dtype=np.uint64), np.zeros(nl * npt + 1)
        pos = np.zeros((nl * npt + 1, 2), dtype=np.float32)
        u64p = C.POINTER(C.c_uint64)
        nk = self.lib.orc_assign_points_to_lines(lines.ctypes.data_as(fp), nl, pts.ctypes.data_as(fp), ids.ctypes.data_as(u64p), npt,
                                                 _ip(kept), _ip(rel_ptr), rel_id.ctypes.data_as(u64p), _dp(rel_d), _ip(pos_ptr),
                                                 pos.ctypes.data_as(fp))
        nr, npos = rel_ptr[nk], pos_ptr[nk]
        return dict(kept=kept[:nk].copy(), rel_ptr=rel_ptr[:nk + 1].copy(), rel_id=rel_id[:nr].copy(), rel_dist=rel_d[:nr].copy(),
                    pos_ptr=pos_ptr[:nk + 1].copy(), pos=pos[:npos].copy())

    def line_match(self, lines_new, rel_ptr_new, rel_id_new, lines_last, rel_ptr_last, rel_id_last):
        ln = np.ascontiguousarray(lines_new, dtype=np.float32).reshape(-1, 4)
        ll = np.ascontiguousarray(lines_last, dtype=np.float32).reshape(-1, 4)
        rpn, rpl = np.ascontiguousarray(rel_ptr_new, dtype=np.int32), np.ascontiguousarray(rel_ptr_last, dtype=np.int32)
        rin, ril = np.ascontiguousarray(rel_id_new, dtype=np.uint64), np.ascontiguousarray(rel_id_last, dtype=np.uint64)
        out = np.zeros(max(1, len(ln)), dtype=np.int32)
        u64p = C.POINTER(C.c_uint64)
        self.lib.orc_line_match(ln.ctypes.data_as(fp), len(ln), _ip(rpn), rin.ctypes.data_as(u64p), ll.ctypes.data_as(fp), len(ll),
                                _ip(rpl), ril.ctypes.data_as(u64p), _ip(out))
        return out[:len(ln)].copy()

    def line_classification(self, line, vps):
        line = np.ascontiguousarray(line, dtype=np.float32)
        vps = np.ascontiguousarray(vps, dtype=np.float64)
        return self.lib.orc_line_classification(line.ctypes.data_as(fp), _dp(vps))

    def vanishing_points(self, R_ItoC, K8):
        R = np.ascontiguousarray(R_ItoC, dtype=np.float64)
        K = np.ascontiguousarray(K8, dtype=np.float64)
        out = np.zeros((3, 2))
        self.lib.orc_vanishing_points(_dp(R), _dp(K), _dp(out))
        return out


_line = None


def load_line():
    global _line
    if _line is None:
        load()
        _line = LineOracle(_inst.lib)
    return _line


# ------------------------------------------------------------------ the compiled CPU frame (oracle/frame_oracle.cpp)
class FrameOracle:
    """One camera of the reference on the CPU, compiled end to end: tracker + databases + try_update (oracle/frame_oracle.cpp).  The
    covariance stays with the caller (a Fortran-ordered numpy array handed to every update call, modified in place)."""

    def __init__(self, pkg, cfg, q95):
        load()
        self.pkg, self.cfg = pkg, cfg
        self.lib = L = C.CDLL(FRAME_LIB) if FRAME_LIB else _inst.lib
        vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
        L.orc_frame_create.restype = vp
        L.orc_frame_create.argtypes = [vp, dp, C.c_int]
        L.orc_frame_destroy.argtypes = [vp]
        L.orc_frame_set_intrinsics.argtypes = [vp, dp]
        L.orc_frame_set_threads.argtypes = [vp, C.c_int]
        L.orc_frame_tracker_feed.argtypes = [vp, C.c_double, vp, C.c_int, vp]
        L.orc_frame_line_feed.argtypes = [vp, C.c_double, dp]
        L.orc_frame_update_points.argtypes = [vp, dp, C.c_int, C.c_int, vp, vp, dp, vp, u64p, u8p, dp]
        L.orc_frame_get_line_features.argtypes = [vp, vp, vp]
        L.orc_frame_last_point_decisions.argtypes = [vp, u64p, dp, C.c_int, ip]
        L.orc_frame_last_line_decisions.argtypes = [vp, u64p, dp, C.c_int, ip]
        L.orc_frame_update_lines.argtypes = [vp, dp, C.c_int, C.c_int, vp, vp, dp, vp, u64p, u8p, dp, C.c_int]
        L.orc_frame_try_update.argtypes = [vp, dp, C.c_int, C.c_int, vp, vp]
        L.orc_frame_camera_frame.argtypes = [vp, dp, C.c_int, C.c_int, vp, vp, dp]
        L.orc_frame_tracker_last.argtypes = [vp, fp, u64p, C.c_int]
        L.orc_frame_line_last.argtypes = [vp, fp, u64p, C.c_int]
        L.orc_frame_db_ids.argtypes = [vp, u64p, ip, C.c_int]
        L.orc_frame_line_db_ids.argtypes = [vp, u64p, ip, C.c_int]
        L.orc_frame_db_track.argtypes = [vp, C.c_uint64, dp, fp, fp, C.c_int]
        L.orc_frame_line_db_track.argtypes = [vp, C.c_uint64, dp, fp, fp, C.c_int, ip, ip]
        L.orc_frame_db_append.argtypes = [vp, C.c_uint64, C.c_int, dp, fp, fp]
        L.orc_frame_line_db_append.argtypes = [vp, C.c_uint64, C.c_int, dp, fp, fp, C.c_int, ip, C.c_int]
        L.orc_frame_used_insert.argtypes = [vp, C.c_uint64, dp, C.c_double]
        L.orc_frame_db_cleanup_measurements.argtypes = [vp, C.c_double]
        for f in ("orc_frame_db_size", "orc_frame_line_db_size", "orc_frame_used_size", "orc_frame_lines_detected", "orc_frame_lk_points"):
            getattr(L, f).argtypes = [vp]
        L.orc_frame_lk_points.restype = C.c_longlong
        self.q95 = np.ascontiguousarray(q95, dtype=np.float64)
        self.h = L.orc_frame_create(C.addressof(cfg), _dp(self.q95), len(self.q95))
        assert self.h
        self.timing_ms = np.zeros(6)       # accumulated by camera_frame: feed points, feed lines, points update, get lines, lines update, whole

    def close(self):
        if self.h:
            self.lib.orc_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_intrinsics(self, K8):
        K8 = np.ascontiguousarray(K8, dtype=np.float64)
        self.lib.orc_frame_set_intrinsics(self.h, _dp(K8))

    def set_threads(self, n):
        self.lib.orc_frame_set_threads(self.h, int(n))

    def tracker_feed(self, t, img, mask=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        rc = self.lib.orc_frame_tracker_feed(self.h, float(t), img.ctypes.data, img.shape[1], m.ctypes.data if m is not None else None)
        assert rc == 0, rc

    def line_feed(self, t, vps):
        vps = np.ascontiguousarray(vps, dtype=np.float64)
        rc = self.lib.orc_frame_line_feed(self.h, float(t), _dp(vps))
        assert rc == 0, rc

    def _P(self, P):
        assert P.flags.f_contiguous and P.dtype == np.float64 and P.shape[0] == P.shape[1]
        return _dp(P), P.shape[0], P.shape[0]

    def update_points(self, P, st, max_msckf, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, min_dist=0.1, max_dist=60.0,
                      max_cond=1e4, max_baseline=40.0, refine=True):
        pkg = self.pkg
        opt = pkg.PlvUpdateOptions(max_msckf, max_obs, chi2_mult, pkg.PlvTriOptions(min_dist, max_dist, max_cond, max_baseline, 1 if refine else 0),
                                   t_prev_frame, state_time, 1 if window_full else 0, 0, 0, None, 10, None)
        res = pkg.PlvUpdateResult()
        n = P.shape[0]
        dx, ids, acc, p = np.zeros(n), np.zeros(max_msckf, dtype=np.uint64), np.zeros(max_msckf, dtype=np.uint8), np.zeros((max_msckf, 3))
        rc = self.lib.orc_frame_update_points(self.h, *self._P(P), C.addressof(st.c), C.addressof(opt), _dp(dx), C.addressof(res),
                                              ids.ctypes.data_as(C.POINTER(C.c_uint64)), acc.ctypes.data_as(u8p), _dp(p))
        assert rc == 0, rc
        m = res.n_msckf
        return dict(dx=dx, n_pool=res.n_pool, n_msckf=m, n_accepted=res.n_accepted, n_rows=res.n_rows, n_returned=res.n_returned, status=res.status,
                    ids=ids[:m].copy(), accepted=acc[:m].copy(), p_FinG=p[:m].copy(), n_slam=0, n_init=0, n_truncated=res.n_truncated)

    def last_point_decisions(self):
        """(ids [n], values [n][11]) of the last point update's pool: the library's Context.last_point_decisions for the oracle"""
        n = C.c_int(0)
        self.lib.orc_frame_last_point_decisions(self.h, None, None, 0, C.byref(n))
        ids, vals = np.zeros(n.value, dtype=np.uint64), np.zeros((n.value, 11))
        if n.value:
            rc = self.lib.orc_frame_last_point_decisions(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(vals), n.value, C.byref(n))
            assert rc == 0, rc
        return ids, vals

    def last_line_decisions(self):
        """(ids [n], values [n][3] = chi2, threshold, residual norm) of the last line update's batch"""
        n = C.c_int(0)
        self.lib.orc_frame_last_line_decisions(self.h, None, None, 0, C.byref(n))
        ids, vals = np.zeros(n.value, dtype=np.uint64), np.zeros((n.value, 3))
        if n.value:
            rc = self.lib.orc_frame_last_line_decisions(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(vals), n.value, C.byref(n))
            assert rc == 0, rc
        return ids, vals

    def _line_opt(self, max_obs, t_prev_frame, state_time, window_full, chi2_mult):
        pkg = self.pkg
        return pkg.PlvUpdateOptions(0, max_obs, chi2_mult, pkg.PlvTriOptions(0, 0, 0, 0, 0), t_prev_frame, state_time, 1 if window_full else 0, 0, 0,
                                    None, 10, None)

    def get_line_features(self, st, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0):
        opt = self._line_opt(max_obs, t_prev_frame, state_time, window_full, chi2_mult)
        rc = self.lib.orc_frame_get_line_features(self.h, C.addressof(st.c), C.addressof(opt))
        assert rc == 0, rc

    def update_lines(self, P, st, max_obs, t_prev_frame, state_time, window_full=True, chi2_mult=1.0, cap=512):
        opt = self._line_opt(max_obs, t_prev_frame, state_time, window_full, chi2_mult)
        res = self.pkg.PlvUpdateResult()
        n = P.shape[0]
        dx, ids, acc, lg = np.zeros(n), np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint8), np.zeros((cap, 6))
        rc = self.lib.orc_frame_update_lines(self.h, *self._P(P), C.addressof(st.c), C.addressof(opt), _dp(dx), C.addressof(res),
                                             ids.ctypes.data_as(C.POINTER(C.c_uint64)), acc.ctypes.data_as(u8p), _dp(lg), cap)
        assert rc == 0, rc
        m = res.n_msckf
        return dict(dx=dx, n_pool=res.n_pool, n_lines=m, n_accepted=res.n_accepted, n_rows=res.n_rows, n_returned=res.n_returned, status=res.status,
                    ids=ids[:m].copy(), accepted=acc[:m].copy(), line_FinG=lg[:m].copy())

    def camera_frame(self, P, st, timestamp, img, mask=None, use_lines=False, update=None):
        """orc_frame_camera_frame with the structures of Context.camera_frame; update = the argument dict of Context._try_update_io"""
        pkg = self.pkg
        img = np.ascontiguousarray(img, dtype=np.uint8)
        m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
        io, results = pkg.Context._try_update_io(None, **update) if update is not None else (None, None)
        f = pkg.PlvCameraFrameIo(float(timestamp), -1, img.ctypes.data, img.shape[1], m.ctypes.data if m is not None else None, 1 if use_lines else 0,
                                 C.addressof(io) if io is not None else None, 0)
        if P is None:
            assert update is None
            rc = self.lib.orc_frame_camera_frame(self.h, None, 0, 0, C.addressof(st.c), C.addressof(f), _dp(self.timing_ms))
        else:
            rc = self.lib.orc_frame_camera_frame(self.h, *self._P(P), C.addressof(st.c), C.addressof(f), _dp(self.timing_ms))
        assert rc == 0, rc
        if results is None:
            return None, None, f.line_db_size
        pts, lns, _ = results()
        return pts, lns, f.line_db_size

    def try_update(self, P, st, update):
        io, results = self.pkg.Context._try_update_io(None, **update)
        rc = self.lib.orc_frame_try_update(self.h, *self._P(P), C.addressof(st.c), C.addressof(io))
        assert rc == 0, rc
        return results()

    # ---- inspection
    def tracker_last(self):
        n = self.lib.orc_frame_tracker_last(self.h, None, None, 0)
        pts, ids = np.zeros((n, 2), dtype=np.float32), np.zeros(n, dtype=np.uint64)
        if n:
            self.lib.orc_frame_tracker_last(self.h, pts.ctypes.data_as(fp), ids.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return pts, ids

    def line_last(self):
        n = self.lib.orc_frame_line_last(self.h, None, None, 0)
        ln, ids = np.zeros((n, 4), dtype=np.float32), np.zeros(n, dtype=np.uint64)
        if n:
            self.lib.orc_frame_line_last(self.h, ln.ctypes.data_as(fp), ids.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return ln, ids

    def db_size(self):
        return self.lib.orc_frame_db_size(self.h)

    def line_db_size(self):
        return self.lib.orc_frame_line_db_size(self.h)

    def used_size(self):
        return self.lib.orc_frame_used_size(self.h)

    def lines_detected(self):
        return self.lib.orc_frame_lines_detected(self.h)

    def lk_points(self):
        """points handed to perform_matching so far (all frames)"""
        return self.lib.orc_frame_lk_points(self.h)

    def db_ids(self, lines=False):
        fn = self.lib.orc_frame_line_db_ids if lines else self.lib.orc_frame_db_ids
        n = fn(self.h, None, None, 0)
        ids, cnt = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.int32)
        if n:
            fn(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), _ip(cnt), n)
        return ids, cnt

    def db_track(self, fid):
        m = self.lib.orc_frame_db_track(self.h, int(fid), None, None, None, 0)
        if m < 0:
            return None
        t, uv, uvn = np.zeros(m), np.zeros((m, 2), dtype=np.float32), np.zeros((m, 2), dtype=np.float32)
        self.lib.orc_frame_db_track(self.h, int(fid), _dp(t), uv.ctypes.data_as(fp), uvn.ctypes.data_as(fp), m)
        return t, uv, uvn

    def line_db_track(self, lid):
        D, npt = C.c_int(), C.c_int()
        m = self.lib.orc_frame_line_db_track(self.h, int(lid), None, None, None, 0, C.byref(D), C.byref(npt))
        if m < 0:
            return None
        t, uv, uvn = np.zeros(m), np.zeros((m, 4), dtype=np.float32), np.zeros((m, 4), dtype=np.float32)
        self.lib.orc_frame_line_db_track(self.h, int(lid), _dp(t), uv.ctypes.data_as(fp), uvn.ctypes.data_as(fp), m, C.byref(D), C.byref(npt))
        return t, uv, uvn, D.value, npt.value

    def db_append(self, fid, t, uv, uvn):
        t = np.ascontiguousarray(t, dtype=np.float64)
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        uvn = np.ascontiguousarray(uvn, dtype=np.float32).reshape(-1, 2)
        self.lib.orc_frame_db_append(self.h, int(fid), len(t), _dp(t), uv.ctypes.data_as(fp), uvn.ctypes.data_as(fp))

    def line_db_append(self, lid, t, uv, uvn, D=0, point_ids=()):
        t = np.ascontiguousarray(t, dtype=np.float64)
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 4)
        uvn = np.ascontiguousarray(uvn, dtype=np.float32).reshape(-1, 4)
        pid = np.ascontiguousarray(point_ids, dtype=np.int32)
        self.lib.orc_frame_line_db_append(self.h, int(lid), len(t), _dp(t), uv.ctypes.data_as(fp), uvn.ctypes.data_as(fp), int(D),
                                          _ip(pid) if len(pid) else None, len(pid))

    def used_insert(self, fid, p, newest):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.lib.orc_frame_used_insert(self.h, int(fid), _dp(p), float(newest))

    def db_cleanup_measurements(self, t):
        self.lib.orc_frame_db_cleanup_measurements(self.h, float(t))
