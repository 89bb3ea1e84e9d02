"""ctypes binding of oracle/liboracle.so — the CPU restatement used ONLY as the checker."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

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


_inst = None


def load():
    global _inst
    if _inst is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        _inst = Oracle(C.CDLL(LIB))
    return _inst
