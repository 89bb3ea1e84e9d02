"""CPU restatement (numpy) of the reference's state initialisers.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing
in the product); parity unpinned (the reference cannot be built here and ships no fixtures for this code).

Follows, function by function:
  REF: PL-VIWO/src/init/imu/I_Initializer.cpp:44-150                 (static IMU initialisation)
  REF: PL-VIWO/src/init/imu_wheel/IW_Initializer.cpp:44-690          (IMU-wheel initialisation)
  REF: PL-VIWO/src/state/Propagator.cpp:93-152,320-331,358-373       (select_imu_readings, interpolate_data, get_bounding_data)
  REF: PL-VIWO/src/update/wheel/UpdaterWheel.cpp:142-215,784-794     (select_wheel_data, interpolate_data)
"""
import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def quat_2_Rot(q):
    return (2 * q[3] ** 2 - 1) * np.eye(3) - 2 * q[3] * skew(q[:3]) + 2 * np.outer(q[:3], q[:3])


def rot_2_quat(R):
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


def Omega(w):
    M = np.zeros((4, 4))
    M[:3, :3], M[3, :3], M[:3, 3] = -skew(w), -w, w
    return M


def quatnorm(q):
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


# --------------------------------------------------------------------------------------------------------- buffers
def select_imu_readings(t, wm, am, time0, time1):
    """Propagator::select_imu_readings.  Returns None where it returns false, else (t, wm, am)."""
    n = len(t)
    if n < 2 or time1 <= time0 or t[0] > time0 or t[-1] < time1:
        return None
    out = []

    def interp(i, ts):
        lam = (ts - t[i]) / (t[i + 1] - t[i])
        out.append((ts, (1 - lam) * wm[i] + lam * wm[i + 1], (1 - lam) * am[i] + lam * am[i + 1]))

    i = 0
    while i < n - 1:
        if t[i] <= time0 <= t[i + 1]:
            interp(i, time0)
            break
        i += 1
    i = i - 1 if i != 0 else 0
    while i < n - 1:
        if time0 < t[i] and t[i + 1] < time1:
            out.append((t[i], wm[i], am[i]))
        if t[i + 1] > time1:
            break
        i += 1
    i = i - 1 if i != 0 else 0
    while i < n - 1:
        if t[i] <= time1 <= t[i + 1]:
            interp(i, time1)
            break
        i += 1
    return np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out])


def select_wheel_data(t, m1, m2, time0, time1):
    """UpdaterWheel::select_wheel_data.  Returns None where it returns false, else a list of (t, m1, m2)."""
    n = len(t)
    if n == 0 or t[-1] <= time1 or t[0] > time0:
        return None

    def interp(i, j, ts):
        lam = (ts - t[i]) / (t[j] - t[i])
        return (ts, (1 - lam) * m1[i] + lam * m1[j], (1 - lam) * m2[i] + lam * m2[j])

    out = []
    for i in range(n - 1):
        if t[i + 1] > time0 and t[i] < time0:
            out.append(interp(i, i + 1, time0))
            continue
        if t[i] >= time0 and t[i + 1] <= time1:
            out.append((t[i], m1[i], m2[i]))
            continue
        if t[i + 1] > time1:
            if t[i] > time1:
                out.append(interp(i - 1, i, time1))
            else:
                out.append((t[i], m1[i], m2[i]))
            if out[-1][0] != time1:
                out.append(interp(i, i + 1, time1))
            break
    if len(out) < 2:
        return None
    i = 0
    while i < len(out) - 1:
        if abs(out[i + 1][0] - out[i][0]) < 1e-12:
            del out[i]
            i -= 1
        i += 1
    if len(out) < 2:
        return None
    return out


# ------------------------------------------------------------------------------------------------ static IMU init
def imu_static_init(t, wm, am, window_time, imu_thresh, gravity):
    """I_Initializer::initialization.  Returns the 17-vector or None."""
    if len(t) < 2:
        return None
    newest, oldest = t[-1], t[0]
    if newest - oldest < 2 * window_time:
        return None
    w10 = [i for i in range(len(t)) if newest - window_time < t[i] <= newest]
    w21 = [i for i in range(len(t)) if newest - 2 * window_time < t[i] <= newest - window_time]
    if len(w10) < 2 or len(w21) < 2:
        return None
    a_avg10 = am[w10].sum(axis=0) / len(w10)
    a_var10 = np.sqrt(sum((am[i] - a_avg10) @ (am[i] - a_avg10) for i in w10) / (len(w10) - 1))
    a_avg21 = am[w21].sum(axis=0) / len(w21)
    w_avg21 = wm[w21].sum(axis=0) / len(w21)
    a_var21 = np.sqrt(sum((am[i] - a_avg21) @ (am[i] - a_avg21) for i in w21) / (len(w21) - 1))
    if a_var10 < imu_thresh or a_var21 > imu_thresh:
        return None
    z = a_avg21 / np.linalg.norm(a_avg21)
    e1 = np.array([1.0, 0, 0])
    x = e1 - z * (z @ e1)
    x = x / np.linalg.norm(x)
    y = skew(z) @ x
    Ro = np.column_stack([x, y, z])
    q = rot_2_quat(Ro)
    out = np.zeros(17)
    out[0] = t[w21[-1]]
    out[1:5] = q
    out[11:14] = w_avg21
    out[14:17] = a_avg21 - quat_2_Rot(q) @ np.asarray(gravity)
    return out


# ------------------------------------------------------------------------------------------------ IMU-wheel init
class IWInitializer:
    """IW_Initializer with its cnt_smooth / prev_init memory."""

    def __init__(self, wheel_type, intrinsics, R_ItoO, p_IinO, toff, threshold, gravity, imu_gravity_aligned):
        self.type, self.intr = wheel_type, np.asarray(intrinsics, dtype=float)
        self.R_OtoI, self.p_IinO = np.asarray(R_ItoO, dtype=float).reshape(3, 3).T, np.asarray(p_IinO, dtype=float)
        self.toff, self.threshold, self.gravity, self.aligned = toff, threshold, np.asarray(gravity, dtype=float), imu_gravity_aligned
        self.cnt_smooth, self.prev_init = -1, None
        self.last_init, self.last_mode = None, -1

    # IMU_prop_rk4
    @staticmethod
    def prop_rk4(dt, w1, w2):
        w = w1.copy()
        alpha = (w2 - w1) / dt
        dq0 = np.array([0, 0, 0, 1.0])
        k1 = 0.5 * Omega(w) @ dq0 * dt
        w = w + 0.5 * alpha * dt
        k2 = 0.5 * Omega(w) @ quatnorm(dq0 + 0.5 * k1) * dt
        k3 = 0.5 * Omega(w) @ quatnorm(dq0 + 0.5 * k2) * dt
        w = w + 0.5 * alpha * dt
        k4 = 0.5 * Omega(w) @ quatnorm(dq0 + k3) * dt
        return quatnorm(dq0 + k1 / 6 + k2 / 3 + k3 / 3 + k4 / 6)

    @staticmethod
    def gram_schmidt(g):
        z = g / np.linalg.norm(g)
        e1 = np.array([1.0, 0, 0])
        x = e1 - z * (z @ e1)
        x = x / np.linalg.norm(x)
        y = skew(z) @ x
        y = y / np.linalg.norm(y)
        return np.column_stack([x, y, z])

    def _intervals(self, imu, bg, wheel):
        """The accumulation every stage of the reference repeats: yields (i, sum_dt, sum_R_a_dt, sum_R_dt, v_ItinI0)."""
        t, wm, am = imu
        sum_dt, sum_R_a_dt, sum_R_dt = 0.0, np.zeros(3), np.zeros((3, 3))
        R_IktoI0, R_O0toOk = np.eye(3), np.eye(3)
        for i in range(1, len(wheel)):
            ts, te = wheel[i - 1][0] + self.toff, wheel[i][0] + self.toff
            w_Os, w_Oe, v_Oe = wheel[i - 1][1][:3], wheel[i][1][:3], wheel[i][1][3:]
            sel = select_imu_readings(t, wm, am, ts, te)
            if sel is None:
                raise AssertionError("select_imu_readings failed inside the initialiser")
            pt, pw, pa = sel
            for j in range(len(pt) - 1):
                dt = pt[j + 1] - pt[j]
                a_I = 0.5 * (pa[j] + pa[j + 1])
                sum_R_a_dt = sum_R_a_dt + R_IktoI0 @ a_I * dt
                sum_R_dt = sum_R_dt + R_IktoI0 * dt
                sum_dt += dt
                R_IktoI0 = R_IktoI0 @ quat_2_Rot(self.prop_rk4(dt, pw[j] - bg, pw[j + 1] - bg)).T
            R_O0toOk = quat_2_Rot(self.prop_rk4(te - ts, w_Os, w_Oe)) @ R_O0toOk
            v_It = self.R_OtoI @ R_O0toOk.T @ (v_Oe + skew(w_Oe) @ self.p_IinO)
            yield i, sum_dt, sum_R_a_dt.copy(), sum_R_dt.copy(), v_It

    def get_data(self, t, wm, am, tw, m1, m2):
        if len(t) < 3 or len(tw) < 3:
            return None
        min_t = max(t[1], tw[1] + self.toff)
        max_t = min(t[-2], tw[-2] + self.toff)
        imu = select_imu_readings(t, wm, am, min_t, max_t)
        if imu is None:
            return None
        raw = select_wheel_data(tw, m1, m2, min_t - self.toff, max_t - self.toff)
        if raw is None:
            return None
        if len(imu[0]) < 20 or len(raw) < 20:
            return None
        rl, rr, b = self.intr
        wheel = []
        for tt, a, c in raw:
            if self.type in ("Wheel2DAng", "Wheel3DAng"):
                wz, vx = (c * rr - a * rl) / b, (c * rr + a * rl) / 2
            elif self.type in ("Wheel2DLin", "Wheel3DLin"):
                wz, vx = (c - a) / b, (c + a) / 2
            else:
                wz, vx = a, c
            wheel.append((tt, np.array([0, 0, wz, vx, 0, 0.0])))
        return imu, wheel

    def init_bg(self, imu_sel, wheel):
        t, wm, _ = imu_sel
        bg, cnt = np.zeros(3), 0
        for tt, wv in wheel:
            tq = tt + self.toff
            if tq > t[-1] or tq < t[0]:
                continue
            for i in range(len(t) - 1):
                if t[i] <= tq < t[i + 1]:
                    lam = (tq - t[i]) / (t[i + 1] - t[i])
                    bg += (1 - lam) * wm[i] + lam * wm[i + 1] - self.R_OtoI @ wv[:3]
                    cnt += 1
                    break
        return bg / cnt if cnt >= 1 else None

    def init_gI_simple(self, imu, bg, v_I0, wheel):
        if self.aligned:
            return self.gravity.copy()
        g = np.zeros(3)
        for _, sdt, sRa, _, v_It in self._intervals(imu, bg, wheel):
            g += (v_I0 + sRa - v_It) / sdt
        g /= len(wheel) - 1
        return g / np.linalg.norm(g) * np.linalg.norm(self.gravity)

    @staticmethod
    def _llt_solve(A, B):
        """Eigen's LLT + solve, including its behaviour on a matrix that is not positive definite (stops at the first
        non-positive pivot, solves with what is there)."""
        A = np.array(A, dtype=float)
        n = A.shape[0]
        for k in range(n):
            x = A[k, k] - A[k, :k] @ A[k, :k]
            if x <= 0:
                break
            A[k, k] = x = np.sqrt(x)
            if k + 1 < n:
                A[k + 1:, k] = (A[k + 1:, k] - A[k + 1:, :k] @ A[k, :k]) / x
        L = np.tril(A)
        B = np.array(B, dtype=float)
        Y = np.zeros_like(B)
        for i in range(n):
            Y[i] = (B[i] - L[i, :i] @ Y[:i]) / L[i, i]
        X = np.zeros_like(B)
        for i in range(n - 1, -1, -1):
            X[i] = (Y[i] - L[i + 1:, i] @ X[i + 1:]) / L[i, i]
        return X

    def init_gI_dongsi(self, imu, bg, v_I0, wheel):
        if self.aligned:
            return self.gravity.copy()
        N = len(wheel)
        A, b = np.zeros((3 * N, 6)), np.zeros(3 * N)
        for i, sdt, sRa, sR, v_It in self._intervals(imu, bg, wheel):
            b[3 * i:3 * i + 3] = v_It - v_I0 - sRa
            A[3 * i:3 * i + 3, :3] = -sR
            A[3 * i:3 * i + 3, 3:] = -sdt * np.eye(3)
        A1, A2 = A[:, :3], A[:, 3:]
        A1A1_inv = self._llt_solve(A1.T @ A1, np.eye(3))
        Temp = A2.T @ (np.eye(3 * N) - A1 @ A1A1_inv @ A1.T)
        D, d = Temp @ A2, Temp @ b
        g = np.linalg.norm(self.gravity)
        # the constraint |(D - lam I)^-1 d| = g as a polynomial in lam: g^2 det(M)^2 - |adj(M) d|^2 with M = D - lam I (degree 6,
        # monic after scaling; compute_dongsi_coeff is a generated closed form of the same thing)
        P = np.poly1d
        M = [[P([-1.0, float(D[i, j])]) if i == j else P([float(D[i, j])]) for j in range(3)] for i in range(3)]

        def cof(i, j):
            r = [k for k in range(3) if k != i]
            c = [k for k in range(3) if k != j]
            return (M[r[0]][c[0]] * M[r[1]][c[1]] - M[r[0]][c[1]] * M[r[1]][c[0]]) * (-1) ** (i + j)

        adj = [[cof(j, i) for j in range(3)] for i in range(3)]
        det = M[0][0] * adj[0][0] + M[0][1] * adj[1][0] + M[0][2] * adj[2][0]
        poly = det * det * float(g * g)
        for i in range(3):
            row = adj[i][0] * float(d[0]) + adj[i][1] * float(d[1]) + adj[i][2] * float(d[2])
            poly = poly - row * row
        coeff = np.zeros(7)
        coeff[7 - len(poly.coeffs):] = poly.coeffs
        coeff = coeff / coeff[0]
        found, lam_min, cost_min = False, -1.0, np.inf
        for val in np.roots(coeff):
            if val.imag == 0:
                lam = val.real
                cost = abs(np.linalg.norm(self._llt_solve(D - lam * np.eye(3), d)) - g)
                if not found or cost < cost_min:
                    found, lam_min, cost_min = True, lam, cost
        if not found:
            return None
        gI = self._llt_solve(D - lam_min * np.eye(3), d)
        if abs(np.linalg.norm(gI) - g) > 1e-3:
            return None
        return gI

    def init_ba(self, imu, bg, v_I0, g_I0, wheel):
        ba = np.zeros(3)
        for _, sdt, sRa, sR, v_It in self._intervals(imu, bg, wheel):
            ba += np.linalg.inv(sR) @ (v_I0 + sRa - sdt * g_I0 - v_It)
        return ba / (len(wheel) - 1)

    def residual(self, imu, bg, ba, v_I0, g_I0, wheel):
        res = np.zeros(3 * len(wheel))
        for i, sdt, sRa, sR, v_It in self._intervals(imu, bg, wheel):
            res[3 * i:3 * i + 3] = v_It - v_I0 - sRa + sR @ ba + sdt * g_I0
        return res

    def initialization(self, t, wm, am, tw, m1, m2):
        """Returns the 17-vector on success, else None; self.last_init holds this call's 12-vector when the stages passed."""
        self.last_init, self.last_mode = None, -1
        data = self.get_data(t, wm, am, tw, m1, m2)
        if data is None:
            return None
        imu_sel, wheel = data
        imu = (t, wm, am)
        static = all(np.linalg.norm(w[1]) == 0 for w in wheel)
        self.last_mode = 0 if static else 1
        bg = self.init_bg(imu_sel, wheel)
        if bg is None:
            return None
        v_I0 = self.R_OtoI @ (wheel[0][1][3:] + skew(wheel[0][1][:3]) @ self.p_IinO)
        g_I0 = self.init_gI_simple(imu, bg, v_I0, wheel) if static else self.init_gI_dongsi(imu, bg, v_I0, wheel)
        if g_I0 is None:
            return None
        ba = self.init_ba(imu, bg, v_I0, g_I0, wheel)
        if not static and np.linalg.norm(ba) > np.linalg.norm(self.gravity):
            return None
        res = self.residual(imu, bg, ba, v_I0, g_I0, wheel)
        if np.linalg.norm(res[-3:]) / 3 > self.threshold * 100:
            self.cnt_smooth = 0
            return None
        init = np.concatenate([bg, ba, g_I0, v_I0])
        self.last_init = init
        if self.cnt_smooth < 0:
            self.cnt_smooth += 1
            self.prev_init = init
            return None
        if np.linalg.norm(self.prev_init - init) < self.threshold:
            self.cnt_smooth += 1
        else:
            self.cnt_smooth = 0
        R_GtoI0 = self.gram_schmidt(init[6:9])
        if self.cnt_smooth > 3:
            out = np.zeros(17)
            out[0] = wheel[0][0] + self.toff
            out[1:5] = rot_2_quat(R_GtoI0)
            out[8:11] = R_GtoI0.T @ init[9:12]
            out[11:14], out[14:17] = init[0:3], init[3:6]
            return out
        self.prev_init = init
        return None
