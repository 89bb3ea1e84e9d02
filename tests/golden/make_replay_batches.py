"""Generates tests/golden/replay_batches.npz: the [Hf | Hx | res] batches (+ covariance, column map, gate parameters) that the CPU
oracle's filter hands to UpdaterCamera::msckf_update / lines_update while it replays a rendered drive, each with the oracle's own
result (Givens null space + Givens compression + EKFUpdate: accepted set, dx, P').  These are Jacobians a running filter produces —
FEJ linearisation points, the gauge directions an MSCKF Jacobian cannot observe, calibration columns, ragged tracks — not i.i.d.
Gaussian matrices.  tests/test_oracle_update.py pins the oracle to them, tests/test_gpu_update_hard.py runs the library on them.

    python tests/golden/make_replay_batches.py          (CPU only, ~1 min)
"""
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge      # noqa: E402
import oracle_context as oc       # noqa: E402
import synth_dataset as sd        # noqa: E402

N_POINT, N_LINE = 20, 6


def main():
    pkg = ge.load_pkg()
    system = importlib.import_module("plviwo_amd.system")
    options = importlib.import_module("plviwo_amd.options")
    sd.set_camera(752, 480)
    hz, n = 10, 70
    sim = sd.simulate(seconds=n / hz + 0.2, cam_hz=hz, style="street")
    tc = sim["cam_times"][:n]
    imgs = sd.render_frames(tc, "street", 8)
    t, wm, am = sim["imu"]
    tw, m1, m2 = sim["wheel"]
    msgs = [(x, 0, i) for i, x in enumerate(t)] + [(x, 1, i) for i, x in enumerate(tw)] + [(x, 2, i) for i, x in enumerate(tc)]
    msgs.sort(key=lambda m: (m[0], m[1]))
    d = tempfile.mkdtemp()
    op = options.load_options(sd.write_config(d, d, d + "/traj.txt", clone_freq=hz, n_pts=120, max_msckf=14, calib_int=True, sigma_px=1.5))
    op.est.cam.use_lines = True
    sm = system.SystemManager(op, max_obs=12, context_factory=oc.PyMirrorContext, iw_initializer_factory=oc.OracleIwInitializer)
    sm.one_call_frame = sm.one_call_update = False
    cap = []
    orc = sm.ctx.o
    inner = orc.msckf_update

    def record(P, rows, Hf, Hx, res, cols, sigma2, q95, chi2_mult=1.0, res_norm_gate=3.0):
        out = inner(P, rows, Hf, Hx, res, cols, sigma2, q95, chi2_mult=chi2_mult, res_norm_gate=res_norm_gate)
        rc, P2, dx, acc, nrows = out
        if rc == 0 and int(np.sum(acc)) > 0:
            cap.append(dict(P=np.array(P), rows=np.array(rows), Hf=np.array(Hf), Hx=np.array(Hx), res=np.array(res), cols=np.array(cols),
                            sigma2=float(sigma2), chi2_mult=float(chi2_mult), gate=float(res_norm_gate), P_new=np.array(P2), dx=np.array(dx),
                            accepted=np.array(acc), n_rows=int(nrows)))
        return out

    sm.ctx.o.msckf_update = record
    sm.ctx.mir.o.msckf_update = record
    for tt, kind, i in msgs:
        if kind == 0:
            r = np.concatenate([[t[i]], wm[i], am[i]])
            sm.feed_measurement_imu(r[0], r[1:4], r[4:7])
        elif kind == 1:
            sm.feed_measurement_wheel(tw[i], m1[i], m2[i])
        else:
            sm.feed_measurement_camera(tt, imgs[i])
    pts = [c for c in cap if c["Hf"].shape[1] == 3]
    lns = [c for c in cap if c["Hf"].shape[1] == 6]
    print(f"captured {len(pts)} point updates, {len(lns)} line updates")
    # the later point updates (full window) and every line update that accepted something
    chosen = pts[-N_POINT:] + lns[-N_LINE:]
    out = {"count": np.array(len(chosen))}
    for j, c in enumerate(chosen):
        for key, v in c.items():
            out[f"b{j}_{key}"] = np.asarray(v)
    path = os.path.join(HERE, "replay_batches.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB;", "k =", sorted({len(c["cols"]) for c in chosen}), "F =", [c["Hf"].shape[0] for c in chosen])
    conds = []
    for c in chosen:
        # condition number of the accepted, projected, stacked Jacobian over its non-null columns (what the compression factors)
        F, fdim, ld = c["Hf"].shape
        rows_all = []
        for f in range(F):
            if not c["accepted"][f]:
                continue
            r = int(c["rows"][f])
            A = c["Hf"][f, :, :r].T
            q, _ = np.linalg.qr(A, mode="complete")
            rows_all.append(q[:, fdim:].T @ c["Hx"][f, :, :r].T)
        H = np.vstack(rows_all)
        s = np.linalg.svd(H / np.maximum(np.linalg.norm(H, axis=0), 1e-300), compute_uv=False)
        conds.append((s[0] / s[s > s[0] * 1e-13][-1], int((s <= s[0] * 1e-13).sum())))
    print("cond of the column-equilibrated stacked Jacobian (over its numerical range), null columns:", [(f"{a:.1e}", b) for a, b in conds])


if __name__ == "__main__":
    main()
