#!/usr/bin/env python3
"""plv_compress (the Householder compression of mode 1 / of a rejected whitened update) timed with hqr_kernel (default) and with the
tree of unblocked factorisations (knob 1 << 29), same process, alternating; and both against numpy's QR (R^T R and |R|)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
ctx = pkg.Context()
rng = np.random.default_rng(3)
for m, k in ((750, 104), (300, 104), (1500, 104), (900, 134), (1900, 149), (200, 104), (120, 104), (2040, 190)):
    H, r = rng.standard_normal((m, k)), rng.standard_normal(m)
    H[:, 5] = H[:, 4] * 2.0           # a dependent column
    H[: m // 2, 7] = 0.0
    Rn = np.linalg.qr(np.column_stack([H, r]), mode="r")
    Rn *= np.sign(np.diag(Rn))[:, None] + (np.diag(Rn) == 0)[:, None]
    out = []
    for name, knob in (("hqr", 0), ("tree", 1 << 29), ("hqr", 0), ("tree", 1 << 29)):
        pkg.debug_knobs(knob)
        try:
            R0, z0 = ctx.compress(H, r)
        except Exception as e:
            out.append("%s: %s" % (name, str(e)[:60]))
            continue
        ts = []
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(20):
            t0 = time.perf_counter(); ctx.compress(H, r); ts.append(time.perf_counter() - t0)
        tab = ctx.prof_table()
        ctx.prof_enable(False)
        kern = sum(v[1] for k_, v in tab.items() if "hqr" in k_ or "qr_accum" in k_) / 20.0 * 1e3
        e1 = np.abs(R0.T @ R0 - H.T @ H).max() / np.abs(H.T @ H).max()
        e2 = np.abs(np.abs(np.triu(R0)) - np.abs(Rn[:k, :k])).max() / np.abs(Rn).max()
        e3 = np.abs(np.abs(z0) - np.abs(Rn[:k, k])).max() / np.abs(Rn).max()
        low = np.abs(np.tril(R0, -1)).max()
        out.append("%s %.0f us, kernels %.0f us (gram %.0e, |R| %.0e, |z| %.0e, below diag %.0e, min diag %.1e)" % (name, np.median(ts) * 1e6, kern, e1, e2, e3, low, np.diag(R0).min()))
    print("%dx%d: " % (m, k) + " | ".join(out))
pkg.debug_knobs(0)
