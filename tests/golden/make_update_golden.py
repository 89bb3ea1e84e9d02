"""Generates tests/golden/update_small.npz: a small seeded MSCKF update problem with the expected
output of the oracle, cross-checked against an independent numpy information-form solution.
(The reference has no golden vectors and cannot be built/imported here — SURVEY.md §8(c).)
Run from the repo root:  python tests/golden/make_update_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

n, k = 45, 32
P = synth.spd_cov(n, seed=21)
cols = synth.col_map(n, k, seed=22)
rows, Hf, Hx, res = synth.msckf_batch(F=10, M=6, k=k, seed=23)
orc = oracle_lib.load()
rc, P2, dx, acc, nrows = orc.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table())
assert rc == 0 and acc.sum() > 0
# independent check
Ps = P[np.ix_(cols, cols)]
Hs, rs = [], []
for f in range(len(rows)):
    if not acc[f]:
        continue
    m = rows[f]
    Q, _ = np.linalg.qr(Hf[f, :, :m].T, mode="complete")
    N = Q[:, 3:]
    Hs.append(N.T @ Hx[f, :, :m].T)
    rs.append(N.T @ res[f, :m])
Hst, rst = np.vstack(Hs), np.concatenate(rs)
Hfull = np.zeros((Hst.shape[0], n))
Hfull[:, cols] = Hst
Pn = np.linalg.inv(np.linalg.inv(P) + Hfull.T @ Hfull)
assert np.allclose(P2, Pn, rtol=1e-6, atol=1e-14)
assert np.allclose(dx, Pn @ Hfull.T @ rst, rtol=1e-6, atol=1e-12)
np.savez_compressed(os.path.join(HERE, "update_small.npz"), P=P, cols=cols, rows=rows, Hf=Hf, Hx=Hx, res=res,
                    sigma2=2.25, accepted=acc, dx=dx, P_new=P2, n_rows=nrows)
print("wrote update_small.npz", acc.sum(), "accepted of", len(rows))
