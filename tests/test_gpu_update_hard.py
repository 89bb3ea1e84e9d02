"""Parity of the SUBSTITUTED algorithms of the update path on inputs that are not benign: where the reference (and the oracle) run
Givens rotations on the stacked Jacobian and then the EKF step (StateHelper.cpp:602-672, :94-173), the library forms the information
matrix of the stacked rows and updates in coordinates whitened by a factor of the prior block (the default, "whitened" route; the
round-2 route — Gram matrix + Cholesky giving the same R as the Givens QR — stays selectable as mode 3), and its fused Jacobian
launches project with Householder reflections where the reference runs Givens (:616-651).  Cases:
  * batches a running filter produced (tests/golden/replay_batches.npz, made by tests/golden/make_replay_batches.py from the CPU
    oracle's replay of a rendered drive): FEJ linearisation points, unobservable gauge directions, calibration columns, ragged tracks;
  * column spaces of prescribed condition number 1e2 .. 1e8;
  * duplicated features and rows scaled by 1e+-3;
  * exact null directions (columns that cancel to rounding).
What is compared is what the filter keeps: the accepted set, dx and P' (both sides fp64).  Tolerances: the default route agrees with
the oracle to 1e-9 (P' and dx, relative to the largest entry) on every replay batch and for column spaces of condition 1e2 .. 1e8; the
Gram + Cholesky route (mode 3) holds 1e-8 up to condition 1e4 and 1e-7 at 1e6 on well-posed priors, loses dx to 3e-5 on replay batches
whose Gram matrix has pivots it cannot tell from zero (it reports them), and cannot resolve condition 1e8 at all (eps c^2 > 1): see
test_compression_reports_what_it_cannot_resolve."""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "replay_batches.npz")


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _batches():
    g = np.load(GOLD)
    for j in range(int(g["count"])):
        yield j, {k[len(f"b{j}_"):]: g[k] for k in g.files if k.startswith(f"b{j}_")}


@pytest.mark.parametrize("mode", [0, 1])
def test_replay_captured_batches(ctx, mode):
    """mode 0 = whitened update (the default), 1 = Householder throughout (plv_update_compression_mode; the Gram + Cholesky
    compression of rounds 2-3 — modes 2 and 3 — was removed in round 5)"""
    ctx.update_compression_mode(mode)
    try:
        n_checked = n_amb = n_redone = 0
        worst_dx = worst_P = worst_dx_clean = 0.0
        for j, b in _batches():
            rc, P1, dx1, acc, nr = ctx.msckf_update(b["P"], b["rows"], b["Hf"], b["Hx"], b["res"], b["cols"], float(b["sigma2"]), float(b["chi2_mult"]),
                                                    float(b["gate"]))
            _, route, amb = ctx.update_compression_mode()
            assert rc == 0, j
            assert np.array_equal(acc, b["accepted"]) and nr == int(b["n_rows"]), j
            e_dx, e_P = _rel(dx1, b["dx"]), _rel(P1, b["P_new"])
            worst_dx, worst_P = max(worst_dx, e_dx), max(worst_P, e_P)
            n_amb += amb > 0
            n_redone += route == 3
            assert np.array_equal(P1, P1.T)
            assert e_P < 1e-9, (j, e_P)
            if mode == 0:
                assert e_dx < 1e-9 and e_P < 1e-10, (j, e_dx, e_P)
                assert route in (0, 4) and amb == 0, (j, route)
            else:
                assert e_dx < 1e-8, (j, e_dx, route, amb)
                assert route in (0, 2), (j, route)
            n_checked += 1
        assert n_checked >= 20
        print(f"mode {mode}: {n_checked} replay batches, {n_amb} with ambiguous pivots, {n_redone} redone by Householder; worst relative difference dx "
              f"{worst_dx:.2e} (without ambiguous pivots {worst_dx_clean:.2e}), P {worst_P:.2e}")
    finally:
        ctx.update_compression_mode(0)


def test_replay_captured_batches_in_the_factor_form(ctx, pkg):
    """The whitened update's other form — what the prior factor picks once it meets a pivot below 1e-4, i.e. late in a drive
    (dense_kernels.hip "whitened update") — forced onto every captured batch (plv_debug_knobs 8192): same verdicts, dx and P' to the
    tolerances of the whitened form where the measurements are not far better than the prior; an update it hands to the Householder
    route (B's diagonal above 100: route 5) agrees like that route."""
    prev = pkg.debug_knobs(8192)
    ctx.update_compression_mode(0)
    try:
        routes = []
        for j, b in _batches():
            rc, P1, dx1, acc, nr = ctx.msckf_update(b["P"], b["rows"], b["Hf"], b["Hx"], b["res"], b["cols"], float(b["sigma2"]), float(b["chi2_mult"]),
                                                    float(b["gate"]))
            _, route, _ = ctx.update_compression_mode()
            routes.append(route)
            assert rc == 0 and np.array_equal(acc, b["accepted"]) and nr == int(b["n_rows"]), j
            assert route in (0, 4, 5), (j, route)
            assert _rel(dx1, b["dx"]) < 1e-8 and _rel(P1, b["P_new"]) < 1e-9 and np.array_equal(P1, P1.T), (j, route, _rel(dx1, b["dx"]), _rel(P1, b["P_new"]))
        assert routes.count(4) >= 5, routes        # (the form itself was exercised, not only its hand-over)
    finally:
        pkg.debug_knobs(prev)


def _conditioned(k, cond, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(k, k)))
    s = np.logspace(0, -np.log10(cond), k)
    return (Q * s) @ Q.T


def _truth(P, rows, Hf, Hx, res, cols, acc):
    """information-form update in extended precision on the accepted, projected rows: P' = (P^-1 + H^T H)^-1, dx = P' H^T r"""
    ld = np.longdouble
    n = P.shape[0]
    Hs, rs = [], []
    for f in range(len(rows)):
        if not acc[f]:
            continue
        m = int(rows[f])
        q, _ = np.linalg.qr(Hf[f, :, :m].T, mode="complete")
        N = q[:, Hf.shape[1]:]
        Hs.append(N.T @ Hx[f, :, :m].T)
        rs.append(N.T @ res[f, :m])
    H = np.zeros((sum(h.shape[0] for h in Hs), n))
    H[:, cols] = np.vstack(Hs)
    r = np.concatenate(rs)
    # K form in long double with a Cholesky solve written out (numpy's linalg has no extended precision)
    Hl, Pl, rl = H.astype(ld), P.astype(ld), r.astype(ld)
    S = Hl @ Pl @ Hl.T + np.eye(H.shape[0], dtype=ld)
    L = np.zeros_like(S)
    for i in range(S.shape[0]):
        for j in range(i + 1):
            v = S[i, j] - L[i, :j] @ L[j, :j]
            L[i, j] = np.sqrt(v) if i == j else v / L[j, j]
    M = Pl @ Hl.T

    def solve(B):   # S^-1 B
        Y = np.zeros_like(B)
        for i in range(L.shape[0]):
            Y[i] = (B[i] - L[i, :i] @ Y[:i]) / L[i, i]
        X = np.zeros_like(B)
        for i in range(L.shape[0] - 1, -1, -1):
            X[i] = (Y[i] - L[i + 1:, i] @ X[i + 1:]) / L[i, i]
        return X
    KT = solve(M.T)
    return (Pl - M @ KT).astype(np.float64), (KT.T @ rl).astype(np.float64)


@pytest.mark.parametrize("mode,cond,tol", [(0, 1e2, 1e-9), (0, 1e4, 1e-9), (0, 1e6, 1e-9), (0, 1e8, 1e-9), (1, 1e4, 1e-9), (1, 1e8, 1e-9)])
def test_msckf_update_conditioned_columns(ctx, oracle, mode, cond, tol):
    """every feature's Jacobian mixed through one k x k matrix of the given condition number: the stacked Jacobian inherits it"""
    ctx.update_compression_mode(mode)
    try:
        _conditioned_case(ctx, oracle, cond, tol)
    finally:
        ctx.update_compression_mode(0)


def _conditioned_case(ctx, oracle, cond, tol):
    n, k, F, M = 60, 44, 12, 6
    P = synth.spd_cov(n, seed=3)
    cols = synth.col_map(n, k, seed=4, skip=15)
    rows, Hf, Hx, res = synth.msckf_batch(F=F, M=M, k=k, seed=5, outlier_frac=0.0)
    W = _conditioned(k, cond, 6)
    Hx = np.einsum("ab,fbi->fai", W.T, Hx)            # H <- H W per feature (Hx is stored [k][ld])
    q95 = synth.q95_table()
    rc0, P0, dx0, acc0, nr0 = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, q95, 1e6, 0.0)
    rc1, P1, dx1, acc1, nr1 = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, 1e6, 0.0)
    assert rc0 == rc1 == 0 and np.array_equal(acc0, acc1) and acc1.all()
    Pt, dxt = _truth(P, rows, Hf, Hx, res, cols, acc1)
    e_lib, e_orc = max(_rel(P1, Pt), _rel(dx1, dxt)), max(_rel(P0, Pt), _rel(dx0, dxt))
    print(f"cond {cond:.0e}: library vs extended precision {e_lib:.2e}, Givens oracle vs extended precision {e_orc:.2e}, "
          f"library vs oracle {max(_rel(P1, P0), _rel(dx1, dx0)):.2e}")
    assert _rel(P1, P0) < tol and _rel(dx1, dx0) < tol
    assert e_lib < max(10 * e_orc, tol)


def _unit_pivots(P, order):
    """pivots of the Cholesky factorisation of P scaled to unit diagonal, states eliminated in `order`: the conditional variance of
    every state given the ones before it, as a fraction of its variance"""
    d = np.sqrt(np.diag(P))
    A = (P / np.outer(d, d))[np.ix_(order, order)].astype(np.longdouble)
    out = np.zeros(len(order))
    for j in range(len(order)):
        out[j] = float(A[j, j])
        A[j + 1:, j] /= A[j, j]
        A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j + 1:, j]) * A[j, j]
    return out


def test_whitened_update_keeps_small_conditional_variances(ctx, oracle):
    """A window of clones as a running filter has it: every clone position is the one before plus 1e-9 of its variance (known to 3e-5
    of the global position's uncertainty).  What an update must not lose is that conditional variance — the pivot of the covariance
    scaled to unit diagonal.  The whitened update obtains W0 = M^-1 P[cols, :] by substitution through the factor's small pivots; its
    columns of the update's own states are rows of the factor and are copied from it (dense_kernels.hip prior_exact_cols_kernel,
    DESIGN 10.3): with them the posterior's small pivots agree with an extended-precision update to 1e-6 of themselves (measured
    9.8e-7; the Givens oracle's P - K H P: 2.9e-7) — eps / pivot, the class of the reference's own form.  One update does not tell
    the forms apart the way a drive does (the factor form of round 4's first scheme was at 2e-7 here and lost the covariance of a drive after
    130 frames: test_gpu_replay.py test_covariance_pivots_follow_the_cpu_oracle); this test pins the default at the kernel."""
    n, k, F, M = 60, 44, 12, 6
    P0 = synth.spd_cov(n, seed=3)
    cols = synth.col_map(n, k, seed=4, skip=15)
    chain = [int(c) for c in cols[8:20]]                 # twelve states of the update's own columns, each "the one before + a little"
    T = np.eye(n)
    for a, b in zip(chain[:-1], chain[1:]):
        T[b] = T[a]
    P = T @ P0 @ T.T
    for i, b in enumerate(chain[1:]):
        for c in chain[i + 1:]:
            for c2 in chain[i + 1:]:
                P[c, c2] += 1e-9 * P0[chain[0], chain[0]]
    P = 0.5 * (P + P.T)
    rows, Hf, Hx, res = synth.msckf_batch(F=F, M=M, k=k, seed=5, outlier_frac=0.0)
    q95 = synth.q95_table()
    rc0, Pg, dx0, acc0, _ = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, q95, 1e6, 0.0)
    rc1, P1, dx1, acc1, _ = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, 1e6, 0.0)
    _, route, _ = ctx.update_compression_mode()
    assert rc0 == rc1 == 0 and np.array_equal(acc0, acc1) and acc1.all() and route == 4
    Pt, dxt = _truth(P, rows, Hf, Hx, res, cols, acc1)
    order = [int(c) for c in cols] + [i for i in range(n) if i not in set(int(c) for c in cols)]
    pos = [order.index(b) for b in chain[1:]]
    piv_prior = _unit_pivots(P, order)[pos]
    piv_t, piv_l, piv_g = (_unit_pivots(X, order)[pos] for X in (Pt, P1, Pg))
    e_l, e_g = np.abs(piv_l - piv_t) / piv_t, np.abs(piv_g - piv_t) / piv_t
    print("small pivots of the prior %.2e .. %.2e, of the posterior %.2e .. %.2e; largest relative error: library %.2e, Givens oracle %.2e"
          % (piv_prior.min(), piv_prior.max(), piv_t.min(), piv_t.max(), e_l.max(), e_g.max()))
    assert piv_prior.max() < 1e-8 and piv_t.min() > 0
    assert (piv_l > 0).all() and e_l.max() < 1e-5 and e_l.max() < max(30 * e_g.max(), 1e-6)
    assert _rel(dx1, dxt) < 1e-8 and _rel(P1, Pt) < 1e-9


def test_unsupported_compression_modes_are_refused(ctx):
    """the Gram + Cholesky compression (modes 2 and 3 of rounds 2-3: it squared the condition number and needed a second route to fall
    back on) is gone: the mode switch refuses them, and the two routes that remain handle cond 1e8 (test_msckf_update_conditioned_columns)"""
    for m in (2, 3, 4):
        with pytest.raises(Exception):
            ctx.update_compression_mode(m)
    assert ctx.update_compression_mode()[0] == 0


def test_msckf_update_duplicated_and_scaled_rows(ctx, oracle):
    n, k, F, M = 70, 50, 10, 6
    P = synth.spd_cov(n, seed=8)
    cols = synth.col_map(n, k, seed=9, skip=15)
    rows, Hf, Hx, res = synth.msckf_batch(F=F, M=M, k=k, seed=10, outlier_frac=0.0)
    # every feature twice (the stacked Jacobian has each row pair twice: rank-deficient row space), then whole features scaled
    rows, Hf, Hx, res = np.tile(rows, 2), np.tile(Hf, (2, 1, 1)), np.tile(Hx, (2, 1, 1)), np.tile(res, (2, 1))
    sc = np.array([1e3 if f % 3 == 0 else (1e-3 if f % 3 == 1 else 1.0) for f in range(len(rows))])
    Hf, Hx, res = Hf * sc[:, None, None], Hx * sc[:, None, None], res * sc[:, None]
    q95 = synth.q95_table()
    rc0, P0, dx0, acc0, nr0 = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, q95, 1e9, 0.0)
    rc1, P1, dx1, acc1, nr1 = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, 1e9, 0.0)
    assert rc0 == rc1 == 0 and np.array_equal(acc0, acc1) and acc1.all() and nr0 == nr1
    assert _rel(P1, P0) < 1e-8 and _rel(dx1, dx0) < 1e-8


def test_msckf_update_exact_null_directions(ctx, oracle):
    """four directions every row is orthogonal to (the gauge freedom an MSCKF Jacobian has): the Gram matrix is singular to rounding,
    the factorisation must treat those pivots as zero rows, and the covariance along them must stay what it was"""
    n, k, F, M = 70, 50, 14, 6
    P = synth.spd_cov(n, seed=12)
    cols = synth.col_map(n, k, seed=13, skip=15)
    rows, Hf, Hx, res = synth.msckf_batch(F=F, M=M, k=k, seed=14, outlier_frac=0.0)
    rng = np.random.default_rng(15)
    G, _ = np.linalg.qr(rng.normal(size=(k, 4)))
    Pr = np.eye(k) - G @ G.T
    Hx = np.einsum("ab,fbi->fai", Pr.T, Hx)            # H <- H (I - G G^T)
    q95 = synth.q95_table()
    rc0, P0, dx0, acc0, nr0 = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, q95, 1e6, 0.0)
    rc1, P1, dx1, acc1, nr1 = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, 1e6, 0.0)
    assert rc0 == rc1 == 0 and np.array_equal(acc0, acc1) and acc1.all()
    assert _rel(P1, P0) < 1e-8 and _rel(dx1, dx0) < 1e-8
    g = np.zeros((n, 4))
    g[cols] = G
    Pinv_g = np.linalg.solve(P, g)
    # information along the null directions is untouched: g^T P'^-1 g == g^T P^-1 g
    assert np.allclose(g.T @ np.linalg.solve(P1, g), g.T @ Pinv_g, rtol=1e-6)


@pytest.mark.parametrize("depth", [20.0, 2e3, 2e5])
def test_fused_householder_nullspace_on_weak_parallax(ctx, pkg, depth):
    """the fused Jacobian launch (Householder reflections) against the oracle's Givens projection where Hf is badly conditioned: points
    so far away that the three columns of Hf are nearly dependent (cond(Hf) grows with depth / baseline).  The projected blocks differ
    by a rotation of the left null space; accepted set, dx and P' must not."""
    import oracle_lib
    jo, orc = oracle_lib.load_jac(pkg), oracle_lib.load()
    scene = synth.vio_scene(n_clones=12, F=16, M=10, seed=21, noise_px=0.3, depth_scale=depth / 20.0)
    st, tr = synth.scene_views(pkg, scene)
    n = scene["n_state"]
    P = synth.spd_cov(n, seed=22)
    cols = jo.columns(st, tr)
    rows, Hf, Hx, res = jo.build_jacobians(st, tr, cols, 20)
    conds = [np.linalg.cond(Hf[f, :, :rows[f]].T) for f in range(len(rows)) if rows[f] > 3]
    rc0, P0, dx0, acc0, nr0 = orc.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table(), 1e6, 0.0)
    ctx.cov_upload(P)
    ctx.build_jacobians_resident(st, tr, cols, 20)      # builds AND projects (Householder) on the device
    rc1, dx1, acc1, nr1 = ctx.msckf_update_resident(n, 2.25, 1e6, 0.0)
    P1 = ctx.cov_download(n)
    assert rc0 == rc1 == 0 and np.array_equal(acc0, acc1) and nr0 == nr1 and acc1.sum() >= 8
    print(f"depth {depth:.0e}: cond(Hf) up to {max(conds):.1e}, dx {_rel(dx1, dx0):.2e}, P {_rel(P1, P0):.2e}")
    assert _rel(P1, P0) < 1e-8 and _rel(dx1, dx0) < 1e-7
