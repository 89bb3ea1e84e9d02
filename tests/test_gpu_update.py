"""GPU parity tests (through the C-ABI) of the update half: HIP vs the CPU oracle on the same
seeded inputs.  Tolerances: fp64 linear algebra, different summation order (MFMA tiles, Householder
vs Givens) -> relative 1e-9 on P / dx; the Givens nullspace keeps the reference's rotation order and
is compared at 1e-12."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("n,k,r", [(113, 98, 98), (143, 128, 128), (30, 12, 40), (17, 17, 1), (113, 98, 27)])
def test_ekf_update_parity(ctx, oracle, n, k, r):
    P = synth.spd_cov(n, seed=n)
    cols = synth.col_map(n, k, seed=k, skip=min(15, n - k))
    rng = np.random.default_rng(r)
    H = rng.normal(size=(r, k))
    res = rng.normal(size=r) * 0.1
    rc0, P0, dx0 = oracle.ekf_update(P, H, cols, res)
    rc1, P1, dx1 = ctx.ekf_update(P, H, cols, res)
    assert rc0 == 0 and rc1 == 0
    assert _rel(dx1, dx0) < 1e-9
    assert _rel(P1, P0) < 1e-9
    assert np.array_equal(P1, P1.T)


def test_ekf_update_rdiag_and_resident(ctx, oracle):
    n, k, r = 60, 40, 33
    P = synth.spd_cov(n, seed=1)
    cols = synth.col_map(n, k, seed=2)
    rng = np.random.default_rng(3)
    H = rng.normal(size=(r, k))
    res = rng.normal(size=r)
    Rd = rng.uniform(0.5, 2.0, r)
    rc0, P0, dx0 = oracle.ekf_update(P, H, cols, res, Rd)
    rc1, P1, dx1 = ctx.ekf_update(P, H, cols, res, Rd)
    assert rc0 == rc1 == 0
    assert _rel(P1, P0) < 1e-9 and _rel(dx1, dx0) < 1e-9
    # device-resident covariance: upload, update twice with P=NULL, download
    ctx.cov_upload(P)
    import ctypes as C
    lib = ctx.lib
    Hf = np.asfortranarray(H)
    dx = np.zeros(n)
    dp = C.POINTER(C.c_double)
    for _ in range(2):
        rc = lib.plv_ekf_update(ctx.h, None, n, n, Hf.ctypes.data_as(dp), r, k, r,
                                cols.ctypes.data_as(C.POINTER(C.c_int)), res.ctypes.data_as(dp),
                                Rd.ctypes.data_as(dp), dx.ctypes.data_as(dp))
        assert rc == 0
    Pd = ctx.cov_download(n)
    _, P0b, dx0b = oracle.ekf_update(P0, H, cols, res, Rd)
    assert _rel(Pd, P0b) < 1e-9 and _rel(dx, dx0b) < 1e-9


def test_ekf_update_not_psd_leaves_state(ctx, oracle, pkg):
    n = 20
    P = np.eye(n) * 1e-4
    P[0, 1] = P[1, 0] = 5e-3
    cols = np.arange(n, dtype=np.int32)
    H = np.eye(n)
    res = np.ones(n)
    Rd = np.full(n, 1e-8)
    rc0, _, _ = oracle.ekf_update(P, H, cols, res, Rd)
    rc1, P1, dx1 = ctx.ekf_update(P, H, cols, res, Rd)
    assert rc0 == -3 and rc1 == pkg.PLV_E_NOT_PSD
    assert np.array_equal(P1, P) and np.all(dx1 == 0)


@pytest.mark.parametrize("fdim,k,M", [(3, 98, 15), (6, 90, 15), (3, 38, 5), (3, 128, 20)])
def test_nullspace_batch_parity(ctx, oracle, fdim, k, M):
    rows, Hf, Hx, res = synth.msckf_batch(F=37, M=M, k=k, fdim=fdim, seed=fdim + k)
    rows = np.maximum(rows, 2 * ((fdim + 3) // 2) + 2).astype(np.int32)
    a = oracle.nullspace_batch(rows, Hf, Hx, res)
    b = ctx.nullspace_batch(rows, Hf, Hx, res)
    for f in range(len(rows)):
        m = rows[f]
        assert np.allclose(b[0][f, :, :m], a[0][f, :, :m], rtol=0, atol=1e-12)
        assert np.allclose(b[1][f, :, :m - fdim], a[1][f, :, :m - fdim], rtol=0, atol=1e-12)
        assert np.allclose(b[2][f, :m - fdim], a[2][f, :m - fdim], rtol=0, atol=1e-12)
    # report how close to bit-exact the rotation replay is
    print("nullspace max |diff|:", max(np.max(np.abs(b[1][f, :, :rows[f] - fdim] - a[1][f, :, :rows[f] - fdim]))
                                      for f in range(len(rows))))


def test_chi2_batch_parity(ctx, oracle):
    n, k = 113, 98
    P = synth.spd_cov(n)
    cols = synth.col_map(n, k)
    rows, Hf, Hx, res = synth.msckf_batch(F=70, M=15, k=k, seed=9)
    rows = rows - 3  # as after a nullspace projection
    chi0 = oracle.chi2_batch(P, rows, Hx, res, cols, 2.25)
    chi1 = ctx.chi2_batch(P, rows, Hx, res, cols, 2.25)
    assert np.allclose(chi1, chi0, rtol=1e-9)


# (round 6b: hqr_kernel — one workgroup up to 1024 / 2048 rows, several above with a second launch over their R's; the tree of
# unblocked factorisations below 1.5 x the width.  Sizes on both sides of every switch, widths that are no multiple of 16.)
@pytest.mark.parametrize("m,k", [(1890, 98), (300, 40), (2600, 128), (150, 98), (99, 98), (5000, 20), (2049, 104), (1025, 33), (1024, 15), (700, 191), (4100, 150),
                                 (160, 104), (8200, 64)])
def test_compress_parity(ctx, oracle, m, k):
    rng = np.random.default_rng(m + k)
    H = rng.normal(size=(m, k))
    if m >= 1.5 * k:  # zero rows (rejected features are stacked as zeros); keep full column rank
        H[rng.uniform(size=m) < 0.2] = 0.0
    r = rng.normal(size=m)
    R0, z0 = oracle.compress(H, r)
    R1, z1 = ctx.compress(H, r)
    assert R1.shape == (k, k)
    assert np.all(np.tril(R1, -1) == 0) and np.all(np.diag(R1) >= 0)
    assert _rel(R1, R0) < 1e-9 and _rel(z1, z0) < 1e-9
    assert np.allclose(R1.T @ R1, H.T @ H, rtol=1e-10, atol=1e-9)


def test_compress_blocked_and_tree_agree(ctx, pkg):
    """hqr_kernel (default) and the tree of rounds 1-6a (knob 1 << 29) on a stack with dependent and empty columns: the same R^T R and the
    same |R|, |z| where the factor is determined (the leading independent columns)."""
    rng = np.random.default_rng(77)
    m, k = 900, 104
    H = rng.normal(size=(m, k))
    H[:, 60] = H[:, 10] - 2.0 * H[:, 20]
    H[:, 70] = 0.0
    r = rng.normal(size=m)
    try:
        pkg.debug_knobs(0)
        R1, z1 = ctx.compress(H, r)
        pkg.debug_knobs(1 << 29)
        R2, z2 = ctx.compress(H, r)
    finally:
        pkg.debug_knobs(0)
    G = H.T @ H
    for R in (R1, R2):
        assert np.all(np.tril(R, -1) == 0) and np.all(np.diag(R) >= 0)
        assert np.abs(R.T @ R - G).max() < 1e-10 * np.abs(G).max()
    assert np.allclose(R1[:60, :60], R2[:60, :60], rtol=1e-9, atol=1e-11) and np.allclose(z1[:60], z2[:60], rtol=1e-9, atol=1e-11)


def test_compress_fat_is_identity(ctx):
    rng = np.random.default_rng(0)
    H = rng.normal(size=(10, 40))
    r = rng.normal(size=10)
    R, z = ctx.compress(H, r)
    assert np.array_equal(R, H) and np.array_equal(z, r)


@pytest.mark.parametrize("F,M,n,k,fdim,gate", [(70, 15, 113, 98, 3, 3.0), (250, 15, 113, 98, 3, 3.0),
                                               (80, 15, 105, 90, 6, 0.0), (70, 20, 143, 128, 3, 3.0), (500, 20, 143, 128, 3, 3.0),
                                               (3, 4, 40, 26, 3, 3.0), (40, 8, 60, 45, 3, 3.0),
                                               (20, 6, 40, 26, 3, 3.0),
                                               # BASELINE configs[3]: 20-clone window + the pose of the newest frame, intrinsics calibrated
                                               # (n = 149, k = 134 columns), 70 point features / 150 lines of 20 observations
                                               (70, 20, 149, 134, 3, 3.0), (150, 20, 149, 127, 6, 0.0), (150, 20, 149, 134, 6, 0.0),
                                               (60, 24, 205, 190, 3, 3.0)])
def test_msckf_update_parity(ctx, oracle, F, M, n, k, fdim, gate):
    P = synth.spd_cov(n, seed=F)
    cols = synth.col_map(n, k, seed=M, skip=min(15, n - k))
    rows, Hf, Hx, res = synth.msckf_batch(F=F, M=M, k=k, fdim=fdim, seed=F + M)
    if fdim == 6:
        rows = np.maximum(rows, 10).astype(np.int32)
    q95 = synth.q95_table()
    rc0, P0, dx0, acc0, nr0 = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, q95, 1.0, gate)
    rc1, P1, dx1, acc1, nr1 = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, 1.0, gate)
    assert rc0 == 0 and rc1 == 0
    assert np.array_equal(acc0, acc1) and nr0 == nr1
    assert 0 < acc1.sum()
    assert _rel(dx1, dx0) < 1e-8
    assert _rel(P1, P0) < 1e-8
    assert np.array_equal(P1, P1.T)


def test_msckf_update_golden(ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "update_small.npz"))
    rc, P1, dx1, acc, nr = ctx.msckf_update(g["P"], g["rows"], g["Hf"], g["Hx"], g["res"], g["cols"],
                                            float(g["sigma2"]))
    assert rc == 0 and np.array_equal(acc, g["accepted"]) and nr == int(g["n_rows"])
    assert _rel(dx1, g["dx"]) < 1e-9 and _rel(P1, g["P_new"]) < 1e-9


def test_msckf_update_resident_repeatable(ctx, oracle):
    """Staged batch + device-resident covariance + rollback: the bench's step is repeatable."""
    n, k = 113, 98
    P = synth.spd_cov(n)
    cols = synth.col_map(n, k)
    rows, Hf, Hx, res = synth.msckf_batch(F=70, M=15, k=k, seed=1)
    rc0, P0, dx0, acc0, _ = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table())
    ctx.cov_upload(P)
    ctx.cov_checkpoint()
    ctx.feat_batch_upload(rows, Hf, Hx, res, cols)
    for _ in range(3):
        ctx.cov_rollback()
        rc, dx, acc, nr = ctx.msckf_update_resident(n, 2.25)
        assert rc == 0 and np.array_equal(acc, acc0)
        assert _rel(dx, dx0) < 1e-8
    assert _rel(ctx.cov_download(n), P0) < 1e-8


def test_update_graph_replay_matches_eager(pkg, oracle):
    """plv_update_graph_mode: eager, captured and replayed launches give the same bits; a new shape retires the graph."""
    ctx = pkg.Context(pkg.default_config(752, 480))
    n, k = 113, 98
    P = synth.spd_cov(n)
    cols = synth.col_map(n, k)
    rows, Hf, Hx, res = synth.msckf_batch(F=40, M=15, k=k, seed=5)
    ref = None
    ctx.update_graph_mode(1)
    for it in range(5):
        ctx.cov_upload(P)
        ctx.feat_batch_upload(rows, Hf, Hx, res, cols)
        rc, dx, acc, nr = ctx.msckf_update_resident(n, 2.25)
        out = (rc, dx.copy(), acc.copy(), nr, ctx.cov_download(n))
        if ref is None:
            ref = out
            rc_o, P_o, dx_o, acc_o, _ = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table())
            assert rc == rc_o == 0 and np.array_equal(acc, acc_o) and np.abs(out[4] - P_o).max() < 1e-8 * np.abs(P_o).max()
        else:
            assert out[0] == ref[0] and np.array_equal(out[1], ref[1]) and np.array_equal(out[2], ref[2]) and np.array_equal(out[4], ref[4])
    cap, rep = ctx.update_graph_mode()
    assert cap == 1 and rep == 3          # eager, capture (+ its launch), three replays
    rows2, Hf2, Hx2, res2 = synth.msckf_batch(F=25, M=15, k=k, seed=6)   # another shape: eager again, then a new capture
    for it in range(3):
        ctx.cov_upload(P)
        ctx.feat_batch_upload(rows2, Hf2, Hx2, res2, cols)
        rc, dx, acc, nr = ctx.msckf_update_resident(n, 2.25)
        rc_o, P_o, dx_o, acc_o, _ = oracle.msckf_update(P, rows2, Hf2, Hx2, res2, cols, 2.25, synth.q95_table())
        assert rc == rc_o and np.array_equal(acc, acc_o) and np.abs(ctx.cov_download(n) - P_o).max() < 1e-8 * np.abs(P_o).max()
    cap2, rep2 = ctx.update_graph_mode()
    assert cap2 == 2 and rep2 == 4
    ctx.update_graph_mode(0)
    ctx.close()


@pytest.mark.parametrize("k", [12, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 95, 96, 97, 111, 112, 113, 127, 128, 129, 143, 144, 145,
                               159, 160, 161, 176, 191, 192])
def test_msckf_update_tile_boundaries(ctx, oracle, k):
    """Every 16-column tile boundary of the blocked factorisations (the kernels are instantiated for 2, 4, 7, 8, 10 and 12 tiles), with
    fewer stacked rows than columns for the small ones (no compression) and more for the rest."""
    n = k + 15
    F, M = (4, 4) if k >= 31 and k % 2 else (30, 8)
    P = synth.spd_cov(n, seed=k)
    cols = synth.col_map(n, k, seed=k, skip=15)
    rows, Hf, Hx, res = synth.msckf_batch(F=F, M=M, k=k, seed=k)
    q95 = synth.q95_table()
    rc0, P0, dx0, acc0, nr0 = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, q95, 1.0, 3.0)
    rc1, P1, dx1, acc1, nr1 = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, 1.0, 3.0)
    assert rc0 == rc1 == 0 and np.array_equal(acc0, acc1) and nr0 == nr1
    tol = 1e-8 if k <= 160 else 1e-7   # (the Gram matrix squares the condition number: 190 columns of this generator cost a digit)
    if acc1.sum():
        assert _rel(dx1, dx0) < tol and _rel(P1, P0) < tol
    else:
        assert np.array_equal(P1, P)


@pytest.mark.parametrize("r", [1, 2, 15, 16, 17, 31, 33, 63, 65, 96, 97, 111, 113, 127, 128, 129, 150, 260])
def test_ekf_update_row_boundaries(ctx, oracle, r):
    """Measurement-row counts around every tile boundary, including r > 128 (the factorisation that does not fit LDS takes the
    general path) and r > k (more rows than columns, as the wheel / GPS updaters may pass)."""
    n, k = 113, 98
    P = synth.spd_cov(n, seed=r)
    cols = synth.col_map(n, k, seed=r, skip=15)
    rng = np.random.default_rng(r)
    H = rng.normal(size=(r, k))
    res = rng.normal(size=r)
    Rd = rng.uniform(0.5, 2.0, r)
    rc0, P0, dx0 = oracle.ekf_update(P, H, cols, res, Rd)
    rc1, P1, dx1 = ctx.ekf_update(P, H, cols, res, Rd)
    assert rc0 == rc1 == 0
    assert _rel(dx1, dx0) < 1e-8 and _rel(P1, P0) < 1e-8 and np.array_equal(P1, P1.T)


def test_ekf_update_chunked_rejection_restores_state(ctx, pkg):
    """r > 128 goes block after block; a rejection in a LATER block must undo the earlier ones (EKFUpdate is all-or-nothing)."""
    n = 20
    P = np.eye(n) * 1e-4
    P[0, 1] = P[1, 0] = 5e-3                      # indefinite in the (0, 1) plane
    cols = np.arange(n, dtype=np.int32)
    r = 150
    H = np.zeros((r, n))
    H[:128, 2:] = np.random.default_rng(0).normal(size=(128, n - 2)) * 0.1   # the first block never touches states 0 and 1
    H[128:128 + n, :] = np.eye(n)                 # the second block does
    res = np.ones(r)
    Rd = np.full(r, 1e-8)
    ctx.cov_upload(P)
    rc, P1, dx1 = ctx.ekf_update(P, H, cols, res, Rd)
    assert rc == pkg.PLV_E_NOT_PSD
    assert np.array_equal(P1, P) and np.all(dx1 == 0)
    assert np.array_equal(ctx.cov_download(n), P)
