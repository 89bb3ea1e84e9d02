"""Seeded synthetic problems for the update half of the path (sizes follow SURVEY.md §8(d))."""
import numpy as np


def spd_cov(n, seed=7, lo=1e-6, hi=1e-2):
    """P = A A^T + 1e-3 I scaled so that diag lies in [lo, hi] (SURVEY §8(d) cfg B)."""
    rng = np.random.default_rng(seed)
    A = rng.normal(0, 1e-2, (n, n))
    P = A @ A.T + 1e-3 * np.eye(n) * 1e-2
    d = np.sqrt(np.diag(P))
    target = np.sqrt(np.exp(rng.uniform(np.log(lo), np.log(hi), n)))
    s = target / d
    P = P * s[:, None] * s[None, :]
    return 0.5 * (P + P.T)


def col_map(n, k, seed=3, skip=15):
    """k distinct state columns, skipping the first `skip` (IMU) entries when possible."""
    rng = np.random.default_rng(seed)
    pool = np.arange(skip, n) if n - skip >= k else np.arange(n)
    cols = np.sort(rng.choice(pool, size=k, replace=False)).astype(np.int32)
    return cols


def msckf_batch(F=70, M=15, k=98, fdim=3, seed=11, ragged=True, outlier_frac=0.15, ld=None, sigma=1.0):
    """Per-feature whitened systems (Hf [F,fdim,ld], Hx [F,k,ld], res [F,ld], rows [F]).
    Each feature touches a contiguous window of 6-wide clone blocks plus the last 8 columns
    (intrinsics), mimicking CamHelper::get_feature_jacobian_full's sparsity."""
    rng = np.random.default_rng(seed)
    ld = ld or 2 * M
    rows = np.full(F, 2 * M, dtype=np.int32)
    if ragged:
        rows = (2 * rng.integers(max(3, M // 3), M + 1, F)).astype(np.int32)
        rows[0] = 2 * M
    Hf = np.zeros((F, fdim, ld))
    Hx = np.zeros((F, k, ld))
    res = np.zeros((F, ld))
    nblk = max(1, (k - 8) // 6)
    for f in range(F):
        r = rows[f]
        Hf[f, :, :r] = rng.normal(0, 1.0, (fdim, r))
        nobs = r // 2
        start = rng.integers(0, max(1, nblk - nobs + 1))
        for o in range(nobs):
            b = min(nblk - 1, start + o)
            Hx[f, 6 * b:6 * b + 6, 2 * o:2 * o + 2] = rng.normal(0, 1.0, (6, 2))
            # interpolation spreads a little onto the neighbours
            if b + 1 < nblk:
                Hx[f, 6 * b + 6:6 * b + 12, 2 * o:2 * o + 2] = rng.normal(0, 0.05, (6, 2))
        if k >= 8:
            Hx[f, k - 8:, :r] = rng.normal(0, 0.3, (8, r))
        scale = 5.0 if rng.uniform() < outlier_frac else 0.35
        res[f, :r] = rng.normal(0, sigma * scale, r)
    return rows, Hf, Hx, res


def q95_table(n=1024):
    from scipy.stats import chi2
    t = np.zeros(n)
    t[1:] = chi2.ppf(0.95, np.arange(1, n))
    return t
