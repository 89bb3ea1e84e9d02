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


# ------------------------------------------------------------------ images (SURVEY.md §8(d) cfg 2)
def texture_canvas(w, h, seed=42, margin=64, blobs=300, lines=0):
    """Band-limited noise texture (sigma_blur 1.5 px, contrast-stretched) + Harris-strong blobs,
    optionally straight high-contrast edges, on a canvas larger than the image by `margin`."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    W, H = w + 2 * margin, h + 2 * margin
    img = rng.normal(0, 1, (H, W))
    img = ndi.gaussian_filter(img, 1.5) * 3.0 + ndi.gaussian_filter(rng.normal(0, 1, (H, W)), 6.0) * 6.0
    for _ in range(blobs):
        x, y = rng.uniform(8, W - 8), rng.uniform(8, H - 8)
        s = rng.uniform(2.0, 4.0)
        amp = rng.choice([-1, 1]) * rng.uniform(1.0, 2.0)
        x0, x1, y0, y1 = int(x - s), int(x + s), int(y - s), int(y + s)
        img[y0:y1, x0:x1] += amp
    for _ in range(lines):
        L = rng.uniform(60, 300)
        th = rng.uniform(0, np.pi)
        cx, cy = rng.uniform(40, W - 40), rng.uniform(40, H - 40)
        n = int(L)
        t = np.linspace(-L / 2, L / 2, n)
        for off in range(-1, 2):
            xs = np.clip((cx + t * np.cos(th) - off * np.sin(th)).astype(int), 0, W - 1)
            ys = np.clip((cy + t * np.sin(th) + off * np.cos(th)).astype(int), 0, H - 1)
            img[ys, xs] += 3.0
    img = ndi.gaussian_filter(img, 0.8)
    lo, hi = np.percentile(img, [1, 99])
    img = np.clip((img - lo) / (hi - lo), 0, 1)
    return img


def render_frame(canvas, w, h, tx=0.0, ty=0.0, rot_deg=0.0, scale=1.0, margin=64):
    """Samples the canvas under a similarity warp about the image centre (cubic interpolation).
    Returns the u8 frame; a canvas point c maps to frame pixel p = s R (c - c0) + c0 + t."""
    from scipy import ndimage as ndi
    th = np.deg2rad(rot_deg)
    cx, cy = margin + w / 2.0, margin + h / 2.0
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    # inverse map: frame pixel -> canvas
    px, py = xs + margin - cx - tx, ys + margin - cy - ty
    c, s = np.cos(-th), np.sin(-th)
    qx = (c * px - s * py) / scale + cx
    qy = (s * px + c * py) / scale + cy
    out = ndi.map_coordinates(canvas, [qy, qx], order=3, mode="reflect")
    return np.clip(np.rint(out * 255.0), 0, 255).astype(np.uint8)


def warp_points(pts, w, h, tx, ty, rot_deg, scale):
    """Where frame-0 (identity warp) pixel positions land in the warped frame."""
    th = np.deg2rad(rot_deg)
    c0 = np.array([w / 2.0, h / 2.0])
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    return ((pts - c0) @ R.T) * scale + c0 + np.array([tx, ty])


def grid_points(w, h, n, seed=1, border=24):
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(border, w - border, n), rng.uniform(border, h - border, n)]).astype(np.float32)


EUROC_K8 = np.array([458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
