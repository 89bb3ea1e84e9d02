"""Algorithmic work per kernel launch (bytes for HBM-class kernels, flops for the dense fp64 ones): the per-unit figures of
SURVEY.md §8(d), restated in DESIGN.md §4, times the units one launch processes.  Shared by bench.py and tests/bench_chain.py."""

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TF = 78.6    # vendor FP64 matrix figure (v_mfma_f64_16x16x4_f64)
HBM_KERNELS = {"gram_reduce_kernel", "gather_cov_kernel", "ekf_commit_kernel", "prior_exact_cols_kernel"}

# Where each per-launch figure comes from (bench.py prints it next to the figure):
#   "8d"       SURVEY.md 8(d)'s formula, verbatim (front-end bytes B_frame; null space, chi2 gate, compression, EKF flops)
#   "8d-split" a term of 8(d)'s EKF / compression sum, attributed to the kernel that does it
#   "estimate" the builder's own operation count for a step 8(d) does not price (Jacobian rows, triangulation, RANSAC, poses): scalar
#              fp64 chains — these kernels are latency-bound and their "frac" against the matrix peak says only that
PROVENANCE = {
    "hist_kernel": "8d", "equalize_kernel": "8d", "pyrdown_kernel": "8d", "pyrdown2_kernel": "8d", "lk_kernel": "8d", "undistort_kernel": "8d",
    "fast_cells_kernel": "8d", "half_kernel": "8d", "canny_kernel": "8d", "half_canny_kernel": "8d", "subpix_kernel": "estimate",
    "ransac_hyp_kernel": "estimate", "ransac_select_kernel": "estimate", "campose_kernel": "estimate", "triangulate_kernel": "estimate",
    "line_triangulate_kernel": "estimate", "jacobian_kernel": "estimate", "line_jacobian_kernel": "estimate",
    "jacobian_nullspace_kernel": "8d null space + estimate (rows)", "line_jacobian_nullspace_kernel": "8d null space + estimate (rows)",
    "tri_jacobian_nullspace_kernel": "8d null space + chi2 gate; estimate (triangulation, rows)",
    "line_tri_jacobian_nullspace_kernel": "8d null space + chi2 gate; estimate (triangulation, rows)",
    "nullspace_kernel": "8d", "chi2_gate_kernel": "8d-split", "chi2_t_kernel": "8d-split", "qr_accum_kernel": "8d",
    "gram_chunk_kernel": "8d-split", "gram_direct_kernel": "8d-split", "gram_reduce_kernel": "estimate", "bchol_compress_kernel": "8d-split",
    "bchol_ekf_kernel": "8d-split", "bchol_prior_kernel": "estimate (whitened route: not in 8d)", "prior_gain_kernel": "estimate (whitened route: not in 8d)",
    "prior_exact_cols_kernel": "estimate (whitened route: not in 8d)",
    "gather_cov_kernel": "estimate", "ekf_dc_kernel": "8d-split", "ekf_commit_kernel": "8d-split", "ekf_mt_kernel": "8d-split",
    "ekf_s_kernel": "8d-split", "ekf_ms_kernel": "8d-split",
}


def update_bytes(F, rows_f, fdim, k, n):
    """Algorithmic BYTES per launch of the update kernels (every operand once, every result once; fp64): what their counter traffic is
    set against (roofline.per_kernel[].traffic_over_algorithmic), whatever bounds them."""
    mp = max(rows_f - fdim, 0)
    m = F * mp
    nc = k + 1
    r = min(k, m) if m > 0 else k
    d = 8.0
    return {
        "bchol_prior_kernel": (k * n + k * k + k * (n + 1)) * d,          # P[cols, :] in; Lp^T, W0 out
        "prior_gain_kernel": (k * (n + 1) + n * n / 2.0) * d,              # W0 in; W0^T W0 (upper) out
        "prior_exact_cols_kernel": 2.0 * k * k * d,                        # Lp^T in, the k columns of W0 that are its rows out
        "nullspace_kernel": 2.0 * F * rows_f * (fdim + k + 1) * d,
        "chi2_t_kernel": (F * mp * k * 2 + k * k) * d,                     # H' in, T out, Ps once
        "chi2_gate_kernel": (F * mp * (2 * k + 1) + F * mp * nc) * d,      # T, H', r in; accepted rows to the stack
        "qr_accum_kernel": 2.0 * m * nc * d,
        "gram_chunk_kernel": (m * nc + (m / 64.0) * nc * nc / 2.0) * d,
        "gram_direct_kernel": (m * nc + nc * nc / 2.0) * d,
        "gram_reduce_kernel": (m / 64.0) * (nc * nc / 2.0) * d,
        "bchol_compress_kernel": (nc * nc / 2.0 + k * k / 2.0 + k) * d,
        "bchol_ekf_kernel": (r * r / 2.0 + 2.0 * r * (n + 1)) * d,        # S (upper), [M ; res] in, W out
        "gather_cov_kernel": 2.0 * (k * n + k * k) * d,
        "ekf_dc_kernel": (r * (n + 1) + n * n / 2.0 + n) * d,
        "ekf_commit_kernel": 3.0 * n * n * d,
        "ekf_mt_kernel": (r * k + k * n + r * n) * d,
        "ekf_s_kernel": (r * k + k * k + r * r / 2.0) * d,
        "ekf_ms_kernel": (r * k + k * k + r * r / 2.0 + r) * d,
    }


def frame_bytes(W, H, levels, n_pts, lk_iters, win, F, M, k, n, L=0, Ml=0, kl=0, n_new=0, pool_pts=0, pool_lines=0, n_clones=16):
    """Algorithmic bytes per launch for EVERY kernel of a frame (the HBM-class ones: the same figure frame_work gives)."""
    fw = frame_work(W, H, levels, n_pts, lk_iters, win, F, M, k, n, L=L, Ml=Ml, kl=kl, n_new=n_new, pool_pts=pool_pts, pool_lines=pool_lines)
    out = {name: v for name, (kind, v) in fw.items() if kind == "hbm"}
    win_b = n_clones * (8 + 72 + 24 + 72 + 24 + 4)              # the clone window: times, poses, first estimates, columns
    out["ransac_hyp_kernel"] = n_pts * 32.0 + 1000 * (3 * 72 + 16)    # the matches once; per hypothesis its models and inlier count
    out["ransac_select_kernel"] = 1000 * 16.0 + n_pts * (16 + 16 + 1)
    out["campose_kernel"] = (pool_pts * M + pool_lines * Ml) * (8 + 96.0)
    out["triangulate_kernel"] = pool_pts * (M * (8 + 8 + 96) + 40.0)
    out["line_triangulate_kernel"] = pool_lines * (Ml * (16 + 192) + 56.0)
    out["jacobian_kernel"] = F * (M * 16 + 24) + win_b + 8.0 * F * 2 * M * (3 + k + 1)
    out["line_jacobian_kernel"] = L * (Ml * 24 + 48) + win_b + 8.0 * L * 2 * Ml * (6 + kl + 1)
    # fused launches: every pool track + the window once per launch, the covariance block the gate contracts with once, and what
    # leaves the launch: the accepted entries' projected rows [H' | r] + verdicts (~0.45 MB at workload C)
    mp, mpl = max(2 * M - 3, 0), max(2 * Ml - 6, 0)
    out["jacobian_nullspace_kernel"] = pool_pts * M * 24.0 + win_b + 8.0 * F * mp * (k + 1) + 8.0 * F * 3 * 2 * M
    out["tri_jacobian_nullspace_kernel"] = out["jacobian_nullspace_kernel"] + 8.0 * k * k + pool_pts * 40.0
    if L > 0 or pool_lines > 0:
        out["line_jacobian_nullspace_kernel"] = pool_lines * Ml * 40.0 + win_b + 8.0 * L * mpl * (kl + 1) + 8.0 * L * 6 * 2 * Ml
        out["line_tri_jacobian_nullspace_kernel"] = out["line_jacobian_nullspace_kernel"] + 8.0 * kl * kl + pool_lines * 56.0
    ub = update_bytes(F, 2 * M, 3, k, n)
    if L > 0:
        ul = update_bytes(L, 2 * Ml, 6, kl, n)
        for name, v in ub.items():
            ub[name] = ul[name] if name == "nullspace_kernel" else 0.5 * (v + ul[name])
    for name, v in ub.items():
        out.setdefault(name, v)
    return out


def update_work(F, rows_f, fdim, k, n, qr_launches=1, whitened=True):
    """F features of rows_f rows (before the null-space projection removes fdim of them) on k columns of an n-state filter.
    whitened: the default route when there are more rows than columns (DESIGN.md "Whitened update") — ekf_ms_kernel then forms
    B = I + Lp^T G Lp and Lp^T g instead of Mt and S."""
    mp = max(rows_f - fdim, 0)
    m = F * mp
    nc = k + 1
    r = min(k, m) if m > 0 else k
    wh = whitened and m > k and k <= 192
    return {
        "bchol_prior_kernel": k ** 3 / 3.0 + 1.0 * k * k * n,      # factor of P[cols, cols] + the n border rows P[:, cols]
        "prior_gain_kernel": 1.0 * n * n * k,                      # W0^T W0, upper tiles
        "prior_exact_cols_kernel": 2.0 * k * k * 8.0,              # a copy: bytes (HBM_KERNELS)
        "nullspace_kernel": F * 6.0 * (fdim + k + 1) * max(rows_f * fdim - fdim * (fdim + 1) / 2, 0),
        "chi2_gate_kernel": F * (2.0 * mp * mp * k + mp ** 3 / 3.0),
        "chi2_t_kernel": F * 2.0 * mp * k * k,
        "qr_accum_kernel": (2.0 * m * nc * nc - (2.0 / 3) * nc ** 3) / max(1, qr_launches),
        "gram_chunk_kernel": 1.0 * m * nc * nc,
        "gram_direct_kernel": 1.0 * m * nc * nc,     # (upper tiles: m (k+1)^2 multiply-adds over the accepted rows)
        "gram_reduce_kernel": (m / 64.0) * (nc * nc / 2.0) * 8,
        "bchol_compress_kernel": nc ** 3 / 3.0,
        "bchol_ekf_kernel": r ** 3 / 3.0 + 1.0 * r * r * (n + 1),
        "gather_cov_kernel": 2.0 * (k * n + k * k) * 8,
        "ekf_dc_kernel": 1.0 * n * n * r + 2.0 * n * r,
        "ekf_commit_kernel": 3.0 * n * n * 8,
        "ekf_mt_kernel": 2.0 * n * k * r,
        "ekf_s_kernel": 2.0 * r * r * k,
        # Mt tiles + every strip's own H Ps + the upper S tiles (whitened: one column instead of the n of Mt)
        "ekf_ms_kernel": 2.0 * (1 if wh else n) * k * r + 2.0 * r * k * k + 1.0 * r * r * k,
    }


def frame_work(W, H, levels, n_pts, lk_iters, win, F, M, k, n, L=0, Ml=0, kl=0, n_new=0, pool_pts=0, pool_lines=0):
    """Per-launch work of every kernel of a frame: n_pts tracked points (lk_iters LK iterations in total), F point features of M
    observations on k columns, L lines of Ml observations on kl columns, n_new corners refined, pool_* triangulated candidates."""
    pyr = (4.0 / 3 + 1.0 / 3) * W * H
    it = lk_iters / float(max(1, n_pts * levels))
    work = {
        "hist_kernel": ("hbm", W * H),
        "equalize_kernel": ("hbm", 2.0 * W * H),
        "pyrdown_kernel": ("hbm", pyr / max(1, levels - 1)),
        "pyrdown2_kernel": ("hbm", pyr / 2.0),
        "lk_kernel": ("hbm", n_pts * levels * ((win + 2) ** 2 + it * (win + 1) ** 2) + n_pts * 17),
        "undistort_kernel": ("hbm", 2.0 * n_pts * 16),
        "fast_cells_kernel": ("hbm", 1.0 * W * H),
        "subpix_kernel": ("hbm", n_new * 20 * 13 * 13 * 1.0),
        "half_kernel": ("hbm", W * H * 1.25),
        "canny_kernel": ("hbm", 2.0 * W * H / 4.0),
        "half_canny_kernel": ("hbm", W * H * 1.25),
        "ransac_hyp_kernel": ("mfma", 1000 * 3 * n_pts * 40.0),
        "ransac_select_kernel": ("mfma", n_pts * 40.0),
        "campose_kernel": ("mfma", (pool_pts * M + pool_lines * Ml) * 60.0),
        "triangulate_kernel": ("mfma", pool_pts * (M * 120.0 + 5 * M * 200.0)),
        "line_triangulate_kernel": ("mfma", pool_lines * Ml * 300.0),
        "jacobian_kernel": ("mfma", F * M * 3000.0),
        "line_jacobian_kernel": ("mfma", L * Ml * 6000.0),
    }
    up = update_work(F, 2 * M, 3, k, n)
    # (the gate runs as the tail of these launches when an entry has at most 32 projected rows and 128 columns: gate_core.hpp)
    gate_in = (2 * M - 3) <= 32 and k <= 128
    gate_pts = (up["chi2_t_kernel"] + up["chi2_gate_kernel"]) if gate_in else 0.0
    work["jacobian_nullspace_kernel"] = ("mfma", F * M * 3000.0 + up["nullspace_kernel"])
    work["tri_jacobian_nullspace_kernel"] = ("mfma", pool_pts * (M * 120.0 + 5 * M * 200.0) + F * M * 3000.0 + up["nullspace_kernel"] + gate_pts)
    if L > 0:
        ul = update_work(L, 2 * Ml, 6, kl, n)
        work["line_jacobian_nullspace_kernel"] = ("mfma", L * Ml * 6000.0 + ul["nullspace_kernel"])
        gate_l = (ul["chi2_t_kernel"] + ul["chi2_gate_kernel"]) if ((2 * Ml - 6) <= 32 and kl <= 128) else 0.0
        work["line_tri_jacobian_nullspace_kernel"] = ("mfma", pool_lines * Ml * 300.0 + L * Ml * 6000.0 + ul["nullspace_kernel"] + gate_l)
        for name, v in up.items():   # kernels both updates launch: the mean of the two launches
            up[name] = ul[name] if name == "nullspace_kernel" else 0.5 * (v + ul[name])
    for name, v in up.items():
        work[name] = ("hbm" if name in HBM_KERNELS else "mfma", v)
    return work
