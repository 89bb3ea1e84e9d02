"""Algorithmic work per kernel launch (bytes for HBM-class kernels, flops for the dense fp64 ones): the per-unit figures of
SURVEY.md §8(d), restated in DESIGN.md §4, times the units one launch processes.  Shared by bench.py and tests/bench_chain.py."""

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TF = 78.6    # vendor FP64 matrix figure (v_mfma_f64_16x16x4_f64)
HBM_KERNELS = {"gram_reduce_kernel", "gather_cov_kernel", "ekf_commit_kernel"}


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
