// blocked_chol.hip — the two factorisation kernels of the update path on top of blocked_chol.hpp:
// compression (Gram matrix -> [R z]) and the EKF solve (S, [M; res^T] -> W).
#include "blocked_chol.hpp"
#include "update_kernels.hpp"

namespace plv {

// pivots of the unit-diagonal prior block below this are exact dependencies (measured on the replay batches: dead pivots <= 1e-14,
// the smallest live one 1.5e-7; DESIGN.md "Whitened update")
#define PLV_PRIOR_TAU 2e-13
// Pivots of the unit-diagonal prior block below this are NEAR dependencies (counted for the record, n_near[4]): clone positions reach
// 1e-8 within seconds of a drive — the global position's variance grows without bound while a clone's position given its neighbour
// stays at tenths of a millimetre — orientations 1e-4.  The columns of W0 that belong to the update's own states are copied from the
// factor (prior_exact_cols_kernel), which is what keeps the whitened form exact there.
// PLV_WHITEN_LAMBDA_MAX: the factor form's own limit (B's largest diagonal entry beyond which it hands the update to Householder).
#define PLV_PRIOR_AMB 1e-4
#define PLV_WHITEN_LAMBDA_MAX 1e2
// ------------------------------------------------------------------------------------------ prior factor
// The whitened route of the compressed update (DESIGN.md "Whitened update"): Ps = P[cols, cols] = Lp Lp^T with the rows
// P[:, cols] as borders,  W0^T = P[:, cols] Lp^-T.  Ps is factored after scaling to unit diagonal; a pivot below tau marks a state that
// is an exact linear function of earlier ones (the IMU pose and the clone just taken of it): its column of Lp is zero and it carries
// no weight, which is what the semi-definite prior says.  Outputs in the layouts the EKF kernels already use:
//   Lt (k x k, Lt(c, i) = Lp(i, c) at Lt[i * ldl + c]) — the "H" operand of ekf_ms_kernel;  W0 (border b = state b at W0[b * ldw + c]).
struct PriorOps {
  static constexpr bool kStoreL = true;
  const double *P;  // n x n, both triangles valid
  int ldp, n;
  const int *cols;
  int k;
  const double *sc;
  double *Lt;
  int ldl;
  double *W0;
  int ldw;
  double dval;
  double *scw;
  bool store_l;  // one workgroup writes the factor, every workgroup its own border strip
  int *n_near;   // (that workgroup) counts the near-dependent pivots here: a diagonal entry l_cc of the unit-diagonal factor with
                 // 0 < l_cc^2 < PLV_PRIOR_AMB.  Counted where the factor is stored, not inside the pivot chain: that code is as
                 // fragile under hipcc 7.2 as the panel loop (an extra test in it gave a wrong factor for ten tiles)
  __device__ __forceinline__ double sym_raw(int i, int c) const {
    return P[(size_t)cols[min(i, k - 1)] * ldp + cols[min(c, k - 1)]];
  }
  __device__ __forceinline__ double border_raw(int b, int c) const { return P[(size_t)cols[min(c, k - 1)] * ldp + min(b, n - 1)]; }
  __device__ __forceinline__ void scales_ready() const {
    const int j = threadIdx.x;
    if (j < 192) {
      const bool ok = j < k && dval > 0.0;
      const double rt = sqrt(ok ? dval : 1.0);
      scw[j] = ok ? 1.0 / rt : 0.0;
      scw[192 + j] = ok ? rt : 0.0;
    }
    __syncthreads();
  }
  __device__ __forceinline__ double sym_fix(int i, int c, double g) const {
    const bool pad = i >= k || c >= k;
    return pad ? (i == c ? 1.0 : 0.0) : g * sc[min(i, k - 1)] * sc[min(c, k - 1)];
  }
  __device__ __forceinline__ double border_fix(int b, int c, double g) const { return (b < n && c < k) ? g * sc[min(c, k - 1)] : 0.0; }
  __device__ __forceinline__ void store_sym(int i, int c, double l) const {
    if (store_l && i < k && c <= i) {
      if (c == i && n_near && l * l < PLV_PRIOR_AMB) atomicAdd(n_near + (l == 0.0 ? 1 : 0), 1);  // [0] near, [1] dead  // (a dead pivot — l = 0, below PLV_PRIOR_TAU — counts: round 4 met whitened-form updates with one that came back with dC ten times P)
      Lt[(size_t)i * ldl + c] = l * sc[192 + i];
      if (c < i) Lt[(size_t)c * ldl + i] = 0.0;
    }
  }
  __device__ __forceinline__ void store_border(int b, int c, double v) const {
    if (W0 && b < n && c < k) W0[(size_t)b * ldw + c] = v;
  }
};

template <int NT>
__global__ void __launch_bounds__(64 * (NT + 1)) bchol_prior_kernel(const double *__restrict__ P, int ldp, int n, const int *__restrict__ cols_g,
                                                                   int k, double *__restrict__ Lt, int ldl, double *__restrict__ W0, int ldw,
                                                                   int *__restrict__ n_near /* near-dependent pivots (workgroup 0 writes it) */) {
  __shared__ BcLds lds;
  __shared__ double sc[384];
  __shared__ int scols[192];
  // the column map may sit in pinned host memory (the one-submission updates start this kernel before their upload has run): it is
  // read once, coalesced, and indexed from LDS afterwards
  if (threadIdx.x < 192) scols[threadIdx.x] = cols_g[min((int)threadIdx.x, k - 1)];
  __syncthreads();
  const int *cols = scols;
  const int jd = cols[min((int)threadIdx.x, k - 1)];
  PriorOps ops{P, ldp, n, cols, k, sc, Lt, ldl, W0, ldw, P[(size_t)jd * ldp + jd], sc, blockIdx.x == 0, blockIdx.x == 0 ? n_near : nullptr};
  if (threadIdx.x == 0) {
    if (n_near && blockIdx.x == 0) n_near[0] = n_near[1] = 0;
    lds.bad = 0;
    lds.step_flag = 0;
    lds.rs_flag = 0;
    lds.n_amb = 0;
  }
  // (no barrier needed here: scales_ready() has one before any wave reads the flags)
  blocked_chol<NT>(ops, lds, k, W0 ? n : 1, PLV_PRIOR_TAU, (int)blockIdx.x);  // (W0 null: the factor alone — one border the kernel forms and drops)
}

// ------------------------------------------------------------------------------------------ EKF
struct EkfOps {
  static constexpr bool kStoreL = false;
  const double *S;
  int lds_, r;
  const double *Mt;
  int ldm, n;
  const double *res;
  double *W;
  int ldw;
  __device__ __forceinline__ double sym_raw(int i, int c) const {
    const int hi = min(max(i, c), r - 1), lo = min(min(i, c), r - 1);
    return S[(size_t)hi * lds_ + lo];  // REF: S.selfadjointView<Upper>()
  }
  __device__ __forceinline__ double border_raw(int b, int c) const {
    const double *src = b < n ? Mt + (size_t)b * ldm : res;  // clamped, select afterwards: no branch
    return src[min(c, r - 1)];
  }
  __device__ __forceinline__ void scales_ready() const {}
  __device__ __forceinline__ double sym_fix(int i, int c, double g) const {
    return (i >= r || c >= r) ? (i == c ? 1.0 : 0.0) : g;
  }
  __device__ __forceinline__ double border_fix(int b, int c, double g) const { return (c < r && b <= n) ? g : 0.0; }
  __device__ __forceinline__ void store_sym(int, int, double) const {}
  __device__ __forceinline__ void store_border(int b, int c, double v) const {
    if (b <= n && c < r) W[(size_t)b * ldw + c] = v;
  }
};

// Whitened update (dense_kernels.hip "whitened update"): the workgroups from `first` on do not factor anything — each of their waves
// forms one 16 x 16 tile (on or above the diagonal) of  C1 = P[:, cols] GP  from the GP = G P[cols, :] the launch before stored
// (update_kernels.hip WhitenPanels).  C1 is wanted by the launch after this one; here it costs nothing, the factorisation is a chain.
struct WhitenC1 {
  const double *P;
  int ldp, k;
  const int *cols;
  const double *GP;  // GP[b * k + c]
  double *C1;        // C1[b * n + b']
  const double *Y0;  // factor form: the borders the factorisation solves for instead of Mt
  const int *use_m;  // device word: != 0 = this update takes the factor form (else these workgroups have nothing to do)
  double lam_max;    // PLV_WHITEN_LAMBDA_MAX (a launch parameter so that tools can move it: PLV_WHITEN_LAMBDA_MAX in the environment)
  int first;         // < 0: none
};

// W = L^-1 [Mt | res] with S = L L^T (upper triangle of S valid).  flag |= 2 when S is not PD.
template <int NT>
__global__ void __launch_bounds__(64 * (NT + 1)) bchol_ekf_kernel(const double *__restrict__ S, int lds_, int r,
                                                         const double *__restrict__ Mt, int ldm, int n,
                                                         const double *__restrict__ res, double *__restrict__ W, int ldw,
                                                         int *__restrict__ flag, const int *__restrict__ skip, WhitenC1 c1) {
  __shared__ BcLds lds;
  if (skip && *skip == 0) return;
  const bool factor_form = c1.first >= 0 && (c1.use_m[0] | c1.use_m[1]) != 0;  // (near or dead pivots in the prior factor)
  if (c1.first >= 0 && (int)blockIdx.x >= c1.first) {
    __shared__ int scols[192];
    if ((int)blockIdx.x == c1.first && threadIdx.x < 64) {
      // The factor form loses eps x lambda^2 of the posterior covariance in a direction the measurements know lambda times better than
      // the prior (B's diagonal is 1 + lambda): beyond PLV_WHITEN_LAMBDA_MAX the update is left to the reference's route (status bit 8,
      // plv_api.hip RedoW).  Only with a near-dependent prior: otherwise the whitened form runs, which loses nothing there.
      double m = 0.0;
      for (int i = threadIdx.x; i < r; i += 64) m = fmax(m, S[(size_t)i * lds_ + i]);
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
      if (threadIdx.x == 0 && factor_form && m > c1.lam_max) atomicOr(flag, 8);
      if (threadIdx.x == 0) ((double *)(c1.use_m + 2))[0] = m;  // (for the record: plv_whiten_stats)
    }
    if (!factor_form) return;
    if (threadIdx.x < 192) scols[threadIdx.x] = c1.cols[min((int)threadIdx.x, c1.k - 1)];
    __syncthreads();
    const int tn = (n + 15) >> 4, ntri = tn * (tn + 1) / 2;
    const int tile = ((int)blockIdx.x - c1.first) * (NT + 1) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 15;
    if (tile >= ntri) return;
    int ti = 0, rem = tile;
    while (rem >= tn - ti) {
      rem -= tn - ti;
      ++ti;
    }
    const int tj = ti + rem;
    const int bq = min(ti * 16 + li, n - 1);
    const double *Gq = c1.GP + (size_t)min(tj * 16 + li, n - 1) * c1.k;
    d4 acc = {0, 0, 0, 0};
    auto fa = [&](int, int kk) { return c1.P[(size_t)scols[kk] * c1.ldp + bq]; };
    auto fb = [&](int kk, int) { return Gq[kk]; };
    acc = mfma_tile_f64_pipe<16>(fa, fb, c1.k, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bp = ti * 16 + (lane >> 4) + 4 * q, b = tj * 16 + li;
      if (bp < n && b < n) c1.C1[(size_t)b * n + bp] = acc[q];
    }
    return;
  }
  EkfOps ops{S, lds_, r, factor_form ? c1.Y0 : Mt, ldm, n, res, W, ldw};
  if (threadIdx.x == 0) {
    lds.bad = 0;
    lds.step_flag = 0;
    lds.rs_flag = 0;
  }
  __syncthreads();
  blocked_chol<NT>(ops, lds, r, n + 1, 0.0, (int)blockIdx.x);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && lds.bad) atomicOr(flag, 2);
}

#ifndef PLV_BCHOL_NO_LAUNCHERS
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }


int launch_bchol_ekf(plv_ctx *ctx, const double *d_S, int lds_, int r, const double *d_Mt, int ldm, int n,
                     const double *d_res, double *d_W, int ldw, int *d_flag, const WhitenC1Args *wc) {
  if (r > 192) return PLV_E_CAPACITY;
  const int strips = cdiv(n + 1, 16);  // one border strip per workgroup; every workgroup factors S itself
  const int nt_waves = r <= 32 ? 3 : r <= 64 ? 5 : r <= 112 ? 8 : r <= 128 ? 9 : r <= 160 ? 11 : 13;
  const int tn = cdiv(n, 16), c1_groups = wc ? cdiv(tn * (tn + 1) / 2, nt_waves) : 0, groups = strips + c1_groups;
  const double lam_max = PLV_WHITEN_LAMBDA_MAX;
  const WhitenC1 c1 = wc ? WhitenC1{wc->P, wc->ldp, r, wc->cols, wc->GP, wc->C1, wc->Y0, wc->use_m, lam_max, strips}
                         : WhitenC1{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, -1};
  ProfScope ps(ctx->prof, "bchol_ekf_kernel", ctx->stream);
  if (r <= 32)
    hipLaunchKernelGGL(bchol_ekf_kernel<2>, dim3(groups), dim3(64 * 3), 0, ctx->stream, d_S, lds_, r, d_Mt, ldm, n, d_res, d_W,
                       ldw, d_flag, ctx->skip_word, c1);
  else if (r <= 64)
    hipLaunchKernelGGL(bchol_ekf_kernel<4>, dim3(groups), dim3(64 * 5), 0, ctx->stream, d_S, lds_, r, d_Mt, ldm, n, d_res, d_W,
                       ldw, d_flag, ctx->skip_word, c1);
  else if (r <= 112)
    hipLaunchKernelGGL(bchol_ekf_kernel<7>, dim3(groups), dim3(64 * 8), 0, ctx->stream, d_S, lds_, r, d_Mt, ldm, n, d_res, d_W,
                       ldw, d_flag, ctx->skip_word, c1);
  else if (r <= 128)
    hipLaunchKernelGGL(bchol_ekf_kernel<8>, dim3(groups), dim3(64 * 9), 0, ctx->stream, d_S, lds_, r, d_Mt, ldm, n, d_res, d_W,
                       ldw, d_flag, ctx->skip_word, c1);
  else if (r <= 160)
    hipLaunchKernelGGL(bchol_ekf_kernel<10>, dim3(groups), dim3(64 * 11), 0, ctx->stream, d_S, lds_, r, d_Mt, ldm, n, d_res, d_W,
                       ldw, d_flag, ctx->skip_word, c1);
  else
    hipLaunchKernelGGL(bchol_ekf_kernel<12>, dim3(groups), dim3(64 * 13), 0, ctx->stream, d_S, lds_, r, d_Mt, ldm, n, d_res, d_W,
                       ldw, d_flag, ctx->skip_word, c1);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}
// Prior factor of the whitened route on stream `st` (a side stream: it only needs the covariance).
int launch_bchol_prior(plv_ctx *ctx, hipStream_t st, const double *d_P, int ldp, int n, const int *d_cols, int k, double *d_Lt, int ldl,
                       double *d_W0, int ldw, int *d_n_near) {
  if (k > 192) return PLV_E_CAPACITY;
  const int groups = d_W0 ? cdiv(n, 16) : 1;
  ProfScope ps(ctx->prof, "bchol_prior_kernel", st);
#define PLV_PRIOR_LAUNCH(NT) \
  hipLaunchKernelGGL(bchol_prior_kernel<NT>, dim3(groups), dim3(64 * (NT + 1)), 0, st, d_P, ldp, n, d_cols, k, d_Lt, ldl, d_W0, ldw, d_n_near)
  if (k <= 32)
    PLV_PRIOR_LAUNCH(2);
  else if (k <= 64)
    PLV_PRIOR_LAUNCH(4);
  else if (k <= 112)
    PLV_PRIOR_LAUNCH(7);
  else if (k <= 128)
    PLV_PRIOR_LAUNCH(8);
  else if (k <= 160)
    PLV_PRIOR_LAUNCH(10);
  else
    PLV_PRIOR_LAUNCH(12);
#undef PLV_PRIOR_LAUNCH
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}
#endif  // PLV_BCHOL_NO_LAUNCHERS

}  // namespace plv
