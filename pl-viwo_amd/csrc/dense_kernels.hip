// dense_kernels.hip — tile-parallel products of the EKF-update half (fp64) and the launchers that chain
// them with the blocked factorisations of blocked_chol.hip.
//
// Measured on MI355X (profiles/r01): a Householder TSQR of the stacked Jacobian is latency-bound
// (99 reflector steps x 4 row chunks x 4 tree levels ~ 1600 dependent steps).  The default route therefore forms no factor of the
// measurements at all (REF: StateHelper::measurement_compress_inplace + EKFUpdate as one whitened step, launch_ekf_whitened below):
//     [G | g] = H^T [H | r] over the accepted rows      gram_direct_kernel  (v_mfma_f64_16x16x4_f64)
// (Rounds 2-3 factored G by a blocked Cholesky into the R the reference's Givens QR produces; that route squared the condition
// number and was removed in round 5 — the Householder TSQR of update_kernels.hip is the reference-equivalent route that remains.)
//
// The EKF step (REF: StateHelper::EKFUpdate, StateHelper.cpp:94-173):
//     Mt = H P[cols,:], S = Mt[:,cols] H^T + R      ekf_mt_kernel, ekf_s_kernel (update_kernels.hip)
//     W  = L^-1 [Mt | res],  S = L L^T               bchol_ekf_kernel
//     dC = W^T W (= K M^T),  dx = W^T y              ekf_dc_kernel
//     diagonal test, P -= dC                         ekf_commit_kernel
#include "mfma_tile.hpp"
#include "update_kernels.hpp"
#include "wave_ops.hpp"

namespace plv {


static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// G = A^T A in ONE launch, over the rows that exist: the stack holds a slot of mp_max rows per batch entry, and an entry the gate
// did not take (or an empty system: a pool candidate the selection skipped) is all zeros, so the chunk + reduce pair below multiplied
// mostly padding (rocprofv3, round 2: 2 MB moved for 0.3 MB of rows).  Here one workgroup owns an upper tile; its four waves share
// the accepted entries (acc_rows[f] > 0) and walk only their rows straight from the L2-resident stack — a wave's operand loads go
// down 16 columns, 16 k-steps in flight — and the four accumulators are added in a fixed order (deterministic).
__global__ void __launch_bounds__(256) gram_direct_kernel(const double *__restrict__ A, int lda, int nc, const int *__restrict__ acc_rows, int F,
                                                          int mp_max, double *__restrict__ G, const int *__restrict__ skip,
                                                          double *__restrict__ Gs, double *__restrict__ gv) {
  if (skip && *skip == 0) return;
  __shared__ double part[3][256];
  const int nt = (nc + 15) >> 4;
  int rem = blockIdx.x, ti = 0;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15;
  const double *ai = A + (size_t)min(ti * 16 + li, nc - 1) * lda;  // columns beyond nc only feed entries that are never stored
  const double *aj = A + (size_t)min(tj * 16 + li, nc - 1) * lda;
  d4 acc = {0, 0, 0, 0};
  int turn = 0;
  for (int f0 = 0; f0 < F; f0 += 64) {  // 64 entries' row counts per load; the accepted ones are dealt to the waves in turn
    const int fr = f0 + lane < F ? acc_rows[f0 + lane] : 0;
    unsigned long long mask = __ballot(fr > 0);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      if ((turn++ & 3) != wave) continue;
      const int rows = min(__shfl(fr, b, 64), mp_max);
      const double *af = ai + (size_t)(f0 + b) * mp_max, *bf = aj + (size_t)(f0 + b) * mp_max;
      auto fa = [&](int, int kk) { return af[kk]; };
      auto fb = [&](int kk, int) { return bf[kk]; };
      acc = mfma_tile_f64_pipe<8>(fa, fb, rows, acc);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) part[wave - 1][q * 64 + lane] = acc[q];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double v = ((acc[q] + part[0][q * 64 + lane]) + part[1][q * 64 + lane]) + part[2][q * 64 + lane];
      const int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + li;
      if (Gs) {  // whitened route: the information matrix H^T H as a full symmetric k x k block and H^T r as a vector
        const int k = nc - 1;
        if (i < k && j < k) Gs[(size_t)j * k + i] = v, Gs[(size_t)i * k + j] = v;
        if (i < k && j == k) gv[i] = v;
      } else if (i < nc && j < nc) {
        G[(size_t)j * nc + i] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// [dC | dx] = W[:, :n]^T [W[:, :n] | y]  (upper tiles; dC = K M^T of the reference, y = W[:, n]).
__global__ void __launch_bounds__(256) ekf_dc_kernel(const double *__restrict__ W, int ldw, int r, int n,
                                                     double *__restrict__ dC, int ldc, double *__restrict__ dx,
                                                     const double *__restrict__ P, int ldp, int *__restrict__ flag, const int *__restrict__ skip,
                                                     const double *__restrict__ dW, const double *__restrict__ C1,
                                                     const double *__restrict__ d0, const int *__restrict__ use_m) {
  // whitened route, its two forms (see "whitened update" below; use_m: device word, != 0 = factor form):
  //   whitened form  dC = dW - W^T W, dx = W^T y         dW = W0^T W0, formed ahead of time on the side stream (this kernel with
  //                                                      dx = P = null: products only)
  //   factor form    dC = C1 - W^T W, dx = d0 - W^T y    C1 = P[:, cols] G P[cols, :], d0 = P[:, cols] g (same tiles, same layout)
  if (skip && *skip == 0) return;
  const bool fm = use_m && (use_m[0] | use_m[1]) != 0;
  if (fm) dW = C1;
  const int tn = (n + 1 + 15) >> 4;
  const int ntri = tn * (tn + 1) / 2;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= ntri) return;
  int ti = 0, rem = wid;
  while (rem >= tn - ti) {
    rem -= tn - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const double *Wi = W + (size_t)min(ti * 16 + (lane & 15), n) * ldw;
  const double *Wj = W + (size_t)min(tj * 16 + (lane & 15), n) * ldw;
  d4 acc = {0, 0, 0, 0};
  auto fa = [&](int, int kk) { return Wi[kk]; };
  auto fb = [&](int kk, int) { return Wj[kk]; };
  acc = mfma_tile_f64_pipe<16>(fa, fb, r, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
    const double v = (dW && i < n && j < n) ? dW[(size_t)j * ldc + i] - acc[q] : acc[q];
    if (i < n && j < n) dC[(size_t)j * ldc + i] = v;
    if (dx && i < n && j == n) dx[i] = fm ? d0[i] - acc[q] : acc[q];
    // REF: StateHelper.cpp:143-152 — any P_ii - (K M^T)_ii < 0 rejects the update; the commit kernel reads the flag
    if (P && i == j && i < n && P[(size_t)i * ldp + i] - v < 0.0) atomicOr(flag, 1);
  }
}

// REF: StateHelper.cpp:143-156 — the update is rejected (nothing modified) when ekf_dc_kernel found a negative
// diagonal or the factorisation was not positive definite; else P.upper -= dC, mirrored.
// The last kernel of the update also mirrors the small result block (dx, flag, accepted, rows) into the caller's pinned host
// buffer, so that no copy command sits between the end of the chain and the host's wait.
__global__ void __launch_bounds__(1024) ekf_commit_kernel(double *__restrict__ P, int ldp, int n,
                                                         const double *__restrict__ dC, int ldc, int *__restrict__ flag,
                                                         const unsigned *__restrict__ mirror_src, unsigned *__restrict__ mirror_dst,
                                                         int mirror_words, const int *__restrict__ skip, double *__restrict__ dx,
                                                         const unsigned *__restrict__ mirror2_src, unsigned *__restrict__ mirror2_dst,
                                                         int mirror2_words, unsigned *done_word, unsigned done_val,
                                                         int *__restrict__ applied_out, const int *__restrict__ cap_words, int cap) {
  // done_word (pinned, optional; the launch then has ONE workgroup): behind the mirrors AND the covariance commit the workgroup stores
  // done_val there, and the host, spinning on the word, knows both the results and the covariance to be final
  const bool skipped = skip && *skip == 0;  // (the gate accepted nothing: no correction, the covariance stays)
  // cap_words (a speculative point batch, SpecSelectArgs::words + 3): [0] the pool was larger than the selection loop's cap, [1] the
  // candidates the Jacobian launch selected on their own verdicts.  When that count reaches the cap the loop would have stopped
  // somewhere inside the pool (REF CamHelper.cpp:651-653) and this batch is not the reference's: status bit 16, nothing is committed,
  // the host runs the update the long way.  (Read here, behind the kernel boundaries that complete the count.)
  const bool capped = cap_words && cap_words[0] != 0 && cap_words[1] >= cap;
  if (capped && blockIdx.x == 0) {
    if (threadIdx.x == 0) atomicOr(flag, 16);
    __syncthreads();  // (in front of the mirror of the status block below)
  }
  // applied_out: "this update changed the state" — StateHelper::EKFUpdate reached its mean update (:156-168): read by a launch that is
  // enqueued behind the update before the host has seen its result and applies dx to its own copy of the state (the chained line launch)
  if (applied_out && blockIdx.x == 0 && threadIdx.x == 0) *applied_out = (skipped || capped || *flag != 0) ? 0 : 1;
  if (blockIdx.x == 0) {
    if (skipped && dx) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) dx[i] = 0.0;
      __syncthreads();
    }
    if (mirror_dst)
      for (int i = threadIdx.x; i < mirror_words; i += blockDim.x) mirror_dst[i] = mirror_src[i];
  }
  if (mirror2_dst && blockIdx.x == gridDim.x - 1)  // (written by kernels launched earlier: complete)
    for (int i = threadIdx.x; i < mirror2_words; i += blockDim.x) mirror2_dst[i] = mirror2_src[i];
  if (!(skipped || capped || *flag != 0)) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
      int j = idx / n, i = idx - j * n;
      if (i <= j) {
        double v = P[(size_t)j * ldp + i] - dC[(size_t)j * ldc + i];
        P[(size_t)j * ldp + i] = v;
        P[(size_t)i * ldp + j] = v;
      }
    }
  }
  if (done_word) {  // (gridDim.x == 1)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done_word, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ========================================================================================== launchers
bool ekf_fast_fits(int r) { return r <= 192; }
static int launch_ekf_commit(plv_ctx *ctx, double *d_P, int n, int ldp, const double *dC, double *d_dx, int *d_flag, const void *mirror_src,
                             void *mirror_dst, size_t mirror_bytes);

// EKF update with the identity-border Cholesky.  Requires ekf_fast_fits(r).
int launch_ekf_fast(plv_ctx *ctx, double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh, const int *d_cols,
                    const double *d_res, const double *d_Rdiag, double *d_dx, int *d_flag, bool gathered, const void *mirror_src,
                    void *mirror_dst, size_t mirror_bytes) {
  int rc;
  const int ldm = r, ldw = r;
  if ((rc = ctx->d_Mt.reserve((size_t)r * (n + 1 + k) * 8)) || (rc = ctx->d_S.reserve((size_t)r * r * 8 * 2)) ||
      (rc = ctx->d_W.reserve((size_t)r * (n + 1) * 8)) || (rc = ctx->d_y.reserve((size_t)n * n * 8)))
    return rc;
  double *Mt = ctx->d_Mt.as<double>(), *S = ctx->d_S.as<double>(), *W = ctx->d_W.as<double>(), *dC = ctx->d_y.as<double>();
  launch_ekf_ms(ctx, d_P, n, ldp, d_H, r, k, ldh, d_cols, d_Rdiag, Mt, ldm, S, gathered, d_flag);
  if ((rc = launch_bchol_ekf(ctx, S, r, r, Mt, ldm, n, d_res, W, ldw, d_flag))) return rc;
  {
    ProfScope ps(ctx->prof, "ekf_dc_kernel", ctx->stream);
    int tn = cdiv(n + 1, 16);
    int waves = tn * (tn + 1) / 2;
    hipLaunchKernelGGL(ekf_dc_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, W, ldw, r, n, dC, n, d_dx, d_P, ldp, d_flag, ctx->skip_word,
                       (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (const int *)nullptr);
  }
  return launch_ekf_commit(ctx, d_P, n, ldp, dC, d_dx, d_flag, mirror_src, mirror_dst, mirror_bytes);
}

static int launch_ekf_commit(plv_ctx *ctx, double *d_P, int n, int ldp, const double *dC, double *d_dx, int *d_flag, const void *mirror_src,
                             void *mirror_dst, size_t mirror_bytes) {
  {
    ProfScope ps(ctx->prof, "ekf_commit_kernel", ctx->stream);
    // with a completion word: one workgroup of 1024 threads does mirrors + commit + word (14 passes over a 119 x 119 covariance)
    unsigned *dw = (mirror_dst && ctx->update_word_armed) ? (unsigned *)ctx->done_word(16) : nullptr;
    const dim3 grid(dw ? 1 : std::min(64, cdiv(n * n, 256))), block(dw ? 1024 : 256);
    hipLaunchKernelGGL(ekf_commit_kernel, grid, block, 0, ctx->stream, d_P, ldp, n, dC, n,
                       d_flag, (const unsigned *)mirror_src, (unsigned *)mirror_dst, (int)(mirror_bytes / 4), ctx->skip_word, d_dx,
                       (const unsigned *)(mirror_dst ? ctx->mirror2_src : nullptr), (unsigned *)(mirror_dst ? ctx->mirror2_dst : nullptr),
                       (int)((ctx->mirror2_bytes + 3) / 4), dw, ctx->update_seq, mirror_dst ? ctx->applied_word : nullptr,
                       mirror_dst ? ctx->cap_words : nullptr, ctx->cap);
    ctx->update_word_used = dw != nullptr;
    if (mirror_dst && ctx->applied_word) ctx->applied_used = true;
    if (mirror_dst && ctx->mirror2_dst) ctx->mirror2_taken = true;
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// The gate leaves only the accepted entries in the stack when the Gram-based routes follow (they walk the accepted entries); before a
// Householder factorisation of the WHOLE stack (the last resort of a whitened update that was run again, plv_api.hip RedoW) the slots
// of the other entries are cleared: a zero row changes no QR factor.
__global__ void __launch_bounds__(256) stack_zero_rejected_kernel(double *__restrict__ A, int lda, int nc, const int *__restrict__ acc_rows, int F, int mp_max) {
  const int f = blockIdx.x;
  if (f >= F) return;
  const int keep = max(acc_rows[f], 0);
  for (int idx = threadIdx.x; idx < nc * mp_max; idx += blockDim.x) {
    const int j = idx / mp_max, i = idx - j * mp_max;
    if (i >= keep) A[(size_t)j * lda + (size_t)f * mp_max + i] = 0.0;
  }
}
// The accepted rows of an accepted-only stack gathered into a dense matrix (dst, ldd rows per column, zeroed by the caller): entry f's
// acc_rows[f] rows go to rows [sum of the accepted rows before it, ...).  The Householder route then works on ~500 rows instead of
// the stack's F x mp_max = 3000 slots, most of them empty: three levels of the tree instead of six.
__global__ void __launch_bounds__(256) stack_compact_kernel(const double *__restrict__ A, int lda, int nc, const int *__restrict__ acc_rows, int F, int mp_max,
                                                            double *__restrict__ dst, int ldd, int *__restrict__ total_out) {
  __shared__ int s_off;
  const int f = blockIdx.x;
  if (total_out && f == 0 && threadIdx.x < 64) {  // (for a reader on the device: the rows dst holds, launch_tsqr's m_dev)
    int tot = 0;
    for (int g = threadIdx.x; g < F; g += 64) tot += max(acc_rows[g], 0);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
    if (threadIdx.x == 0) *total_out = tot;
  }
  const int rows = max(acc_rows[f], 0);
  if (rows == 0) return;
  if (threadIdx.x < 64) {
    int part = 0;
    for (int g = threadIdx.x; g < f; g += 64) part += max(acc_rows[g], 0);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    if (threadIdx.x == 0) s_off = part;
  }
  __syncthreads();
  const int off = s_off;
  if (off + rows > ldd) return;  // (cannot happen: the host sized dst from the same counts)
  for (int idx = threadIdx.x; idx < nc * rows; idx += blockDim.x) {
    const int j = idx / rows, i = idx - j * rows;
    dst[(size_t)j * ldd + off + i] = A[(size_t)j * lda + (size_t)f * mp_max + i];
  }
}
int launch_stack_compact(plv_ctx *ctx, const double *d_A, int lda, int nc, const int *d_acc_rows, int F, int mp_max, double *d_dst, int ldd, bool exact_rows,
                         int *d_total_out) {
  // (exact_rows: ldd is the number of accepted rows — every row of dst is written, nothing to clear)
  if (!exact_rows) PLV_HIP_CHECK(hipMemsetAsync(d_dst, 0, (size_t)ldd * nc * 8, ctx->stream));
  ProfScope ps(ctx->prof, "stack_compact_kernel", ctx->stream);
  hipLaunchKernelGGL(stack_compact_kernel, dim3(F), dim3(256), 0, ctx->stream, d_A, lda, nc, d_acc_rows, F, mp_max, d_dst, ldd, d_total_out);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_stack_zero_rejected(plv_ctx *ctx, double *d_A, int lda, int nc, const int *d_acc_rows, int F, int mp_max) {
  ProfScope ps(ctx->prof, "stack_zero_rejected_kernel", ctx->stream);
  hipLaunchKernelGGL(stack_zero_rejected_kernel, dim3(F), dim3(256), 0, ctx->stream, d_A, lda, nc, d_acc_rows, F, mp_max);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// ------------------------------------------------------------------------------------------ whitened update
// The compressed update without a factorisation of the measurement side (DESIGN.md "Whitened update").  With G = H^T H, g = H^T r
// (noise-normalised, as the reference's compression leaves them), Pc = P[cols, :] and Ps = P[cols, cols] = M M^T, S^-1 on the
// compressed system is the k x k matrix  B = I + M^T G M = Lb Lb^T  on the whitened one.  Two ways to the update from there:
//   whitened form   P' = P - W0^T W0 + V^T V,  dx = V^T v       W0 = M^-1 Pc,  [V | v] = Lb^-1 [W0 | M^T g]
//   factor form     P' = P - C1 + Z^T Z,       dx = d0 - Z^T z   GP = G Pc, C1 = Pc^T GP, d0 = Pc^T g, [Z | z] = Lb^-1 M^T [GP | g]
//                   (H^T S^-1 H = (I + G Ps)^-1 G = G - G M B^-1 M^T G)
// Neither divides by a pivot of G: directions the measurements do not observe (the gauge freedom of an MSCKF Jacobian) simply add
// nothing to B.  They differ in what costs digits.  The quantity at stake is the CONDITIONAL variance of a state given the ones
// before it — pivot x its variance, the pivot being that of the prior block's unit-diagonal factor: clone positions of this filter
// sit at 1e-9 .. 1e-8 (known to 1e-4 of the global position's uncertainty, which grows without bound), orientations at 1e-4.
// Relative to it an update loses (round 4, measured against the Householder route and the CPU oracle on the configs[2] / [3] drives;
// DESIGN 10.3 has the table):
//   * whitened form with every column of W0 obtained by substitution (round 3): eps / pivot^2.  The columns of the update's OWN states
//     are M^-1 Ps = M^T, which substitution delivers only to its own rounding, amplified by the factor's small pivots; after 24 s of the configs[2] drive the
//     position pivots had gone from +1e-8 to -1e-6 and the next update left the covariance indefinite;
//   * whitened form with those columns COPIED from the factor (prior_exact_cols_kernel; the default): eps / pivot — P[cols, cols] -
//     W0c^T W0c is then the factorisation's backward error and the block's posterior the sum of squares Vc^T Vc.  The library's
//     covariance pivots are the oracle's to two or three digits over 34 s of the configs[3] drive, the trajectories micrometres apart;
//   * factor form: eps x lambda / pivot (with a constant that grows with the window: covariance pivots negative from frame 130 of that
//     drive on, indefinite at frame 550) and eps x lambda^2 of the posterior variance itself, lambda = how much better than the prior
//     the measurements know a direction (B's diagonal - 1): 1e4 .. 1e5 in the first updates after an initialisation with the
//     intrinsics in the state (dx of the NEXT update off by 2e-5).  Only multiplies by M;
//   * the reference's P - K H P: eps / pivot.
// The prior factor decides on the device: the whitened form unless a pivot is DEAD (below PLV_PRIOR_TAU: M has a zero column, W0 a
// zero row, and the whitened form returned dC ten times P); then the factor form, and if B's diagonal exceeds PLV_WHITEN_LAMBDA_MAX
// (1e2) the update is handed to the reference's route (status bit 8 -> plv_api.hip RedoW).  Near-dependent pivots (below
// PLV_PRIOR_AMB) are counted for the record.  The prior factor, W0 and W0^T W0 only need the covariance, so they run on a side stream while
// the main stream triangulates, builds Jacobians and gates; the main chain after the gate is
// gram -> [B | GP, M^T GP] -> [factor B, solve | C1] -> dC -> commit  (four launches; "|": workgroups of the same launch, those of
// the factor form return at once when the update takes the whitened one).
// W0 = M^-1 Pc holds M^T in the columns of the update's own states, and the prior factor has M itself: those columns are COPIED from
// the factor instead of kept as the substitution left them (M^-1 Ps is M^T only to the substitution's rounding, amplified by the factor's small pivots: see "whitened update").
// Near-dependent pivots do not select the factor form — their count moves to n_near[4] for the record — only dead ones do.
__global__ void __launch_bounds__(64) prior_exact_cols_kernel(const double *__restrict__ Lt, int ldl, int k, const int *__restrict__ cols,
                                                              double *__restrict__ W0, int ldw, int *__restrict__ n_near) {
  const int j = blockIdx.x;
  double *dst = W0 + (size_t)cols[j] * ldw;
  const double *src = Lt + (size_t)j * ldl;
  for (int c = threadIdx.x; c < k; c += 64) dst[c] = c <= j ? src[c] : 0.0;
  if (j == 0 && threadIdx.x == 0) {  // near-dependent pivots do not select the factor form: their count moves to n_near[4] for the record
    n_near[4] = n_near[0];
    n_near[0] = 0;
  }
}

int launch_prior_factor(plv_ctx *ctx, hipStream_t st, const double *d_P, int n, int ldp, const int *d_cols, int k) {
  int rc;
  if ((rc = ctx->d_Lt.reserve((size_t)k * k * 8)) || (rc = ctx->d_W0.reserve((size_t)k * (n + 1) * 8)) ||
      (rc = ctx->d_dW.reserve((size_t)n * n * 8)) || (rc = ctx->d_prior_near.reserve(64)))
    return rc;
  if ((rc = launch_bchol_prior(ctx, st, d_P, ldp, n, d_cols, k, ctx->d_Lt.as<double>(), k, ctx->d_W0.as<double>(), k, ctx->d_prior_near.as<int>()))) return rc;
  {  // the columns of W0 that belong to the update's own states, exact (copied from the factor); near-dependent pivots only counted
    ProfScope ps(ctx->prof, "prior_exact_cols_kernel", st);
    hipLaunchKernelGGL(prior_exact_cols_kernel, dim3(k), dim3(64), 0, st, ctx->d_Lt.as<double>(), k, k, d_cols, ctx->d_W0.as<double>(), k,
                       ctx->d_prior_near.as<int>());
  }
  {
    ProfScope ps(ctx->prof, "prior_gain_kernel", st);
    const int tn = cdiv(n + 1, 16), waves = tn * (tn + 1) / 2;
    hipLaunchKernelGGL(ekf_dc_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, st, ctx->d_W0.as<double>(), k, k, n, ctx->d_dW.as<double>(), n,
                       (double *)nullptr, (const double *)nullptr, 0, (int *)nullptr, (const int *)nullptr, (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, (const int *)nullptr);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// Information matrix of the accepted rows of the stack (d_Gs k x k, d_gv k), on the main stream.
int launch_gram_information(plv_ctx *ctx, const double *d_A, int lda, int nc, const int *d_acc_rows, int F, int mp_max) {
  const int k = nc - 1, nt = cdiv(nc, 16), ntri = nt * (nt + 1) / 2;
  int rc;
  if ((rc = ctx->d_Gs.reserve(((size_t)k * k + k) * 8))) return rc;
  ProfScope ps(ctx->prof, "gram_direct_kernel", ctx->stream);
  hipLaunchKernelGGL(gram_direct_kernel, dim3(ntri), dim3(256), 0, ctx->stream, d_A, lda, nc, d_acc_rows, F, mp_max, (double *)nullptr,
                     ctx->skip_word, ctx->d_Gs.as<double>(), ctx->d_Gs.as<double>() + (size_t)k * k);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// The main-stream part after launch_gram_information; the caller has made the stream wait for launch_prior_factor's end.
int launch_ekf_whitened(plv_ctx *ctx, double *d_P, int n, int ldp, int k, const int *d_cols, double *d_dx, int *d_flag, const void *mirror_src,
                        void *mirror_dst, size_t mirror_bytes) {
  int rc;
  if ((rc = ctx->d_Mt.reserve((size_t)k * 8)) || (rc = ctx->d_S.reserve((size_t)k * k * 8)) || (rc = ctx->d_W.reserve((size_t)k * (n + 1) * 8)) ||
      (rc = ctx->d_y.reserve((size_t)n * n * 8)))
    return rc;
  double *cv = ctx->d_Mt.as<double>(), *B = ctx->d_S.as<double>(), *V = ctx->d_W.as<double>(), *dC = ctx->d_y.as<double>();
  const double *Gs = ctx->d_Gs.as<double>(), *gv = Gs + (size_t)k * k;
  if ((rc = ctx->d_C1.reserve(((size_t)n * n + n) * 8)) || (rc = ctx->d_Y0.reserve((size_t)k * (n + 1) * 8)) || (rc = ctx->d_GP.reserve((size_t)k * n * 8)))
    return rc;
  double *Y0 = ctx->d_Y0.as<double>(), *C1 = ctx->d_C1.as<double>(), *d0 = C1 + (size_t)n * n, *GP = ctx->d_GP.as<double>();
  const int *use_m = ctx->d_prior_near.as<int>();  // (written by the prior factor; the caller has joined the side stream)
  if (plv::knob(plv::PLV_KNOB_FORCE_FACTOR_FORM)) {  // (tools / tests: every update takes the factor form)
    static const int one[2] = {1, 0};
    PLV_HIP_CHECK(hipMemcpyAsync(ctx->d_prior_near.p, one, 8, hipMemcpyHostToDevice, ctx->stream));
  }
  launch_whiten_b(ctx, ctx->d_Lt.as<double>(), k, Gs, gv, cv, B, d_flag, d_P, ldp, n, d_cols, Y0, GP, d0, use_m);
  const WhitenC1Args wc{d_P, ldp, d_cols, GP, C1, Y0, use_m};
  if ((rc = launch_bchol_ekf(ctx, B, k, k, ctx->d_W0.as<double>(), k, n, cv, V, k, d_flag, &wc))) return rc;
  {
    ProfScope ps(ctx->prof, "ekf_dc_kernel", ctx->stream);
    const int tn = cdiv(n + 1, 16), waves = tn * (tn + 1) / 2;
    hipLaunchKernelGGL(ekf_dc_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, V, k, k, n, dC, n, d_dx, d_P, ldp, d_flag, ctx->skip_word,
                       ctx->d_dW.as<double>(), C1, d0, use_m);
  }
  return launch_ekf_commit(ctx, d_P, n, ldp, dC, d_dx, d_flag, mirror_src, mirror_dst, mirror_bytes);
}

}  // namespace plv
