// update_kernels.hpp — launcher declarations of update_kernels.hip (product code).
#pragma once
#include "plv_ctx.hpp"

namespace plv {

struct Chi2Args {
  const double *P;
  int ldp;
  int k, ld, fdim_off;   // fdim_off: rows already removed by the nullspace (mp = rows - fdim_off)
  const int *rows;
  const double *Hx, *res;
  const int *cols;
  double sigma2;
  double *chi2;
  double *dec;            // (optional, plv_decision_trace) [F][3]: chi2, threshold, norm of the projected residual
  // gate + stack (optional)
  double *stack;
  int lds, mp_max;
  double chi2_mult, res_norm_gate;
  const double *q95;      // device table, index = dof
  int q95_n;
  int min_rows;
  unsigned char *accepted;
  int *acc_rows;          // projected rows of each accepted feature (0 when rejected)
  int *n_acc;             // (optional) number of accepted features: zeroed by chi2_t_kernel, counted by chi2_gate_kernel
  int stack_accepted_only;  // the consumer of the stack walks acc_rows (gram_direct_kernel): a rejected entry's slot is left unwritten
  // filled by launch_chi2
  int F;
  const double *Ps;       // dense P[cols, cols] (row-major k x k) from gather_cov_kernel
  double *T;              // [F][k][ld] H' * Ps
  // gate probe (optional): every workgroup also leaves its verdict and a slice of a second block (the triangulation result of its
  // candidate) in pinned host memory, so that the host can read the gate's outcome as soon as the launch has finished — no copy
  // command, no further launch
  unsigned char *h_accepted;          // [F]
  int *h_acc_rows;                    // [F]
  const unsigned char *probe_src;     // block f copies [f * stride_a, +stride_a) and [off_b + f * stride_b, +stride_b)
  unsigned char *probe_dst;
  int probe_stride_a, probe_off_b, probe_stride_b;
};

int launch_nullspace(plv_ctx *ctx, int F, int fdim, int k, int ld, const int *d_rows, double *d_Hf, double *d_Hx,
                     double *d_res, const double *d_P = nullptr, int n = 0, int ldp = 0, const int *d_cols = nullptr,
                     int shift = -1 /* rows dropped from Hx / res on write-back; -1 = fdim */);
struct GatherArgs;
int gather_args(plv_ctx *ctx, const double *d_P, int n, int ldp, const int *d_cols, int k, GatherArgs &g);
int launch_gather_cov(plv_ctx *ctx, const double *d_P, int n, int ldp, const int *d_cols, int k);
int launch_chi2(plv_ctx *ctx, int F, const Chi2Args &a, int max_mp);
// m_dev (nullable, device word): the rows of d_A that hold anything — d_A is a gathered stack whose row count only the device knows
// (launch_stack_compact's d_total_out); m is then the bound the launch is sized for, rows beyond *m_dev are not read
int launch_tsqr(plv_ctx *ctx, double *d_A, int lda, int m, int nc, double *d_tmp, size_t tmp_elems, double **result,
                int *ld_out, const int *m_dev = nullptr);
int launch_ekf(plv_ctx *ctx, double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh, const int *d_cols,
               const double *d_res, const double *d_Rdiag, double *d_dx, int *d_flag, bool gathered = false);

void launch_ekf_ms(plv_ctx *ctx, const double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh,
                   const int *d_cols, const double *d_Rdiag, double *Mt, int ldm, double *S, bool gathered = false,
                   int *d_flag = nullptr);
// `gathered`: launch_gather_cov already ran for this (P, cols) and P has not changed since
// dense_kernels.hip
bool ekf_fast_fits(int r);
int launch_ekf_fast(plv_ctx *ctx, double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh, const int *d_cols,
                    const double *d_res, const double *d_Rdiag, double *d_dx, int *d_flag, bool gathered = false, const void *mirror_src = nullptr,
                    void *mirror_dst = nullptr, size_t mirror_bytes = 0);

// whitened route (dense_kernels.hip; DESIGN.md "Whitened update")
void launch_whiten_b(plv_ctx *ctx, const double *Lt, int k, const double *Gs, const double *gv, double *cv, double *B, int *d_flag,
                     const double *d_P, int ldp, int n, const int *d_cols, double *Y0, double *GP, double *d0, const int *use_m);
int launch_stack_compact(plv_ctx *ctx, const double *d_A, int lda, int nc, const int *d_acc_rows, int F, int mp_max, double *d_dst, int ldd, bool exact_rows = false,
                         int *d_total_out = nullptr /* device word: the rows gathered */);
int launch_stack_zero_rejected(plv_ctx *ctx, double *d_A, int lda, int nc, const int *d_acc_rows, int F, int mp_max);
int launch_prior_factor(plv_ctx *ctx, hipStream_t st, const double *d_P, int n, int ldp, const int *d_cols, int k);
int launch_gram_information(plv_ctx *ctx, const double *d_A, int lda, int nc, const int *d_acc_rows, int F, int mp_max);
int launch_ekf_whitened(plv_ctx *ctx, double *d_P, int n, int ldp, int k, const int *d_cols, double *d_dx, int *d_flag, const void *mirror_src, void *mirror_dst,
                        size_t mirror_bytes);
int launch_bchol_prior(plv_ctx *ctx, hipStream_t st, const double *d_P, int ldp, int n, const int *d_cols, int k, double *d_Lt, int ldl,
                       double *d_W0, int ldw, int *d_n_near);

// blocked_chol.hip
struct WhitenC1Args {  // (whitened update: the tiles of C1 = P[:, cols] GP ride along in spare workgroups: blocked_chol.hip WhitenC1)
  const double *P;
  int ldp;
  const int *cols;
  const double *GP;
  double *C1;
  const double *Y0;   // the borders of the factor form (d_Mt of the call: those of the whitened form)
  const int *use_m;   // device word: != 0 = factor form
};
int launch_bchol_ekf(plv_ctx *ctx, const double *d_S, int lds_, int r, const double *d_Mt, int ldm, int n,
                     const double *d_res, double *d_W, int ldw, int *d_flag, const WhitenC1Args *wc = nullptr);

}  // namespace plv
