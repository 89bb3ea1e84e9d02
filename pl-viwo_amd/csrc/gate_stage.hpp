// gate_stage.hpp — the kernel argument of the gate inside the Jacobian launches (gate_core.hpp); plain data, no device code.
#pragma once

namespace plv {

#define GATE_KMAX 128  // columns the fused gate holds (T in LDS: 32 x GATE_KMAX doubles)
#define GATE_MMAX 32   // projected rows per entry (two 16-row strips)
#define GATE_TLD (GATE_KMAX + 4)  // row stride of T in LDS: 16 rows x 4 k-slabs of an MFMA operand read land on distinct banks (a stride of 128 doubles put all 16 rows on one)

struct GateStage {  // kernel argument; on == 0: the launch ends with the projected blocks as before
  int on;
  int lds_off;      // byte offsets inside the launch's dynamic shared memory (set by the launcher): the gate's block (GateLds: S, the
  int ps_off;       // factorisation's block — it overlays scratch that is dead when the gate starts), the prior block Ps (upper
  int t_off;        // triangle, staged while the entry is triangulated: gate_stage_prior) and T = H' Ps
  const double *P;  // covariance, n x n, both triangles valid
  int ldp;
  double sigma2, chi2_mult, res_norm_gate;
  const double *q95;
  int q95_n, min_rows;
  double *chi2;             // [F]
  double *dec;              // (optional, plv_decision_trace) [F][3]: chi2, the threshold it was held against, the norm of the projected residual
  unsigned char *accepted;  // [F]
  int *acc_rows;            // [F]
  int *n_acc;               // counter of accepted entries (zero when the launch starts)
  int *n_acc_next;          // the counter the NEXT update will use: zeroed here (the two alternate, see plv_api.hip)
  double *stack;            // accepted rows [H' | r] of entry f at rows f * mp_max .. (col-major, lds)
  int lds, mp_max, stack_accepted_only;
  unsigned char *h_accepted;  // (optional) pinned copies of the verdicts + a second block per entry: the gate probe of the line update
  int *h_acc_rows;
  const unsigned char *probe_src;
  unsigned char *probe_dst;
  int probe_stride_a, probe_off_b, probe_stride_b;
};

}  // namespace plv
