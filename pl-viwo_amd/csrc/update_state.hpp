// update_state.hpp — per-ctx state of the update side shared by plv_api.hip and jacobian_api.hip.
#pragma once
#include <chrono>
#include <vector>

#include "plv_ctx.hpp"

struct plv_ctx_update_state {
  plv::DevBuf q95;
  plv::DevBuf result;   // [dx: max_n doubles][flag: int + pad][accepted: bytes]
  plv::DevBuf covck;    // covariance checkpoint
  int covck_n = 0;      // dimension of the checkpointed covariance (a rollback restores it together with the data)
  plv::DevBuf bHf, bHx, bres, brows, bcols, bwork;  // staged feature batch ([Hf|Hx|res] in bHf) + working copy
  plv::DevBuf bcols_l;                               // the line batch's column map (same reason as plv_ctx::d_stack_l)
  plv::DevBuf &bcols_of(int fdim) { return fdim == 6 ? bcols_l : bcols; }
  int bF = 0, bfdim = 0, bk = 0, bld = 0, bmaxrows = 0;
  bool b_on_device_rows = false;  // rows[] produced on the device (plv_build_jacobians_resident)
  bool b_single_use = false;      // batch is rebuilt every frame: consume it in place, no working copy
  bool b_projected = false;       // the batch is already null-space projected (jacobian_nullspace_kernel)
  unsigned long long b_gather_token = 0;  // plv_ctx::gather_stamp right after the gathers that rode on the batch's launch (0: none)
  std::vector<int> brows_host;
  // measurement compression (plv_update_compression_mode): 0 = whitened update (information matrix + factor of the prior block; no
  // factor of the measurements), 1 = Householder TSQR on the stacked rows, 2 = Gram + Cholesky first, redone through the Householder
  // route when its factorisation reports pivots it could not resolve, 3 = Gram matrix + blocked Cholesky (the round-2 default)
  int compress_mode = 0;
  int last_route = 0;       // of the last update: 0 none / not compressed, 1 Gram + Cholesky, 2 Householder, 3 Gram vetoed and redone by Householder, 4 whitened,
                            // 5 whitened came back rejected / withheld and was run again by Householder reflections (redo_w)
  int last_ambiguous = 0;   // pivots the last Gram factorisation could not tell from zero
  // Status block of a LINE update (round 4): the chained line launch reads the point update's dx from `result` while its own gate
  // writes verdicts — and a larger line batch may make its block grow — so the two measurement kinds keep separate blocks, each with
  // its own pair of alternating accepted-entry counters.
  plv::DevBuf result_l;
  struct AccWords {
    int word = 2;             // word of the status block the fused gate's next update counts its accepted entries in (1 or 2, alternating)
    const void *z_ptr = nullptr;  // (z_ptr, z_n, z_word): the counter word known to be zero on the stream — the gate of the update before
    int z_n = 0, z_word = 0;      // zeroed it; anything else (first use, the block moved or grew, cov_n changed, a launch declined the gate) is zeroed by a memset
  } acc[2];
  plv::DevBuf &result_of(int fdim) { return fdim == 6 ? result_l : result; }
  AccWords &acc_of(int fdim) { return acc[fdim == 6 ? 1 : 0]; }
  int acc_word_used = 1;   // the word the last launched update's chain read as its skip word (1: chi2_gate_kernel's)
  // A whitened update that comes back rejected (negative diagonal of P', B not positive definite) is not final: its formulas are
  // not the reference's and lose digits elsewhere (dense_kernels.hip "whitened update": eps x how much better than the prior the
  // measurements know a direction — large for the first update after an initialisation).  The stacked rows are still on the device
  // and the update is run again through a Householder compression and the EKF step on R (last_route 5) before the verdict is reported.
  struct RedoW {
    bool armed = false;
    int Mtot = 0, k = 0, n = 0, F = 0, mp_max = 0, fdim = 3;
    size_t tmp_elems = 0, rb = 0;
    double *d_dx = nullptr;
    int *d_flag = nullptr, *d_acc_rows = nullptr;
  } redo_w;
  int pending_F = 0;  // features of a launched, not yet collected plv_msckf_update_resident_launch
  unsigned long long done_stamp = 0;  // plv_ctx::gather_stamp when done_ev was recorded
  unsigned word_seq = 0;              // nonzero: the launched update's last kernel stores this number to plv_ctx::done_word(16)
  hipEvent_t done_ev = nullptr;  // behind the update's last command: the wait does not cover what the caller enqueues after the launch
  // optional hipGraph replay of the update launch sequence (plv_update_graph_mode): key = every pointer / size / scalar a
  // kernel argument is made of; first sight of a key runs eagerly (sizes every buffer), the second captures, later ones replay
  struct GraphKey {
    const void *P, *Hf, *rows, *cols, *result, *hpin;
    const void *m2_src, *m2_dst, *skip;  // the second mirror block of ekf_commit_kernel and the skip word are kernel arguments too
    size_t m2_bytes;
    int F, fdim, k, ld, n, mp_max;
    double s2, cm, rg;
    unsigned long long epoch;
    bool operator==(const GraphKey &o) const {
      return P == o.P && Hf == o.Hf && rows == o.rows && cols == o.cols && result == o.result && hpin == o.hpin &&
             m2_src == o.m2_src && m2_dst == o.m2_dst && skip == o.skip && m2_bytes == o.m2_bytes && F == o.F &&
             fdim == o.fdim && k == o.k && ld == o.ld && n == o.n && mp_max == o.mp_max && s2 == o.s2 && cm == o.cm && rg == o.rg &&
             epoch == o.epoch;
    }
  };
  bool graph_mode = false, gseen = false;
  GraphKey gkey_seen{}, gkey{};
  hipGraphExec_t gexec = nullptr;
  int graph_replays = 0, graph_captures = 0;
  // jacobian inputs
  plv::DevBuf jin, tri, eval;
  plv::PinBuf h_jin;  // dedicated pinned staging: its upload is not followed by a host sync
  plv::PinBuf h_tri;  // triangulation results of the one-submission updates
  // the line half's own staging, triangulation and result blocks (round 4): the line launch of a frame is staged and enqueued while
  // the point update is still running and being collected (plv_camera_try_update's chained line launch), so nothing of the two
  // halves may share a buffer the host writes or reads
  plv::DevBuf jin_l, tri_l;
  plv::PinBuf h_jin_l, h_tri_l;
  size_t lt_o_lines = 0;  // plv_lines_update_fused_submit -> _finish: where the triangulated lines sit in tri_l, how many
  int lt_L = 0;
  int pending_fdim = 3;  // measurement size of the launched, not yet collected update (selects the pinned result block)
  // where the last fused point launch left its triangulation results on the device (the chained line launch reads its anchors there)
  const double *pt_tri_p = nullptr;
  const unsigned char *pt_tri_ok = nullptr;
  int pt_tri_F = 0;
  plv::DevBuf chain_words;      // home of applied_word
  int *applied_word = nullptr;  // device word ekf_commit_kernel sets to 1 when the point update changed the state (0: dx is not to be applied)
  bool applied_armed = false;   // ... and the last point launch ended in that kernel with the word as its argument
  hipStream_t spec_stream = nullptr;  // upload of a speculative batch (stage_inputs)
  hipEvent_t spec_ev = nullptr;
  // a submitted point update between plv_points_update_submit and plv_points_update_collect
  struct PointJob {
    bool pending = false, mirrored = false, chain_events = false, spec = false;
    int rc = 0, F = 0, max_sel = 0;
    size_t o_p = 0, o_member = 0, o_words = 0;
    std::chrono::steady_clock::time_point t_entry;
  } point_job;
};
plv_ctx_update_state *plv_update_state(plv_ctx *ctx);

extern "C" int plv_update_gate_prepare(plv_ctx *ctx, int F, int fdim, int k, int ld, double sigma2, double chi2_mult, double res_norm_gate, int probe);  // plv_api.hip
extern "C" int plv_prior_prefetch(plv_ctx *ctx, int phase, const int *d_cols, int k, int F, int mp_max);  // plv_api.hip

// Speculative submission of the point update (round 6): host arrays per candidate + where the flow leaves its results on the device
// (SpecSelectArgs, jacobian_kernels.hpp).  The candidates' observation ranges end with a slot for the frame's own observation when
// li >= 0 (time staged, image point and normalised point written by the device).
struct plv_points_spec {
  int n_flow;
  const int *li;
  const uint8_t *meta, *prevalid;
  const float *d_flow_p1, *d_flow_n1;
  const uint8_t *d_flow_mask;
};
// internal entry points of jacobian_api.hip used by the one-call camera updates (tracker_api.hip, line_api.hip)
// plv_points_update_fused = plv_points_update_submit (everything enqueued: upload, [spec_select,] triangulation + Jacobians + null space +
// gate, compression, EKFUpdate) + plv_points_update_collect (host work inside the wait, the wait, results).  With `spec` the batch is
// the speculative one; collect then also returns the device's membership (member [F], may be null) and *spec_over (1: the pool exceeded
// max_sel and every candidate was left empty — nothing was updated, the caller runs the update the long way).
extern "C" int plv_points_update_submit(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *all, const plv_tri_options *tri,
                                        const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2,
                                        double chi2_mult, double res_norm_gate, const plv_points_spec *spec);
extern "C" int plv_points_update_collect(plv_ctx *ctx, double *p_out, uint8_t *ok_out, double *err_out, uint8_t *accepted, int *n_rows, double *dx,
                                         void (*before_wait)(void *), void *before_wait_arg, uint8_t *member, int *spec_count, int *spec_over);
extern "C" int plv_points_update_fused(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *all, const plv_tri_options *tri,
                                       const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2,
                                       double chi2_mult, double res_norm_gate, double *p_out, uint8_t *ok_out, double *err_out,
                                       uint8_t *accepted, int *n_rows, double *dx, void (*before_wait)(void *), void *before_wait_arg);
extern "C" int plv_lines_update_fused(plv_ctx *ctx, const plv_state_view *st, const plv_state_view *st_tri, const plv_line_tracks *all,
                                      const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2,
                                      double chi2_mult, double *lines_out, uint8_t *ok_out, uint8_t *accepted, int *n_rows, double *dx,
                                      void (*before_wait)(void *), void *before_wait_arg);
extern "C" int plv_lines_update_fused_submit(plv_ctx *ctx, const plv_state_view *st, const plv_state_view *st_tri, const plv_line_tracks *all,
                                             const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2, double chi2_mult);
extern "C" int plv_lines_update_fused_finish(plv_ctx *ctx, double sigma2, double chi2_mult, double *lines_out, uint8_t *ok_out, uint8_t *accepted,
                                             int *n_rows, double *dx, void (*before_wait)(void *), void *before_wait_arg);
extern "C" void plv_tracker_run_deferred(void *ctx);  // tracker_api.hip
