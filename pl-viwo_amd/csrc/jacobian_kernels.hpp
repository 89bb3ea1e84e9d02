// jacobian_kernels.hpp — device-side view of plv_state_view / plv_tracks for jacobian_kernel.
#pragma once
#include "plv_ctx.hpp"

namespace plv {

struct JacParams {
  // state (device pointers)
  int n_clones;
  const double *clone_time, *clone_R, *clone_p, *clone_R_fej, *clone_p_fej;
  const int *clone_col;  // [n_clones] first column of each clone in the stacked Jacobian, or -1
  double R_ItoC[9], p_IinC[3], K[8];
  double cam_dt, dt_exp, sigma_pix, intr_ori_cov, intr_pos_cov;
  int use_pol_cov, feat_rep;
  int col_ext, col_int, col_dt;
  // tracks (device pointers)
  int n_feat, n_obs;
  const int *obs_ptr, *obs_feat;
  const double *obs_time;
  const float *obs_uv;
  const double *p_FinG, *p_FinG_fej, *res_R, *res_p;
  // line tracks (plv_line_tracks): obs_ptr / obs_time / res_R / res_p are shared with the point view
  const float *seg_uv, *seg_uvn;
  const double *line_FinG, *anchor_pt;
  const int *lineD;
  const unsigned char *has_pt;
  // outputs
  int k, ld;
  int *rows;
  double *Hf, *Hx, *res;
  // col_to_state rides in the packed input block; workgroup 0 copies it to its resident home (no separate H2D on the chain)
  const int *cols_in;
  int *cols_out;
  // one-submission update (plv_camera_update_points / _lines): the batch holds EVERY pool candidate and the Jacobian kernels decide
  // on the device which ones the reference's selection loop would take (CamHelper.cpp:648-699): sel_flags = the host's part of the
  // test (enough observations with bounding clones), tri_ok / tri_err = the triangulation kernel's verdict (err: mean reprojection
  // error, < 3 px; null for lines), at most max_sel candidates in batch order.  The others become empty (zero-row) systems.
  const unsigned char *sel_flags, *tri_ok;
  const double *tri_err;
  int max_sel;
  // the packed input block all the pointers above lead into (stage_inputs): the fused launches touch its lines first thing, so that
  // the chains of dependent loads behind (obs_ptr -> obs_time -> clone poses) meet one cold miss, not one each
  const char *in_base;
  int in_bytes;
  // (optional, plv_decision_trace) [n_feat][4]: the values the triangulation's tests looked at — condition number and depth of the
  // linear solution, depth and baseline ratio of the refined one (NaN: not reached)
  double *tri_dbg;
  // Chained launch (the line update enqueued behind a point update whose result the host has not seen, plv_camera_try_update): the
  // state above is the one BEFORE that update's correction; the kernel forms x (+) dx itself (StateHelper::EKFUpdate's mean update,
  // the arithmetic of plv_state_boxplus) for what it linearises on — clone poses, extrinsics, intrinsics, time offset — when the
  // update's commit kernel left 1 in *chain_applied.  chain_id: covariance index of every clone pose (orientation; position = + 3),
  // then of the extrinsics (same), the intrinsics and the time offset (-1: not estimated).  null chain_dx: not chained.
  const double *chain_dx;
  const int *chain_applied;
  const int *chain_status;  // status word of that update: nonzero = rejected — the host may run it again from the buffers both updates
                            // share (stack, column map, gathers), so this launch then ends at once, touching nothing (plv_api.hip, RedoW)
  const double *chain_q;  // [n_clones][4] JPL quaternions of the clones
  const int *chain_id;    // [n_clones + 3]
  double chain_qe[4];     // extrinsic quaternion (R_ItoC)
  // ... and a line's anchor point (LineHelper.cpp:233-247: the first point of the line that is triangulated) is looked up by the
  // kernel: per line a list of candidates in the line's point order — index into that point update's triangulation results (or -1)
  // and the value point_used held before (if any).  null anc_ptr: anchor_pt / has_pt as staged.
  const int *anc_ptr, *anc_f;          // [n_feat + 1], [candidates]
  const unsigned char *anc_has_old;    // [candidates]
  const double *anc_old;               // [candidates][3]
  const double *anc_tri_p;             // [F][3] of the point launch
  const unsigned char *anc_tri_ok;     // [F]
  // Speculative submission (round 6, plv_camera_frame): the batch holds EVERY track the frame's flow could send into the pool and was
  // staged before the flow's result was known; spec_select_kernel, behind the flow on the stream, decides the membership, appends the
  // frame's own observation to the tracks that survived and leaves the end of every candidate's observation range here (obs_ptr[f] for
  // a track that is not in the pool: an empty candidate).  null: obs_ptr[f + 1] as staged.
  const int *obs_end;
  // ... and the launch has one workgroup per POOL entry, not per candidate: workgroup b works on candidate spec_order[b] while
  // b < *spec_count (spec_select_kernel lists the pool's candidates in batch order and leaves the others' outputs empty)
  const int *spec_order, *spec_count;
  // ... and counts the candidates its workgroups selected (own verdict) in *spec_pass: a pool larger than max_sel is worked on as long
  // as it fits the launch (SpecSelectArgs::grid); whether the selection loop's cap would have cut it is known when all of them are
  // done — the update's commit kernel reads the count and leaves the state alone when it reached max_sel (plv_ctx::cap_words)
  int *spec_pass;
  // use_imu_cov: CPI covariance (6 x 6 row-major) of the pose each observation was made at and the clone it hangs on
  int use_imu_cov;
  double intr_err_mlt;
  const double *res_Q;
  const int *res_clone;
};

struct CpiParams {  // device pointers; State::cpis as a table sorted by time + the clone window
  int n, n_clones, n_q;
  const double *t, *clone_t, *dt, *R, *alpha, *v;
  const double *clone_time, *clone_R, *clone_p;
  double gravity[3];
};
int launch_cpi_poses(plv_ctx *ctx, const CpiParams &C, const double *d_tq, double *d_R, double *d_p, unsigned char *d_ok);

// The pool of CamHelper::get_features decided on the device (REF: PL-VIWO/src/update/cam/CamHelper.cpp:630-637 features_containing_older /
// features_not_containing_newer + :740-775 remove_unusable_measurements, open_vins/ov_core/src/track/TrackKLT.cpp:158-179 the frame's
// survivors) for a batch of candidates staged before the frame's flow had finished.  Per candidate f (device arrays unless said):
//   li [F]        index of the track's point in the flow's batch, -1: the track was not tracked into this frame
//   meta [F]      bit 0: the track holds an observation older than the second-oldest clone (in the pool whatever the flow says)
//                 bit 1: the frame's own observation would be inside the window (usable), bit 2: ... and has bounding clones
//                 bit 3: the track holds an observation newer than the previous frame already (a positive camera time offset)
//   prevalid [F]  the track's observations before this frame that have bounding clones
// and the flow's outputs flow_p1 / flow_n1 [n][2], flow_mask [n] (RANSAC inlier and LK status), the image size.  A track survives
// when its point is an inlier inside the image; it is in the pool when it carries bit 0, or neither survived nor carries bit 3; a pool track of fewer
// than two usable observations is dropped (CamHelper.cpp:766-771).  Outputs: obs_end [F], sel_flags [F] (>= 2 observations with
// bounding clones), member [F], words [0] = pool size, [1] = 1 when the pool exceeds `grid` (the workgroups of the Jacobian launch) —
// then EVERY candidate is left empty and the host runs the update the long way —, [2] = the pool entries the launch works on,
// [3] = 1 when the pool exceeds max_sel but not grid: the selection loop's cap (CamHelper.cpp:651-653: it stops after max_sel
// selected candidates) cuts such a pool only when max_sel of its candidates pass their own tests, which the launch counts in [4]
// (JacParams::spec_pass) — and the survivors' observation of this frame written into the last slot of their range (time staged by
// the host).
struct SpecSelectArgs {
  int F, n_flow, W, H, max_sel, grid;
  const int *obs_ptr, *li;
  const unsigned char *meta, *prevalid;
  const float *flow_p1, *flow_n1;
  const unsigned char *flow_mask;
  float *obs_uv, *obs_uvn;
  int *obs_end;
  unsigned char *sel_flags, *member;
  int *words;
  // the pool's candidates in batch order [grid] and, for every candidate that is NOT worked on by the Jacobian launch, the outputs
  // that launch and its gate would have left: rows, triangulation result, verdict, accepted rows
  int *order, *rows_out;
  double *tri_p, *tri_err, *chi2;
  unsigned char *tri_ok, *accepted;
  int *acc_rows;
  // ... and what workgroup 0 of that launch does for the launch as a whole: the gate counter of the NEXT update zeroed, the column map
  // copied to its resident home
  int *zero_word, *cols_out;
  const int *cols_in;
  int k;
};
int launch_spec_select(plv_ctx *ctx, const SpecSelectArgs &A);

int launch_jacobians(plv_ctx *ctx, const JacParams &P);
struct GatherArgs;
// the batch built AND null-space projected in one launch (resident update path); g != null: the covariance gathers ride along
// tri_opt != null: every workgroup first triangulates its feature (arguments as launch_triangulate takes them)
int launch_jacobians_projected(plv_ctx *ctx, const JacParams &P, const GatherArgs *g, int gather_blocks, const plv_tri_options *tri_opt = nullptr,
                               double *d_poses = nullptr, unsigned char *d_valid = nullptr, const float *d_uvn = nullptr, double *d_p = nullptr,
                               unsigned char *d_ok = nullptr, double *d_err = nullptr, int max_obs = 0);
int launch_line_jacobians(plv_ctx *ctx, const JacParams &P);
// Pt != null: every workgroup first triangulates its line on the state Pt (scratch and results as launch_triangulate_lines takes them)
int launch_line_jacobians_projected(plv_ctx *ctx, const JacParams &P, const GatherArgs *g, int gather_blocks, const JacParams *Pt = nullptr,
                                    double *d_cam = nullptr, double *d_imu = nullptr, unsigned char *d_valid = nullptr, double *d_lines = nullptr,
                                    unsigned char *d_ok = nullptr, int max_obs = 0);
int launch_triangulate_lines(plv_ctx *ctx, const JacParams &P, double *d_poses, double *d_imu, unsigned char *d_valid,
                             double *d_lines, unsigned char *d_ok);
int launch_triangulate(plv_ctx *ctx, const JacParams &P, double *d_poses, unsigned char *d_valid, const float *d_uvn,
                       const plv_tri_options &opt, double *d_p, unsigned char *d_ok, double *d_err, int max_obs);

}  // namespace plv
