/*
 * plviwo.h — C-ABI of the MI355X-native per-frame (track + EKF update) path of PL-VIWO.
 *
 * This is the drop-in boundary: everything HIP lives behind these entry points.  The host
 * side (the C++ mirror of ov_core::TrackBase / viw::UpdaterCamera / viw::StateHelper under
 * pl-viwo_amd/host/, or the reference's own classes through the stubs in INTEGRATION.md)
 * calls only what is declared here.  No torch / Eigen / OpenCV types cross the boundary:
 * plain pointers and sizes, `int` status codes, no C++ exceptions.
 *
 * Conventions
 *   - dense matrices are COLUMN-MAJOR with an explicit leading dimension, like Eigen's default,
 *     so `MatrixXd::data()` passes straight through;
 *   - images are ROW-MAJOR u8 with an explicit byte stride, like `cv::Mat(CV_8UC1)`;
 *   - the caller owns every host buffer; the library owns device memory inside `plv_ctx`;
 *   - one `plv_ctx` per camera, calls on one ctx are serialised by the caller, one HIP stream
 *     per ctx, every call is synchronous at return unless its name ends in `_async`;
 *   - pointers are HOST pointers unless the parameter is documented as "device".
 *
 * Citations `REF:` are relative to /root/reference (see SURVEY.md for the prefixes).
 */
#ifndef PLVIWO_H
#define PLVIWO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLV_ABI_VERSION 1

/* ---------------------------------------------------------------- status codes */
enum {
  PLV_OK = 0,
  PLV_E_BADARG = -1,    /* null pointer / size mismatch. REF: TrackKLT.cpp:37-43 exits; we return */
  PLV_E_DEVICE = -2,    /* HIP runtime error (plv_last_error has the text)                        */
  PLV_E_NOT_PSD = -3,   /* EKFUpdate negative diagonal: P and dx untouched. REF: StateHelper.cpp:143-152 */
  PLV_E_NOMEM = -4,
  PLV_E_CAPACITY = -5,  /* problem larger than the ctx was sized for                              */
  PLV_E_NO_DEVICE = -6, /* no gfx950 device visible: the product path never falls back to a CPU   */
  PLV_E_NUMERIC = -7    /* NaN / non-finite in a numerical step (REF: UpdaterStatistics.cpp:118)  */
};

/* ---------------------------------------------------------------- configuration */
enum { PLV_HIST_NONE = 0, PLV_HIST_HISTOGRAM = 1, PLV_HIST_CLAHE = 2 }; /* REF: TrackBase.h:78 */

typedef struct plv_config {
  /* image / tracker.  REF: TrackKLT ctor TrackKLT.h:57-62, UpdaterCamera.cpp:41 */
  int width, height;
  int num_features;      /* n_pts */
  int fast_threshold;
  int grid_x, grid_y;
  int min_px_dist;
  int histogram_method;  /* PLV_HIST_* */
  int win_size;          /* 15  REF: TrackKLT.h:144 */
  int pyr_levels;        /* 5   REF: TrackKLT.h:143 (OpenCV maxLevel) */
  int lk_max_iters;      /* 30  REF: TrackKLT.cpp:857 */
  float lk_eps;          /* 0.01 */
  double ransac_thr_px;  /* 2.0  REF: TrackKLT.cpp:873 (divided by max focal length inside) */
  double ransac_conf;    /* 0.999 */
  int ransac_max_iters;  /* 1000 (OpenCV default for findFundamentalMat) */
  double intrinsics[8];  /* fx fy cx cy k1 k2 p1 p2  REF: CamBase.h:56-82 */
  /* line front-end.  REF: TrackLSD.h:269-273, TrackLSD.cpp:200-231,780,824 */
  int line_length_threshold;   /* 20 (half-res px) */
  float line_distance_threshold; /* 1.41421356 */
  int canny_th1, canny_th2, canny_aperture; /* 50,50,3 */
  float line_min_length_px;    /* 40 (full-res) */
  float line_assign_px;        /* 5 */
  float line_similar_px;       /* 6 */
  /* update sizing */
  int max_state_dim;     /* capacity for n (rows of P) */
  int max_meas_rows;     /* capacity for stacked rows before compression */
  int max_features;      /* capacity for features per update batch */
  int max_rows_per_feat; /* 2*M_max */
  double sigma_pix;      /* REF: OptionsCamera.h (sigma_pix) */
  double chi2_mult;      /* REF: OptionsCamera.h (chi2_mult) */
  int device;            /* HIP device ordinal */
} plv_config;

/* Fills every field with the reference's hard-coded / KAIST defaults for a W x H camera. */
void plv_config_default(plv_config *cfg, int width, int height);

typedef struct plv_ctx plv_ctx;

int plv_abi_version(void);
const char *plv_last_error(void);
/* number of visible gfx950 devices (0 => every compute call returns PLV_E_NO_DEVICE) */
int plv_device_count(void);
/* NUMA node of HIP device `device` (from its PCI function in sysfs), -1 when the system does not say.  Host threads that drive a
 * ctx — and the library's own, which inherit the creating thread's affinity — are best kept on that node's cores. */
int plv_device_numa_node(int device);

int plv_ctx_create(const plv_config *cfg, plv_ctx **out);
void plv_ctx_destroy(plv_ctx *ctx);
int plv_ctx_synchronize(plv_ctx *ctx);

/* ---------------------------------------------------------------- profiling hooks
 * HIP-event timing of the library's own kernels on the ctx stream (bench.py's roofline leg).
 * `plv_prof_get` returns, for kernel class `name`, the number of launches and the total
 * device time in milliseconds accumulated since plv_prof_reset. */
int plv_prof_enable(plv_ctx *ctx, int on);
int plv_prof_reset(plv_ctx *ctx);
int plv_prof_count(plv_ctx *ctx);
int plv_prof_get(plv_ctx *ctx, int idx, char *name, int name_cap, int *launches, double *total_ms);

/* ================================================================ EKF update side (fp64)
 *
 * plv_ekf_update replaces viw::StateHelper::EKFUpdate  (REF: PL/state/StateHelper.cpp:94-173):
 *   M = P[:,cols] H^T ; S = H P[cols,cols] H^T + R ; K = M S^-1 ; dx = K res ; P -= K M^T.
 * `col_to_state[j]` is the row/col of P that column j of H multiplies — the flat equivalent of
 * the reference's H_order (vector<shared_ptr<Type>> with id()/size()).
 * R is diagonal (`Rdiag`, r entries) or identity when NULL (REF: UpdaterCamera.cpp:290).
 * On PLV_E_NOT_PSD neither P nor dx is modified.
 * If `P` is NULL the device-resident covariance (plv_cov_upload) is updated in place.
 */
int plv_ekf_update(plv_ctx *ctx, double *P, int n, int ldp, const double *H, int r, int k, int ldh,
                   const int *col_to_state, const double *res, const double *Rdiag, double *dx);

/* device-resident covariance (REF: State.h:226 `MatrixXd cov`) */
int plv_cov_upload(plv_ctx *ctx, const double *P, int n, int ldp);
int plv_cov_download(plv_ctx *ctx, double *P, int n, int ldp);

/* plv_compress replaces StateHelper::measurement_compress_inplace (REF: StateHelper.cpp:602-614,
 * 653-672): QR of [H | res], keep the top k rows.  If m <= k nothing happens (m_out = m).
 * Output R is upper-triangular with non-negative diagonal (the sign convention Eigen's
 * makeGivens produces), so it is directly comparable with the reference's result. */
int plv_compress(plv_ctx *ctx, double *H, int m, int k, int ldh, double *res, int *m_out);

/* plv_nullspace_batch replaces StateHelper::nullspace_project_inplace for F features at once
 * (REF: StateHelper.cpp:616-651).  Feature f owns rows[f] rows; Hf is [F][fdim][ld] , Hx is
 * [F][k][ld], res is [F][ld] (col-major per feature, ld = padded rows).  On return the first
 * rows[f]-fdim rows of Hx/res of feature f hold the projected system (the reference drops the
 * first fdim rows; we shift up the same way). Same rotation order as the reference. */
int plv_nullspace_batch(plv_ctx *ctx, int F, int fdim, int k, int ld, const int *rows, double *Hf,
                        double *Hx, double *res);

/* plv_chi2_batch replaces UpdaterStatistics::get_chi2 for F features (REF:
 * UpdaterStatistics.cpp:94-117): chi2[f] = r^T (H P_s H^T + sigma2 I)^-1 r with
 * P_s = P[cols, cols]; uses the device-resident covariance when P is NULL. */
int plv_chi2_batch(plv_ctx *ctx, const double *P, int n, int ldp, int F, int k, int ld,
                   const int *rows, const double *Hx, const double *res, const int *col_to_state,
                   double sigma2, double *chi2);

/* 95% chi-square quantile for `dof` degrees of freedom (REF: UpdaterStatistics.cpp:31-37 uses
 * boost::math::quantile(chi_squared(k), 0.95)). */
double plv_chi2_quantile95(int dof);

/* plv_msckf_update: the whole of UpdaterCamera::msckf_update after the per-feature Jacobians
 * (REF: UpdaterCamera.cpp:230-293) in one device pass without host round trips:
 *   nullspace -> R = sigma^2 I gate (||res|| < res_norm_gate && chi2 < mult*q95) -> stack accepted
 *   -> compress -> R = I -> EKFUpdate.
 * `res_norm_gate` <= 0 disables the norm term (lines: REF UpdaterCamera.cpp:420).
 * accepted[f] (may be NULL) receives 1/0 per feature (REF: Feature::Chi_test).
 * Returns PLV_OK, or PLV_E_NOT_PSD (state untouched). n_accepted_rows may be NULL. */
int plv_msckf_update(plv_ctx *ctx, double *P, int n, int ldp, int F, int fdim, int k, int ld,
                     const int *rows, const double *Hf, const double *Hx, const double *res,
                     const int *col_to_state, double sigma2, double chi2_mult,
                     double res_norm_gate, uint8_t *accepted, int *n_accepted_rows, double *dx);

/* Device-resident variants (the per-frame path keeps its operands in HBM between calls):
 * plv_feat_batch_upload stages one batch of per-feature systems (same layout as
 * plv_msckf_update) in the ctx; plv_msckf_update_resident runs the update on the staged batch
 * and the device-resident covariance (plv_cov_upload) and returns only dx / accepted.  The staged
 * batch is preserved (the kernels work on a device copy), so the call can be repeated.
 * plv_cov_checkpoint / plv_cov_rollback keep and restore a device copy of the covariance
 * (REF: StateHelper::initialize rolls the state back on a failed update, StateHelper.cpp:430-435). */
int plv_feat_batch_upload(plv_ctx *ctx, int F, int fdim, int k, int ld, const int *rows, const double *Hf,
                          const double *Hx, const double *res, const int *col_to_state);
int plv_msckf_update_resident(plv_ctx *ctx, double sigma2, double chi2_mult, double res_norm_gate,
                              uint8_t *accepted, int *n_accepted_rows, double *dx);
/* The same in two halves, for callers that overlap the update with other work (e.g. the front-end of the next
 * frame on another context / stream): _launch enqueues every kernel and the result copy and returns without
 * waiting; _wait blocks until they are done and unpacks accepted / rows / dx.  No other call may use this ctx in
 * between. */
int plv_msckf_update_resident_launch(plv_ctx *ctx, double sigma2, double chi2_mult, double res_norm_gate);
int plv_msckf_update_resident_wait(plv_ctx *ctx, uint8_t *accepted, int *n_accepted_rows, double *dx);
/* Optional hipGraph replay of the launch sequence behind plv_msckf_update_resident_launch (nullspace .. EKF commit + the copy
 * of the result block): on = 1 / 0 switches it, -1 only queries.  The first call with a given shape (batch sizes, state
 * dimension, gate parameters, buffer addresses) runs eagerly, the second is captured, later ones are one hipGraphLaunch.
 * Any change of shape or any device (re)allocation retires the graph.  Off by default.  captures / replays (nullable) count. */
int plv_update_graph_mode(plv_ctx *ctx, int on, int *captures, int *replays);
/* How StateHelper::measurement_compress_inplace + StateHelper::EKFUpdate (REF: StateHelper.cpp:602-672 Givens rotations on the stacked
 * Jacobian, :94-173 the update with the compressed system) are carried out inside plv_msckf_update* / the one-call camera updates
 * when there are more rows than columns:
 *   0 (default)  whitened update: G = H^T H, g = H^T r over the accepted rows; Ps = P[cols, cols] = Lp Lp^T (unit-diagonal scaling,
 *                exact dependencies such as the IMU pose and its fresh clone dropped); B = I + Lp^T G Lp = Lb Lb^T;
 *                P' = P - W0^T W0 + V^T V, dx = V^T v with W0 = Lp^-1 P[cols, :] (its columns of the update's own states are
 *                Lp^T and are copied from the factor, which keeps the conditional variances of nearly dependent states — clone
 *                positions — to eps / pivot as the reference's form does), [V | v] = Lb^-1 [W0 | Lp^T g].  The same P', dx as
 *                the reference's R-based update in exact arithmetic; no pivot of the measurement side is ever divided by, so the
 *                gauge directions of an MSCKF Jacobian cost nothing: agrees with the Givens oracle to 1e-10 (P') / 1e-9 (dx) on
 *                every captured replay batch and up to condition 1e8 of the stacked Jacobian.  The prior factor runs on a side
 *                stream during the Jacobian / gate launches;
 *   1            Householder TSQR on the stacked rows themselves (orthogonal transformations: the reference's accuracy, ~0.7 ms);
 * (Rounds 2-3 also offered a Gram + Cholesky compression, modes 2 and 3; it squared the condition number of the stacked Jacobian and was
 * removed in round 5: modes above 1 return PLV_E_BADARG.  Beyond 192 measured columns, or in graph mode, 0 falls back to 1.)
 * mode < 0 only queries.  Returns the mode in force.  last_route (nullable): how the last update was carried out — 0 not compressed
 * (fewer rows than columns), 2 Householder, 4 whitened, 5 whitened rejected on the device and redone by Householder (1 and 3 were the
 * removed routes); last_ambiguous (nullable): always 0 (kept for the signature). */
int plv_update_compression_mode(plv_ctx *ctx, int mode, int *last_route, int *last_ambiguous);
/* Process-wide activity counters since load (measurement aid, no reference counterpart): out[0] kernel launches, [1] host
 * synchronisations (stream / event waits), [2] asynchronous copies, [3] bytes copied, [4] LK iterations over all points and levels
 * (plv_perform_matching), [5] line segments the detector returned inside plv_line_tracker_feed*, [6] nanoseconds spent inside
 * plv_camera_frame and [7] inside plv_ctx_synchronize (std::chrono::steady_clock: the library's own per-frame time, the twin of the
 * timer inside the CPU oracle's frame). */
void plv_counters(unsigned long long *out8);
/* (measurement aid) updates collected since the library was loaded, by route: index = plv_update_compression_mode's last_route
 * (0 no compression, 2 Householder, 4 whitened, 5 whitened rejected, then Householder; 1 and 3: unused since round 5); [6] counts
 * the route-5 point updates of plv_camera_try_update that had a chained line launch behind them (which is then withdrawn and redone);
 * [7] the point updates plv_camera_frame had enqueued behind the frame's flow, before its result was known, and that stood */
void plv_route_counts(unsigned long long *out8);
/* (measurement aid) speculative point updates of plv_camera_frame since the library was loaded: [0] enqueued behind the frame's flow
 * and used as they ran (= plv_route_counts[7]), [1] of those: the pool held more tracks than max_msckf but fewer than max_msckf of
 * them passed their tests (the selection loop of CamHelper.cpp:648-699 never reaches its cap), [2] withdrawn and run again after the
 * flow's result: max_msckf tracks or more of such a pool passed (the device committed nothing), [3] withdrawn: the pool was larger
 * than the launch (twice max_msckf) */
void plv_speculation_counts(unsigned long long *out4);
/* (no reference counterpart) What the library holds, all contexts of the process together: out[0] bytes of device memory, [1] bytes of
 * pinned host memory, [2] / [3] their peaks since the library was loaded or plv_memory_policy was last called.  Buffers grow on demand
 * and are never shrunk; a context's buffers are released by plv_ctx_destroy. */
void plv_memory_bytes(unsigned long long *out4);
/* How a buffer is sized when it has to grow: growth_percent of what is asked for (100 .. 1000, default 200: batch sizes wander from
 * frame to frame and a buffer that regrows inside a frame costs that frame 0.3-0.4 ms) and never less than device_floor_kb (default
 * 1024) / pinned_floor_kb (default 256).  -1 leaves a value as it is.  Process-wide; buffers that exist keep their size.  Resets the
 * peaks of plv_memory_bytes.  PLV_E_BADARG for a growth below 100 or above 1000. */
int plv_memory_policy(int growth_percent, int device_floor_kb, int pinned_floor_kb);
/* (test aid) Decision trace.  With it on, plv_camera_update_points (alone or inside plv_camera_try_update / plv_camera_frame) keeps, for
 * every feature of its pool, the values its verdicts were taken on; plv_last_point_decisions returns them for the last update:
 * ids [n] in pool order and vals [n][PLV_DECISION_VALUES] =
 *   0 usable observations   1 triangulated (0 / 1)   2 mean reprojection error, px (the 3 px test, REF UpdaterCamera.cpp:656-683)
 *   3 passed the gate (0 / 1)
 *   4 condition number and 5 depth of the linear triangulation (REF FeatureInitializer.cpp:105-117: max_cond_number, min / max_dist)
 *   6 depth and 7 baseline ratio after the refinement (REF FeatureInitializer.cpp:283-301: min / max_dist, max_baseline)
 *   8 chi2, 9 the threshold it was held against (chi2_mult x the 95 % quantile), 10 the norm of the projected residual
 * NaN: a test that was not reached (or a route that does not report: the two-step update of CPI poses / in-state landmarks).  cap = 0
 * asks for n only.  The device values cost two small copies per update: not for timed runs. */
#define PLV_DECISION_VALUES 11
int plv_decision_trace(plv_ctx *ctx, int on);
int plv_last_point_decisions(plv_ctx *ctx, uint64_t *ids, double *vals, int cap, int *n);
/* ... and of the last line update: ids [n] as plv_camera_update_lines returned them, vals [n][3] = chi2, its threshold, the norm of the
 * projected residual (REF UpdaterCamera.cpp:406-419; NaN: the line did not reach the gate). */
int plv_last_line_decisions(plv_ctx *ctx, uint64_t *ids, double *vals, int cap, int *n);
/* (measurement aid) line launches plv_camera_try_update enqueued behind a point update that was still running */
unsigned long long plv_chain_count(void);
int plv_cov_checkpoint(plv_ctx *ctx);
int plv_cov_rollback(plv_ctx *ctx);

/* ================================================================ point front-end
 *
 * plv_feed_image replaces the per-image preprocessing of TrackKLT::feed_new_camera (REF:
 * ov_core/src/track/TrackKLT.cpp:54-75): cv::equalizeHist (histogram_method == HISTOGRAM) and
 * cv::buildOpticalFlowPyramid(win_size, pyr_levels).  The previously current pyramid becomes the
 * "last" pyramid (REF: the img_pyramid_last <- img_pyramid_curr swap, TrackKLT.cpp:182-189).
 * `img` is a host CV_8UC1 image, row stride `stride` bytes.  plv_image_stage / plv_feed_staged do
 * the same from an image already resident in HBM (slots 0..7); plv_feed_staged only enqueues
 * (stream-ordered, no host synchronisation: the next data-returning call on the ctx waits for it).
 * Size mismatch -> PLV_E_BADARG (the reference exits: TrackKLT.cpp:37-43). */
enum { PLV_PYR_CUR = 0, PLV_PYR_LAST = 1 };
int plv_feed_image(plv_ctx *ctx, const uint8_t *img, int stride);
int plv_image_stage(plv_ctx *ctx, int slot, const uint8_t *img, int stride);
int plv_feed_staged(plv_ctx *ctx, int slot);
/* A page-locked host block of the library (index 0..3, width x height bytes, packed rows: *stride = width) for the caller to
 * produce the next image in — the target of the copy the reference's ROS callback makes anyway (cv_bridge's clone() of the
 * message: `cv::Mat(h, w, CV_8UC1, ptr)`) or of a camera driver's DMA.  plv_feed_image / plv_tracker_feed / plv_camera_frame
 * recognise an `img` that points at such a block and read it from where it lies (their first kernel brings the pixels across
 * PCIe: no host copy, no copy command); any other `img` is first copied into a block like these by the call.  A block must not
 * be overwritten before the call it was handed to has returned.  REF: the CameraData the caller builds, ROSSubscriber /
 * run_bag -> UpdaterCamera::feed_measurement (UpdaterCamera.cpp:77). */
int plv_image_buffer(plv_ctx *ctx, int index, uint8_t **ptr, int *stride);
int plv_pyramid_levels(plv_ctx *ctx, int which);
/* level geometry and (if out != NULL, w*h bytes, packed) pixels of one pyramid level */
int plv_pyramid_download(plv_ctx *ctx, int which, int level, int *w, int *h, uint8_t *out);

/* plv_lk_track replaces cv::calcOpticalFlowPyrLK(last_pyr, cur_pyr, pts0, pts1, status, err,
 * win, maxLevel, {COUNT|EPS, lk_max_iters, lk_eps}, OPTFLOW_USE_INITIAL_FLOW) (REF call site:
 * TrackKLT.cpp:857-858).  pts0 / pts1 are n x 2 float (x,y); pts1 carries the initial flow in and
 * the tracked position out; status n bytes; iters (nullable) n ints = LK iterations actually run. */
int plv_lk_track(plv_ctx *ctx, int n, const float *pts0, float *pts1, uint8_t *status, int *iters);

/* plv_undistort replaces CamRadtan::undistort_f per point (REF: ov_core/src/cam/CamRadtan.h:99-120,
 * cv::undistortPoints with K, D from cfg.intrinsics): pixel uv -> normalised xy, float. */
int plv_undistort(plv_ctx *ctx, int n, const float *uv, float *xy);

/* plv_ransac_fundamental replaces cv::findFundamentalMat(m1, m2, FM_RANSAC, thr, ransac_conf,
 * mask) (REF call site: TrackKLT.cpp:870-873); m1/m2 are normalised coordinates, thr in the
 * same units.  mask n bytes; n_inliers / iters_used nullable. */
int plv_ransac_fundamental(plv_ctx *ctx, int n, const float *m1, const float *m2, double thr, uint32_t seed,
                           uint8_t *mask, int *n_inliers, int *iters_used);

/* plv_perform_matching replaces TrackKLT::perform_matching (REF: TrackKLT.cpp:829-886) on the
 * last -> current pyramids: LK, undistort both point sets, RANSAC with thr = ransac_thr_px /
 * max(fx,fy), mask_out = klt & ransac.  n < 10 -> all-zero mask, PLV_OK (REF :848-852).
 * n0 / n1 (nullable) receive the normalised coordinates; lk_iters (nullable) the total LK
 * iteration count of the call. */
int plv_perform_matching(plv_ctx *ctx, int n, const float *pts0, float *pts1, uint8_t *mask_out, float *n0,
                         float *n1, long long *lk_iters);
/* The same in two halves, as plv_msckf_update_resident_launch / _wait: _launch enqueues the copy-in, LK, undistortion, RANSAC
 * and the copy-back on the context's stream and returns; _wait blocks for them and hands the results over.  No other call
 * that returns data may be made on this context in between (they share the pinned staging block). */
int plv_perform_matching_launch(plv_ctx *ctx, int n, const float *pts0, const float *pts1_init);
int plv_perform_matching_wait(plv_ctx *ctx, float *pts1, uint8_t *mask_out, float *n0, float *n1, long long *lk_iters);

/* plv_perform_detection replaces TrackKLT::perform_detection_monocular (REF: open_vins/ov_core/src/
 * track/TrackKLT.cpp:395-528, Grider_GRID::perform_griding Grider_GRID.h:74-180): drops tracked
 * points that are near the border / masked / closer than min_px_dist to another one, and when
 * fewer than num_features remain tops them up with FAST corners (per under-filled grid cell, best
 * num_features/(grid_x*grid_y)+1 by response, cv::cornerSubPix refined) on level 0 of the chosen
 * pyramid (PLV_PYR_LAST for the top-up on the previous image, PLV_PYR_CUR for initialisation).
 * pts [cap][2] / ids [cap] hold n_in points on entry and *n_out on return; new points get ids
 * ++*currid (REF: TrackBase::currid).  mask: optional host W x H u8 (255 = masked), may be NULL. */
int plv_perform_detection(plv_ctx *ctx, int which, const uint8_t *mask, float *pts, uint64_t *ids, int n_in, int cap,
                          uint64_t *currid, int *n_out);

/* ================================================================ tracker frame logic + feature database
 *
 * Host mirror of ov_core::TrackKLT (monocular) and ov_core::FeatureDatabase for callers that do
 * not bring the reference's own classes (INTEGRATION.md §2 shows the adapter that keeps them).
 * plv_tracker_feed = TrackKLT::feed_new_camera (REF: open_vins/ov_core/src/track/TrackKLT.cpp:34-200):
 * equalize + pyramid, first-frame detection or top-up on the last image, temporal KLT + RANSAC,
 * in-bounds / mask filter, database update (u, v, u_n, v_n, timestamp per id).  mask may be NULL. */
int plv_tracker_feed(plv_ctx *ctx, double timestamp, const uint8_t *img, int stride, const uint8_t *mask);
/* On (the default), the NEXT frame's top-up detection (TrackKLT.cpp:127-131 runs it on the then-last image with the then-last
 * points: this image, these points) is started ahead of time: plv_camera_update_points starts it once the point update of this frame
 * is submitted (its host stage and launches sit in the update's wait) — on a side stream, or, inside plv_camera_try_update with a
 * line update to follow, on the ctx stream behind the update, whose wait ends at its own last kernel, so that the detection runs in
 * the device's idle time before the line update is submitted; the next feed collects it; without an update in between the next feed
 * detects in place.  on = 1: at the end of the feed, on a side stream.  Same points, same ids.  Falls back to the in-line detection whenever the inputs differ (e.g. after
 * plv_tracker state was edited) or the per-kernel profiler is on. */
int plv_tracker_detect_ahead(plv_ctx *ctx, int on);
/* plv_tracker_feed from an image already resident in HBM (plv_image_stage, slots 0..7): the camera driver's DMA target in a
 * deployment, and the form bench.py times (no PCIe copy inside the frame). */
int plv_tracker_feed_staged(plv_ctx *ctx, double timestamp, int slot, const uint8_t *mask);
/* feed_measurement with OptionsCamera::downsample (REF: UpdaterCamera.cpp:85-98): cv::pyrDown(img, Size(cols / 2.0,
 * rows / 2.0)) of image and mask on the device, then the same path.  The context is created at the halved
 * resolution with halved intrinsics, as the reference's option loader does (OptionsCamera.cpp:123-138). */
int plv_tracker_feed_downsampled(plv_ctx *ctx, double timestamp, const uint8_t *img, int stride, int src_w, int src_h,
                                 const uint8_t *mask, int mask_stride);
/* The pyrDown alone (host in, host out): dst is (src_w / 2) x (src_h / 2). */
int plv_downsample(plv_ctx *ctx, const uint8_t *src, int stride, int src_w, int src_h, uint8_t *dst, int dstride);
/* plv_feed_image on the halved image, without the tracker bookkeeping. */
int plv_feed_image_downsampled(plv_ctx *ctx, const uint8_t *img, int stride, int src_w, int src_h);
/* TrackBase::get_last_obs / get_last_ids (REF: TrackBase.h:121-131) */
int plv_tracker_last(plv_ctx *ctx, float *pts, uint64_t *ids, int cap, int *n);
int plv_db_size(plv_ctx *ctx);
/* mode 0 = FeatureDatabase::features_not_containing_newer(t), 1 = features_containing_older(t)
 * (REF: FeatureDatabase.cpp:147-232); ids ascending. */
int plv_db_select(plv_ctx *ctx, int mode, double t, uint64_t *ids, int cap, int *n);
/* CSR export of tracks in plv_tracks layout: obs_ptr [n+1], obs_time / obs_uv / obs_uvn [cap_obs] */
int plv_db_export_tracks(plv_ctx *ctx, const uint64_t *ids, int n, int *obs_ptr, double *obs_time, float *obs_uv,
                         float *obs_uvn, int cap_obs);
int plv_db_cleanup_measurements(plv_ctx *ctx, double t); /* REF: FeatureDatabase.cpp:286-323 */
int plv_db_remove(plv_ctx *ctx, const uint64_t *ids, int n);

/* ================================================================ per-feature Jacobians (K10)
 *
 * Flat views of what CamHelper::get_feature_jacobian_full (REF: PL-VIWO/src/update/cam/
 * CamHelper.cpp:58-267) and State::get_interpolated_jacobian (REF: PL-VIWO/src/state/State.cpp:
 * 833-973) read from the pointer-linked State / Feature objects.
 * Rotations are 3x3 ROW-major R_GtoI (PoseJPL::Rot()), positions p_IinG; the 6-dof pose error is
 * ordered (theta, p) like ov_type::PoseJPL. */
enum { PLV_FEAT_GLOBAL_3D = 0, PLV_FEAT_GLOBAL_FULL_INVERSE_DEPTH = 1 }; /* REF: LandmarkRepresentation.h */

typedef struct plv_state_view {
  int n_clones;
  const double *clone_time;    /* [n_clones] ascending (State::clones is a std::map<double, PoseJPL>) */
  const double *clone_R;       /* [n_clones][9]  current estimate (Rot())            */
  const double *clone_p;       /* [n_clones][3]                    (pos())            */
  const double *clone_R_fej;   /* [n_clones][9]  first estimates   (Rot_fej())        */
  const double *clone_p_fej;   /* [n_clones][3]                    (pos_fej())        */
  const int *clone_state_id;   /* [n_clones] covariance index of each clone's 6-dof error (Type::id()) */
  double R_ItoC[9], p_IinC[3]; /* camera extrinsics, State::cam_extrinsic             */
  double intrinsics[8];        /* State::cam_intrinsic value (fx fy cx cy k1 k2 p1 p2) */
  double cam_dt;               /* State::cam_dt value                                  */
  int extrinsic_state_id;      /* covariance index, or -1 unless do_calib_ext          */
  int intrinsic_state_id;      /* ... -1 unless do_calib_int                           */
  int dt_state_id;             /* ... -1 unless do_calib_dt                            */
  int intr_order;              /* 3 (REF: OptionsEstimator.h intr_order); only 3 is built */
  double dt_exp;               /* 0.01 s extrapolation allowance (OptionsEstimator.h:56) */
  double sigma_pix;
  int use_pol_cov;             /* REF: CamHelper.cpp:214-217 */
  double intr_ori_cov, intr_pos_cov; /* interpolation_error::ori_cov / pos_cov (OptionsEstimator.h:60-80) */
  int feat_rep;                /* PLV_FEAT_* */
  int use_imu_cov;             /* OptionsEstimator::use_imu_cov (REF: CamHelper.cpp:217-224, LineHelper's twin): the CPI covariance of the
                                  pose an observation was made at inflates its noise; read only when use_pol_cov is 0 and the
                                  tracks carry res_Q / res_clone */
  double intr_err_mlt;         /* OptionsEstimator::intr_err.mlt, the factor on that covariance */
} plv_state_view;

typedef struct plv_tracks {
  int n_feat;
  const int *obs_ptr;      /* [n_feat+1] CSR into the observation arrays                       */
  const double *obs_time;  /* [n_obs] measurement time stamps (cam_dt is added inside)         */
  const float *obs_uv;     /* [n_obs][2] raw pixel measurement   (Feature::uvs)                */
  const double *p_FinG;    /* [n_feat][3] triangulated position                                */
  const double *p_FinG_fej;/* [n_feat][3] (== p_FinG for MSCKF features, CamHelper.cpp:556-557) */
  const double *res_R;     /* optional [n_obs][9]: IMU pose for the RESIDUAL at each obs, e.g. the CPI
                              pose when use_imu_res (State.cpp:1138-1155); NULL -> estimate polynomial */
  const double *res_p;     /* optional [n_obs][3] */
  const float *obs_uvn;    /* [n_obs][2] normalised coordinates (Feature::uvs_norm); only plv_triangulate reads it */
  const double *res_Q;     /* optional [n_obs][36], row-major 6x6: State::CPI::Q of the record behind res_R / res_p (use_imu_cov) */
  const int *res_clone;    /* optional [n_obs]: index into the view's clone arrays of that record's clone (CPI::clone_t)      */
} plv_tracks;

/* FeatureInitializerOptions (REF: open_vins/ov_core/src/feat/FeatureInitializerOptions.h:36-69;
 * KAIST overrides PL-VIWO/config/kaist/kaist_C/config_camera.yaml:32-35) */
typedef struct plv_tri_options {
  double min_dist, max_dist, max_cond_number, max_baseline;
  int refine_features; /* single_gaussnewton after the linear solve (default true) */
} plv_tri_options;

/* plv_triangulate replaces, for all features at once, CamHelper::get_imu_poses / get_cam_poses /
 * feature_triangulation / the reprojection part of moving_consistency (REF: PL-VIWO/src/update/cam/
 * CamHelper.cpp:327-483) and FeatureInitializer::single_triangulation + single_gaussnewton (REF:
 * open_vins/ov_core/src/feat/FeatureInitializer.cpp:30-112,197-375).  Camera poses come from the
 * residual poses of `tr` when given (use_imu_res) or from the estimate polynomial.  ok[f] = 1 when
 * feature f passed every check of the reference; reproj_err[f] (nullable) = mean pixel
 * reprojection error (moving_consistency compares it with 3 px). */
int plv_triangulate(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *tr, const plv_tri_options *opt,
                    double *p_FinG, uint8_t *ok, double *reproj_err);

/* Column order of the stacked Jacobians: [extrinsics 6][intrinsics 8][dt 1] when calibrated, then
 * every clone (6 columns) an observation interpolates over, in first-seen order.  Returns k and
 * fills col_to_state (capacity cap).  Host-side integer logic only. */
int plv_jacobian_columns(const plv_state_view *st, const plv_tracks *tr, int *col_to_state, int cap, int *k_out);

/* plv_build_jacobians replaces, for all features at once, get_feature_jacobian_full +
 * get_interpolated_jacobian + the (optional) polynomial residual pose: writes the whitened
 * per-feature systems Hf [F][3][ld], Hx [F][k][ld], res [F][ld] and rows[f] = 2 * (valid
 * observations of feature f) in the layout plv_msckf_update consumes.  Observations whose time has
 * no bounding clones are dropped (REF: CamHelper.cpp:115-121).  Host buffers out. */
int plv_build_jacobians(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *tr, int k,
                        const int *col_to_state, int ld, int *rows, double *Hf, double *Hx, double *res);

/* The same, leaving the systems staged on the device as the current feature batch (the input of
 * plv_msckf_update_resident): the whole update then runs without any host round trip. */
int plv_build_jacobians_resident(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *tr, int k,
                                 const int *col_to_state, int ld);

/* a19, use_imu_res branch: State::get_interpolated_pose_imu (REF: PL-VIWO/src/state/State.cpp:1138-1155) on
 * State::cpis given as a table sorted by time (the std::map's order).  For each query time: the record stored at
 * exactly that time whose clone is in the window (have_cpi, :273-277), else create_new_cpi_linear between the
 * neighbouring records when both were integrated from the same clone (:286-355: geodesic interpolation of
 * R_I0toIk, linear alpha), then R_GtoI = R_I0toIk R_GtoI0, p = p_I0 + v_I0 dt - g dt^2 / 2 + R_GtoI0^T alpha.
 * ok[q] = 0 where the reference would go on to create_new_cpi_integrate (re-integration from the IMU buffer,
 * SURVEY 8(f) rank 2 — not built) or would throw on a missing clone / missing record at clone_t.  The poses are
 * what plv_tracks::res_R / res_p and plv_line_tracks::res_R / res_p take. */
typedef struct plv_cpi_table {
  int n;
  const double *t;        /* [n] strictly ascending                               */
  const double *clone_t;  /* [n] CPI::clone_t: time of the clone it integrates from */
  const double *dt;       /* [n] CPI::dt = t - clone_t                            */
  const double *R_I0toIk; /* [n][9] row-major                                     */
  const double *alpha;    /* [n][3] CPI::alpha_I0toIk                             */
  const double *v;        /* [n][3] CPI::v (global velocity at t)                 */
  double gravity[3];      /* OptionsEstimator::gravity                            */
  const double *Q;        /* optional [n][36] row-major: CPI::Q (plv_cpi_record::Q); read by plv_cpi_noise */
} plv_cpi_table;
int plv_cpi_poses(plv_ctx *ctx, const plv_state_view *st, const plv_cpi_table *cpi, int n_q, const double *t_q, double *R_GtoI,
                  double *p_IinG, uint8_t *ok);
/* The noise side of the same lookup (use_imu_cov): Q of the record stored at t_q, else (1 - lambda) Q0 + lambda Q1 between the
 * neighbouring records of the same clone (create_new_cpi_linear, REF: State.cpp:287-355), and the index of that clone in the
 * view.  ok = 0 where plv_cpi_poses answers 0 as well.  Host arithmetic. */
int plv_cpi_noise(const plv_state_view *st, const plv_cpi_table *cpi, int n_q, const double *t_q, double *Q /*[n_q][36]*/,
                  int *clone_index, uint8_t *ok);

/* ---------------------------------------------------------------------------------------------
 * Line features on the update side (a27-a29).
 * A line is a Pluecker 6-vector line_FinG = [moment n (3); direction v (3)] in the global frame
 * (REF: PL-VIWO/src/update/cam/linefeat/LineFeature.h:22-107, head = moment, tail = direction).
 * ------------------------------------------------------------------------------------------- */
typedef struct plv_line_tracks {
  int n_lines;
  const int *obs_ptr;       /* [n_lines+1] CSR into the observation arrays                               */
  const double *obs_time;   /* [n_obs] measurement time stamps (cam_dt is added inside)                  */
  const float *seg_uv;      /* [n_obs][4] raw pixel end points x1 y1 x2 y2 (LineFeature::line_uvs)       */
  const float *seg_uvn;     /* [n_obs][4] normalised end points (line_uvs_norm); plv_triangulate_lines    */
  const double *line_FinG;  /* [n_lines][6] triangulated lines; input of the Jacobians                   */
  const int *D;             /* [n_lines] structural class 0..3 (LineFeature::D); NULL = all 0             */
  const double *anchor_pt;  /* [n_lines][3] first triangulated point feature lying on the line            */
  const uint8_t *has_pt;    /* [n_lines] anchor_pt valid (REF: LineHelper.cpp:231-247); NULL = none       */
  const double *res_R;      /* optional [n_obs][9] / [n_obs][3]: IMU pose of the residual, as plv_tracks  */
  const double *res_p;
  const double *res_Q;      /* optional [n_obs][36] + [n_obs]: CPI covariance and clone index, as plv_tracks (use_imu_cov) */
  const int *res_clone;
} plv_line_tracks;

/* plv_triangulate_lines replaces, for all lines at once, LineHelper::get_imu_poses / get_cam_poses /
 * line_triangulation (REF: LineHelper.cpp:132-229): with D > 0 and a triangulated point on the line
 * the direction is R_GtoI^T e_D and the moment p x direction (:231-293); otherwise the plane through
 * the first view's end points is intersected with the plane of every later view and the Pluecker
 * results are averaged (:372-495, CompoutePlaneFromPoints :615-623, ComputeLineFramePlanes :625-650).
 * ok[l] = 0 when the reference returns false (fewer than two usable views, all plane pairs with
 * |cos| >= 0.99).  LineFeature::EndPoints is left to the caller (the reference stores an
 * uninitialised vector there, :489-493). */
int plv_triangulate_lines(plv_ctx *ctx, const plv_state_view *st, const plv_line_tracks *lt, double *line_FinG,
                          uint8_t *ok);

/* Column order of the stacked line Jacobians: per observation the four interpolation clones, then the
 * time offset when it is calibrated, first-seen order (REF: LineHelper.cpp:757-788 with
 * State::get_interpolated_jacobian's `order`, State.cpp:905-958).  No extrinsic / intrinsic columns:
 * the reference's line model has no Jacobian with respect to them. */
int plv_line_jacobian_columns(const plv_state_view *st, const plv_line_tracks *lt, int *col_to_state, int cap, int *k_out);

/* plv_build_line_jacobians replaces LineHelper::get_line_feature_jacobian_full (REF: LineHelper.cpp:
 * 733-1024, point-line coupling off as in UpdaterCamera::lines_update :373) for all lines at once:
 * Hf [L][6][ld], Hx [L][k][ld], res [L][ld], rows[l] = 2 * (valid observations).  The reference's
 * arithmetic is kept, including dz/dl built from Identity(2,3) (third column zero) and
 * ln_2 = l0^2 + l1 + l1 (:921-928). */
int plv_build_line_jacobians(plv_ctx *ctx, const plv_state_view *st, const plv_line_tracks *lt, int k,
                             const int *col_to_state, int ld, int *rows, double *Hf, double *Hx, double *res);
/* ... staged on the device as the current feature batch: follow with
 * plv_msckf_update_resident(fdim = 6, res_norm_gate = 0) = UpdaterCamera::lines_update (:371-464). */
int plv_build_line_jacobians_resident(plv_ctx *ctx, const plv_state_view *st, const plv_line_tracks *lt, int k,
                                      const int *col_to_state, int ld);

/* ---------------------------------------------------------------------------------------------
 * Line front-end (a9-a14): TrackLSD behind the boundary.
 * ------------------------------------------------------------------------------------------- */
/* plv_detect_lines replaces the numeric part of TrackLSD::perform_detection_monocular (REF: PL-VIWO/src/
 * update/cam/TrackLSD.cpp:194-235): half-resolution image, FastLineDetector(cfg.line_length_threshold,
 * cfg.line_distance_threshold, cfg.canny_th1, cfg.canny_th2, 3, no merge), x2, drop length^2 <=
 * cfg.line_min_length_px^2.  Works on the equalised image of the current (PLV_PYR_CUR) or previous
 * frame held by the ctx; lines = x1 y1 x2 y2 full-resolution pixels, in the detector's output order. */
int plv_detect_lines(plv_ctx *ctx, int which, float *lines, int cap, int *n_out);
/* With the prefetch on, plv_tracker_feed / _staged / _downsampled run the line detector of the new image themselves: resize + Canny
 * go on the stream ahead of the point front-end, the host walks the edge chains and grows the segments while the device runs LK and
 * RANSAC, and the following plv_line_tracker_feed of the same frame takes the finished detection (same segments as without). */
int plv_line_prefetch_mode(plv_ctx *ctx, int on);
/* The library's own threads (process-wide).  With the prefetch on the line detector's host stage runs on ONE worker thread per
 * context plus up to two segment-fitter threads (default: at most 2, one of them used below 150 000 half-resolution pixels); a thread that waits for work polls for spin_us microseconds before it blocks
 * (default 300: the hand-overs inside one frame follow each other within that time and then cost no wake-up; between frames the
 * threads sleep — at 15 Hz that is at most 3 x 0.3 ms of polling per 66 ms frame).  spin_us = 0: block at once (no polling at all, every hand-over pays a wake-up
 * of tens of microseconds); fit_threads = 0: the worker grows the segments itself after the walk.  Negative arguments only query.
 * Environment: PLV_LINE_SPIN_US, PLV_LINE_FIT_THREADS. */
int plv_line_worker_config(int spin_us, int fit_threads, int *spin_us_out, int *fit_threads_out);
/* plv_line_tracker_feed without waiting for it: with the prefetch on, the rest of TrackLSD::feed_monocular (point-line assignment,
 * matching, classification, track store) runs on the library's line worker thread behind the detection, so the caller can enqueue
 * the point update (plv_camera_update_points) meanwhile; both only read the tracker's output of this frame, as feed_measurement
 * precedes try_update in the reference (UpdaterCamera.cpp:77-116 before :139-195).  Every line entry point joins the feed first;
 * plv_line_tracker_feed_wait joins and returns its status.  Without a detection in flight the call is plv_line_tracker_feed. */
int plv_line_tracker_feed_async(plv_ctx *ctx, double timestamp, const double *vps);
int plv_line_tracker_feed_wait(plv_ctx *ctx);
/* Optional first half of the detector for the image `which`: enqueues the resize, the Canny map and their copies to the host on the
 * ctx's stream and returns.  The next detection of the same image (plv_detect_lines, plv_line_tracker_feed[_points]) then only
 * waits for those copies before its host stage, so that work enqueued in between (plv_perform_matching_launch) runs on the device
 * while the host walks the edge chains.  Feeding another image in between is the caller's error (the maps would be stale). */
int plv_line_detect_launch(plv_ctx *ctx, int which);
/* Second half, ahead of time: runs the detection of image `which` now (after plv_line_detect_launch: only its host stage) and keeps
 * the segments; the next plv_line_tracker_feed[_points] of the SAME frame (no image fed in between) takes them instead of
 * detecting again.  Lets the caller place the host stage between plv_perform_matching_launch and _wait. */
int plv_line_detect_finish(plv_ctx *ctx, int which);
/* Where the detector's chain walk runs: 0 (default) = on the host between the device stages (Canny map down,
 * chains up), 1 = fld_walk_kernel on the device.  Same results; the walk is one dependent chain of
 * ~10^4 scalar steps, which a single GPU lane retires ~25x slower than a host core. */
int plv_line_walk_mode(plv_ctx *ctx, int on_device);

/* TrackLSD::AssignPointToLines (REF :744-792, including its bounding-box test on (x1,y1) / (x2,y2)):
 * kept[q] = input index of the q-th line that owns at least one point; CSR lists per kept line:
 * rel_id / rel_dist (ascending point id, as the reference's std::map) and pos_xy (input order).
 * Capacities: kept n_lines, rel_ptr / pos_ptr n_lines + 1, the lists n_lines * n_pts. Host logic. */
int plv_assign_points_to_lines(const float *lines, int n_lines, const float *pts, const uint64_t *ids, int n_pts, int *kept,
                               int *rel_ptr, uint64_t *rel_id, double *rel_dist, int *pos_ptr, float *pos_xy, int *n_kept);
/* TrackLSD::LineMatch (REF :368-407): match_of_new[i] = index of the last-frame line or -1. Host logic. */
int plv_line_match(const float *lines_new, int n_new, const int *rel_ptr_new, const uint64_t *rel_id_new, const float *lines_last,
                   int n_last, const int *rel_ptr_last, const uint64_t *rel_id_last, int *match_of_new);
/* TrackLSD::LineClassification (REF :318-366): 0..3 for vps = [x y z][2]. */
int plv_line_classification(const float *line, const double *vps);
/* LineHelper::Vanishing_Points (REF: linefeat/LineHelper.cpp:1026-1088): vps[3][2] from R_ItoC (row-major) and
 * the 8 intrinsics. */
int plv_vanishing_points(const double *R_ItoC, const double *K8, double *vps);

/* TrackLSD::feed_monocular (REF :70-192) for the image currently held by the ctx — call after
 * plv_tracker_feed (the point tracker's current observations are the ones lines are attached to).
 * Updates lines_last / ids_last and the line track store (LineFeatureDatabase::update_feature). */
int plv_line_tracker_feed(plv_ctx *ctx, double timestamp, const double *vps);
/* The same with the caller's point observations of this frame (pts [n][2] pixels, ids [n]) in place of the ctx's own point
 * tracker: what an adapter that keeps the reference's TrackKLT object hands to TrackLSD (REF: TrackLSD.cpp:106-107 reads
 * trackFEATS->get_last_obs() / get_last_ids()). */
int plv_line_tracker_feed_points(plv_ctx *ctx, double timestamp, const double *vps, int n, const float *pts, const uint64_t *ids);
int plv_line_tracker_last(plv_ctx *ctx, float *lines, uint64_t *ids, int cap, int *n);
int plv_line_db_size(plv_ctx *ctx);
int plv_line_db_ids(plv_ctx *ctx, uint64_t *ids, int cap, int *n); /* ascending */
/* CSR export of line tracks in the layout of plv_line_tracks (+ LineFeature::D and ::points). */
int plv_line_db_export_tracks(plv_ctx *ctx, const uint64_t *ids, int n_ids, int *obs_ptr, double *obs_time, float *seg_uv,
                              float *seg_uvn, int obs_cap, int *D, int *pts_ptr, int *pt_ids, int pts_cap);
int plv_line_db_remove(plv_ctx *ctx, const uint64_t *ids, int n_ids);

/* ---------------------------------------------------------------------------------------------
 * UpdaterCamera::try_update, point half, as one call (a15-a17, a24, a31).
 * ------------------------------------------------------------------------------------------- */
/* FeatureDatabase::append_new_measurements for one feature (REF: open_vins/ov_core/src/feat/
 * FeatureDatabase.cpp:338-387): appends n observations (time, raw uv, normalised uv) to track `id`,
 * creating it if needed.  The tracker does this itself; the entry point serves callers that keep
 * their own tracker (and the tests). */
int plv_db_append_measurements(plv_ctx *ctx, uint64_t id, int n, const double *t, const float *uv, const float *uvn);

typedef struct plv_update_options {
  int max_msckf;        /* OptionsCamera::max_msckf (REF: CamHelper.cpp:651-653)                      */
  int max_obs;          /* capacity: observations per feature kept in the batch (ld = 2 * max_obs)     */
  double chi2_mult;     /* OptionsCamera::chi2_mult                                                    */
  plv_tri_options tri;  /* FeatureInitializerOptions                                                   */
  double t_prev_frame;  /* t_hist[size-2]: features without a newer observation are used (REF :636)    */
  double state_time;    /* State::time: observations newer than state_time + dt_exp go back to the DB  */
  int window_full;      /* state->clone_window() > window_size: drop observations older than the oldest clone (:733-737) */
  int max_slam;         /* OptionsCamera::max_slam (0 in the shipped configuration)                    */
  int n_slam;           /* State::cam_SLAM_features.size()                                             */
  const uint64_t *slam_ids; /* [n_slam] feature ids of the landmarks in the state                      */
  int init_min_meas;    /* min(window_size * (int)cam_hz - 1, 10): track length that qualifies for SLAM initialisation (:685) */
  const plv_cpi_table *cpi; /* OptionsEstimator::use_imu_res: State::get_interpolated_pose = get_interpolated_pose_imu (REF:
                         * State.cpp:975-977): the pose of every observation (validity, triangulation, residual) comes from this
                         * table through plv_cpi_poses; an observation it cannot serve goes back to the database as one without
                         * bounding clones does (CamHelper.cpp:360-365).  NULL = polynomial poses.                               */
} plv_update_options;

typedef struct plv_update_result {
  int n_pool;       /* features taken out of the database                                   */
  int n_msckf;      /* features that reached msckf_update (triangulated, consistent, capped) */
  int n_accepted;   /* of those, passed the gate                                             */
  int n_rows;       /* stacked rows before compression                                       */
  int n_returned;   /* features handed back to the database                                  */
  int status;       /* PLV_OK, or PLV_E_NOT_PSD from the EKF step (state untouched)          */
  int n_slam;       /* landmarks of the state with a usable track (list PLV_LIST_SLAM)       */
  int n_init;       /* features chosen for SLAM initialisation (list PLV_LIST_INIT)          */
  int n_truncated;  /* tracks with more than max_obs usable observations: the newest max_obs were used */
} plv_update_result;

/* CamHelper::get_features (pool = features_containing_older(2nd-oldest clone) + features_not_containing_newer
 * (t_prev_frame), remove_unusable_measurements, sort by track length, triangulation + moving-consistency
 * (3 px) until max_msckf, REF: CamHelper.cpp:613-707) -> UpdaterCamera::msckf_update (:197-294) on the
 * device-resident covariance -> CamHelper::cleanup_features (:709-738: everything not consumed goes back
 * to the tracker's database; old observations are dropped when the window is full).
 * With max_slam > 0 the call also classifies: landmarks already in the state (opt->slam_ids) with a live track
 * form the SLAM list (:621-628; the reference does not take them out of the database, so they can enter the
 * pool as well), and a consistent feature with >= init_min_meas observations becomes a SLAM-initialisation
 * candidate while n_slam + n_init < max_slam (:685-693) instead of an MSCKF feature.  Both lists are read with
 * plv_camera_update_list and fed to plv_slam_update / plv_slam_initialize by the caller, who applies each dx to
 * its state in between as the reference's EKFUpdate does.
 * Ties in the track-length sort are broken by ascending feature id (the reference's std::sort on an
 * unordered_map leaves them unspecified).  dx (n) is the state correction of EKFUpdate; msckf_ids /
 * accepted (capacity max_msckf, may be NULL) list the features of the update in batch order. */
int plv_camera_update_points(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt, double *dx,
                             plv_update_result *res, uint64_t *msckf_ids, uint8_t *accepted, double *p_FinG);

/* The SLAM / SLAM-init lists of the last plv_camera_update_points (a17; REF: CamHelper.cpp:621-628,685-693), as
 * CSR tracks holding only observations with bounding clones (get_imu_poses, :327-372).  p_FinG (nullable) is
 * the triangulated position for PLV_LIST_INIT and zero for PLV_LIST_SLAM (the landmark lives in the caller's
 * state).  Call with ids = obs_ptr = NULL to query *n_feat only.  A candidate whose initialisation fails goes
 * back to the database with plv_db_append_measurements (REF: UpdaterCamera.cpp:363-364). */
enum { PLV_LIST_SLAM = 0, PLV_LIST_INIT = 1 };
int plv_camera_update_list(plv_ctx *ctx, int which, int cap_feat, int cap_obs, int *n_feat, uint64_t *ids, int *obs_ptr,
                           double *obs_time, float *obs_uv, float *obs_uvn, double *p_FinG);
/* UpdaterCamera::marginalize_slam_features' flags (REF: UpdaterCamera.cpp:118-137): should_marg[i] = 1 when the
 * landmark's feature is no longer in the tracker database or update_fail_count[i] > 1 (nullable = all 0).  The
 * caller then removes flagged landmarks with plv_cov_marginalize (StateHelper::marginalize_slam). */
int plv_slam_marg_flags(plv_ctx *ctx, int n_slam, const uint64_t *slam_ids, const int *update_fail_count, uint8_t *should_marg);

/* ---------------------------------------------------------------------------------------------
 * UpdaterCamera::try_update, line half, as one call (a15/a16, a27-a29, a31).  Call after the caller has
 * applied the dx of plv_camera_update_points to its state (the reference's EKFUpdate does, StateHelper.cpp:
 * 159-160) and rebuilt the state view.
 * ------------------------------------------------------------------------------------------- */
/* LineFeatureDatabase::update_feature for one line track (REF: linefeat/LineFeatureDatabase.cpp:40-76): appends n
 * observations (time, raw end points, normalised end points); D is taken when the track is created;
 * point_ids (n_pts, may be 0) are appended to LineFeature::points. */
int plv_line_db_append_measurements(plv_ctx *ctx, uint64_t id, int n, const double *t, const float *seg_uv,
                                    const float *seg_uvn, int D, const int *point_ids, int n_pts);
/* Triangulated point features the line path may anchor on (the reference's `point_used` database,
 * UpdaterCamera.h / CamHelper.cpp:671-695): plv_camera_update_points records every triangulated feature it
 * examined; this entry point lets a caller with its own point pipeline do the same. */
int plv_point_used_insert(plv_ctx *ctx, uint64_t id, const double *p_FinG, double newest_obs_time);

/* LineHelper::get_line_features' place in try_update (REF: UpdaterCamera.cpp:148-152): the reference forms and triangulates the
 * line pool after get_features and BEFORE msckf_update's correction reaches the state; lines_update then linearises on the
 * updated state.  Call this with the state as it is before the dx of plv_camera_update_points is applied: it records the clone
 * poses / extrinsics / time offset the following plv_camera_update_lines triangulates on (the pool and the triangulation
 * themselves run inside that call; nothing they read changes in between).  Without it plv_camera_update_lines triangulates on
 * the state it is handed.  plv_camera_try_update / plv_camera_frame do this themselves. */
int plv_camera_get_line_features(plv_ctx *ctx, const plv_state_view *st);

/* LineHelper::get_line_features (REF: linefeat/LineHelper.cpp:19-72: pool, remove_unusable_measurements (0.01 s
 * margins), sort by track length, line_triangulation — on the state recorded by plv_camera_get_line_features when
 * there is one) -> UpdaterCamera::lines_update (UpdaterCamera.cpp:371-464) on `st`
 * -> LineHelper::cleanup_lines (:522-553).  opt->max_msckf is ignored (no cap on lines).  line_ids / accepted
 * (capacity cap, may be NULL) list the lines of the update in batch order. */
int plv_camera_update_lines(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt, double *dx,
                            plv_update_result *res, uint64_t *line_ids, uint8_t *accepted, double *line_FinG, int cap);

/* x <- x [+] dx for every variable of the state in one call (REF: StateHelper::EKFUpdate, PL-VIWO/src/state/StateHelper.cpp:156-160,
 * which calls Type::update(dx.block(id, 0, size, 1)) variable by variable: ov_type::Vec adds, JPLQuat composes on the left, PoseJPL is
 * a quaternion at id and a 3-vector at id + 3).  `out` of a quaternion (nullable) receives its rotation matrix [9]; `mirror` (nullable)
 * receives a second copy of what the variable now holds (value of a vector, rotation matrix of a quaternion), e.g. the field of a
 * plv_state_view the caller keeps current.  Host arithmetic. */
enum { PLV_VAR_VEC = 0, PLV_VAR_QUAT = 1 };
typedef struct plv_state_var {
  int kind;       /* PLV_VAR_* */
  int id;         /* covariance index of the variable's error state (Type::id()) */
  int size;       /* values of a vector (its error state has the same size); ignored for a quaternion (4 values, 3 error states) */
  double *val;    /* vector [size] / quaternion [4] (JPL: x y z w), updated in place */
  double *out;    /* quaternion: rotation matrix [9], row-major, or NULL */
  double *mirror; /* or NULL */
} plv_state_var;
int plv_state_boxplus(int n_var, const plv_state_var *vars, const double *dx, int n_dx);

/* UpdaterCamera::try_update as ONE call (REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:139-195 without the SLAM branch, which needs
 * the two-call form): plv_camera_update_points; its dx applied to the caller's state (plv_state_boxplus over `vars` when the update
 * was accepted, StateHelper.cpp:156-160, and plv_set_camera_intrinsics when the intrinsics are calibrated, :163-168); then, when
 * opt_lines is given, the join of an asynchronous line feed, plv_camera_update_lines on the updated state and its dx applied the
 * same way.  `st` must stay current under plv_state_boxplus: its clone arrays are the `val` / `out` arrays of the clone variables
 * and its calibration fields are their `mirror`s.  The point half's database hand-back (CamHelper::cleanup_features) is run while
 * the line update executes on the device.  Outputs as in the two calls; line_db_size = LineFeatureDatabase size after the feed. */
typedef struct plv_try_update {
  const plv_update_options *opt_points;
  const plv_update_options *opt_lines;   /* NULL: points only */
  int n_var;
  const plv_state_var *vars;             /* every variable of the state (n_var may be 0: dx is returned, nothing applied) */
  double *dx_points, *dx_lines;          /* [cov_n] each */
  plv_update_result *res_points, *res_lines;
  uint64_t *msckf_ids;                   /* capacity opt_points->max_msckf (nullable, like the next two) */
  uint8_t *msckf_accepted;
  double *p_FinG;
  uint64_t *line_ids;                    /* capacity line_cap (nullable, like the next two) */
  uint8_t *line_accepted;
  double *line_FinG;
  int line_cap;
  int line_db_size;                      /* out */
} plv_try_update;
int plv_camera_try_update(plv_ctx *ctx, const plv_state_view *st, plv_try_update *io);
/* One camera frame as one call: UpdaterCamera::feed_measurement (REF: UpdaterCamera.cpp:77-116: TrackKLT::feed_new_camera, then
 * LineHelper::Vanishing_Points + TrackLSD::feed_new_camera when lines are on) followed by try_update (:139-195) through
 * plv_camera_try_update.  The image comes from HBM slot `slot` (plv_image_stage; slot >= 0) or from the host (img, stride).
 * update == NULL: feed only (not initialised yet / fewer than intr_order + 1 clones, CamHelper.cpp:615-616); with lines on and
 * update->opt_lines == NULL the lines are tracked but not used.  `st` is the state at the time of the call (the extrinsics and
 * intrinsics the vanishing points are made from are st's).  line_db_size: LineFeatureDatabase size after the feed. */
typedef struct plv_camera_frame_io {
  double timestamp;
  int slot;                /* >= 0: staged image; < 0: img / stride */
  const uint8_t *img;
  int stride;
  const uint8_t *mask;     /* W x H, > 127 = masked out; may be NULL */
  int use_lines;           /* OptionsCamera::use_lines */
  plv_try_update *update;  /* or NULL */
  int line_db_size;        /* out */
} plv_camera_frame_io;
int plv_camera_frame(plv_ctx *ctx, const plv_state_view *st, plv_camera_frame_io *io);

/* ---------------------------------------------------------------------------------------------
 * In-state (SLAM) landmarks (a30) on the device-resident covariance, one landmark per call as the reference
 * loops.  Inactive at the shipped configuration (max_slam: 0).
 * ------------------------------------------------------------------------------------------- */
/* UpdaterCamera::slam_update for one landmark (REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:296-338): H is the
 * system of get_feature_jacobian_full with the landmark's columns appended ([Hx | Hf], rows x k col-major,
 * ld), R = I (whitened rows).  Chi2Check (threshold chi2_mult * q95[rows]) then EKFUpdate.  *accepted = 0
 * when the gate fails or rows < 2 (nothing changed); PLV_E_NOT_PSD when EKFUpdate rejects. */
int plv_slam_update(plv_ctx *ctx, int rows, int k, int ld, const double *H, const double *res, const int *col_to_state,
                    double chi2_mult, uint8_t *accepted, double *dx);
/* StateHelper::initialize (REF: PL-VIWO/src/state/StateHelper.cpp:357-439) for a 3-dof landmark: Givens split of
 * [Hf (3 cols) | Hx (k cols) | res], Mahalanobis gate on the updating rows (threshold on ALL rows, :415),
 * initialize_invertible (:495-600: P_LL = H_L^-1 (H P H^T + I) H_L^-T, cross terms -P H^T H_L^-T, the suspicious
 * / negative-diagonal rejections), EKFUpdate with the updating rows, revert when that fails (:430-435).
 * On success *ok = 1, the covariance has grown to (n+3) with the landmark at index n, dx_init (3) is the
 * landmark's own correction H_L^-1 res_init and dx (n+3) the EKF correction that follows it. */
int plv_slam_initialize(plv_ctx *ctx, int rows, int k, int ld, const double *Hf, const double *Hx, const double *res,
                        const int *col_to_state, double chi2_mult, uint8_t *ok, double *dx_init, double *dx);
/* StateHelper::marginalize (REF: StateHelper.cpp:235-303): drop rows / columns [id, id + size). */
int plv_cov_marginalize(plv_ctx *ctx, int id, int size);

/* ---------------------------------------------------------------------------------------------
 * IMU propagation + window maintenance on the device-resident covariance.
 * ------------------------------------------------------------------------------------------- */
typedef struct plv_imu_state { /* ov_type::IMU value() and fej() (REF: PL-VIWO/src/types/IMU.h) */
  double q[4], p[3], v[3], bg[3], ba[3]; /* JPL quaternion global->IMU, p_IinG, v_IinG, gyro / accel bias */
  double q_fej[4], p_fej[3], v_fej[3];
} plv_imu_state;
typedef struct plv_imu_noise { /* OptionsIMU sigma_w / sigma_wb / sigma_a / sigma_ab, OptionsEstimator::gravity */
  double sigma_w, sigma_wb, sigma_a, sigma_ab;
  double gravity[3];
} plv_imu_noise;
typedef struct plv_cpi_accum { /* CpiV1's running state between two Propagator::reset_cpi calls */
  double clone_t, DT;
  double R_k2tau[9], alpha_tau[3], beta_tau[3];
  double b_w_lin[3], b_a_lin[3];
  double v_clone[3];   /* cpis.at(cpi_clone_t).v */
  double P_meas[225];  /* 15 x 15 row-major */
} plv_cpi_accum;
typedef struct plv_cpi_record { /* State::CPI as Propagator::propagate fills it (REF: Propagator.cpp:65-82) */
  double t, dt, clone_t;
  double R_I0toIk[9], alpha[3], v[3], w[3];
  double Q[36];        /* 6 x 6 row-major: P_meas blocks (0,0) (0,12) (12,0) (12,12) */
} plv_cpi_record;

/* Propagator::select_imu_readings + interpolate_data (REF: PL-VIWO/src/state/Propagator.cpp:93-152,320-331) on an
 * ascending IMU buffer t / wm [n][3] / am [n][3].  *n_out = 0 and PLV_OK with *ok = 0 where the reference returns
 * false (fewer than two samples, time1 <= time0, window outside the buffer).  Host logic. */
int plv_select_imu_readings(int n, const double *t, const double *wm, const double *am, double time0, double time1, int cap,
                            double *out_t, double *out_wm, double *out_am, int *n_out, int *ok);
/* Propagator::reset_cpi for the accumulator (REF: Propagator.cpp:333-357); the caller keeps State::cpis. */
void plv_reset_cpi(plv_cpi_accum *acc, const plv_imu_state *imu, double clone_t);
/* Propagator::propagate over already selected samples (REF: Propagator.cpp:30-91): per interval predict_mean_rk4
 * (:240-318) + predict_and_compute (:154-238: F, G Qc G^T with first-estimate Jacobians, value and fej replaced by the
 * propagated mean), Phi = F Phi, Qd = F Qd F^T + Qdi symmetrised, optionally CpiV1::feed_IMU (REF: open_vins/ov_core/
 * src/cpi/CpiV1.cpp:32-315, means + RK4 measurement covariance, imu_avg) with one State::CPI record per interval;
 * then StateHelper::EKFPropagation (REF: PL-VIWO/src/state/StateHelper.cpp:20-92) of the resident covariance with the
 * IMU block at imu_id.  imu is updated in place; acc / records / Phi / Qd (15 x 15 row-major) are nullable. */
int plv_propagate(plv_ctx *ctx, plv_imu_state *imu, const plv_imu_noise *noise, int n_data, const double *t, const double *wm,
                  const double *am, plv_cpi_accum *acc, plv_cpi_record *records, int n, int imu_id, double *Phi, double *Qd);
/* State::create_new_cpi_integrate (REF: PL-VIWO/src/state/State.cpp:357-415), the last stage of have_cpi: CpiV1 from the
 * clone at clone_t to t_given over the IMU buffer (select_imu_readings between the two, reversed when the clone is the later
 * one), producing the State::CPI record the caller inserts into its table; plv_cpi_poses then answers t_given from it.
 * R_GtoI_clone = clones.at(clone_t)->Rot(); v_clone / bg / ba = cpis.at(clone_t).v / .bg / .ba.  *ok = 0 when the IMU
 * buffer does not cover the interval (the reference returns false). */
int plv_cpi_integrate(plv_ctx *ctx, const plv_imu_noise *noise, double t_given, double clone_t, const double *R_GtoI_clone,
                      const double *v_clone, const double *bg, const double *ba, int n_imu, const double *t, const double *wm,
                      const double *am, plv_cpi_record *out, int *ok);
/* State::closest_clone_time_not_imu (REF: State.cpp:524-538) as intended: the clone nearest to t_given, leaving out the
 * newest entry when it is the IMU pose itself (exclude_newest).  The reference's loop compares against
 * abs(t - clone.first) with t read before its first assignment (uninitialised on the first pass), so its result is not
 * defined; this is the evident intent.  Host logic. */
int plv_closest_clone_time(const plv_state_view *st, int exclude_newest, double t_given, double *clone_t, int *found);
/* SystemManager::get_next_clone_time (REF: PL-VIWO/src/core/SystemManager.cpp:172-267) for one camera: the time of the next
 * clone, snapped to the nearest camera measurement.  dynamic_cloning's choice of clone_freq (:290-310, from the
 * interpolation-error tables of the configuration) is the caller's: pass the frequency in use.  *ok = 0 where the
 * reference returns false (no IMU coverage, or no measurement at / before the desired time).  Host logic. */
typedef struct plv_clone_schedule {
  int n_clones;                     /* state->clones.size() */
  double state_time, meas_t;        /* State::time, the time stamp of the IMU message being fed */
  double newest_clone_time, second_newest_clone_time;
  int newest_is_imu_pose;           /* clones.at(newest)->id() == imu->id() */
  int clone_freq;                   /* OptionsEstimator::clone_freq (after dynamic_cloning) */
  int n_sensor_times;               /* UpdaterCamera::t_hist of the camera, ascending */
  const double *sensor_times;
  double sensor_dt;                 /* State::cam_dt */
  double imu_oldest_t, imu_newest_t;/* Propagator::imu_data front / back */
  int wheel_enabled;
} plv_clone_schedule;
int plv_next_clone_time(const plv_clone_schedule *in, double *clone_time, int *ok);
/* StateHelper::clone as augment_clone uses it (REF: StateHelper.cpp:175-201,305-355): the resident covariance grows
 * from n to n + size, the new rows / columns copy those at src_id (the IMU pose: size 6). */
int plv_cov_clone(plv_ctx *ctx, int n, int src_id, int size);

/* ---------------------------------------------------------------------------------------------
 * Wheel odometry updater (REF: PL-VIWO/src/update/wheel/UpdaterWheel.cpp).
 * Wheel3DAng (the shipped configuration): m1 / m2 = left / right wheel angular velocity.  The 3D types measure the
 * relative rotation and translation of the odometry frame (6 rows), the 2D types yaw and planar translation (3 rows).
 * ------------------------------------------------------------------------------------------- */
enum { PLV_WHEEL3D_ANG = 0, PLV_WHEEL3D_LIN = 1, PLV_WHEEL3D_CEN = 2, PLV_WHEEL2D_ANG = 3, PLV_WHEEL2D_LIN = 4,
       PLV_WHEEL2D_CEN = 5 }; /* REF: WheelTypes.h */
typedef struct plv_wheel_options { /* OptionsWheel */
  int type;
  double noise_w, noise_v, noise_p;
  int do_calib_ext, do_calib_dt, do_calib_int;
  double chi2_mult;
} plv_wheel_options;
typedef struct plv_wheel_state { /* what compute_linear_system_3D reads from the State (UpdaterWheel.cpp:327-424) */
  double intr[3];                 /* wheel_intrinsic: r_l, r_r, base length                   */
  double R_ItoO[9], p_IinO[3];    /* wheel_extrinsic                                          */
  double R0[9], p0[3], R0_fej[9], p0_fej[3]; /* clones.at(time0): Rot(), pos(), first estimates */
  double R1[9], p1[3], R1_fej[9], p1_fej[3]; /* clones.at(time1)                                */
  double w0[3], v0[3], w1[3], v1[3];         /* cpis.at(time0 / time1).w / .v (do_calib_dt only) */
  int pose0_id, pose1_id, ext_id, dt_id, intr_id; /* covariance indices (Type::id()); unused ones -1 */
} plv_wheel_state;

/* UpdaterWheel::select_wheel_data + interpolate_data (REF: UpdaterWheel.cpp:142-215,784-794) on an ascending buffer
 * t / m1 / m2 [n].  *ok = 0 where the reference returns false.  Host logic. */
int plv_select_wheel_data(int n, const double *t, const double *m1, const double *m2, double time0, double time1, int cap,
                          double *out_t, double *out_m1, double *out_m2, int *n_out, int *ok);
/* preintegration_3D / _2D (+ preintegration_intrinsics_*) over the selected samples and compute_linear_system_3D / _2D (REF:
 * :86-101, 217-325, 327-424, 426-500, 502-646, 648-782) on the device: rows = 6 (3D) or 3 (2D); res (rows), H (rows x k,
 * col-major, k = 12 [+6 ext] [+1 dt] [+3 intr]), Cov (rows x rows row-major), col_to_state (k).  Capacities: H 6 x 22, res 6,
 * Cov 36.  Also returns the preintegrated measurement when asked (nullable): R_3D (9) / p_3D (3); for the 2D types
 * p_3D = (theta, x, y) and R_3D is the identity. */
int plv_wheel_linear_system(plv_ctx *ctx, const plv_wheel_options *opt, const plv_wheel_state *st, int n_data, const double *t,
                            const double *m1, const double *m2, double *H, double *res, double *Cov, int *col_to_state, int *k,
                            int *rows, double *R_3D, double *p_3D);
/* UpdaterWheel::update from the selected samples on (REF: :72-139): the system above, Chi2Check with the full Cov_3D and
 * StateHelper::EKFUpdate on the resident covariance.  The full (6 x 6 or 3 x 3) noise is applied by whitening (H <- L^-1 H, res <- L^-1 res
 * with Cov_3D = L L^T), which is the same update.  *accepted = 0 when the gate fails; PLV_E_NOT_PSD as plv_ekf_update. */
int plv_wheel_update(plv_ctx *ctx, const plv_wheel_options *opt, const plv_wheel_state *st, int n_data, const double *t,
                     const double *m1, const double *m2, uint8_t *accepted, double *dx);

/* ---------------------------------------------------------------------------------------------
 * State initialisation (REF: PL-VIWO/src/init/Initializer.cpp:93-113 picks one of the two).
 * imustate [17] = time, q_GtoI (JPL x y z w), p_IinG, v_IinG, bias_g, bias_a, as Initializer::set_state takes it
 * (REF: Initializer.cpp:175-180; the initial covariance is cov_size * I_15 there).  Host logic, no device work.
 * ------------------------------------------------------------------------------------------- */
/* I_Initializer::initialization (REF: PL-VIWO/src/init/imu/I_Initializer.cpp:44-150) on the ascending IMU buffer:
 * waits for two windows of window_time, wants the older one still (accel. std <= imu_thresh) and the newer one moving
 * (>= imu_thresh); orientation from the mean specific force of the still window, biases from its means.  *ok = 0
 * where the reference returns false. */
int plv_init_imu_static(int n, const double *t, const double *wm, const double *am, double window_time, double imu_thresh,
                        const double *gravity, double *imustate, int *ok);
typedef struct plv_iw_init_options { /* IW_Initializer::IW_Initializer_Options (REF: IW_Initializer.h:50-58) */
  int wheel_type;                 /* PLV_WHEEL*                                                          */
  double intrinsics[3];           /* r_l, r_r, base length                                               */
  double R_ItoO[9], p_IinO[3];    /* wheel_extrinsic->Rot() (row-major), ->pos()                         */
  double toff;                    /* wheel_dt                                                            */
  double threshold;               /* init.imu_wheel_thresh                                               */
  double gravity[3];
  int imu_gravity_aligned;        /* init.imu_gravity_aligned: gravity in {I0} = `gravity`, no estimate  */
} plv_iw_init_options;
typedef struct plv_iw_init_state { /* what IW_Initializer keeps between calls (REF: IW_Initializer.h:104-108) */
  int cnt_smooth;
  int reserved;
  double prev_init[12];           /* bg, ba, g in {I0}, v in {I0}                                        */
} plv_iw_init_state;
void plv_iw_init_reset(plv_iw_init_state *state); /* cnt_smooth = -1 */
/* IW_Initializer::initialization (REF: PL-VIWO/src/init/imu_wheel/IW_Initializer.cpp:44-104) on the IMU buffer
 * (Propagator::imu_data) and the wheel buffer (UpdaterWheel::data_stack): common window, gyro bias against the wheel
 * yaw rate, velocity from the wheels, gravity direction (mean of per-interval estimates when every wheel reading is
 * zero, the norm-constrained least squares of :266-432 otherwise), accelerometer bias, residual check, and success
 * only after more than three consecutive calls whose 12-vector moved by less than `threshold`.  *mode (nullable) =
 * 0 static / 1 dynamic / -1 not enough data; init12 (nullable) receives this call's [bg ba g_I0 v_I0] when the
 * stages all succeeded.  Where a nested select_imu_readings fails the reference asserts; this returns *ok = 0. */
int plv_init_imu_wheel(const plv_iw_init_options *opt, plv_iw_init_state *state, int n_imu, const double *t, const double *wm,
                       const double *am, int n_wheel, const double *tw, const double *m1, const double *m2, double *imustate, int *ok,
                       int *mode, double *init12);
/* StateHelper::EKFUpdate refreshes the tracker's camera model after every update when the intrinsics are calibrated
 * online (REF: PL-VIWO/src/state/StateHelper.cpp:163-168): same for the ctx (undistortion, RANSAC threshold). */
int plv_set_camera_intrinsics(plv_ctx *ctx, const double *K8);
/* plv_config::win_size after the context was made (REF: TrackKLT.h:143-144 `win_size`, the winSize of calcOpticalFlowPyrLK at
 * TrackKLT.cpp:857-858 and of buildOpticalFlowPyramid at :71): odd, 3 .. 21.  The pyramids are laid out for the window they were
 * made with (a level exists while both its sides exceed the window): a window that would change the number of levels is refused
 * (PLV_E_BADARG) once an image was fed.  Takes effect with the next flow. */
int plv_set_lk_window(plv_ctx *ctx, int win_size);
/* ov_type::JPLQuat::update for n orientations at once (REF: open_vins/ov_core/src/types/JPLQuat.h:62-73): q [n][4] <- quatnorm([dth / 2, 1]) (x) q
 * with w >= 0 (dth [n][3]; NULL = leave q as it is), and R [n][9] (nullable) = quat_2_Rot(q) row-major.  Host arithmetic for the driver's
 * application of dx to the clone window. */
void plv_jpl_left_update(int n, double *q, const double *dth, double *R);


/* ---------------------------------------------------------------------------------------------
 * Trajectory I/O and the ATE evaluator: the accuracy half of the metric.
 * Poses are [n][7] = tx ty tz qx qy qz qw (JPL quaternion), as the reference logs and loads them.
 * ------------------------------------------------------------------------------------------- */
/* State_Logger's header line and one `t x y z qx qy qz qw [12 covariance terms]` line (REF: PL-VIWO/src/utils/
 * State_Logger.h:166-205: ios::fixed, precision 6, covariance terms precision 10; P = 6x6 row-major marginal of the
 * IMU pose [orientation; position], NULL = the 8-column form).  Return the number of characters written. */
int plv_traj_header(char *buf, int cap);
int plv_traj_format(char *buf, int cap, double t, const double *p, const double *q, const double *P);
/* ov_eval::Loader::load_data (REF: open_vins/ov_eval/src/utils/Loader.cpp:26-90): space-separated, lines starting
 * with '#' skipped, 8 or 20 columns.  times [cap], poses [cap][7], cov_ori / cov_pos [cap][9] (nullable).  *n = lines
 * parsed, *n_cov = lines that carried a covariance.  Call with times = NULL to count.  PLV_E_BADARG when the file
 * cannot be opened or holds no pose (the reference exits), PLV_E_CAPACITY when n > cap. */
int plv_traj_load(const char *path, int cap, double *times, double *poses, double *cov_ori, double *cov_pos, int *n, int *n_cov);
/* Loader::get_total_length (:388-398) */
double plv_traj_length(int n, const double *poses);
/* AlignUtils::perform_association (REF: open_vins/ov_eval/src/alignment/AlignUtils.cpp:101-189): for every estimate
 * the closest ground-truth time within max_difference of est + offset, the ground-truth pointer only ever
 * advancing.  est_idx / gt_idx (capacity n_est) list the matched pairs. */
int plv_traj_associate(double offset, double max_difference, int n_est, const double *est_times, int n_gt, const double *gt_times,
                       int *est_idx, int *gt_idx, int *n_match);
enum { PLV_ALIGN_POSYAW = 0, PLV_ALIGN_POSYAW_SINGLE = 1, PLV_ALIGN_SE3 = 2, PLV_ALIGN_SE3_SINGLE = 3, PLV_ALIGN_SIM3 = 4,
       PLV_ALIGN_NONE = 5 }; /* REF: AlignTrajectory.cpp:31-53 (rosbag.launch uses posyaw) */
typedef struct plv_stats { double min, max, median, mean, rmse, std, ninetynine; } plv_stats; /* REF: ov_eval Statistics.h:72-119 */
/* ResultTrajectory's alignment + calculate_ate on n associated pose pairs (REF: open_vins/ov_eval/src/calc/
 * ResultTrajectory.cpp:55-121): align est to gt with AlignTrajectory::align_trajectory (Umeyama's two passes over
 * the positions and the per-pose errors run on the device), aligned[i] = {s R p_est + t, q_est (x) q_ESTtoGT^-1},
 * ori_err[i] = |log_so3(R_aligned^T R_gt)| in degrees, pos_err[i] = |p_gt - p_aligned|.  Every output is nullable. */
int plv_traj_ate(plv_ctx *ctx, int method, int n, const double *est_poses, const double *gt_poses, int n_aligned, double *R,
                 double *t, double *s, double *aligned, double *ori_err, double *pos_err, plv_stats *ori, plv_stats *pos);

#ifdef __cplusplus
}
#endif
#endif /* PLVIWO_H */
