// jacobian_api.hip — plv_jacobian_columns / plv_build_jacobians[_resident] (include/plviwo.h).
// Host side: integer bookkeeping only (column order, CSR -> per-observation feature index, one
// packed upload); all arithmetic is in jacobian_kernel.  No CPU compute path.
#include <algorithm>
#include <vector>

#include <chrono>
#include "jacobian_kernels.hpp"
#include "nullspace_core.hpp"
#include "update_kernels.hpp"
#include "update_state.hpp"

using namespace plv;

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

namespace {

// State::bounding_times + bounding_poses_n (order 3), host copy used for the column bookkeeping.
// REF: PL-VIWO/src/state/State.cpp:1023-1136
int bounding_start_host(const plv_state_view &st, double t) {
  const int N = st.n_clones;
  if (N < 4) return -1;
  const double *ct = st.clone_time;
  if (t < ct[0] - st.dt_exp || t > ct[N - 1] + st.dt_exp) return -1;
  if (t > ct[N - 1]) return -1;
  int n_b = -1;
  for (int i = 0; i < N - 1; ++i)
    if (ct[i] - st.dt_exp <= t && t <= ct[i + 1] + st.dt_exp) {
      n_b = i;
      break;
    }
  if (n_b < 0) return -1;
  int start = n_b - 1;
  if (n_b - 1 < 0)
    start = 0;
  else if (n_b + 2 >= N)
    start = N - 4;
  if (start < 0 || start + 4 > N) return -1;
  return start;
}
// the same answer from a small memo: a batch's observations carry the time stamps of the last few frames (~16 distinct values), asked
// for hundreds of times per call of the column functions below (8-9 us of the caller's thread in front of the line launch)
struct BoundingStartMemo {
  const plv_state_view &st;
  double t[24];
  int s0[24];
  int n = 0;
  explicit BoundingStartMemo(const plv_state_view &s) : st(s) {}
  int operator()(double tq) {
    for (int i = n - 1; i >= 0; --i)
      if (t[i] == tq) return s0[i];
    const int r = bounding_start_host(st, tq);
    if (n < 24) t[n] = tq, s0[n++] = r;
    return r;
  }
};

int check_views(const plv_state_view *st, const plv_tracks *tr) {
  if (!st || !tr || st->n_clones < 1 || !st->clone_time || !st->clone_R || !st->clone_p || !st->clone_R_fej ||
      !st->clone_p_fej || !st->clone_state_id || tr->n_feat < 1 || !tr->obs_ptr || !tr->obs_time || !tr->obs_uv ||
      !tr->p_FinG || !tr->p_FinG_fej) {
    set_last_error("jacobians: null view field");
    return PLV_E_BADARG;
  }
  if (st->intr_order != 3) {
    set_last_error("jacobians: only intr_order = 3 is built (got %d)", st->intr_order);
    return PLV_E_BADARG;
  }
  if ((tr->res_R == nullptr) != (tr->res_p == nullptr)) return PLV_E_BADARG;
  return PLV_OK;
}

// Packs every input array into one pinned block, uploads it with one copy and fills JacParams.
struct StageExtra {  // optional riders of the packed block (one-submission update): normalised coordinates, admissibility flags
  const float *uvn = nullptr;
  const uint8_t *flags = nullptr;
  const float *d_uvn = nullptr;
  const uint8_t *d_flags = nullptr;
  // speculative submission (plv_points_spec): per candidate the flow index, the meta bits and the count of valid older observations
  // (SpecSelectArgs, jacobian_kernels.hpp); out: where they, the ranges' ends the device writes and the staged arrays it patches sit
  const int *spec_li = nullptr;
  const uint8_t *spec_meta = nullptr, *spec_prevalid = nullptr;
  const int *d_spec_li = nullptr;
  const uint8_t *d_spec_meta = nullptr, *d_spec_prevalid = nullptr;
  int *d_obs_end = nullptr;
  float *d_obs_uv = nullptr;
};
int stage_inputs(plv_ctx *ctx, plv_ctx_update_state *us, const plv_state_view *st, const plv_tracks *tr, int k,
                 const int *col_to_state, int ld, JacParams &P, StageExtra *ex = nullptr) {
  const int N = st->n_clones, F = tr->n_feat, nobs = tr->obs_ptr[F];
  if (nobs < 1) {
    set_last_error("jacobians: no observations");
    return PLV_E_BADARG;
  }
  auto col_of = [&](int sid) {
    if (sid < 0) return -1;
    for (int j = 0; j < k; ++j)
      if (col_to_state[j] == sid) return j;
    return -1;
  };
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_time = take(8 * N), o_R = take(72 * N), o_p = take(24 * N), o_Rf = take(72 * N), o_pf = take(24 * N),
               o_ccol = take(4 * N), o_ptr = take(4 * (F + 1)), o_of = take(4 * nobs), o_ot = take(8 * nobs),
               o_uv = take(8 * nobs), o_pg = take(24 * F), o_pgf = take(24 * F),
               o_rR = tr->res_R ? take(72 * nobs) : 0, o_rp = tr->res_R ? take(24 * nobs) : 0, o_cols = take(4 * (size_t)k),
               o_xuvn = ex && ex->uvn ? take(8 * nobs) : 0, o_xfl = ex && ex->flags ? take(F) : 0,
               o_rQ = tr->res_Q ? take(288 * (size_t)nobs) : 0, o_rc = tr->res_Q ? take(4 * (size_t)nobs) : 0,
               o_sli = ex && ex->spec_li ? take(4 * (size_t)F) : 0, o_sme = ex && ex->spec_li ? take(F) : 0, o_spv = ex && ex->spec_li ? take(F) : 0,
               o_send = ex && ex->spec_li ? take(4 * (size_t)F) : 0;
  const size_t total = off;
  TRY(us->h_jin.reserve(total));
  TRY(us->jin.reserve(total));
  char *h = us->h_jin.as<char>();
  memcpy(h + o_cols, col_to_state, 4 * (size_t)k);
  if (ex && ex->uvn) memcpy(h + o_xuvn, ex->uvn, 8 * nobs);
  if (ex && ex->flags) memcpy(h + o_xfl, ex->flags, F);
  if (ex && ex->spec_li) {
    memcpy(h + o_sli, ex->spec_li, 4 * (size_t)F);
    memcpy(h + o_sme, ex->spec_meta, F);
    memcpy(h + o_spv, ex->spec_prevalid, F);
    memcpy(h + o_send, tr->obs_ptr + 1, 4 * (size_t)F);  // (overwritten by spec_select_kernel)
  }
  memcpy(h + o_time, st->clone_time, 8 * N);
  memcpy(h + o_R, st->clone_R, 72 * N);
  memcpy(h + o_p, st->clone_p, 24 * N);
  memcpy(h + o_Rf, st->clone_R_fej, 72 * N);
  memcpy(h + o_pf, st->clone_p_fej, 24 * N);
  int *ccol = (int *)(h + o_ccol);
  for (int i = 0; i < N; ++i) ccol[i] = col_of(st->clone_state_id[i]);
  memcpy(h + o_ptr, tr->obs_ptr, 4 * (F + 1));
  int *of = (int *)(h + o_of);
  for (int f = 0; f < F; ++f) {
    if (tr->obs_ptr[f + 1] < tr->obs_ptr[f]) return PLV_E_BADARG;
    for (int o = tr->obs_ptr[f]; o < tr->obs_ptr[f + 1]; ++o) of[o] = f;
  }
  memcpy(h + o_ot, tr->obs_time, 8 * nobs);
  memcpy(h + o_uv, tr->obs_uv, 8 * nobs);
  memcpy(h + o_pg, tr->p_FinG, 24 * F);
  memcpy(h + o_pgf, tr->p_FinG_fej, 24 * F);
  if (tr->res_R) {
    memcpy(h + o_rR, tr->res_R, 72 * nobs);
    memcpy(h + o_rp, tr->res_p, 24 * nobs);
  }
  if (tr->res_Q) {
    if (!tr->res_clone) return PLV_E_BADARG;
    memcpy(h + o_rQ, tr->res_Q, 288 * (size_t)nobs);
    memcpy(h + o_rc, tr->res_clone, 4 * (size_t)nobs);
  }
  // measurement knob PLV_KNOB_INPUTS_PINNED: no upload — the kernels read the pinned staging block over PCIe (every byte once or a
  // few times; what a workgroup reuses it keeps in LDS)
  const bool pinned_inputs = plv::knob(plv::PLV_KNOB_INPUTS_PINNED) && !(ex && ex->spec_li);  // (the speculative batch is patched on the device)
  // (an upload by a kernel of the ctx stream instead of the copy command was measured, alternating frame by frame: no difference)
  if (ex && ex->spec_li) {
    // the speculative batch is staged while the frame's flow occupies the ctx stream: its upload goes onto a stream of its own at once
    // (a copy command behind the flow would sit between the flow's last kernel and the update's first); the ctx stream waits for it
    if (!us->spec_stream) {
      PLV_HIP_CHECK(hipStreamCreateWithFlags(&us->spec_stream, hipStreamNonBlocking));
      PLV_HIP_CHECK(hipEventCreateWithFlags(&us->spec_ev, hipEventDisableTiming));
    }
    PLV_HIP_CHECK(plv::memcpy_async(us->jin.p, h, total, hipMemcpyHostToDevice, us->spec_stream));
    PLV_HIP_CHECK(hipEventRecord(us->spec_ev, us->spec_stream));
    PLV_HIP_CHECK(hipStreamWaitEvent(ctx->stream, us->spec_ev, 0));
  } else if (!pinned_inputs)
    PLV_HIP_CHECK(plv::memcpy_async(us->jin.p, h, total, hipMemcpyHostToDevice, ctx->stream));
  const char *d = pinned_inputs ? (const char *)h : us->jin.as<char>();
  P.n_clones = N;
  P.clone_time = (const double *)(d + o_time);
  P.clone_R = (const double *)(d + o_R);
  P.clone_p = (const double *)(d + o_p);
  P.clone_R_fej = (const double *)(d + o_Rf);
  P.clone_p_fej = (const double *)(d + o_pf);
  P.clone_col = (const int *)(d + o_ccol);
  memcpy(P.R_ItoC, st->R_ItoC, 72);
  memcpy(P.p_IinC, st->p_IinC, 24);
  memcpy(P.K, st->intrinsics, 64);
  P.cam_dt = st->cam_dt;
  P.dt_exp = st->dt_exp;
  P.sigma_pix = st->sigma_pix;
  P.intr_ori_cov = st->intr_ori_cov;
  P.intr_pos_cov = st->intr_pos_cov;
  P.use_pol_cov = st->use_pol_cov;
  P.use_imu_cov = st->use_imu_cov && tr->res_Q ? 1 : 0;
  P.intr_err_mlt = st->intr_err_mlt;
  P.res_Q = tr->res_Q ? (const double *)(d + o_rQ) : nullptr;
  P.res_clone = tr->res_Q ? (const int *)(d + o_rc) : nullptr;
  P.feat_rep = st->feat_rep;
  P.col_ext = col_of(st->extrinsic_state_id);
  P.col_int = col_of(st->intrinsic_state_id);
  P.col_dt = col_of(st->dt_state_id);
  P.n_feat = F;
  P.n_obs = nobs;
  P.obs_ptr = (const int *)(d + o_ptr);
  P.obs_feat = (const int *)(d + o_of);
  P.obs_time = (const double *)(d + o_ot);
  P.obs_uv = (const float *)(d + o_uv);
  P.p_FinG = (const double *)(d + o_pg);
  P.p_FinG_fej = (const double *)(d + o_pgf);
  P.res_R = tr->res_R ? (const double *)(d + o_rR) : nullptr;
  P.res_p = tr->res_R ? (const double *)(d + o_rp) : nullptr;
  P.k = k;
  P.ld = ld;
  P.cols_in = (const int *)(d + o_cols);
  P.cols_out = nullptr;
  P.in_base = d;
  // (what every workgroup touches first thing: the whole block — or, of a speculative batch, whose block holds every candidate's
  //  observations, the state and the ranges in front of them)
  P.in_bytes = (ex && ex->spec_li) ? (int)o_of : (int)total;
  if (ex) {
    ex->d_uvn = ex->uvn ? (const float *)(d + o_xuvn) : nullptr;
    ex->d_flags = ex->flags ? (const uint8_t *)(d + o_xfl) : nullptr;
    if (ex->spec_li) {
      ex->d_spec_li = (const int *)(d + o_sli), ex->d_spec_meta = (const uint8_t *)(d + o_sme), ex->d_spec_prevalid = (const uint8_t *)(d + o_spv);
      ex->d_obs_end = (int *)(us->jin.as<char>() + o_send), ex->d_obs_uv = (float *)(us->jin.as<char>() + o_uv);
      P.obs_end = ex->d_obs_end;
    }
  }
  return PLV_OK;
}

// builds the batch into us->bHf ([Hf | Hx | res]) and us->brows on the device
// the column map as the host staged it: same offset in the pinned block as in the device copy (or the pinned block itself)
static const int *host_copy_of(plv_ctx_update_state *us, const int *cols_in, bool lines = false) {
  plv::PinBuf &hb = lines ? us->h_jin_l : us->h_jin;
  plv::DevBuf &db = lines ? us->jin_l : us->jin;
  const char *c = (const char *)cols_in, *hj = hb.as<char>();
  if (c >= hj && c < hj + hb.cap) return cols_in;
  return (const int *)(hj + (c - db.as<char>()));
}
struct FusedTri {  // triangulate on the device first and let the Jacobian launch take its candidates from the result
  const plv_tri_options *opt;
  const float *uvn;
  const uint8_t *flags;
  int max_sel;
  size_t o_p, o_err, o_ok;  // out: where the results sit in us->tri (p [F][3], err [F], ok [F], contiguous)
  const plv_points_spec *spec = nullptr;  // speculative submission: the candidates' membership is decided on the device (spec_select_kernel)
  size_t o_member = 0, o_words = 0;       // out (spec): member [F] and the words (SpecSelectArgs) behind ok, part of the mirrored result block
  const int *d_words = nullptr;           // out (spec): the words on the device
};
int build_on_device(plv_ctx *ctx, plv_ctx_update_state *us, const plv_state_view *st, const plv_tracks *tr, int k,
                    const int *col_to_state, int ld, bool project, FusedTri *ft = nullptr) {
  TRY(check_views(st, tr));
  if (k < 1 || ld < 2 || !col_to_state) return PLV_E_BADARG;
  const int F = tr->n_feat;
  const size_t nHf = (size_t)F * 3 * ld, nHx = (size_t)F * k * ld;
  TRY(us->bHf.reserve_units((size_t)F, (size_t)std::max(ctx->cfg.num_features, 64), (size_t)(3 + k + 1) * ld * 8));
  TRY(us->brows.reserve((size_t)F * 4));
  TRY(us->bcols.reserve((size_t)k * 4));
  {
    plv::HostPhase ph("build: prior prefetch, phase 0");
    if (project && ctx->cov_n > 0) TRY(plv_prior_prefetch(ctx, 0, nullptr, k, F, ld - 3));  // (before the upload goes onto the stream)
  }
  plv::HostPhase ph_stage("build: inputs staged + upload enqueued");
  JacParams P{};
  bool fuse_tri = false;
  int tri_max_obs = 1;
  double *tri_poses = nullptr, *tri_p = nullptr, *tri_err = nullptr;
  unsigned char *tri_valid = nullptr, *tri_ok = nullptr;
  const float *tri_uvn = nullptr;
  const plv_tri_options *tri_opt = nullptr;
  for (int f = 0; f < F; ++f) tri_max_obs = std::max(tri_max_obs, tr->obs_ptr[f + 1] - tr->obs_ptr[f]);  // (sizes the launch's LDS)
  if (ft) {
    StageExtra ex;
    ex.uvn = ft->uvn;
    ex.flags = ft->flags;
    if (ft->spec) ex.spec_li = ft->spec->li, ex.spec_meta = ft->spec->meta, ex.spec_prevalid = ft->spec->prevalid;
    TRY(stage_inputs(ctx, us, st, tr, k, col_to_state, ld, P, &ex));
    const int nobs = tr->obs_ptr[F];
    const size_t o_pose = 0, o_valid = (size_t)nobs * 96, o_p = (o_valid + nobs + 15) & ~(size_t)15, o_err = o_p + (size_t)F * 24,
                 o_ok = o_err + (size_t)F * 8, o_member = o_ok + F, o_words = (o_member + F + 7) & ~(size_t)7, o_order = o_words + 32,
                 total = o_order + 4 * (size_t)spec_grid(ft->max_sel) + 16;
    TRY(us->tri.reserve(total));
    char *d = us->tri.as<char>();
    ft->o_member = o_member, ft->o_words = o_words;
    // one launch for triangulation + Jacobians + null space while the selection has no cap to enforce (see the kernel); a speculative
    // batch holds more candidates than the cap, but its pool does not (spec_select_kernel empties every candidate otherwise)
    fuse_tri = project && (F <= ft->max_sel || ft->spec) && !plv::knob(plv::PLV_KNOB_POINT_TRI_SEPARATE);
    if (ft->spec) {
      if (!fuse_tri || plv::knob(plv::PLV_KNOB_INPUTS_PINNED)) {
        set_last_error("speculative point submission needs the fused triangulation launch");
        return PLV_E_BADARG;
      }
      SpecSelectArgs A{};
      A.F = F, A.n_flow = ft->spec->n_flow, A.W = ctx->cfg.width, A.H = ctx->cfg.height, A.max_sel = ft->max_sel, A.grid = spec_grid(ft->max_sel);
      A.obs_ptr = P.obs_ptr, A.li = ex.d_spec_li, A.meta = ex.d_spec_meta, A.prevalid = ex.d_spec_prevalid;
      A.flow_p1 = ft->spec->d_flow_p1, A.flow_n1 = ft->spec->d_flow_n1, A.flow_mask = ft->spec->d_flow_mask;
      A.obs_uv = ex.d_obs_uv, A.obs_uvn = const_cast<float *>(ex.d_uvn), A.obs_end = ex.d_obs_end;
      A.sel_flags = const_cast<unsigned char *>(ex.d_flags), A.member = (unsigned char *)(d + o_member), A.words = (int *)(d + o_words);
      if (!ctx->gate_stage.on) {
        set_last_error("speculative point submission needs the gate inside the Jacobian launch");
        return PLV_E_BADARG;
      }
      A.order = (int *)(d + o_order), A.rows_out = us->brows.as<int>();
      A.tri_p = (double *)(d + o_p), A.tri_err = (double *)(d + o_err), A.tri_ok = (unsigned char *)(d + o_ok);
      A.chi2 = ctx->gate_stage.chi2, A.accepted = ctx->gate_stage.accepted, A.acc_rows = ctx->gate_stage.acc_rows;
      A.zero_word = ctx->gate_stage.n_acc_next, A.cols_out = us->bcols.as<int>(), A.cols_in = P.cols_in, A.k = k;
      TRY(launch_spec_select(ctx, A));
      P.spec_order = A.order, P.spec_count = A.words + 2, P.spec_pass = A.words + 4;
      ft->d_words = A.words;
    }
    tri_poses = (double *)(d + o_pose), tri_valid = (unsigned char *)(d + o_valid), tri_uvn = ex.d_uvn;
    tri_p = (double *)(d + o_p), tri_ok = (unsigned char *)(d + o_ok), tri_err = (double *)(d + o_err);
    tri_opt = ft->opt;
    if (ctx->decision_trace) {
      TRY(ctx->d_tri_dbg.reserve((size_t)F * 32));
      P.tri_dbg = ctx->d_tri_dbg.as<double>();
      ctx->dec_F = F;
    }
    if (!fuse_tri) TRY(launch_triangulate(ctx, P, tri_poses, tri_valid, tri_uvn, *ft->opt, tri_p, tri_ok, tri_err, tri_max_obs));
    P.p_FinG = P.p_FinG_fej = (const double *)(d + o_p);  // MSCKF features: FEJ value = estimate (REF CamHelper.cpp:556-557)
    P.sel_flags = ex.d_flags;
    P.tri_ok = (const unsigned char *)(d + o_ok);
    P.tri_err = (const double *)(d + o_err);
    P.max_sel = ft->max_sel;
    ft->o_p = o_p, ft->o_err = o_err, ft->o_ok = o_ok;

  } else {
    TRY(stage_inputs(ctx, us, st, tr, k, col_to_state, ld, P));
  }
  ph_stage.stop();
  P.cols_out = us->bcols.as<int>();
  P.rows = us->brows.as<int>();
  P.Hf = us->bHf.as<double>();
  P.Hx = P.Hf + nHf;
  P.res = P.Hx + nHx;
  us->b_projected = false;
  us->b_gather_token = 0;
  if (project) {
    // resident update path: build + project in one launch; when a covariance of matching size is resident its gathers ride along
    bool can_gather = ctx->cov_n > 0;
    for (int j = 0; j < k && can_gather; ++j) can_gather = col_to_state[j] >= 0 && col_to_state[j] < ctx->cov_n;
    GatherArgs g{};
    int gblocks = 0;
    plv::HostPhase ph_l("build: gather arguments + Jacobian launch");
    if (can_gather) {
      const int n = ctx->cov_n;
      TRY(gather_args(ctx, ctx->d_P.as<double>(), n, n, P.cols_in, k, g));
      gblocks = (std::max(k * n, std::max(k * k, n)) + 255) / 256;
    }
    if (fuse_tri)
      TRY(launch_jacobians_projected(ctx, P, can_gather ? &g : nullptr, gblocks, tri_opt, tri_poses, tri_valid, tri_uvn, tri_p, tri_ok, tri_err, tri_max_obs));
    else
      TRY(launch_jacobians_projected(ctx, P, can_gather ? &g : nullptr, gblocks, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tri_max_obs));
    ph_l.stop();
    plv::HostPhase ph_p1("build: prior prefetch, phase 1 (side stream)");
    if (can_gather)  // (the column map as the host staged it: same offset in the pinned block as in the device copy)
      TRY(plv_prior_prefetch(ctx, 1, host_copy_of(us, P.cols_in), k, F, ld - 3));
    ph_p1.stop();
    us->b_projected = true;
    us->b_gather_token = can_gather ? ctx->gather_stamp : 0;
  } else {
    TRY(launch_jacobians(ctx, P));
  }
  us->bF = F;
  us->bfdim = 3;
  us->bk = k;
  us->bld = ld;
  us->bmaxrows = ld;
  us->b_on_device_rows = true;
  return PLV_OK;
}

}  // namespace

extern "C" {

int plv_jacobian_columns(const plv_state_view *st, const plv_tracks *tr, int *col_to_state, int cap, int *k_out) {
  if (!col_to_state || !k_out) return PLV_E_BADARG;
  TRY(check_views(st, tr));
  int k = 0;
  auto push = [&](int id, int size) {
    if (id < 0) return true;
    for (int j = 0; j < k; ++j)
      if (col_to_state[j] == id) return true;
    if (k + size > cap) return false;
    for (int d = 0; d < size; ++d) col_to_state[k++] = id + d;
    return true;
  };
  // REF: CamHelper.cpp:74-95 calibration blocks first, then :98-110 interpolation poses in first-seen order
  if (!push(st->extrinsic_state_id, 6) || !push(st->intrinsic_state_id, 8) || !push(st->dt_state_id, 1)) return PLV_E_CAPACITY;
  // (a window start that has been seen adds nothing: 700 observations meet ~14 distinct windows, and the search through the
  // column list per pose was 29 us of the caller's thread in front of the point launch at configs[2])
  std::vector<uint8_t> seen_s0((size_t)std::max(st->n_clones, 1), 0);
  BoundingStartMemo start_of(*st);
  for (int f = 0; f < tr->n_feat; ++f)
    for (int o = tr->obs_ptr[f]; o < tr->obs_ptr[f + 1]; ++o) {
      const int s0 = start_of(tr->obs_time[o] + st->cam_dt);
      if (s0 < 0 || seen_s0[s0]) continue;
      seen_s0[s0] = 1;
      for (int w = 0; w < 4; ++w)
        if (!push(st->clone_state_id[s0 + w], 6)) return PLV_E_CAPACITY;
    }
  *k_out = k;
  return PLV_OK;
}

// One-submission point update (plv_camera_update_points): triangulation of every pool candidate, the selection loop, Jacobians +
// null-space projection, gate, compression and EKFUpdate are enqueued back to back; one upload, one result download, one host
// synchronisation.  `all` carries obs_uvn; flags[f] = the host's part of the selection test.  Returns as plv_msckf_update_resident
// (PLV_E_NOT_PSD: covariance untouched); p / ok / err (per candidate) and accepted (per candidate, 0 for unselected ones) are filled
// in both cases.
typedef plv_ctx_update_state::PointJob PointJob;
static PointJob &point_job(plv_ctx *ctx) { return plv_update_state(ctx)->point_job; }
static hipEvent_t g_ce[3] = {nullptr, nullptr, nullptr};

int plv_points_update_submit(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *all, const plv_tri_options *tri, const uint8_t *flags,
                             int max_sel, int k, const int *col_to_state, int ld, double sigma2, double chi2_mult, double res_norm_gate,
                             const plv_points_spec *spec) {
  if (!ctx || !all || !tri || !flags || !all->obs_uvn) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  PointJob &J = point_job(ctx);
  J = PointJob();
  std::vector<double> p_dummy(3 * (size_t)std::max(all->n_feat, 1), 0.0);
  plv_tracks t2 = *all;
  t2.p_FinG = t2.p_FinG_fej = p_dummy.data();  // (outputs of the triangulation: the staged copy is never read)
  FusedTri ft{tri, all->obs_uvn, flags, max_sel, 0, 0, 0};
  ft.spec = spec;
  // PLV_KNOB_CHAIN_EVENTS (with PLV_KNOB_HOST_TIMING): three timed events on the stream — at entry (the stream is idle: stamped at once),
  // behind the Jacobian launch, behind the update's last kernel — to set the device's view of the chain against the host's phases
  J.chain_events = plv::knob(plv::PLV_KNOB_CHAIN_EVENTS) && plv::host_phases().on;
  J.t_entry = std::chrono::steady_clock::now();
  if (J.chain_events) {
    for (auto &e : g_ce)
      if (!e) (void)hipEventCreate(&e);
    (void)hipEventRecord(g_ce[0], ctx->stream);
  }
  plv::HostPhase ph_a("points fused: stage + triangulate + jacobians enqueued");
  TRY(plv_update_gate_prepare(ctx, all->n_feat, 3, k, ld, sigma2, chi2_mult, res_norm_gate, 0));
  TRY(build_on_device(ctx, us, st, &t2, k, col_to_state, ld, true, &ft));
  ph_a.stop();
  plv::frame_mark("@ point Jacobian launch enqueued");
  if (J.chain_events) (void)hipEventRecord(g_ce[1], ctx->stream);
  us->b_single_use = true;
  const int F = all->n_feat;
  const size_t mirror_bytes = spec ? (ft.o_words + 20 - ft.o_p) : (size_t)F * 33;
  TRY(us->h_tri.reserve(mirror_bytes + 16));
  plv::HostPhase ph_b("points fused: gate .. EKF enqueued");
  // the triangulation results reach the host with the update's result block (copied by its last kernel), or by a copy command when
  // the chain ended another way; either lands before the wait below returns
  ctx->mirror2_src = us->tri.as<char>() + ft.o_p, ctx->mirror2_dst = us->h_tri.p, ctx->mirror2_bytes = mirror_bytes, ctx->mirror2_taken = false;
  // (for a line launch chained behind this update, plv_camera_try_update: the commit kernel leaves "state changed" in a device word)
  TRY(us->chain_words.reserve(64));
  us->applied_word = us->chain_words.as<int>();
  ctx->applied_word = us->applied_word, ctx->applied_used = false;
  ctx->cap_words = spec ? ft.d_words + 3 : nullptr, ctx->cap = max_sel;
  int rc = plv_msckf_update_resident_launch(ctx, sigma2, chi2_mult, res_norm_gate);
  ctx->cap_words = nullptr;
  us->applied_armed = rc == PLV_OK && ctx->applied_used;
  ctx->applied_word = nullptr, ctx->applied_used = false;
  us->pt_tri_p = (const double *)(us->tri.as<char>() + ft.o_p), us->pt_tri_ok = (const unsigned char *)(us->tri.as<char>() + ft.o_ok), us->pt_tri_F = F;
  const bool mirrored = ctx->mirror2_taken;
  ctx->mirror2_src = nullptr, ctx->mirror2_dst = nullptr, ctx->mirror2_bytes = 0, ctx->mirror2_taken = false;
  if (!mirrored)
    PLV_HIP_CHECK(plv::memcpy_async(us->h_tri.p, us->tri.as<char>() + ft.o_p, mirror_bytes, hipMemcpyDeviceToHost, ctx->stream));
  ph_b.stop();
  plv::frame_mark("@ point chain enqueued");
  if (J.chain_events) (void)hipEventRecord(g_ce[2], ctx->stream);
  J.pending = true, J.mirrored = mirrored, J.rc = rc, J.F = F, J.spec = spec != nullptr, J.max_sel = max_sel;
  J.o_p = ft.o_p, J.o_member = ft.o_member, J.o_words = ft.o_words;
  return PLV_OK;
}

int plv_points_update_collect(plv_ctx *ctx, double *p_out, uint8_t *ok_out, double *err_out, uint8_t *accepted, int *n_rows, double *dx,
                              void (*before_wait)(void *), void *before_wait_arg, uint8_t *member, int *spec_count, int *spec_over) {
  if (!ctx || !p_out || !ok_out || !err_out || !accepted || !dx) return PLV_E_BADARG;
  auto *us = plv_update_state(ctx);
  PointJob &J = point_job(ctx);
  if (!J.pending) {
    set_last_error("plv_points_update_collect: no point update was submitted");
    return PLV_E_BADARG;
  }
  J.pending = false;
  int rc = J.rc;
  const int F = J.F;
  plv::HostPhase ph_c("points fused: host work inside the wait");
  if (before_wait) before_wait(before_wait_arg);  // host work of the caller that fits into the wait
  // ... and work that becomes possible DURING the wait (the line pool, once the line worker has finished the frame's feed): the
  // caller's poll function is tried until it reports that nothing is left, or the update is done
  if (rc == PLV_OK && ctx->wait_poll && us->done_ev) {
    auto running = [&]() {
      if (us->word_seq) return __atomic_load_n((const unsigned *)ctx->done_word(16), __ATOMIC_ACQUIRE) != us->word_seq;
      return hipEventQuery(us->done_ev) == hipErrorNotReady;
    };
    while (running() || plv::knob(plv::PLV_KNOB_CHAIN_ALWAYS)) {
      if (ctx->wait_poll(ctx->wait_poll_arg)) break;
      for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
    }
  }
  ph_c.stop();
  plv::frame_mark("@ host work inside the point wait done");
  plv::NsScope ns_pw(plv::counters().points_wait_ns);
  plv::HostPhase ph_d("points fused: wait");
  if (rc == PLV_OK) rc = plv_msckf_update_resident_wait(ctx, accepted, n_rows, dx);  // (ends at the update's last kernel)
  if (rc != PLV_OK || !J.mirrored) PLV_HIP_CHECK(plv::stream_sync(ctx->stream));    // (the copy command enqueued behind it)
  ph_d.stop();
  plv::frame_mark("@ point update collected");
  if (J.chain_events) {
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - J.t_entry).count();
    (void)hipEventSynchronize(g_ce[2]);
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, g_ce[0], g_ce[1]) == hipSuccess && hipEventElapsedTime(&b, g_ce[1], g_ce[2]) == hipSuccess) {
      plv::host_phases().add("points fused: DEVICE entry -> Jacobian launch done", a * 1e3);
      plv::host_phases().add("points fused: DEVICE Jacobian done -> last kernel done", b * 1e3);
      plv::host_phases().add("points fused: HOST entry -> results read", host_us);
    }
  }
  const char *h = us->h_tri.as<char>();
  memcpy(p_out, h, (size_t)F * 24);
  memcpy(err_out, h + (size_t)F * 24, (size_t)F * 8);
  memcpy(ok_out, h + (size_t)F * 32, (size_t)F);
  if (J.spec) {
    if (member) memcpy(member, h + (J.o_member - J.o_p), (size_t)F);
    const int *w = (const int *)(h + (J.o_words - J.o_p));
    if (spec_count) *spec_count = w[0];
    if (spec_over) *spec_over = w[1] ? 1 : ((w[3] && w[4] >= J.max_sel) ? 2 : 0);  // (2: worked on, but the selection loop's cap would have cut the pool)
  } else {
    if (spec_count) *spec_count = 0;
    if (spec_over) *spec_over = 0;
  }
  return rc;
}

int plv_points_update_fused(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *all, const plv_tri_options *tri,
                            const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2, double chi2_mult,
                            double res_norm_gate, double *p_out, uint8_t *ok_out, double *err_out, uint8_t *accepted, int *n_rows,
                            double *dx, void (*before_wait)(void *), void *before_wait_arg) {
  if (!ctx || !all || !tri || !flags || !p_out || !ok_out || !err_out || !accepted || !dx || !all->obs_uvn) return PLV_E_BADARG;
  TRY(plv_points_update_submit(ctx, st, all, tri, flags, max_sel, k, col_to_state, ld, sigma2, chi2_mult, res_norm_gate, nullptr));
  return plv_points_update_collect(ctx, p_out, ok_out, err_out, accepted, n_rows, dx, before_wait, before_wait_arg, nullptr, nullptr, nullptr);
}

int plv_build_jacobians_resident(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *tr, int k,
                                 const int *col_to_state, int ld) {
  if (!ctx) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  TRY(build_on_device(ctx, us, st, tr, k, col_to_state, ld, true));
  us->b_single_use = true;  // consumed in place by the next plv_msckf_update_resident
  return PLV_OK;            // stream-ordered; no host synchronisation
}

int plv_build_jacobians(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *tr, int k, const int *col_to_state,
                        int ld, int *rows, double *Hf, double *Hx, double *res) {
  if (!ctx || !rows || !Hf || !Hx || !res) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  TRY(build_on_device(ctx, us, st, tr, k, col_to_state, ld, false));  // host out: the un-projected systems
  us->b_single_use = false;
  const int F = tr->n_feat;
  const size_t nHf = (size_t)F * 3 * ld, nHx = (size_t)F * k * ld, nr = (size_t)F * ld;
  const double *d = us->bHf.as<double>();
  PLV_HIP_CHECK(plv::memcpy_async(Hf, d, nHf * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(Hx, d + nHf, nHx * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(res, d + nHf + nHx, nr * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(rows, us->brows.p, (size_t)F * 4, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  return PLV_OK;
}


int plv_triangulate(plv_ctx *ctx, const plv_state_view *st, const plv_tracks *tr, const plv_tri_options *opt, double *p_FinG,
                    uint8_t *ok, double *reproj_err) {
  if (!ctx || !opt || !p_FinG || !ok) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  // p_FinG of the view is an OUTPUT here: allow it to be missing by pointing the checks at the output
  plv_tracks t2 = *tr;
  if (!t2.p_FinG) t2.p_FinG = p_FinG;
  if (!t2.p_FinG_fej) t2.p_FinG_fej = t2.p_FinG;
  TRY(check_views(st, &t2));
  if (!tr->obs_uvn) {
    set_last_error("plv_triangulate: plv_tracks.obs_uvn is required");
    return PLV_E_BADARG;
  }
  const int F = tr->n_feat, nobs = tr->obs_ptr[F];
  int dummy_cols[1] = {-1};
  JacParams P{};
  TRY(stage_inputs(ctx, us, st, &t2, 0, dummy_cols, 2, P));
  const size_t o_pose = 0, o_valid = (size_t)nobs * 96, o_uvn = (o_valid + nobs + 15) & ~(size_t)15, o_p = o_uvn + (size_t)nobs * 8,
               o_err = o_p + (size_t)F * 24, o_ok = o_err + (size_t)F * 8, total = o_ok + F + 16;
  TRY(us->tri.reserve(total));
  char *d = us->tri.as<char>();
  PLV_HIP_CHECK(plv::memcpy_async(d + o_uvn, tr->obs_uvn, (size_t)nobs * 8, hipMemcpyHostToDevice, ctx->stream));
  int max_obs = 1;
  for (int f = 0; f < F; ++f) max_obs = std::max(max_obs, tr->obs_ptr[f + 1] - tr->obs_ptr[f]);
  TRY(launch_triangulate(ctx, P, (double *)(d + o_pose), (unsigned char *)(d + o_valid), (const float *)(d + o_uvn), *opt,
                         (double *)(d + o_p), (unsigned char *)(d + o_ok), (double *)(d + o_err), max_obs));
  PLV_HIP_CHECK(plv::memcpy_async(p_FinG, d + o_p, (size_t)F * 24, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(ok, d + o_ok, (size_t)F, hipMemcpyDeviceToHost, ctx->stream));
  if (reproj_err) PLV_HIP_CHECK(plv::memcpy_async(reproj_err, d + o_err, (size_t)F * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  return PLV_OK;
}

}  // extern "C"

namespace {

// ------------------------------------------------------------------------------------------ lines
int check_line_views(const plv_state_view *st, const plv_line_tracks *lt, bool need_lines, bool need_uvn) {
  if (!st || !lt || st->n_clones < 1 || !st->clone_time || !st->clone_R || !st->clone_p || !st->clone_R_fej ||
      !st->clone_p_fej || !st->clone_state_id || lt->n_lines < 1 || !lt->obs_ptr || !lt->obs_time || !lt->seg_uv ||
      (need_lines && !lt->line_FinG) || (need_uvn && !lt->seg_uvn)) {
    set_last_error("line jacobians: null view field");
    return PLV_E_BADARG;
  }
  if (st->intr_order != 3) {
    set_last_error("line jacobians: only intr_order = 3 is built (got %d)", st->intr_order);
    return PLV_E_BADARG;
  }
  if ((lt->res_R == nullptr) != (lt->res_p == nullptr)) return PLV_E_BADARG;
  if (lt->has_pt && !lt->anchor_pt) return PLV_E_BADARG;
  return PLV_OK;
}

// one packed upload of the state view and the line tracks; fills JacParams (n_feat = n_lines)
// st_tri (optional): a second state whose clone poses ride in the same block; *Pt then is P with the poses, extrinsics and time
// offset of that state (the view the line triangulation works on when it differs from the one the Jacobians are taken at)
int stage_line_inputs(plv_ctx *ctx, plv_ctx_update_state *us, const plv_state_view *st, const plv_line_tracks *lt, int k,
                      const int *col_to_state, int ld, JacParams &P, StageExtra *ex = nullptr, const plv_state_view *st_tri = nullptr,
                      JacParams *Pt = nullptr) {
  const int N = st->n_clones, L = lt->n_lines, nobs = lt->obs_ptr[L];
  if (nobs < 1) {
    set_last_error("line jacobians: no observations");
    return PLV_E_BADARG;
  }
  auto col_of = [&](int sid) {
    if (sid < 0) return -1;
    for (int j = 0; j < k; ++j)
      if (col_to_state[j] == sid) return j;
    return -1;
  };
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_time = take(8 * N), o_R = take(72 * N), o_p = take(24 * N), o_Rf = take(72 * N), o_pf = take(24 * N),
               o_ccol = take(4 * N), o_ptr = take(4 * (L + 1)), o_ot = take(8 * nobs), o_uv = take(16 * nobs),
               o_uvn = lt->seg_uvn ? take(16 * nobs) : 0, o_lg = lt->line_FinG ? take(48 * L) : 0,
               o_D = lt->D ? take(4 * L) : 0, o_ap = lt->has_pt ? take(24 * L) : 0, o_hp = lt->has_pt ? take(L) : 0,
               o_rR = lt->res_R ? take(72 * nobs) : 0, o_rp = lt->res_R ? take(24 * nobs) : 0,
               o_rQ = lt->res_Q ? take(288 * (size_t)nobs) : 0, o_rc = lt->res_Q ? take(4 * (size_t)nobs) : 0,
               o_cols = take(4 * (size_t)k), o_xfl = ex && ex->flags ? take(L) : 0;
  const bool two_states = st_tri && Pt && st_tri != st && st_tri->n_clones == N;
  const size_t o_tR = two_states ? take(72 * N) : 0, o_tp = two_states ? take(24 * N) : 0;
  // chained launch (plv_ctx::chain): quaternions + covariance indices for the kernel's own x (+) dx, anchor candidates
  const plv_ctx::ChainState &ch = ctx->chain;
  const bool chained = ch.on && ch.ready && !two_states && (int)ch.q.size() == 4 * N && (int)ch.ids.size() == N + 3 && (int)ch.anc_ptr.size() == L + 1;
  const size_t n_anc = chained ? ch.anc_f.size() : 0;
  const size_t o_cq = chained ? take(32 * (size_t)N) : 0, o_cid = chained ? take(4 * (size_t)(N + 3)) : 0, o_aptr = chained ? take(4 * (size_t)(L + 1)) : 0,
               o_af = chained ? take(4 * n_anc + 4) : 0, o_aho = chained ? take(n_anc + 4) : 0, o_aold = chained ? take(24 * n_anc + 8) : 0;
  if (ch.on && !chained) {
    set_last_error("line jacobians: chained launch asked for without its state");
    return PLV_E_BADARG;
  }
  const size_t total = off;
  TRY(us->h_jin_l.reserve(total));
  TRY(us->jin_l.reserve(total));
  char *h = us->h_jin_l.as<char>();
  memcpy(h + o_cols, col_to_state, 4 * (size_t)k);
  if (ex && ex->flags) memcpy(h + o_xfl, ex->flags, L);
  if (two_states) {
    memcpy(h + o_tR, st_tri->clone_R, 72 * N);
    memcpy(h + o_tp, st_tri->clone_p, 24 * N);
  }
  if (chained) {
    memcpy(h + o_cq, ch.q.data(), 32 * (size_t)N);
    memcpy(h + o_cid, ch.ids.data(), 4 * (size_t)(N + 3));
    memcpy(h + o_aptr, ch.anc_ptr.data(), 4 * (size_t)(L + 1));
    if (n_anc) {
      memcpy(h + o_af, ch.anc_f.data(), 4 * n_anc);
      memcpy(h + o_aho, ch.anc_has_old.data(), n_anc);
      memcpy(h + o_aold, ch.anc_old.data(), 24 * n_anc);
    }
  }
  memcpy(h + o_time, st->clone_time, 8 * N);
  memcpy(h + o_R, st->clone_R, 72 * N);
  memcpy(h + o_p, st->clone_p, 24 * N);
  memcpy(h + o_Rf, st->clone_R_fej, 72 * N);
  memcpy(h + o_pf, st->clone_p_fej, 24 * N);
  int *ccol = (int *)(h + o_ccol);
  for (int i = 0; i < N; ++i) ccol[i] = col_of(st->clone_state_id[i]);
  for (int l = 0; l < L; ++l)
    if (lt->obs_ptr[l + 1] < lt->obs_ptr[l]) return PLV_E_BADARG;
  memcpy(h + o_ptr, lt->obs_ptr, 4 * (L + 1));
  memcpy(h + o_ot, lt->obs_time, 8 * nobs);
  memcpy(h + o_uv, lt->seg_uv, 16 * nobs);
  if (lt->seg_uvn) memcpy(h + o_uvn, lt->seg_uvn, 16 * nobs);
  if (lt->line_FinG) memcpy(h + o_lg, lt->line_FinG, 48 * L);
  if (lt->D) memcpy(h + o_D, lt->D, 4 * L);
  if (lt->has_pt) {
    memcpy(h + o_ap, lt->anchor_pt, 24 * L);
    memcpy(h + o_hp, lt->has_pt, L);
  }
  if (lt->res_R) {
    memcpy(h + o_rR, lt->res_R, 72 * nobs);
    memcpy(h + o_rp, lt->res_p, 24 * nobs);
  }
  if (lt->res_Q) {
    if (!lt->res_clone) return PLV_E_BADARG;
    memcpy(h + o_rQ, lt->res_Q, 288 * (size_t)nobs);
    memcpy(h + o_rc, lt->res_clone, 4 * (size_t)nobs);
  }
  // measurement knob PLV_KNOB_INPUTS_PINNED: no upload — the kernels read the pinned staging block over PCIe (every byte once or a
  // few times; what a workgroup reuses it keeps in LDS)
  const bool pinned_inputs = plv::knob(plv::PLV_KNOB_INPUTS_PINNED) && !(ex && ex->spec_li);  // (the speculative batch is patched on the device)
  // (an upload by a kernel of the ctx stream instead of the copy command was measured, alternating frame by frame: no difference)
  if (!pinned_inputs) PLV_HIP_CHECK(plv::memcpy_async(us->jin_l.p, h, total, hipMemcpyHostToDevice, ctx->stream));
  const char *d = pinned_inputs ? (const char *)h : us->jin_l.as<char>();
  P.n_clones = N;
  P.clone_time = (const double *)(d + o_time);
  P.clone_R = (const double *)(d + o_R);
  P.clone_p = (const double *)(d + o_p);
  P.clone_R_fej = (const double *)(d + o_Rf);
  P.clone_p_fej = (const double *)(d + o_pf);
  P.clone_col = (const int *)(d + o_ccol);
  memcpy(P.R_ItoC, st->R_ItoC, 72);
  memcpy(P.p_IinC, st->p_IinC, 24);
  memcpy(P.K, st->intrinsics, 64);
  P.cam_dt = st->cam_dt;
  P.dt_exp = st->dt_exp;
  P.sigma_pix = st->sigma_pix;
  P.intr_ori_cov = st->intr_ori_cov;
  P.intr_pos_cov = st->intr_pos_cov;
  P.use_pol_cov = st->use_pol_cov;
  P.use_imu_cov = st->use_imu_cov && lt->res_Q ? 1 : 0;
  P.intr_err_mlt = st->intr_err_mlt;
  P.res_Q = lt->res_Q ? (const double *)(d + o_rQ) : nullptr;
  P.res_clone = lt->res_Q ? (const int *)(d + o_rc) : nullptr;
  P.feat_rep = st->feat_rep;
  P.col_ext = P.col_int = -1;
  P.col_dt = col_of(st->dt_state_id);
  P.n_feat = L;
  P.n_obs = nobs;
  P.obs_ptr = (const int *)(d + o_ptr);
  P.obs_time = (const double *)(d + o_ot);
  P.seg_uv = (const float *)(d + o_uv);
  P.seg_uvn = lt->seg_uvn ? (const float *)(d + o_uvn) : nullptr;
  P.line_FinG = lt->line_FinG ? (const double *)(d + o_lg) : nullptr;
  P.lineD = lt->D ? (const int *)(d + o_D) : nullptr;
  P.anchor_pt = lt->has_pt ? (const double *)(d + o_ap) : nullptr;
  P.has_pt = lt->has_pt ? (const unsigned char *)(d + o_hp) : nullptr;
  P.res_R = lt->res_R ? (const double *)(d + o_rR) : nullptr;
  P.res_p = lt->res_R ? (const double *)(d + o_rp) : nullptr;
  P.k = k;
  P.ld = ld;
  P.cols_in = (const int *)(d + o_cols);
  P.in_base = d;
  P.in_bytes = (int)total;
  P.cols_out = nullptr;
  if (chained) {
    auto *us2 = us;
    P.chain_dx = us2->result.as<double>();  // (dx leads the status block of the update launched last: the point update)
    P.chain_applied = us2->applied_word;
    P.chain_status = (const int *)(us2->result.as<char>() + (size_t)ctx->cov_n * 8);
    P.chain_q = (const double *)(d + o_cq);
    P.chain_id = (const int *)(d + o_cid);
    memcpy(P.chain_qe, ch.qe, 32);
    P.anc_ptr = (const int *)(d + o_aptr);
    P.anc_f = (const int *)(d + o_af);
    P.anc_has_old = (const unsigned char *)(d + o_aho);
    P.anc_old = (const double *)(d + o_aold);
    P.anc_tri_p = us2->pt_tri_p;
    P.anc_tri_ok = us2->pt_tri_ok;
  }
  if (ex) ex->d_flags = ex->flags ? (const uint8_t *)(d + o_xfl) : nullptr;
  if (Pt) {
    *Pt = P;
    if (two_states) {
      Pt->clone_R = (const double *)(d + o_tR);
      Pt->clone_p = (const double *)(d + o_tp);
      memcpy(Pt->R_ItoC, st_tri->R_ItoC, 72);
      memcpy(Pt->p_IinC, st_tri->p_IinC, 24);
      Pt->cam_dt = st_tri->cam_dt;
    }
  }
  return PLV_OK;
}

struct FusedLineTri {
  const uint8_t *flags;
  int max_sel;
  size_t o_lines, o_ok;  // out: results in us->tri (lines [L][6], ok [L])
  const plv_state_view *st_tri = nullptr;  // the state of the triangulation when it is not the one of the Jacobians
};
int build_lines_on_device(plv_ctx *ctx, plv_ctx_update_state *us, const plv_state_view *st, const plv_line_tracks *lt, int k,
                          const int *col_to_state, int ld, bool project, FusedLineTri *ft = nullptr) {
  TRY(check_line_views(st, lt, ft == nullptr, ft != nullptr));
  if (k < 1 || ld < 2 || !col_to_state) return PLV_E_BADARG;
  const int L = lt->n_lines;
  const size_t nHf = (size_t)L * 6 * ld, nHx = (size_t)L * k * ld;
  TRY(us->bHf.reserve_units((size_t)L, (size_t)std::max(ctx->cfg.num_features, 64), (size_t)(6 + k + 1) * ld * 8));
  TRY(us->brows.reserve((size_t)L * 4));
  TRY(us->bcols_l.reserve((size_t)k * 4));
  // The prior factor of the whitened update, ahead of time on the side stream — not for the one-submission line update (gate probe): it
  // accepts a line or two, fewer rows than columns, and then goes through EKFUpdate on the rows themselves (plv_api.hip, round 6); the
  // four enqueue calls were 8-10 us of the caller's thread in front of the line launch, the factor 50 us of side-stream work per frame
  // that nothing read.  (A probed update that does accept more rows than columns starts the factor when it knows.)
  const bool prior_ahead = project && ctx->cov_n > 0 && !(ctx->gate_stage.on && ctx->gate_stage.probe_dst);
  if (prior_ahead) TRY(plv_prior_prefetch(ctx, 0, nullptr, k, L, ld - 6));  // (before the upload goes onto the stream)
  JacParams P{}, Pt{};
  bool fuse_tri = false;
  double *tri_cam = nullptr, *tri_imu = nullptr, *tri_lines = nullptr;
  unsigned char *tri_valid = nullptr, *tri_ok = nullptr;
  StageExtra ex;
  if (ft) ex.flags = ft->flags;
  {
    plv::HostPhase ph_stage("build lines: inputs staged + upload enqueued");
    TRY(stage_line_inputs(ctx, us, st, lt, k, col_to_state, ld, P, &ex, ft ? ft->st_tri : nullptr, &Pt));
  }
  plv::HostPhase ph_rest("build lines: gather arguments + launch + prior prefetch");
  us->b_projected = false;
  us->b_gather_token = 0;
  if (ft) {
    const int nobs = lt->obs_ptr[L];
    const size_t o_cam = 0, o_imu = (size_t)nobs * 96, o_lines = o_imu + (size_t)nobs * 96, o_ok = o_lines + (size_t)L * 48,
                 o_valid = (o_ok + L + 15) & ~(size_t)15, total = o_valid + nobs + 16;
    TRY(us->tri_l.reserve(total));
    char *d = us->tri_l.as<char>();
    // one launch for triangulation + Jacobians + null space while the selection has no cap to enforce (see the kernel)
    fuse_tri = project && L <= ft->max_sel && !plv::knob(plv::PLV_KNOB_LINE_TRI_SEPARATE);
    tri_cam = (double *)(d + o_cam), tri_imu = (double *)(d + o_imu), tri_valid = (unsigned char *)(d + o_valid);
    tri_lines = (double *)(d + o_lines), tri_ok = (unsigned char *)(d + o_ok);
    if (!fuse_tri) TRY(launch_triangulate_lines(ctx, Pt, tri_cam, tri_imu, tri_valid, tri_lines, tri_ok));
    P.line_FinG = (const double *)(d + o_lines);
    P.sel_flags = ex.d_flags;
    P.tri_ok = (const unsigned char *)(d + o_ok);
    P.tri_err = nullptr;
    P.max_sel = ft->max_sel;
    ft->o_lines = o_lines, ft->o_ok = o_ok;
    if (ctx->gate_stage.on && ctx->gate_stage.probe_dst) {  // the gate inside the launch also carries the triangulated lines to the host
      ctx->gate_stage.probe_src = (const unsigned char *)(d + o_lines);
      ctx->gate_stage.probe_stride_a = 48, ctx->gate_stage.probe_off_b = L * 48, ctx->gate_stage.probe_stride_b = 1;
    }
  }
  P.rows = us->brows.as<int>();
  P.Hf = us->bHf.as<double>();
  P.Hx = P.Hf + nHf;
  P.res = P.Hx + nHx;
  if (project) {
    // resident update path: build + project in one launch (the column map is published by its workgroup 0); when a covariance of
    // matching size is resident its gathers ride along
    P.cols_out = us->bcols_l.as<int>();
    bool can_gather = ctx->cov_n > 0;
    for (int j = 0; j < k && can_gather; ++j) can_gather = col_to_state[j] >= 0 && col_to_state[j] < ctx->cov_n;
    GatherArgs g{};
    int gblocks = 0;
    if (can_gather) {
      const int n = ctx->cov_n;
      TRY(gather_args(ctx, ctx->d_P.as<double>(), n, n, P.cols_in, k, g));
      gblocks = (std::max(k * n, std::max(k * k, n)) + 255) / 256;
    }
    int line_max_obs = 1;  // (sizes the launch's LDS)
    for (int l = 0; l < L; ++l) line_max_obs = std::max(line_max_obs, lt->obs_ptr[l + 1] - lt->obs_ptr[l]);
    if (fuse_tri)
      TRY(launch_line_jacobians_projected(ctx, P, can_gather ? &g : nullptr, gblocks, &Pt, tri_cam, tri_imu, tri_valid, tri_lines, tri_ok, line_max_obs));
    else
      TRY(launch_line_jacobians_projected(ctx, P, can_gather ? &g : nullptr, gblocks, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, line_max_obs));
    if (can_gather && prior_ahead)
      TRY(plv_prior_prefetch(ctx, 1, host_copy_of(us, P.cols_in, true), k, L, ld - 6));
    us->b_projected = true;
    us->b_gather_token = can_gather ? ctx->gather_stamp : 0;
  } else {
    PLV_HIP_CHECK(plv::memcpy_async(us->bcols_l.p, col_to_state, (size_t)k * 4, hipMemcpyHostToDevice, ctx->stream));
    TRY(launch_line_jacobians(ctx, P));
  }
  us->bF = L;
  us->bfdim = 6;
  us->bk = k;
  us->bld = ld;
  us->bmaxrows = ld;
  us->b_on_device_rows = true;
  return PLV_OK;
}

}  // namespace

extern "C" {

// The line twin (plv_camera_update_lines): line triangulation, selection, Pluecker Jacobians, null space, gate, compression, EKF.
// st_tri: the state the lines are triangulated on (the one before the point update, plv_camera_get_line_features); the Jacobians
// are taken at st.
// The line half's one-submission update in two steps: `submit` stages the pool and enqueues triangulation + Jacobians + null space +
// gate (one launch, the gate's verdicts and the triangulated lines go to pinned memory); `finish` waits for the gate, enqueues
// compression + EKFUpdate only when something was accepted, and collects.  plv_camera_try_update calls `submit` while the point
// update of the frame is still running (plv_ctx::chain.on: the launch then forms the corrected state itself, JacParams::chain_dx).
int plv_lines_update_fused_submit(plv_ctx *ctx, const plv_state_view *st, const plv_state_view *st_tri, const plv_line_tracks *all,
                                  const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2, double chi2_mult) {
  if (!ctx || !all || !flags) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  FusedLineTri ft{flags, max_sel, 0, 0, st_tri};
  const int L = all->n_lines;
  TRY(us->h_tri_l.reserve((size_t)L * 49 + 16));
  TRY(plv_update_gate_prepare(ctx, L, 6, k, ld, sigma2, chi2_mult, 0.0, 1));
  ctx->gate_stage.probe_dst = (unsigned char *)us->h_tri_l.p;  // (probe_src and the strides: build_lines_on_device, where the results' place is decided)
  TRY(build_lines_on_device(ctx, us, st, all, k, col_to_state, ld, true, &ft));
  us->b_single_use = true;
  us->lt_o_lines = ft.o_lines, us->lt_L = L;
  return PLV_OK;
}
int plv_lines_update_fused_finish(plv_ctx *ctx, double sigma2, double chi2_mult, double *lines_out, uint8_t *ok_out, uint8_t *accepted, int *n_rows,
                                  double *dx, void (*before_wait)(void *), void *before_wait_arg) {
  if (!ctx || !lines_out || !ok_out || !accepted || !dx) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  const int L = us->lt_L;
  // the gate's workgroups leave their verdicts and the triangulated lines in pinned memory and the launch function looks at them
  // before it enqueues compression + EKF (plv_ctx::probe): a line update in which nothing passes the gate ends there
  ctx->probe = true, ctx->probe_done = false;
  ctx->probe_src = us->tri_l.as<char>() + us->lt_o_lines, ctx->probe_dst = us->h_tri_l.p;
  ctx->probe_stride_a = 48, ctx->probe_off_b = L * 48, ctx->probe_stride_b = 1;
  ctx->probe_hook = before_wait, ctx->probe_hook_arg = before_wait_arg;
  int rc = plv_msckf_update_resident_launch(ctx, sigma2, chi2_mult, 0.0);
  ctx->probe = false, ctx->probe_src = nullptr, ctx->probe_dst = nullptr, ctx->probe_hook = nullptr, ctx->probe_hook_arg = nullptr;
  if (rc == PLV_OK) rc = plv_msckf_update_resident_wait(ctx, accepted, n_rows, dx);
  else PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  const char *h = us->h_tri_l.as<char>();
  memcpy(lines_out, h, (size_t)L * 48);
  memcpy(ok_out, h + (size_t)L * 48, (size_t)L);
  return rc;
}
int plv_lines_update_fused(plv_ctx *ctx, const plv_state_view *st, const plv_state_view *st_tri, const plv_line_tracks *all,
                           const uint8_t *flags, int max_sel, int k, const int *col_to_state, int ld, double sigma2, double chi2_mult,
                           double *lines_out, uint8_t *ok_out, uint8_t *accepted, int *n_rows, double *dx, void (*before_wait)(void *),
                           void *before_wait_arg) {
  if (!ctx || !all || !flags || !lines_out || !ok_out || !accepted || !dx) return PLV_E_BADARG;
  TRY(plv_lines_update_fused_submit(ctx, st, st_tri, all, flags, max_sel, k, col_to_state, ld, sigma2, chi2_mult));
  return plv_lines_update_fused_finish(ctx, sigma2, chi2_mult, lines_out, ok_out, accepted, n_rows, dx, before_wait, before_wait_arg);
}

int plv_line_jacobian_columns(const plv_state_view *st, const plv_line_tracks *lt, int *col_to_state, int cap, int *k_out) {
  if (!col_to_state || !k_out) return PLV_E_BADARG;
  TRY(check_line_views(st, lt, false, false));
  int k = 0;
  auto push = [&](int id, int size) {
    if (id < 0) return true;
    for (int j = 0; j < k; ++j)
      if (col_to_state[j] == id) return true;
    if (k + size > cap) return false;
    for (int d = 0; d < size; ++d) col_to_state[k++] = id + d;
    return true;
  };
  // REF: LineHelper.cpp:757-788 — `order` of get_interpolated_jacobian: four poses, then the time offset
  std::vector<uint8_t> seen_s0((size_t)std::max(st->n_clones, 1), 0);  // (see plv_jacobian_columns)
  BoundingStartMemo start_of(*st);
  for (int l = 0; l < lt->n_lines; ++l)
    for (int o = lt->obs_ptr[l]; o < lt->obs_ptr[l + 1]; ++o) {
      const int s0 = start_of(lt->obs_time[o] + st->cam_dt);
      if (s0 < 0 || seen_s0[s0]) continue;
      seen_s0[s0] = 1;
      for (int w = 0; w < 4; ++w)
        if (!push(st->clone_state_id[s0 + w], 6)) return PLV_E_CAPACITY;
      if (!push(st->dt_state_id, 1)) return PLV_E_CAPACITY;
    }
  *k_out = k;
  return PLV_OK;
}

int plv_build_line_jacobians_resident(plv_ctx *ctx, const plv_state_view *st, const plv_line_tracks *lt, int k,
                                      const int *col_to_state, int ld) {
  if (!ctx) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  TRY(build_lines_on_device(ctx, us, st, lt, k, col_to_state, ld, true));
  us->b_single_use = true;
  return PLV_OK;
}

int plv_build_line_jacobians(plv_ctx *ctx, const plv_state_view *st, const plv_line_tracks *lt, int k,
                             const int *col_to_state, int ld, int *rows, double *Hf, double *Hx, double *res) {
  if (!ctx || !rows || !Hf || !Hx || !res) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  TRY(build_lines_on_device(ctx, us, st, lt, k, col_to_state, ld, false));
  us->b_single_use = false;
  const int L = lt->n_lines;
  const size_t nHf = (size_t)L * 6 * ld, nHx = (size_t)L * k * ld, nr = (size_t)L * ld;
  const double *d = us->bHf.as<double>();
  PLV_HIP_CHECK(plv::memcpy_async(Hf, d, nHf * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(Hx, d + nHf, nHx * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(res, d + nHf + nHx, nr * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(rows, us->brows.p, (size_t)L * 4, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  return PLV_OK;
}

int plv_triangulate_lines(plv_ctx *ctx, const plv_state_view *st, const plv_line_tracks *lt, double *line_FinG,
                          uint8_t *ok) {
  if (!ctx || !line_FinG || !ok) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  TRY(check_line_views(st, lt, false, true));
  const int L = lt->n_lines, nobs = lt->obs_ptr[L];
  int dummy_cols[1] = {-1};
  JacParams P{};
  TRY(stage_line_inputs(ctx, us, st, lt, 0, dummy_cols, 2, P));
  const size_t o_cam = 0, o_imu = (size_t)nobs * 96, o_lines = o_imu + (size_t)nobs * 96, o_valid = o_lines + (size_t)L * 48,
               o_ok = o_valid + ((nobs + 15) & ~(size_t)15), total = o_ok + L + 16;
  TRY(us->tri.reserve(total));
  char *d = us->tri.as<char>();
  TRY(launch_triangulate_lines(ctx, P, (double *)(d + o_cam), (double *)(d + o_imu), (unsigned char *)(d + o_valid),
                               (double *)(d + o_lines), (unsigned char *)(d + o_ok)));
  PLV_HIP_CHECK(plv::memcpy_async(line_FinG, d + o_lines, (size_t)L * 48, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(ok, d + o_ok, (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  return PLV_OK;
}

// The record behind a query time, as cpi_pose_kernel finds it, for its covariance (host arithmetic: a search and one blend).
int plv_cpi_noise(const plv_state_view *st, const plv_cpi_table *cpi, int n_q, const double *t_q, double *Q, int *clone_index, uint8_t *ok) {
  if (!st || !cpi || n_q < 0 || (n_q > 0 && (!t_q || !Q || !clone_index || !ok)) || st->n_clones < 1) return PLV_E_BADARG;
  if (cpi->n > 0 && (!cpi->t || !cpi->clone_t || !cpi->Q)) {
    set_last_error("plv_cpi_noise: the table carries no covariances (plv_cpi_table::Q)");
    return PLV_E_BADARG;
  }
  const int n = cpi->n;
  auto find_t = [&](double t) {
    const double *e = std::lower_bound(cpi->t, cpi->t + n, t);
    return (e != cpi->t + n && *e == t) ? (int)(e - cpi->t) : -1;
  };
  auto find_clone = [&](double t) {
    for (int i = 0; i < st->n_clones; ++i)
      if (st->clone_time[i] == t) return i;
    return -1;
  };
  for (int q = 0; q < n_q; ++q) {
    ok[q] = 0;
    clone_index[q] = -1;
    std::fill(Q + 36 * (size_t)q, Q + 36 * (size_t)q + 36, 0.0);
    const double t = t_q[q];
    double clone_t;
    const int e = find_t(t);
    if (e >= 0 && find_clone(cpi->clone_t[e]) >= 0) {  // REF State.cpp:275-277
      std::copy(cpi->Q + 36 * (size_t)e, cpi->Q + 36 * (size_t)e + 36, Q + 36 * (size_t)q);
      clone_t = cpi->clone_t[e];
    } else {  // create_new_cpi_linear :286-355
      if (n == 0 || t < cpi->t[0] || t > cpi->t[n - 1]) continue;
      const int lo = (int)(std::lower_bound(cpi->t, cpi->t + n, t) - cpi->t);
      const int i0 = (t == cpi->t[0]) ? 0 : lo - 1;
      const int up = (int)(std::upper_bound(cpi->t, cpi->t + n, t) - cpi->t);
      const int i1 = (t == cpi->t[n - 1]) ? n - 1 : up;
      if (cpi->clone_t[i0] != cpi->clone_t[i1] || cpi->clone_t[i0] < st->clone_time[0]) continue;
      const double lambda = (t - cpi->t[i0]) / (cpi->t[i1] - cpi->t[i0]);
      for (int j = 0; j < 36; ++j) Q[36 * (size_t)q + j] = (1 - lambda) * cpi->Q[36 * (size_t)i0 + j] + lambda * cpi->Q[36 * (size_t)i1 + j];
      clone_t = cpi->clone_t[i0];
    }
    const int ci = find_clone(clone_t);
    if (ci < 0 || find_t(clone_t) < 0) continue;
    clone_index[q] = ci;
    ok[q] = 1;
  }
  return PLV_OK;
}

int plv_cpi_poses(plv_ctx *ctx, const plv_state_view *st, const plv_cpi_table *cpi, int n_q, const double *t_q, double *R_GtoI,
                  double *p_IinG, uint8_t *ok) {
  if (!ctx || !st || !cpi || n_q < 0 || (n_q > 0 && (!t_q || !R_GtoI || !p_IinG || !ok)) || cpi->n < 0 || st->n_clones < 1)
    return PLV_E_BADARG;
  if (cpi->n > 0 && (!cpi->t || !cpi->clone_t || !cpi->dt || !cpi->R_I0toIk || !cpi->alpha || !cpi->v)) return PLV_E_BADARG;
  for (int i = 1; i < cpi->n; ++i)
    if (!(cpi->t[i - 1] < cpi->t[i])) {
      set_last_error("plv_cpi_poses: the table must be strictly ascending in time (record %d)", i);
      return PLV_E_BADARG;
    }
  if (n_q == 0) return PLV_OK;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  const size_t n = (size_t)cpi->n, nc = (size_t)st->n_clones, nq = (size_t)n_q;
  // one packed upload: [t n][clone_t n][dt n][R 9n][alpha 3n][v 3n][clone_time nc][clone_R 9nc][clone_p 3nc][t_q nq]
  const size_t in_d = 18 * n + 13 * nc + nq, out_d = 12 * nq;
  std::vector<double> h(in_d);
  double *w = h.data();
  auto put = [&](const double *src, size_t cnt) {
    if (cnt) std::copy(src, src + cnt, w);
    w += cnt;
  };
  put(cpi->t, n), put(cpi->clone_t, n), put(cpi->dt, n), put(cpi->R_I0toIk, 9 * n), put(cpi->alpha, 3 * n), put(cpi->v, 3 * n);
  put(st->clone_time, nc), put(st->clone_R, 9 * nc), put(st->clone_p, 3 * nc), put(t_q, nq);
  TRY(us->tri.reserve((in_d + out_d) * sizeof(double) + nq + 16));
  double *d = us->tri.as<double>();
  PLV_HIP_CHECK(plv::memcpy_async(d, h.data(), in_d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  CpiParams C{};
  C.n = cpi->n, C.n_clones = st->n_clones, C.n_q = n_q;
  C.t = d, C.clone_t = d + n, C.dt = d + 2 * n, C.R = d + 3 * n, C.alpha = d + 12 * n, C.v = d + 15 * n;
  C.clone_time = d + 18 * n, C.clone_R = C.clone_time + nc, C.clone_p = C.clone_R + 9 * nc;
  const double *d_tq = C.clone_p + 3 * nc;
  double *d_R = d + in_d, *d_p = d_R + 9 * nq;
  unsigned char *d_ok = (unsigned char *)(d_p + 3 * nq);
  std::copy(cpi->gravity, cpi->gravity + 3, C.gravity);
  TRY(launch_cpi_poses(ctx, C, d_tq, d_R, d_p, d_ok));
  PLV_HIP_CHECK(plv::memcpy_async(R_GtoI, d_R, 9 * nq * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(p_IinG, d_p, 3 * nq * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(ok, d_ok, nq, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  for (size_t q = 0; q < nq; ++q)
    if (!ok[q]) {
      std::fill(R_GtoI + 9 * q, R_GtoI + 9 * q + 9, 0.0);
      std::fill(p_IinG + 3 * q, p_IinG + 3 * q + 3, 0.0);
    }
  return PLV_OK;
}

}  // extern "C"
