// tracker_api.hip — host mirror of ov_core::TrackKLT's monocular frame logic and of
// ov_core::FeatureDatabase behind the C-ABI (plv_tracker_* / plv_db_*).
//   TrackKLT::feed_new_camera / feed_monocular   REF: open_vins/ov_core/src/track/TrackKLT.cpp:34-200
//   FeatureDatabase                              REF: open_vins/ov_core/src/feat/FeatureDatabase.cpp:60-323
// This is bookkeeping (vectors and a hash map), exactly what the reference keeps on the host;
// every image / point computation goes through the device entry points of frontend_api.hip.
#include <algorithm>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <sys/resource.h>
#include "plv_ctx.hpp"
#include "update_state.hpp"
#include "gate_stage.hpp"

namespace {

struct Track {  // ov_core::Feature, one camera  (REF: open_vins/ov_core/src/feat/Feature.h:43-77)
  std::vector<double> t;
  std::vector<float> uv, uvn;  // 2 per observation
  // (transient, Tracker::Spec) index of the track's point in the flow's batch of feed number li_seq
  int li = -1;
  unsigned long long li_seq = 0;
};

struct Tracker {
  std::vector<float> pts_last;     // 2 per point
  std::vector<uint64_t> ids_last;
  std::vector<uint8_t> mask_last;  // W*H or empty
  uint64_t currid = 0;             // REF: TrackBase::currid (4*num_aruco + 1 - 1 = 0 without ArUco tags)
  int detect_ahead = 2;            // plv_tracker_detect_ahead: start the next frame's top-up detection ahead of time (1: at the end of the feed, 2: once the point update is submitted)
  bool ahead_deferred = false;     // ... asked for by the last feed, not started yet (start_detection_ahead)
  bool ahead_on_ctx_stream = false;  // plv_camera_try_update with a line update to follow: the detection goes behind the point update
  bool defer_db = false;           // plv_camera_try_update: the point update leaves its database hand-back to run_deferred_db
  const plv_state_view *early_st = nullptr;       // plv_camera_try_update with a line update to follow: the line pool is formed inside
  const plv_update_options *early_lines = nullptr;  // the point update's wait when the frame's line feed has finished by then
  int early_cap = 0;                              // ... and, chained, the whole first half of the line update (line_cap)
  // the pool of the fused point update being run: feature id -> index (only where the selection cannot hit its cap and the
  // candidate has enough observations with bounding clones); the chained line launch finds its anchors through it
  std::unordered_map<uint64_t, int> chain_index;
  bool chain_ok = false;
  std::function<void()> deferred_db;
  std::unordered_map<uint64_t, Track> db;
  struct UsedPoint {
    double p[3], newest;
  };
  std::unordered_map<uint64_t, UsedPoint> used;  // the reference's `point_used` database (triangulated features)
  struct Listed {  // a feature get_features classified as SLAM / SLAM-init in the last update call
    uint64_t id;
    Track tr;
    double p[3];
  };
  std::vector<Listed> last_slam, last_init;
  // plv_decision_trace: the last point update's pool, feature by feature (plv_last_point_decisions)
  std::vector<uint64_t> dec_ids;
  std::vector<double> dec_vals;  // [pool][PLV_DECISION_VALUES]
  // Speculative submission of the point update (round 6, plv_camera_frame): while the frame's flow runs on the device, every track that
  // flow could send into the update's pool is staged and the whole update chain is enqueued behind the flow; a kernel decides the
  // membership from the flow's result (spec_select_kernel).  plv_camera_update_points, which forms the pool on the host as before (it
  // keeps the database), then finds its update already running.
  const plv_state_view *spec_st = nullptr;        // set by plv_camera_frame around the feed: the update this frame will ask for
  const plv_update_options *spec_opt = nullptr;
  // (plv_camera_frame) called by the feed the moment the frame's point list (pts_last / ids_last) stands, in front of the database
  // update: posts the line tracker's feed to its worker ~20 us earlier (the line worker's path is the frame's longer one)
  std::function<void(int, const float *, const uint64_t *)> points_ready;
  unsigned long long feed_seq = 0;                // feeds so far (Track::li_seq)
  std::vector<double> frame_t;                    // time stamps of the last feeds, ascending (a track's observations carry these very values)
  struct Spec {
    bool active = false;                          // a speculative batch is on the stream (not collected yet)
    double t_prev_frame = 0, state_time = 0, dt = 0, t_oldest = 0, t_oldest2 = 0;   // what the batch was staged for
    int n_clones = 0, max_msckf = 0, max_obs = 0, k = 0, F = 0;
    std::unordered_map<uint64_t, int> index;      // track id -> candidate
    std::vector<uint64_t> ids;
    std::vector<int> ptr, li, cols;
    std::vector<uint8_t> meta, prevalid, flags;
    std::vector<double> ot;
    std::vector<float> ouv, ouvn;
  } spec;
  std::mutex mtx;
};

std::mutex g_mtx;
std::unordered_map<plv_ctx *, Tracker *> g_trk;
Tracker *trk(plv_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_mtx);
  auto it = g_trk.find(ctx);
  if (it != g_trk.end()) return it->second;
  Tracker *t = new Tracker();
  g_trk[ctx] = t;
  return t;
}

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

}  // namespace

// the same test answered from a small memo: the observations of a window carry ~16 distinct time stamps (the camera frames), asked
// for thousands of times per update
template <class F> struct BoundingMemo {
  F f;
  double t[40];
  bool v[40];
  int n = 0;
  explicit BoundingMemo(F f_) : f(f_) {}
  bool operator()(double tq) {
    for (int i = n - 1; i >= 0; --i)
      if (t[i] == tq) return v[i];
    const bool r = f(tq);
    if (n < 40) t[n] = tq, v[n++] = r;
    return r;
  }
};
template <class F> BoundingMemo<F> bounding_memo(F f) { return BoundingMemo<F>(f); }

extern "C" {

void plv_tracker_destroy(plv_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_mtx);
  auto it = g_trk.find(ctx);
  if (it != g_trk.end()) {
    delete it->second;
    g_trk.erase(it);
  }
}

static int tracker_feed_fed(plv_ctx *ctx, Tracker *T, double timestamp, const uint8_t *mask);
extern "C" int plv_feed_image_enqueue(plv_ctx *ctx, const uint8_t *img, int stride);  // frontend_api.hip
extern "C" int plv_line_prefetch_enabled(plv_ctx *ctx);                               // line_api.hip
extern "C" void plv_line_edges_early(plv_ctx *ctx, const uint8_t *d_raw, int W, int H, const unsigned *d_hist);
// the image feed with the line detector's edge kernel between its histogram and its pyramid (plv_ctx::edges_hook) when the frame's
// lines are detected ahead of the line tracker's feed anyway; PLV_KNOB_EDGES_LATE / the other edge knobs keep the older orders
static int feed_with_early_edges(plv_ctx *ctx, const std::function<int()> &feed) {
  const bool early = !plv::knob(plv::PLV_KNOB_EDGES_LATE | plv::PLV_KNOB_EDGES_SIDE | plv::PLV_KNOB_EDGES_AFTER_PYRAMID) && plv_line_prefetch_enabled(ctx) != 0;
  ctx->edges_hook_fired = false;
  ctx->edges_hook = early ? plv_line_edges_early : nullptr;
  const int rc = feed();
  ctx->edges_hook = nullptr;
  return rc;
}

int plv_tracker_feed(plv_ctx *ctx, double timestamp, const uint8_t *img, int stride, const uint8_t *mask) {
  if (!ctx || !img) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);  // REF: mtx_feeds.at(cam_id), TrackKLT.cpp:54,100
  // :59 equalizeHist, :71 buildOpticalFlowPyramid (enqueued; the feed below waits for its flow)
  TRY(feed_with_early_edges(ctx, [&] { return plv_feed_image_enqueue(ctx, img, stride); }));
  return tracker_feed_fed(ctx, T, timestamp, mask);
}

int plv_tracker_feed_staged(plv_ctx *ctx, double timestamp, int slot, const uint8_t *mask) {
  if (!ctx) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  TRY(feed_with_early_edges(ctx, [&] { return plv_feed_staged(ctx, slot); }));  // the image is already in HBM (plv_image_stage)
  return tracker_feed_fed(ctx, T, timestamp, mask);
}

int plv_tracker_feed_downsampled(plv_ctx *ctx, double timestamp, const uint8_t *img, int stride, int src_w, int src_h,
                                 const uint8_t *mask, int mask_stride) {
  if (!ctx || !img) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  const int W = ctx->cfg.width, H = ctx->cfg.height;
  std::vector<uint8_t> small_mask;
  if (mask) {  // REF UpdaterCamera.cpp:93 the mask is pyrDown'ed like the image (and then thresholded at 127 as usual)
    small_mask.resize((size_t)W * H);
    TRY(plv_downsample(ctx, mask, mask_stride, src_w, src_h, small_mask.data(), W));
  }
  TRY(plv_feed_image_downsampled(ctx, img, stride, src_w, src_h));
  return tracker_feed_fed(ctx, T, timestamp, mask ? small_mask.data() : nullptr);
}

static bool has_bounding_poses(const plv_state_view &st, double t);
extern "C++" {
namespace plv { int plv_front_match_device(plv_ctx *ctx, const float **d_p1, const float **d_n1, const uint8_t **d_mask, int *n); }  // frontend_api.hip
}
// Stages every track the frame's flow could send into the point update's pool and enqueues the update behind the flow (Tracker::Spec).
// Called with T->mtx held, between the flow's launch and the wait for it; tp[i]: the database track of flow point i (or null).
// Anything unusual leaves spec.active false: plv_camera_update_points then submits the update itself, as it always did.
static void spec_submit(plv_ctx *ctx, Tracker *T, double t_now, int n_flow, const uint64_t *flow_ids, Track *const *tp) {
  Tracker::Spec &S = T->spec;
  S.active = false;
  const plv_state_view *st = T->spec_st;
  const plv_update_options *opt = T->spec_opt;
  if (plv::knob(plv::PLV_KNOB_NO_SPECULATION | plv::PLV_KNOB_GATE_SEPARATE | plv::PLV_KNOB_POINT_TRI_SEPARATE | plv::PLV_KNOB_INPUTS_PINNED) || !st || !opt || opt->cpi || opt->max_slam > 0 || opt->n_slam > 0 || st->n_clones < 4 || opt->max_msckf < 1 ||
      opt->max_obs < 2 || ctx->cov_n < 1 || ctx->decision_trace || n_flow < 10 || ctx->prof.on)
    return;
  {
    const plv_ctx_update_state *us0 = plv_update_state(ctx);
    if (us0->compress_mode != 0 || us0->graph_mode) return;  // (the chain must need no host decision: the whitened route)
  }
  plv::HostPhase ph("speculative point update: candidates staged + chain enqueued");
  const double dt = st->cam_dt, t_oldest = st->clone_time[0], t_oldest2 = st->clone_time[1];
  auto has_bounding = bounding_memo([st](double tq) { return has_bounding_poses(*st, tq); });
  const double tm_new = t_now + dt;
  const bool new_usable = !(tm_new > opt->state_time + st->dt_exp) && !(tm_new < t_oldest - st->dt_exp);
  const bool new_bounded = new_usable && has_bounding(tm_new);
  if (!(t_now > opt->t_prev_frame - dt)) return;  // (a surviving track counts as "seen in the newest frame": CamHelper.cpp:635, the usual case)
  // What the window tests say about an observation depends on its time alone, and the observations of all tracks carry the time
  // stamps of the last few feeds: the tests are evaluated once per stamp, the tracks are then walked with integer work only.
  const std::vector<double> &ft = T->frame_t;
  const int nt = (int)ft.size();
  struct TimeInfo {
    bool older, newer_old, too_new, too_old, bounded;
  };
  std::vector<TimeInfo> ti(nt);
  for (int j = 0; j < nt; ++j) {
    const double t = ft[j], tm = t + dt;
    ti[j] = TimeInfo{t < t_oldest2 - dt, t > opt->t_prev_frame - dt, tm > opt->state_time + st->dt_exp, tm < t_oldest - st->dt_exp, false};
    ti[j].bounded = !ti[j].too_new && !ti[j].too_old && has_bounding_poses(*st, tm);
  }
  struct C {
    uint64_t id;
    const Track *tr;
    int li, keep;
    uint8_t meta, prevalid;
  };
  std::vector<C> cand;
  cand.reserve(T->db.size());
  bool bail = false;
  auto consider = [&](uint64_t id, const Track &tr, int li) {
    const int n = (int)tr.t.size();
    if (n == 0) return;
    bool older = false, newer_old = false;
    int keep = 0, pv = 0;
    int j = (int)(std::lower_bound(ft.begin(), ft.end(), tr.t[0]) - ft.begin());
    for (int i = 0; i < n; ++i) {
      while (j < nt && ft[j] < tr.t[i]) ++j;
      if (j >= nt || ft[j] != tr.t[i])  // (a track handed back in pieces need not be in time order: look this one up on its own)
        j = (int)(std::lower_bound(ft.begin(), ft.end(), tr.t[i]) - ft.begin());
      if (j >= nt || ft[j] != tr.t[i]) {  // (an observation that is not of a recent feed — plv_db_append_measurements: the long way)
        if (plv::host_phases().on) plv::host_phases().add("speculative point update: no - an observation time is not a recent feed's (count)", 1.0);
        bail = true;
        return;
      }
      const TimeInfo &q = ti[j];
      older = older || q.older;
      newer_old = newer_old || q.newer_old;
      if (q.too_new) {  // (an observation newer than the window: the long way handles the hand-back)
        if (plv::host_phases().on) plv::host_phases().add("speculative point update: no - an observation newer than the window (count)", 1.0);
        bail = true;
        return;
      }
      if (q.too_old) continue;
      ++keep;
      pv += q.bounded ? 1 : 0;
    }
    if (!older && newer_old) return;                                  // never in the pool: in the pool = older || !(newer_old || survived)
    if (keep + (li >= 0 && new_usable ? 1 : 0) < 2) return;            // cannot reach two usable observations
    if (pv > 255) {
      bail = true;
      return;
    }
    cand.push_back(C{id, &tr, li, keep, (uint8_t)((older ? 1 : 0) | (new_usable ? 2 : 0) | (new_bounded ? 4 : 0) | (newer_old ? 8 : 0)), (uint8_t)pv});
  };
  // the candidates in an order that does not depend on the hash map's: the tracked points in the flow's order, then the tracks of the
  // database that were not tracked into this frame by id
  int n_live = 0;
  for (int i = 0; i < n_flow && !bail; ++i)
    if (tp[i]) {
      tp[i]->li = i, tp[i]->li_seq = T->feed_seq;
      ++n_live;
      consider(flow_ids[i], *tp[i], i);
    }
  if ((int)T->db.size() > n_live && !bail) {
    std::vector<std::pair<uint64_t, const Track *>> rest;
    for (const auto &kv : T->db)
      if (kv.second.li_seq != T->feed_seq) rest.emplace_back(kv.first, &kv.second);
    std::sort(rest.begin(), rest.end(), [](const std::pair<uint64_t, const Track *> &a, const std::pair<uint64_t, const Track *> &b) { return a.first < b.first; });
    for (const auto &r : rest) {
      if (bail) break;
      consider(r.first, *r.second, -1);
    }
  }
  if (bail) return;
  if (cand.empty()) return;
  const int F = (int)cand.size();
  S.index.clear();
  S.ids.resize(F), S.ptr.assign(F + 1, 0), S.li.resize(F), S.meta.resize(F), S.prevalid.resize(F), S.flags.assign(F, 0);
  int most_valid = 0;
  for (int f = 0; f < F; ++f) {
    S.ptr[f + 1] = S.ptr[f] + cand[f].keep + (cand[f].li >= 0 ? 1 : 0);
    most_valid = std::max(most_valid, (int)cand[f].prevalid + ((cand[f].li >= 0 && new_bounded) ? 1 : 0));
  }
  if (most_valid > opt->max_obs) {  // (the two-step route of over-long tracks)
    if (plv::host_phases().on) plv::host_phases().add("speculative point update: no - a track longer than max_obs (count)", 1.0);
    return;
  }
  const int nobs = S.ptr[F];
  S.ot.resize(nobs), S.ouv.resize(2 * (size_t)nobs), S.ouvn.resize(2 * (size_t)nobs);
  for (int f = 0; f < F; ++f) {
    const C &c = cand[f];
    const Track &tr = *c.tr;
    S.ids[f] = c.id, S.li[f] = c.li, S.meta[f] = c.meta, S.prevalid[f] = c.prevalid;
    S.index.emplace(c.id, f);
    int o = S.ptr[f];
    for (size_t i = 0; i < tr.t.size(); ++i) {
      if (tr.t[i] + dt < t_oldest - st->dt_exp) continue;
      S.ot[o] = tr.t[i];
      S.ouv[2 * o] = tr.uv[2 * i], S.ouv[2 * o + 1] = tr.uv[2 * i + 1];
      S.ouvn[2 * o] = tr.uvn[2 * i], S.ouvn[2 * o + 1] = tr.uvn[2 * i + 1];
      ++o;
    }
    if (c.li >= 0) {  // the slot of this frame's observation: the time now, the image points by the device
      S.ot[o] = t_now;
      S.ouv[2 * o] = S.ouv[2 * o + 1] = S.ouvn[2 * o] = S.ouvn[2 * o + 1] = 0.f;
    }
  }
  std::vector<double> pf(3 * (size_t)F, 0.0);
  plv_tracks all{};
  all.n_feat = F;
  all.obs_ptr = S.ptr.data();
  all.obs_time = S.ot.data();
  all.obs_uv = S.ouv.data();
  all.obs_uvn = S.ouvn.data();
  all.p_FinG = all.p_FinG_fej = pf.data();
  S.cols.resize(ctx->cfg.max_state_dim > 0 ? ctx->cfg.max_state_dim : 1024);
  int k = 0;
  {  // the column set of the candidates = that of one track observed at every feed time inside the window (+ this frame's)
    std::vector<double> tt;
    for (int j = 0; j < nt; ++j)
      if (!ti[j].too_new && !ti[j].too_old) tt.push_back(ft[j]);
    if (tt.empty() || tt.back() != t_now) tt.push_back(t_now);
    const int one_ptr[2] = {0, (int)tt.size()};
    std::vector<float> zuv(2 * tt.size(), 0.f);
    const double zp[3] = {0, 0, 0};
    plv_tracks one{};
    one.n_feat = 1, one.obs_ptr = one_ptr, one.obs_time = tt.data(), one.obs_uv = zuv.data(), one.obs_uvn = zuv.data(), one.p_FinG = one.p_FinG_fej = zp;
    if (plv_jacobian_columns(st, &one, S.cols.data(), (int)S.cols.size(), &k) != PLV_OK || k < 1) return;
  }
  if (k > GATE_KMAX || 2 * most_valid - 3 > GATE_MMAX) {  // (the gate must run inside the Jacobian launch: with it outside every candidate's block is written and walked)
    if (plv::host_phases().on) plv::host_phases().add("speculative point update: no - the gate would not fit the Jacobian launch (count)", 1.0);
    return;
  }
  plv_points_spec sp{};
  sp.n_flow = n_flow, sp.li = S.li.data(), sp.meta = S.meta.data(), sp.prevalid = S.prevalid.data();
  int n_dev = 0;
  if (plv::plv_front_match_device(ctx, &sp.d_flow_p1, &sp.d_flow_n1, &sp.d_flow_mask, &n_dev) != PLV_OK || n_dev != n_flow) return;
  (void)flow_ids;
  ctx->gate_rows_hint = 2 * most_valid;
  if (plv_points_update_submit(ctx, st, &all, &opt->tri, S.flags.data(), opt->max_msckf, k, S.cols.data(), 2 * opt->max_obs,
                               st->sigma_pix * st->sigma_pix, opt->chi2_mult, 3.0, &sp) != PLV_OK) {
    // (nothing usable was enqueued: whatever the failed call left on the stream works on empty or stale candidates and commits nothing
    //  it was not told to; the long way follows)
    plv_update_state(ctx)->point_job.pending = false;
    return;
  }
  S.active = true;
  S.F = F, S.k = k;
  S.t_prev_frame = opt->t_prev_frame, S.state_time = opt->state_time, S.dt = dt, S.t_oldest = t_oldest, S.t_oldest2 = t_oldest2;
  S.n_clones = st->n_clones, S.max_msckf = opt->max_msckf, S.max_obs = opt->max_obs;
}

// the rest of TrackKLT::feed_monocular once the image is equalised and its pyramid built
extern "C" int plv_line_edges_fork(plv_ctx *ctx);         // line_api.hip
extern "C" void plv_line_defer_finish(plv_ctx *ctx, int on);
extern "C" void plv_line_run_deferred(plv_ctx *ctx);
extern "C" int plv_perform_detection_ahead(plv_ctx *ctx, const uint8_t *mask, const float *pts, const uint64_t *ids, int n_in, int on_ctx_stream);  // frontend_api.hip
static int tracker_feed_fed(plv_ctx *ctx, Tracker *T, double timestamp, const uint8_t *mask) {
  const int W = ctx->cfg.width, H = ctx->cfg.height;
  ++T->feed_seq;
  if (T->frame_t.empty() || timestamp > T->frame_t.back()) {
    T->frame_t.push_back(timestamp);
    if (T->frame_t.size() > 96) T->frame_t.erase(T->frame_t.begin(), T->frame_t.begin() + 32);
  } else if (timestamp < T->frame_t.back()) {
    T->frame_t.clear();  // (time went backwards: a new sequence on the same context)
    T->frame_t.push_back(timestamp);
  }
  plv::HostPhase ph_all("tracker_feed (after the image feed)");
  // With the line prefetch on (plv_line_prefetch_mode), resize + Canny of the new image and the copies of the two maps go first on
  // the stream and the library's line worker thread walks the edge chains and grows the segments while this thread runs the point
  // front-end; plv_line_tracker_feed of the same frame joins it.
  // (enqueued further down, behind the flow + RANSAC of this frame: the point front-end starts the moment the pyramid is built, and
  // the line worker, whose host stage has slack against the point update, gets its edge maps ~0.1 ms later)
  bool prefetch_lines = plv_line_prefetch_enabled(ctx) != 0 && !ctx->edges_hook_fired;  // (fired: the image feed launched it already)
  ctx->edges_hook_fired = false;
  auto launch_prefetch = [&]() {
    if (prefetch_lines) (void)plv_line_detect_launch(ctx, PLV_PYR_CUR);
    prefetch_lines = false;
  };
  T->ahead_deferred = false;  // (no update came in between: the detection below runs in place)
  const int cap = std::max(ctx->cfg.num_features * 4, 1024) + (int)T->ids_last.size();
  std::vector<float> pts(2 * (size_t)cap);
  std::vector<uint64_t> ids(cap);
  int n = 0;
  auto keep_mask = [&]() {
    if (mask)
      T->mask_last.assign(mask, mask + (size_t)W * H);
    else
      T->mask_last.clear();
  };
  if (T->ids_last.empty()) {  // :110-122 first frame / lost everything: detect on the current image
    launch_prefetch();
    TRY(plv_perform_detection(ctx, PLV_PYR_CUR, mask, pts.data(), ids.data(), 0, cap, &T->currid, &n));
    T->pts_last.assign(pts.begin(), pts.begin() + 2 * (size_t)n);
    T->ids_last.assign(ids.begin(), ids.begin() + n);
    keep_mask();
    return PLV_OK;
  }
  // :127-131 top-up on the LAST image
  n = (int)T->ids_last.size();
  std::copy(T->pts_last.begin(), T->pts_last.end(), pts.begin());
  std::copy(T->ids_last.begin(), T->ids_last.end(), ids.begin());
  {
    plv::HostPhase ph("tracker_feed: perform_detection");
    TRY(plv_perform_detection(ctx, PLV_PYR_LAST, T->mask_last.empty() ? nullptr : T->mask_last.data(), pts.data(), ids.data(), n,
                              cap, &T->currid, &n));
  }
  // :134-139 temporal KLT with the previous positions as the initial flow
  std::vector<float> pts_new(pts.begin(), pts.begin() + 2 * (size_t)n), n1(2 * (size_t)std::max(n, 1));
  std::vector<uint8_t> mask_ll((size_t)std::max(n, 1), 0);
  std::vector<Track *> tp;  // the listed points' tracks (null: not in the database yet), filled inside the wait for the flow
  if (n == 0) {  // :143-152
    launch_prefetch();
    T->pts_last.clear();
    T->ids_last.clear();
    keep_mask();
    return PLV_OK;
  }
  {
    plv::HostPhase ph("tracker_feed: perform_matching");
    // the line detector's pixel work (18 us) goes in FRONT of the flow: the edge maps then reach the library's line worker ~0.1 ms
    // earlier than behind flow + RANSAC, and the worker (chain walk, segment growth, assignment, matching) is the longer of the two
    // paths that meet at the line update; PLV_KNOB_EDGES_LATE restores the old order
    // (plv_line_edges_fork: a measurement knob that puts the kernel on its own stream behind the pyramid instead — the flow then does
    //  not wait 18 us for it, and yet the frame is 6-15 us slower, measured alternating frame by frame)
    if (!plv::knob(plv::PLV_KNOB_EDGES_LATE) && prefetch_lines && plv_line_edges_fork(ctx) != PLV_OK) launch_prefetch();
    const int rc_l = plv_perform_matching_launch(ctx, n, pts.data(), pts_new.data());
    launch_prefetch();  // (inside the wait for the flow)
    if (rc_l == PLV_OK) plv_line_run_deferred(ctx);  // the previous frame's line database hand-back, if one was left behind
    TRY(rc_l);
    // ... and, while the flow runs: where every listed point's track sits in the database (one look-up each, the track's vectors
    // touched), so that the database update behind the wait appends through pointers into warm cache lines (30 -> ~12 us at 340 points)
    tp.assign((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {
      auto it = T->db.find(ids[i]);
      if (it == T->db.end()) continue;
      Track &tr = it->second;
      tp[i] = &tr;
      if (!tr.t.empty()) {
        __builtin_prefetch(&tr.t.back() + 1, 1);
        __builtin_prefetch(&tr.uv.back() + 1, 1);
        __builtin_prefetch(&tr.uvn.back() + 1, 1);
      }
    }
    // the frame's point update, enqueued behind the flow before its result is known (see Tracker::Spec)
    if (T->spec_st && T->spec_opt && !mask) spec_submit(ctx, T, timestamp, n, ids.data(), tp.data());
    plv::NsScope ns_wait(plv::counters().flow_wait_ns);
    TRY(plv_perform_matching_wait(ctx, pts_new.data(), mask_ll.data(), nullptr, n1.data(), nullptr));
  }
  plv::HostPhase ph_db("tracker_feed: database update");
  // :158-173 keep in-bounds, unmasked, matched points; :176-179 database update
  std::vector<float> good;
  std::vector<uint64_t> good_ids;
  std::vector<uint8_t> is_good((size_t)n, 0);
  good.reserve(2 * (size_t)n), good_ids.reserve((size_t)n);
  for (int i = 0; i < n; ++i) {
    const float x = pts_new[2 * i], y = pts_new[2 * i + 1];
    if (x < 0 || y < 0 || (int)x >= W || (int)y >= H) continue;
    if (mask && mask[(size_t)(int)y * W + (int)x] > 127) continue;
    if (!mask_ll[i]) continue;
    is_good[i] = 1;
    good.push_back(x);
    good.push_back(y);
    good_ids.push_back(ids[i]);
  }
  if (T->points_ready) T->points_ready((int)good_ids.size(), good.data(), good_ids.data());  // (the frame's point list stands: the line feed can start)
  for (int i = 0; i < n; ++i) {
    if (!is_good[i]) continue;
    const float x = pts_new[2 * i], y = pts_new[2 * i + 1];
    Track &tr = tp[i] ? *tp[i] : T->db[ids[i]];
    if (tr.t.capacity() == 0) {  // a new track: room for a window's worth of observations (no regrowth frame after frame)
      tr.t.reserve(32);
      tr.uv.reserve(64);
      tr.uvn.reserve(64);
    }
    tr.t.push_back(timestamp);
    tr.uv.push_back(x);
    tr.uv.push_back(y);
    tr.uvn.push_back(n1[2 * i]);
    tr.uvn.push_back(n1[2 * i + 1]);
  }
  T->pts_last.swap(good);
  T->ids_last.swap(good_ids);
  keep_mask();
  // the next frame's top-up works on this image with these points (:127-131): it is started on a side stream once the point update
  // of this frame has been submitted (start_detection_ahead below), so that its host stage and launches sit in the update's wait
  // instead of in front of it; without an update in between, the next feed detects in place as usual
  T->ahead_deferred = T->detect_ahead == 2 && !T->ids_last.empty();
  if (T->detect_ahead == 1 && !T->ids_last.empty())
    (void)plv_perform_detection_ahead(ctx, mask, T->pts_last.data(), T->ids_last.data(), (int)T->ids_last.size(), 0);
  return PLV_OK;
}

extern "C" int plv_line_pool_prepare(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt);  // line_api.hip
extern "C" int plv_line_tracker_feed_async_points(plv_ctx *ctx, double timestamp, const double *vps, int np, const float *pts, const uint64_t *pids);
extern "C" void plv_line_pool_discard(plv_ctx *ctx);
extern "C" int plv_line_db_size_after_feed(plv_ctx *ctx);
// host work placed inside the point update's wait (called without T->mtx held)
static void start_detection_ahead(void *arg) {
  plv_ctx *ctx = (plv_ctx *)arg;
  Tracker *T = trk(ctx);
  {
    std::lock_guard<std::mutex> lk(T->mtx);
    if (T->ahead_deferred) {
      T->ahead_deferred = false;
      // (with a line update to follow: behind the point update on the ctx stream — its wait ends at the update's own last kernel and
      // the detection fills the device's idle time until the line update is submitted; else on the side stream, next to the update)
      (void)plv_perform_detection_ahead(ctx, T->mask_last.empty() ? nullptr : T->mask_last.data(), T->pts_last.data(), T->ids_last.data(),
                                        (int)T->ids_last.size(), T->ahead_on_ctx_stream ? 1 : 0);
    }
  }
}
// LineHelper::get_line_features' pool (times only: nothing the point update changes) while the device is busy with that update;
// polled inside the update's wait (plv_ctx::wait_poll) until the line worker has finished the frame's feed
extern "C" void plv_line_feed_pool_args(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt);  // line_api.hip
extern "C" int plv_camera_lines_submit_chained(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt, int cap);  // line_api.hip
extern "C" int plv_camera_lines_job_pending(plv_ctx *ctx);
extern "C" void plv_camera_lines_job_abort(plv_ctx *ctx);
extern "C" void plv_camera_lines_job_abort2(plv_ctx *ctx, int keep_pool);
static int poll_line_pool(void *arg) {
  plv_ctx *ctx = (plv_ctx *)arg;
  Tracker *T = trk(ctx);
  if (!T->early_lines || !T->early_st) return 1;
  if (!plv_line_pool_prepare(ctx, T->early_st, T->early_lines)) return 0;  // (the frame's line feed is still on the worker: try again)
  if (plv::host_phases().on) {  // (why a frame's line launch was or was not chained: counts in the phase table)
    plv::host_phases().add("chain: pool ready inside the point wait (count)", 1.0);
    if (!T->chain_ok) plv::host_phases().add("chain: no - the point pool exceeds max_msckf or holds no candidate (count)", 1.0);
    if (!ctx->chain.ready) plv::host_phases().add("chain: no - state variables not chainable (count)", 1.0);
    if (!plv_update_state(ctx)->applied_armed) plv::host_phases().add("chain: no - the point launch did not arm the applied word (count)", 1.0);
  }
  // The line half's first half right here, its launch enqueued behind the point update that is still running (round 4): what used to
  // sit between the two device chains — wake-up, selection, dx applied, line staging, upload, launch: ~45 us of idle device — is gone.
  if (T->chain_ok && ctx->chain.ready && plv_update_state(ctx)->applied_armed && !plv::knob(plv::PLV_KNOB_NO_CHAIN))
    (void)plv_camera_lines_submit_chained(ctx, T->early_st, T->early_lines, T->early_cap);
  return 1;
}
// index of a feature in the pool of the point update being run (its triangulation result decides whether point_used gets the point)
int plv_point_chain_lookup(plv_ctx *ctx, uint64_t id) {
  Tracker *T = trk(ctx);
  auto it = T->chain_index.find(id);
  return it == T->chain_index.end() ? -1 : it->second;
}

int plv_decision_trace(plv_ctx *ctx, int on) {
  if (!ctx) return PLV_E_BADARG;
  ctx->decision_trace = on != 0;
  return PLV_OK;
}
int plv_last_point_decisions(plv_ctx *ctx, uint64_t *ids, double *vals, int cap, int *n) {
  if (!ctx || !n || cap < 0 || (cap > 0 && (!ids || !vals))) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  *n = (int)T->dec_ids.size();
  if (*n > cap) return cap == 0 ? PLV_OK : PLV_E_CAPACITY;
  std::copy(T->dec_ids.begin(), T->dec_ids.end(), ids);
  std::copy(T->dec_vals.begin(), T->dec_vals.end(), vals);
  return PLV_OK;
}

int plv_tracker_detect_ahead(plv_ctx *ctx, int on) {
  if (!ctx) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  T->detect_ahead = on < 0 ? 0 : on > 2 ? 2 : on;
  return PLV_OK;
}

// TrackBase::get_last_obs / get_last_ids   REF: open_vins/ov_core/src/track/TrackBase.h:121-131
int plv_tracker_last(plv_ctx *ctx, float *pts, uint64_t *ids, int cap, int *n) {
  if (!ctx || !n) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  const int m = (int)T->ids_last.size();
  *n = m;
  if (m > cap) return PLV_E_CAPACITY;
  if (pts) std::copy(T->pts_last.begin(), T->pts_last.end(), pts);
  if (ids) std::copy(T->ids_last.begin(), T->ids_last.end(), ids);
  return PLV_OK;
}

int plv_db_size(plv_ctx *ctx) {
  if (!ctx) return 0;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  return (int)T->db.size();
}

// mode 0: FeatureDatabase::features_not_containing_newer(t)   REF: FeatureDatabase.cpp:147-190
// mode 1: FeatureDatabase::features_containing_older(t)       REF: FeatureDatabase.cpp:192-232
// ids are returned in ascending order (the reference iterates an unordered_map; SURVEY §7 H6 asks
// for a defined order).
int plv_db_select(plv_ctx *ctx, int mode, double t, uint64_t *ids, int cap, int *n) {
  if (!ctx || !n) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  std::vector<uint64_t> out;
  for (auto &kv : T->db) {
    const Track &tr = kv.second;
    if (tr.t.empty()) continue;
    if (mode == 0 ? !(tr.t.back() >= t) : (tr.t.front() < t)) out.push_back(kv.first);
  }
  std::sort(out.begin(), out.end());
  *n = (int)out.size();
  if (*n > cap) return PLV_E_CAPACITY;
  if (ids) std::copy(out.begin(), out.end(), ids);
  return PLV_OK;
}

// CSR export of the chosen tracks (the observation arrays of plv_tracks); missing ids get 0 observations
int plv_db_export_tracks(plv_ctx *ctx, const uint64_t *ids, int n, int *obs_ptr, double *obs_time, float *obs_uv,
                         float *obs_uvn, int cap_obs) {
  if (!ctx || !ids || !obs_ptr) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  int at = 0;
  obs_ptr[0] = 0;
  for (int f = 0; f < n; ++f) {
    auto it = T->db.find(ids[f]);
    if (it != T->db.end()) {
      const Track &tr = it->second;
      const int m = (int)tr.t.size();
      if (at + m > cap_obs) return PLV_E_CAPACITY;
      for (int i = 0; i < m; ++i) {
        if (obs_time) obs_time[at + i] = tr.t[i];
        if (obs_uv) obs_uv[2 * (at + i)] = tr.uv[2 * i], obs_uv[2 * (at + i) + 1] = tr.uv[2 * i + 1];
        if (obs_uvn) obs_uvn[2 * (at + i)] = tr.uvn[2 * i], obs_uvn[2 * (at + i) + 1] = tr.uvn[2 * i + 1];
      }
      at += m;
    }
    obs_ptr[f + 1] = at;
  }
  return PLV_OK;
}

// FeatureDatabase::cleanup_measurements(t): drop observations older than t (REF: FeatureDatabase.cpp:286-323)
int plv_db_cleanup_measurements(plv_ctx *ctx, double t) {
  if (!ctx) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  for (auto it = T->db.begin(); it != T->db.end();) {
    Track &tr = it->second;
    size_t keep = 0;
    for (size_t i = 0; i < tr.t.size(); ++i)
      if (!(tr.t[i] < t)) {
        tr.t[keep] = tr.t[i];
        tr.uv[2 * keep] = tr.uv[2 * i], tr.uv[2 * keep + 1] = tr.uv[2 * i + 1];
        tr.uvn[2 * keep] = tr.uvn[2 * i], tr.uvn[2 * keep + 1] = tr.uvn[2 * i + 1];
        ++keep;
      }
    tr.t.resize(keep);
    tr.uv.resize(2 * keep);
    tr.uvn.resize(2 * keep);
    if (keep == 0)
      it = T->db.erase(it);
    else
      ++it;
  }
  return PLV_OK;
}

// remove features (used ones are deleted after an update: REF CamHelper::cleanup_features, to_delete + cleanup())
int plv_db_remove(plv_ctx *ctx, const uint64_t *ids, int n) {
  if (!ctx || (n > 0 && !ids)) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  for (int i = 0; i < n; ++i) T->db.erase(ids[i]);
  return PLV_OK;
}

int plv_db_append_measurements(plv_ctx *ctx, uint64_t id, int n, const double *t, const float *uv, const float *uvn) {
  if (!ctx || n < 0 || (n > 0 && (!t || !uv || !uvn))) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  Track &tr = T->db[id];
  tr.t.insert(tr.t.end(), t, t + n);
  tr.uv.insert(tr.uv.end(), uv, uv + 2 * (size_t)n);
  tr.uvn.insert(tr.uvn.end(), uvn, uvn + 2 * (size_t)n);
  return PLV_OK;
}

// State::bounding_times + bounding_poses_n (order 3): is there an interpolation window for time t?
// REF: PL-VIWO/src/state/State.cpp:1023-1136 (same test as the kernels' bounding_start)
static bool has_bounding_poses(const plv_state_view &st, double t) {
  const int N = st.n_clones;
  if (N < 4) return false;
  const double *ct = st.clone_time;
  if (t < ct[0] - st.dt_exp || t > ct[N - 1] + st.dt_exp) return false;
  if (t > ct[N - 1]) return false;
  for (int i = 0; i < N - 1; ++i)
    if (ct[i] - st.dt_exp <= t && t <= ct[i + 1] + st.dt_exp) return true;
  return false;
}


int plv_camera_update_points(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt, double *dx,
                             plv_update_result *res, uint64_t *msckf_ids, uint8_t *accepted_out, double *p_out) {
  if (!ctx || !st || !opt || !dx || !res || st->n_clones < 2 || opt->max_msckf < 1 || opt->max_obs < 2) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  *res = plv_update_result{0, 0, 0, 0, 0, PLV_OK, 0, 0, 0};
  plv::NsScope ns_points(plv::counters().points_ns);
  plv::HostPhase ph_all("update_points: whole call");
  plv::HostPhase ph_pool("update_points: pool + staging");
  plv::RoctxRange rx_get("[Time-Cam] get features");
  const double dt = st->cam_dt;
  auto has_bounding = bounding_memo([st](double tq) { return has_bounding_poses(*st, tq); });
  const double t_oldest = st->clone_time[0], t_oldest2 = st->clone_time[1];  // no keyframes on this path
  struct Cand {
    uint64_t id;
    Track tr;
  };
  std::vector<Cand> pool;
  std::unordered_map<uint64_t, Track> unused;  // db_unused: goes back to the database at the end
  auto give_back = [&](uint64_t id, double t, const float *uv, const float *uvn) {
    Track &u = unused[id];
    u.t.push_back(t);
    u.uv.insert(u.uv.end(), uv, uv + 2);
    u.uvn.insert(u.uvn.end(), uvn, uvn + 2);
  };
  T->last_slam.clear();
  T->last_init.clear();
  if (opt->n_slam > 0 && !opt->slam_ids) return PLV_E_BADARG;
  auto is_slam = [&](uint64_t id) {
    for (int i = 0; i < opt->n_slam; ++i)
      if (opt->slam_ids[i] == id) return true;
    return false;
  };
  {
    // REF CamHelper.cpp:621-628 — landmarks of the state with a live track.  get_feature(id) is called without
    // `remove`, so the Feature object stays in the database (and may enter the pool below as well);
    // remove_unusable_measurements then edits that shared object: too-new observations are parked in db_unused,
    // too-old ones are dropped, a landmark may keep a single observation (:766-769).
    std::lock_guard<std::mutex> lk(T->mtx);
    for (int i = 0; i < opt->n_slam; ++i) {
      auto it = T->db.find(opt->slam_ids[i]);
      if (it == T->db.end()) continue;
      Track &tr = it->second;
      size_t keep = 0;
      for (size_t q = 0; q < tr.t.size(); ++q) {
        const double tm = tr.t[q] + dt;
        if (tm > opt->state_time + st->dt_exp) {
          give_back(it->first, tr.t[q], &tr.uv[2 * q], &tr.uvn[2 * q]);
          continue;
        }
        if (tm < t_oldest - st->dt_exp) continue;
        tr.t[keep] = tr.t[q];
        tr.uv[2 * keep] = tr.uv[2 * q], tr.uv[2 * keep + 1] = tr.uv[2 * q + 1];
        tr.uvn[2 * keep] = tr.uvn[2 * q], tr.uvn[2 * keep + 1] = tr.uvn[2 * q + 1];
        ++keep;
      }
      tr.t.resize(keep);
      tr.uv.resize(2 * keep);
      tr.uvn.resize(2 * keep);
      if (keep == 0) continue;
      // slam_update's own get_imu_poses (UpdaterCamera.cpp:303-304) leaves out what has no bounding clones
      Tracker::Listed e{it->first, Track{}, {0, 0, 0}};
      for (size_t q = 0; q < keep; ++q) {
        if (!has_bounding(tr.t[q] + dt)) continue;
        e.tr.t.push_back(tr.t[q]);
        e.tr.uv.insert(e.tr.uv.end(), &tr.uv[2 * q], &tr.uv[2 * q] + 2);
        e.tr.uvn.insert(e.tr.uvn.end(), &tr.uvn[2 * q], &tr.uvn[2 * q] + 2);
      }
      if (!e.tr.t.empty()) T->last_slam.push_back(std::move(e));
    }
  }
  res->n_slam = (int)T->last_slam.size();
  {
    std::lock_guard<std::mutex> lk(T->mtx);
    // REF CamHelper.cpp:631-637 — features_containing_older(oldest_2nd_clone_time), then
    // features_not_containing_newer(t_hist[size-2]); both take the feature out of the database
    std::vector<uint64_t> take;
    for (const auto &kv : T->db) {
      bool older = false, newer = false;
      for (double t : kv.second.t) {
        older = older || t < t_oldest2 - dt;
        newer = newer || t > opt->t_prev_frame - dt;
      }
      if (older || !newer) take.push_back(kv.first);
    }
    std::sort(take.begin(), take.end());
    pool.reserve(take.size());
    for (uint64_t id : take) {
      auto node = T->db.extract(id);  // (one lookup: the track leaves the database with its node)
      pool.push_back(Cand{id, std::move(node.mapped())});
    }
  }
  res->n_pool = (int)pool.size();
  // REF :740-775 remove_unusable_measurements, and get_imu_poses' bounding-pose test (:327-372)
  for (auto it = pool.begin(); it != pool.end();) {
    Track &tr = it->tr;
    size_t keep = 0;
    for (size_t i = 0; i < tr.t.size(); ++i) {
      const double tm = tr.t[i] + dt;
      if (tm > opt->state_time + st->dt_exp) {
        give_back(it->id, tr.t[i], &tr.uv[2 * i], &tr.uvn[2 * i]);  // newer than the window: later
        continue;
      }
      if (tm < t_oldest - st->dt_exp) continue;  // older than the window: discarded
      tr.t[keep] = tr.t[i];
      tr.uv[2 * keep] = tr.uv[2 * i], tr.uv[2 * keep + 1] = tr.uv[2 * i + 1];
      tr.uvn[2 * keep] = tr.uvn[2 * i], tr.uvn[2 * keep + 1] = tr.uvn[2 * i + 1];
      ++keep;
    }
    tr.t.resize(keep);
    tr.uv.resize(2 * keep);
    tr.uvn.resize(2 * keep);
    if (keep == 1 && is_slam(it->id)) {  // REF :766-769 kept, then handed back by the n_meas < 2 test (:656-661)
      give_back(it->id, tr.t[0], &tr.uv[0], &tr.uvn[0]);
      it = pool.erase(it);
    } else if (keep < 2) {
      it = pool.erase(it);  // REF :766-771 a single measurement is dropped
    } else {
      ++it;
    }
  }
  // REF :640 sort(feats_pool, feat_sort): long tracks first
  std::stable_sort(pool.begin(), pool.end(), [](const Cand &a, const Cand &b) { return a.tr.t.size() > b.tr.t.size(); });
  auto finish = [&](int rc) {
    res->n_returned = (int)unused.size();
    const bool window_full = opt->window_full != 0;
    auto hand_back = [ctx, T, window_full, t_oldest](std::unordered_map<uint64_t, Track> &un) {
      {
        std::lock_guard<std::mutex> lk(T->mtx);
        for (auto &kv : un) {  // REF :702-703 / :727-729 append_new_measurements
          Track &d = T->db[kv.first];
          if (d.t.empty()) {
            d = std::move(kv.second);
            continue;
          }
          d.t.insert(d.t.end(), kv.second.t.begin(), kv.second.t.end());
          d.uv.insert(d.uv.end(), kv.second.uv.begin(), kv.second.uv.end());
          d.uvn.insert(d.uvn.end(), kv.second.uvn.begin(), kv.second.uvn.end());
        }
      }
      // REF CamHelper.cpp:733-737: cleanup_features runs on every try_update, whether or not anything was updated
      if (window_full) (void)plv_db_cleanup_measurements(ctx, t_oldest);
    };
    if (T->defer_db) {  // plv_camera_try_update: nothing before the line update's submission reads the point database
      auto held = std::make_shared<std::unordered_map<uint64_t, Track>>(std::move(unused));
      T->deferred_db = [hand_back, held]() { hand_back(*held); };
    } else {
      hand_back(unused);
    }
    return rc;
  };
  auto give_back_all = [&](Cand &c) {
    if (unused.find(c.id) == unused.end()) {  // nothing of this feature went back earlier: hand the track over as it is
      unused.emplace(c.id, std::move(c.tr));
      c.tr = Track{};
      return;
    }
    for (size_t i = 0; i < c.tr.t.size(); ++i) give_back(c.id, c.tr.t[i], &c.tr.uv[2 * i], &c.tr.uvn[2 * i]);
  };
  // ---- this frame's update may be on the stream already (Tracker::Spec: enqueued behind the flow by the feed, its pool decided on the
  // device).  The pool above is the same one — formed from the database as always — and says which candidates' results to read.
  Tracker::Spec &S = T->spec;
  bool spec_done = false;
  std::vector<double> s_pf, s_err;
  std::vector<uint8_t> s_ok, s_acc;
  int s_rows = 0, s_status = PLV_OK;
  if (S.active) {
    S.active = false;
    ph_pool.stop();
    rx_get.stop();
    const int Fp0 = (int)pool.size();
    const bool same = S.t_prev_frame == opt->t_prev_frame && S.state_time == opt->state_time && S.dt == dt && S.n_clones == st->n_clones &&
                      S.t_oldest == t_oldest && S.t_oldest2 == t_oldest2 && S.max_msckf == opt->max_msckf && S.max_obs == opt->max_obs && !opt->cpi &&
                      opt->max_slam == 0 && opt->n_slam == 0;
    std::vector<int> cand_of(Fp0, -1);
    bool known = true;
    for (int f = 0; f < Fp0; ++f) {
      const auto it = S.index.find(pool[f].id);
      if (it == S.index.end()) known = false;
      else cand_of[f] = it->second;
    }
    T->chain_index.clear();
    // (the device works on a pool of up to spec_grid(max_msckf) tracks; whether the selection loop's cap would have cut it is known with
    //  the results — then the chained launch has seen the status and ended without touching anything, like behind a rejected update)
    T->chain_ok = same && known && Fp0 <= plv::spec_grid(opt->max_msckf);
    if (T->chain_ok && T->early_lines)
      for (int f = 0; f < Fp0; ++f) {
        int v = 0;
        for (double t : pool[f].tr.t) v += has_bounding(t + dt);
        if (v >= 2) T->chain_index.emplace(pool[f].id, cand_of[f]);  // (the chained line launch reads the candidates' triangulation results)
      }
    std::vector<double> cp(3 * (size_t)S.F), ce(S.F);
    std::vector<uint8_t> cok(S.F), cacc(S.F), cmem(S.F);
    int count = 0, over = 0, nrows = 0;
    int rc_s;
    {
      plv::RoctxRange rx_upd("[Time-Cam] MSCKF update");
      plv::HostPhase ph_dev("update_points: device submission + wait");
      rc_s = plv_points_update_collect(ctx, cp.data(), cok.data(), ce.data(), cacc.data(), &nrows, dx, start_detection_ahead, ctx, cmem.data(), &count, &over);
    }
    T->chain_ok = false;
    auto fail = [&](const char *why) {
      plv::set_last_error("speculative point update: %s (host pool %d, device pool %d%s)", why, Fp0, count, over ? ", over the cap" : "");
      for (Cand &c : pool) give_back_all(c);
      return finish(PLV_E_DEVICE);
    };
    if (!same) return fail("the batch was staged for another update than the one asked for");
    if (over || Fp0 > plv::spec_grid(opt->max_msckf)) {
      // over == 1: the pool exceeds the launch — the device left every candidate empty and updated nothing; over == 2: the pool exceeds
      // max_msckf and max_msckf of its candidates passed their tests — the selection loop would have stopped inside the pool
      // (REF CamHelper.cpp:651-653), the device committed nothing (ekf_commit_kernel).  The long way below.
      if (!(over && Fp0 > opt->max_msckf && count == Fp0 && (over == 1) == (Fp0 > plv::spec_grid(opt->max_msckf))))
        return fail("host and device disagree on the pool's size");
      if (plv::host_phases().on) plv::host_phases().add(over == 1 ? "speculative point update: pool larger than the launch, run again the long way (count)" : "speculative point update: pool cut by the selection cap, run again the long way (count)", 1.0);
      if (plv_camera_lines_job_pending(ctx)) plv_camera_lines_job_abort2(ctx, 1);  // (a chained line launch saw the status and did nothing; its pool stays formed)
      ++plv::counters().spec_over[over == 1 ? 2 : 1];
    } else {
      if (rc_s != PLV_OK && rc_s != PLV_E_NOT_PSD) {
        for (Cand &c : pool) give_back_all(c);
        return finish(rc_s);
      }
      if (!known || count != Fp0) return fail("host and device disagree on the pool");
      for (int f = 0; f < Fp0; ++f)
        if (!cmem[cand_of[f]]) return fail("a track of the host's pool is not in the device's");
      s_pf.resize(3 * (size_t)Fp0), s_err.resize(Fp0), s_ok.resize(Fp0), s_acc.resize(Fp0);
      for (int f = 0; f < Fp0; ++f) {
        const int c = cand_of[f];
        std::copy(cp.begin() + 3 * (size_t)c, cp.begin() + 3 * (size_t)c + 3, s_pf.begin() + 3 * (size_t)f);
        s_err[f] = ce[c], s_ok[f] = cok[c], s_acc[f] = cacc[c];
      }
      s_rows = nrows;
      s_status = rc_s == PLV_E_NOT_PSD ? rc_s : PLV_OK;
      if (rc_s == PLV_E_NOT_PSD) std::fill(dx, dx + ctx->cov_n, 0.0);  // EKFUpdate returned false: nothing changed
      spec_done = true;
      ++plv::counters().speculated;
      if (Fp0 > opt->max_msckf) ++plv::counters().spec_over[0];
    }
  }
  if (pool.empty()) {
    if (!spec_done) std::fill(dx, dx + ctx->cov_n, 0.0);
    return finish(PLV_OK);
  }
  // ---- triangulate the whole pool in one call (the reference goes feature by feature until the cap; a
  //      feature it never reaches keeps its observations either way)
  const int Fp = (int)pool.size();
  // ---- use_imu_res: poses from the CPI table; what it cannot serve leaves the track here (get_imu_poses, :356-365)
  std::vector<std::vector<double>> cpiR(opt->cpi ? Fp : 0), cpip(opt->cpi ? Fp : 0), cpiQ(opt->cpi ? Fp : 0);
  std::vector<std::vector<int>> cpiC(opt->cpi ? Fp : 0);
  const bool imu_cov = opt->cpi && opt->cpi->Q && st->use_imu_cov && !st->use_pol_cov;  // REF CamHelper.cpp:214-224
  if (opt->cpi) {
    std::vector<double> tq;
    for (const Cand &c : pool)
      for (double t : c.tr.t) tq.push_back(t + dt);
    std::vector<double> Rq(9 * tq.size()), pq(3 * tq.size());
    std::vector<uint8_t> okq(tq.size());
    int rc0 = plv_cpi_poses(ctx, st, opt->cpi, (int)tq.size(), tq.data(), Rq.data(), pq.data(), okq.data());
    std::vector<double> Qq(imu_cov ? 36 * tq.size() : 0);
    std::vector<int> Cq(imu_cov ? tq.size() : 0);
    if (rc0 == PLV_OK && imu_cov) {
      std::vector<uint8_t> okn(tq.size());
      rc0 = plv_cpi_noise(st, opt->cpi, (int)tq.size(), tq.data(), Qq.data(), Cq.data(), okn.data());
      for (size_t i = 0; i < tq.size(); ++i) okq[i] = okq[i] && okn[i];
    }
    if (rc0 != PLV_OK) {
      for (Cand &c : pool) give_back_all(c);
      return finish(rc0);
    }
    size_t o = 0;
    for (int f = 0; f < Fp; ++f) {
      Cand &c = pool[f];
      Track kept;
      for (size_t i = 0; i < c.tr.t.size(); ++i, ++o) {
        if (!okq[o]) {
          give_back(c.id, c.tr.t[i], &c.tr.uv[2 * i], &c.tr.uvn[2 * i]);
          continue;
        }
        kept.t.push_back(c.tr.t[i]);
        kept.uv.insert(kept.uv.end(), &c.tr.uv[2 * i], &c.tr.uv[2 * i] + 2);
        kept.uvn.insert(kept.uvn.end(), &c.tr.uvn[2 * i], &c.tr.uvn[2 * i] + 2);
        cpiR[f].insert(cpiR[f].end(), &Rq[9 * o], &Rq[9 * o] + 9);
        cpip[f].insert(cpip[f].end(), &pq[3 * o], &pq[3 * o] + 3);
        if (imu_cov) {
          cpiQ[f].insert(cpiQ[f].end(), &Qq[36 * o], &Qq[36 * o] + 36);
          cpiC[f].push_back(Cq[o]);
        }
      }
      c.tr = std::move(kept);
    }
  }
  std::vector<int> ptr(Fp + 1, 0), valid_n(Fp, 0);
  int most_valid = 0;
  for (int f = 0; f < Fp; ++f) {
    ptr[f + 1] = ptr[f] + (int)pool[f].tr.t.size();
    // get_imu_poses (:327-372): observations without bounding clones do not count (and go back to the database below)
    for (double t : pool[f].tr.t) valid_n[f] += has_bounding(t + dt);
    most_valid = std::max(most_valid, valid_n[f]);
  }
  const int nobs = ptr[Fp];
  if (nobs == 0) {
    if (!spec_done) std::fill(dx, dx + ctx->cov_n, 0.0);
    return finish(PLV_OK);
  }
  std::vector<double> ot(nobs), pf(3 * (size_t)Fp), err(Fp);
  std::vector<float> ouv(2 * (size_t)nobs), ouvn(2 * (size_t)nobs);
  std::vector<uint8_t> ok(Fp);
  for (int f = 0; f < Fp; ++f) {
    const Track &tr = pool[f].tr;
    std::copy(tr.t.begin(), tr.t.end(), ot.begin() + ptr[f]);
    std::copy(tr.uv.begin(), tr.uv.end(), ouv.begin() + 2 * (size_t)ptr[f]);
    std::copy(tr.uvn.begin(), tr.uvn.end(), ouvn.begin() + 2 * (size_t)ptr[f]);
  }
  plv_tracks all{};
  all.n_feat = Fp;
  all.obs_ptr = ptr.data();
  all.obs_time = ot.data();
  all.obs_uv = ouv.data();
  all.obs_uvn = ouvn.data();
  std::vector<double> allR, allp;
  if (opt->cpi) {
    for (int f = 0; f < Fp; ++f) {
      allR.insert(allR.end(), cpiR[f].begin(), cpiR[f].end());
      allp.insert(allp.end(), cpip[f].begin(), cpip[f].end());
    }
    all.res_R = allR.data();
    all.res_p = allp.data();
  }
  std::vector<int> cols(ctx->cfg.max_state_dim > 0 ? ctx->cfg.max_state_dim : 1024);
  int k = 0, n_rows = 0, rc = PLV_OK;
  // ---- one submission (the default): every pool candidate is triangulated, the selection loop below runs on the device too
  // (jacobian_nullspace_kernel: candidate_selected), Jacobians / gate / compression / EKFUpdate follow on the stream, and the host
  // reads triangulation results, gate decisions and dx after ONE synchronisation.  The batch then holds all candidates (the ones
  // the loop does not take are empty systems) and its column set is the union over all of them: a permutation / zero columns of the
  // reference's, which changes neither dx nor P beyond rounding.  Poses from the CPI table, in-state landmarks and tracks longer
  // than the batch rows take the two-step route (triangulate, select on the host, then build + update).
  const bool fused = !opt->cpi && opt->max_slam == 0 && most_valid <= opt->max_obs;
  ph_pool.stop();
  rx_get.stop();
  plv::frame_mark("@ update_points: pool + staging done");
  plv::RoctxRange rx_upd("[Time-Cam] MSCKF update");
  plv::HostPhase ph_dev("update_points: device submission + wait");
  std::vector<uint8_t> acc_all(Fp, 0);
  bool fused_ran = false;
  if (fused && spec_done) {  // the update ran behind the flow: its results, candidate by candidate
    std::copy(s_pf.begin(), s_pf.end(), pf.begin());
    std::copy(s_err.begin(), s_err.end(), err.begin());
    std::copy(s_ok.begin(), s_ok.end(), ok.begin());
    std::copy(s_acc.begin(), s_acc.end(), acc_all.begin());
    n_rows = s_rows;
    res->status = s_status;
    fused_ran = true;
  } else if (fused) {
    std::vector<uint8_t> flags(Fp);
    bool any = false;
    for (int f = 0; f < Fp; ++f) any = (flags[f] = valid_n[f] >= 2) || any;
    if (any) {
      all.p_FinG = all.p_FinG_fej = pf.data();
      rc = plv_jacobian_columns(st, &all, cols.data(), (int)cols.size(), &k);
      if (rc == PLV_OK && k > 0) {
        ctx->gate_rows_hint = 2 * most_valid;
        T->chain_index.clear();
        T->chain_ok = Fp <= opt->max_msckf;  // (the selection loop cannot reach its cap: a candidate is taken on its own verdict)
        if (T->chain_ok && T->early_lines)
          for (int f = 0; f < Fp; ++f)
            if (flags[f]) T->chain_index.emplace(pool[f].id, f);
        plv::frame_mark("@ update_points: columns done, fused call");
        rc = plv_points_update_fused(ctx, st, &all, &opt->tri, flags.data(), opt->max_msckf, k, cols.data(), 2 * opt->max_obs,
                                     st->sigma_pix * st->sigma_pix, opt->chi2_mult, 3.0, pf.data(), ok.data(), err.data(), acc_all.data(),
                                     &n_rows, dx, start_detection_ahead, ctx);
        T->chain_ok = false;
        res->status = rc == PLV_E_NOT_PSD ? rc : PLV_OK;
        if (rc == PLV_E_NOT_PSD) {
          rc = PLV_OK;  // EKFUpdate returned false: nothing changed, the call itself succeeded
          std::fill(dx, dx + ctx->cov_n, 0.0);
        }
        fused_ran = rc == PLV_OK;
      }
      if (rc != PLV_OK) {
        for (Cand &c : pool) give_back_all(c);
        return finish(rc);
      }
    }
  } else {
    start_detection_ahead(ctx);
    rc = plv_triangulate(ctx, st, &all, &opt->tri, pf.data(), ok.data(), err.data());
    if (rc != PLV_OK) {
      for (Cand &c : pool) give_back_all(c);
      return finish(rc);
    }
  }
  ph_dev.stop();
  rx_upd.stop();
  if (ctx->decision_trace) {
    // the values behind the verdicts, for the comparison of two runs decision by decision (tests/decision_trace.py): what the host
    // already holds, and what the kernels left on the device (NaN: a test that was not reached, or a route that does not report)
    const double nan = std::numeric_limits<double>::quiet_NaN();
    T->dec_ids.resize(Fp);
    T->dec_vals.assign((size_t)Fp * PLV_DECISION_VALUES, nan);
    std::vector<double> tri(4 * (size_t)Fp, nan), gate(3 * (size_t)Fp, nan);
    if (fused_ran && ctx->dec_F == Fp) {
      PLV_HIP_CHECK(hipMemcpy(tri.data(), ctx->d_tri_dbg.p, tri.size() * 8, hipMemcpyDeviceToHost));
      if (ctx->dec_gate) PLV_HIP_CHECK(hipMemcpy(gate.data(), ctx->d_gate_dec.p, gate.size() * 8, hipMemcpyDeviceToHost));
    }
    ctx->dec_F = 0, ctx->dec_gate = false;
    for (int f = 0; f < Fp; ++f) {
      double *v = &T->dec_vals[(size_t)f * PLV_DECISION_VALUES];
      T->dec_ids[f] = pool[f].id;
      v[0] = valid_n[f], v[1] = ok[f], v[2] = ok[f] ? err[f] : nan, v[3] = fused_ran ? acc_all[f] : nan;
      for (int i = 0; i < 4; ++i) v[4 + i] = tri[4 * (size_t)f + i];
      for (int i = 0; i < 3; ++i) v[8 + i] = gate[3 * (size_t)f + i];
    }
  }
  plv::RoctxRange rx_db("[Time-Cam] DB clan up");  // (sic, UpdaterCamera.cpp:174)
  plv::HostPhase ph_post("update_points: selection + database");
  // ---- REF :648-699 the selection loop
  std::vector<int> sel;
  std::vector<int> n_skip(Fp, 0);  // usable observations a truncated track leaves out (its oldest)
  for (int f = 0; f < Fp; ++f) {
    Cand &c = pool[f];
    if ((int)sel.size() >= opt->max_msckf) {  // :651-653 break; the rest returns to the database (:702)
      give_back_all(c);
      continue;
    }
    const int valid = valid_n[f];
    if (valid >= 2 && ok[f]) {  // copy_to_db(db_used, feat): dynamic (:677) and MSCKF (:697) features, Triangulated = true
      std::lock_guard<std::mutex> lk(T->mtx);
      Tracker::UsedPoint &u = T->used[c.id];
      std::copy(pf.begin() + 3 * (size_t)f, pf.begin() + 3 * (size_t)f + 3, u.p);
      u.newest = c.tr.t.back();
    }
    if (valid < 2 || !ok[f] || !(err[f] < 3.0)) {  // :656-683
      give_back_all(c);
      continue;
    }
    // :685-693 a long track becomes a new in-state landmark while there is room
    if (valid >= opt->init_min_meas && opt->n_slam + (int)T->last_init.size() < opt->max_slam) {
      Tracker::Listed e{c.id, Track{}, {pf[3 * (size_t)f], pf[3 * (size_t)f + 1], pf[3 * (size_t)f + 2]}};
      for (size_t i = 0; i < c.tr.t.size(); ++i) {
        if (!has_bounding(c.tr.t[i] + dt)) {
          give_back(c.id, c.tr.t[i], &c.tr.uv[2 * i], &c.tr.uvn[2 * i]);
          continue;
        }
        e.tr.t.push_back(c.tr.t[i]);
        e.tr.uv.insert(e.tr.uv.end(), &c.tr.uv[2 * i], &c.tr.uv[2 * i] + 2);
        e.tr.uvn.insert(e.tr.uvn.end(), &c.tr.uvn[2 * i], &c.tr.uvn[2 * i] + 2);
      }
      T->last_init.push_back(std::move(e));
      continue;
    }
    // batch capacity of the MSCKF update (the reference has none): the newest max_obs observations are used, the older ones are
    // consumed with the feature; counted in res->n_truncated
    if (valid > opt->max_obs) {
      n_skip[f] = valid - opt->max_obs;  // the first (oldest) usable observations are left out
      ++res->n_truncated;
    }
    sel.push_back(f);
  }
  res->n_msckf = (int)sel.size();
  res->n_init = (int)T->last_init.size();
  if (plv::host_phases().on && Fp > opt->max_msckf) fprintf(stderr, "[plv over] pool %d selected %d\n", Fp, (int)sel.size());
  if (sel.empty()) {
    if (!fused_ran) std::fill(dx, dx + ctx->cov_n, 0.0);
    return finish(PLV_OK);
  }
  // ---- UpdaterCamera::msckf_update on the selected features
  const int F = (int)sel.size();
  std::vector<int> sptr(F + 1, 0);
  std::vector<double> st_t, sp(3 * (size_t)F), selR, selp, selQ;
  std::vector<int> selC;
  std::vector<float> suv;
  for (int q = 0; q < F; ++q) {
    const Cand &c = pool[sel[q]];
    int seen = 0;
    for (size_t i = 0; i < c.tr.t.size(); ++i) {
      if (!has_bounding(c.tr.t[i] + dt)) {
        give_back(c.id, c.tr.t[i], &c.tr.uv[2 * i], &c.tr.uvn[2 * i]);
        continue;
      }
      if (seen++ < n_skip[sel[q]]) continue;
      st_t.push_back(c.tr.t[i]);
      suv.push_back(c.tr.uv[2 * i]);
      suv.push_back(c.tr.uv[2 * i + 1]);
      if (opt->cpi) {
        selR.insert(selR.end(), &cpiR[sel[q]][9 * i], &cpiR[sel[q]][9 * i] + 9);
        selp.insert(selp.end(), &cpip[sel[q]][3 * i], &cpip[sel[q]][3 * i] + 3);
        if (imu_cov) {
          selQ.insert(selQ.end(), &cpiQ[sel[q]][36 * i], &cpiQ[sel[q]][36 * i] + 36);
          selC.push_back(cpiC[sel[q]][i]);
        }
      }
    }
    sptr[q + 1] = (int)st_t.size();
    std::copy(pf.begin() + 3 * (size_t)sel[q], pf.begin() + 3 * (size_t)sel[q] + 3, sp.begin() + 3 * (size_t)q);
    if (msckf_ids) msckf_ids[q] = c.id;
  }
  if (p_out) std::copy(sp.begin(), sp.end(), p_out);
  std::vector<uint8_t> acc(F, 0);
  if (fused_ran) {
    for (int q = 0; q < F; ++q) acc[q] = acc_all[sel[q]];
  } else {
    plv_tracks tr{};
    tr.n_feat = F;
    tr.obs_ptr = sptr.data();
    tr.obs_time = st_t.data();
    tr.obs_uv = suv.data();
    tr.p_FinG = sp.data();
    tr.p_FinG_fej = sp.data();  // MSCKF features: FEJ value = estimate (REF CamHelper.cpp:556-557)
    if (opt->cpi) {
      tr.res_R = selR.data();
      tr.res_p = selp.data();
      if (imu_cov) {
        tr.res_Q = selQ.data();
        tr.res_clone = selC.data();
      }
    }
    rc = plv_jacobian_columns(st, &tr, cols.data(), (int)cols.size(), &k);
    if (rc == PLV_OK) rc = plv_build_jacobians_resident(ctx, st, &tr, k, cols.data(), 2 * opt->max_obs);
    if (rc == PLV_OK) {
      rc = plv_msckf_update_resident(ctx, st->sigma_pix * st->sigma_pix, opt->chi2_mult, 3.0, acc.data(), &n_rows, dx);
      res->status = rc;
      if (rc == PLV_E_NOT_PSD) rc = PLV_OK;  // EKFUpdate returned false: nothing changed, the call itself succeeded
    }
    if (rc != PLV_OK) {
      for (int q = 0; q < F; ++q) give_back_all(pool[sel[q]]);
      return finish(rc);
    }
  }
  res->n_rows = n_rows;
  for (int q = 0; q < F; ++q) {
    res->n_accepted += acc[q];
    if (accepted_out) accepted_out[q] = acc[q];
    if (!acc[q]) {  // REF UpdaterCamera.cpp:266-268: only gate failures go back; what EKFUpdate then rejects is consumed all the same
      const Cand &c = pool[sel[q]];
      for (size_t i = 0; i < c.tr.t.size(); ++i)
        if (has_bounding(c.tr.t[i] + dt)) give_back(c.id, c.tr.t[i], &c.tr.uv[2 * i], &c.tr.uvn[2 * i]);
    }
  }
  return finish(PLV_OK);
}

int plv_camera_update_list(plv_ctx *ctx, int which, int cap_feat, int cap_obs, int *n_feat, uint64_t *ids, int *obs_ptr,
                           double *obs_time, float *obs_uv, float *obs_uvn, double *p_FinG) {
  if (!ctx || !n_feat || (which != PLV_LIST_SLAM && which != PLV_LIST_INIT)) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  const std::vector<Tracker::Listed> &L = which == PLV_LIST_SLAM ? T->last_slam : T->last_init;
  *n_feat = (int)L.size();
  size_t nobs = 0;
  for (const auto &e : L) nobs += e.tr.t.size();
  if (!ids && !obs_ptr) return PLV_OK;  // size query
  if ((int)L.size() > cap_feat || (int)nobs > cap_obs || !ids || !obs_ptr || !obs_time || !obs_uv) return PLV_E_BADARG;
  int o = 0;
  obs_ptr[0] = 0;
  for (size_t f = 0; f < L.size(); ++f) {
    const Tracker::Listed &e = L[f];
    ids[f] = e.id;
    std::copy(e.tr.t.begin(), e.tr.t.end(), obs_time + o);
    std::copy(e.tr.uv.begin(), e.tr.uv.end(), obs_uv + 2 * (size_t)o);
    if (obs_uvn) std::copy(e.tr.uvn.begin(), e.tr.uvn.end(), obs_uvn + 2 * (size_t)o);
    if (p_FinG) std::copy(e.p, e.p + 3, p_FinG + 3 * f);
    o += (int)e.tr.t.size();
    obs_ptr[f + 1] = o;
  }
  return PLV_OK;
}

int plv_slam_marg_flags(plv_ctx *ctx, int n_slam, const uint64_t *slam_ids, const int *update_fail_count, uint8_t *should_marg) {
  if (!ctx || n_slam < 0 || (n_slam > 0 && (!slam_ids || !should_marg))) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  for (int i = 0; i < n_slam; ++i) {
    const bool lost = T->db.find(slam_ids[i]) == T->db.end();  // REF UpdaterCamera.cpp:125-129 feat == nullptr
    should_marg[i] = (lost || (update_fail_count && update_fail_count[i] > 1)) ? 1 : 0;  // :130-131
  }
  return PLV_OK;
}

int plv_point_used_insert(plv_ctx *ctx, uint64_t id, const double *p_FinG, double newest_obs_time) {
  if (!ctx || !p_FinG) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  Tracker::UsedPoint &u = T->used[id];
  std::copy(p_FinG, p_FinG + 3, u.p);
  u.newest = newest_obs_time;
  return PLV_OK;
}

// line_api.hip: FeatureDatabase::get_feature(id) on point_used (Triangulated features only)
int plv_point_used_lookup(plv_ctx *ctx, uint64_t id, double *p) {
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  auto it = T->used.find(id);
  if (it == T->used.end()) return 0;
  std::copy(it->second.p, it->second.p + 3, p);
  return 1;
}
// The anchors of a pool of lines in one pass under one lock (the line update's first half looked every point of every line up through
// two locked calls: 12-19 us of the chained first half at configs[2]).  pt_ptr / pt_ids: the lines' related points (CSR, in the
// order LineHelper.cpp:233-247 walks them).  chained: the candidates for the launch to decide from (plv_ctx::chain: per point its
// index in the running point update's pool and what point_used holds now; the walk of a line ends at the first point point_used
// holds); else the first point point_used holds (anchor [Lp][3], has [Lp]).
void plv_point_anchor_fill(plv_ctx *ctx, int Lp, const int *pt_ptr, const int *pt_ids, int chained, double *anchor, uint8_t *has) {
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  plv_ctx::ChainState &ch = ctx->chain;
  if (chained) ch.anc_ptr.assign(1, 0), ch.anc_f.clear(), ch.anc_has_old.clear(), ch.anc_old.clear();
  for (int l = 0; l < Lp; ++l) {
    // (a line's list names its points once per observation — the same few ids fifteen times over: an id that was looked at gives the
    //  same answer again, and the first answer is the one that counts: two hash look-ups per DISTINCT id, 14 -> ~5 us per frame)
    uint64_t seen[16];
    int n_seen = 0;
    for (int q = pt_ptr[l]; q < pt_ptr[l + 1]; ++q) {
      const uint64_t id = (uint64_t)pt_ids[q];
      bool dup = false;
      for (int z = 0; z < n_seen && !dup; ++z) dup = seen[z] == id;
      if (dup) continue;
      if (n_seen < 16) seen[n_seen++] = id;
      const auto it = T->used.find(id);
      const bool has_old = it != T->used.end();
      if (!chained) {
        if (!has_old) continue;
        std::copy(it->second.p, it->second.p + 3, anchor + 3 * (size_t)l);
        has[l] = 1;
        break;
      }
      const auto ci = T->chain_index.find(id);
      const int pf = ci == T->chain_index.end() ? -1 : ci->second;
      if (pf < 0 && !has_old) continue;
      static const double zero[3] = {0, 0, 0};
      const double *old = has_old ? it->second.p : zero;
      ch.anc_f.push_back(pf);
      ch.anc_has_old.push_back(has_old ? 1 : 0);
      ch.anc_old.insert(ch.anc_old.end(), old, old + 3);
      if (has_old) break;  // (point_used holds this point whatever the update does to it: the search ends here either way)
    }
    if (chained) ch.anc_ptr.push_back((int)ch.anc_f.size());
  }
}
// point_used->cleanup_measurements(oldest_clone_time)   REF: UpdaterCamera.cpp:186-188
void plv_point_used_cleanup(plv_ctx *ctx, double t_oldest) {
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  for (auto it = T->used.begin(); it != T->used.end();)
    it = it->second.newest < t_oldest ? T->used.erase(it) : std::next(it);
}

// the database hand-back a point update left behind (plv_camera_try_update); runs inside the line update's wait
void plv_tracker_run_deferred(void *arg) {
  plv_ctx *ctx = (plv_ctx *)arg;
  Tracker *T = trk(ctx);
  if (!T->deferred_db) return;
  std::function<void()> f;
  f.swap(T->deferred_db);
  f();
}

int plv_camera_try_update(plv_ctx *ctx, const plv_state_view *st, plv_try_update *io) {
  if (!ctx || !st || !io || !io->opt_points || !io->dx_points || !io->res_points || (io->n_var > 0 && !io->vars)) return PLV_E_BADARG;
  if (io->opt_lines && (!io->dx_lines || !io->res_lines)) return PLV_E_BADARG;
  if (io->opt_points->max_slam > 0) {  // the in-state landmark updates sit between the two halves and are the caller's
    plv::set_last_error("plv_camera_try_update: max_slam > 0 needs the two-call form (plv_slam_update in between)");
    return PLV_E_BADARG;
  }
  Tracker *T = trk(ctx);
  const int n = ctx->cov_n;
  auto apply = [&](const plv_update_result &r, const double *dx) {  // StateHelper::EKFUpdate's mean update (:156-168)
    if (r.status != PLV_OK || r.n_accepted < 1 || io->n_var < 1) return (int)PLV_OK;
    TRY(plv_state_boxplus(io->n_var, io->vars, dx, n));
    if (st->intrinsic_state_id >= 0) TRY(plv_set_camera_intrinsics(ctx, st->intrinsics));
    return (int)PLV_OK;
  };
  // the next frame's top-up detection runs on the side stream next to the point update.  (Round 2 placed it on the ctx stream behind
  // the update when a line update follows; since the line pool is formed inside the point update's wait, the line update is submitted
  // right after that wait and would queue behind the detection: PLV_KNOB_AHEAD_CTX restores that placement for measurements.)
  T->defer_db = io->opt_lines != nullptr;
  T->ahead_on_ctx_stream = io->opt_lines != nullptr && plv::knob(plv::PLV_KNOB_AHEAD_CTX);
  T->early_st = io->opt_lines ? st : nullptr;
  T->early_lines = io->opt_lines;
  T->early_cap = io->line_cap;
  T->chain_ok = false;
  // what a line launch chained behind the point update needs to form x (+) dx itself: the quaternions behind the view's rotation
  // matrices and the covariance indices, from the caller's variable list (poses are PoseJPL: orientation at id, position at id + 3)
  plv_ctx::ChainState &ch = ctx->chain;
  ch.ready = false;
  if (io->opt_lines && io->n_var > 0 && st->dt_state_id < 0 && !io->opt_lines->cpi) {
    const int N = st->n_clones;
    auto find = [&](int kind, int id, int size) -> const plv_state_var * {
      for (int i = 0; i < io->n_var; ++i)
        if (io->vars[i].kind == kind && io->vars[i].id == id && (kind == PLV_VAR_QUAT || io->vars[i].size == size)) return &io->vars[i];
      return nullptr;
    };
    bool okc = true;
    ch.q.assign(4 * (size_t)N, 0.0), ch.ids.assign(N + 3, -1);
    for (int i = 0; i < N && okc; ++i) {
      const int sid = st->clone_state_id[i];
      const plv_state_var *vq = find(PLV_VAR_QUAT, sid, 3), *vp = find(PLV_VAR_VEC, sid + 3, 3);
      // (the view must be what plv_state_boxplus keeps current: the variable's own array or its mirror)
      const double *vR = st->clone_R + 9 * (size_t)i, *vP = st->clone_p + 3 * (size_t)i;
      okc = sid >= 0 && vq && vp && (vq->out == vR || vq->mirror == vR) && (vp->val == vP || vp->mirror == vP);
      if (okc) std::copy(vq->val, vq->val + 4, ch.q.begin() + 4 * (size_t)i), ch.ids[i] = sid;
    }
    if (okc && st->extrinsic_state_id >= 0) {
      const plv_state_var *vq = find(PLV_VAR_QUAT, st->extrinsic_state_id, 3), *vp = find(PLV_VAR_VEC, st->extrinsic_state_id + 3, 3);
      okc = vq && vp;
      if (okc) std::copy(vq->val, vq->val + 4, ch.qe), ch.ids[N] = st->extrinsic_state_id;
    }
    if (okc && st->intrinsic_state_id >= 0) {
      okc = find(PLV_VAR_VEC, st->intrinsic_state_id, 8) != nullptr;
      ch.ids[N + 1] = st->intrinsic_state_id;
    }
    ch.ready = okc;
  }
  plv::frame_mark("@ try_update: chain state ready");
  ctx->wait_poll = io->opt_lines ? poll_line_pool : nullptr;
  ctx->wait_poll_arg = ctx;
  int rc = plv_camera_update_points(ctx, st, io->opt_points, io->dx_points, io->res_points, io->msckf_ids, io->msckf_accepted, io->p_FinG);
  ctx->wait_poll = nullptr;
  ch.ready = false;
  T->defer_db = T->ahead_on_ctx_stream = false;
  T->early_st = nullptr, T->early_lines = nullptr;
  bool chained = io->opt_lines && plv_camera_lines_job_pending(ctx);
  if (chained && (plv_update_state(ctx)->last_route >= 5 || io->res_points->status != PLV_OK)) {
    // the point update came back rejected — and was perhaps run again on the host's verdict (update_state.hpp, RedoW): the chained
    // line launch saw the rejection and ended without touching anything (JacParams::chain_status).  The line half goes the unchained way.
    if (plv_update_state(ctx)->last_route >= 5) ++plv::counters().route[6];  // (plv_route_counts[6]: re-runs next to a chained line launch)
    plv_camera_lines_job_abort2(ctx, rc == PLV_OK ? 1 : 0);  // (the frame goes on: its line pool stays formed)
    chained = false;
  }
  // REF UpdaterCamera.cpp:148-152: get_line_features runs between get_features and msckf_update — the line pool is triangulated on
  // the state as it is before the point update's correction is applied (chained: its launch is already on the stream, staged from st)
  if (rc == PLV_OK && io->opt_lines && !chained) rc = plv_camera_get_line_features(ctx, st);
  plv::frame_mark("@ update_points returned");
  if (rc == PLV_OK) rc = apply(*io->res_points, io->dx_points);
  if (rc == PLV_OK && io->opt_lines) {
    rc = plv_line_tracker_feed_wait(ctx);
    io->line_db_size = plv_line_db_size_after_feed(ctx);
    if (rc == PLV_OK) {
      plv_line_defer_finish(ctx, 1);
      rc = plv_camera_update_lines(ctx, st, io->opt_lines, io->dx_lines, io->res_lines, io->line_ids, io->line_accepted, io->line_FinG,
                                   io->line_cap);
      plv_line_defer_finish(ctx, 0);
      plv::frame_mark("@ update_lines returned");
    }
    plv_tracker_run_deferred(ctx);
    if (rc == PLV_OK) rc = apply(*io->res_lines, io->dx_lines);
    plv::frame_mark("@ line dx applied");
  }
  if (rc != PLV_OK && io->opt_lines) {  // (every failing exit — a pool formed ahead of time or a chained launch must not outlive the call)
    if (plv_camera_lines_job_pending(ctx)) plv_camera_lines_job_abort(ctx);
    plv_line_pool_discard(ctx);
  }
  plv_tracker_run_deferred(ctx);
  return rc;
}

// UpdaterCamera::feed_measurement followed by try_update (REF: UpdaterCamera.cpp:77-116, 139-195), one call per camera frame.
int plv_camera_frame(plv_ctx *ctx, const plv_state_view *st, plv_camera_frame_io *io) {
  if (!ctx || !st || !io || (io->slot < 0 && !io->img)) return PLV_E_BADARG;
  plv::NsScope ns(plv::counters().frame_ns);
  struct RuScope {  // PLV_HOST_TIMING: page faults and system time of the caller's thread inside the frame
    struct rusage a;
    bool on = plv::host_phases().on;
    RuScope() {
      if (on) getrusage(RUSAGE_THREAD, &a);
    }
    ~RuScope() {
      if (!on) return;
      struct rusage b;
      getrusage(RUSAGE_THREAD, &b);
      plv::host_phases().add("frame: minor page faults of the caller's thread (count)", (double)(b.ru_minflt - a.ru_minflt));
      plv::host_phases().add("frame: system time of the caller's thread", (double)((b.ru_stime.tv_sec - a.ru_stime.tv_sec) * 1000000L + (b.ru_stime.tv_usec - a.ru_stime.tv_usec)));
      plv::host_phases().add("frame: involuntary context switches (count)", (double)(b.ru_nivcsw - a.ru_nivcsw));
      plv::host_phases().add("frame: voluntary context switches (count)", (double)(b.ru_nvcsw - a.ru_nvcsw));
    }
  } ru_scope;
  if (plv::host_phases().on)
    plv::frame_t0_ns().store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
  plv::RoctxRange rx_feed("[Time-Cam] feed measurement");
  // (round 6) with an update to follow, the feed enqueues that update behind the frame's flow (Tracker::Spec); whatever happens
  // between here and plv_camera_update_points, a batch that was enqueued is collected before the call returns (SpecGuard)
  Tracker *Tf = trk(ctx);
  struct SpecGuard {
    plv_ctx *c;
    Tracker *T;
    ~SpecGuard() {
      T->spec_st = nullptr, T->spec_opt = nullptr;
      if (!T->spec.active) return;
      T->spec.active = false;  // (an early exit: the update ran, nobody reads it; the covariance on the device is what it left)
      const int F = T->spec.F, n = c->cov_n;
      std::vector<double> p(3 * (size_t)F), e(F), dx((size_t)std::max(n, 1));
      std::vector<uint8_t> ok(F), acc(F);
      int rows = 0;
      (void)plv_points_update_collect(c, p.data(), ok.data(), e.data(), acc.data(), &rows, dx.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
    }
  } spec_guard{ctx, Tf};
  Tf->spec.active = false;
  if (io->update && io->update->opt_points && io->update->opt_points->max_slam == 0) Tf->spec_st = st, Tf->spec_opt = io->update->opt_points;
  const bool lines = io->use_lines != 0;
  // with lines and an update to follow, the line tracker's host logic runs on the worker thread next to the point update (joined inside
  // plv_camera_try_update) and is posted by the point tracker's feed itself, the moment the frame's point list stands
  // (Tracker::points_ready); otherwise in place after the feed
  bool line_feed_posted = false;
  int line_feed_rc = PLV_OK;
  double vps[6] = {0, 0, 0, 0, 0, 0};
  if (lines) TRY(plv_vanishing_points(st->R_ItoC, st->intrinsics, vps));
  if (lines && io->update && io->update->opt_lines) {
    plv_line_feed_pool_args(ctx, st, io->update->opt_lines);  // (the worker forms the line update's pool at the end of the feed)
    Tf->points_ready = [&](int np, const float *pts, const uint64_t *pids) {
      line_feed_posted = true;
      line_feed_rc = plv_line_tracker_feed_async_points(ctx, io->timestamp, vps, np, pts, pids);
    };
  }
  int rc_feed;
  if (io->slot >= 0)
    rc_feed = plv_tracker_feed_staged(ctx, io->timestamp, io->slot, io->mask);
  else
    rc_feed = plv_tracker_feed(ctx, io->timestamp, io->img, io->stride, io->mask);
  Tf->points_ready = nullptr;
  Tf->spec_st = nullptr, Tf->spec_opt = nullptr;
  TRY(rc_feed);
  if (lines) {
    if (io->update && io->update->opt_lines) {
      if (!line_feed_posted) TRY(plv_line_tracker_feed_async(ctx, io->timestamp, vps));  // (a feed that ended before its point list: first frame, nothing tracked)
      else TRY(line_feed_rc);
    }
    else
      TRY(plv_line_tracker_feed(ctx, io->timestamp, vps));
  }
  if (!io->update) {
    io->line_db_size = lines ? plv_line_db_size(ctx) : 0;
    return PLV_OK;
  }
  rx_feed.stop();
  plv::frame_mark("@ feed done, try_update starts");
  const int rc = plv_camera_try_update(ctx, st, io->update);
  plv::frame_mark("@ try_update returned");
  if (lines && !io->update->opt_lines) (void)plv_line_tracker_feed_wait(ctx);
  io->line_db_size = io->update->opt_lines ? io->update->line_db_size : (lines ? plv_line_db_size(ctx) : 0);
  return rc;
}

}  // extern "C"
