// tracker_api.hip — host mirror of ov_core::TrackKLT's monocular frame logic and of
// ov_core::FeatureDatabase behind the C-ABI (plv_tracker_* / plv_db_*).
//   TrackKLT::feed_new_camera / feed_monocular   REF: open_vins/ov_core/src/track/TrackKLT.cpp:34-200
//   FeatureDatabase                              REF: open_vins/ov_core/src/feat/FeatureDatabase.cpp:60-323
// This is bookkeeping (vectors and a hash map), exactly what the reference keeps on the host;
// every image / point computation goes through the device entry points of frontend_api.hip.
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "plv_ctx.hpp"

namespace {

struct Track {  // ov_core::Feature, one camera  (REF: open_vins/ov_core/src/feat/Feature.h:43-77)
  std::vector<double> t;
  std::vector<float> uv, uvn;  // 2 per observation
};

struct Tracker {
  std::vector<float> pts_last;     // 2 per point
  std::vector<uint64_t> ids_last;
  std::vector<uint8_t> mask_last;  // W*H or empty
  uint64_t currid = 0;             // REF: TrackBase::currid (4*num_aruco + 1 - 1 = 0 without ArUco tags)
  std::unordered_map<uint64_t, Track> db;
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

extern "C" {

void plv_tracker_destroy(plv_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_mtx);
  auto it = g_trk.find(ctx);
  if (it != g_trk.end()) {
    delete it->second;
    g_trk.erase(it);
  }
}

int plv_tracker_feed(plv_ctx *ctx, double timestamp, const uint8_t *img, int stride, const uint8_t *mask) {
  if (!ctx || !img) return PLV_E_BADARG;
  Tracker *T = trk(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);  // REF: mtx_feeds.at(cam_id), TrackKLT.cpp:54,100
  const int W = ctx->cfg.width, H = ctx->cfg.height;
  const int cap = std::max(ctx->cfg.num_features * 4, 1024) + (int)T->ids_last.size();
  TRY(plv_feed_image(ctx, img, stride));  // :59 equalizeHist, :71 buildOpticalFlowPyramid
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
  TRY(plv_perform_detection(ctx, PLV_PYR_LAST, T->mask_last.empty() ? nullptr : T->mask_last.data(), pts.data(), ids.data(), n,
                            cap, &T->currid, &n));
  // :134-139 temporal KLT with the previous positions as the initial flow
  std::vector<float> pts_new(pts.begin(), pts.begin() + 2 * (size_t)n), n1(2 * (size_t)std::max(n, 1));
  std::vector<uint8_t> mask_ll((size_t)std::max(n, 1), 0);
  if (n == 0) {  // :143-152
    T->pts_last.clear();
    T->ids_last.clear();
    keep_mask();
    return PLV_OK;
  }
  TRY(plv_perform_matching(ctx, n, pts.data(), pts_new.data(), mask_ll.data(), nullptr, n1.data(), nullptr));
  // :158-173 keep in-bounds, unmasked, matched points; :176-179 database update
  std::vector<float> good;
  std::vector<uint64_t> good_ids;
  for (int i = 0; i < n; ++i) {
    const float x = pts_new[2 * i], y = pts_new[2 * i + 1];
    if (x < 0 || y < 0 || (int)x >= W || (int)y >= H) continue;
    if (mask && mask[(size_t)(int)y * W + (int)x] > 127) continue;
    if (!mask_ll[i]) continue;
    good.push_back(x);
    good.push_back(y);
    good_ids.push_back(ids[i]);
    Track &tr = T->db[ids[i]];
    tr.t.push_back(timestamp);
    tr.uv.push_back(x);
    tr.uv.push_back(y);
    tr.uvn.push_back(n1[2 * i]);
    tr.uvn.push_back(n1[2 * i + 1]);
  }
  T->pts_last.swap(good);
  T->ids_last.swap(good_ids);
  keep_mask();
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

}  // extern "C"
