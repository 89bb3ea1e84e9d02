// line_api.hip — TrackLSD behind the C-ABI (plv_detect_lines, plv_line_tracker_*, plv_line_db_*).
//   TrackLSD::feed_new_camera / feed_monocular      REF: PL-VIWO/src/update/cam/TrackLSD.cpp:39-192
//   TrackLSD::perform_detection_monocular           REF: :194-235   (device: line_kernels.hip)
//   TrackLSD::AssignPointToLines / PointLineDistance    REF: :744-814
//   TrackLSD::LineMatch / LineSimilar               REF: :368-407, :816-830
//   TrackLSD::LineClassification / LineClass        REF: :318-366
//   LineHelper::Vanishing_Points / Distort          REF: linefeat/LineHelper.cpp:1026-1088
//   LineFeatureDatabase::update_feature             REF: linefeat/LineFeatureDatabase.cpp:40-76
// Pixel work (half-resolution resize, Sobel, non-maximum suppression, hysteresis) runs on the device; the
// detector's list processing (chain walk + segment growth) is a HOST stage by default (plv_line_walk_mode
// selects the device kernels; DESIGN.md "Line detector" has the measurements behind that choice).  The rest
// here is the reference's own host bookkeeping: id hand-over between frames and the line track store.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "line_host.hpp"
#include "radtan_core.hpp"
#include "line_kernels.hpp"
#include "update_state.hpp"

using namespace plv;
using namespace plv::linehost;

int plv_tracker_last(plv_ctx *ctx, float *pts, uint64_t *ids, int cap, int *n);
namespace {

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

struct LineTrack {  // LineFeature, one camera   REF: linefeat/LineFeature.h:22-107
  std::vector<double> t;
  std::vector<float> uv, uvn;  // 4 per observation
  std::vector<int> points;     // ids of the point features assigned at every observation (appended, REF :50-52)
  int D = 0;
};

struct LineCand {
  uint64_t id;
  LineTrack tr;
};
// The pool of LineHelper::get_line_features (REF: linefeat/LineHelper.cpp:33-44: features_containing_older + features_not_containing_newer,
// remove_unusable_measurements, sort) taken out of the database.  It reads times only, so plv_camera_try_update forms it while the
// point update is still running on the device (form_line_pool below); plv_camera_update_lines consumes it.
struct PoolArgs {  // what form_line_pool reads of the state and the update options: times only
  double t_prev_frame = 0, state_time = 0, dt = 0, t_oldest = 0, t_oldest2 = 0;
  int n_clones = 0;
  static PoolArgs of(const plv_state_view *st, const plv_update_options *opt) {
    PoolArgs a;
    a.t_prev_frame = opt->t_prev_frame, a.state_time = opt->state_time, a.dt = st->cam_dt, a.n_clones = st->n_clones;
    a.t_oldest = st->clone_time[0], a.t_oldest2 = st->clone_time[1];
    return a;
  }
};
struct LinePool {
  bool valid = false;
  double t_prev_frame = 0, state_time = 0, t_oldest = 0, t_oldest2 = 0, dt = 0;
  int n_clones = 0, n_pool = 0, db_size_before = 0;
  std::vector<LineCand> pool;                       // trimmed, sorted long to short
  std::unordered_map<uint64_t, LineTrack> unused;   // db_unused so far (observations newer than the window)
};

// plv_camera_update_lines in two halves: what the first (pool, staging, launch of triangulation + Jacobians + gate) hands to the
// second (gate verdicts, compression + EKFUpdate, selection, database).  plv_camera_try_update runs the first half inside the point
// update's wait ("chained": the launch sits behind that update on the stream and forms the corrected state itself), the second
// after it has applied the point update's dx; called on its own, plv_camera_update_lines runs one after the other.
struct LinesJob {
  enum Stage { NONE = 0, EMPTY, FAILED, FUSED_LAUNCHED, FUSED_NOTHING, TWO_STEP };
  bool pending = false;  // a chained first half waits for its second
  Stage stage = NONE;
  int rc = 0;            // FAILED: the status to return
  LinePool LP;
  int Lp = 0, nobs = 0, most_valid = 0, k = 0, cap = 0, n_clones = 0;
  double state_time = 0, t_prev_frame = 0;
  std::vector<int> ptr, D, valid_n, cols, pt_ptr, pt_ids;
  std::vector<double> anchor, ot;
  std::vector<float> uv, uvn;
  std::vector<uint8_t> has, flags;
  std::vector<std::vector<double>> cpiR, cpip, cpiQ;
  std::vector<std::vector<int>> cpiC;
  std::vector<double> allR, allp;
  std::vector<double> lg_two_step;   // TWO_STEP: the triangulation ran as its own call inside the first half
  std::vector<uint8_t> ok_two_step;
  double us_pool = 0;
};

struct LineTracker {
  LinePool pool_prep;
  LinesJob ujob;  // the line update in two halves (plv_camera_update_lines)
  std::vector<float> lines_last;  // 4 per line
  std::vector<uint64_t> ids_last;
  std::vector<int> rel_ptr_last{0};  // CSR: point ids on each last line (ascending, the reference keeps a std::map)
  std::vector<uint64_t> rel_id_last;
  uint64_t currid = 1;  // REF: TrackLSD.cpp:32, ids are pre-incremented (:234)
  bool walk_on_device = false;  // plv_line_walk_mode
  bool prefetch = false;        // plv_line_prefetch_mode: plv_tracker_feed* detects the lines of the new image ahead of plv_line_tracker_feed
  std::unordered_map<uint64_t, LineTrack> db;
  // device buffers of the detector
  DevBuf half, map, work, pts, chains, counts, segs, seg_count, uv_in, uv_out;
  // Component labels of the edge map (line_kernels.hip ccl_*_kernel) for the worker's host stage: formed on a stream of their own behind
  // the edge kernel, next to the point front-end's flow; the worker waits for labels_ready before it splits the detection by them
  DevBuf lab, lab_cnt, lab_roots;
  hipStream_t ccl_stream = nullptr;
  hipEvent_t canny_done = nullptr, labels_ready = nullptr;
  PinBuf pin;
  hipEvent_t edges_ready = nullptr;  // plv_line_detect_launch: the maps of image `pending_which` are on their way to the host
  // measurement knob PLV_KNOB_EDGES_SIDE: the edge kernel of a prefetched detection on its own stream, behind the pyramid only
  // (plv_line_edges_fork records pyr_done on the ctx stream before the flow is enqueued).  Slower than the default by 6-15 us per frame.
  hipStream_t edge_stream = nullptr;
  hipEvent_t pyr_done = nullptr;
  bool edge_fork = false;
  // plv_line_edges_early: the detection being launched reads the RAW image of the frame being fed (the pyramid that will hold its
  // equalised form is not current yet) through its histogram
  std::vector<uint64_t> dec_ids;  // plv_decision_trace: the last line update's batch and its gate values (plv_last_line_decisions)
  std::vector<double> dec_vals;
  const uint8_t *early_raw = nullptr;
  const unsigned *early_hist = nullptr;
  int early_w = 0, early_h = 0;
  int pending_which = -1, pending_fed = -1;
  bool defer_finish = false;        // plv_camera_try_update: the line update leaves its database hand-back (cleanup_lines) behind ...
  std::function<void()> deferred;   // ... to run before anything else reads the tracker: in the next frame's wait for the flow (ltr())
  std::atomic<int> defer_state{0};  // 0: not posted; 1: `deferred` is the worker's third kind of job (round 5: it runs there while the
                                    // caller finishes the frame); 2: done.  Every entry point joins it like the feed (ltr())
  // plv_camera_get_line_features: the state the line pool is triangulated on (the reference runs get_line_features BEFORE the point
  // update's correction is applied, UpdaterCamera.cpp:148-152), kept until the next plv_camera_update_lines
  struct TriState {
    bool valid = false;
    plv_state_view view;
    std::vector<double> clone_time, clone_R, clone_p;
    std::vector<int> clone_id;
  } tri_state;
  std::vector<float> cached;  // plv_line_detect_finish: the segments of image `cached_which` of frame `cached_fed`
  int cached_which = -1, cached_fed = -1;
  std::mutex mtx;
  // Host stage of the detector (wait for the edge maps, chain walk, segment growth) on a worker thread: plv_line_detect_launch
  // hands it a job, the caller's thread goes on enqueuing / waiting for the point front-end, whoever needs the segments joins.
  linehost::Job job;
  std::thread worker;
  std::mutex jm;
  std::condition_variable jcv;
  std::atomic<int> job_state{0};  // 0 idle, 1 posted, 2 done; -1 quit (written under jm, polled without it: wait_polling)
  // plv_line_tracker_feed_async: the rest of TrackLSD::feed_monocular (assignment, matching, classification, track store) as a
  // second job of the same thread.  While it is posted the worker owns the tracker state; every entry point joins it first (ltr()).
  struct FeedJob {
    plv_ctx *ctx = nullptr;
    double timestamp = 0, vps[6] = {0, 0, 0, 0, 0, 0};
    double K8[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the camera model at post time (the caller may refresh ctx->cfg.intrinsics while the job runs)
    std::vector<float> pts;
    std::vector<uint64_t> pids;
    int rc = PLV_OK;
    // plv_camera_frame with a line update to follow: the worker forms the update's pool (form_line_pool: times only) as the last step
    // of the feed — the caller's thread, busy enqueuing the point update, finds it ready (round 3 formed it on that thread, inside
    // the point update's wait; with the chained line launch that was 30 us in front of the line launch)
    bool pool_on = false;
    PoolArgs pool_args;
  } feed;
  std::atomic<int> feed_state{0};  // 0 idle, 1 posted, 2 done
  std::chrono::steady_clock::time_point job_posted, feed_posted;
  // Second half of the host stage on a thread of its own: segments are grown along chain c while the walk is still producing chain
  // c + 1 (the walk publishes its chain count after every chain; both halves are sequential in themselves, the two overlap)
  linehost::HostStage host;  // chain walk + segment growth (line_host.hpp); host.fit = the fitter threads
};

std::mutex g_mtx;
std::unordered_map<plv_ctx *, LineTracker *> g_lt;
LineTracker *ltr(plv_ctx *ctx, bool run_deferred = true);  // (defined after LineTracker's worker protocol)


void form_line_pool(LineTracker *T, const PoolArgs &A, LinePool &R);
void discard_line_pool(LineTracker *T);
// (feed_points_impl is defined further down, outside this namespace: the worker reaches it through this pointer)
int (*g_feed_impl)(plv_ctx *, LineTracker *, double, const double *, int, const float *, const uint64_t *, const double *) = nullptr;
void line_worker(LineTracker *T) {
  for (;;) {
    bool do_detect = false, do_feed = false, do_deferred = false;
    {
      std::unique_lock<std::mutex> lk(T->jm);
      wait_polling(lk, T->jcv, [&] { return T->job_state == 1 || T->job_state == -1 || T->feed_state == 1 || T->defer_state == 1; });
      if (T->job_state == -1) return;
      // the database hand-back first: it is short (~20 us), every entry point joins it (ltr()), and behind a detect job it would
      // wait for that job's edge maps and its whole host stage while the caller spins
      do_deferred = T->defer_state == 1;
      do_detect = !do_deferred && T->job_state == 1;
      do_feed = !do_deferred && !do_detect && T->feed_state == 1;
    }
    if (do_deferred) {  // the line update's database hand-back (plv_camera_update_lines: finish), off the caller's thread
      std::function<void()> f;
      f.swap(T->deferred);
      if (f) f();
      {
        std::lock_guard<std::mutex> lk(T->jm);
        T->defer_state = 2;
      }
      T->jcv.notify_all();
      continue;
    }
    if (do_feed) {
      LineTracker::FeedJob &F = T->feed;
      if (plv::host_phases().on)
        plv::host_phases().add("line worker: feed job starts after its post", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T->feed_posted).count());
      const auto Fs = std::chrono::steady_clock::now();
      plv::counters().w_feed_start_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(Fs - T->feed_posted).count();
      const int rc = g_feed_impl(F.ctx, T, F.timestamp, F.vps, (int)F.pids.size(), F.pts.data(), F.pids.data(), F.K8);
      if (rc == PLV_OK && F.pool_on) {
        discard_line_pool(T);
        form_line_pool(T, F.pool_args, T->pool_prep);
      }
      F.pool_on = false;
      plv::frame_mark("@ (worker) line feed + pool done");
      plv::counters().w_feed_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - Fs).count();
      {
        std::lock_guard<std::mutex> lk(T->jm);
        F.rc = rc;
        T->feed_state = 2;
      }
      T->jcv.notify_all();
      continue;
    }
    const bool timing = plv::knob(plv::PLV_KNOB_LINE_TIMING);
    auto W0 = std::chrono::steady_clock::now();
    (void)hipSetDevice(T->job.device);
    int rc = PLV_OK;
    if (plv::event_sync(T->edges_ready) != hipSuccess) rc = PLV_E_DEVICE;  // the two maps are on the host
    if (rc == PLV_OK && T->job.hlab && plv::event_sync(T->labels_ready) != hipSuccess) rc = PLV_E_DEVICE;  // ... and the component labels
    auto W1 = std::chrono::steady_clock::now();
    if (rc == PLV_OK) rc = host_extract(&T->host, T->job, timing);
    {
      auto ns = [](auto a, auto b) { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
      plv::counters().w_wake_ns += ns(T->job_posted, W0);
      plv::counters().w_maps_ns += ns(W0, W1);
      plv::counters().w_extract_ns += ns(W1, std::chrono::steady_clock::now());
    }
    if (plv::host_phases().on)
      plv::host_phases().add("line worker: detect job, post to done", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T->job_posted).count());
    plv::frame_mark("@ (worker) line detection done");
    if (timing) {
      auto W2 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      fprintf(stderr, "line worker: woke %.1f us after the post, edge maps %.1f us, walk + fit %.1f us\n", us(T->job_posted, W0), us(W0, W1), us(W1, W2));
    }
    {
      std::lock_guard<std::mutex> lk(T->jm);
      T->job.rc = rc;
      if (T->job_state == 1) T->job_state = 2;
    }
    T->jcv.notify_all();
  }
}
// waits for a posted job; true when one was in flight (its result is then in T->job)
bool join_job(LineTracker *T) {
  std::unique_lock<std::mutex> lk(T->jm);
  if (T->job_state == 0) return false;
  auto J0 = std::chrono::steady_clock::now();
  wait_polling(lk, T->jcv, [&] { return T->job_state == 2; });
  T->job_state = 0;
  if (plv::knob(plv::PLV_KNOB_LINE_TIMING))
    fprintf(stderr, "line join: waited %.1f us (posted %.1f us ago)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - J0).count(),
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T->job_posted).count());
  return true;
}

// Every entry point reaches the tracker through here: an asynchronous feed still running on the worker is joined first (its status
// is kept for plv_line_tracker_feed_wait).
LineTracker *ltr(plv_ctx *ctx, bool run_deferred) {
  LineTracker *T;
  {
    std::lock_guard<std::mutex> lk(g_mtx);
    auto it = g_lt.find(ctx);
    if (it == g_lt.end()) {
      T = new LineTracker();
      g_lt[ctx] = T;
      return T;
    }
    T = it->second;
  }
  std::unique_lock<std::mutex> lk(T->jm);
  if (T->feed_state != 0) {
    plv::NsScope ns_join(plv::counters().line_join_ns);
    plv::HostPhase ph("line feed join: wait");
    wait_polling(lk, T->jcv, [&] { return T->feed_state == 2; });
    T->feed_state = 0;
    if (plv::host_phases().on)
      plv::host_phases().add("line feed join: time since the post", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T->feed_posted).count());
  }
  if (T->defer_state != 0) {  // a hand-back posted to the worker: done before anything reads the tracker
    wait_polling(lk, T->jcv, [&] { return T->defer_state == 2; });
    T->defer_state = 0;
  }
  lk.unlock();
  if (run_deferred && T->deferred) {  // (only the thread that owns the ctx gets here: the worker reaches the tracker directly)
    std::function<void()> f;
    f.swap(T->deferred);
    f();
  }
  return T;
}

// detection on the device + the host tail of perform_detection_monocular (x2, FilterShortLines)
int detect(plv_ctx *ctx, LineTracker *T, int which, std::vector<float> &lines, bool launch_only = false) {
  const bool prelaunched = !launch_only && !T->walk_on_device && T->pending_which == which && T->pending_fed == plv_front_fed_count(ctx);
  if (!launch_only) T->pending_which = -1;
  int W = 0, H = 0;
  const bool early = launch_only && T->early_raw != nullptr;
  const uint8_t *d_img = early ? T->early_raw : plv_front_level0(ctx, which, &W, &H);
  if (early) W = T->early_w, H = T->early_h;
  if (!d_img) {
    set_last_error("plv_detect_lines: no image has been fed");
    return PLV_E_BADARG;
  }
  const int w = W / 2, h = H / 2;
  const size_t npix = (size_t)w * h;
  // (a helper thread the last detection's host stage was closed without — cut off in the middle of a part — reads the job, the maps and
  // the lists this call rewrites, possibly in buffers it regrows: nobody is inside a job from here on.  Immediate, unless that thread
  // has been off its CPU for a whole frame.)
  if (!prelaunched) plv::linehost::quiesce_helpers(T->host.fit);
  TRY(T->half.reserve(npix));
  TRY(T->map.reserve(npix));
  TRY(T->work.reserve(npix));
  TRY(T->pts.reserve(npix * sizeof(int2)));
  TRY(T->chains.reserve(kChainCap * sizeof(FldChain)));
  TRY(T->counts.reserve(4 * sizeof(int)));
  const size_t slot_cap = npix / (size_t)std::max(1, ctx->cfg.line_length_threshold) + kChainCap;
  TRY(T->segs.reserve(slot_cap * sizeof(float4)));
  TRY(T->seg_count.reserve(kChainCap * sizeof(int)));
  FldBuffers b{T->half.as<uint8_t>(), T->map.as<uint8_t>(), T->work.as<uint8_t>(), T->pts.as<int2>(), T->chains.as<FldChain>(),
               kChainCap,             T->counts.as<int>(),  T->segs.as<float4>(), T->seg_count.as<int>()};
  FldParams fp{ctx->cfg.line_length_threshold, (float)ctx->cfg.line_distance_threshold, ctx->cfg.canny_th1, ctx->cfg.canny_th2};
  const size_t bytes = 16 + kChainCap * (sizeof(FldChain) + sizeof(int));
  const size_t lab_off = bytes + ((2 * npix + 63) & ~(size_t)63) + npix * sizeof(int2);  // (behind the host stage's chain points)
  // (behind the labels: the parts' pixel lists, ccl_flatten_kernel)
  const size_t nblk = (npix + 255) / 256, sorted_off = (lab_off + npix + 63) & ~(size_t)63, bins_off = sorted_off + nblk * 256;
  const size_t pin_end = bins_off + nblk * (size_t)(plv::line_label_parts() + 1) * sizeof(unsigned short);
  TRY(T->pin.reserve(std::max(bytes + slot_cap * sizeof(float4), pin_end + 64)));
  char *hp = T->pin.as<char>();
  // host walk without hysteresis (the shipped thresholds are equal): the edge kernel writes the two maps straight into the pinned
  // buffer the host stage reads — no device copies of them, no copy commands behind the kernel
  const bool maps_to_host = !T->walk_on_device && fp.canny_low == fp.canny_high;
  if (maps_to_host) {
    b.map = (uint8_t *)(hp + bytes);
    b.half = b.map + npix;
  }
  // the worker's host stage splits its work by the components of the edge map when the device labels them (PLV_KNOB_LINE_LABELS_OFF:
  // it walks the map as one sequence, as rounds 2-4 did)
  const bool labels_off = plv::knob(plv::PLV_KNOB_LINE_LABELS_OFF);
  const bool with_labels = launch_only && maps_to_host && !labels_off;
  if (with_labels) {
    TRY(T->lab.reserve(npix * sizeof(int)));
    TRY(T->lab_cnt.reserve(npix * sizeof(int)));
    TRY(T->lab_roots.reserve(plv::line_label_roots_bytes()));
    b.lab_work = T->lab.as<int>();
    b.lab_cnt = T->lab_cnt.as<int>();
    b.lab_roots = T->lab_roots.as<int>();
    b.lab_out = (uint8_t *)(hp + lab_off);
    if (!plv::knob(plv::PLV_KNOB_PART_LISTS_OFF)) {
      b.blk_sorted = (uint8_t *)(hp + sorted_off);
      b.blk_bins = (unsigned short *)(hp + bins_off);
    }
  }
  hipStream_t es = ctx->stream;
  if (launch_only && maps_to_host && T->edge_fork) {
    PLV_HIP_CHECK(hipStreamWaitEvent(T->edge_stream, T->pyr_done, 0));
    es = T->edge_stream;
  }
  T->edge_fork = false;
  if (!prelaunched) TRY(launch_line_edges(ctx, d_img, W, H, fp, b, es, early ? T->early_hist : nullptr));
  if (with_labels && !T->ccl_stream) {
    PLV_HIP_CHECK(hipStreamCreateWithFlags(&T->ccl_stream, hipStreamNonBlocking));
    PLV_HIP_CHECK(hipEventCreateWithFlags(&T->canny_done, hipEventDisableTiming));
    PLV_HIP_CHECK(hipEventCreateWithFlags(&T->labels_ready, hipEventDisableTiming));
  }
  if (with_labels) PLV_HIP_CHECK(hipEventRecord(T->canny_done, es));
  if (launch_only && !T->edges_ready) PLV_HIP_CHECK(hipEventCreateWithFlags(&T->edges_ready, hipEventDisableTiming));
  if (launch_only && maps_to_host) PLV_HIP_CHECK(hipEventRecord(T->edges_ready, es));  // (the two maps are the edge kernel's own stores)
  // the image feed's remaining launches (the pyramid) go behind the edge kernel NOW: what follows here is host work
  if (early && ctx->after_edges) TRY(ctx->after_edges(ctx->after_edges_arg));
  if (with_labels) {
    PLV_HIP_CHECK(hipStreamWaitEvent(T->ccl_stream, T->canny_done, 0));
    TRY(launch_line_labels(ctx, w, h, b, T->ccl_stream, fp.length_threshold));
    PLV_HIP_CHECK(hipEventRecord(T->labels_ready, T->ccl_stream));
  }
  const float thr2 = ctx->cfg.line_min_length_px * ctx->cfg.line_min_length_px;
  auto emit = [&](const float4 &sg) {
    const float x1 = sg.x * 2, y1 = sg.y * 2, x2 = sg.z * 2, y2 = sg.w * 2;  // REF :218-220
    const float l2 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
    if (!(l2 > thr2)) return;  // FilterShortLines(lines0, 40)   REF :232, :435-448
    lines.insert(lines.end(), {x1, y1, x2, y2});
  };
  lines.clear();
  if (!T->walk_on_device) {
    // The chain walk consumes edge pixels in raster order and every step depends on the one before, and so does the growth of
    // the segments along a chain: one dependent scalar sequence of ~10^4 .. 3*10^4 steps.  One MI355X lane retires such a step
    // in ~0.8 us (measured: 28 ms per frame for the walk on the dense-edge test image, 160 us for fld_fit_kernel's longest
    // chain), a host core in ~30 ns, so both run on the host on the 90 KB edge map and the half-resolution image (DESIGN.md "Line
    // detector"); the pixel work (resize, Sobel, non-maximum suppression, hysteresis) stays on the device.
    const bool timing = plv::knob(plv::PLV_KNOB_LINE_TIMING);
    if (prelaunched) {  // plv_line_detect_launch posted the job: the worker thread has been walking meanwhile
      const bool had = join_job(T);
      if (had && T->job.rc == PLV_OK) {
        lines.swap(T->job.lines);  // (no profiler collection here: this branch may run on the worker thread)
        return PLV_OK;
      }
      if (had && T->job.rc != PLV_OK) return T->job.rc;
    } else {
      (void)join_job(T);  // a stale job (another image / frame) must be off the buffers before they are reused
    }
    // (... and so must a helper thread that job was closed without: it reads the job's fields, rewritten below — found by the test
    // suite run with napping helpers, PLV_DEBUG_KNOBS = 1 << 28: a part started after the fields of a detection without labels were in)
    plv::linehost::quiesce_helpers(T->host.fit);
    Job &J = T->job;
    J.device = ctx->device, J.w = w, J.h = h, J.length_threshold = fp.length_threshold, J.distance_threshold = fp.distance_threshold;
    J.thr2 = thr2;
    uint8_t *hmap = (uint8_t *)(hp + bytes);
    J.hmap = hmap, J.hhalf = hmap + npix;
    J.hpts = (int2 *)(hp + bytes + ((2 * npix + 63) & ~(size_t)63));
    J.hc = (FldChain *)(hp + 16);
    J.hlab = with_labels ? b.lab_out : nullptr;
    J.parts = with_labels ? plv::line_label_parts() : 0;
    J.hsorted = with_labels ? b.blk_sorted : nullptr, J.hbins = with_labels ? b.blk_bins : nullptr;
    if (!maps_to_host) {
      PLV_HIP_CHECK(plv::memcpy_async(hmap, T->map.p, npix, hipMemcpyDeviceToHost, ctx->stream));
      PLV_HIP_CHECK(plv::memcpy_async(hmap + npix, T->half.p, npix, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (launch_only) {
      if (!maps_to_host) PLV_HIP_CHECK(hipEventRecord(T->edges_ready, es));  // (behind the two copy commands above)
      T->pending_which = which;
      T->pending_fed = plv_front_fed_count(ctx) + (early ? 1 : 0);  // (early: the image becomes the current one when its feed returns)
      if (!T->worker.joinable()) T->worker = std::thread(line_worker, T);
      {
        std::lock_guard<std::mutex> lk(T->jm);
        T->job_state = 1;
        T->job_posted = std::chrono::steady_clock::now();
      }
      T->jcv.notify_all();
      prewake_helpers(&T->host);  // (the host stage's helper threads: awake and polling by the time the maps and labels are on the host)
      return PLV_OK;
    }
    PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
    TRY(host_extract(&T->host, J, timing));
    lines.swap(J.lines);
    ctx->prof.collect();
    return PLV_OK;
  }
  TRY(launch_line_walk(ctx, w, h, fp, b));
  TRY(launch_line_fit(ctx, w, h, fp, b));
  // download: counts, chain table, per-chain segment counts, segment slots
  PLV_HIP_CHECK(plv::memcpy_async(hp, T->counts.p, 16, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  const int n_chain = ((int *)hp)[0], n_slot = ((int *)hp)[1];
  if (n_chain >= kChainCap) {
    set_last_error("plv_detect_lines: more than %d edge chains", kChainCap);
    return PLV_E_CAPACITY;
  }
  if (n_chain > 0) {
    FldChain *hc = (FldChain *)(hp + 16);
    int *hn = (int *)(hp + 16 + kChainCap * sizeof(FldChain));
    float4 *hs = (float4 *)(hp + bytes);
    PLV_HIP_CHECK(plv::memcpy_async(hc, T->chains.p, n_chain * sizeof(FldChain), hipMemcpyDeviceToHost, ctx->stream));
    PLV_HIP_CHECK(plv::memcpy_async(hn, T->seg_count.p, n_chain * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PLV_HIP_CHECK(plv::memcpy_async(hs, T->segs.p, (size_t)n_slot * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
    for (int c = 0; c < n_chain; ++c)  // chains are in raster order of their seeds = the detector's output order
      for (int q = 0; q < hn[c]; ++q) emit(hs[hc[c].slot + q]);
  }
  ctx->prof.collect();
  return PLV_OK;
}


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

void plv_line_tracker_destroy(plv_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_mtx);
  auto it = g_lt.find(ctx);
  if (it != g_lt.end()) {
    LineTracker *T = it->second;
    if (T->worker.joinable()) {
      {
        std::unique_lock<std::mutex> lk(T->jm);
        T->jcv.wait(lk, [&] { return T->job_state != 1 && T->feed_state != 1 && T->defer_state != 1; });  // let posted jobs finish: they read what is released below
        T->job_state = -1;
      }
      T->jcv.notify_all();
      T->worker.join();
    }
    // (the fitter threads end with T->host: ~HostStage)
    DevBuf *bufs[] = {&T->half, &T->map, &T->work, &T->pts, &T->chains, &T->counts, &T->segs, &T->seg_count, &T->uv_in, &T->uv_out, &T->lab, &T->lab_cnt, &T->lab_roots};
    for (DevBuf *b : bufs) b->release();
    T->pin.release();
    if (T->edges_ready) (void)hipEventDestroy(T->edges_ready);
    if (T->pyr_done) (void)hipEventDestroy(T->pyr_done);
    if (T->canny_done) (void)hipEventDestroy(T->canny_done);
    if (T->labels_ready) (void)hipEventDestroy(T->labels_ready);
    if (T->ccl_stream) (void)hipStreamDestroy(T->ccl_stream);
    if (T->edge_stream) (void)hipStreamDestroy(T->edge_stream);
    delete T;
    g_lt.erase(it);
  }
}

int plv_line_walk_mode(plv_ctx *ctx, int on_device) {
  if (!ctx) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  T->walk_on_device = on_device != 0;
  return PLV_OK;
}

int plv_line_worker_config(int spin_us, int fit_threads_n, int *spin_us_out, int *fit_threads_out) {
  if (spin_us >= 0) spin_budget_us().store(spin_us);
  if (fit_threads_n >= 0) fit_threads().store(std::min(fit_threads_n, (int)plv::linehost::Fit::kThreads));
  if (spin_us_out) *spin_us_out = spin_budget_us().load();
  if (fit_threads_out) *fit_threads_out = fit_threads().load();
  return PLV_OK;
}

int plv_line_prefetch_mode(plv_ctx *ctx, int on) {
  if (!ctx) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  T->prefetch = on != 0;
  return PLV_OK;
}
int plv_line_prefetch_enabled(plv_ctx *ctx) {
  {
    std::lock_guard<std::mutex> lk(g_mtx);
    if (g_lt.find(ctx) == g_lt.end()) return 0;  // no line tracker was ever touched on this ctx
  }
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  return T->prefetch && !T->walk_on_device ? 1 : 0;
}

int plv_detect_lines(plv_ctx *ctx, int which, float *lines, int cap, int *n_out) {
  if (!ctx || !n_out) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  std::vector<float> v;
  TRY(detect(ctx, T, which, v));
  *n_out = (int)v.size() / 4;
  if (*n_out > cap) return PLV_E_CAPACITY;
  if (lines) std::copy(v.begin(), v.end(), lines);
  return PLV_OK;
}

int plv_assign_points_to_lines(const float *lines, int n_lines, const float *pts, const uint64_t *ids, int n_pts, int *kept,
                               int *rel_ptr, uint64_t *rel_id, double *rel_dist, int *pos_ptr, float *pos_xy, int *n_kept) {
  if (!lines || !pts || !ids || !kept || !rel_ptr || !rel_id || !rel_dist || !pos_ptr || !pos_xy || !n_kept) return PLV_E_BADARG;
  Assign A;
  assign_points(lines, n_lines, pts, ids, n_pts, A);
  *n_kept = (int)A.kept.size();
  std::copy(A.kept.begin(), A.kept.end(), kept);
  std::copy(A.rel_ptr.begin(), A.rel_ptr.end(), rel_ptr);
  std::copy(A.rel_id.begin(), A.rel_id.end(), rel_id);
  std::copy(A.rel_dist.begin(), A.rel_dist.end(), rel_dist);
  std::copy(A.pos_ptr.begin(), A.pos_ptr.end(), pos_ptr);
  std::copy(A.pos.begin(), A.pos.end(), pos_xy);
  return PLV_OK;
}

int plv_line_match(const float *lines_new, int n_new, const int *rel_ptr_new, const uint64_t *rel_id_new, const float *lines_last,
                   int n_last, const int *rel_ptr_last, const uint64_t *rel_id_last, int *match_of_new) {
  if (!match_of_new || (n_new > 0 && (!lines_new || !rel_ptr_new)) || (n_last > 0 && (!lines_last || !rel_ptr_last)))
    return PLV_E_BADARG;
  match_lines(lines_new, n_new, rel_ptr_new, rel_id_new, lines_last, n_last, rel_ptr_last, rel_id_last, match_of_new);
  return PLV_OK;
}

int plv_line_classification(const float *line, const double *vps) {
  if (!line || !vps) return PLV_E_BADARG;
  if (line_class(line, vps + 4)) return 3;
  if (line_class(line, vps + 2)) return 2;
  if (line_class(line, vps)) return 1;
  return 0;
}

int plv_vanishing_points(const double *R_ItoC, const double *K8, double *vps) {
  if (!R_ItoC || !K8 || !vps) return PLV_E_BADARG;
  for (int a = 0; a < 3; ++a) {
    // column a of R_ItoC, first two components used as normalised coordinates (REF :1037-1046: no division by z)
    const double x = R_ItoC[a], y = R_ItoC[3 + a];
    const double r = std::sqrt(x * x + y * y), r_2 = r * r, r_4 = r_2 * r_2;
    const double x1 = x * (1 + K8[4] * r_2 + K8[5] * r_4) + 2 * K8[6] * x * y + K8[7] * (r_2 + 2 * x * x);
    const double y1 = y * (1 + K8[4] * r_2 + K8[5] * r_4) + K8[6] * (r_2 + 2 * y * y) + 2 * K8[7] * x * y;
    vps[2 * a] = (double)(float)(K8[0] * x1 + K8[2]);
    vps[2 * a + 1] = (double)(float)(K8[1] * y1 + K8[3]);
  }
  vps[5] *= 1000;  // REF :1051-1054
  return PLV_OK;
}

// TrackLSD::feed_monocular for the image currently in the ctx (fed by plv_tracker_feed / plv_feed_image,
// which also is where the reference's second equalizeHist comes from: same input, same result).
// (internal, the tracker feed) marks the point on the ctx stream the prefetched edge kernel has to wait for — the pyramid of the image
// just fed — so that the kernel can be enqueued on its own stream after the flow has been enqueued on the ctx stream.
int plv_line_edges_fork(plv_ctx *ctx) {
  if (!ctx) return PLV_E_BADARG;
  if (!plv::knob(plv::PLV_KNOB_EDGES_SIDE)) return 1;  // default: on the ctx stream, in front of the flow (plv_ctx.hpp "Measurement knobs")
  LineTracker *T = ltr(ctx, false);
  std::lock_guard<std::mutex> lk(T->mtx);
  if (T->walk_on_device) return 1;
  if (!T->edge_stream) {
    PLV_HIP_CHECK(hipStreamCreateWithFlags(&T->edge_stream, hipStreamNonBlocking));
    PLV_HIP_CHECK(hipEventCreateWithFlags(&T->pyr_done, hipEventDisableTiming));
  }
  PLV_HIP_CHECK(hipEventRecord(T->pyr_done, ctx->stream));
  T->edge_fork = true;
  return PLV_OK;
}

// The tracker feed's hook into the image feed (plv_ctx::edges_hook): plv_line_detect_launch for the image being fed, between its
// histogram and its pyramid — the edge kernel equalises the raw image itself (canny_kernel), the worker gets the maps two launches
// earlier, and the flow starts when it always did.  Host-walk configuration without hysteresis only (the shipped one).
extern "C" void plv_line_edges_early(plv_ctx *ctx, const uint8_t *d_raw, int W, int H, const unsigned *d_hist) {
  LineTracker *T = ltr(ctx, false);
  std::lock_guard<std::mutex> lk(T->mtx);
  if (T->walk_on_device || !T->prefetch || ctx->cfg.canny_th1 != ctx->cfg.canny_th2) return;
  T->early_raw = d_raw, T->early_hist = d_hist, T->early_w = W, T->early_h = H;
  std::vector<float> none;
  if (detect(ctx, T, PLV_PYR_CUR, none, true) == PLV_OK) ctx->edges_hook_fired = true;
  T->early_raw = nullptr, T->early_hist = nullptr;
}

// (test aid, plv_decision_trace) the last line update's batch: ids [n] as plv_camera_update_lines returned them and vals [n][3] = chi2,
// the threshold it was held against, the norm of the projected residual (NaN: the line did not reach the gate)
int plv_last_line_decisions(plv_ctx *ctx, uint64_t *ids, double *vals, int cap, int *n) {
  if (!ctx || !n || cap < 0 || (cap > 0 && (!ids || !vals))) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  *n = (int)T->dec_ids.size();
  if (cap == 0) return PLV_OK;
  if (*n > cap) return PLV_E_CAPACITY;
  std::copy(T->dec_ids.begin(), T->dec_ids.end(), ids);
  std::copy(T->dec_vals.begin(), T->dec_vals.end(), vals);
  return PLV_OK;
}

int plv_line_detect_launch(plv_ctx *ctx, int which) {
  if (!ctx || (which != PLV_PYR_CUR && which != PLV_PYR_LAST)) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  LineTracker *T = ltr(ctx, false);  // (touches the detector's buffers only: a hand-back left behind may still wait)
  std::lock_guard<std::mutex> lk(T->mtx);
  if (T->walk_on_device) return PLV_OK;  // nothing to overlap: the device variant has no host stage
  std::vector<float> none;
  return detect(ctx, T, which, none, true);
}

int plv_line_detect_finish(plv_ctx *ctx, int which) {
  if (!ctx || (which != PLV_PYR_CUR && which != PLV_PYR_LAST)) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  T->cached_which = -1;
  TRY(detect(ctx, T, which, T->cached));
  T->cached_which = which;
  T->cached_fed = plv_front_fed_count(ctx);
  return PLV_OK;
}

int plv_line_tracker_feed(plv_ctx *ctx, double timestamp, const double *vps) {
  if (!ctx || !vps) return PLV_E_BADARG;
  // the point tracker's current observations (REF :106-107, :131-134)
  int np = 0;
  TRY(plv_tracker_last(ctx, nullptr, nullptr, 1 << 30, &np));
  std::vector<float> pts(2 * (size_t)std::max(np, 1));
  std::vector<uint64_t> pids((size_t)std::max(np, 1));
  TRY(plv_tracker_last(ctx, pts.data(), pids.data(), np, &np));
  return plv_line_tracker_feed_points(ctx, timestamp, vps, np, pts.data(), pids.data());
}

static int feed_points_impl(plv_ctx *ctx, LineTracker *T, double timestamp, const double *vps, int np, const float *pts, const uint64_t *pids, const double *K8);
int plv_line_tracker_feed_points(plv_ctx *ctx, double timestamp, const double *vps, int np, const float *pts, const uint64_t *pids) {
  if (!ctx || !vps || np < 0 || (np > 0 && (!pts || !pids))) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  return feed_points_impl(ctx, T, timestamp, vps, np, pts, pids, ctx->cfg.intrinsics);
}

// TrackLSD::feed_monocular after the histogram equalisation, on the tracker state of T (the caller holds T->mtx or is the worker).
// K8: the camera model of feed_measurement's time (REF: UpdaterCamera.cpp:77-116 runs before try_update refreshes it,
// StateHelper.cpp:163-168) — a snapshot when the call runs on the worker next to the point update.
static int feed_points_impl(plv_ctx *ctx, LineTracker *T, double timestamp, const double *vps, int np, const float *pts, const uint64_t *pids, const double *K8) {
  plv::HostPhase ph_all("line_tracker_feed: whole call");
  std::vector<float> lines;
  if (T->cached_which == PLV_PYR_CUR && T->cached_fed == plv_front_fed_count(ctx)) {
    lines.swap(T->cached);  // detected ahead of time for this very frame (plv_line_detect_finish)
    T->cached_which = -1;
  } else {
    TRY(detect(ctx, T, PLV_PYR_CUR, lines));
  }
  const bool timing = plv::knob(plv::PLV_KNOB_LINE_TIMING);
  auto F0 = std::chrono::steady_clock::now();
  const int nl = (int)lines.size() / 4;
  plv::counters().lines_detected += (unsigned long long)nl;
  std::vector<uint64_t> ids(nl);
  for (int i = 0; i < nl; ++i) ids[i] = ++T->currid;  // REF :233-236
  Assign A;
  assign_points_parallel(&T->host, fit_threads().load(std::memory_order_relaxed), lines.data(), nl, pts, pids, np, A);
  auto F1 = std::chrono::steady_clock::now();
  const int nk = (int)A.kept.size();
  std::vector<float> fl(4 * (size_t)nk);
  std::vector<uint64_t> fid(nk);
  for (int q = 0; q < nk; ++q) {
    std::copy(lines.begin() + 4 * A.kept[q], lines.begin() + 4 * A.kept[q] + 4, fl.begin() + 4 * q);
    fid[q] = ids[A.kept[q]];
  }
  const bool first = T->lines_last.empty();  // REF :100 (first frame or lost everything: no matching, no DB update)
  auto F_match = F1, F_und = F1;
  if (!first) {
    std::vector<int> match((size_t)std::max(nk, 1));
    match_lines_parallel(&T->host, fit_threads().load(std::memory_order_relaxed), fl.data(), nk, A.rel_ptr.data(), A.rel_id.data(), T->lines_last.data(),
                         (int)T->ids_last.size(), T->rel_ptr_last.data(), T->rel_id_last.data(), match.data());
    for (int q = 0; q < nk; ++q)
      if (match[q] >= 0) fid[q] = (uint64_t)(int)T->ids_last[match[q]];  // REF :153-158 (`int id`)
    if (timing) F_match = std::chrono::steady_clock::now();
    // CamBase::undistort_line: both end points through undistort_f.  A few dozen points: the arithmetic of undistort_kernel
    // (radtan_core.hpp, bit-identical on host and device) run here instead of a launch + copy + synchronisation round trip
    std::vector<float> un(4 * (size_t)std::max(nk, 1));
    for (int q = 0; q < 2 * nk; ++q)
      undistort_radtan(K8, fl[2 * (size_t)q], fl[2 * (size_t)q + 1], un[2 * (size_t)q], un[2 * (size_t)q + 1]);
    if (timing) F_und = std::chrono::steady_clock::now();
    for (int q = 0; q < nk; ++q) {
      const int D = plv_line_classification(fl.data() + 4 * q, vps);
      auto ins = T->db.try_emplace(fid[q]);
      const bool is_new = ins.second;
      LineTrack &tr = ins.first->second;
      if (is_new) {
        tr.D = D;  // REF LineFeatureDatabase.cpp:62-63: only a new feature takes D
        tr.t.reserve(32);
        tr.uv.reserve(128);
        tr.uvn.reserve(128);
      }
      tr.t.push_back(timestamp);
      tr.uv.insert(tr.uv.end(), fl.begin() + 4 * q, fl.begin() + 4 * q + 4);
      tr.uvn.insert(tr.uvn.end(), un.begin() + 4 * q, un.begin() + 4 * q + 4);
      for (int p = A.rel_ptr[q]; p < A.rel_ptr[q + 1]; ++p) tr.points.push_back((int)A.rel_id[p]);
    }
  }
  T->lines_last.swap(fl);
  T->ids_last.swap(fid);
  T->rel_ptr_last = A.rel_ptr;
  T->rel_id_last = A.rel_id;
  if (timing) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "line feed: kept-line copy + match %.1f us, undistort %.1f us, classify + store -> end %.1f us\n", us(F1, F_match), us(F_match, F_und),
            us(F_und, std::chrono::steady_clock::now()));
    fprintf(stderr, "line feed: assignment %.1f us (%d lines x %d points), match + undistort + classify + store %.1f us (%d kept)\n", us(F0, F1), nl, np,
            us(F1, std::chrono::steady_clock::now()), nk);
  }
  return PLV_OK;
}

// (internal) plv_line_tracker_feed_async with the frame's tracked points handed in (pts / pids = what plv_tracker_last returns once the
// point tracker's feed is over): the point tracker's feed posts the line feed through this the moment its point list stands, in front
// of its own database update (plv_camera_frame, round 6: the line worker's feed is the longer path of the frame)
int plv_line_tracker_feed_async_points(plv_ctx *ctx, double timestamp, const double *vps, int np, const float *pts, const uint64_t *pids) {
  if (!ctx || !vps || np < 0 || (np > 0 && (!pts || !pids))) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  LineTracker::FeedJob &F = T->feed;
  F.pts.assign(pts, pts + 2 * (size_t)np);
  F.pids.assign(pids, pids + (size_t)np);
  const bool detecting = !T->walk_on_device && T->worker.joinable() && T->pending_which == PLV_PYR_CUR && T->pending_fed == plv_front_fed_count(ctx);
  if (!detecting) {  // no detection of this frame on the worker: nothing to overlap with, and the detector's HIP calls stay on this thread
    F.pool_on = false;
    F.rc = feed_points_impl(ctx, T, timestamp, vps, np, F.pts.data(), F.pids.data(), ctx->cfg.intrinsics);
    return F.rc;
  }
  g_feed_impl = feed_points_impl;
  F.ctx = ctx;
  F.timestamp = timestamp;
  std::copy(vps, vps + 6, F.vps);
  std::copy(ctx->cfg.intrinsics, ctx->cfg.intrinsics + 8, F.K8);
  F.rc = PLV_OK;
  {
    std::lock_guard<std::mutex> lk2(T->jm);
    T->feed_state = 1;
    T->feed_posted = std::chrono::steady_clock::now();
  }
  T->jcv.notify_all();
  return PLV_OK;
}

int plv_line_tracker_feed_async(plv_ctx *ctx, double timestamp, const double *vps) {
  if (!ctx || !vps) return PLV_E_BADARG;
  int np = 0;
  TRY(plv_tracker_last(ctx, nullptr, nullptr, 1 << 30, &np));
  std::vector<float> pts(2 * (size_t)std::max(np, 1));
  std::vector<uint64_t> pids((size_t)std::max(np, 1));
  TRY(plv_tracker_last(ctx, pts.data(), pids.data(), np, &np));
  return plv_line_tracker_feed_async_points(ctx, timestamp, vps, np, pts.data(), pids.data());
}

int plv_line_tracker_feed_wait(plv_ctx *ctx) {
  if (!ctx) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);  // joins
  return T->feed.rc;
}

int plv_line_tracker_last(plv_ctx *ctx, float *lines, uint64_t *ids, int cap, int *n) {
  if (!ctx || !n) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  *n = (int)T->ids_last.size();
  if (*n > cap) return PLV_E_CAPACITY;
  if (lines) std::copy(T->lines_last.begin(), T->lines_last.end(), lines);
  if (ids) std::copy(T->ids_last.begin(), T->ids_last.end(), ids);
  return PLV_OK;
}

int plv_line_db_size(plv_ctx *ctx) {
  if (!ctx) return 0;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  return (int)T->db.size();
}

int plv_line_db_ids(plv_ctx *ctx, uint64_t *ids, int cap, int *n) {
  if (!ctx || !n) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  std::vector<uint64_t> v;
  for (const auto &kv : T->db) v.push_back(kv.first);
  std::sort(v.begin(), v.end());
  *n = (int)v.size();
  if (*n > cap) return PLV_E_CAPACITY;
  if (ids) std::copy(v.begin(), v.end(), ids);
  return PLV_OK;
}

// CSR export of the chosen line tracks in the layout of plv_line_tracks (+ the assigned point ids)
int plv_line_db_export_tracks(plv_ctx *ctx, const uint64_t *ids, int n_ids, int *obs_ptr, double *obs_time, float *seg_uv,
                              float *seg_uvn, int obs_cap, int *D, int *pts_ptr, int *pt_ids, int pts_cap) {
  if (!ctx || !ids || !obs_ptr) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  int no = 0, npt = 0;
  obs_ptr[0] = 0;
  if (pts_ptr) pts_ptr[0] = 0;
  for (int i = 0; i < n_ids; ++i) {
    auto it = T->db.find(ids[i]);
    if (it != T->db.end()) {
      const LineTrack &tr = it->second;
      const int m = (int)tr.t.size();
      if (no + m > obs_cap) return PLV_E_CAPACITY;
      if (obs_time) std::copy(tr.t.begin(), tr.t.end(), obs_time + no);
      if (seg_uv) std::copy(tr.uv.begin(), tr.uv.end(), seg_uv + 4 * (size_t)no);
      if (seg_uvn) std::copy(tr.uvn.begin(), tr.uvn.end(), seg_uvn + 4 * (size_t)no);
      no += m;
      if (D) D[i] = tr.D;
      if (pts_ptr) {
        if (npt + (int)tr.points.size() > pts_cap) return PLV_E_CAPACITY;
        if (pt_ids) std::copy(tr.points.begin(), tr.points.end(), pt_ids + npt);
        npt += (int)tr.points.size();
      }
    } else if (D) {
      D[i] = 0;
    }
    obs_ptr[i + 1] = no;
    if (pts_ptr) pts_ptr[i + 1] = npt;
  }
  return PLV_OK;
}

int plv_line_db_remove(plv_ctx *ctx, const uint64_t *ids, int n_ids) {
  if (!ctx || (n_ids > 0 && !ids)) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  for (int i = 0; i < n_ids; ++i) T->db.erase(ids[i]);
  return PLV_OK;
}

int plv_line_db_append_measurements(plv_ctx *ctx, uint64_t id, int n, const double *t, const float *seg_uv,
                                    const float *seg_uvn, int D, const int *point_ids, int n_pts) {
  if (!ctx || n < 0 || (n > 0 && (!t || !seg_uv || !seg_uvn)) || (n_pts > 0 && !point_ids)) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  std::lock_guard<std::mutex> lk(T->mtx);
  const bool is_new = T->db.find(id) == T->db.end();
  LineTrack &tr = T->db[id];
  if (is_new) tr.D = D;
  tr.t.insert(tr.t.end(), t, t + n);
  tr.uv.insert(tr.uv.end(), seg_uv, seg_uv + 4 * (size_t)n);
  tr.uvn.insert(tr.uvn.end(), seg_uvn, seg_uvn + 4 * (size_t)n);
  tr.points.insert(tr.points.end(), point_ids, point_ids + n_pts);
  return PLV_OK;
}

int plv_point_used_lookup(plv_ctx *ctx, uint64_t id, double *p);   // tracker_api.hip
void plv_point_used_cleanup(plv_ctx *ctx, double t_oldest);

static bool line_has_bounding_poses(const plv_state_view &st, double t) {  // as the kernels' bounding_start
  const int N = st.n_clones;
  if (N < 4) return false;
  const double *ct = st.clone_time;
  if (t < ct[0] - st.dt_exp || t > ct[N - 1] + st.dt_exp || t > ct[N - 1]) return false;
  for (int i = 0; i < N - 1; ++i)
    if (ct[i] - st.dt_exp <= t && t <= ct[i + 1] + st.dt_exp) return true;
  return false;
}


// (internal) plv_camera_try_update turns the deferral on around its line update; plv_tracker_feed* runs what was left behind
void plv_line_defer_finish(plv_ctx *ctx, int on) { ltr(ctx, false)->defer_finish = on != 0; }
void plv_line_run_deferred(plv_ctx *ctx) { (void)ltr(ctx); }

extern "C" int plv_point_chain_lookup(plv_ctx *ctx, uint64_t id);  // tracker_api.hip: index of a feature in the running point update's pool, or -1
extern "C" void plv_point_anchor_fill(plv_ctx *ctx, int Lp, const int *pt_ptr, const int *pt_ids, int chained, double *anchor, uint8_t *has);  // tracker_api.hip
extern "C" int plv_camera_get_line_features(plv_ctx *ctx, const plv_state_view *st);
static void line_give_back(std::unordered_map<uint64_t, LineTrack> &unused, const LineCand &c, size_t i) {
  LineTrack &u = unused[c.id];
  if (u.t.empty() && u.points.empty()) {
    u.D = c.tr.D;
    u.points = c.tr.points;  // copy_to_db copies the feature's point list
  }
  u.t.push_back(c.tr.t[i]);
  u.uv.insert(u.uv.end(), c.tr.uv.begin() + 4 * i, c.tr.uv.begin() + 4 * i + 4);
  u.uvn.insert(u.uvn.end(), c.tr.uvn.begin() + 4 * i, c.tr.uvn.begin() + 4 * i + 4);
}

namespace {
// cleanup_measurements on one track (REF LineHelper.cpp:549-551): observations older than the oldest clone go; true = nothing left
inline bool line_track_drop_before(LineTrack &tr, double t_oldest) {
  size_t keep = 0;
  for (size_t i = 0; i < tr.t.size(); ++i)
    if (!(tr.t[i] < t_oldest)) {
      if (keep != i) {
        tr.t[keep] = tr.t[i];
        std::copy(tr.uv.begin() + 4 * i, tr.uv.begin() + 4 * i + 4, tr.uv.begin() + 4 * keep);
        std::copy(tr.uvn.begin() + 4 * i, tr.uvn.begin() + 4 * i + 4, tr.uvn.begin() + 4 * keep);
      }
      ++keep;
    }
  tr.t.resize(keep);
  tr.uv.resize(4 * keep);
  tr.uvn.resize(4 * keep);
  return keep == 0;
}
// host work placed inside the line update's wait: the point database's hand-back, then the caller's own
struct LineWaitHook {
  plv_ctx *ctx;
  std::function<void()> *fn;
};
void line_wait_hook(void *arg) {
  LineWaitHook *h = (LineWaitHook *)arg;
  plv_tracker_run_deferred(h->ctx);
  if (h->fn && *h->fn) (*h->fn)();
}
}  // namespace

namespace {
void form_line_pool(LineTracker *T, const PoolArgs &A, LinePool &R) {
  R = LinePool();
  R.valid = true;
  R.t_prev_frame = A.t_prev_frame, R.state_time = A.state_time, R.dt = A.dt, R.n_clones = A.n_clones;
  R.t_oldest = A.t_oldest, R.t_oldest2 = A.t_oldest2;
  const double dt = R.dt, t_oldest = R.t_oldest, t_oldest2 = R.t_oldest2;
  const PoolArgs *opt = &A;  // (t_prev_frame, state_time)
  plv::HostPhase ph_scan("line pool: scan + take");
  {
    std::lock_guard<std::mutex> lk(T->mtx);
    R.db_size_before = (int)T->db.size();
    // (the tracks to take are remembered by position: extracting by iterator needs no second look-up — 180 hash look-ups were 8 of
    //  this stage's 15 us on the worker's path in front of the line launch)
    static thread_local std::vector<std::pair<uint64_t, decltype(T->db)::iterator>> take;
    take.clear();
    const double t_old = t_oldest2 - dt, t_new = opt->t_prev_frame - dt;
    for (auto it = T->db.begin(); it != T->db.end(); ++it) {  // REF LineHelper.cpp:33-38 (:74-130)
      bool older = false, newer = false;
      for (double t : it->second.t) {
        older = older || t < t_old;
        newer = newer || t > t_new;
      }
      if (older || !newer) take.emplace_back(it->first, it);
    }
    std::sort(take.begin(), take.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    R.pool.reserve(take.size());
    for (auto &tk : take) {
      auto node = T->db.extract(tk.second);  // (the track leaves the database with its node)
      R.pool.push_back(LineCand{tk.first, std::move(node.mapped())});
    }
  }
  ph_scan.stop();
  plv::HostPhase ph_trim("line pool: trim + sort");
  R.n_pool = (int)R.pool.size();
  R.unused.reserve(R.pool.size());
  size_t kept_cands = 0;  // (candidates that stay are moved down once: erasing from the middle of the vector shifted the rest every time)
  for (size_t ci = 0; ci < R.pool.size(); ++ci) {  // REF :652-682 (hard-coded 0.01 s margins)
    LineCand *it = &R.pool[ci];
    LineTrack &tr = it->tr;
    size_t keep = 0;
    for (size_t i = 0; i < tr.t.size(); ++i) {
      const double tm = tr.t[i] + dt;
      if (tm > opt->state_time + 0.01) {
        line_give_back(R.unused, *it, i);
        continue;
      }
      if (tm < t_oldest - 0.01) continue;
      if (keep != i) {
        tr.t[keep] = tr.t[i];
        std::copy(tr.uv.begin() + 4 * i, tr.uv.begin() + 4 * i + 4, tr.uv.begin() + 4 * keep);
        std::copy(tr.uvn.begin() + 4 * i, tr.uvn.begin() + 4 * i + 4, tr.uvn.begin() + 4 * keep);
      }
      ++keep;
    }
    tr.t.resize(keep);
    tr.uv.resize(4 * keep);
    tr.uvn.resize(4 * keep);
    if (keep >= 2) {
      if (kept_cands != ci) R.pool[kept_cands] = std::move(*it);
      ++kept_cands;
    }
  }
  R.pool.resize(kept_cands);
  // REF :640 sort by track length, long tracks first, ties in the order they stand (ascending id): the order is found on (length,
  // position) pairs and every candidate moved once — a stable sort of the candidates themselves moves each ~8 times, ~120 bytes a move
  static thread_local std::vector<std::pair<int, int>> ord;
  ord.clear();
  for (size_t i = 0; i < R.pool.size(); ++i) ord.emplace_back(-(int)R.pool[i].tr.t.size(), (int)i);
  std::sort(ord.begin(), ord.end());
  std::vector<LineCand> sorted;
  sorted.reserve(R.pool.size());
  for (const auto &o : ord) sorted.push_back(std::move(R.pool[(size_t)o.second]));
  R.pool.swap(sorted);
}
}  // namespace

// a pool that was formed ahead of time and is not going to be used: everything goes back where it came from
namespace {
void discard_line_pool(LineTracker *T) {
  LinePool &R = T->pool_prep;
  if (!R.valid) return;
  std::lock_guard<std::mutex> lk(T->mtx);
  auto put = [&](uint64_t id, LineTrack &tr) {
    const bool is_new = T->db.find(id) == T->db.end();
    LineTrack &d = T->db[id];
    if (is_new) {
      d = std::move(tr);
      return;
    }
    // the database has the id again (a feed re-created it with newer observations while the pool held the older ones): the track
    // stays ordered in time (appended, the returning older observations ended up behind the newer ones)
    const size_t na = d.t.size(), nb = tr.t.size();
    std::vector<std::pair<double, std::pair<int, int>>> order;  // (time, (source, index))
    order.reserve(na + nb);
    for (size_t i = 0; i < na; ++i) order.push_back({d.t[i], {0, (int)i}});
    for (size_t i = 0; i < nb; ++i) order.push_back({tr.t[i], {1, (int)i}});
    std::stable_sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    LineTrack m;
    m.D = d.D, m.points = d.points;
    for (const auto &e : order) {
      const LineTrack &src = e.second.first ? tr : d;
      const size_t i = (size_t)e.second.second;
      m.t.push_back(src.t[i]);
      m.uv.insert(m.uv.end(), src.uv.begin() + 4 * i, src.uv.begin() + 4 * i + 4);
      m.uvn.insert(m.uvn.end(), src.uvn.begin() + 4 * i, src.uvn.begin() + 4 * i + 4);
    }
    d = std::move(m);
  };
  for (auto &c : R.pool) put(c.id, c.tr);
  for (auto &kv : R.unused) put(kv.first, kv.second);
  R = LinePool();
}
}  // namespace

// (internal, plv_camera_try_update) forms the line pool now if the line feed of this frame has finished — polled while the point
// update runs on the device.  Never blocks: returns 0 while the feed is still on the worker (try again), 1 when the pool is formed or
// cannot be formed ahead of time (plv_camera_update_lines then forms it).
int plv_line_pool_prepare(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt) {
  if (plv::knob(plv::PLV_KNOB_POOL_LATE) || !ctx || !st || !opt || opt->cpi || st->n_clones < 2 || st->dt_state_id >= 0) return 1;  // (a calibrated time offset moves the window test)
  LineTracker *T;
  {
    std::lock_guard<std::mutex> lk(g_mtx);
    auto it = g_lt.find(ctx);
    if (it == g_lt.end()) return 1;
    T = it->second;
  }
  {
    std::lock_guard<std::mutex> lk(T->jm);
    if (T->feed_state == 1) return 0;  // still running
  }
  T = ltr(ctx);  // (joins a finished feed, runs a hand-back left behind)
  if (T->feed.rc != PLV_OK) return 1;
  const PoolArgs A = PoolArgs::of(st, opt);
  const LinePool &R = T->pool_prep;
  if (R.valid && R.t_prev_frame == A.t_prev_frame && R.state_time == A.state_time && R.dt == A.dt && R.n_clones == A.n_clones && R.t_oldest == A.t_oldest &&
      R.t_oldest2 == A.t_oldest2)
    return 1;  // (the worker formed it at the end of the feed: plv_line_feed_pool_args)
  plv::HostPhase ph("update_lines: pool formed inside the point update's wait");
  discard_line_pool(T);
  form_line_pool(T, A, T->pool_prep);
  return 1;
}
// (internal, plv_camera_frame) the frame's line feed is about to be posted and a line update follows: the worker forms that update's
// pool at the end of the feed.  Same conditions as plv_line_pool_prepare.
void plv_line_feed_pool_args(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt) {
  LineTracker *T = ltr(ctx);
  T->feed.pool_on = false;
  if (plv::knob(plv::PLV_KNOB_POOL_LATE) || !st || !opt || opt->cpi || st->n_clones < 2 || st->dt_state_id >= 0) return;
  T->feed.pool_args = PoolArgs::of(st, opt);
  T->feed.pool_on = true;
}
void plv_line_pool_discard(plv_ctx *ctx) { discard_line_pool(ltr(ctx, false)); }
int plv_line_db_size_after_feed(plv_ctx *ctx) {
  LineTracker *T = ltr(ctx);
  if (T->pool_prep.valid) return T->pool_prep.db_size_before;
  if (T->ujob.pending && T->ujob.LP.valid) return T->ujob.LP.db_size_before;  // (the chained first half holds the pool)
  std::lock_guard<std::mutex> lk(T->mtx);
  return (int)T->db.size();
}

// LineHelper::get_line_features' place in try_update (REF: UpdaterCamera.cpp:148-152: after get_features, before msckf_update's
// correction reaches the state): records the state the line pool is to be triangulated on.  The pool itself (LineHelper.cpp:33-44)
// and its triangulation (:45-63) are formed inside the following plv_camera_update_lines, on this state; nothing they read changes
// in between (the line database, point_used), so the result is what the reference computes at this point.
int plv_camera_get_line_features(plv_ctx *ctx, const plv_state_view *st) {
  if (!ctx || !st || st->n_clones < 1 || !st->clone_time || !st->clone_R || !st->clone_p || !st->clone_state_id) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx, false);
  LineTracker::TriState &S = T->tri_state;
  const int N = st->n_clones;
  S.clone_time.assign(st->clone_time, st->clone_time + N);
  S.clone_R.assign(st->clone_R, st->clone_R + 9 * (size_t)N);
  S.clone_p.assign(st->clone_p, st->clone_p + 3 * (size_t)N);
  S.clone_id.assign(st->clone_state_id, st->clone_state_id + N);
  S.view = *st;
  S.view.clone_time = S.clone_time.data();
  S.view.clone_R = S.view.clone_R_fej = S.clone_R.data();  // (the triangulation reads estimates only)
  S.view.clone_p = S.view.clone_p_fej = S.clone_p.data();
  S.view.clone_state_id = S.clone_id.data();
  S.valid = true;
  return PLV_OK;
}

// First half of plv_camera_update_lines (see LinesJob).  st: the state of the pool's window tests and — unless `chained` — of the
// linearisation; st_tri: the state of the triangulation.  chained: plv_ctx::chain carries the caller's quaternions / covariance
// indices, the launch is enqueued behind the point update and linearises on st (+) that update's dx.
static int lines_first_half(plv_ctx *ctx, LineTracker *T, const plv_state_view *st, const plv_state_view *st_tri, const plv_update_options *opt, int cap,
                            LinesJob &J, bool chained) {
  J = LinesJob();
  J.cap = cap, J.n_clones = st->n_clones, J.state_time = opt->state_time, J.t_prev_frame = opt->t_prev_frame;
  const auto U0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  plv::HostPhase ph_pool("update_lines: pool + staging");
  const double dt = st->cam_dt, t_oldest = st->clone_time[0];
  auto has_bounding = bounding_memo([st](double tq) { return line_has_bounding_poses(*st, tq); });
  typedef LineCand Cand;
  LinePool &LP = J.LP;
  if (T->pool_prep.valid && T->pool_prep.t_prev_frame == opt->t_prev_frame && T->pool_prep.state_time == opt->state_time && T->pool_prep.dt == dt &&
      T->pool_prep.n_clones == st->n_clones && T->pool_prep.t_oldest == t_oldest && T->pool_prep.t_oldest2 == st->clone_time[1]) {
    LP = std::move(T->pool_prep);  // formed while the point update was running (plv_line_pool_prepare)
    T->pool_prep = LinePool();
  } else {
    discard_line_pool(T);
    form_line_pool(T, PoolArgs::of(st, opt), LP);
  }
  std::vector<Cand> &pool = LP.pool;
  std::unordered_map<uint64_t, LineTrack> &unused = LP.unused;
  auto give_back = [&](const Cand &c, size_t i) { line_give_back(unused, c, i); };
  plv::HostPhase ph_p1("update_lines: pool a (scan + take) done -> b (trim + sort)");
  if (pool.empty()) {
    J.stage = LinesJob::EMPTY;
    return PLV_OK;
  }
  // ---- triangulate every pool line (REF :45-63; get_imu_poses drops views without bounding clones)
  const int Lp = J.Lp = (int)pool.size();
  // ---- use_imu_res: poses from the CPI table (plv_update_options::cpi); views it cannot serve go back to the database
  J.cpiR.resize(opt->cpi ? Lp : 0), J.cpip.resize(opt->cpi ? Lp : 0), J.cpiQ.resize(opt->cpi ? Lp : 0), J.cpiC.resize(opt->cpi ? Lp : 0);
  auto &cpiR = J.cpiR, &cpip = J.cpip, &cpiQ = J.cpiQ;
  auto &cpiC = J.cpiC;
  const bool imu_cov = opt->cpi && opt->cpi->Q && st->use_imu_cov && !st->use_pol_cov;
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
      J.stage = LinesJob::FAILED, J.rc = rc0;
      return PLV_OK;
    }
    size_t o = 0;
    for (int l = 0; l < Lp; ++l) {
      Cand &c = pool[l];
      LineTrack kept;
      kept.D = c.tr.D;
      kept.points = c.tr.points;
      for (size_t i = 0; i < c.tr.t.size(); ++i, ++o) {
        if (!okq[o]) {
          give_back(c, i);
          continue;
        }
        kept.t.push_back(c.tr.t[i]);
        kept.uv.insert(kept.uv.end(), c.tr.uv.begin() + 4 * i, c.tr.uv.begin() + 4 * i + 4);
        kept.uvn.insert(kept.uvn.end(), c.tr.uvn.begin() + 4 * i, c.tr.uvn.begin() + 4 * i + 4);
        cpiR[l].insert(cpiR[l].end(), &Rq[9 * o], &Rq[9 * o] + 9);
        cpip[l].insert(cpip[l].end(), &pq[3 * o], &pq[3 * o] + 3);
        if (imu_cov) {
          cpiQ[l].insert(cpiQ[l].end(), &Qq[36 * o], &Qq[36 * o] + 36);
          cpiC[l].push_back(Cq[o]);
        }
      }
      c.tr = std::move(kept);
    }
  }
  ph_p1.stop();
  plv::HostPhase ph_p2("update_lines: pool c (anchors + arrays + valid)");
  J.ptr.assign(Lp + 1, 0), J.D.resize(Lp), J.anchor.assign(3 * (size_t)Lp, 0.0), J.has.assign(Lp, 0);
  std::vector<int> &ptr = J.ptr, &D = J.D;
  plv_ctx::ChainState &ch = ctx->chain;
  {
    std::vector<int> &pp = J.pt_ptr, &pi = J.pt_ids;
    pp.assign(1, 0), pi.clear();
    for (int l = 0; l < Lp; ++l) {
      ptr[l + 1] = ptr[l] + (int)pool[l].tr.t.size();
      D[l] = pool[l].tr.D;
      pi.insert(pi.end(), pool[l].tr.points.begin(), pool[l].tr.points.end());
      pp.push_back((int)pi.size());
    }
    // first triangulated point of every line (REF :233-247) — chained: the candidates, the launch decides (the point update whose
    // triangulation may (re)write point_used is still running)
    plv_point_anchor_fill(ctx, Lp, pp.data(), pi.data(), chained ? 1 : 0, J.anchor.data(), J.has.data());
  }
  const int nobs = J.nobs = ptr[Lp];
  if (nobs == 0) {
    J.stage = LinesJob::EMPTY;
    return PLV_OK;
  }
  J.ot.resize(nobs), J.uv.resize(4 * (size_t)nobs), J.uvn.resize(4 * (size_t)nobs);
  for (int l = 0; l < Lp; ++l) {
    const LineTrack &tr = pool[l].tr;
    std::copy(tr.t.begin(), tr.t.end(), J.ot.begin() + ptr[l]);
    std::copy(tr.uv.begin(), tr.uv.end(), J.uv.begin() + 4 * (size_t)ptr[l]);
    std::copy(tr.uvn.begin(), tr.uvn.end(), J.uvn.begin() + 4 * (size_t)ptr[l]);
  }
  plv_line_tracks all{};
  all.n_lines = Lp;
  all.obs_ptr = ptr.data();
  all.obs_time = J.ot.data();
  all.seg_uv = J.uv.data();
  all.seg_uvn = J.uvn.data();
  all.D = D.data();
  all.anchor_pt = J.anchor.data();
  all.has_pt = J.has.data();
  if (opt->cpi) {
    for (int l = 0; l < Lp; ++l) {
      J.allR.insert(J.allR.end(), cpiR[l].begin(), cpiR[l].end());
      J.allp.insert(J.allp.end(), cpip[l].begin(), cpip[l].end());
    }
    all.res_R = J.allR.data();
    all.res_p = J.allp.data();
  }
  J.valid_n.assign(Lp, 0);
  for (int l = 0; l < Lp; ++l) {
    for (double t : pool[l].tr.t) J.valid_n[l] += has_bounding(t + dt);
    J.most_valid = std::max(J.most_valid, J.valid_n[l]);
  }
  J.cols.resize(ctx->cfg.max_state_dim > 0 ? ctx->cfg.max_state_dim : 1024);
  // ---- one submission (see plv_camera_update_points): line triangulation, the selection below, Jacobians, null space, gate,
  // compression and EKFUpdate back to back on the stream, one synchronisation.  CPI poses and over-long tracks take the two-step route.
  const bool fused = !opt->cpi && J.most_valid <= opt->max_obs;
  J.us_pool = since(U0);
  ph_p2.stop();
  ph_pool.stop();
  if (!fused) {
    J.stage = LinesJob::TWO_STEP;
    return PLV_OK;
  }
  J.flags.resize(Lp);
  bool any = false;
  for (int l = 0; l < Lp; ++l) any = (J.flags[l] = J.valid_n[l] >= 2) || any;
  if (!any) {
    J.stage = LinesJob::FUSED_NOTHING;
    return PLV_OK;
  }
  plv::HostPhase ph_cols("update_lines: columns");
  int rc = plv_line_jacobian_columns(st, &all, J.cols.data(), (int)J.cols.size(), &J.k);
  ph_cols.stop();
  if (rc == PLV_OK && J.k > 0) {
    plv::HostPhase ph_sub("update_lines: fused submit (gate prepare + stage + upload + launch)");
    ctx->gate_rows_hint = 2 * J.most_valid;
    ch.on = chained;
    rc = plv_lines_update_fused_submit(ctx, st, st_tri, &all, J.flags.data(), cap, J.k, J.cols.data(), 2 * opt->max_obs, st->sigma_pix * st->sigma_pix,
                                       opt->chi2_mult);
    ch.on = false;
    if (rc == PLV_OK) {
      J.stage = LinesJob::FUSED_LAUNCHED;
      return PLV_OK;
    }
  }
  if (rc == PLV_OK) {  // (no column: nothing to linearise)
    J.stage = LinesJob::FUSED_NOTHING;
    return PLV_OK;
  }
  J.stage = LinesJob::FAILED, J.rc = rc;
  return PLV_OK;
}

// (internal, plv_camera_try_update) the chained first half: called inside the point update's wait once the frame's line feed has
// finished.  0: not possible now (the caller goes on as before), 1: the line launch is on the stream behind the point update.
int plv_camera_lines_submit_chained(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt, int cap) {
  if (!ctx || !st || !opt || opt->cpi || st->n_clones < 2 || opt->max_obs < 2 || st->dt_state_id >= 0 || !ctx->chain.ready) return 0;
  LineTracker *T = ltr(ctx);
  if (T->ujob.pending || T->feed.rc != PLV_OK) return 0;
  plv::frame_mark("@ chained first half starts");
  plv::HostPhase ph("update_lines: first half inside the point update's wait (chained)");
  T->tri_state.valid = false;  // (REF UpdaterCamera.cpp:148-152: the pool is triangulated on st — the one state the chained launch stages; the corrected one it forms itself)
  (void)lines_first_half(ctx, T, st, st, opt, cap, T->ujob, true);
  if (T->ujob.stage == LinesJob::TWO_STEP) {  // (tracks longer than the batch rows: the two-step route needs the corrected state on the host)
    T->pool_prep = std::move(T->ujob.LP);     // the pool as it was formed goes back to where plv_camera_update_lines looks for it
    T->pool_prep.valid = true;
    T->ujob = LinesJob();
    return 0;
  }
  T->ujob.pending = true;
  plv::frame_mark("@ chained first half done (line launch enqueued)");
  if (T->ujob.stage == LinesJob::FUSED_LAUNCHED) ++plv::counters().chained;
  return 1;
}

int plv_camera_lines_job_pending(plv_ctx *ctx) { return ltr(ctx, false)->ujob.pending ? 1 : 0; }
// (internal) a chained first half whose second half will not run (the point update failed or was run again): its launch is waited
// for.  keep_pool = 0: everything it took out of the line database goes back (a failing exit: nothing of the frame may outlive the
// call); keep_pool = 1: the pool stays formed for plv_camera_get_line_features / plv_camera_update_lines of the same frame, which go on
// the unchained way (formed again it would hold the same candidates, but report the pool size after the single-view tracks were
// dropped: plv_update_result::n_pool of such a frame was short by those — tests/test_gpu_kaist_replay.py, round 6)
void plv_camera_lines_job_abort2(plv_ctx *ctx, int keep_pool) {
  LineTracker *T = ltr(ctx, false);
  if (!T->ujob.pending) return;
  (void)plv::stream_sync(ctx->stream);
  ctx->gate_stage.on = 0, ctx->gate_stage_taken = false;
  T->pool_prep = std::move(T->ujob.LP);
  T->pool_prep.valid = true;
  if (!keep_pool) discard_line_pool(T);
  T->ujob = LinesJob();
}
void plv_camera_lines_job_abort(plv_ctx *ctx) { plv_camera_lines_job_abort2(ctx, 0); }

int plv_camera_update_lines(plv_ctx *ctx, const plv_state_view *st, const plv_update_options *opt, double *dx,
                            plv_update_result *res, uint64_t *line_ids, uint8_t *accepted_out, double *lines_out, int cap) {
  if (!ctx || !st || !opt || !dx || !res || st->n_clones < 2 || opt->max_obs < 2) return PLV_E_BADARG;
  LineTracker *T = ltr(ctx);
  *res = plv_update_result{0, 0, 0, 0, 0, PLV_OK, 0, 0, 0};
  const bool timing = plv::knob(plv::PLV_KNOB_UPDATE_TIMING);
  plv::NsScope ns_lines(plv::counters().lines_ns);
  plv::HostPhase ph_all("update_lines: whole call");
  plv::RoctxRange rx_line("[Time-Cam] LINE update");
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  LinesJob Jlocal;
  const bool resumed = T->ujob.pending && T->ujob.cap == cap && T->ujob.n_clones == st->n_clones && T->ujob.state_time == opt->state_time &&
                       T->ujob.t_prev_frame == opt->t_prev_frame;
  if (T->ujob.pending && !resumed) {  // (a chained first half for other arguments than these: cannot be — its launch may be on the stream)
    plv::set_last_error("plv_camera_update_lines: a chained submission for another window is pending");
    return PLV_E_BADARG;
  }
  LinesJob &J = resumed ? T->ujob : Jlocal;
  if (!resumed) {
    // the state of the triangulation: what plv_camera_get_line_features recorded (same window), else the state handed in
    LineTracker::TriState tri_keep;
    std::swap(tri_keep, T->tri_state);
    T->tri_state.valid = false;
    if (tri_keep.valid) {  // (the vectors moved: re-point the view)
      tri_keep.view.clone_time = tri_keep.clone_time.data();
      tri_keep.view.clone_R = tri_keep.view.clone_R_fej = tri_keep.clone_R.data();
      tri_keep.view.clone_p = tri_keep.view.clone_p_fej = tri_keep.clone_p.data();
      tri_keep.view.clone_state_id = tri_keep.clone_id.data();
      if (tri_keep.view.n_clones != st->n_clones || memcmp(tri_keep.clone_time.data(), st->clone_time, 8 * (size_t)st->n_clones) != 0) tri_keep.valid = false;
    }
    const plv_state_view *st_tri = tri_keep.valid ? &tri_keep.view : st;
    TRY(lines_first_half(ctx, T, st, st_tri, opt, cap, J, false));
    if (J.stage == LinesJob::TWO_STEP) {  // (triangulation as its own synchronous call, on st_tri while it is in scope)
      J.ok_two_step.resize(J.Lp), J.lg_two_step.resize(6 * (size_t)J.Lp);
      plv_line_tracks all{};
      all.n_lines = J.Lp, all.obs_ptr = J.ptr.data(), all.obs_time = J.ot.data(), all.seg_uv = J.uv.data(), all.seg_uvn = J.uvn.data();
      all.D = J.D.data(), all.anchor_pt = J.anchor.data(), all.has_pt = J.has.data();
      if (opt->cpi) all.res_R = J.allR.data(), all.res_p = J.allp.data();
      const int rc2 = plv_triangulate_lines(ctx, st_tri, &all, J.lg_two_step.data(), J.ok_two_step.data());
      if (rc2 != PLV_OK) J.stage = LinesJob::FAILED, J.rc = rc2;
    }
  }
  T->ujob.pending = false;
  const double dt = st->cam_dt, t_oldest = st->clone_time[0];
  auto has_bounding = bounding_memo([st](double tq) { return line_has_bounding_poses(*st, tq); });
  typedef LineCand Cand;
  LinePool &LP = J.LP;
  std::vector<Cand> &pool = LP.pool;
  std::unordered_map<uint64_t, LineTrack> &unused = LP.unused;
  auto give_back = [&](const Cand &c, size_t i) { line_give_back(unused, c, i); };
  res->n_pool = LP.n_pool;
  auto give_back_all = [&](Cand &c) {
    if (unused.find(c.id) == unused.end()) {  // nothing of this line went back earlier: hand the track over as it is
      unused.emplace(c.id, std::move(c.tr));
      c.tr = LineTrack{};
      return;
    }
    for (size_t i = 0; i < c.tr.t.size(); ++i) give_back(c, i);
  };
  std::vector<int> lazy_back;  // pool candidates whose whole track returns to the database: moved there by the deferred hand-back
  // cleanup_measurements over the tracks that stayed in the database, placed inside the update's wait for the device (they do not
  // depend on its result); finish() then cleans only what returns
  bool db_scanned_early = false;
  std::function<void()> scan_db_early = [&]() {
    if (!opt->window_full || db_scanned_early) return;
    std::lock_guard<std::mutex> lk(T->mtx);
    for (auto it = T->db.begin(); it != T->db.end();) it = line_track_drop_before(it->second, t_oldest) ? T->db.erase(it) : std::next(it);
    db_scanned_early = true;
  };
  auto finish = [&](int rc) {
    plv::HostPhase ph_fin("update_lines: finish (counts, point_used clean-up, hand-back closure)");
    res->n_returned = (int)unused.size();
    for (int l : lazy_back) res->n_returned += unused.find(pool[l].id) == unused.end() ? 1 : 0;
    const bool window_full = opt->window_full != 0;
    const bool scanned = db_scanned_early;
    // REF :71 / cleanup_lines :545-546 append_new_measurements, then LineHelper.cpp:549-551, UpdaterCamera.cpp:186-188 (on every
    // try_update) cleanup_measurements(oldest clone).  Whole tracks (`whole`: candidates the update did not take and of which nothing
    // went back earlier) enter the database with one insertion; the cleanup touches every track, or — when the database was cleaned
    // inside the update's wait (scan_db_early below) — only the tracks that return now.
    auto hand_back = [T, window_full, t_oldest, scanned](std::unordered_map<uint64_t, LineTrack> &un, std::vector<Cand> *cands, const std::vector<int> *whole) {
      std::lock_guard<std::mutex> lk(T->mtx);
      auto put = [&](uint64_t id, LineTrack &tr) {
        auto ins = T->db.try_emplace(id);
        LineTrack &d = ins.first->second;
        if (ins.second) {
          d = std::move(tr);
        } else {
          d.t.insert(d.t.end(), tr.t.begin(), tr.t.end());
          d.uv.insert(d.uv.end(), tr.uv.begin(), tr.uv.end());
          d.uvn.insert(d.uvn.end(), tr.uvn.begin(), tr.uvn.end());
        }
        if (window_full && scanned && line_track_drop_before(d, t_oldest)) T->db.erase(ins.first);
      };
      if (whole)
        for (int l : *whole) {
          Cand &c = (*cands)[l];
          if (un.find(c.id) == un.end())
            put(c.id, c.tr);
          else  // (parts of it went back earlier: behind those, as give_back_all does)
            for (size_t i = 0; i < c.tr.t.size(); ++i) line_give_back(un, c, i);
        }
      for (auto &kv : un) put(kv.first, kv.second);
      if (window_full && !scanned)
        for (auto it = T->db.begin(); it != T->db.end();) it = line_track_drop_before(it->second, t_oldest) ? T->db.erase(it) : std::next(it);
    };
    // (point_used->cleanup_measurements is not deferred: it takes the point tracker's lock, which a feed in progress holds)
    if (window_full) plv_point_used_cleanup(ctx, t_oldest);
    if (T->defer_finish) {
      // plv_camera_try_update: nothing reads the line database before the next frame's feed; the hand-back (and the release of the
      // pooled tracks) runs in that frame's wait for the flow, or at the next call that reaches the tracker
      auto held = std::make_shared<std::unordered_map<uint64_t, LineTrack>>(std::move(unused));
      auto used_up = std::make_shared<std::vector<Cand>>(std::move(pool));
      auto lazy = std::make_shared<std::vector<int>>(std::move(lazy_back));
      T->deferred = [hand_back, held, used_up, lazy]() { hand_back(*held, used_up.get(), lazy.get()); };
      if (T->worker.joinable()) {  // the worker is idle (and still polling) at this point of the frame: it starts at once
        {
          std::lock_guard<std::mutex> lk(T->jm);
          T->defer_state = 1;
        }
        T->jcv.notify_all();
      }
    } else {
      hand_back(unused, &pool, &lazy_back);
      lazy_back.clear();
    }
    return rc;
  };
  std::fill(dx, dx + ctx->cov_n, 0.0);
  if (J.stage == LinesJob::EMPTY) return finish(PLV_OK);
  if (J.stage == LinesJob::FAILED) {
    for (Cand &c : pool) give_back_all(c);
    return finish(J.rc);
  }
  const int Lp = J.Lp, nobs = J.nobs;
  std::vector<int> &valid_n = J.valid_n, &cols = J.cols;
  auto &cpiR = J.cpiR, &cpip = J.cpip, &cpiQ = J.cpiQ;
  auto &cpiC = J.cpiC;
  const bool imu_cov = opt->cpi && opt->cpi->Q && st->use_imu_cov && !st->use_pol_cov;
  std::vector<double> lg(6 * (size_t)Lp);
  std::vector<uint8_t> ok(Lp);
  int k = J.k, n_rows = 0, rc = PLV_OK;
  const double us_pool = J.us_pool;
  plv::HostPhase ph_dev("update_lines: device submission + wait");
  const auto U1 = std::chrono::steady_clock::now();
  std::vector<uint8_t> acc_all(Lp, 0);
  bool fused_ran = false;
  if (J.stage == LinesJob::FUSED_LAUNCHED) {
    LineWaitHook hook{ctx, &scan_db_early};
    rc = plv_lines_update_fused_finish(ctx, st->sigma_pix * st->sigma_pix, opt->chi2_mult, lg.data(), ok.data(), acc_all.data(), &n_rows, dx,
                                       line_wait_hook, &hook);
    res->status = rc == PLV_E_NOT_PSD ? rc : PLV_OK;
    if (rc == PLV_E_NOT_PSD) {
      rc = PLV_OK;
      std::fill(dx, dx + ctx->cov_n, 0.0);
    }
    fused_ran = rc == PLV_OK;
    if (rc != PLV_OK) {
      for (Cand &c : pool) give_back_all(c);
      return finish(rc);
    }
  } else if (J.stage == LinesJob::FUSED_NOTHING) {
    std::fill(ok.begin(), ok.end(), 0);
  } else {  // TWO_STEP
    lg = J.lg_two_step;
    ok = J.ok_two_step;
  }
  const double us_dev = since(U1);
  ph_dev.stop();
  plv::frame_mark("@ line gate / update collected");
  plv::HostPhase ph_post("update_lines: selection + database");
  if (timing) fprintf(stderr, "update lines: pool + staging %.1f us (%d lines, %d observations), device submission + wait %.1f us\n", us_pool, Lp, nobs, us_dev);
  std::vector<int> sel;
  std::vector<int> n_skip(Lp, 0);  // usable observations a truncated track leaves out (its first ones)
  for (int l = 0; l < Lp; ++l) {
    const int valid = valid_n[l];
    if (!ok[l] || valid < 2 || (int)sel.size() >= cap) {
      lazy_back.push_back(l);  // (the whole track goes back: finish() does it, deferred when the caller allows)
      continue;
    }
    if (valid > opt->max_obs) {  // batch capacity (none in the reference): the last max_obs usable observations, counted in n_truncated
      n_skip[l] = valid - opt->max_obs;
      ++res->n_truncated;
    }
    sel.push_back(l);
  }
  res->n_msckf = (int)sel.size();
  plv::frame_mark("@ line selection loop done");
  if (sel.empty()) {
    if (fused_ran) std::fill(dx, dx + ctx->cov_n, 0.0);
    return finish(PLV_OK);
  }
  // ---- UpdaterCamera::lines_update
  const int L = (int)sel.size();
  std::vector<int> sptr(L + 1, 0);
  std::vector<double> st_t, sl(6 * (size_t)L), selR, selp, selQ;
  std::vector<int> selC;
  std::vector<float> suv;
  for (int q = 0; q < L; ++q) {
    const Cand &c = pool[sel[q]];
    int seen = 0;
    // (behind a fused launch the loop only hands back views without bounding clones: none when every view counted as usable)
    const bool nothing_to_do = fused_ran && valid_n[sel[q]] == (int)c.tr.t.size();
    for (size_t i = 0; !nothing_to_do && i < c.tr.t.size(); ++i) {
      if (!has_bounding(c.tr.t[i] + dt)) {
        give_back(c, i);
        continue;
      }
      if (seen++ < n_skip[sel[q]]) continue;
      if (fused_ran) continue;  // (the batch was built on the device: the two-step route's arrays are not needed)
      st_t.push_back(c.tr.t[i]);
      suv.insert(suv.end(), c.tr.uv.begin() + 4 * i, c.tr.uv.begin() + 4 * i + 4);
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
    std::copy(lg.begin() + 6 * (size_t)sel[q], lg.begin() + 6 * (size_t)sel[q] + 6, sl.begin() + 6 * (size_t)q);
    if (line_ids) line_ids[q] = c.id;
  }
  if (lines_out) std::copy(sl.begin(), sl.end(), lines_out);
  plv_line_tracks lt{};
  lt.n_lines = L;
  lt.obs_ptr = sptr.data();
  lt.obs_time = st_t.data();
  lt.seg_uv = suv.data();
  lt.line_FinG = sl.data();
  if (opt->cpi) {
    lt.res_R = selR.data();
    lt.res_p = selp.data();
    if (imu_cov) {
      lt.res_Q = selQ.data();
      lt.res_clone = selC.data();
    }
  }
  std::vector<uint8_t> acc(L, 0);
  if (fused_ran) {
    for (int q = 0; q < L; ++q) acc[q] = acc_all[sel[q]];
  } else {
    rc = plv_line_jacobian_columns(st, &lt, cols.data(), (int)cols.size(), &k);
    if (rc == PLV_OK) rc = plv_build_line_jacobians_resident(ctx, st, &lt, k, cols.data(), 2 * opt->max_obs);
    if (rc == PLV_OK) {
      rc = plv_msckf_update_resident(ctx, st->sigma_pix * st->sigma_pix, opt->chi2_mult, 0.0, acc.data(), &n_rows, dx);
      res->status = rc;
      if (rc == PLV_E_NOT_PSD) rc = PLV_OK;
    }
    if (rc != PLV_OK) {
      for (int q = 0; q < L; ++q) give_back_all(pool[sel[q]]);
      return finish(rc);
    }
  }
  res->n_rows = n_rows;
  if (ctx->decision_trace) {  // (plv_last_line_decisions: the gate's values of the lines that reached it, in the order of line_ids)
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const int Fg = ctx->dec_F_l;
    std::vector<double> gv(3 * (size_t)std::max(Fg, 1), nan);
    if (Fg > 0) PLV_HIP_CHECK(hipMemcpy(gv.data(), ctx->d_gate_dec_l.p, (size_t)Fg * 24, hipMemcpyDeviceToHost));
    ctx->dec_F_l = 0;
    T->dec_ids.resize(L), T->dec_vals.assign(3 * (size_t)L, nan);
    for (int q = 0; q < L; ++q) {
      const int gi = fused_ran ? sel[q] : q;
      T->dec_ids[q] = pool[sel[q]].id;
      if (gi < Fg) std::copy(gv.begin() + 3 * (size_t)gi, gv.begin() + 3 * (size_t)gi + 3, T->dec_vals.begin() + 3 * (size_t)q);
    }
  }
  plv::frame_mark("@ line arrays of the selected done");
  for (int q = 0; q < L; ++q) {
    res->n_accepted += acc[q];
    if (accepted_out) accepted_out[q] = acc[q];
    if (!acc[q]) {  // REF UpdaterCamera.cpp:441-444 copy_to_db(lbd_unused, line): gate failures only
      const Cand &c = pool[sel[q]];
      // (every view usable and nothing of the line handed back earlier — the usual case: the copy the reference makes view by view
      // is the track itself, returned whole by the hand-back like the candidates the update never took)
      const bool whole = valid_n[sel[q]] == (int)c.tr.t.size() && unused.find(c.id) == unused.end();  // (valid_n: the views with bounding clones)
      if (whole) {
        lazy_back.push_back(sel[q]);
        continue;
      }
      for (size_t i = 0; i < c.tr.t.size(); ++i)
        if (has_bounding(c.tr.t[i] + dt)) give_back(c, i);
    }
  }
  return finish(PLV_OK);
}

}  // extern "C"
