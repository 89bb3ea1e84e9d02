// plv_api.hip — the extern "C" boundary declared in include/plviwo.h (context + update side).
// Product code: no CPU fallback anywhere — without a gfx950 device every compute entry point
// returns PLV_E_NO_DEVICE.
#include <algorithm>
#include <cstdarg>
#include <dlfcn.h>
#include <cmath>
#include <mutex>

#include "plv_ctx.hpp"
#include "update_kernels.hpp"
#include "update_state.hpp"

namespace plv {

static thread_local char g_err[512] = "";
void set_last_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

double chi2_quantile(int dof, double p);

Roctx::Roctx() {
  if (!getenv("PLV_ROCTX")) return;
  for (const char *name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
    void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (push && pop) return;
    push = nullptr, pop = nullptr;
  }
}

static const int Q95_N = 1024;

static int h2d(plv_ctx *ctx, void *dst, const void *src, size_t bytes) {
  PLV_HIP_CHECK(plv::memcpy_async(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return PLV_OK;
}
static int d2h(plv_ctx *ctx, void *dst, const void *src, size_t bytes) {
  PLV_HIP_CHECK(plv::memcpy_async(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return PLV_OK;
}
static int sync(plv_ctx *ctx) {
  const unsigned long long stamp = ctx->gather_stamp;
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->cov_host_synced = stamp;
  ctx->prof.collect();
  return PLV_OK;
}
// upload a col-major matrix with arbitrary ld into a dense (ld = rows) device matrix
static int upload_mat(plv_ctx *ctx, double *dst, const double *src, int rows, int cols, int ld) {
  if (ld == rows) return h2d(ctx, dst, src, (size_t)rows * cols * 8);
  PLV_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)rows * 8, src, (size_t)ld * 8, (size_t)rows * 8, cols,
                                 hipMemcpyHostToDevice, ctx->stream));
  return PLV_OK;
}
static int download_mat(plv_ctx *ctx, double *dst, const double *src, int rows, int cols, int ld) {
  if (ld == rows) return d2h(ctx, dst, src, (size_t)rows * cols * 8);
  PLV_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)ld * 8, src, (size_t)rows * 8, (size_t)rows * 8, cols,
                                 hipMemcpyDeviceToHost, ctx->stream));
  return PLV_OK;
}

}  // namespace plv

using namespace plv;

#define REQUIRE_CTX(ctx)                     \
  do {                                       \
    if (!(ctx)) {                            \
      set_last_error("null ctx");            \
      return PLV_E_BADARG;                   \
    }                                        \
    (void)hipSetDevice((ctx)->device);       \
  } while (0)
#define TRY(expr)              \
  do {                         \
    int _rc = (expr);          \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

static std::mutex g_state_mtx;
static std::vector<std::pair<plv_ctx *, plv_ctx_update_state *>> g_states;
plv_ctx_update_state *plv_update_state(plv_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_state_mtx);
  for (auto &p : g_states)
    if (p.first == ctx) return p.second;
  return nullptr;
}
static plv_ctx_update_state *ustate(plv_ctx *ctx) { return plv_update_state(ctx); }

extern "C" {

int plv_abi_version(void) { return PLV_ABI_VERSION; }
// (measurement aid, not part of the drop-in surface: plv_ctx.hpp "Measurement knobs")  set < 0 only queries; returns the previous mask
// (measurement aid) wall time inside the parts of plv_camera_frame since the library was loaded: [0] the wait for flow + RANSAC,
// [1] plv_camera_update_points, [2] of it the wait for the device, [3] plv_camera_update_lines, [4] the join of the line worker
// [5..9] the line worker: post -> wake-up, wait for the edge maps, walk + fit, feed post -> start, the feed
void plv_phase_counters(unsigned long long *out10) {
  if (!out10) return;
  auto &c = plv::counters();
  out10[0] = c.flow_wait_ns, out10[1] = c.points_ns, out10[2] = c.points_wait_ns, out10[3] = c.lines_ns, out10[4] = c.line_join_ns;
  out10[5] = c.w_wake_ns, out10[6] = c.w_maps_ns, out10[7] = c.w_extract_ns, out10[8] = c.w_feed_start_ns, out10[9] = c.w_feed_ns;
}
// (measurement aid) device / pinned (re)allocations since the library was loaded: a frame that grows a buffer pays a hipMalloc
unsigned long long plv_alloc_count(void) { return plv::alloc_epoch().load(); }
// (measurement aid) line launches that plv_camera_try_update enqueued behind a point update still running (the chained line launch)
unsigned long long plv_chain_count(void) { return plv::counters().chained.load(); }
// (measurement aid) updates collected since the library was loaded, by the route they took (plv_update_compression_mode's last_route:
// 0 no compression, 1 Gram + Cholesky, 2 Householder, 3 Gram then Householder, 4 whitened, 5 whitened rejected, then Householder)
void plv_route_counts(unsigned long long *out8) {
  if (!out8) return;
  for (int i = 0; i < 8; ++i) out8[i] = plv::counters().route[i].load();
  out8[7] = plv::counters().speculated.load();  // (point updates enqueued behind the frame's flow and used as they ran)
}
void plv_memory_bytes(unsigned long long *out4) {
  if (!out4) return;
  plv::MemoryBook &b = plv::memory_book();
  out4[0] = (unsigned long long)std::max(0ll, b.dev.load()), out4[1] = (unsigned long long)std::max(0ll, b.pin.load());
  out4[2] = (unsigned long long)std::max(0ll, b.dev_peak.load()), out4[3] = (unsigned long long)std::max(0ll, b.pin_peak.load());
}
int plv_memory_policy(int growth_percent, int device_floor_kb, int pinned_floor_kb) {
  plv::MemoryBook &b = plv::memory_book();
  if ((growth_percent >= 0 && growth_percent < 100) || growth_percent > 1000) return PLV_E_BADARG;
  if (growth_percent >= 100) b.growth_percent.store(growth_percent);
  if (device_floor_kb >= 0) b.dev_floor_kb.store(device_floor_kb);
  if (pinned_floor_kb >= 0) b.pin_floor_kb.store(pinned_floor_kb);
  b.dev_peak.store(b.dev.load()), b.pin_peak.store(b.pin.load());
  return PLV_OK;
}
void plv_speculation_counts(unsigned long long *out4) {
  if (!out4) return;
  out4[0] = plv::counters().speculated.load();
  for (int i = 0; i < 3; ++i) out4[1 + i] = plv::counters().spec_over[i].load();
}
unsigned plv_debug_knobs(long long set) {
  const unsigned prev = plv::knobs().load();
  if (set >= 0) plv::knobs().store((unsigned)set);
  return prev;
}
const char *plv_last_error(void) { return g_err; }

int plv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

// NUMA node of a HIP device's PCI function (sysfs), -1 when unknown: the caller's threads belong on that node's cores — the ctx
// stream's doorbell, the pinned result blocks the host polls and the library's worker threads all live next to the device then
int plv_device_numa_node(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return -1;
  for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

void plv_config_default(plv_config *c, int width, int height) {
  memset(c, 0, sizeof(*c));
  c->width = width;
  c->height = height;
  c->num_features = 250;
  c->fast_threshold = 20;
  c->grid_x = 5;
  c->grid_y = 5;
  c->min_px_dist = 10;
  c->histogram_method = PLV_HIST_HISTOGRAM;
  c->win_size = 15;
  c->pyr_levels = 5;
  c->lk_max_iters = 30;
  c->lk_eps = 0.01f;
  c->ransac_thr_px = 2.0;
  c->ransac_conf = 0.999;
  c->ransac_max_iters = 1000;
  const double intr[8] = {458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05};
  for (int i = 0; i < 8; ++i) c->intrinsics[i] = intr[i];
  c->intrinsics[2] *= width / 752.0;
  c->intrinsics[3] *= height / 480.0;
  c->line_length_threshold = 20;
  c->line_distance_threshold = 1.414213562f;
  c->canny_th1 = 50;
  c->canny_th2 = 50;
  c->canny_aperture = 3;
  c->line_min_length_px = 40.f;
  c->line_assign_px = 5.f;
  c->line_similar_px = 6.f;
  c->max_state_dim = 160;
  c->max_meas_rows = 8192;
  c->max_features = 512;
  c->max_rows_per_feat = 48;
  c->sigma_pix = 1.5;
  c->chi2_mult = 1.0;
  c->device = 0;
}

int plv_set_camera_intrinsics(plv_ctx *ctx, const double *K8) {
  if (!ctx || !K8) return PLV_E_BADARG;
  for (int i = 0; i < 8; ++i)
    if (!std::isfinite(K8[i])) return PLV_E_BADARG;
  // cfg.intrinsics is read at every undistortion / RANSAC-threshold use, so this takes effect with the next frame
  for (int i = 0; i < 8; ++i) ctx->cfg.intrinsics[i] = K8[i];
  return PLV_OK;
}

int plv_ctx_create(const plv_config *cfg, plv_ctx **out) {
  if (!cfg || !out) {
    set_last_error("plv_ctx_create: null argument");
    return PLV_E_BADARG;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device) {
    set_last_error("no HIP device %d visible (this library has no CPU fallback)", cfg->device);
    return PLV_E_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_last_error("device %d is not gfx950 (the code objects are gfx950-only)", cfg->device);
    return PLV_E_NO_DEVICE;
  }
  PLV_HIP_CHECK(hipSetDevice(cfg->device));
  plv_ctx *ctx = new plv_ctx();
  ctx->cfg = *cfg;
  ctx->device = cfg->device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    set_last_error("hipStreamCreate failed");
    return PLV_E_DEVICE;
  }
  auto *us = new plv_ctx_update_state();
  // chi-square table (REF: UpdaterStatistics.cpp:31-37)
  std::vector<double> q(Q95_N, 0.0);
  for (int i = 1; i < Q95_N; ++i) q[i] = chi2_quantile(i, 0.95);
  int rc = us->q95.reserve(Q95_N * 8);
  if (rc == PLV_OK && hipMemcpy(us->q95.p, q.data(), Q95_N * 8, hipMemcpyHostToDevice) != hipSuccess) rc = PLV_E_DEVICE;
  if (rc != PLV_OK) {
    delete us;
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_state_mtx);
    g_states.push_back({ctx, us});
  }
  *out = ctx;
  return PLV_OK;
}

void plv_frontend_destroy(plv_ctx *ctx);  // frontend_api.hip
void plv_tracker_destroy(plv_ctx *ctx);   // tracker_api.hip
void plv_line_tracker_destroy(plv_ctx *ctx);  // line_api.hip

void plv_ctx_destroy(plv_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)plv::stream_sync(ctx->stream);
  plv_tracker_destroy(ctx);
  plv_line_tracker_destroy(ctx);
  plv_frontend_destroy(ctx);
  ctx->prof.destroy();
  plv::DevBuf *bufs[] = {&ctx->d_P, &ctx->d_P2, &ctx->d_H, &ctx->d_res, &ctx->d_cols, &ctx->d_Rdiag, &ctx->d_dx, &ctx->d_flag,
                         &ctx->d_Mt, &ctx->d_S, &ctx->d_W, &ctx->d_y, &ctx->d_fHf, &ctx->d_fHx, &ctx->d_fres,
                         &ctx->d_frows, &ctx->d_chi2, &ctx->d_acc, &ctx->d_stack, &ctx->d_stack_l, &ctx->d_stack2, &ctx->d_Pc, &ctx->d_Ps,
                         &ctx->d_inv, &ctx->d_T, &ctx->d_Lt, &ctx->d_W0, &ctx->d_dW, &ctx->d_Gs};
  for (auto *b : bufs) b->release();
  if (ctx->aux_stream) {
    (void)plv::stream_sync(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux_stream);
  }
  if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
  if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
  ctx->h_pin.release();
  ctx->h_pin_flow.release();
  ctx->h_pin_l.release();
  ctx->h_done.release();
  plv_ctx_update_state *us = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_state_mtx);
    for (size_t i = 0; i < g_states.size(); ++i)
      if (g_states[i].first == ctx) {
        us = g_states[i].second;
        g_states.erase(g_states.begin() + i);
        break;
      }
  }
  if (us) {
    plv::DevBuf *ub[] = {&us->q95, &us->result, &us->result_l, &us->covck, &us->bHf, &us->bHx, &us->bres, &us->brows, &us->bcols, &us->bcols_l, &us->bwork};
    for (auto *b : ub) b->release();
    us->jin.release();
    us->tri.release();
    us->jin_l.release();
    us->tri_l.release();
    us->h_jin_l.release();
    us->h_tri_l.release();
    us->chain_words.release();
    us->eval.release();
    if (us->done_ev) (void)hipEventDestroy(us->done_ev);
    if (us->gexec) (void)hipGraphExecDestroy(us->gexec);
    us->gexec = nullptr;
    us->h_jin.release();
    us->h_tri.release();
    delete us;
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int plv_front_quiesce(plv_ctx *ctx);        // frontend_api.hip: a detection started ahead of time on the side stream
extern "C" int plv_line_tracker_feed_wait(plv_ctx *ctx);
// Everything the calls so far have started is finished at return: the ctx stream, the side stream of the detection that a point
// update starts ahead of time, and the library's line worker (an asynchronous feed is joined; its status stays with
// plv_line_tracker_feed_wait).  bench.py ends every timed step here.
int plv_ctx_synchronize(plv_ctx *ctx) {
  REQUIRE_CTX(ctx);
  plv::NsScope ns(plv::counters().sync_ns);
  {
    plv::HostPhase ph("ctx_synchronize: detection side stream");
    TRY(plv_front_quiesce(ctx));
  }
  {
    plv::HostPhase ph("ctx_synchronize: line worker");
    (void)plv_line_tracker_feed_wait(ctx);
  }
  plv::HostPhase ph("ctx_synchronize: ctx stream");
  TRY(sync(ctx));
  ctx->prof.collect();
  return PLV_OK;
}

int plv_prof_enable(plv_ctx *ctx, int on) {
  REQUIRE_CTX(ctx);
  TRY(sync(ctx));
  ctx->prof.on = on != 0;
  return PLV_OK;
}
int plv_prof_reset(plv_ctx *ctx) {
  REQUIRE_CTX(ctx);
  TRY(sync(ctx));
  ctx->prof.reset();
  return PLV_OK;
}
int plv_prof_count(plv_ctx *ctx) { return ctx ? (int)ctx->prof.recs.size() : 0; }
int plv_prof_get(plv_ctx *ctx, int idx, char *name, int name_cap, int *launches, double *total_ms) {
  if (!ctx || idx < 0 || idx >= (int)ctx->prof.recs.size()) return PLV_E_BADARG;
  const auto &r = ctx->prof.recs[idx];
  if (name && name_cap > 0) {
    strncpy(name, r.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (launches) *launches = r.launches;
  if (total_ms) *total_ms = r.total_ms;
  return PLV_OK;
}

double plv_chi2_quantile95(int dof) { return chi2_quantile(dof, 0.95); }

// ------------------------------------------------------------------------------ covariance
int plv_cov_upload(plv_ctx *ctx, const double *P, int n, int ldp) {
  REQUIRE_CTX(ctx);
  if (!P || n < 1 || ldp < n) {
    set_last_error("plv_cov_upload: bad argument");
    return PLV_E_BADARG;
  }
  TRY(ctx->d_P.reserve((size_t)n * n * 8));
  TRY(upload_mat(ctx, ctx->d_P.as<double>(), P, n, n, ldp));
  ctx->cov_n = n;
  ++ctx->gather_stamp;
  return sync(ctx);
}
int plv_cov_download(plv_ctx *ctx, double *P, int n, int ldp) {
  REQUIRE_CTX(ctx);
  if (!P || n != ctx->cov_n || ldp < n) {
    set_last_error("plv_cov_download: bad argument (resident n = %d)", ctx->cov_n);
    return PLV_E_BADARG;
  }
  TRY(download_mat(ctx, P, ctx->d_P.as<double>(), n, n, ldp));
  return sync(ctx);
}
void plv_counters(unsigned long long *out) {
  plv::Counters &c = plv::counters();
  out[0] = c.launches, out[1] = c.syncs, out[2] = c.copies, out[3] = c.copy_bytes, out[4] = c.lk_iters, out[5] = c.lines_detected;
  out[6] = c.frame_ns, out[7] = c.sync_ns;
}
int plv_cov_checkpoint(plv_ctx *ctx) {
  REQUIRE_CTX(ctx);
  auto *us = ustate(ctx);
  if (ctx->cov_n < 1) return PLV_E_BADARG;
  size_t bytes = (size_t)ctx->cov_n * ctx->cov_n * 8;
  TRY(us->covck.reserve(bytes));
  PLV_HIP_CHECK(plv::memcpy_async(us->covck.p, ctx->d_P.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  us->covck_n = ctx->cov_n;
  return sync(ctx);
}
int plv_cov_rollback(plv_ctx *ctx) {
  REQUIRE_CTX(ctx);
  auto *us = ustate(ctx);
  if (us->covck_n < 1 || !us->covck.p) {
    set_last_error("plv_cov_rollback: no checkpoint");
    return PLV_E_BADARG;
  }
  // the checkpoint carries its own dimension: after plv_cov_clone / plv_cov_marginalize / plv_slam_initialize changed the resident
  // one, the rollback restores the dimension together with the data (the clone / marginalise calls may have swapped d_P for a
  // buffer of another size)
  size_t bytes = (size_t)us->covck_n * us->covck_n * 8;
  TRY(ctx->d_P.reserve(bytes));
  PLV_HIP_CHECK(plv::memcpy_async(ctx->d_P.p, us->covck.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  ctx->cov_n = us->covck_n;
  ++ctx->gather_stamp;
  return PLV_OK;  // ordered on the ctx stream; no host sync needed
}

// ------------------------------------------------------------------------------ EKF update
static size_t result_rows_off(int n, int F) { return ((size_t)n * 8 + 16 + (size_t)F + 7) & ~(size_t)7; }
static int result_buf(plv_ctx *ctx, plv_ctx_update_state *us, int n, int F, double **dx, int **flag, unsigned char **acc,
                      int **acc_rows = nullptr, int fdim = 3) {
  size_t bytes = result_rows_off(n, F) + (size_t)F * 4 + 16;
  plv::DevBuf &rb = us->result_of(fdim);
  const void *before = rb.p;
  TRY(rb.reserve(bytes));
  if (rb.p != before) PLV_HIP_CHECK(hipMemsetAsync(rb.p, 0, rb.cap, ctx->stream));
  char *b = rb.as<char>();
  *dx = (double *)b;
  *flag = (int *)(b + (size_t)n * 8);
  *acc = (unsigned char *)(b + (size_t)n * 8 + 16);
  if (acc_rows) *acc_rows = (int *)(b + result_rows_off(n, F));
  return PLV_OK;
}

// rows one EKF factorisation takes: the blocked kernel up to 128, the LDS-resident general one up to r (r | 1) 8 B <= 160 KB
static const int EKF_MAX_ROWS = 128;

int plv_ekf_update(plv_ctx *ctx, double *P, int n, int ldp, const double *H, int r, int k, int ldh,
                   const int *col_to_state, const double *res, const double *Rdiag, double *dx) {
  REQUIRE_CTX(ctx);
  if (!H || !col_to_state || !res || !dx || r < 1 || k < 1 || ldh < r || n < 1) {
    set_last_error("plv_ekf_update: bad argument");
    return PLV_E_BADARG;
  }
  for (int j = 0; j < k; ++j)
    if (col_to_state[j] < 0 || col_to_state[j] >= n) {
      set_last_error("plv_ekf_update: col_to_state[%d]=%d out of range", j, col_to_state[j]);
      return PLV_E_BADARG;
    }
  auto *us = ustate(ctx);
  ++ctx->gather_stamp;  // this call rewrites the covariance (and gathers for its own columns)
  if (P) {
    if (ldp < n) return PLV_E_BADARG;
    TRY(ctx->d_P.reserve((size_t)n * n * 8));
    TRY(upload_mat(ctx, ctx->d_P.as<double>(), P, n, n, ldp));
    ctx->cov_n = n;
  } else if (ctx->cov_n != n) {
    set_last_error("plv_ekf_update: no device-resident covariance of dimension %d", n);
    return PLV_E_BADARG;
  }
  if (r > EKF_MAX_ROWS) {
    // More rows than one factorisation holds.  The noise is diagonal, so the rows are independent measurements and the update
    // equals the same rows applied block after block (each block's residual taken at the state the earlier blocks produced).
    // A rejection anywhere rejects everything: the covariance is restored and dx left untouched, as EKFUpdate would.
    const size_t bytes = (size_t)n * n * 8;
    TRY(ctx->d_P2.reserve(bytes));
    PLV_HIP_CHECK(plv::memcpy_async(ctx->d_P2.p, ctx->d_P.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    std::vector<double> dx_tot(n, 0.0), dx_b(n), res_b;
    int rc = PLV_OK;
    for (int r0 = 0; r0 < r && rc == PLV_OK; r0 += EKF_MAX_ROWS) {
      const int rb_ = std::min(EKF_MAX_ROWS, r - r0);
      res_b.assign(res + r0, res + r0 + rb_);
      for (int j = 0; j < k; ++j) {
        const double d = dx_tot[col_to_state[j]];
        if (d != 0.0)
          for (int i = 0; i < rb_; ++i) res_b[i] -= H[(size_t)j * ldh + r0 + i] * d;
      }
      rc = plv_ekf_update(ctx, nullptr, n, n, H + r0, rb_, k, ldh, col_to_state, res_b.data(), Rdiag ? Rdiag + r0 : nullptr, dx_b.data());
      if (rc == PLV_OK)
        for (int i = 0; i < n; ++i) dx_tot[i] += dx_b[i];
    }
    if (rc != PLV_OK) {
      PLV_HIP_CHECK(plv::memcpy_async(ctx->d_P.p, ctx->d_P2.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      TRY(sync(ctx));
      return rc;
    }
    std::copy(dx_tot.begin(), dx_tot.end(), dx);
    if (P) {
      TRY(download_mat(ctx, P, ctx->d_P.as<double>(), n, n, ldp));
      TRY(sync(ctx));
    }
    return PLV_OK;
  }
  TRY(ctx->d_H.reserve((size_t)r * k * 8));
  TRY(ctx->d_res.reserve((size_t)r * 8));
  TRY(ctx->d_cols.reserve((size_t)k * 4));
  TRY(upload_mat(ctx, ctx->d_H.as<double>(), H, r, k, ldh));
  TRY(h2d(ctx, ctx->d_res.p, res, (size_t)r * 8));
  TRY(h2d(ctx, ctx->d_cols.p, col_to_state, (size_t)k * 4));
  const double *dR = nullptr;
  if (Rdiag) {
    TRY(ctx->d_Rdiag.reserve((size_t)r * 8));
    TRY(h2d(ctx, ctx->d_Rdiag.p, Rdiag, (size_t)r * 8));
    dR = ctx->d_Rdiag.as<double>();
  }
  double *d_dx;
  int *d_flag;
  unsigned char *d_acc;
  TRY(result_buf(ctx, us, n, 0, &d_dx, &d_flag, &d_acc));
  if (ekf_fast_fits(r))
    TRY(launch_ekf_fast(ctx, ctx->d_P.as<double>(), n, n, ctx->d_H.as<double>(), r, k, r, ctx->d_cols.as<int>(),
                        ctx->d_res.as<double>(), dR, d_dx, d_flag));
  else
    TRY(launch_ekf(ctx, ctx->d_P.as<double>(), n, n, ctx->d_H.as<double>(), r, k, r, ctx->d_cols.as<int>(),
                   ctx->d_res.as<double>(), dR, d_dx, d_flag));
  size_t rb = (size_t)n * 8 + 16;
  TRY(ctx->h_pin.reserve(rb));
  TRY(d2h(ctx, ctx->h_pin.p, us->result.p, rb));
  TRY(sync(ctx));
  int flag = *(int *)(ctx->h_pin.as<char>() + (size_t)n * 8);
  if (flag != 0) {
    set_last_error("EKFUpdate rejected: %s", (flag & 16) ? "speculative batch over the selection cap (run again by the caller)" : (flag & 2) ? "S not positive definite" : "negative covariance diagonal");
    return PLV_E_NOT_PSD;
  }
  memcpy(dx, ctx->h_pin.p, (size_t)n * 8);
  if (P) {
    TRY(download_mat(ctx, P, ctx->d_P.as<double>(), n, n, ldp));
    TRY(sync(ctx));
  }
  return PLV_OK;
}

// ------------------------------------------------------------------------------ compress
int plv_compress(plv_ctx *ctx, double *H, int m, int k, int ldh, double *res, int *m_out) {
  REQUIRE_CTX(ctx);
  if (!H || !res || !m_out || m < 1 || k < 1 || ldh < m) {
    set_last_error("plv_compress: bad argument");
    return PLV_E_BADARG;
  }
  if (m <= k) {  // REF: StateHelper.cpp:605 fat matrix -> nothing to do
    *m_out = m;
    return PLV_OK;
  }
  const int nc = k + 1;
  TRY(ctx->d_stack.reserve((size_t)m * nc * 8));
  size_t tmp_elems = (size_t)(m / (2 * nc) + 2) * nc * nc;
  TRY(ctx->d_stack2.reserve(tmp_elems * 8));
  double *A = ctx->d_stack.as<double>();
  TRY(upload_mat(ctx, A, H, m, k, ldh));
  TRY(h2d(ctx, A + (size_t)m * k, res, (size_t)m * 8));
  double *R;
  int ldr;
  TRY(launch_tsqr(ctx, A, m, m, nc, ctx->d_stack2.as<double>(), tmp_elems, &R, &ldr));
  // top k rows of [R | z]
  PLV_HIP_CHECK(hipMemcpy2DAsync(H, (size_t)ldh * 8, R, (size_t)ldr * 8, (size_t)k * 8, k, hipMemcpyDeviceToHost,
                                 ctx->stream));
  TRY(d2h(ctx, res, R + (size_t)k * ldr, (size_t)k * 8));
  TRY(sync(ctx));
  *m_out = k;
  return PLV_OK;
}

// ------------------------------------------------------------------------------ batches
static int check_batch(int F, int fdim, int k, int ld, const int *rows) {
  if (F < 1 || fdim < 1 || k < 1 || ld < 1 || !rows) {
    set_last_error("feature batch: bad argument");
    return PLV_E_BADARG;
  }
  for (int f = 0; f < F; ++f)
    if (rows[f] < 0 || rows[f] > ld) {
      set_last_error("feature batch: rows[%d]=%d exceeds ld=%d", f, rows[f], ld);
      return PLV_E_BADARG;
    }
  return PLV_OK;
}

int plv_nullspace_batch(plv_ctx *ctx, int F, int fdim, int k, int ld, const int *rows, double *Hf, double *Hx,
                        double *res) {
  REQUIRE_CTX(ctx);
  if (!Hf || !Hx || !res) return PLV_E_BADARG;
  TRY(check_batch(F, fdim, k, ld, rows));
  size_t nHf = (size_t)F * fdim * ld, nHx = (size_t)F * k * ld, nr = (size_t)F * ld;
  TRY(ctx->d_fHf.reserve(nHf * 8));
  TRY(ctx->d_fHx.reserve(nHx * 8));
  TRY(ctx->d_fres.reserve(nr * 8));
  TRY(ctx->d_frows.reserve((size_t)F * 4));
  TRY(h2d(ctx, ctx->d_fHf.p, Hf, nHf * 8));
  TRY(h2d(ctx, ctx->d_fHx.p, Hx, nHx * 8));
  TRY(h2d(ctx, ctx->d_fres.p, res, nr * 8));
  TRY(h2d(ctx, ctx->d_frows.p, rows, (size_t)F * 4));
  TRY(launch_nullspace(ctx, F, fdim, k, ld, ctx->d_frows.as<int>(), ctx->d_fHf.as<double>(), ctx->d_fHx.as<double>(),
                       ctx->d_fres.as<double>()));
  TRY(d2h(ctx, Hf, ctx->d_fHf.p, nHf * 8));
  TRY(d2h(ctx, Hx, ctx->d_fHx.p, nHx * 8));
  TRY(d2h(ctx, res, ctx->d_fres.p, nr * 8));
  return sync(ctx);
}

int plv_chi2_batch(plv_ctx *ctx, const double *P, int n, int ldp, int F, int k, int ld, const int *rows,
                   const double *Hx, const double *res, const int *col_to_state, double sigma2, double *chi2) {
  REQUIRE_CTX(ctx);
  if (!Hx || !res || !col_to_state || !chi2) return PLV_E_BADARG;
  TRY(check_batch(F, 1, k, ld, rows));
  auto *us = ustate(ctx);
  ++ctx->gather_stamp;  // (gathers for its own columns; may replace the covariance)
  if (P) {
    TRY(ctx->d_P.reserve((size_t)n * n * 8));
    TRY(upload_mat(ctx, ctx->d_P.as<double>(), P, n, n, ldp));
    ctx->cov_n = n;
  } else if (ctx->cov_n != n) {
    return PLV_E_BADARG;
  }
  size_t nHx = (size_t)F * k * ld, nr = (size_t)F * ld;
  TRY(ctx->d_fHx.reserve(nHx * 8));
  TRY(ctx->d_fres.reserve(nr * 8));
  TRY(ctx->d_frows.reserve((size_t)F * 4));
  TRY(ctx->d_cols.reserve((size_t)k * 4));
  TRY(ctx->d_chi2.reserve((size_t)F * 8));
  TRY(h2d(ctx, ctx->d_fHx.p, Hx, nHx * 8));
  TRY(h2d(ctx, ctx->d_fres.p, res, nr * 8));
  TRY(h2d(ctx, ctx->d_frows.p, rows, (size_t)F * 4));
  TRY(h2d(ctx, ctx->d_cols.p, col_to_state, (size_t)k * 4));
  int max_mp = 0;
  for (int f = 0; f < F; ++f) max_mp = rows[f] > max_mp ? rows[f] : max_mp;
  Chi2Args a{};
  a.P = ctx->d_P.as<double>();
  a.ldp = n;
  a.k = k;
  a.ld = ld;
  a.fdim_off = 0;
  a.rows = ctx->d_frows.as<int>();
  a.Hx = ctx->d_fHx.as<double>();
  a.res = ctx->d_fres.as<double>();
  a.cols = ctx->d_cols.as<int>();
  a.sigma2 = sigma2;
  a.chi2 = ctx->d_chi2.as<double>();
  a.stack = nullptr;
  a.q95 = us->q95.as<double>();
  a.q95_n = Q95_N;
  a.min_rows = 1;
  TRY(launch_gather_cov(ctx, ctx->d_P.as<double>(), n, n, ctx->d_cols.as<int>(), k));
  TRY(launch_chi2(ctx, F, a, max_mp));
  TRY(d2h(ctx, chi2, ctx->d_chi2.p, (size_t)F * 8));
  return sync(ctx);
}

int plv_feat_batch_upload(plv_ctx *ctx, int F, int fdim, int k, int ld, const int *rows, const double *Hf,
                          const double *Hx, const double *res, const int *col_to_state) {
  REQUIRE_CTX(ctx);
  if (!Hf || !Hx || !res || !col_to_state) return PLV_E_BADARG;
  TRY(check_batch(F, fdim, k, ld, rows));
  auto *us = ustate(ctx);
  size_t nHf = (size_t)F * fdim * ld, nHx = (size_t)F * k * ld, nr = (size_t)F * ld;
  // pristine batch in ONE allocation [Hf | Hx | res] so that a single D2D restores the working copy
  TRY(us->bHf.reserve((nHf + nHx + nr) * 8));
  TRY(us->brows.reserve((size_t)F * 4));
  TRY(us->bcols_of(fdim).reserve((size_t)k * 4));
  TRY(h2d(ctx, us->bHf.p, Hf, nHf * 8));
  TRY(h2d(ctx, us->bHf.as<double>() + nHf, Hx, nHx * 8));
  TRY(h2d(ctx, us->bHf.as<double>() + nHf + nHx, res, nr * 8));
  TRY(h2d(ctx, us->brows.p, rows, (size_t)F * 4));
  TRY(h2d(ctx, us->bcols_of(fdim).p, col_to_state, (size_t)k * 4));
  us->bF = F;
  us->bfdim = fdim;
  us->bk = k;
  us->bld = ld;
  us->brows_host.assign(rows, rows + F);
  us->bmaxrows = 0;
  for (int f = 0; f < F; ++f) us->bmaxrows = rows[f] > us->bmaxrows ? rows[f] : us->bmaxrows;
  us->b_on_device_rows = false;
  us->b_single_use = false;
  us->b_projected = false;
  us->b_gather_token = 0;
  return sync(ctx);
}

static bool whitened_route(const plv_ctx_update_state *us, int Mtot, int k);
// Called by the one-submission updates before they build their batch: fills plv_ctx::gate_stage so that the projected Jacobian
// launch ends with the gate of every entry (gate_core.hpp) — verdicts, counter, stack — and plv_msckf_update_resident_launch starts
// behind it.  The buffers are the ones that launch function reserves (same sizes: no reallocation in between).
int plv_update_gate_prepare(plv_ctx *ctx, int F, int fdim, int k, int ld, double sigma2, double chi2_mult, double res_norm_gate, int probe) {
  auto *us = ustate(ctx);
  ctx->gate_stage = plv::GateStage{};
  ctx->gate_stage_taken = false;
  const int n = ctx->cov_n, mp_max = ld - fdim;
  const int rows_most = ctx->gate_rows_hint > 0 ? std::min(ctx->gate_rows_hint, ld) : ld;
  ctx->gate_rows_hint = 0;
  if (plv::knob(plv::PLV_KNOB_GATE_SEPARATE) || us->graph_mode || n < 1 || F < 1 || mp_max < 1 || rows_most - fdim > GATE_MMAX || k > GATE_KMAX)
    return PLV_OK;
  const int nc = k + 1, Mtot = F * mp_max;
  TRY(ctx->d_chi2.reserve((size_t)F * 8));
  TRY(ctx->stack_of(fdim).reserve_units((size_t)F, (size_t)std::max(ctx->cfg.num_features, 64), (size_t)mp_max * nc * 8));
  double *d_dx;
  int *d_flag;
  unsigned char *d_acc;
  int *d_acc_rows;
  TRY(result_buf(ctx, us, n, F, &d_dx, &d_flag, &d_acc, &d_acc_rows, fdim));
  const size_t rb = result_rows_off(n, F) + (size_t)F * 4;
  plv::PinBuf &hpin = ctx->res_pin(fdim);
  TRY(hpin.reserve(rb));
  plv_ctx_update_state::AccWords &aw = us->acc_of(fdim);
  aw.word = aw.word == 1 ? 2 : 1;  // the two counters alternate: this update counts in one and zeroes the other for the next
  // the word must be zero when the launch starts: the gate of the update before in this block zeroed it — unless this is the first
  // use, the block moved, cov_n changed (the words sit behind dx) or that launch declined the gate: then a memset does
  if (!(aw.z_ptr == us->result_of(fdim).p && aw.z_n == n && aw.z_word == aw.word)) PLV_HIP_CHECK(hipMemsetAsync(d_flag + aw.word, 0, 4, ctx->stream));
  aw.z_ptr = nullptr;  // (known again once a launch has taken the gate: plv_msckf_update_resident_launch)
  plv::GateStage &g = ctx->gate_stage;
  g.on = 1;
  g.P = ctx->d_P.as<double>();
  g.ldp = n;
  g.sigma2 = sigma2, g.chi2_mult = chi2_mult, g.res_norm_gate = res_norm_gate;
  g.q95 = us->q95.as<double>();
  g.q95_n = Q95_N;
  g.min_rows = fdim == 3 ? 4 : 5;  // REF: UpdaterCamera.cpp:228 / :406
  g.chi2 = ctx->d_chi2.as<double>();
  g.dec = nullptr;
  if (ctx->decision_trace && fdim == 3) {
    TRY(ctx->d_gate_dec.reserve((size_t)F * 24));
    g.dec = ctx->d_gate_dec.as<double>();
    ctx->dec_gate = true;
  } else if (ctx->decision_trace) {
    TRY(ctx->d_gate_dec_l.reserve((size_t)F * 24));
    g.dec = ctx->d_gate_dec_l.as<double>();
    ctx->dec_F_l = F;
  }
  g.accepted = d_acc;
  g.acc_rows = d_acc_rows;
  g.n_acc = d_flag + aw.word;
  g.n_acc_next = d_flag + (3 - aw.word);
  g.stack = ctx->stack_of(fdim).as<double>();
  g.lds = Mtot;
  g.mp_max = mp_max;
  g.stack_accepted_only = (Mtot > k && k <= 192 && whitened_route(us, Mtot, k)) ? 1 : 0;
  if (probe) {  // the verdicts also go to pinned memory (the caller adds the second block: probe_src / probe_dst / strides)
    char *hb = hpin.as<char>();
    g.h_accepted = (unsigned char *)(hb + (size_t)n * 8 + 16);
    g.h_acc_rows = (int *)(hb + result_rows_off(n, F));
  }
  return PLV_OK;
}
static bool whitened_route(const plv_ctx_update_state *us, int Mtot, int k) {
  return Mtot > k && k <= 192 && us->compress_mode == 0 && !us->graph_mode;
}
// Side stream: behind everything the main stream held when prior_mark() was called (the previous update's commit, the upload that
// carries d_cols), factor the prior block and form W0^T W0; aux_join is recorded behind them.  Two steps so that a caller can mark,
// enqueue its own next launch on the main stream first (the host's enqueue time of the side work then overlaps that launch's run
// time instead of delaying it), and start the side work afterwards.
static int prior_mark(plv_ctx *ctx, bool always_fork = false) {
  if (!ctx->aux_stream) {
    PLV_HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    PLV_HIP_CHECK(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
    PLV_HIP_CHECK(hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
  }
  // the host has waited for the covariance's last writer (plv_ctx::cov_host_synced): nothing to order the side stream behind
  // always_fork: the side work reads something the main stream is still producing besides the covariance (the column map a Jacobian
  // launch publishes: that launch does not move gather_stamp) — ADVICE r3
  ctx->aux_fork_needed = always_fork || ctx->cov_host_synced != ctx->gather_stamp;
  if (plv::host_phases().on) plv::host_phases().add("prior_mark: fork event needed (1 = yes)", ctx->aux_fork_needed ? 1.0 : 0.0);
  plv::HostPhase ph("prior_mark: fork event recorded");
  if (ctx->aux_fork_needed) PLV_HIP_CHECK(hipEventRecord(ctx->aux_fork, ctx->stream));
  return PLV_OK;
}
static int prior_start(plv_ctx *ctx, const int *d_cols, int k) {
  const int n = ctx->cov_n;
  if (ctx->aux_fork_needed) PLV_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
  TRY(launch_prior_factor(ctx, ctx->aux_stream, ctx->d_P.as<double>(), n, n, d_cols, k));
  PLV_HIP_CHECK(hipEventRecord(ctx->aux_join, ctx->aux_stream));
  return PLV_OK;
}
// Called by the one-submission updates (jacobian_api.hip) with the update's shape: phase 0 before anything of the update is on the main
// stream (its upload included), phase 1 right after the Jacobian launch with the column map in memory the device can read NOW (the
// pinned staging block: the side stream does not wait for the upload).  When the update will take the whitened route its prior factor runs next to triangulation,
// Jacobians and gate.
int plv_prior_prefetch(plv_ctx *ctx, int phase, const int *d_cols, int k, int F, int mp_max) {
  auto *us = ustate(ctx);
  if (plv::knob(plv::PLV_KNOB_PRIOR_LATE) || !whitened_route(us, F * mp_max, k) || ctx->cov_n < 1) return PLV_OK;
  if (phase == 0) return prior_mark(ctx);
  TRY(prior_start(ctx, d_cols, k));
  ctx->prior_pending = true;
  ctx->prior_k = k;
  return PLV_OK;
}

int plv_msckf_update_resident_launch(plv_ctx *ctx, double sigma2, double chi2_mult, double res_norm_gate) {
  REQUIRE_CTX(ctx);
  auto *us = ustate(ctx);
  const int F_guard = us->bF;
  us->pending_F = 0;
  ctx->probe_done = false;  // (a launch that failed after setting it, or a wait that was skipped, must not short-cut this update's wait)
  if (F_guard < 1 || ctx->cov_n < 1) {
    set_last_error("plv_msckf_update_resident: no staged batch / covariance");
    return PLV_E_BADARG;
  }
  const int F = us->bF, fdim = us->bfdim, k = us->bk, ld = us->bld, n = ctx->cov_n;
  size_t nHf = (size_t)F * fdim * ld, nHx = (size_t)F * k * ld, nr = (size_t)F * ld;
  TRY(ctx->d_chi2.reserve((size_t)F * 8));
  double *wHf;
  if (us->b_single_use) {
    // the batch was just built on the device (plv_build_jacobians_resident): consume it in place
    wHf = us->bHf.as<double>();
    us->b_single_use = false;
    us->bF = 0;
  } else {
    // working copy (the nullspace projection is in place): one D2D
    TRY(us->bwork.reserve((nHf + nHx + nr) * 8));
    PLV_HIP_CHECK(plv::memcpy_async(us->bwork.p, us->bHf.p, (nHf + nHx + nr) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    wHf = us->bwork.as<double>();
  }
  double *wHx = wHf + nHf, *wres = wHx + nHx;

  const int mp_max = (us->b_on_device_rows ? ld : us->bmaxrows) - fdim;
  if (mp_max < 1) {
    set_last_error("plv_msckf_update: no feature has more than fdim rows");
    return PLV_E_BADARG;
  }
  const int nc = k + 1;
  const int Mtot = F * mp_max;
  TRY(ctx->stack_of(fdim).reserve_units((size_t)F, (size_t)std::max(ctx->cfg.num_features, 64), (size_t)mp_max * nc * 8));
  size_t tmp_elems = (size_t)std::max(Mtot / (2 * nc) + 2, 16) * nc * nc;  // TSQR tree levels
  {  // Gram path: per-chunk partial tiles (64-row chunks, upper 16x16 tiles) + the reduced matrix
    const size_t nt = (size_t)(nc + 15) / 16, ntri = nt * (nt + 1) / 2;
    tmp_elems = std::max(tmp_elems, (size_t)((Mtot + 63) / 64) * ntri * 256 + (size_t)nc * nc);
  }
  TRY(ctx->d_stack2.reserve(tmp_elems * 8));

  double *d_dx;
  int *d_flag;
  unsigned char *d_acc;
  int *d_acc_rows;
  TRY(result_buf(ctx, us, n, F, &d_dx, &d_flag, &d_acc, &d_acc_rows, fdim));
  plv::DevBuf &resbuf = us->result_of(fdim);

  // a batch from plv_build_jacobians_resident arrives projected, with the covariance gathers done on its launch: they stand as
  // long as nothing has touched the covariance or the gathered blocks since (plv_ctx::gather_stamp)
  const bool projected = us->b_projected;
  const bool gathers_valid = projected && us->b_gather_token != 0 && us->b_gather_token == ctx->gather_stamp;
  us->b_gather_token = 0;
  // (the prefetched prior factor stands exactly as long as the gathers that rode on the same launch do)
  const bool prior_was_pending = ctx->prior_pending;
  const bool prefetched = prior_was_pending && ctx->prior_k == k && gathers_valid;
  ctx->prior_pending = false;
  const size_t rb = result_rows_off(n, F) + (size_t)F * 4;
  plv::PinBuf &hpin = ctx->res_pin(fdim);
  TRY(hpin.reserve(rb));
  struct SkipGuard {  // the words are only meaningful for the kernels of this update
    plv_ctx *c;
    ~SkipGuard() { c->skip_word = nullptr; }
  } skip_guard{ctx};
  // Whitened route (mode 0, DESIGN.md "Whitened update").  Its prior factor only needs the covariance: the one-submission updates
  // start it on the side stream before their Jacobian launch (plv_prior_prefetch); otherwise it starts here.
  const bool whiten = whitened_route(us, Mtot, k);
  bool aux_open = prior_was_pending;  // (a prefetch that is not taken after all is still joined: the main stream rewrites the covariance)
  auto aux_join = [&]() -> int {  // the main stream goes on only when the side work is done (it reads the covariance)
    if (!aux_open) return PLV_OK;
    aux_open = false;
    PLV_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));
    return PLV_OK;
  };
  TRY(ctx->h_done.reserve(256));
  ++ctx->update_seq;
  ctx->update_word_armed = !us->graph_mode && plv::knob(plv::PLV_KNOB_DONE_WORDS);
  ctx->update_word_used = false;
  us->word_seq = 0;
  const bool probing = ctx->probe && !us->graph_mode;
  auto start_prior = [&]() -> int {
    TRY(aux_join());
    TRY(prior_mark(ctx, true));  // (us->bcols is written by the Jacobian launch queued on the main stream)
    TRY(prior_start(ctx, us->bcols_of(fdim).as<int>(), k));
    aux_open = true;
    return PLV_OK;
  };
  auto enqueue = [&]() -> int {
  // (a probed update starts the factor once it knows that it takes the whitened route: few accepted rows go another way)
  if (whiten && !prefetched && !probing) TRY(start_prior());
  if (!projected) {
    // (+ the covariance gathers the gate and the EKF step read: independent of the projection, same launch)
    TRY(launch_nullspace(ctx, F, fdim, k, ld, us->brows.as<int>(), wHf, wHx, wres, ctx->d_P.as<double>(), n, n, us->bcols_of(fdim).as<int>()));
  } else if (!gathers_valid) {
    TRY(launch_gather_cov(ctx, ctx->d_P.as<double>(), n, n, us->bcols_of(fdim).as<int>(), k));
  }
  Chi2Args a{};
  a.P = ctx->d_P.as<double>();
  a.ldp = n;
  a.k = k;
  a.ld = ld;
  a.fdim_off = fdim;
  a.rows = us->brows.as<int>();
  a.Hx = wHx;
  a.res = wres;
  a.cols = us->bcols_of(fdim).as<int>();
  a.sigma2 = sigma2;
  a.chi2 = ctx->d_chi2.as<double>();
  a.dec = nullptr;
  if (ctx->decision_trace && fdim == 3) {
    TRY(ctx->d_gate_dec.reserve((size_t)F * 24));
    a.dec = ctx->d_gate_dec.as<double>();
    ctx->dec_gate = true;
  } else if (ctx->decision_trace) {
    TRY(ctx->d_gate_dec_l.reserve((size_t)F * 24));
    a.dec = ctx->d_gate_dec_l.as<double>();
    ctx->dec_F_l = F;
  }
  a.stack = ctx->stack_of(fdim).as<double>();
  a.lds = Mtot;
  a.mp_max = mp_max;
  a.chi2_mult = chi2_mult;
  a.res_norm_gate = res_norm_gate;
  a.q95 = us->q95.as<double>();
  a.q95_n = Q95_N;
  a.min_rows = fdim == 3 ? 4 : 5;  // REF: UpdaterCamera.cpp:228 / :406
  a.accepted = d_acc;
  a.acc_rows = d_acc_rows;
  a.n_acc = d_flag + 1;           // second word of the status block
  // with more rows than columns the stack goes to gram_direct_kernel, which walks the accepted entries only (not in the modes that may
  // hand the whole stack to the Householder route, nor beyond its 192-column capacity)
  a.stack_accepted_only = (Mtot > k && k <= 192 && whiten) ? 1 : 0;
  // read by every kernel enqueued from here on (cleared after enqueue()); only the blocked-Cholesky route honours it in all of
  // its kernels, so the Householder / LDS-resident fallbacks (more than 192 columns) run unconditionally
  ctx->skip_word = (Mtot > k ? k <= 192 : ekf_fast_fits(Mtot)) ? d_flag + 1 : nullptr;
  const bool probe = ctx->probe && !us->graph_mode;
  if (probe) {
    char *hb = hpin.as<char>();
    a.h_accepted = (unsigned char *)(hb + (size_t)n * 8 + 16);
    a.h_acc_rows = (int *)(hb + result_rows_off(n, F));
    a.probe_src = (const unsigned char *)ctx->probe_src;
    a.probe_dst = (unsigned char *)ctx->probe_dst;
    a.probe_stride_a = ctx->probe_stride_a, a.probe_off_b = ctx->probe_off_b, a.probe_stride_b = ctx->probe_stride_b;
  }
  // the gate already ran as the tail of the Jacobian launch (plv_update_gate_prepare): same buffers, its own counter word
  const bool gated = ctx->gate_stage_taken && projected && ctx->gate_stage.stack == a.stack && ctx->gate_stage.lds == a.lds &&
                     ctx->gate_stage.accepted == a.accepted && ctx->gate_stage.stack_accepted_only == a.stack_accepted_only;
  if (ctx->gate_stage_taken && !gated) {  // (the launch that took the gate did not write the projected blocks a separate gate would read)
    set_last_error("plv_msckf_update_resident: the batch was gated inside its Jacobian launch for other buffers than this update's");
    return PLV_E_BADARG;
  }
  if (gated) {
    ctx->skip_word = ctx->skip_word ? ctx->gate_stage.n_acc : nullptr;
    plv_ctx_update_state::AccWords &aw = us->acc_of(fdim);
    us->acc_word_used = aw.word;
    aw.z_ptr = resbuf.p, aw.z_n = n, aw.z_word = 3 - aw.word;  // (workgroup 0 of the gated launch zeroed the other counter)
  } else {
    us->acc_word_used = 1;
    TRY(launch_chi2(ctx, F, a, mp_max));
  }
  us->last_route = 0;  // (before the probe's early return: a line update that ends at the gate must not report the point update's route)
  us->redo_w.armed = false;
  if (probe) {
    // the gate's verdicts are in pinned memory when its launch has finished: most line updates end here (three frames in four at
    // BASELINE configs[2] accept no line), without the six launches that would find nothing to do
    if (ctx->probe_hook) ctx->probe_hook(ctx->probe_hook_arg);
    TRY(sync(ctx));
    const unsigned char *hacc = (const unsigned char *)(hpin.as<char>() + (size_t)n * 8 + 16);
    int any = 0;
    for (int f = 0; f < F; ++f) any |= hacc[f];
    if (!any) {
      memset(hpin.p, 0, (size_t)n * 8 + 16);  // dx = 0, status = updated-with-nothing (as the skipped chain reports it)
      ctx->probe_done = true;
      return aux_join();
    }
    // (round 6) The host holds the gate's verdicts here.  With no more accepted rows than columns the reference does not compress at
    // all (measurement_compress_inplace returns at once when H has no more rows than columns, StateHelper.cpp:604-606) and EKFUpdate
    // factors S = H P H^T + R of the accepted rows' size — a line update accepts one or two lines, ~20 rows, where the whitened form
    // factors a k x k matrix (k ~ 100: 34 us of blocked Cholesky against 6) behind an information matrix it has to form first.
    // The accepted rows are gathered into a dense block and go through EKFUpdate as they are.
    const int *hrows = (const int *)(hpin.as<char>() + result_rows_off(n, F));
    int m_acc = 0;
    for (int f = 0; f < F; ++f) m_acc += std::max(hrows[f], 0);
    if (whiten && m_acc > 0 && m_acc <= k && ekf_fast_fits(m_acc) && !plv::knob(plv::PLV_KNOB_FORCE_FACTOR_FORM)) {
      TRY(ctx->d_stackc.reserve((size_t)m_acc * nc * 8));
      TRY(launch_stack_compact(ctx, ctx->stack_of(fdim).as<double>(), Mtot, nc, d_acc_rows, F, mp_max, ctx->d_stackc.as<double>(), m_acc, true));
      TRY(aux_join());  // (a prior factor started ahead of time is not used, but it reads the covariance this update rewrites)
      const double *Hc = ctx->d_stackc.as<double>();
      TRY(launch_ekf_fast(ctx, ctx->d_P.as<double>(), n, n, Hc, m_acc, k, m_acc, us->bcols_of(fdim).as<int>(), Hc + (size_t)k * m_acc, nullptr, d_dx, d_flag,
                          gathers_valid, resbuf.p, hpin.p, (rb + 3) & ~(size_t)3));
      us->last_route = 0;
      return PLV_OK;
    }
    if (whiten && !prefetched) TRY(start_prior());
  }

  const double *dH, *dr;
  int r, ldh;
  if (whiten) {
    // REF: measurement_compress_inplace + EKFUpdate as one whitened step: no triangular factor of the measurements is formed
    TRY(launch_gram_information(ctx, ctx->stack_of(fdim).as<double>(), Mtot, nc, d_acc_rows, F, mp_max));
    TRY(aux_join());
    TRY(launch_ekf_whitened(ctx, ctx->d_P.as<double>(), n, n, k, us->bcols_of(fdim).as<int>(), d_dx, d_flag, resbuf.p, hpin.p, (rb + 3) & ~(size_t)3));
    us->last_route = 4;
    us->redo_w = plv_ctx_update_state::RedoW{!us->graph_mode, Mtot, k, n, F, mp_max, fdim, tmp_elems, rb, d_dx, d_flag, d_acc_rows};
    return PLV_OK;
  }
  if (Mtot > k) {
    // REF: measurement_compress_inplace — [R z] by Householder reflections on the stacked rows (TSQR): mode 1, and whatever the
    // whitened update does not take (more than 192 columns, graph mode)
    double *R;
    int ldr;
    if (projected && d_acc_rows && !plv::knob(plv::PLV_KNOB_TSQR_TREE)) {
      // (round 6b) the stack is F slots of mp_max rows, the accepted rows a fraction of them and the rest zeros: gathered first (the
      // count stays on the device: hqr_kernel reads it), 2.3 ms of six tree levels over 2100 rows became ~0.8 at workload C
      TRY(ctx->d_stackc.reserve((size_t)Mtot * nc * 8));
      TRY(ctx->d_count_words.reserve(64));
      int *m_dev = ctx->d_count_words.as<int>();
      TRY(launch_stack_compact(ctx, ctx->stack_of(fdim).as<double>(), Mtot, nc, d_acc_rows, F, mp_max, ctx->d_stackc.as<double>(), Mtot, false, m_dev));
      TRY(launch_tsqr(ctx, ctx->d_stackc.as<double>(), Mtot, Mtot, nc, ctx->d_stack2.as<double>(), tmp_elems, &R, &ldr, m_dev));
    } else {
      TRY(launch_tsqr(ctx, ctx->stack_of(fdim).as<double>(), Mtot, Mtot, nc, ctx->d_stack2.as<double>(), tmp_elems, &R, &ldr));
    }
    dH = R;
    dr = R + (size_t)k * ldr;
    r = k;
    ldh = ldr;
    us->last_route = 2;
  } else {
    dH = ctx->stack_of(fdim).as<double>();
    dr = dH + (size_t)k * Mtot;
    r = Mtot;
    ldh = Mtot;
  }
  if (ekf_fast_fits(r)) {  // (its last kernel mirrors the result block into h_pin: no copy command after the chain)
    TRY(launch_ekf_fast(ctx, ctx->d_P.as<double>(), n, n, dH, r, k, ldh, us->bcols_of(fdim).as<int>(), dr, nullptr, d_dx, d_flag, true,
                        resbuf.p, hpin.p, (rb + 3) & ~(size_t)3));
    return PLV_OK;
  }
  TRY(launch_ekf(ctx, ctx->d_P.as<double>(), n, n, dH, r, k, ldh, us->bcols_of(fdim).as<int>(), dr, nullptr, d_dx, d_flag, true));
  TRY(d2h(ctx, hpin.p, resbuf.p, rb));
    return PLV_OK;
  };
  if (us->graph_mode && !ctx->prof.on) {
    // (the skip word is chosen inside enqueue() from Mtot, k and d_flag: all functions of the key's own fields and of `result`)
    plv_ctx_update_state::GraphKey key{ctx->d_P.p, wHf, us->brows.p, us->bcols_of(fdim).p, resbuf.p, hpin.p, ctx->mirror2_src, ctx->mirror2_dst,
                                       (const void *)(d_flag + 1), ctx->mirror2_bytes, F, fdim + 16 * (projected ? 1 : 0) + 32 * (gathers_valid ? 1 : 0), k, ld, n, mp_max,
                                       sigma2, chi2_mult, res_norm_gate, plv::alloc_epoch().load()};
    if (us->gexec && key == us->gkey) {
      PLV_HIP_CHECK(hipGraphLaunch(us->gexec, ctx->stream));
      ++us->graph_replays;
    } else if (us->gseen && key == us->gkey_seen) {
      if (us->gexec) {
        (void)hipGraphExecDestroy(us->gexec);
        us->gexec = nullptr;
      }
      hipGraph_t g = nullptr;
      PLV_HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
      const int crc = enqueue();
      const hipError_t ce = hipStreamEndCapture(ctx->stream, &g);
      if (crc != PLV_OK || ce != hipSuccess || !g || plv::alloc_epoch().load() != key.epoch) {
        if (g) (void)hipGraphDestroy(g);
        us->gseen = false;
        if (crc != PLV_OK) return crc;
        TRY(enqueue());  // capture refused or a buffer grew under it: run this one eagerly
      } else {
        const hipError_t ie = hipGraphInstantiate(&us->gexec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) {
          us->gexec = nullptr;
          us->gseen = false;
          TRY(enqueue());
        } else {
          us->gkey = key;
          ++us->graph_captures;
          PLV_HIP_CHECK(hipGraphLaunch(us->gexec, ctx->stream));
        }
      }
    } else {
      TRY(enqueue());
      us->gkey_seen = key;
      us->gkey_seen.epoch = plv::alloc_epoch().load();  // the eager run may have grown buffers
      us->gseen = true;
    }
  } else {
    const int erc = enqueue();
    const int jrc = aux_join();  // (also on an error path: what follows on the main stream may rewrite the covariance)
    if (erc) return erc;
    if (jrc) return jrc;
  }
  ctx->gate_stage.on = 0, ctx->gate_stage_taken = false;
  ++ctx->gather_stamp;  // the update rewrites the covariance
  us->pending_fdim = fdim;
  us->pending_F = F;  // stream-ordered: the result block lands in h_pin; plv_msckf_update_resident_wait reads it
  if (!us->done_ev) PLV_HIP_CHECK(hipEventCreateWithFlags(&us->done_ev, hipEventDisableTiming));
  PLV_HIP_CHECK(hipEventRecord(us->done_ev, ctx->stream));
  us->done_stamp = ctx->gather_stamp;
  us->word_seq = ctx->update_word_used ? ctx->update_seq : 0;
  ctx->update_word_armed = ctx->update_word_used = false;
  return PLV_OK;
}

int plv_update_compression_mode(plv_ctx *ctx, int mode, int *last_route, int *last_ambiguous) {
  REQUIRE_CTX(ctx);
  auto *us = ustate(ctx);
  if (mode >= 0) {
    if (mode > 1) return PLV_E_BADARG;  // (round 5: the Gram + Cholesky compression of rounds 2-3 — modes 2 and 3 — is gone)
    us->compress_mode = mode;
  }
  if (last_route) *last_route = us->last_route;
  if (last_ambiguous) *last_ambiguous = us->last_ambiguous;
  return us->compress_mode;
}

int plv_update_graph_mode(plv_ctx *ctx, int on, int *captures, int *replays) {
  REQUIRE_CTX(ctx);
  auto *us = ustate(ctx);
  if (on >= 0) {
    us->graph_mode = on != 0;
    if (!us->graph_mode && us->gexec) {
      (void)hipGraphExecDestroy(us->gexec);
      us->gexec = nullptr;
      us->gseen = false;
    }
  }
  if (captures) *captures = us->graph_captures;
  if (replays) *replays = us->graph_replays;
  return PLV_OK;
}

int plv_msckf_update_resident_wait(plv_ctx *ctx, uint8_t *accepted, int *n_accepted_rows, double *dx) {
  REQUIRE_CTX(ctx);
  auto *us = ustate(ctx);
  const int F = us->pending_F, n = ctx->cov_n, wfdim = us->pending_fdim;
  plv::PinBuf &hpin = ctx->res_pin(us->pending_fdim);
  if (F < 1 || !dx) {
    set_last_error("plv_msckf_update_resident_wait: nothing was launched");
    return PLV_E_BADARG;
  }
  us->pending_F = 0;
  if (ctx->probe_done)
    ctx->probe_done = false;  // (ended at the gate: already synchronised, the result block in h_pin is complete)
  else if (ctx->prof.on || !us->done_ev)
    TRY(sync(ctx));
  else
  {
    // (not the whole stream: the caller may have enqueued more behind the update)
    if (us->word_seq)
      PLV_HIP_CHECK(plv::wait_done_word(ctx->done_word(16), us->word_seq, us->done_ev));
    else
      PLV_HIP_CHECK(plv::event_sync(us->done_ev));
    if (us->done_stamp > ctx->cov_host_synced) ctx->cov_host_synced = us->done_stamp;
  }
  us->word_seq = 0;
  const char *hb = hpin.as<char>();
  const int *hrows = (const int *)(hb + result_rows_off(n, F));
  us->last_ambiguous = 0;
  if (us->redo_w.armed && us->last_route == 4 && *(const int *)(hb + (size_t)n * 8) != 0 && !(*(const int *)(hb + (size_t)n * 8) & 16) &&
      ((const int *)(hb + (size_t)n * 8))[us->acc_word_used] > 0) {  // (bit 16: not a rejection — a speculative batch the selection loop's cap would have cut, ekf_commit_kernel)
    // The whitened update came back rejected (update_state.hpp, RedoW): nothing was committed.  The stacked rows are run again the
    // reference's way — compression, then S = R P R^T + I — with the compression by Householder reflections on the rows themselves
    // (on the KAIST-layout drive with stamps of 1.5e9 s, where this happens to the first update after the initialisation, the Gram +
    // Cholesky compression ended 5.6 cm from this one; and the updates the factor form hands over on the bench drive — the car at rest,
    // a prior block of rank 50 of 110, lambda above 1e4 — all have 5 .. 70 pivots the Gram + Cholesky compression cannot tell from zero,
    // so trying that compression first would only add its 0.1 ms to the 2 ms of this one).  Its verdict is the update's.
    const plv_ctx_update_state::RedoW rd = us->redo_w;
    us->redo_w.armed = false;
    const int nc = rd.k + 1;
    const size_t mb = ((size_t)rd.n * 8 + 16 + 3) & ~(size_t)3;
    ctx->skip_word = nullptr;
    double *R;
    int ldr;
    // (the accepted rows gathered into a dense matrix first: the stack is F x mp_max slots, most of them empty, and the tree of
    // Householder reductions costs 0.4 ms per level of 224 rows)
    int m_acc = 0;
    for (int f = 0; f < rd.F; ++f) m_acc += std::max(hrows[f], 0);
    const int m_c = std::max(m_acc, 2 * nc);
    TRY(ctx->d_stackc.reserve((size_t)m_c * nc * 8));
    TRY(launch_stack_compact(ctx, ctx->stack_of(wfdim).as<double>(), rd.Mtot, nc, rd.d_acc_rows, rd.F, rd.mp_max, ctx->d_stackc.as<double>(), m_c));
    TRY(launch_tsqr(ctx, ctx->d_stackc.as<double>(), m_c, m_c, nc, ctx->d_stack2.as<double>(), rd.tmp_elems, &R, &ldr));
    TRY(launch_ekf_fast(ctx, ctx->d_P.as<double>(), rd.n, rd.n, R, rd.k, rd.k, ldr, us->bcols_of(wfdim).as<int>(), R + (size_t)rd.k * ldr, nullptr, rd.d_dx, rd.d_flag,
                        true, us->result_of(rd.fdim).p, hpin.p, mb));
    TRY(sync(ctx));
    us->last_route = 5;
    ++ctx->gather_stamp;
    ctx->cov_host_synced = ctx->gather_stamp;
  }
  us->redo_w.armed = false;
  ++plv::counters().route[us->last_route & 7];
  int flag = *(const int *)(hb + (size_t)n * 8);
  const unsigned char *hacc = (const unsigned char *)(hb + (size_t)n * 8 + 16);
  int nrows = 0, overflow = 0;
  for (int f = 0; f < F; ++f) {
    if (accepted) accepted[f] = hacc[f];
    if (hrows[f] < 0) ++overflow;  // (gate_core.hpp: an entry with more rows than the gate inside the Jacobian launch holds)
    else nrows += hrows[f];
  }
  if (n_accepted_rows) *n_accepted_rows = nrows;
  if (overflow) {
    set_last_error("update: %d batch entries had more projected rows than the gate inside the Jacobian launch holds (%d): they were left out", overflow, GATE_MMAX);
    return PLV_E_CAPACITY;
  }
  if (flag != 0) {
    set_last_error("EKFUpdate rejected: %s", (flag & 16) ? "speculative batch over the selection cap (run again by the caller)" : (flag & 2) ? "S not positive definite" : "negative covariance diagonal");
    return PLV_E_NOT_PSD;
  }
  memcpy(dx, hb, (size_t)n * 8);
  return PLV_OK;
}

int plv_msckf_update_resident(plv_ctx *ctx, double sigma2, double chi2_mult, double res_norm_gate, uint8_t *accepted,
                              int *n_accepted_rows, double *dx) {
  if (!dx) return PLV_E_BADARG;
  TRY(plv_msckf_update_resident_launch(ctx, sigma2, chi2_mult, res_norm_gate));
  return plv_msckf_update_resident_wait(ctx, accepted, n_accepted_rows, dx);
}

int plv_msckf_update(plv_ctx *ctx, double *P, int n, int ldp, int F, int fdim, int k, int ld, const int *rows,
                     const double *Hf, const double *Hx, const double *res, const int *col_to_state, double sigma2,
                     double chi2_mult, double res_norm_gate, uint8_t *accepted, int *n_accepted_rows, double *dx) {
  REQUIRE_CTX(ctx);
  if (P) {
    TRY(plv_cov_upload(ctx, P, n, ldp));
  } else if (ctx->cov_n != n) {
    set_last_error("plv_msckf_update: no device-resident covariance of dimension %d", n);
    return PLV_E_BADARG;
  }
  for (int j = 0; j < k; ++j)
    if (!col_to_state || col_to_state[j] < 0 || col_to_state[j] >= n) {
      set_last_error("plv_msckf_update: col_to_state out of range");
      return PLV_E_BADARG;
    }
  TRY(plv_feat_batch_upload(ctx, F, fdim, k, ld, rows, Hf, Hx, res, col_to_state));
  int rc = plv_msckf_update_resident(ctx, sigma2, chi2_mult, res_norm_gate, accepted, n_accepted_rows, dx);
  if (rc == PLV_OK && P) rc = plv_cov_download(ctx, P, n, ldp);
  return rc;
}

}  // extern "C"
