// plv_ctx.hpp — library context: HIP stream, device workspaces, profiling events.
// Product code (never includes anything from oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <sys/resource.h>

#include "gate_stage.hpp"
#include <cstdint>
#include <cstdio>
#include <execinfo.h>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/plviwo.h"

namespace plv {

void set_last_error(const char *fmt, ...);

#define PLV_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      plv::set_last_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return PLV_E_DEVICE;                                                               \
    }                                                                                    \
  } while (0)

// Process-wide activity counters (plv_counters): what a frame costs in submissions, read by bench.py around the timed steps.
struct Counters {
  std::atomic<unsigned long long> launches{0}, syncs{0}, copies{0}, copy_bytes{0}, lk_iters{0}, lines_detected{0};
  std::atomic<unsigned long long> frame_ns{0}, sync_ns{0};  // wall time inside plv_camera_frame / plv_ctx_synchronize (steady_clock)
  // ... and inside its parts (plv_phase_counters): waiting for flow + RANSAC, the point update, the point update's device wait, the
  // line update, the join of the line worker
  std::atomic<unsigned long long> flow_wait_ns{0}, points_ns{0}, points_wait_ns{0}, lines_ns{0}, line_join_ns{0};
  // the line worker's side: post -> wake-up, the wait for the edge maps, chain walk + segment growth, feed post -> start, the feed
  std::atomic<unsigned long long> w_wake_ns{0}, w_maps_ns{0}, w_extract_ns{0}, w_feed_start_ns{0}, w_feed_ns{0};
  std::atomic<unsigned long long> chained{0};  // line launches enqueued behind a running point update (plv_chain_count)
  std::atomic<unsigned long long> spec_over[3] = {};  // plv_speculation_counts [1] .. [3]
  std::atomic<unsigned long long> speculated{0};  // point updates enqueued behind the frame's flow and used (plv_route_counts[7])
  std::atomic<unsigned long long> route[8] = {};  // collected updates by plv_ctx_update_state::last_route (plv_route_counts)
};
inline Counters &counters() {
  static Counters c;
  return c;
}
// workgroups of a speculative batch's Jacobian launch (pool entries it can work on): twice the selection loop's cap — at BASELINE
// configs[2] one frame in ten pools 71 .. 100 tracks of which 40 .. 65 pass their tests (the cap, 70, is not reached)
inline int spec_grid(int max_sel) { return 2 * (max_sel > 0 ? max_sel : 1); }
inline hipError_t stream_sync(hipStream_t s) {
  ++counters().syncs;
  return hipStreamSynchronize(s);
}
inline hipError_t event_sync(hipEvent_t e) {
  ++counters().syncs;
  return hipEventSynchronize(e);
}
inline hipError_t memcpy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  ++counters().copies;
  counters().copy_bytes += bytes;
  return hipMemcpyAsync(dst, src, bytes, kind, s);
}

// Measurement knobs (plv_debug_knobs): alternative placements kept in the library so that tools can switch them frame by frame inside
// one process — run-to-run drift on a box (+-25 us per frame) is larger than what most single changes move.
//   1  the prefetched edge kernel on its own stream behind the pyramid instead of on the ctx stream in front of the flow (measured
//      with tools: 6-15 us per frame SLOWER, four alternating runs of 600 frames; the default stays on the ctx stream)
//   2  the whitened update's prior factor started behind the Jacobian launch instead of before the update's upload
//   4  the edge kernel behind flow + RANSAC (PLV_KNOB_EDGES_LATE)          8  the next frame's detection on the ctx stream (PLV_KNOB_AHEAD_CTX)
//  16  the line pool formed after the point update (PLV_LINE_POOL_LATE)    32 / 64  point / line triangulation as its own launch
// 128  the Jacobian launches read their inputs from the pinned staging block instead of an uploaded copy
// 1024 the gate as chi2_t_kernel + chi2_gate_kernel behind the Jacobian launch instead of as that launch's tail (gate_core.hpp)
// 256  the flow's and the updates' waits on completion words their last kernels write into pinned memory (plv_ctx::h_done) instead of
//      on HIP events: a bare word is seen 4.8 us earlier (tools/ubench/waitlat.hip), in the frame it gains nothing (0 .. 9 us SLOWER over
//      four alternating runs: the commit then runs as one workgroup so that the word also covers the covariance)
//  512 the prefetched edge kernel behind the pyramid (rounds 2-3) instead of between the histogram and the pyramid (round 4: it equalises
//      the raw image itself, canny_kernel; the maps reach the line worker two launches earlier)
// 2048 no chained line launch: the line half is staged and enqueued after the host has collected and applied the point update (round 3)
// 8192 every whitened update takes its factor form (dense_kernels.hip "whitened update"; tests: the form the prior factor would pick
//      only late in a drive runs on every batch)
// 4096 the point update's wait keeps polling its hook until the hook is done even when the device has finished (tests: every frame's
//      line launch is chained, whatever the timing of the line worker)
enum : unsigned { PLV_KNOB_CHAIN_ALWAYS = 4096u, PLV_KNOB_NO_CHAIN = 2048u, PLV_KNOB_EDGES_SIDE = 1u, PLV_KNOB_PRIOR_LATE = 2u, PLV_KNOB_EDGES_LATE = 4u, PLV_KNOB_AHEAD_CTX = 8u, PLV_KNOB_POOL_LATE = 16u,
                  PLV_KNOB_POINT_TRI_SEPARATE = 32u, PLV_KNOB_LINE_TRI_SEPARATE = 64u, PLV_KNOB_INPUTS_PINNED = 128u, PLV_KNOB_DONE_WORDS = 256u, PLV_KNOB_GATE_SEPARATE = 1024u, PLV_KNOB_EDGES_AFTER_PYRAMID = 512u, PLV_KNOB_FORCE_FACTOR_FORM = 8192u,
                  // reporting aids (round 5: every measurement switch that used to be an environment variable of its own is a bit here)
                  PLV_KNOB_HOST_TIMING = 1u << 14,    // phase table of the host side on stderr when the library unloads (HostPhases)
                  PLV_KNOB_LINE_TIMING = 1u << 15,    // the line worker's stages, per frame, on stderr
                  PLV_KNOB_UPDATE_TIMING = 1u << 16,  // the line update's stages, per call, on stderr
                  PLV_KNOB_KERNEL_STAMPS = 1u << 17,  // s_memtime stamps of the fused Jacobian launches' phases
                  PLV_KNOB_CHAIN_EVENTS = 1u << 18,   // HIP events around the chained launches (with HOST_TIMING)
                  PLV_KNOB_ALLOC_DEBUG = 1u << 19,    // every (re)allocation of a library buffer with a backtrace
                  PLV_KNOB_HOST_FAULTS = 1u << 20,    // HostPhase counts minor page faults instead of time
                  PLV_KNOB_LK_LEGACY_LOOP = 1u << 21, // lk_kernel<0>: the iteration of rounds 2-4 (tools/lk_exp.py; same bits, slower)
                  PLV_KNOB_LINE_LABELS_OFF = 1u << 22,
                  PLV_KNOB_LK_AHEAD = 1u << 23,
                  PLV_KNOB_TSQR_TREE = 1u << 29,          // the Householder compression as the tree of unblocked workgroup factorisations (rounds 1-6a) instead of hqr_kernel
                  PLV_KNOB_HELPER_NAPS = 1u << 28,        // tests: the line detector's helper threads fall asleep (up to 200 us) at random when they pick a job up or start a part
                  PLV_KNOB_WAIT_ALL_HELPERS = 1u << 27,   // the line worker's jobs wait for every helper thread's report (rounds 5-6a) instead of doing a late helper's share themselves
                  PLV_KNOB_PART_LISTS_OFF = 1u << 26,     // the host stage builds a part's map from the labels of the whole image (round 5) instead of the part's own pixel list
                  PLV_KNOB_ASSIGN_ONE_THREAD = 1u << 25,  // the line feed's point-line assignment on the worker alone (rounds 1-5)
                  PLV_KNOB_NO_SPECULATION = 1u << 24 };    // plv_camera_frame submits the point update after the flow's result has reached the host (rounds 1-5), not behind the flow          // lk_ahead_kernel (round 6 experiment: all levels' templates first, search tiles a level ahead; same bits, no faster)  // the line detector's host stage walks the edge map as one sequence (rounds 2-4) instead of by labelled components
// The mask starts from PLV_DEBUG_KNOBS in the environment (the library's only measurement variable; plv_debug_knobs changes it at run time)
inline std::atomic<unsigned> &knobs() {
  static std::atomic<unsigned> k{getenv("PLV_DEBUG_KNOBS") ? (unsigned)strtoul(getenv("PLV_DEBUG_KNOBS"), nullptr, 0) : 0u};
  return k;
}
inline bool knob(unsigned bit) { return (knobs().load(std::memory_order_relaxed) & bit) != 0; }
inline bool alloc_debug() { return knob(PLV_KNOB_ALLOC_DEBUG); }

// Host-side phase timing (PLV_KNOB_HOST_TIMING): accumulated wall time per label, printed to stderr when the library unloads.
struct HostPhases {
  struct Rec {
    const char *label;
    double us = 0;
    long n = 0;
    float last[64] = {};  // the last 64 samples: their median is what a steady-state call costs (the mean carries the first call's
                          // stream / buffer creation: milliseconds)
    void put(double v) {
      last[n & 63] = (float)v;
      us += v;
      ++n;
    }
    double median() const {
      float tmp[64];
      const int m = (int)(n < 64 ? n : 64);
      for (int i = 0; i < m; ++i) tmp[i] = last[i];
      std::sort(tmp, tmp + m);
      return m ? tmp[m / 2] : 0.0;
    }
  };
  // latched when the table is first touched (the library's first call): PLV_KNOB_HOST_TIMING — and the LINE / UPDATE timing bits,
  // latched the same way by their readers — are honoured from PLV_DEBUG_KNOBS in the environment only; setting them later through
  // plv_debug_knobs does nothing (every scope timer would otherwise pay an atomic load per use)
  bool on = knob(PLV_KNOB_HOST_TIMING);
  std::mutex mtx;
  std::vector<Rec> recs;
  void add(const char *label, double us) {
    std::lock_guard<std::mutex> lk(mtx);
    // by address first: comparing the text of every earlier label walks the library's string pages, and those page in on first touch
    // (measured: 16 minor faults, 30 us, charged to whatever phase came next)
    for (auto &r : recs)
      if (r.label == label) {
        r.put(us);
        return;
      }
    for (auto &r : recs)
      if (!strcmp(r.label, label)) {
        r.put(us);
        return;
      }
    recs.push_back(Rec{label});
    recs.back().put(us);
  }
  ~HostPhases() {
    if (!on) return;
    for (auto &r : recs)
      fprintf(stderr, "[plv host] %-44s %8ld calls  %9.1f us/call  (median of the last %d: %.1f)\n", r.label, r.n, r.us / (double)r.n, (int)(r.n < 64 ? r.n : 64), r.median());
  }
};
inline HostPhases &host_phases() {
  static HostPhases h;
  return h;
}
// PLV_KNOB_HOST_TIMING: where inside the frame an event falls (microseconds since plv_camera_frame was entered), per label
inline std::atomic<long long> &frame_t0_ns() {
  static std::atomic<long long> t{0};
  return t;
}
inline void frame_mark(const char *label) {
  if (!host_phases().on) return;
  const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const long long t0 = frame_t0_ns().load(std::memory_order_relaxed);
  if (t0) host_phases().add(label, (double)(now - t0) * 1e-3);
}
struct NsScope {  // adds the scope's wall time to one of the counters above
  std::atomic<unsigned long long> &acc;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit NsScope(std::atomic<unsigned long long> &a) : acc(a) {}
  ~NsScope() { acc += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
inline long thread_minor_faults() {
  struct rusage ru;
  getrusage(RUSAGE_THREAD, &ru);
  return ru.ru_minflt;
}
struct HostPhase {  // scope timer (PLV_KNOB_HOST_FAULTS: the scope's minor page faults instead of its time)
  const char *label;
  std::chrono::steady_clock::time_point t0;
  long f0 = 0;
  bool on;
  static bool faults() {
    return knob(PLV_KNOB_HOST_FAULTS);
  }
  explicit HostPhase(const char *l) : label(l), on(host_phases().on) {
    if (on && faults()) f0 = thread_minor_faults();
    if (on) t0 = std::chrono::steady_clock::now();
  }
  void stop() {
    if (!on) return;
    if (faults())
      host_phases().add(label, (double)(thread_minor_faults() - f0));
    else
      host_phases().add(label, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    on = false;
  }
  ~HostPhase() { stop(); }
};

// roctx ranges with the reference's TimeChecker labels (REF: PL-VIWO/src/utils/TimeChecker.h:65-135, the dingdong() pairs of
// UpdaterCamera.cpp:79-190) so that a rocprofv3 --marker-trace timeline reads like the reference's own timing printout.  The roctx
// library is looked up at run time (PLV_ROCTX=1): no link dependency, no cost when it is off.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx();
};
inline Roctx &roctx() {
  static Roctx r;
  return r;
}
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char *label) : on(roctx().push != nullptr) {
    if (on) roctx().push(label);
  }
  void stop() {
    if (on) roctx().pop();
    on = false;
  }
  ~RoctxRange() { stop(); }
};

// A grow-only device buffer.
// Bumped by every device (re)allocation: a captured graph holds raw device pointers, so any growth anywhere retires it.
inline std::atomic<unsigned long long> &alloc_epoch() {  // contexts may live on different threads
  static std::atomic<unsigned long long> e{0};
  return e;
}

// What the library holds in HBM and in pinned host memory (plv_memory_bytes), and how generously a buffer is sized when it has to
// grow (plv_memory_policy): the defaults suit one context per GPU with 288 GB behind it, a process with many contexts or a
// smaller device lowers them.
struct MemoryBook {
  std::atomic<long long> dev{0}, pin{0}, dev_peak{0}, pin_peak{0};
  std::atomic<int> growth_percent{200}, dev_floor_kb{1024}, pin_floor_kb{256};
  void add(std::atomic<long long> &now, std::atomic<long long> &peak, long long bytes) {
    const long long v = now.fetch_add(bytes) + bytes;
    long long p = peak.load();
    while (v > p && !peak.compare_exchange_weak(p, v)) {
    }
  }
  size_t sized(size_t bytes, int floor_kb) const {
    const size_t grown = (size_t)((double)bytes * std::max(100, growth_percent.load()) / 100.0) + 256;
    return std::max(grown, (size_t)std::max(0, floor_kb) << 10);
  }
};
inline MemoryBook &memory_book() {
  static MemoryBook b;
  return b;
}

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return PLV_OK;
    ++alloc_epoch();
    if (alloc_debug()) {
      fprintf(stderr, "[plv alloc] device buffer %p: %zu -> %zu bytes asked\n", (void *)this, cap, bytes);
      void *bt[8];
      backtrace_symbols_fd(bt, backtrace(bt, 8), 2);
    }
    release();
    // By default twice what is asked for and never less than 1 MB (the device has 288 GB): batch sizes wander from frame to frame
    // (pool sizes, accepted rows, chain counts) and a buffer that regrows inside a frame costs that frame 0.3 - 0.4 ms (hipFree waits
    // for the device, hipMalloc maps pages) — round 4's driver-timed run had three such frames in its twenty.  plv_memory_policy.
    const size_t want = memory_book().sized(bytes, memory_book().dev_floor_kb.load());
    if (hipMalloc(&p, want) != hipSuccess) {
      p = nullptr;
      set_last_error("hipMalloc(%zu) failed", want);
      return PLV_E_NOMEM;
    }
    cap = want;
    memory_book().add(memory_book().dev, memory_book().dev_peak, (long long)want);
    return PLV_OK;
  }
  void release() {
    if (p) {
      (void)hipFree(p);
      memory_book().dev.fetch_sub((long long)cap);
    }
    p = nullptr;
    cap = 0;
  }
  // reserve for a batch of `units` entries of `bytes_per_unit` with room for `units_floor` of them: the buffers of the per-feature
  // batches are sized by the pool of the frame, which peaks when many tracks end at once (up to every tracked feature)
  int reserve_units(size_t units, size_t units_floor, size_t bytes_per_unit) { return reserve(std::max(units, units_floor) * bytes_per_unit); }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// Pinned host staging buffer (grow-only).
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return PLV_OK;
    ++alloc_epoch();  // kernels write into pinned blocks too (result mirrors): a captured graph holds their addresses
    if (alloc_debug()) fprintf(stderr, "[plv alloc] pinned buffer %p: %zu -> %zu bytes asked\n", (void *)this, cap, bytes);
    release();
    const size_t want = memory_book().sized(bytes, memory_book().pin_floor_kb.load());  // (as DevBuf::reserve: generous once instead of regrowing inside a frame)
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      set_last_error("hipHostMalloc(%zu) failed", want);
      return PLV_E_NOMEM;
    }
    cap = want;
    memory_book().add(memory_book().pin, memory_book().pin_peak, (long long)want);
    return PLV_OK;
  }
  void release() {
    if (p) {
      (void)hipHostFree(p);
      memory_book().pin.fetch_sub((long long)cap);
    }
    p = nullptr;
    cap = 0;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize only has to grow: remember the largest size set per kernel and skip the runtime call
// (a lock + a driver query on every launch otherwise) when the request fits.
inline hipError_t ensure_dyn_smem(const void *fn, int bytes) {
  static std::mutex mtx;
  static std::unordered_map<unsigned long long, int> seen;  // (device, kernel): the attribute is per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long key = (unsigned long long)(uintptr_t)fn * 64ull + (unsigned)(dev & 63);
  std::lock_guard<std::mutex> lk(mtx);
  auto it = seen.find(key);
  if (it != seen.end() && it->second >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) seen[key] = bytes;
  return e;
}

// A kernel's static LDS (its __shared__ arrays), which the limit of 160 KB holds next to the dynamic block; asked once per kernel.
inline size_t static_smem_bytes(const void *fn) {
  static std::mutex mtx;
  static std::unordered_map<uintptr_t, size_t> seen;
  std::lock_guard<std::mutex> lk(mtx);
  auto it = seen.find((uintptr_t)fn);
  if (it != seen.end()) return it->second;
  hipFuncAttributes a{};
  const size_t v = hipFuncGetAttributes(&a, fn) == hipSuccess ? a.sharedSizeBytes : 4096;
  seen[(uintptr_t)fn] = v;
  return v;
}

// HIP-event timing per kernel class, on the ctx stream.
struct Profiler {
  struct Rec {
    std::string name;
    int launches = 0;
    double total_ms = 0.0;
  };
  struct Pending {
    int rec;
    hipEvent_t a, b;
  };
  bool on = false;
  std::vector<Rec> recs;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;

  int find(const char *name) {
    for (size_t i = 0; i < recs.size(); ++i)
      if (recs[i].name == name) return (int)i;
    recs.push_back(Rec{name, 0, 0.0});
    return (int)recs.size() - 1;
  }
  hipEvent_t get_event() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
  void begin(const char *name, hipStream_t s) {
    if (!on) return;
    Pending p;
    p.rec = find(name);
    p.a = get_event();
    p.b = get_event();
    (void)hipEventRecord(p.a, s);
    pending.push_back(p);
  }
  void end(hipStream_t s) {
    if (!on) return;
    (void)hipEventRecord(pending.back().b, s);
  }
  // call after a stream sync.  Launches on another stream (the detection started ahead of time) may still be running: their
  // records stay pending until a later collect finds them finished.
  void collect() {
    size_t keep = 0;
    for (auto &p : pending) {
      if (hipEventQuery(p.b) == hipErrorNotReady) {
        pending[keep++] = p;
        continue;
      }
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        recs[p.rec].launches++;
        recs[p.rec].total_ms += ms;
      }
      pool.push_back(p.a);
      pool.push_back(p.b);
    }
    pending.resize(keep);
  }
  void reset() {
    for (auto &r : recs) r.launches = 0, r.total_ms = 0.0;
  }
  void destroy() {
    (void)hipDeviceSynchronize();
    collect();
    for (auto &p : pending) pool.push_back(p.a), pool.push_back(p.b);
    pending.clear();
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
  }
};

// Waits until *word == want (a kernel stores it behind its results).  `ev` (recorded behind that kernel) is looked at now and then: a
// chain that ended another way, or failed, must not hang the host.
inline hipError_t wait_done_word(volatile unsigned *word, unsigned want, hipEvent_t ev) {
  ++counters().syncs;
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n((const unsigned *)word, __ATOMIC_ACQUIRE) == want) return hipSuccess;
    if ((spins & 8191u) == 0 && ev && hipEventQuery(ev) != hipErrorNotReady) {
      const hipError_t e = hipEventSynchronize(ev);
      std::atomic_thread_fence(std::memory_order_acquire);
      return e;
    }
    __builtin_ia32_pause();
  }
}

struct ProfScope {
  Profiler &p;
  hipStream_t s;
  ProfScope(Profiler &p_, const char *name, hipStream_t s_) : p(p_), s(s_) {
    ++counters().launches;
    p.begin(name, s);
  }
  ~ProfScope() { p.end(s); }
};

}  // namespace plv

struct plv_ctx {
  plv_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  plv::Profiler prof;

  // ---- update side
  int cov_n = 0;          // dimension of the device-resident covariance (0 = none)
  // Bumped whenever the resident covariance changes OR the gathered blocks (d_Pc / d_Ps / d_inv) are rewritten: a batch whose
  // covariance gathers rode on an earlier launch (plv_build_jacobians_resident) may only reuse them while the stamp stands.
  unsigned long long gather_stamp = 1;
  // value of gather_stamp up to which the HOST knows the main stream's work to be finished (set where the host waits on the stream
  // or on an event recorded with the stamp of that moment).  Equal to gather_stamp: the covariance's last writer has finished, work
  // on another stream may read it without an event (plv_api.hip, prior_mark: the event itself is cheap, the first event call on a
  // stream the host has just waited on is not — 35 us measured).
  unsigned long long cov_host_synced = 0;
  plv::DevBuf d_P;        // n x n col-major, ld = n
  plv::DevBuf d_P2;       // second covariance buffer: state augmentation / marginalisation write here, then swap
  plv::DevBuf d_H, d_res, d_cols, d_Rdiag, d_dx, d_flag;
  plv::DevBuf d_Mt, d_S, d_W, d_y;          // EKF workspaces
  plv::DevBuf d_Pc, d_Ps, d_inv, d_T;       // dense covariance gathers, H'Ps
  plv::DevBuf d_fHf, d_fHx, d_fres, d_frows, d_chi2, d_acc;  // per-feature batches
  // Set by the tracker feed around the image feed while the line prefetch is on: called between the histogram launch and the first
  // pyramid launch with the raw image and its histogram (frontend_kernels.hip launch_equalize_pyramid) — the line detector's edge
  // kernel goes there (line_api.hip plv_line_edges_early); edges_hook_fired: it did, the feed must not launch the detection again
  void (*edges_hook)(plv_ctx *, const uint8_t *d_raw, int W, int H, const unsigned *d_hist) = nullptr;
  bool edges_hook_fired = false;
  // set around the call of edges_hook: the rest of the image feed's launches (the pyramid).  The hook runs it right behind its edge
  // kernel — in front of the label kernels, the events and the hand-over to the line worker, ~35 us of host work that used to sit
  // between the edge kernel and the pyramid on the ctx stream (the flow started that much later).
  int (*after_edges)(void *) = nullptr;
  void *after_edges_arg = nullptr;
  // plv_decision_trace: the values behind every verdict of the point update stay on the device until plv_last_point_decisions asks
  bool decision_trace = false;
  plv::DevBuf d_tri_dbg, d_gate_dec, d_gate_dec_l;  // (_l: the line update's gate)
  int dec_F_l = 0;        // entries of the last line batch whose gate values are in d_gate_dec_l (0: none)
  int dec_F = 0;          // entries of the last point batch whose values are in d_tri_dbg
  bool dec_gate = false;  // ... and in d_gate_dec: that batch reached a gate
  plv::DevBuf d_stack, d_stack2;            // stacked [H | r] and TSQR ping-pong
  // The line half's stacked rows have their own buffer: a line launch chained behind a point update that is still running
  // reserves its stack inside the point update's wait, and a reserve that grows frees the old block — the point update's re-run
  // (plv_msckf_update_resident_wait: redo / redo_w) would read rows that are gone.
  plv::DevBuf d_stack_l;
  plv::DevBuf &stack_of(int fdim) { return fdim == 6 ? d_stack_l : d_stack; }
  // whitened route: prior factor Lp^T, W0 = Lp^-1 P[cols, :], W0^T W0 (side stream), information matrix [G | g] (main stream)
  plv::DevBuf d_Lt, d_W0, d_dW, d_Gs, d_GP, d_Y0, d_C1, d_stackc;
  plv::DevBuf d_count_words; // int: [0] the rows a device-side gather of the stack left (launch_stack_compact -> hqr_kernel, mode 1)
  plv::DevBuf d_prior_near;  // int: near-dependent pivots the last prior factor met (blocked_chol.hip, PLV_PRIOR_AMB)
  hipStream_t aux_stream = nullptr;  // work that only needs the covariance, concurrent with the Jacobians and the gate
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  bool aux_fork_needed = false;  // the main stream held work when the side work was marked: aux_fork orders the side stream behind it
  bool prior_pending = false;  // plv_prior_prefetch started the prior factor for the update about to be launched (k = prior_k)
  int prior_k = 0;
  plv::PinBuf h_pin;
  plv::PinBuf h_pin_flow;  // points in / results out of the flow (plv_perform_matching_launch / _wait): a block of its own since round 6 — the point
                           // update enqueued behind the flow (Tracker::Spec) may write ITS result block (h_pin) before the host has read the flow's
  plv::PinBuf h_pin_l;  // result block of a LINE update (fdim 6): the point update's block may still be unread when the line gate writes
  plv::PinBuf &res_pin(int fdim) { return fdim == 6 ? h_pin_l : h_pin; }
  // What a line launch chained behind the point update needs beyond its own inputs (JacParams::chain_dx ...): filled by
  // plv_camera_try_update (quaternions and covariance indices from the caller's variable list) and by the line half's submit
  // (anchor candidates); `on` while a chained submit is being staged
  struct ChainState {
    bool ready = false, on = false;
    std::vector<double> q;  // [n_clones][4]
    std::vector<int> ids;   // [n_clones + 3]
    double qe[4] = {0, 0, 0, 1};
    std::vector<int> anc_ptr, anc_f;
    std::vector<unsigned char> anc_has_old;
    std::vector<double> anc_old;
  } chain;
  int *applied_word = nullptr;  // set by the caller of a launch that may end in ekf_commit_kernel: the kernel stores "state changed" there
  bool applied_used = false;
  // set by the caller of a speculative point update (plv_points_update_submit) around its launch: ekf_commit_kernel's cap test
  const int *cap_words = nullptr;
  int cap = 0;
  // Measurement knob PLV_KNOB_DONE_WORDS — completion words in pinned memory: the last kernel of the flow (word 0) and of an update
  // (word 16) stores the call's sequence number there behind its result block (system-scope release) and the host spins on the word
  // instead of waiting on an event.  Off by default (no gain in the frame, see the knob list).
  plv::PinBuf h_done;
  unsigned match_seq = 0, update_seq = 0;
  bool update_word_armed = false;  // the update being enqueued may end in a kernel that stores update_seq to done_word(16) ...
  bool update_word_used = false;   // ... and did
  volatile unsigned *done_word(int i) { return h_done.p ? (volatile unsigned *)h_done.p + i : nullptr; }

  // Device word holding the number of features the gate accepted in the update being enqueued (null outside
  // plv_msckf_update_resident_launch): the compression and EKF kernels return at once when it is zero — an update in which the gate
  // took nothing (most line updates) costs their launches, not their pivot chains; ekf_commit_kernel then reports dx = 0.
  const int *skip_word = nullptr;
  // a second block the update's last kernel copies to pinned host memory next to its result block (the triangulation results of the
  // one-submission updates): set by the caller before plv_msckf_update_resident_launch, cleared (taken = true) when the chain ended in
  // the kernel that does it — else the caller enqueues a copy command as before
  const void *mirror2_src = nullptr;
  void *mirror2_dst = nullptr;
  size_t mirror2_bytes = 0;
  bool mirror2_taken = false;
  // gate probe of the one-submission line update: plv_msckf_update_resident_launch waits for the gate, reads its verdicts from
  // pinned memory and enqueues compression + EKF only when something was accepted (probe_* describe the second block the gate's
  // workgroups copy, update_kernels.hpp; probe_hook runs right before that wait).  probe_done: the update ended at the gate.
  bool probe = false, probe_done = false;
  const void *probe_src = nullptr;
  void *probe_dst = nullptr;
  int probe_stride_a = 0, probe_off_b = 0, probe_stride_b = 0;
  void (*probe_hook)(void *) = nullptr;
  void *probe_hook_arg = nullptr;

  // the gate as the tail of the next projected Jacobian launch (gate_core.hpp): filled by plv_update_gate_prepare for the update the
  // one-submission entry points are about to build; the launcher takes it (gate_stage_taken) when the batch fits, and
  // plv_msckf_update_resident_launch then starts behind the gate
  plv::GateStage gate_stage{};
  bool gate_stage_taken = false;
  int gate_rows_hint = 0;  // most rows any entry of the batch about to be built can have (0: unknown, the batch's row capacity counts)

  // host work of the caller that becomes possible while a point update runs on the device (plv_points_update_fused polls it inside
  // its wait until it returns nonzero = done / nothing to do)
  int (*wait_poll)(void *) = nullptr;
  void *wait_poll_arg = nullptr;

  // ---- front-end (frontend_api.hip owns the object)
  void *fe_state = nullptr;
};
