// line_host.hpp — the HOST stage of the line front-end, free of device calls: FastLineDetector's chain walk on the Canny map, the
// segment growth along the chains on the library's fitter threads (fld_fit_core.hpp), and TrackLSD's point-line assignment, line
// matching and classification (REF: PL-VIWO/src/update/cam/TrackLSD.cpp:318-407,744-830; OpenCV contract of
// ximgproc::FastLineDetector).  line_api.hip builds the tracker around it; tests/host_sanitize/ compiles this
// header alone with -fsanitize=thread / address and drives the thread protocol with recorded edge maps.
#pragma once
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "fld_fit_core.hpp"
#include "line_kernels.hpp"

namespace plv {
namespace linehost {

// one detection's host stage: the maps a job points to and its result
struct Job {
  int device = 0, w = 0, h = 0, length_threshold = 0;
  float distance_threshold = 0, thr2 = 0;
  const uint8_t *hmap = nullptr, *hhalf = nullptr;
  // Component labels of the edge pixels (ccl_merge_kernel + ccl_flatten_kernel, line_kernels.hip): 0 = not an edge, else 1 + (a hash of
  // the root of the pixel's 8-connected component) % parts.  With them the host stage splits the detection by components (host_extract).
  const uint8_t *hlab = nullptr;
  int parts = 0;
  // (nullable) the parts' pixels in raster order per run of 256 pixels (ccl_flatten_kernel: blk_sorted / blk_bins; lists_from_labels
  // forms the same on the host): a part is then staged and seeded from its own pixels (detect_part)
  const uint8_t *hsorted = nullptr;
  const unsigned short *hbins = nullptr;
  int2 *hpts = nullptr;
  FldChain *hc = nullptr;
  std::vector<float> lines;
  int rc = PLV_OK;
};

// Second half of the host stage on threads of their own (see host_extract)
struct Fit {
  static const int kThreads = 15;  // helper threads next to the walking thread (fit_threads() says how many a job uses)
  static const int kParts = 32;   // parts a labelled detection is split into (Job::parts <= kParts)
  std::thread th[kThreads];
  std::mutex m;
  std::condition_variable cv;
  // (atomics: written under m, read without it by a polling thread — wait_polling spins on them lock-free)
  std::atomic<int> gen{0};  // a job = a new generation
  // Round 6: the hand-over of a job takes no lock while the threads are awake.  The poster fills desc[g & 1] for the generation g it
  // is about to post and then stores g (under m only so that a helper asleep on cv cannot miss it); a helper that sees gen change
  // reads desc[g & 1] and checks that gen still is g — the slot is written again only for g + 2, after g + 1 has been stored, so an
  // unchanged gen proves the copy whole (a helper the job of generation g did not count may wake that late; one it counts is
  // waited for).  It reports in done_gen and takes m only when the poster has gone to sleep (poster_waiting).  Until round 5 every
  // pick-up and every report locked m: eight threads at once, 6 us until the last helper had started and 8 us until the poster had
  // seen the last report (measured on the assignment job, PLV_KNOB_LINE_TIMING).
  struct Desc {
    // a detection (job): by_parts = its parts are claimed by the walking thread and the helpers, a part = the components whose label
    // it is, walked and fitted by the thread that claimed it (detect_part) into part[p]; or a generic job for the same threads
    // (run_on_helpers): every thread it counts calls (*task)(slot), slot 1 .. nfit (0 = the poster)
    std::atomic<const Job *> job{nullptr};
    std::atomic<int> nfit{0};  // helper threads the job counts (thread i takes part when i < nfit): read with the rest of the slot,
                               // so that a thread waking late for a job it was not part of cannot mix two jobs
    std::atomic<bool> closable{false};  // the poster closes the job itself (parts, generic jobs); false: it waits for every report
    std::atomic<bool> by_parts{false};
    std::atomic<const std::function<void(int)> *> task{nullptr};
  } desc[2];
  alignas(64) std::atomic<int> done_gen[kThreads] = {};  // thread i reports the last generation it finished
  // Round 6b: a job no longer waits for a helper that is late or has been descheduled in the middle of its share (on a host shared
  // with other tenants one frame in ten used to wait 1-9 ms for such a thread: bench trace, w_extract).  picked_gen[i] = the last
  // generation helper i has started on (stored before it touches anything of the job); closed_gen = every share of the jobs up to
  // this generation is taken care of: a helper that arrives later leaves the job alone, one inside a part of it stops at its next
  // seed / chain.  The poster does what nobody has started (generic jobs: slot_claim) or does again what somebody has started and not
  // finished (parts: winner / part[p][1]), closes the job, and waits only for helpers that picked it up before it was closed (they
  // are running, or they hold the task's closure).  quiesce_helpers() — nobody is inside a job — comes before the buffers a part
  // reads or writes are reused: at the start of the next detection, not at the end of this one.
  // (tests: helper threads go to sleep for up to this many microseconds when they pick a job up and at every part they start — the
  // late and descheduled helpers of a busy host, on demand: tests/host_sanitize)
  std::atomic<int> chaos_us{0};
  std::atomic<long> second_runs{0}, second_run_wins{0};  // parts the poster ran again; of those, the ones whose second run counted
  alignas(64) std::atomic<int> picked_gen[kThreads] = {};
  alignas(64) std::atomic<int> closed_gen{0};
  std::atomic<int> slot_claim[kThreads + 1] = {};  // generic job: slot s has been started (by its helper or by the poster)
  std::atomic<int> winner[kParts] = {};            // detection by parts: -1 undecided, else whose output counts (0 the claimer's, 1 the poster's second run)
  alignas(64) std::atomic<int> poster_waiting{0};       // the poster blocks on cv (under m) for the reports
  std::atomic<int> prewake{0};  // bumped (under m) when a job is on its way (prewake_helpers): sleeping helpers wake and poll for it
  alignas(64) std::atomic<int> next_part{0};
  int claim_order[kParts] = {};  // the q-th claim takes part claim_order[q] (the largest parts first: host_extract)
  std::vector<std::pair<int, int>> merged, merge_tmp;  // (seed, part << 16 | chain) of the parts taken in so far, sorted
  struct PartOut {
    std::vector<int> seed, seg_at, seg_n;  // per chain: raster index of its seed, first segment, segments
    std::vector<float4> segs;
    std::vector<float> lines;   // the chains' segments that pass the detector's tail (x2, FilterShortLines), chain after chain
    std::vector<int> line_at;   // [chains + 1] chain c's lines: line_at[c] .. line_at[c + 1]
    int chains = 0;
    float us_build = 0, us_walk = 0, us_fit = 0, us_start = 0;  // (reporting: PLV_KNOB_LINE_TIMING)
    int slot = 0, pixels = 0;
  } part[kParts][2];
  std::chrono::steady_clock::time_point job_t0;
  struct Scratch {  // one per thread (index 0: the walking thread)
    std::vector<uint8_t> pad;
    int clean_w = 0, clean_h = 0;  // pad is the bordered map of a clean_w x clean_h image with no edge left in it (what a finished walk leaves)
    std::vector<uint32_t> seeds;   // a listed part's pixels in raster order, (y << 16) | x
    std::vector<int2> pts;
    std::vector<FldChain> chains;
  } scratch[kThreads + 1];
  std::atomic<bool> quit{false};
  alignas(64) std::atomic<int> published{0};  // (own cache line: written by the walk after every chain, polled by this thread)
  alignas(64) std::atomic<bool> walk_done{false};
  alignas(64) std::atomic<int> next{0};  // next chain to fit: this thread and, once its walk is over, the walking thread claim chains here
  alignas(64) std::vector<float4> segs;  // chain c's segments at its slot (FldChain::slot)
  std::vector<int> seg_n;    // per chain
};

struct HostStage {  // what the host stage keeps between detections
  Fit fit;
  std::vector<uint8_t> pad;  // bordered copy of the edge map for the walk
  ~HostStage() {
    if (fit.th[0].joinable()) {
      {
        std::lock_guard<std::mutex> lk(fit.m);
        fit.quit = true;
      }
      fit.cv.notify_all();
      for (auto &th : fit.th)
        if (th.joinable()) th.join();
    }
  }
};

const int kChainCap = 4096;

// Condition wait that polls first: the library's threads hand each other work several times per frame and a thread that blocked
// pays tens of microseconds (sometimes a millisecond) to be woken.  Polls for up to spin_us (the lock is released between probes),
// then blocks as usual — at camera rates the threads sleep between frames, back to back frames keep them awake.
// How long a waiting library thread polls before it blocks: process-wide, PLV_LINE_SPIN_US (default 300: the hand-overs inside one
// frame follow each other within that time, between frames the threads sleep; 0 = block at once, measured +10..25 us per frame) or
// plv_line_worker_config.
inline std::atomic<int> &spin_budget_us() {
  static std::atomic<int> v{getenv("PLV_LINE_SPIN_US") ? atoi(getenv("PLV_LINE_SPIN_US")) : 300};
  return v;
}
// The most fitter threads the segment growth uses next to the walking thread (0 .. Fit::kThreads; 0 = the walk's own thread fits
// afterwards).  host_extract takes one of them for a small map and both for a large one (see there).  Whole runs at workload C: 0.56 ms
// per frame with one or two, 0.70 ms with none (without a fitter the worker finishes after the point update and the caller waits for it).
// Default: seven (one 8-core complex holds the walking thread + seven helpers; 5, 7 and 11 measured the same there), but never more
// than the CPUs this process may run on minus two — the caller's thread and the line worker are on the frame's critical path and a
// polling helper that shares their CPU preempts them (a small control group, several contexts per node) — and at most
// two when fewer than six CPUs are allowed.  INTEGRATION.md "Host threads".
inline int default_fit_threads() {
  int ncpu = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) ncpu = CPU_COUNT(&set);
  if (ncpu <= 0) ncpu = (int)std::thread::hardware_concurrency();
  int n = ncpu - 2;
  if (ncpu < 6) n = std::min(n, 2);
  return std::max(0, std::min(7, n));
}
inline std::atomic<int> &fit_threads() {
  static std::atomic<int> v{getenv("PLV_LINE_FIT_THREADS") ? std::max(0, std::min((int)Fit::kThreads, atoi(getenv("PLV_LINE_FIT_THREADS")))) : default_fit_threads()};
  return v;
}
template <class Pred>
inline void wait_polling(std::unique_lock<std::mutex> &lk, std::condition_variable &cv, Pred pred, int spin_us = -1) {
  // `pred` reads atomics only (they are written under the mutex): the polling phase runs WITHOUT the mutex — round 5: eight threads
  // polling by locking and unlocking it kept the thread that wanted to post a job out of it for tens of microseconds
  if (pred()) return;
  if (spin_us < 0) spin_us = spin_budget_us().load(std::memory_order_relaxed);
  if (spin_us > 0) {
    lk.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
      if (pred()) break;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
    lk.lock();
    if (pred()) return;
  }
  cv.wait(lk, pred);
}

inline float point_line_distance(const float *line, float x0, float y0) {
  const float x1 = line[0], y1 = line[1], x2 = line[2], y2 = line[3];
  const float along = (x2 - x1) * (x0 - x1) + (y2 - y1) * (y0 - y1);
  if (along <= 0) return std::sqrt((x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1));
  const float len2 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
  if (along > len2) return std::sqrt((x0 - x2) * (x0 - x2) + (y0 - y2) * (y0 - y2));
  return std::abs(std::fabs((y2 - y1) * x0 + (x1 - x2) * y0 + ((x2 * y1) - (x1 * y2))) /
                  (std::sqrt(std::pow(y2 - y1, 2) + std::pow(x1 - x2, 2))));
}

// FastLineDetector's seed loop + getPointChain on a host copy of the Canny map (2 = edge).  Same
// algorithm as fld_walk_kernel; see detect() for why the default runs it here.  The map is copied into a
// buffer with a one-pixel non-edge border so that the eight neighbour tests need no bounds checks.
inline void walk_padded(uint8_t *m, int w, int h, int length_threshold, int2 *pts, FldChain *chains, int chain_cap, int *counts,
                        std::atomic<int> *published = nullptr);
inline void walk_chains(const uint8_t *map, int w, int h, int length_threshold, int2 *pts, FldChain *chains, int chain_cap, int *counts,
                 std::vector<uint8_t> &pad, std::atomic<int> *published = nullptr) {
  const int pw = w + 2;
  // bordered copy of the map (border = 1: never an edge)
  pad.resize((size_t)pw * (h + 2));
  uint8_t *m = pad.data();
  memset(m, 1, (size_t)pw);
  memset(m + (size_t)(h + 1) * pw, 1, (size_t)pw);
  for (int r = 0; r < h; ++r) {
    uint8_t *row = m + (size_t)(r + 1) * pw;
    row[0] = 1;
    memcpy(row + 1, map + (size_t)r * w, w);
    row[w + 1] = 1;
  }
  walk_padded(m, w, h, length_threshold, pts, chains, chain_cap, counts, published);
}
// (the walk proper, on a map with a one-pixel border that is never an edge; it clears what it consumes)
// next(x, y): the next pixel in raster order that may still be an edge (the walk tests it), false when there is none left
// (a seed source may end the walk early — a part somebody else has finished meanwhile: the map then still holds edges)
template <class NextSeed>
inline void walk_core(uint8_t *m, int w, int length_threshold, int2 *pts, FldChain *chains, int chain_cap, int *counts, std::atomic<int> *published,
                      NextSeed next) {
  static const int dx[8] = {1, 0, -1, -1, -1, 0, 1, 1}, dy[8] = {1, 1, 1, 0, -1, -1, -1, 0};
  const int pw = w + 2;
  int off[8];
  for (int i = 0; i < 8; ++i) off[i] = dy[i] * pw + dx[i];
  int n_chain = 0, n_slot = 0, n_pts = 0;
  for (int x, y; next(x, y);) {
    size_t idx = (size_t)(y + 1) * pw + x + 1;
    if (m[idx] != 2) continue;
    const int start = n_pts;
    pts[n_pts++] = make_int2(x, y);
    m[idx] = 1;
    float direction = 0.0f;
    for (int step = 0;; ++step) {
      // (eight byte tests per step.  Measured against it on the bench scene's maps: the neighbourhood read as three 32-bit words
      // and a bit mask is 1.7x SLOWER — the words overlap the byte the previous step has just cleared and wait for that store — and
      // eight branch-free byte loads + a loop over the set bits 1.15x slower: chains are mostly straight, the branches predict, and
      // what a step costs is the float recurrence of `direction` (multiply, add, divide: ~20 cycles), which must stay as it is.)
      int pick = -1;
      float best = 7.0f;
      if (step == 0) {
        for (int i = 0; i < 8; ++i)
          if (m[idx + off[i]] == 2) {
            pick = i;
            break;
          }
      } else {
        for (int i = 0; i < 8; ++i) {
          if (m[idx + off[i]] != 2) continue;
          const float curr = i > 4 ? (float)(i - 8) : (float)i;
          float diff = std::fabs(curr - direction);
          diff = diff > 4.0f ? 8.0f - diff : diff;
          if (diff <= best) {
            best = diff;
            pick = i;
          }
        }
      }
      if (pick < 0 || (step > 0 && !(best < 2.0f))) break;
      const int cdir = pick > 4 ? pick - 8 : pick;
      direction = step == 0 ? (float)cdir : (direction * (float)step + (float)cdir) / (float)(step + 1);
      x += dx[pick];
      y += dy[pick];
      idx += off[pick];
      pts[n_pts++] = make_int2(x, y);
      m[idx] = 1;
    }
    const int len = n_pts - start;
    if (len >= length_threshold + 1 && n_chain < chain_cap) {
      chains[n_chain++] = FldChain{start, len, n_slot};
      n_slot += len / length_threshold + 1;
      if (published) published->store(n_chain, std::memory_order_release);  // chain n_chain - 1 and its points are final
    } else {
      n_pts = start;
    }
  }
  counts[0] = n_chain;
  counts[1] = n_slot;
  counts[2] = n_pts;
}
inline void walk_padded(uint8_t *m, int w, int h, int length_threshold, int2 *pts, FldChain *chains, int chain_cap, int *counts,
                        std::atomic<int> *published) {
  // the seeds of a whole map: a vectorised byte search per row instead of a test per pixel (90 000 pixels, 20 000 of them edges, most
  // of those consumed by earlier chains by the time the scan reaches them)
  const int pw = w + 2;
  int r = 0, c = 0;
  walk_core(m, w, length_threshold, pts, chains, chain_cap, counts, published, [&](int &x, int &y) {
    for (; r < h; ++r, c = 0) {
      const uint8_t *row = m + (size_t)(r + 1) * pw + 1;
      const uint8_t *nx = c < w ? (const uint8_t *)memchr(row + c, 2, (size_t)(w - c)) : nullptr;
      if (!nx) continue;
      x = (int)(nx - row), y = r;
      c = x + 1;
      return true;
    }
    return false;
  });
}
// the seeds of a listed part: its pixels in raster order ((y << 16) | x)
template <class Stop>
inline void walk_listed(uint8_t *m, int w, const uint32_t *seeds, size_t n_seeds, int length_threshold, int2 *pts, FldChain *chains, int chain_cap,
                        int *counts, Stop stop) {
  size_t at = 0;
  walk_core(m, w, length_threshold, pts, chains, chain_cap, counts, nullptr, [&](int &x, int &y) {
    if (at >= n_seeds || stop()) return false;
    const uint32_t s = seeds[at++];
    x = (int)(s & 0xffffu), y = (int)(s >> 16);
    return true;
  });
}

// What ccl_flatten_kernel leaves next to the labels, formed from the labels on the host (the sanitizer driver and the tests): per run
// of 256 pixels the run's pixels grouped by part, raster order inside a group.
inline void lists_from_labels(const uint8_t *lab, int n, int parts, std::vector<uint8_t> &sorted, std::vector<unsigned short> &bins) {
  const int nblk = (n + 255) / 256;
  sorted.assign((size_t)nblk * 256, 0);
  bins.assign((size_t)nblk * (parts + 1), 0);
  for (int b = 0; b < nblk; ++b) {
    unsigned short *bn = bins.data() + (size_t)b * (parts + 1);
    int at = 0;
    for (int p = 0; p < parts; ++p) {
      bn[p] = (unsigned short)at;
      for (int o = 0; o < 256 && b * 256 + o < n; ++o)
        if (lab[b * 256 + o] == p + 1) sorted[(size_t)b * 256 + at++] = (uint8_t)o;
    }
    bn[parts] = (unsigned short)at;
  }
}

// The host stage on the maps a job points to: chains in raster order of their seeds = the detector's output order; the tail of
// perform_detection_monocular (x2, FilterShortLines) on every segment.
// claims the next unfitted chain below `avail`, or -1
inline int claim_chain(Fit &F, int avail) {
  int c = F.next.load(std::memory_order_relaxed);
  while (c < avail && !F.next.compare_exchange_weak(c, c + 1, std::memory_order_relaxed)) {
  }
  return c < avail ? c : -1;
}
inline void fit_one(Fit &F, const Job &J, int c) {
  F.seg_n[c] = fit_chain(J.hhalf, J.w, J.h, J.length_threshold, J.distance_threshold, J.hpts + J.hc[c].start, J.hc[c].len,
                         F.segs.data() + J.hc[c].slot);
}

// One part of a labelled detection: the components whose label is `p + 1`, alone on a map of their own — a pixel's eight neighbours
// that are edges lie in its own component, and the walk only ever reads and clears those, so the raster walk over this map yields
// exactly the chains the raster walk over the whole map yields inside these components, in the same order; the segments of a chain
// depend on nothing but the chain.  The caller puts the parts' chains back into the raster order of their seeds.
// kind: whose run this is — 0 the thread that claimed the part, 1 the poster doing it again because that thread has not finished
// (host_extract); g: the job's generation.  The run gives up at its next seed / chain when the part has been decided by the other run
// or the job has been closed; the one that gets to the end first sets winner[p] and its output (part[p][kind]) counts.
inline void detect_part(Fit &F, const Job &J, int p, Fit::Scratch &S, int slot, int kind, int g) {
  const auto tp0 = std::chrono::steady_clock::now();
  auto stop = [&] { return F.winner[p].load(std::memory_order_relaxed) >= 0 || F.closed_gen.load(std::memory_order_relaxed) >= g; };
  if (stop()) return;  // (claimed just as the job was closed: nothing of the job is touched)
  const int w = J.w, h = J.h, pw = w + 2;
  const bool listed = J.hsorted != nullptr && J.hbins != nullptr && w < 65536 && h < 65536;
  if (S.pad.size() != (size_t)pw * (h + 2) || (listed && (S.clean_w != w || S.clean_h != h))) {
    S.pad.assign((size_t)pw * (h + 2), 1);
    S.clean_w = w, S.clean_h = h;
  }
  uint8_t *m = S.pad.data();
  if (listed) {
    // The part's own pixels (a fraction of a per cent of the map) instead of a pass over the labels of the whole image per part: the
    // thread's map holds no edge when a walk is over — every listed pixel is a seed or a chain's — so staging a part is writing its
    // pixels, and its seeds are those pixels in the order they are listed in (raster order).
    const int n = w * h, nblk = (n + 255) / 256, stride = J.parts + 1;
    S.seeds.clear();
    int row = 0, col = 0;  // of the run's first pixel (a run is 256 pixels: it ends on the same row or the next, w >= 256, or later)
    for (int b = 0; b < nblk; ++b) {
      const unsigned short *bn = J.hbins + (size_t)b * stride;
      const uint8_t *so = J.hsorted + (size_t)b * 256;
      // (bounds held against any content: a detection that has been superseded — plv_line_detect_launch for another image, then a
      // detection from scratch — may still be reading these lists while the next edge launch rewrites them; its result is dropped, but
      // a pixel outside the image would put an edge on the map's border and the walk off the map)
      for (int q = std::min<int>(bn[p], 256), q1 = std::min<int>(bn[p + 1], 256); q < q1; ++q) {
        if (b * 256 + (int)so[q] >= n) continue;
        int x = col + so[q], y = row;
        while (x >= w) x -= w, ++y;
        m[(size_t)(y + 1) * pw + x + 1] = 2;
        S.seeds.push_back((uint32_t)y << 16 | (uint32_t)x);
      }
      col += 256;
      while (col >= w) col -= w, ++row;
    }
  } else {
    memset(m, 1, (size_t)pw);
    memset(m + (size_t)(h + 1) * pw, 1, (size_t)pw);
    const uint8_t want = (uint8_t)(p + 1);
    for (int r = 0; r < h; ++r) {
      uint8_t *row = m + (size_t)(r + 1) * pw;
      const uint8_t *lr = J.hlab + (size_t)r * w;
      row[0] = 1;
      for (int x = 0; x < w; ++x) row[x + 1] = (uint8_t)(1 + (lr[x] == want));
      row[w + 1] = 1;
    }
  }
  S.pts.resize((size_t)w * h);
  S.chains.resize(kChainCap);
  int counts[4] = {0, 0, 0, 0};
  const auto tp1 = std::chrono::steady_clock::now();
  if (listed)
    walk_listed(m, w, S.seeds.data(), S.seeds.size(), J.length_threshold, S.pts.data(), S.chains.data(), kChainCap, counts, stop);
  else
    walk_padded(m, w, h, J.length_threshold, S.pts.data(), S.chains.data(), kChainCap, counts);
  S.clean_w = w, S.clean_h = h;  // (either walk has consumed every edge of the map)
  if (stop()) {  // (a walk that was cut short may have left edges behind)
    if (listed)
      for (const uint32_t sd : S.seeds) m[(size_t)((sd >> 16) + 1) * pw + (sd & 0xffffu) + 1] = 1;
    return;
  }
  const auto tp2 = std::chrono::steady_clock::now();
  Fit::PartOut &O = F.part[p][kind];
  O.chains = counts[0];
  O.seed.resize(counts[0]), O.seg_at.resize(counts[0]), O.seg_n.resize(counts[0]);
  O.segs.resize((size_t)counts[1] + 1);
  for (int c = 0; c < counts[0] && c < kChainCap; ++c) {
    const FldChain &ch = S.chains[c];
    O.seed[c] = S.pts[ch.start].y * w + S.pts[ch.start].x;
    O.seg_at[c] = ch.slot;
    O.seg_n[c] = fit_chain(J.hhalf, w, h, J.length_threshold, J.distance_threshold, S.pts.data() + ch.start, ch.len, O.segs.data() + ch.slot);
    if ((c & 7) == 7 && stop()) return;
  }
  // the detector's tail on the part's own thread: what host_extract copies out in the order of the seeds
  O.lines.clear();
  O.line_at.resize((size_t)counts[0] + 1);
  for (int c = 0; c < counts[0] && c < kChainCap; ++c) {
    O.line_at[c] = (int)O.lines.size() / 4;
    for (int q = 0; q < O.seg_n[c]; ++q) {
      const float4 &sg = O.segs[O.seg_at[c] + q];
      const float x1 = sg.x * 2, y1 = sg.y * 2, x2 = sg.z * 2, y2 = sg.w * 2;  // REF :218-220
      const float l2 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
      if (!(l2 > J.thr2)) continue;  // FilterShortLines(lines0, 40)   REF :232, :435-448
      O.lines.insert(O.lines.end(), {x1, y1, x2, y2});
    }
  }
  O.line_at[counts[0]] = (int)O.lines.size() / 4;
  const auto tp3 = std::chrono::steady_clock::now();
  auto us = [](auto a, auto b) { return std::chrono::duration<float, std::micro>(b - a).count(); };
  O.us_start = us(F.job_t0, tp0), O.us_build = us(tp0, tp1), O.us_walk = us(tp1, tp2), O.us_fit = us(tp2, tp3), O.slot = slot, O.pixels = counts[2];
  int undecided = -1;
  F.winner[p].compare_exchange_strong(undecided, kind, std::memory_order_acq_rel);
}
inline void chaos_nap(Fit &F, unsigned a, unsigned b) {
  int c = F.chaos_us.load(std::memory_order_relaxed);
  if (c <= 0 && plv::knob(plv::PLV_KNOB_HELPER_NAPS)) {
    static const int env_us = [] {
      const char *e = getenv("PLV_HELPER_NAP_US");  // (tests: longer naps than the knob's 200 us)
      return e ? atoi(e) : 0;
    }();
    c = env_us > 0 ? env_us : 200;
  }
  if (c <= 0) return;
  unsigned x = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u;
  x ^= x >> 15, x *= 2246822519u, x ^= x >> 13;
  if (x & 1) std::this_thread::sleep_for(std::chrono::microseconds((x >> 8) % (unsigned)c));
}
inline void claim_parts(Fit &F, const Job &J, int slot, int g) {
  while (F.closed_gen.load(std::memory_order_acquire) < g) {
    const int q = F.next_part.fetch_add(1, std::memory_order_relaxed);
    if (q >= J.parts) break;
    const int p = F.claim_order[q];
    if (slot) chaos_nap(F, (unsigned)(g * 64 + p), (unsigned)slot);
    detect_part(F, J, p, F.scratch[slot], slot, 0, g);
  }
}

// posts generation g = gen + 1 (the caller has filled nothing yet): fills its slot, stores g, wakes sleepers.  One poster at a time
// (the line worker, or the caller's thread while the worker is idle: both hold the tracker's lock).
inline int post_job(Fit &F, const Job *job, int nfit, bool by_parts, const std::function<void(int)> *task) {
  const int g = F.gen.load(std::memory_order_relaxed) + 1;
  Fit::Desc &D = F.desc[g & 1];
  D.job.store(job, std::memory_order_relaxed), D.nfit.store(nfit, std::memory_order_relaxed);
  D.by_parts.store(by_parts, std::memory_order_relaxed), D.task.store(task, std::memory_order_relaxed);
  D.closable.store(by_parts || task != nullptr, std::memory_order_relaxed);
  {
    std::lock_guard<std::mutex> lk(F.m);
    F.gen.store(g, std::memory_order_release);
  }
  F.cv.notify_all();
  return g;
}
// polls, then blocks on cv (a report then takes m and notifies) until done() holds
template <class Done>
inline void wait_helpers(Fit &F, Done done);
// waits until the first nfit helpers have reported generation g
inline void wait_reports(Fit &F, int nfit, int g) {
  wait_helpers(F, [&] {
    for (int i = 0; i < nfit; ++i)
      if (F.th[i].joinable() && F.done_gen[i].load() != g) return false;
    return true;
  });
}
// closes job g (the poster has seen to every share of it) and waits for the helpers that picked it up before that: they are inside
// one of its shares — a part stops at its next seed / chain — or about to find nothing left.  A helper that looks at the job later
// sees it closed and leaves it alone (either its picked_gen store comes first in the single order of these operations and is seen
// here, or the closed_gen store does and is seen there), so what the job points to — a closure on the poster's stack — may go.
inline void close_job(Fit &F, int nfit, int g, bool wait = true) {
  F.closed_gen.store(g);
  if (!wait) return;  // (a detection by parts: what its late helpers touch stays in place until quiesce_helpers)
  wait_helpers(F, [&] {
    for (int i = 0; i < nfit; ++i)
      if (F.th[i].joinable() && F.picked_gen[i].load() == g && F.done_gen[i].load() != g) return false;
    return true;
  });
}
// nobody is inside a job (a helper cut off in the middle of a part of the last detection, which was closed without it, may still be):
// before the buffers the parts read (the pinned maps and lists) or write (Fit::part, winner) are used again
inline void quiesce_helpers(Fit &F) {
  wait_helpers(F, [&] {
    for (int i = 0; i < Fit::kThreads; ++i)  // (the two words only: this may run on another thread than the one that starts helpers)
      if (F.picked_gen[i].load() != F.done_gen[i].load()) return false;
    return true;
  });
}
template <class Done>
inline void wait_helpers(Fit &F, Done done) {
  if (done()) return;
  const int spin_us = spin_budget_us().load(std::memory_order_relaxed);
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int i = 0; i < 8; ++i) __builtin_ia32_pause();
      if (done()) return;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
  }
  std::unique_lock<std::mutex> lk(F.m);
  F.poster_waiting.fetch_add(1);  // (a count: the worker and, in quiesce_helpers, the caller's thread may both be asleep here)
  F.cv.wait(lk, done);
  F.poster_waiting.fetch_sub(1);
}

inline void fit_worker(HostStage *T, int me, int seen /* the generation current when the thread was made: it waits for the next */) {
  Fit &F = T->fit;
  int seen_pw = 0;
  for (;;) {
    auto changed = [&] { return F.gen.load(std::memory_order_acquire) != seen || F.prewake.load(std::memory_order_acquire) != seen_pw || F.quit.load(); };
    if (!changed()) {
      // (a pre-wake ends the wait without a job: the thread comes round and polls again — awake when the job arrives; a helper beyond
      // the configured count — the count was lowered after it was started — blocks at once: a pre-wake makes only the helpers the
      // next job will use poll)
      const int spin_us = me < fit_threads().load(std::memory_order_relaxed) ? spin_budget_us().load(std::memory_order_relaxed) : 0;
      bool hit = false;
      if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
          for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
          if ((hit = changed())) break;
          if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
      }
      if (!hit) {
        std::unique_lock<std::mutex> lk(F.m);
        F.cv.wait(lk, changed);
      }
    }
    if (F.quit.load()) return;
    seen_pw = F.prewake.load(std::memory_order_acquire);
    const int g = F.gen.load(std::memory_order_acquire);
    if (g == seen) continue;
    const Fit::Desc &D = F.desc[g & 1];
    const Job *job = D.job.load(std::memory_order_relaxed);
    const int nfit = D.nfit.load(std::memory_order_relaxed);
    const bool by_parts = D.by_parts.load(std::memory_order_relaxed), closable = D.closable.load(std::memory_order_relaxed);
    const std::function<void(int)> *task = D.task.load(std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_acquire);
    if (F.gen.load(std::memory_order_relaxed) != g) continue;  // (posted over while this thread was reading: it was not part of g)
    seen = g;
    if (me >= nfit) continue;  // (a thread this job does not count takes nothing and is not waited for)
    chaos_nap(F, (unsigned)g, (unsigned)me + 77u);
    F.picked_gen[me].store(g);  // (before anything of the job is touched: close_job)
    if (closable && F.closed_gen.load() >= g) {  // too late: the poster has done this thread's share
      F.done_gen[me].store(g);
      if (F.poster_waiting.load()) {
        { std::lock_guard<std::mutex> lk(F.m); }
        F.cv.notify_all();
      }
      continue;
    }
    if (task) {
      int unclaimed = 0;
      if (F.slot_claim[me + 1].compare_exchange_strong(unclaimed, 1)) (*task)(me + 1);
    } else if (job) {
      const Job &J = *job;
      if (by_parts) claim_parts(F, J, me + 1, g);
      for (; !by_parts;) {
        const int avail = F.published.load(std::memory_order_acquire);
        const int c = claim_chain(F, avail);
        if (c >= 0) {
          fit_one(F, J, c);
          continue;
        }
        if (F.walk_done.load(std::memory_order_acquire) && F.next.load(std::memory_order_relaxed) >= F.published.load(std::memory_order_acquire)) break;
        for (int i = 0; i < 64; ++i) __builtin_ia32_pause();  // (poll every few hundred ns: the walk owns the counter's cache line meanwhile)
      }
    }
    F.done_gen[me].store(g);
    if (F.poster_waiting.load()) {
      { std::lock_guard<std::mutex> lk(F.m); }
      F.cv.notify_all();
    }
  }
}

inline int host_extract(HostStage *T, Job &J, bool timing) {
  auto T1 = std::chrono::steady_clock::now();
  int hcounts[4] = {0, 0, 0, 0};
  Fit &F = T->fit;
  F.segs.resize((size_t)J.w * J.h / (size_t)std::max(1, J.length_threshold) + kChainCap);
  F.seg_n.resize(kChainCap);
  F.published.store(0, std::memory_order_relaxed);
  F.next.store(0, std::memory_order_relaxed);
  F.walk_done.store(false, std::memory_order_relaxed);
  // one fitter keeps pace with the walk of a 376 x 240 map (BASELINE configs[2]) on a quiet host; a 640 x 360 map (configs[3]: 2.5 x the
  // chains, the fit left over at the end of the walk 110 us with one fitter, 20 with two) needs the second one.  Since round 4 the
  // worker's path decides whether the line launch can be chained behind the point update (tracker_api.hip poll_line_pool), so the
  // second fitter joins at configs[2] as well: alternating frame by frame -9 us on the mean and -80 .. -130 us on p99
  // (bench.py --alternate-fit 1,2).  Tiny maps keep one: every hand-over between threads is a chance of a delayed wake-up.
  // A labelled job is split by components over the walking thread and every configured helper (detect_part); without labels the walk
  // is one sequence and the helpers only grow segments behind it.
  const bool by_parts = J.hlab != nullptr && J.parts >= 1 && J.parts <= Fit::kParts;
  const int nfit = by_parts ? std::min(fit_threads().load(std::memory_order_relaxed), J.parts - 1)
                            : std::min(std::min(fit_threads().load(std::memory_order_relaxed), 2), (size_t)J.w * J.h >= 60000 ? 2 : 1);
  quiesce_helpers(F);  // (a helper cut off inside a part of the last detection: Fit::part and winner are about to be written again, and
                       // the job it reads may be another one by now)
  if (by_parts) {
    for (int p = 0; p < J.parts; ++p) F.winner[p].store(-1, std::memory_order_relaxed);
    // the order the parts are claimed in: the largest first (the threads' shares end level), sizes from the parts' pixel lists
    int size[Fit::kParts] = {};
    if (J.hbins) {
      const int nblk = (J.w * J.h + 255) / 256, stride = J.parts + 1;
      for (int b = 0; b < nblk; ++b) {
        const unsigned short *bn = J.hbins + (size_t)b * stride;
        for (int p = 0; p < J.parts; ++p) size[p] += bn[p + 1] - bn[p];
      }
    }
    for (int p = 0; p < J.parts; ++p) F.claim_order[p] = p;
    if (J.hbins) std::stable_sort(F.claim_order, F.claim_order + J.parts, [&](int a, int b) { return size[a] > size[b]; });
  }
  F.next_part.store(0, std::memory_order_relaxed);
  F.job_t0 = T1;
  {
    int gen_now;
    {
      std::lock_guard<std::mutex> lk(F.m);
      gen_now = F.gen;
    }
    for (int i = 0; i < nfit; ++i)
      if (!F.th[i].joinable()) F.th[i] = std::thread(fit_worker, T, i, gen_now);
  }
  const int gen = post_job(F, &J, nfit, by_parts, nullptr);
  int second_runs = 0;
  if (by_parts) {
    claim_parts(F, J, 0, gen);
    // Every part is claimed; those still undecided are with threads that have not finished — most of the time about to, now and then
    // descheduled for milliseconds (other tenants' work on the same cores).  This thread has nothing else to do: it runs such a part
    // again, into the part's second output, the earliest claimed first; whichever run ends first counts and the other gives up.
    // Every part is claimed; those still undecided are with threads that have not finished — most of the time about to, now and then
    // descheduled for milliseconds (other tenants' work on the same cores).  This thread takes the decided parts' chains into the
    // order of their seeds meanwhile (a part's chains come in that order: one merge of two sorted lists per part, all but the last
    // of them inside the wait), and when nothing is left to take in it runs an undecided part again, into the part's second output,
    // the earliest claimed first; whichever run ends first counts and the other gives up.
    const bool wait_all = plv::knob(plv::PLV_KNOB_WAIT_ALL_HELPERS);  // (measurement: rounds 5-6a waited for every helper's report)
    if (wait_all) wait_reports(F, nfit, gen);
    F.merged.clear();
    bool taken[Fit::kParts] = {};
    for (int left = J.parts; left > 0;) {
      bool progress = false;
      for (int p = 0; p < J.parts; ++p) {
        int wn;
        if (taken[p] || (wn = F.winner[p].load(std::memory_order_acquire)) < 0) continue;
        const Fit::PartOut &O = F.part[p][wn];
        F.merge_tmp.resize(F.merged.size() + (size_t)O.chains);
        size_t a = 0, o = 0;
        for (int c = 0; c < O.chains;) {
          if (a < F.merged.size() && F.merged[a].first < O.seed[c])
            F.merge_tmp[o++] = F.merged[a++];
          else
            F.merge_tmp[o++] = std::make_pair(O.seed[c], (p << 16) | c), ++c;
        }
        while (a < F.merged.size()) F.merge_tmp[o++] = F.merged[a++];
        F.merged.swap(F.merge_tmp);
        taken[p] = true, progress = true, --left;
      }
      if (progress || left == 0) continue;
      int q = 0;
      while (q < J.parts && F.winner[F.claim_order[q]].load(std::memory_order_acquire) >= 0) ++q;
      if (q == J.parts) continue;
      const int p = F.claim_order[q];
      ++second_runs;
      detect_part(F, J, p, F.scratch[0], 0, 1, gen);
      F.second_runs.fetch_add(1, std::memory_order_relaxed);
      if (F.winner[p].load(std::memory_order_relaxed) == 1) F.second_run_wins.fetch_add(1, std::memory_order_relaxed);
    }
  } else {
    walk_chains(J.hmap, J.w, J.h, J.length_threshold, J.hpts, J.hc, kChainCap, hcounts, T->pad, &F.published);
    F.walk_done.store(true, std::memory_order_release);
  }
  auto T2 = std::chrono::steady_clock::now();
  if (!by_parts)
    for (int c; (c = claim_chain(F, std::min(hcounts[0], kChainCap))) >= 0;) fit_one(F, J, c);  // the walk is over: share what is left
  if (by_parts)
    close_job(F, nfit, gen, false);  // (every part is decided; a helper still inside one stops at its next seed and is waited for where
                                     // the job's buffers are used again: quiesce_helpers)
  else
    wait_reports(F, nfit, gen);
  auto T2b = std::chrono::steady_clock::now();
  J.lines.clear();
  if (by_parts) {
    // the parts' chains in the raster order of their seeds = the detector's output order
    const int total = (int)F.merged.size();
    if (total >= kChainCap) {
      set_last_error("plv_detect_lines: more than %d edge chains", kChainCap);
      return PLV_E_CAPACITY;
    }
    const Fit::PartOut *out[Fit::kParts];
    for (int p = 0; p < J.parts; ++p) out[p] = &F.part[p][F.winner[p].load(std::memory_order_acquire) == 1 ? 1 : 0];
    for (const auto &o : F.merged) {
      const Fit::PartOut &O = *out[o.second >> 16];
      const int c = o.second & 0xffff;
      J.lines.insert(J.lines.end(), O.lines.begin() + 4 * (size_t)O.line_at[c], O.lines.begin() + 4 * (size_t)O.line_at[c + 1]);
    }
    hcounts[0] = total;
  } else if (hcounts[0] >= kChainCap) {
    set_last_error("plv_detect_lines: more than %d edge chains", kChainCap);
    return PLV_E_CAPACITY;
  }
  for (int c = 0; !by_parts && c < hcounts[0]; ++c) {
    const float4 *seg = F.segs.data() + J.hc[c].slot;
    for (int q = 0; q < F.seg_n[c]; ++q) {
      const float4 &sg = seg[q];
      const float x1 = sg.x * 2, y1 = sg.y * 2, x2 = sg.z * 2, y2 = sg.w * 2;  // REF :218-220
      const float l2 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
      if (!(l2 > J.thr2)) continue;  // FilterShortLines(lines0, 40)   REF :232, :435-448
      J.lines.insert(J.lines.end(), {x1, y1, x2, y2});
    }
  }
  if (plv::host_phases().on) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    plv::host_phases().add("detect host stage: the worker's own parts (or walk)", us(T1, T2));
    plv::host_phases().add("detect host stage: job closed (helpers inside a part)", us(T2, T2b));
    plv::host_phases().add("detect host stage: parts run a second time (count)", second_runs);
    plv::host_phases().add("detect host stage: chains merged, segments out", us(T2b, std::chrono::steady_clock::now()));
  }
  if (timing) {
    auto T3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    size_t edges = 0;
    for (size_t i = 0; i < (size_t)J.w * J.h; ++i) edges += J.hmap[i] == 2;
    fprintf(stderr, "walk %.1f us, rest of the fit %.1f us; %d chains, %d chain points, %zu edge pixels\n", us(T1, T2), us(T2, T3), hcounts[0],
            hcounts[2], edges);
    if (by_parts) {
      fprintf(stderr, "  parts (thread: start + build + walk + fit us, chain points):");
      for (int p = 0; p < J.parts; ++p) {
        const Fit::PartOut &O = F.part[p][F.winner[p].load() == 1 ? 1 : 0];
        fprintf(stderr, " %d(t%d%s: %.0f + %.0f + %.0f + %.0f, %d)", p + 1, O.slot, F.winner[p].load() == 1 ? " second run" : "", O.us_start, O.us_build, O.us_walk, O.us_fit, O.pixels);
      }
      fprintf(stderr, "\n");
    }
  }
  return PLV_OK;
}

struct Assign {
  std::vector<int> kept;
  std::vector<int> rel_ptr{0}, pos_ptr{0};
  std::vector<uint64_t> rel_id;
  std::vector<double> rel_dist;
  std::vector<float> pos;
};

inline void assign_points(const float *lines, int nl, const float *pts, const uint64_t *ids, int np, Assign &A, float assign_px = 5.0f) {
  A = Assign();
  // A point is assigned to a line when it lies inside the reference's box (REF :753-764, which reads (x1, y1, x2, y2) as
  // (lx1, lx2, ly1, ly2): kept as is) AND within assign_px of the segment.  The box is so wide that most points pass it (~47 000
  // tests per frame for 190 lines x 250 points, 60-70 us), but only points within assign_px of the segment's TRUE bounding box can
  // pass the distance test: the points are binned once into 32-pixel cells and a line only looks at the cells its true box (grown
  // by the threshold and half a pixel for rounding) touches — a few points instead of all of them, in point order, through the same
  // two tests.  Same assignments, ~10 x less work.
  constexpr int kCell = 32;
  float xmax = 0.f, ymax = 0.f;
  for (int j = 0; j < np; ++j) xmax = std::max(xmax, pts[2 * j]), ymax = std::max(ymax, pts[2 * j + 1]);
  const int gx = std::max(1, (int)(xmax / kCell) + 1), gy = std::max(1, (int)(ymax / kCell) + 1);
  auto cell_of = [&](float v, int g) { return std::min(std::max((int)std::floor(v * (1.0f / kCell)), 0), g - 1); };  // (a power of two: exact)
  // (the scratch of the binning lives with the thread: a frame's assignment allocates nothing but its result)
  static thread_local std::vector<int> cstart, cidx, fill, pcell;
  cstart.assign((size_t)gx * gy + 1, 0), cidx.resize((size_t)std::max(np, 1)), pcell.resize((size_t)std::max(np, 1));
  for (int j = 0; j < np; ++j) ++cstart[(size_t)(pcell[j] = cell_of(pts[2 * j + 1], gy) * gx + cell_of(pts[2 * j], gx)) + 1];
  for (size_t c = 0; c < (size_t)gx * gy; ++c) cstart[c + 1] += cstart[c];
  fill.assign(cstart.begin(), cstart.end() - 1);
  for (int j = 0; j < np; ++j) cidx[fill[pcell[j]]++] = j;  // ascending j within a cell
  constexpr int kCand = 64;
  int cand[kCand];
  std::vector<int> cand_more;
  std::vector<std::pair<int, double>> on;
  const float grow = assign_px + 0.5f;
  for (int i = 0; i < nl; ++i) {
    const float *ln = lines + 4 * i;
    const float lx1 = ln[0], lx2 = ln[1], ly1 = ln[2], ly2 = ln[3];  // (sic)
    const float min_lx = std::min(lx1, lx2), max_lx = std::max(lx1, lx2), min_ly = std::min(ly1, ly2), max_ly = std::max(ly1, ly2);
    const float tx0 = std::min(ln[0], ln[2]) - grow, tx1 = std::max(ln[0], ln[2]) + grow, ty0 = std::min(ln[1], ln[3]) - grow, ty1 = std::max(ln[1], ln[3]) + grow;
    // (no point of the true box can pass the reference's box when the two do not meet: nothing to look up)
    if (tx1 < min_lx || tx0 > max_lx || ty1 < min_ly || ty0 > max_ly) continue;
    const int cx0 = cell_of(std::max(tx0, min_lx), gx), cx1 = cell_of(std::min(tx1, max_lx), gx), cy0 = cell_of(std::max(ty0, min_ly), gy),
              cy1 = cell_of(std::min(ty1, max_ly), gy);
    int nc = 0;
    cand_more.clear();
    for (int cy = cy0; cy <= cy1; ++cy) {
      const int q0 = cstart[(size_t)cy * gx + cx0], q1 = cstart[(size_t)cy * gx + cx1 + 1];  // (the cells of a row are contiguous)
      for (int q = q0; q < q1; ++q) {
        if (nc < kCand)
          cand[nc++] = cidx[q];
        else
          cand_more.push_back(cidx[q]);
      }
    }
    if (nc == 0) continue;
    const int *cp = cand;
    if (!cand_more.empty()) {
      cand_more.insert(cand_more.end(), cand, cand + nc);
      cp = cand_more.data(), nc = (int)cand_more.size();
      std::sort(cand_more.begin(), cand_more.end());
    } else {
      for (int a = 1; a < nc; ++a) {  // point order, as the reference's loop meets them (a handful: insertion sort)
        const int v = cand[a];
        int b = a - 1;
        for (; b >= 0 && cand[b] > v; --b) cand[b + 1] = cand[b];
        cand[b + 1] = v;
      }
    }
    on.clear();
    const size_t first_pos = A.pos.size();
    for (int q = 0; q < nc; ++q) {
      const int j = cp[q];
      const float x = pts[2 * j], y = pts[2 * j + 1];
      if (!((x >= min_lx) & (x <= max_lx) & (y >= min_ly) & (y <= max_ly))) continue;
      const float d = point_line_distance(ln, x, y);
      if (d > assign_px) continue;
      on.emplace_back((int)ids[j], (double)d);
      A.pos.push_back(x);
      A.pos.push_back(y);
    }
    if (A.pos.size() == first_pos) continue;  // lines without a point are dropped (REF :784-789)
    A.kept.push_back(i);
    // (the reference keeps the relations in a std::map<int, double>: ascending ids, a later point with the same id replaces the earlier)
    std::stable_sort(on.begin(), on.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
    for (size_t q = 0; q < on.size(); ++q) {
      if (q + 1 < on.size() && on[q + 1].first == on[q].first) continue;
      A.rel_id.push_back((uint64_t)on[q].first);
      A.rel_dist.push_back(on[q].second);
    }
    A.rel_ptr.push_back((int)A.rel_id.size());
    A.pos_ptr.push_back((int)A.pos.size() / 2);
  }
}

// A job for the helpers is on its way (the edge maps of a frame have been launched: the host stage follows within ~0.1 ms): helpers
// asleep on their condition variable are woken now and poll (wait_polling's budget) instead of being woken when the work is there —
// a wake-up costs 30-100 us, which used to be the tail of every frame's detection ("rest of the fit")
inline void prewake_helpers(HostStage *T) {
  Fit &F = T->fit;
  {
    std::lock_guard<std::mutex> lk(F.m);
    ++F.prewake;
  }
  F.cv.notify_all();
}

// fn(slot) for slot 0 .. nhelpers, each exactly once: slot 0 on the calling thread, slot s on helper thread s - 1 of the stage — or on
// the calling thread when that helper has not started it by the time the caller gets there (a late or descheduled helper is not
// waited for: Fit::slot_claim, close_job).  Returns when every slot has returned.
inline void run_on_helpers(HostStage *T, int nhelpers, const std::function<void(int)> &fn) {
  Fit &F = T->fit;
  nhelpers = std::max(0, std::min(nhelpers, (int)Fit::kThreads));
  if (nhelpers == 0) {
    fn(0);
    return;
  }
  {
    int gen_now;
    {
      std::lock_guard<std::mutex> lk(F.m);
      gen_now = F.gen;
    }
    for (int i = 0; i < nhelpers; ++i)
      if (!F.th[i].joinable()) F.th[i] = std::thread(fit_worker, T, i, gen_now);
  }
  for (int sl = 1; sl <= nhelpers; ++sl) F.slot_claim[sl].store(0, std::memory_order_relaxed);  // (nobody is inside a generic job now: the last one was closed)
  const int gen = post_job(F, nullptr, nhelpers, false, &fn);
  fn(0);
  if (plv::knob(plv::PLV_KNOB_WAIT_ALL_HELPERS)) {
    wait_reports(F, nhelpers, gen);
    F.closed_gen.store(gen);
    return;
  }
  for (int sl = 1; sl <= nhelpers; ++sl) {
    int unclaimed = 0;
    if (F.slot_claim[sl].compare_exchange_strong(unclaimed, 1)) fn(sl);
  }
  close_job(F, nhelpers, gen);  // (fn lives with the caller until every thread that may call it has reported)
}

// assign_points with the lines split into contiguous ranges over the stage's threads: a line's assignment depends on nothing but the
// line and the points, and the ranges' results are joined in line order — the same Assign as the serial call.
inline void assign_points_parallel(HostStage *T, int nhelpers, const float *lines, int nl, const float *pts, const uint64_t *ids, int np, Assign &A,
                                   float assign_px = 5.0f) {
  const int nt = plv::knob(plv::PLV_KNOB_ASSIGN_ONE_THREAD) ? 1 : std::max(1, std::min(nhelpers + 1, nl / 50));
  if (nt == 1) {
    assign_points(lines, nl, pts, ids, np, A, assign_px);
    return;
  }
  std::vector<Assign> part(nt);
  const bool timing = plv::knob(plv::PLV_KNOB_LINE_TIMING);
  const auto ta = std::chrono::steady_clock::now();
  float t_start[16] = {0}, t_end[16] = {0};
  run_on_helpers(T, nt - 1, [&](int slot) {
    if (timing) t_start[slot] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - ta).count();
    const int i0 = (int)((long long)nl * slot / nt), i1 = (int)((long long)nl * (slot + 1) / nt);
    assign_points(lines + 4 * (size_t)i0, i1 - i0, pts, ids, np, part[slot], assign_px);
    for (int &k : part[slot].kept) k += i0;
    if (timing) t_end[slot] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - ta).count();
  });
  if (timing) {
    fprintf(stderr, "assign over %d threads (start-end us):", nt);
    for (int i = 0; i < nt; ++i) fprintf(stderr, " %.1f-%.1f", t_start[i], t_end[i]);
    fprintf(stderr, "; all joined %.1f\n", std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - ta).count());
  }
  A = Assign();
  for (const Assign &P : part) {
    A.kept.insert(A.kept.end(), P.kept.begin(), P.kept.end());
    const int r0 = (int)A.rel_id.size(), p0 = (int)A.pos.size() / 2;
    for (size_t q = 1; q < P.rel_ptr.size(); ++q) A.rel_ptr.push_back(r0 + P.rel_ptr[q]);
    for (size_t q = 1; q < P.pos_ptr.size(); ++q) A.pos_ptr.push_back(p0 + P.pos_ptr[q]);
    A.rel_id.insert(A.rel_id.end(), P.rel_id.begin(), P.rel_id.end());
    A.rel_dist.insert(A.rel_dist.end(), P.rel_dist.begin(), P.rel_dist.end());
    A.pos.insert(A.pos.end(), P.pos.begin(), P.pos.end());
  }
}

// TrackLSD's line matching (REF: TrackLSD.cpp LineMatch as the oracle restates it): for a new line i the reference walks every line j
// of the last frame and that line's points; the first shared point accepts j when j's midpoint lies within 6 px of the new segment,
// the second accepts it outright, and a later j overwrites an earlier one.  So match[i] = the LAST j that shares two points with i, or
// one point and the midpoint test.  Only lines that share a point can match: an index from point id to the last frame's lines
// replaces the walk over all pairs (90 x 90 pairs x their points: 30-40 us per frame) by a look-up per point of the new line.
inline void match_lines(const float *lines_new, int n_new, const int *rp_new, const uint64_t *ri_new, const float *lines_last, int n_last,
                 const int *rp_last, const uint64_t *ri_last, int *match) {
  std::fill(match, match + n_new, -1);
  if (n_new == 0 || n_last == 0) return;
  static thread_local std::vector<std::pair<uint64_t, int>> by_id;  // (point id, line of the last frame), sorted
  by_id.clear();
  for (int j = 0; j < n_last; ++j)
    for (int q = rp_last[j]; q < rp_last[j + 1]; ++q) by_id.emplace_back(ri_last[q], j);
  std::sort(by_id.begin(), by_id.end());
  by_id.erase(std::unique(by_id.begin(), by_id.end()), by_id.end());  // (a point listed twice on a line counts once per listing in the
                                                                      //  reference too, but the lists are map keys: no duplicates)
  static thread_local std::vector<std::pair<int, int>> hit;  // (line j, shared points)
  for (int i = 0; i < n_new; ++i) {
    hit.clear();
    for (int p = rp_new[i]; p < rp_new[i + 1]; ++p) {
      if (p > rp_new[i] && ri_new[p] == ri_new[p - 1]) continue;
      auto lo = std::lower_bound(by_id.begin(), by_id.end(), std::make_pair(ri_new[p], -1));
      for (; lo != by_id.end() && lo->first == ri_new[p]; ++lo) {
        bool seen = false;
        for (auto &h : hit)
          if (h.first == lo->second) {
            ++h.second, seen = true;
            break;
          }
        if (!seen) hit.emplace_back(lo->second, 1);
      }
    }
    for (const auto &h : hit) {
      const int j = h.first;
      if (j < match[i]) continue;
      bool ok = h.second >= 2;
      if (!ok) {
        const float mx = (lines_last[4 * j] + lines_last[4 * j + 2]) / 2, my = (lines_last[4 * j + 1] + lines_last[4 * j + 3]) / 2;
        ok = point_line_distance(lines_new + 4 * i, mx, my) <= 6;
      }
      if (ok) match[i] = j;
    }
  }
}

// match_lines with the new lines split into contiguous ranges over the stage's threads (a new line's match depends on nothing but
// that line and the last frame's lines)
inline void match_lines_parallel(HostStage *T, int nhelpers, const float *lines_new, int n_new, const int *rp_new, const uint64_t *ri_new,
                                 const float *lines_last, int n_last, const int *rp_last, const uint64_t *ri_last, int *match) {
  const int nt = std::max(1, std::min(nhelpers + 1, n_new / 2000));  // (the indexed match takes a few us: one thread)
  if (nt == 1 || n_last == 0) {
    match_lines(lines_new, n_new, rp_new, ri_new, lines_last, n_last, rp_last, ri_last, match);
    return;
  }
  run_on_helpers(T, nt - 1, [&](int slot) {
    const int i0 = (int)((long long)n_new * slot / nt), i1 = (int)((long long)n_new * (slot + 1) / nt);
    // (rp_new + i0 still indexes ri_new from its start: the offsets are absolute)
    match_lines(lines_new + 4 * (size_t)i0, i1 - i0, rp_new + i0, ri_new, lines_last, n_last, rp_last, ri_last, match + i0);
  });
}

inline bool line_class(const float *line, const double *vp) {
  const double sx = line[0], sy = line[1], ex = line[2], ey = line[3];
  const double mx = (sx + ex) / 2, my = (sy + ey) / 2;
  // line through the midpoint and the vanishing point, homogeneous
  const double a = my - vp[1], b = vp[0] - mx, c = mx * vp[1] - my * vp[0];
  const double ds = a * sx + b * sy + c, de = a * ex + b * ey + c;
  const double dis_error = std::abs((std::abs(std::sqrt(ds * ds)) + std::abs(std::sqrt(de * de))) / (2 * std::sqrt(a * a + b * b)));
  const double angle1 = (double)(std::atan(line[1] - line[3]) / (line[0] - line[2]));  // (sic) atan(dy)/dx in float
  const double angle2 = std::atan(my - vp[1]) / (mx - vp[0]);
  return dis_error <= 5.0 && std::abs(angle1 - angle2) <= 0.35;
}

}  // namespace linehost
}  // namespace plv
