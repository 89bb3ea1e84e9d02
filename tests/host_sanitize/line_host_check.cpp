// Sanitizer driver of the line front-end's host stage (pl-viwo_amd/csrc/line_host.hpp): the chain walk publishing chains to two
// fitter threads, the hand-over protocol between them, and the assignment / matching logic — compiled with g++ -fsanitize=thread
// or -fsanitize=address,undefined by tests/test_host_sanitize.py and fed recorded edge maps.  No device is touched.
//   usage: line_host_check <maps.bin> <repeats>
//   maps.bin: int32 w, h, n; then n x { w*h bytes Canny map (2 = edge), w*h bytes half-resolution image }
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../pl-viwo_amd/csrc/line_host.hpp"

namespace plv {
void set_last_error(const char *, ...) {}
}  // namespace plv

using namespace plv;
using namespace plv::linehost;

// AssignPointToLines and LineMatch written out plainly (every line against every point / every pair of lines and their points, as
// the reference's loops run and the oracle restates them): the checkers of line_host.hpp's binned assignment and indexed matching.
static void assign_points_plain(const float *lines, int nl, const float *pts, const uint64_t *ids, int np, Assign &A, float assign_px = 5.0f) {
  A = Assign();
  for (int i = 0; i < nl; ++i) {
    const float *ln = lines + 4 * i;
    const float lx1 = ln[0], lx2 = ln[1], ly1 = ln[2], ly2 = ln[3];  // (sic: the reference reads the segment this way)
    const float min_lx = std::min(lx1, lx2), max_lx = std::max(lx1, lx2), min_ly = std::min(ly1, ly2), max_ly = std::max(ly1, ly2);
    std::map<int, double> on;
    const size_t first_pos = A.pos.size();
    for (int j = 0; j < np; ++j) {
      const float x = pts[2 * j], y = pts[2 * j + 1];
      if (!(x >= min_lx && x <= max_lx && y >= min_ly && y <= max_ly)) continue;
      const float d = point_line_distance(ln, x, y);
      if (d > assign_px) continue;
      on[(int)ids[j]] = d;
      A.pos.push_back(x);
      A.pos.push_back(y);
    }
    if (A.pos.size() == first_pos) continue;
    A.kept.push_back(i);
    for (const auto &kv : on) A.rel_id.push_back((uint64_t)kv.first), A.rel_dist.push_back(kv.second);
    A.rel_ptr.push_back((int)A.rel_id.size());
    A.pos_ptr.push_back((int)A.pos.size() / 2);
  }
}
static void match_lines_plain(const float *lines_new, int n_new, const int *rp_new, const uint64_t *ri_new, const float *lines_last, int n_last,
                              const int *rp_last, const uint64_t *ri_last, int *match) {
  std::fill(match, match + n_new, -1);
  for (int i = 0; i < n_new; ++i) {
    if (rp_new[i + 1] == rp_new[i]) continue;
    for (int j = 0; j < n_last; ++j) {
      int shared = 0;
      for (int q = rp_last[j]; q < rp_last[j + 1]; ++q) {
        if (!std::binary_search(ri_new + rp_new[i], ri_new + rp_new[i + 1], ri_last[q])) continue;
        ++shared;
        if (shared >= 2) {
          match[i] = j;
          break;
        }
        const float mx = (lines_last[4 * j] + lines_last[4 * j + 2]) / 2, my = (lines_last[4 * j + 1] + lines_last[4 * j + 3]) / 2;
        if (point_line_distance(lines_new + 4 * i, mx, my) <= 6) {
          match[i] = j;
          break;
        }
      }
    }
  }
}

// The chain walk written out plainly (eight byte tests per step, in the detector's order: REF FastLineDetector's chain loop as the
// oracle restates it): the checker of line_host.hpp's walk_chains, which reads the neighbourhood as three words and a bit mask.
static void walk_chains_plain(const uint8_t *map, int w, int h, int length_threshold, std::vector<int2> &pts, std::vector<FldChain> &chains) {
  static const int dx[8] = {1, 0, -1, -1, -1, 0, 1, 1}, dy[8] = {1, 1, 1, 0, -1, -1, -1, 0};
  std::vector<uint8_t> m(map, map + (size_t)w * h);
  auto edge = [&](int x, int y) { return x >= 0 && y >= 0 && x < w && y < h && m[(size_t)y * w + x] == 2; };
  int n_slot = 0;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      if (m[(size_t)r * w + c] != 2) continue;
      const size_t start = pts.size();
      int x = c, y = r;
      pts.push_back(make_int2(x, y));
      m[(size_t)y * w + x] = 1;
      float direction = 0.0f;
      for (int step = 0;; ++step) {
        int pick = -1;
        float best = 7.0f;
        for (int i = 0; i < 8; ++i) {
          if (!edge(x + dx[i], y + dy[i])) continue;
          if (step == 0) {
            pick = i;
            break;
          }
          const float curr = i > 4 ? (float)(i - 8) : (float)i;
          float diff = std::fabs(curr - direction);
          if (diff > 4.0f) diff = 8.0f - diff;
          if (diff <= best) best = diff, pick = i;
        }
        if (pick < 0 || (step > 0 && !(best < 2.0f))) break;
        const int cdir = pick > 4 ? pick - 8 : pick;
        direction = step == 0 ? (float)cdir : (direction * (float)step + (float)cdir) / (float)(step + 1);
        x += dx[pick], y += dy[pick];
        pts.push_back(make_int2(x, y));
        m[(size_t)y * w + x] = 1;
      }
      const int len = (int)(pts.size() - start);
      if (len >= length_threshold + 1 && (int)chains.size() < kChainCap) {
        chains.push_back(FldChain{(int)start, len, n_slot});
        n_slot += len / length_threshold + 1;
      } else {
        pts.resize(start);
      }
    }
}

// component labels of the edge pixels as ccl_merge_kernel / ccl_flatten_kernel produce them on the device (8-connectivity, root = the
// smallest pixel index of the component): 0 = not an edge, else 1 + hash(root) % parts
static std::vector<uint8_t> component_labels(const uint8_t *map, int w, int h, int parts) {
  std::vector<int> root((size_t)w * h, -1), stack;
  for (int s = 0; s < w * h; ++s) {
    if (map[s] != 2 || root[s] >= 0) continue;
    root[s] = s;
    stack.assign(1, s);
    while (!stack.empty()) {
      const int p = stack.back();
      stack.pop_back();
      const int y = p / w, x = p - y * w;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int qx = x + dx, qy = y + dy;
          if (qx < 0 || qy < 0 || qx >= w || qy >= h) continue;
          const int q = qy * w + qx;
          if (map[q] == 2 && root[q] < 0) root[q] = s, stack.push_back(q);
        }
    }
  }
  std::vector<uint8_t> lab((size_t)w * h, 0);
  for (int s = 0; s < w * h; ++s)
    if (root[s] >= 0) lab[s] = (uint8_t)(1 + ((unsigned)root[s] * 2654435761u >> 8) % (unsigned)parts);
  return lab;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 2;
  const int w = hdr[0], h = hdr[1], n = hdr[2], reps = atoi(argv[2]);
  const size_t npix = (size_t)w * h;
  std::vector<std::vector<uint8_t>> maps(n, std::vector<uint8_t>(npix)), halves(n, std::vector<uint8_t>(npix));
  for (int i = 0; i < n; ++i)
    if (fread(maps[i].data(), 1, npix, f) != npix || fread(halves[i].data(), 1, npix, f) != npix) return 2;
  fclose(f);
  HostStage stage;
  std::vector<int2> pts(npix);
  std::vector<FldChain> chains(kChainCap);
  long total_lines = 0, total_kept = 0, total_matched = 0;
  std::mt19937 rng(7);
  std::vector<float> last_lines;
  Assign last;
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < n; ++i) {
      Job J;
      J.w = w, J.h = h, J.length_threshold = 20, J.distance_threshold = 1.414213562f, J.thr2 = 1600.0f;
      J.hmap = maps[i].data(), J.hhalf = halves[i].data(), J.hpts = pts.data(), J.hc = chains.data();
      // every other pass with the components labelled: the detection split into 16, 3 or 1 parts over the walking thread and the
      // helpers (host_extract -> detect_part) must give the same segments in the same order
      // (the second half of the passes with the parts' pixel lists next to the labels — what ccl_flatten_kernel leaves: a part is then
      // staged and seeded from its own pixels on the thread's persistent map)
      std::vector<uint8_t> lab, sorted;
      std::vector<unsigned short> bins;
      if ((r + i) % 2 == 1) {
        J.parts = r % 3 == 0 ? plv::linehost::Fit::kParts : (r % 3 == 1 ? 3 : 1);
        lab = component_labels(maps[i].data(), w, h, J.parts);
        J.hlab = lab.data();
        if (r >= 3) {
          plv::linehost::lists_from_labels(lab.data(), (int)npix, J.parts, sorted, bins);
          J.hsorted = sorted.data(), J.hbins = bins.data();
        }
      }
      // (the last two passes with helper threads that fall asleep at random — when they pick a job up, when they start a part: jobs
      // are closed without them, parts run a second time, late helpers find closed jobs; same segments, same assignment)
      stage.fit.chaos_us.store(r >= 4 ? 20000 : 0);
      if (host_extract(&stage, J, false) != 0) return 3;
      // the same detection on this thread alone: the threaded one must give the same segments in the same order
      std::vector<int2> p2(npix);
      std::vector<FldChain> c2(kChainCap);
      std::vector<uint8_t> pad;
      int counts[4] = {0, 0, 0, 0};
      walk_chains(maps[i].data(), w, h, 20, p2.data(), c2.data(), kChainCap, counts, pad);
      if (r == 0) {  // the word-and-mask walk against the plain one: same chains, same points, same order
        std::vector<int2> p3;
        std::vector<FldChain> c3;
        walk_chains_plain(maps[i].data(), w, h, 20, p3, c3);
        bool same = (int)c3.size() == counts[0] && (int)p3.size() == counts[2];
        for (int c = 0; same && c < counts[0]; ++c) same = c3[c].start == c2[c].start && c3[c].len == c2[c].len && c3[c].slot == c2[c].slot;
        for (int q = 0; same && q < counts[2]; ++q) same = p3[q].x == p2[q].x && p3[q].y == p2[q].y;
        if (!same) {
          fprintf(stderr, "frame %d: walk_chains differs from the plain walk (%d vs %zu chains, %d vs %zu points)\n", i, counts[0], c3.size(), counts[2], p3.size());
          return 5;
        }
        if (counts[0] < 20) {
          fprintf(stderr, "frame %d: only %d chains: the map does not exercise the walk\n", i, counts[0]);
          return 5;
        }
      }
      std::vector<float> ref;
      std::vector<float4> seg(npix / 20 + kChainCap);
      for (int c = 0; c < counts[0]; ++c) {
        const int ns = fit_chain(halves[i].data(), w, h, 20, 1.414213562f, p2.data() + c2[c].start, c2[c].len, seg.data());
        for (int q = 0; q < ns; ++q) {
          const float x1 = seg[q].x * 2, y1 = seg[q].y * 2, x2 = seg[q].z * 2, y2 = seg[q].w * 2;
          if (!((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) > 1600.0f)) continue;
          ref.insert(ref.end(), {x1, y1, x2, y2});
        }
      }
      if (ref != J.lines) {
        fprintf(stderr, "frame %d: threaded detection differs from the serial one (%zu vs %zu values)\n", i, J.lines.size(), ref.size());
        return 4;
      }
      total_lines += (long)J.lines.size() / 4;
      // TrackLSD's assignment + matching on points scattered along the segments
      const int nl = (int)J.lines.size() / 4;
      std::vector<float> ptsf;
      std::vector<uint64_t> ids;
      std::uniform_real_distribution<float> u(0.f, 1.f);
      for (int q = 0; q < nl; q += 2) {
        const float s = u(rng);
        ptsf.push_back(J.lines[4 * q] + s * (J.lines[4 * q + 2] - J.lines[4 * q]) + 2 * u(rng));
        ptsf.push_back(J.lines[4 * q + 1] + s * (J.lines[4 * q + 3] - J.lines[4 * q + 1]) + 2 * u(rng));
        ids.push_back(1000 + q);
      }
      // ... and points that live on from map to map (ids that meet again: what the matching works on), some of them piled up
      // at the segments' first end points (several points per line, the same point on several lines)
      for (int q = 0; q < 260; ++q) {
        const int l = (q * 7) % std::max(nl, 1);
        const bool near = nl > 0 && q % 3 != 0;
        ptsf.push_back(near ? J.lines[4 * l] + 6 * u(rng) - 3 : 2.f * w * u(rng));
        ptsf.push_back(near ? J.lines[4 * l + 1] + 6 * u(rng) - 3 : 2.f * h * u(rng));
        ids.push_back(50000 + q);
      }
      // ... and two or three points on every third segment whose ids go with the segment's index: lines of consecutive maps that
      // share two points (the matching's first rule), or one (its second)
      for (int l = 0; l < nl; l += 3)
        for (int k = 0; k < 2 + (l % 2); ++k) {
          if ((l + k) % 5 == 0) continue;
          const float s = 0.2f + 0.3f * k;
          ptsf.push_back(J.lines[4 * l] + s * (J.lines[4 * l + 2] - J.lines[4 * l]) + u(rng));
          ptsf.push_back(J.lines[4 * l + 1] + s * (J.lines[4 * l + 3] - J.lines[4 * l + 1]) + u(rng));
          ids.push_back(60000 + 4 * (uint64_t)l + k);
        }
      Assign A, Ap, Aplain;
      assign_points(J.lines.data(), nl, ptsf.data(), ids.data(), (int)ids.size(), A);
      assign_points_plain(J.lines.data(), nl, ptsf.data(), ids.data(), (int)ids.size(), Aplain);
      if (Aplain.kept != A.kept || Aplain.rel_ptr != A.rel_ptr || Aplain.rel_id != A.rel_id || Aplain.rel_dist != A.rel_dist || Aplain.pos_ptr != A.pos_ptr ||
          Aplain.pos != A.pos) {
        fprintf(stderr, "frame %d: the binned assignment differs from the plain one (%zu vs %zu lines kept)\n", i, A.kept.size(), Aplain.kept.size());
        return 8;
      }
      // the same assignment with the lines split over the stage's threads: the same lists
      assign_points_parallel(&stage, 3, J.lines.data(), nl, ptsf.data(), ids.data(), (int)ids.size(), Ap);
      if (Ap.kept != A.kept || Ap.rel_ptr != A.rel_ptr || Ap.rel_id != A.rel_id || Ap.rel_dist != A.rel_dist || Ap.pos_ptr != A.pos_ptr || Ap.pos != A.pos) {
        fprintf(stderr, "frame %d: the parallel assignment differs from the serial one\n", i);
        return 6;
      }
      std::vector<float> kept_lines;
      for (int q : A.kept) kept_lines.insert(kept_lines.end(), J.lines.begin() + 4 * q, J.lines.begin() + 4 * q + 4);
      if (!last_lines.empty() && !A.kept.empty()) {
        std::vector<int> match(A.kept.size());
        match_lines(kept_lines.data(), (int)A.kept.size(), A.rel_ptr.data(), A.rel_id.data(), last_lines.data(), (int)last_lines.size() / 4,
                    last.rel_ptr.data(), last.rel_id.data(), match.data());
        std::vector<int> match_p(A.kept.size());
        match_lines_parallel(&stage, 3, kept_lines.data(), (int)A.kept.size(), A.rel_ptr.data(), A.rel_id.data(), last_lines.data(),
                             (int)last_lines.size() / 4, last.rel_ptr.data(), last.rel_id.data(), match_p.data());
        if (match_p != match) {
          fprintf(stderr, "frame %d: the parallel matching differs from the serial one\n", i);
          return 7;
        }
        std::vector<int> match_plain(A.kept.size());
        match_lines_plain(kept_lines.data(), (int)A.kept.size(), A.rel_ptr.data(), A.rel_id.data(), last_lines.data(), (int)last_lines.size() / 4,
                          last.rel_ptr.data(), last.rel_id.data(), match_plain.data());
        if (match_plain != match) {
          fprintf(stderr, "frame %d: the indexed matching differs from the plain one\n", i);
          return 9;
        }
        for (int m : match) total_matched += m >= 0;
      }
      total_kept += (long)A.kept.size();
      last_lines = kept_lines;
      last = A;
      // (the job and its label lists live in this block; the library keeps its own in place until the next detection is launched and
      // quiesces there)
      plv::linehost::quiesce_helpers(stage.fit);
    }
  printf("ok: %d maps x %d repeats, %ld segments, %ld kept by the assignment, %ld matched to a line of the map before; %ld parts run a second time, %ld of those runs counted\n",
         n, reps, total_lines, total_kept, total_matched, stage.fit.second_runs.load(), stage.fit.second_run_wins.load());
  return 0;
}
