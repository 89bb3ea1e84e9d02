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
  long total_lines = 0, total_kept = 0;
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
      std::vector<uint8_t> lab;
      if ((r + i) % 2 == 1) {
        J.parts = r % 3 == 0 ? plv::linehost::Fit::kParts : (r % 3 == 1 ? 3 : 1);
        lab = component_labels(maps[i].data(), w, h, J.parts);
        J.hlab = lab.data();
      }
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
      Assign A, Ap;
      assign_points(J.lines.data(), nl, ptsf.data(), ids.data(), (int)ids.size(), A);
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
      }
      total_kept += (long)A.kept.size();
      last_lines = kept_lines;
      last = A;
    }
  printf("ok: %d maps x %d repeats, %ld segments, %ld kept by the assignment\n", n, reps, total_lines, total_kept);
  return 0;
}
